#!/bin/bash
timeout -k 10 600 python -m pytest tests/test_gpu_debruijn.py -m gpu -x -q 2>&1 | tail -15
