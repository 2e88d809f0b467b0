#!/bin/bash
timeout -k 10 900 python tools/exp/front_fuzz.py 120 2>&1 | tail -3
