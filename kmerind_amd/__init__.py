"""kmerind_amd -- MI355X-native k-mer index core (host-side Python plumbing over the C ABI).

The product is libkmerind_hip.so (hand-written HIP for gfx950, include/kmerind_hip.h) and the
C++ facade in include/kmerind/. This package is the thin layer bench.py and the tests use:
device buffers, torch.distributed exchange, synthetic inputs."""
from . import _lib
from .core import Context, CountIndex, DeBruijnNodes, PositionIndex, make_config, synth_fastq  # noqa: F401
