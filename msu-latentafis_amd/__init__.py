"""MI355X-native latent-vs-gallery fingerprint matcher (hot path of prip-lab/MSU-LatentAFIS `matching/`).

The product is libafis_hip.so (hand-written HIP kernels behind the C ABI in include/afis_matcher.h) plus the
C++ `match` CLI.  This Python package is the host-side mirror used by tests and bench.py.
"""
