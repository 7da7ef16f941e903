"""motcpp_amd — MI355X-native association hot path of motcpp.

The compute lives in hand-written HIP behind a C ABI (include/motcpp_amd.h); this package is the thin
Python loader/binding used by tests and bench.py. Importing it never falls back to a CPU path.
"""
