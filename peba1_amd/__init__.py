"""peba1_amd -- MI355X-native TFHE gate-bootstrapping engine behind the tfhe boots* C API.

The product is peba1_amd/libtfhe-hip.so (HIP kernels + C ABI, sources in
peba1_amd/csrc, headers in include/).  This package is the Python plumbing used
by tests and bench.py; it never evaluates a gate on the CPU.
"""
from . import lib  # noqa: F401
