"""sfgwas_amd — MI355X-native implementation of SF-GWAS's per-party local linear-algebra hot path.

The product is sfgwas_amd/lib/libsfgwas_hip.so (C-ABI in include/sfgwas_hip.h, HIP sources in
sfgwas_amd/csrc/).  The Python modules here are test/bench plumbing around that library.
"""
from . import capi  # noqa: F401
