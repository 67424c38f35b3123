"""apss -- MI355X-native all-pairs sparse-vector similarity (hot path of mcgill-cpslab/all-pairs-similarity).

The compute path is the HIP library all-pairs-similarity_amd/csrc/libapss_hip.so reached through the C ABI of
include/apss.h; there is NO CPU fallback: creating an index without the built library (or without a GPU)
raises.  `synth` (workload generator) is importable without the library.
"""
from . import synth  # noqa: F401

__all__ = ["synth"]
