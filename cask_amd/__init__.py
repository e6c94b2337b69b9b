"""cask_amd -- MI355X-native SpMV engine behind the CASK operator surface.

The product is ``cask_amd/lib/libcask_hip.so`` (C ABI: ``include/cask_hip.h``)
plus the C++ drop-in headers under ``include/cask``.  This Python package is
plumbing: ``capi`` (ctypes binding), ``synth`` (seeded benchmark matrices) and
``dist`` (one-process-per-GPU row sharding over torch.distributed / RCCL).
"""
__version__ = "0.1.0"
