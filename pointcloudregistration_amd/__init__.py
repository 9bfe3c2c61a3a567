"""pointcloudregistration_amd -- MI355X-native ICP fine-registration hot path.

The product is libpcr.so (hand-written HIP for gfx950 behind the C ABI of include/pcr.h) plus the
C++ plugin classes under host/ that mirror the reference's interfaces.  The Python modules here are
plumbing: `capi` (ctypes binding), `synth` (seeded clouds), `build` (hipcc driver), `dist`
(source-sharded multi-GPU loop over torch.distributed / RCCL).
"""
__version__ = "0.1.0"
