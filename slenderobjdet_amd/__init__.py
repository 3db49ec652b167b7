"""slenderobjdet_amd — MI355X-native implementation of the SlenderObjDet data-parallel training hot path.

Host side: Python mirrors of the reference's ``slender_det`` interfaces (config / registries / build_model /
trainer) for the FCOS path.  Arithmetic: hand-written HIP kernels for gfx950 behind the C ABI of
``include/slender_hip.h`` (``libslender_hip.so``, bound in :mod:`slenderobjdet_amd._C`).
"""
__version__ = "0.1.0"
