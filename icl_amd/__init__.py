"""icl_amd — MI355X-native hot path of Inherent Consistent Learning (zhuye98/ICL).

HIP kernels (csrc/, C ABI in include/icl_hip.h) behind the reference's own Python interface:
``icl_amd.networks.net_factory_3d.net_factory_3d`` and ``icl_amd.utils.losses``.
"""
__version__ = "0.1.0"
