"""`from networks.net_factory_3d import net_factory_3d` (train_inherent_consistent_unet_3D_BraTS.py:21) -> icl_amd."""
from icl_amd.networks.net_factory_3d import *  # noqa: F401,F403
from icl_amd.networks.net_factory_3d import net_factory_3d  # noqa: F401
