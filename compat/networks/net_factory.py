"""`from networks.net_factory import net_factory` (train_inherent_consistent_unet_2D.py:18) -> icl_amd."""
from icl_amd.networks.net_factory import *  # noqa: F401,F403
from icl_amd.networks.net_factory import net_factory  # noqa: F401
