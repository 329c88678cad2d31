"""`from utils import losses, ramps` (train_inherent_consistent_unet_3D_AMOS22.py) -> icl_amd.utils.ramps."""
from icl_amd.utils.ramps import cosine_rampdown, linear_rampup, sigmoid_rampup  # noqa: F401
