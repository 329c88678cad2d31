"""`from utils import losses` (train_inherent_consistent_unet_3D_BraTS.py:22) -> icl_amd.utils.losses."""
from icl_amd.utils.losses import *  # noqa: F401,F403
from icl_amd.utils.losses import (AuxLoss, AuxLoss3D, DiceLoss, PseudoSoftLoss, PseudoSoftLoss3D, dice_loss1,  # noqa: F401
                                  softmax_dice_loss, softmax_mse_loss)
