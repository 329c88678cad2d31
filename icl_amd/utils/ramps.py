"""Consistency-weight ramps imported by the AMOS and 2-D trainers (``from utils import losses, ramps``;
/root/reference/code/utils/ramps.py:16-41).  ``train_inherent_consistent_unet_3D_AMOS22.py:163-165`` evaluates
``args.consistency * ramps.sigmoid_rampup(epoch, args.consistency_rampup)`` (the result is unused in the ICL loss, :224-230)."""
import numpy as np


def sigmoid_rampup(current, rampup_length):
    """exp(-5 (1 - t)^2) with t = clip(current / rampup_length, 0, 1); 1 when the ramp has no length."""
    if rampup_length == 0:
        return 1.0
    t = np.clip(current, 0.0, rampup_length) / rampup_length
    return float(np.exp(-5.0 * (1.0 - t) ** 2))


def linear_rampup(current, rampup_length):
    assert current >= 0 and rampup_length >= 0
    return 1.0 if current >= rampup_length else current / rampup_length


def cosine_rampdown(current, rampdown_length):
    assert 0 <= current <= rampdown_length
    return float(.5 * (np.cos(np.pi * current / rampdown_length) + 1))
