"""`from val_2D import test_single_volume_ours` (train_inherent_consistent_unet_2D.py, ..._swinunet_2D.py) -> icl_amd.val_2D."""
from icl_amd.val_2D import calculate_metric_percase, test_single_volume, test_single_volume_ours  # noqa: F401
