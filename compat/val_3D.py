"""`from val_3D import test_all_case_base` / `test_all_case_amos` (train_inherent_consistent_*_3D_*.py) -> icl_amd.val_3D."""
from icl_amd.val_3D import *  # noqa: F401,F403
from icl_amd.val_3D import (cal_metric, sliding_window_inference, test_all_case_amos, test_all_case_base,  # noqa: F401
                            test_single_case_base)
