"""`from dataloaders.brats2019 import (BraTS2019, RandomCrop, RandomRotFlip, ToTensor, TwoStreamBatchSampler)`
(train_inherent_consistent_unet_3D_BraTS.py:17-19) -> icl_amd.dataloaders.brats2019."""
from icl_amd.dataloaders.brats2019 import (BraTS2019, CenterCrop, DeviceVolumeStore, OnDeviceAugment, RandomCrop,  # noqa: F401
                                           RandomRotFlip, ToTensor, TwoStreamBatchSampler)
