from .imagenet_group import (DataManager, DataManager_test, GroupBatchSampler, GroupBatchSamplerTest, GroupDataset,
                             GroupLoader, ImageDataset)

__all__ = ["DataManager", "DataManager_test", "GroupBatchSampler", "GroupBatchSamplerTest", "GroupDataset", "GroupLoader",
           "ImageDataset"]
