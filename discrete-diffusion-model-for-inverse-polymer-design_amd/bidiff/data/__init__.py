from .bucket_sampler import LengthBucketBatchSampler
from .token_store import TokenStore
from .loader import DeviceBatchLoader

__all__ = ["LengthBucketBatchSampler", "TokenStore", "DeviceBatchLoader"]
