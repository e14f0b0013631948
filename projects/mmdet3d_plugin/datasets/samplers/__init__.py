from .group_in_batch_sampler import GroupInBatchSampler

__all__ = ["GroupInBatchSampler"]
