from .stein_sampler import SteinSampler

__all__ = ["SteinSampler"]
