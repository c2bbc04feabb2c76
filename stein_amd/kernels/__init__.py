from .squared_exponential_kernel import SquaredExponentialKernel

__all__ = ["SquaredExponentialKernel"]
