from .adam_gradient_descent import AdamGradientDescent
from .adagrad_gradient_descent import AdagradGradientDescent

__all__ = ["AdamGradientDescent", "AdagradGradientDescent"]
