from .compute_median import compute_median
from .converters import convert_array_to_dictionary, convert_dictionary_to_array

__all__ = ["compute_median", "convert_array_to_dictionary", "convert_dictionary_to_array"]
