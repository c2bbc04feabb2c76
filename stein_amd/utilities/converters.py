"""dict <-> packed [n, d] matrix, following stein/utilities/converters.py:4-89.

Columns are laid out by variable, variables ordered by their sorted ``name`` (:40);
``access_indices`` maps each variable to its (start, end) column range (:53).  Keys may
be any object with a ``.name`` (and, for unpacking, ``.get_shape().as_list()`` or a
``.shape``), or plain strings.  Works on NumPy arrays (returns float64 NumPy, as the
reference does) and on torch tensors (returns tensors on the same device; unpacking a
tensor yields zero-copy views of the packed matrix).
"""
import numpy as np
import torch


def _name(v):
    return v if isinstance(v, str) else v.name


def _shape(v, shapes=None):
    if shapes is not None and v in shapes:
        return list(shapes[v])
    if hasattr(v, "get_shape"):
        return list(v.get_shape().as_list())
    if hasattr(v, "shape") and not isinstance(v, str):
        return list(v.shape)
    raise ValueError("cannot infer the parameter shape of %r; pass shapes={key: shape}" % (v,))


def convert_dictionary_to_array(dictionary):
    keys = sorted(dictionary.keys(), key=_name)
    first = dictionary[keys[0]]
    n_particles = first.shape[0]
    use_torch = isinstance(first, torch.Tensor)
    access_indices, blocks, at = {}, [], 0
    for v in keys:
        value = dictionary[v]
        width = 1
        for s in value.shape[1:]:
            width *= int(s)
        blocks.append(value.reshape(n_particles, width))
        access_indices[v] = (at, at + width)
        at += width
    if use_torch:
        return torch.cat(blocks, dim=1), access_indices
    array = np.zeros((n_particles, at))
    for v, b in zip(keys, blocks):
        a, e = access_indices[v]
        array[:, a:e] = b
    return array, access_indices


def convert_array_to_dictionary(array, access_indices, shapes=None):
    n_particles = array.shape[0]
    out = {}
    for v, (a, e) in access_indices.items():
        block = array[:, a:e]
        out[v] = block.reshape([n_particles] + _shape(v, shapes))
    return out
