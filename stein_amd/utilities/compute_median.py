"""Exact median of every entry of a matrix, on device.

stein/utilities/compute_median.py:4-16 takes tf.nn.top_k of the flattened input;
here the same order statistics come from libsteinhip's three-level radix select on
the fp32 bit patterns (even count -> mean of the two middle values, in fp32).
"""
import numpy as np
import torch

from .. import _lib
from ..engine import HipStages, tile_distances


def compute_median(D, device="cuda"):
    """Median of all entries of the array/tensor `D` (vector or any [rows, cols] matrix); returns a float32 scalar
    (NumPy scalar for NumPy input, 0-d device tensor for tensor input)."""
    was_numpy = not isinstance(D, torch.Tensor)
    t = torch.as_tensor(np.asarray(D, dtype=np.float32)) if was_numpy else D
    if t.dim() == 1:
        t = t.reshape(1, -1)
    if t.dim() != 2 or t.numel() == 0:
        raise ValueError("compute_median expects a non-empty vector or matrix, got shape %s" % (tuple(t.shape),))
    rows, n = t.shape
    ld = (n + 63) // 64 * 64
    buf = tile_distances(t.to(device=device, dtype=torch.float32), ld)   # the select kernels read [128][32] tiles
    hist = torch.zeros(_lib.HIST_LEVELS, 2, _lib.HIST_BINS, dtype=torch.int64, device=device)
    sel = torch.zeros(64, dtype=torch.uint8, device=device)
    h2 = torch.zeros(1, dtype=torch.float32, device=device)
    med = torch.zeros(1, dtype=torch.float32, device=device)
    st = HipStages()
    st.median_begin(hist, sel, rows * n)
    for level in range(_lib.HIST_LEVELS):
        st.median_hist_pass(buf, ld, rows, n, level, sel, hist)
        st.median_resolve(hist, level, max(rows * n, 2), sel, h2, med)   # h2 output unused here
    return med.cpu().numpy()[0] if was_numpy else med[0]
