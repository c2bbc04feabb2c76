"""NumPy model of the STAGED protocol (row-block distance, 3-level radix select with summed histograms,
row-block contraction)  --  TEST INFRASTRUCTURE ONLY, never imported by stein_amd.

Two uses:
  * tests/test_radix_model.py checks the radix-select algorithm itself (key transform, two-target tracking,
    even/odd handling) against np.partition, on CPU;
  * tests/test_distributed_cpu.py plugs `NumpyStages` into stein_amd.engine.SvgdEngine in place of the HIP
    backend so the multi-rank collective protocol (all-gather of rows, histogram all-reduce or window-table
    all-reduce with the hit read-back, |phi|^2 all-reduce, identical bandwidth on every rank) runs under gloo
    with world_size 2 on CPU.

It mirrors the semantics of the C ABI in include/steinhip.h, not its performance structure.
"""
import numpy as np

from . import svgd_oracle as orc

BINS, LEVELS = 2048, 3
SHIFT = (21, 10, 0)
BITS = (11, 11, 10)


def f32_keys(x):
    """Monotone map fp32 -> uint32 (same as f32_key in steinhip.hip)."""
    u = np.ascontiguousarray(x, dtype=np.float32).view(np.uint32)
    neg = (u & np.uint32(0x80000000)) != 0
    return np.where(neg, ~u, u | np.uint32(0x80000000)).astype(np.uint32)


def key_to_f32(k):
    k = np.uint32(k)
    u = (k & np.uint32(0x7FFFFFFF)) if (k & np.uint32(0x80000000)) else np.uint32(~k)
    return np.array([u], dtype=np.uint32).view(np.float32)[0]


class SelectState:
    """What the 64-byte device struct holds."""

    def __init__(self, total):
        self.even = total % 2 == 0
        self.rank = [total // 2 - 1 if self.even else total // 2, total // 2]
        self.prefix = [0, 0]
        self.diverged = False
        self.median = self.h2 = None


def hist_pass(values, level, st, hist):
    """hist: int64 [2][BINS] for this level; adds the counts of `values` (any shape)."""
    key = f32_keys(values).reshape(-1).astype(np.uint64)
    digit = (key >> np.uint64(SHIFT[level])) & np.uint64((1 << BITS[level]) - 1)
    hi = key >> np.uint64(SHIFT[level] + BITS[level]) if level > 0 else np.zeros_like(key)
    for tg in range(2 if (st.diverged and level > 0) else 1):
        sel = digit if level == 0 else digit[hi == np.uint64(st.prefix[tg])]
        hist[tg] += np.bincount(sel.astype(np.int64), minlength=BINS)[:BINS]


def resolve(hist, level, n, st):
    div_in = st.diverged
    for tg in range(2):
        h = hist[1 if (div_in and tg == 1) else 0]
        cum = np.cumsum(h)
        b = int(np.searchsorted(cum, st.rank[tg], side="right"))
        b = min(b, BINS - 1)
        st.rank[tg] -= int(cum[b - 1]) if b > 0 else 0
        st.prefix[tg] = (st.prefix[tg] << BITS[level]) | b
    st.diverged = st.prefix[0] != st.prefix[1]
    if level == LEVELS - 1:
        lo, hi = key_to_f32(st.prefix[0]), key_to_f32(st.prefix[1])
        st.lo, st.hi = lo, hi
        st.median = np.float32(0.5) * (lo + hi) if st.even else lo
        with np.errstate(invalid="ignore", divide="ignore"):
            st.h2 = orc.bandwidth_sq(st.median, n, np.float32) if n >= 2 else None


def radix_median(values):
    """Exact median of all entries (compute_median semantics) via the 3-level select."""
    total = int(np.asarray(values).size)
    st = SelectState(total)
    for level in range(LEVELS):
        hist = np.zeros((2, BINS), dtype=np.int64)
        hist_pass(values, level, st, hist)
        resolve(hist, level, max(total, 2), st)
    return st.median


class NumpyStages:
    """Stage backend with the HipStages interface, operating on CPU torch tensors through NumPy views."""

    name = "numpy-model"

    def __init__(self, layout_fn):
        self._layout = layout_fn     # stein_amd._lib.workspace_layout (pure host arithmetic in the .so)
        self._st = None
        self._partial = None

    def workspace_layout(self, n_local, n, d, flags=0, dtype=0):
        return self._layout(n_local, n, d, dtype, flags)

    def x3_prepare(self, T, G, n, d, planes):
        pass   # the split-bf16 operand planes are a device-side detail; the model multiplies in fp64

    def rownorms(self, T, n, d, r):
        t = T.numpy()
        r.numpy()[:] = (t * t).sum(axis=1)

    def distance_block(self, T, r, n, d, row0, n_local, D, ld, hist0=None, symmetric=False, planes=None):
        t, rr = T.numpy(), r.numpy()
        blk = rr[row0:row0 + n_local, None] + rr[None, :] - np.float32(2) * (t[row0:row0 + n_local] @ t.T)
        D.numpy()[:n_local, :n] = blk
        if hist0 is not None:      # the HIP kernel takes the level-0 counts from its accumulators
            hist_pass(blk, 0, self._st, hist0.numpy())

    def median_begin(self, hist, sel, total):
        hist.zero_()
        self._st = SelectState(total)

    # ---- speculative window across ranks (stein_common.h: SpecState; steinhip.hip: k_median_init, k_spec_tally,
    #      k_spec_pick, k_spec_update).  The predictor lives in self._sp; the two state words the engine reads back
    #      (hit, skip_l0) are mirrored into the SELECT section at the offsets of the device struct. ----
    HW_MAX, TABLE_HDR, TABLE_OFF = 32767, 8, 1 << 21

    def _flags(self, sel):
        u = sel.numpy().view(np.uint32)
        u[(64 + 28) // 4] = 1 if self._sp["hit"] else 0
        u[(64 + 52) // 4] = 1 if self._sp["skip_l0"] else 0

    def spec_begin(self, hist, sel, spec, total):
        hist.zero_()
        self._st = SelectState(total)
        sp = getattr(self, "_sp", None) or dict(magic=0, center=0, halfwidth=0, last_key=0)
        ok = sp["magic"] in (1, 2) and sp["halfwidth"] <= self.HW_MAX and 0x80000000 + sp["halfwidth"] <= sp["center"] < 0xFF000000 - sp["halfwidth"]
        sp.update(lo=sp["center"] - sp["halfwidth"] if ok else 0xFFFFFFFF, width=2 * sp["halfwidth"] if ok else 0,
                  count=0, hit=False, below=0, total=total, entries=np.zeros(0, dtype=np.int64))
        sp["skip_l0"] = sp["width"] == 0
        self._sp = sp
        self._flags(sel)

    def distance_block_spec(self, T, r, n, d, row0, n_local, D, ld, hist0, sel, spec, planes=None):
        sp = self._sp
        if sp["width"] == 0:     # no window: the level-0 histogram comes from the distance pass, as without the window
            return self.distance_block(T, r, n, d, row0, n_local, D, ld, hist0=hist0)
        self.distance_block(T, r, n, d, row0, n_local, D, ld, hist0=None)
        keys = f32_keys(D.numpy()[:n_local, :n]).reshape(-1).astype(np.int64)
        sp["below"] = int((keys < sp["lo"]).sum())
        inside = keys[(keys >= sp["lo"]) & (keys <= sp["lo"] + sp["width"])]
        sp["entries"], sp["count"] = inside - sp["lo"], int(inside.size)

    def _table(self, spec):
        return spec.numpy().view(np.int64)[self.TABLE_OFF:self.TABLE_OFF + self.TABLE_HDR + 2 * self.HW_MAX + 2]

    def spec_tally(self, sel, spec):
        sp, table = self._sp, self._table(spec)
        table[:] = 0
        table[0], table[2] = sp["below"], sp["count"]
        if sp["width"] == 0:
            table[1] = 1
        else:
            np.add.at(table[self.TABLE_HDR:], sp["entries"], 1)

    def spec_pick(self, sel, spec, n, h2, median):
        sp, st, table = self._sp, self._st, self._table(spec)
        sp["count"] = int(table[2])                         # the GLOBAL count sizes the next window on every rank alike
        if sp["width"] == 0 or table[1] != 0:
            return
        total, below = sp["total"], int(table[0])
        r0 = total // 2 if total % 2 else total // 2 - 1
        r1 = total // 2
        cum = np.cumsum(table[self.TABLE_HDR:self.TABLE_HDR + sp["width"] + 1])
        if r0 < below or r1 - below >= cum[-1]:
            return
        k0 = int(np.searchsorted(cum, r0 - below, side="right"))
        k1 = int(np.searchsorted(cum, r1 - below, side="right"))
        lo, hi = key_to_f32(sp["lo"] + k0), key_to_f32(sp["lo"] + k1)
        st.lo, st.hi = lo, hi
        st.median = np.float32(0.5) * (lo + hi) if st.even else lo
        st.h2 = orc.bandwidth_sq(st.median, n, np.float32)
        h2.numpy()[0], median.numpy()[0] = st.h2, st.median
        sp["hit"] = sp["skip_l0"] = True
        self._flags(sel)

    def spec_update(self, sel):
        sp, st = self._sp, self._st
        lo = getattr(st, "lo", None)
        if lo is None:                                       # the radix passes produced the median: its lower target
            lo = key_to_f32(st.prefix[0])
        key = int(f32_keys(np.float32(lo)).reshape(-1)[0])
        hw, nxt, earned = 4096, key, 0
        if sp["magic"] in (1, 2):
            pred = np.float32(2) * np.float32(lo) - key_to_f32(sp["last_key"])
            c = int(f32_keys(pred if pred == pred else np.float32(lo)).reshape(-1)[0])
            nxt = min(max(c, 65536), 0xFFFE0000)
            if sp["magic"] == 2 and sp["width"] != 0:
                err = abs(key - sp["center"])
                hw = self.HW_MAX if err > self.HW_MAX // 4 else 4 * err + 48
                hw = max(hw, sp.get("earned_hw", 0) - sp.get("earned_hw", 0) // 4)   # never below 3/4 of the last earned width
                earned = hw
                if sp["hit"] and sp["count"] > ((1 << 21) - 2048) // 2 and hw > sp["halfwidth"] // 2:
                    hw = sp["halfwidth"] // 2 + 1
            sp["magic"] = 2
        else:
            sp["magic"] = 1
        sp.update(last_key=key, center=nxt, halfwidth=min(hw, self.HW_MAX), earned_hw=min(earned, self.HW_MAX))

    def median_hist_pass(self, D, ld, n_local, n, level, sel, hist, symmetric=False):
        hist_pass(D.numpy()[:n_local, :n], level, self._st, hist.numpy()[level])

    def median_resolve(self, hist, level, n, sel, h2, median):
        resolve(hist.numpy()[level], level, n, self._st)
        if level == LEVELS - 1:
            h2.numpy()[0] = self._st.h2
            median.numpy()[0] = self._st.median

    def contract_partial(self, D, ld, T, G, n, d, row0, n_local, h2, ws, planes=None, upper=False):
        K = np.exp(-D.numpy()[:n_local, :n] / h2.numpy()[0] / np.float32(2)).astype(np.float64)
        self._partial = (K @ G.numpy().astype(np.float64), K @ T.numpy().astype(np.float64), K.sum(axis=1))

    def contract_finish(self, T, n, d, row0, n_local, h2, phi, sqnorm, dK, ws, flags=0):
        kg, kt, rs = self._partial
        th = T.numpy()[row0:row0 + n_local].astype(np.float64)
        dk = (rs[:, None] * th - kt) / float(h2.numpy()[0])
        ph = (kg + dk) / n
        phi.numpy()[:] = ph
        sqnorm.numpy()[0] = float((ph.astype(np.float32).astype(np.float64) ** 2).sum())
        if dK is not None:
            dK.numpy()[:] = dk

    def kernel_contract(self, D, ld, T, G, n, d, row0, n_local, h2, phi, sqnorm, dK, ws, planes=None, upper=False):
        self.contract_partial(D, ld, T, G, n, d, row0, n_local, h2, ws)
        self.contract_finish(T, n, d, row0, n_local, h2, phi, sqnorm, dK, ws)

    def kernel_matrix(self, D, ld, n_local, n, h2, K, upper=False):
        K.numpy()[:] = np.exp(-D.numpy()[:n_local, :n] / h2.numpy()[0] / np.float32(2))

    def svgd_phi(self, T, G, n, d, phi, h2, sqnorm, K, dK, ws, flags=0):
        raise RuntimeError("the model backend only implements the staged calls")
