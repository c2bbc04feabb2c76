"""Host-side orchestration of one SVGD direction computation (phi) on one rank.

The arithmetic lives in libsteinhip.so; this module owns device buffers (PyTorch
tensors), the caller-owned workspace, and -- when particles are sharded over
ranks -- the step's collectives.  On GPUs the library issues them itself on its
own RCCL communicator, the whole step being ONE C call (stein_rank_step,
comm="native"); the same protocol written out with torch.distributed
collectives between the rank segments (comm="torch") serves every other backend
(the gloo tests) and as the cross-check of the native path:

    all_gather(theta rows), all_gather(score rows)          once per step
    all_reduce(level histogram)  x3                          median select
    all_reduce(|phi|^2 partial)                              norm clip

Rank p owns rows [p*n/P, (p+1)*n/P) of theta / score / phi / optimizer state and
never sees more than its own [n/P, n] block of the distance matrix; K and D never
cross a link.  Mirrors the single call at
stein/samplers/abstract_stein_sampler.py:103-105 (K, dK = kernel_and_grad; phi).
"""
import ctypes

import torch

from . import _lib


def _ptr(t):
    return ctypes.c_void_p(t.data_ptr()) if t is not None else ctypes.c_void_p(0)


def _dt(t):
    """C-ABI element code of an input tensor (theta / score): fp32, or bf16 for BASELINE config 2."""
    if t.dtype == torch.float32:
        return _lib.F32
    if t.dtype == torch.bfloat16:
        return _lib.BF16
    raise ValueError("theta / score must be float32 or bfloat16 tensors, got %s" % t.dtype)


def _stream(t):
    if t.is_cuda:
        return ctypes.c_void_p(torch.cuda.current_stream(t.device).cuda_stream)
    raise RuntimeError("libsteinhip stages need device (cuda/HIP) tensors; got a %s tensor. "
                       "stein_amd has no CPU path." % t.device)


def untile_distances(block, n_local, n, upper=False):
    """tile-major distance image (flat [rows_padded, ld] view, tiles of [128][32]) -> row-major [n_local, n] copy.
    upper: the image holds only the 128 x 128 tiles on and above the diagonal of a symmetric matrix (the split path's
    symmetric distance pass); the others are filled in from their mirror images."""
    rows, ld = block.shape
    t = block.reshape(rows // 128, ld // 32, 128, 32).permute(0, 2, 1, 3).reshape(rows, ld)
    m = t[:n_local, :n].contiguous()
    if upper:
        bi = torch.arange(n_local, device=m.device) // 128
        lower = bi[:, None] > bi[None, :]              # entries of tiles strictly below the block diagonal
        m = torch.where(lower, m.T, m)
    return m


def tile_distances(matrix, ld):
    """row-major [rows, cols] matrix -> tile-major image [rows padded to 128, ld] (zero padded), ld % 32 == 0"""
    rows, cols = matrix.shape
    rp = (rows + 127) // 128 * 128
    full = torch.zeros(rp, ld, dtype=matrix.dtype, device=matrix.device)
    full[:rows, :cols] = matrix
    return full.reshape(rp // 128, 128, ld // 32, 32).permute(0, 2, 1, 3).contiguous().reshape(rp, ld)


class HipStages:
    """The staged C-ABI calls on torch device tensors (the only product backend)."""

    name = "hip"

    def workspace_layout(self, n_local, n, d, flags=0, dtype=_lib.F32):
        return _lib.workspace_layout(n_local, n, d, dtype, flags)

    def svgd_phi(self, T, G, n, d, phi, h2, sqnorm, K, dK, ws, flags=0):
        _lib.call_on(T.device, "stein_svgd_phi", _ptr(T), _ptr(G), n, d, 0, n, _dt(T), _ptr(phi), _ptr(h2), _ptr(sqnorm),
                  _ptr(K), _ptr(dK), _ptr(ws), ws.numel(), flags, _stream(T))

    def x3_prepare(self, T, G, n, d, planes):
        """T or G may be None: only the other matrix's scales and planes are rebuilt."""
        ref = T if T is not None else G
        _lib.call_on(ref.device, "stein_x3_prepare", _ptr(T), _ptr(G), n, d, _dt(ref), _ptr(planes), planes.numel(), _stream(ref))

    def rownorms(self, T, n, d, r):
        _lib.call_on(T.device, "stein_rownorms", _ptr(T), n, d, _dt(T), _ptr(r), _stream(T))

    def distance_block(self, T, r, n, d, row0, n_local, D, ld, hist0=None, symmetric=False, planes=None, kernel=0):
        """kernel: 0 = the library's choice, _lib.STAGE_TILES / _lib.STAGE_PANEL = force one form of the split path's pass"""
        _lib.call_on(T.device, "stein_distance_block", _ptr(T), _ptr(r), n, d, row0, n_local, _dt(T), _ptr(D), ld, _ptr(hist0),
                  _ptr(planes), (_lib.STAGE_SYMMETRIC if symmetric else 0) | kernel, _stream(T))

    # -- speculative median window across ranks (include/steinhip.h) --
    def spec_begin(self, hist, sel, spec, total):
        _lib.call_on(hist.device, "stein_spec_begin", _ptr(hist), _ptr(sel), _ptr(spec), total, _stream(hist))

    def distance_block_spec(self, T, r, n, d, row0, n_local, D, ld, hist0, sel, spec, planes=None, kernel=0, symmetric=False):
        _lib.call_on(T.device, "stein_distance_block_spec", _ptr(T), _ptr(r), n, d, row0, n_local, _dt(T), _ptr(D), ld, _ptr(hist0),
                  _ptr(planes), (_lib.STAGE_SYMMETRIC if symmetric else 0) | kernel, _ptr(sel), _ptr(spec), _stream(T))

    def spec_tally(self, sel, spec):
        _lib.call_on(sel.device, "stein_spec_tally", _ptr(sel), _ptr(spec), _stream(sel))

    def spec_pick(self, sel, spec, n, h2, median):
        _lib.call_on(sel.device, "stein_spec_pick", _ptr(sel), _ptr(spec), n, _ptr(h2), _ptr(median), _stream(sel))

    def spec_update(self, sel):
        _lib.call_on(sel.device, "stein_spec_update", _ptr(sel), _stream(sel))

    def median_begin(self, hist, sel, total):
        _lib.call_on(hist.device, "stein_median_begin", _ptr(hist), _ptr(sel), total, _stream(hist))

    def median_hist_pass(self, D, ld, n_local, n, level, sel, hist, symmetric=False):
        _lib.call_on(D.device, "stein_median_hist_pass", _ptr(D), ld, n_local, n, level, _ptr(sel), _ptr(hist),
                  _lib.STAGE_SYMMETRIC if symmetric else 0, _stream(D))

    def median_resolve(self, hist, level, n, sel, h2, median):
        _lib.call_on(hist.device, "stein_median_resolve", _ptr(hist), level, n, _ptr(sel), _ptr(h2), _ptr(median), _stream(hist))

    # upper: D is what distance_block(symmetric=True, planes=...) leaves -- only the tiles on and above the diagonal
    def kernel_matrix(self, D, ld, n_local, n, h2, K, upper=False):
        _lib.call_on(D.device, "stein_kernel_matrix", _ptr(D), ld, n_local, n, _ptr(h2), _ptr(K), K.stride(0),
                     _lib.STAGE_UPPER if upper else 0, _stream(D))

    def contract_partial(self, D, ld, T, G, n, d, row0, n_local, h2, ws, planes=None, upper=False):
        _lib.call_on(D.device, "stein_contract_partial", _ptr(D), ld, _ptr(T), _ptr(G), n, d, row0, n_local, _dt(T), _ptr(h2),
                  _ptr(planes), _ptr(ws), ws.numel(), _lib.STAGE_UPPER if upper else 0, _stream(D))

    def contract_finish(self, T, n, d, row0, n_local, h2, phi, sqnorm, dK, ws, flags=0):
        _lib.call_on(T.device, "stein_contract_finish", _ptr(T), n, d, row0, n_local, _dt(T), _ptr(h2), _ptr(phi),
                  _ptr(sqnorm), _ptr(dK), _ptr(ws), ws.numel(), flags, _stream(T))

    # -- rank-step segments: everything one rank does between two collectives, one C call each (include/steinhip.h) --
    def rank_begin(self, T, n, d, row0, n_local, ws, flags):
        _lib.call_on(T.device, "stein_rank_begin", _ptr(T), n, d, row0, n_local, _dt(T), _ptr(ws), ws.numel(), flags, _stream(T))

    def rank_pick(self, T, n, d, row0, n_local, ws, flags, h2, median, flags_host):
        _lib.call_on(T.device, "stein_rank_pick", n, d, row0, n_local, _dt(T), _ptr(ws), ws.numel(), flags, _ptr(h2), _ptr(median),
                     ctypes.c_void_p(flags_host.data_ptr()), _stream(T))

    def rank_radix(self, T, level, need_pass, n, d, row0, n_local, ws, flags, h2, median):
        _lib.call_on(T.device, "stein_rank_radix", level, 1 if need_pass else 0, n, d, row0, n_local, _dt(T), _ptr(ws), ws.numel(),
                     flags, _ptr(h2), _ptr(median), _stream(T))

    def rank_finish(self, T, G, n, d, row0, n_local, h2, phi, sqnorm, dK, ws, flags):
        _lib.call_on(T.device, "stein_rank_finish", _ptr(T), _ptr(G), n, d, row0, n_local, _dt(T), _ptr(h2), _ptr(phi), _ptr(sqnorm),
                     _ptr(dK), _ptr(ws), ws.numel(), flags, _stream(T))

    # -- the library's own RCCL communicator: the whole sharded step as one call (include/steinhip.h) --
    def comm_unique_id(self):
        buf = (ctypes.c_ubyte * _lib.COMM_ID_BYTES)()
        _lib.call("stein_comm_unique_id", buf, _lib.COMM_ID_BYTES)
        return bytes(buf)

    def comm_init(self, device, uid, nranks, rank):
        """collective over the ranks; `device` becomes the communicator's device"""
        out = ctypes.c_void_p(0)
        buf = (ctypes.c_ubyte * _lib.COMM_ID_BYTES).from_buffer_copy(uid)
        _lib.call_on(device, "stein_comm_init", buf, _lib.COMM_ID_BYTES, nranks, rank, ctypes.byref(out))
        return out

    def comm_destroy(self, comm):
        _lib.call("stein_comm_destroy", comm)

    def rank_step(self, comm, theta_local, score_local, T_all, G_all, n, d, phi, h2, median, sqnorm, dK, ws, flags):
        """-> window hit (True / False; None in the radix form)"""
        hit = ctypes.c_int(-1)
        _lib.call_on(T_all.device, "stein_rank_step", comm, _ptr(theta_local), _ptr(score_local), _ptr(T_all), _ptr(G_all), n, d,
                     _dt(T_all), _ptr(phi), _ptr(h2), _ptr(median), _ptr(sqnorm), _ptr(dK), _ptr(ws), ws.numel(), flags,
                     ctypes.byref(hit), _stream(T_all))
        return None if hit.value < 0 else bool(hit.value)

    def kernel_contract(self, D, ld, T, G, n, d, row0, n_local, h2, phi, sqnorm, dK, ws, planes=None, upper=False):
        self.contract_partial(D, ld, T, G, n, d, row0, n_local, h2, ws, planes, upper)
        self.contract_finish(T, n, d, row0, n_local, h2, phi, sqnorm, dK, ws, _lib.FLAG_X3 if planes is not None else 0)


class SvgdEngine:
    """phi for the rows this rank owns.

    Parameters
    ----------
    n, d    : total particle count and parameters per particle.
    device  : torch device of every buffer.
    group   : torch.distributed process group (None -> single rank).  n must divide evenly.
    stages  : backend implementing the staged calls; the product default is HipStages.
              (tests substitute a NumPy model to exercise the collective protocol on CPU/gloo.)
    comm    : who issues the collectives of a sharded step.  "torch" (and "auto", the default): torch.distributed
              collectives between the rank segments -- the path every multi-rank test has run.  "native": the library,
              on its own RCCL communicator, the whole step one C call (stein_rank_step; 68 us of host time per step
              against ~190).  Native is opt-in until a run on two or more GPUs has passed bench.py's native-vs-torch
              cross-check: so far it has only ever seen one-rank groups, where every collective is a self-copy.
    """

    _full_distance_image = False   # set by scratch/ab.py for -DSTEIN_NO_UPPER builds (the mirrored image of round 1)

    def __init__(self, n, d, device="cuda", group=None, stages=None, x3=None, dtype=torch.float32, small=True,
                 window=True, force_collectives=False, comm="auto", tile_distance=False, dist_window=None):
        self.n, self.d = int(n), int(d)
        # dtype of the theta / score tensors handed to compute_phi: float32, or bfloat16 (BASELINE config 2: the
        # values are used as they are, K is rounded to bf16, one bf16 MFMA per product, fp32 accumulation)
        if dtype not in (torch.float32, torch.bfloat16):
            raise ValueError("dtype must be torch.float32 or torch.bfloat16")
        self.dtype = dtype
        # x3 (default): both GEMMs run on the 16-bit matrix cores with every fp32 operand scaled by a power of two and
        # split into two fp16 terms -- fp32-level accuracy at a fraction of the fp32-MFMA time (stein_x3.hip).
        # x3=False selects the fp32-input MFMA kernels (an exact k-ordered fmaf chain).  None = the default (split path).
        # Constructor arguments are the only switches: neither this module nor the library reads the environment.
        if x3 is None:
            x3 = True
        self.x3 = bool(x3) or dtype == torch.bfloat16   # bf16 inputs only exist on the bf16-MFMA kernels
        # small=False: the fused call never takes the one-kernel path for n <= 160 (tests of the tiled kernels)
        # window=False: the fused call never uses the speculative median window (every step pays the radix-select passes;
        # same results -- bench.py times the miss path this way)
        # tile_distance=True: the fused call's distance pass always runs the per-tile kernel (A/B against the
        # panel-resident kernel that large blocks with d <= 256 take by default, stein_dpanel.hip)
        self.flags = ((_lib.FLAG_X3 if self.x3 else 0) | (0 if small else _lib.FLAG_TILED) |
                      (0 if window else _lib.FLAG_NO_WINDOW) | (_lib.FLAG_TILE_DISTANCE if tile_distance else 0))
        # several ranks: use the speculative median window (ONE 512 KB all-reduce and a hit-flag read-back per step
        # instead of three histogram all-reduces and two passes over the local distance block) when the block is large
        # enough for that to pay (>= 2^24 entries; every collective costs ~30 us of host time from Python, and the
        # read-back no longer leaves a bubble on the stream: _sharded_step); dist_window=True / False forces it
        self.dist_window = False
        self.window_hit = None
        self.device = torch.device(device)
        self.stages = stages if stages is not None else HipStages()
        self.group = group
        if group is not None:
            import torch.distributed as dist
            self.world, self.rank = dist.get_world_size(group), dist.get_rank(group)
        else:
            self.world, self.rank = 1, 0
        # sharded: run the multi-rank protocol (gathers, histogram / table / scalar all-reduces between the staged calls).
        # force_collectives runs it on a one-rank group as well -- how the tests drive every collective through RCCL on
        # a single card.
        self.sharded = self.world > 1 or (group is not None and force_collectives)
        if self.n < 2:
            raise ValueError("n_particles = %d: the median-heuristic bandwidth divides by ln(n); need n >= 2" % self.n)
        if self.n % self.world:
            raise ValueError("n_particles (%d) must be divisible by the number of ranks (%d)" % (self.n, self.world))
        self.n_local = self.n // self.world
        self.row0 = self.rank * self.n_local

        if self.sharded:
            self.flags |= _lib.FLAG_TILED        # a rank never takes the one-kernel path; its workspace has every section
        total, offs, extra = self.stages.workspace_layout(self.n_local, self.n, self.d, self.flags,
                                                          _lib.BF16 if dtype == torch.bfloat16 else _lib.F32)
        self.ws_bytes, self._offs = total, offs
        self.ld_dist, self.split = extra[_lib.WSX_LD_DIST], extra[_lib.WSX_SPLIT]
        dev = self.device
        self.ws = torch.empty(total, dtype=torch.uint8, device=dev)
        # the SELECT section carries the median predictor from call to call and is trusted once its magic word matches:
        # start it clean, so that which select path runs never depends on what the allocator's block held before
        o = offs[_lib.WS_SELECT]
        self.ws[o:o + _lib.SELECT_BYTES].zero_()
        self.phi = torch.empty(self.n_local, self.d, dtype=torch.float32, device=dev)
        self.h2 = torch.zeros(1, dtype=torch.float32, device=dev)
        self.median = torch.zeros(1, dtype=torch.float32, device=dev)
        self.sqnorm = torch.zeros(1, dtype=torch.float64, device=dev)
        if self.sharded and hasattr(self.stages, "spec_begin"):
            self.dist_window = bool(dist_window) if dist_window is not None else self.n_local * self.n >= (1 << 24)
        if self.sharded:
            self.T_all = torch.empty(self.n, self.d, dtype=dtype, device=dev)
            self.G_all = torch.empty(self.n, self.d, dtype=dtype, device=dev)
        self.dist_upper = False      # the distance image holds only the tiles on and above the diagonal (set per step)
        # the fused call takes the one-kernel path for this shape (stein_small.hip: n <= 160): no distance image exists.
        # The layout tells: only then is the SPEC section of the workspace empty (stein_make_layout)
        self._one_kernel = (not self.sharded) and offs[_lib.WS_PLANES] == offs[_lib.WS_SPEC]
        self._have_dist = not self._one_kernel   # (a caller that drives the staged calls on this engine's buffers gets an image too)
        self._flags_host = None      # page-locked landing place of the window's hit flag (HIP stages, window form)
        self._flags_event = None
        if comm not in ("auto", "native", "torch"):
            raise ValueError("comm must be 'auto', 'native' or 'torch'")
        self._comm = None
        self.comm_error = None       # why comm="native" could not be set up (the same text on every rank)
        if self.sharded and comm == "native":
            import torch.distributed as dist
            able = self.device.type == "cuda" and hasattr(self.stages, "rank_step") and "nccl" in str(dist.get_backend(group))
            if not able:             # a property of the group and the stages: every rank decides the same
                raise ValueError("comm='native' needs CUDA/HIP tensors, the HIP stages and an nccl (RCCL) process group")
            self._comm = self._make_native_comm()
            if self._comm is None:   # decided collectively: every rank raises
                raise RuntimeError("comm='native': " + self.comm_error)
        self.comm = "native" if self._comm is not None else ("torch" if self.sharded else None)

    def _make_native_comm(self):
        """Group rank 0 makes the 128-byte RCCL id, the group broadcasts it, every rank joins -- collective, and
        collective in failure too: every rank takes part in the same two torch collectives whatever happens to it, so
        that a rank whose RCCL is missing (or whose ncclCommInitRank fails) cannot leave the others waiting inside a
        broadcast.  Returns the communicator, or None on EVERY rank (self.comm_error says why)."""
        import torch.distributed as dist
        src = dist.get_global_rank(self.group, 0)
        msg = torch.zeros(_lib.COMM_ID_BYTES + 1, dtype=torch.uint8, device=self.device)   # id | status byte (1 = id valid)
        if self.rank == 0:
            try:
                uid0 = self.stages.comm_unique_id()
                msg[:_lib.COMM_ID_BYTES].copy_(torch.frombuffer(bytearray(uid0), dtype=torch.uint8))
                msg[_lib.COMM_ID_BYTES] = 1
            except Exception as e:      # rank 0 still broadcasts: the status byte tells the others
                self.comm_error = "rank 0 could not make an RCCL id: %s" % e
        dist.broadcast(msg, src=src, group=self.group)
        host = msg.cpu()
        comm, err = None, None
        if int(host[_lib.COMM_ID_BYTES]) == 1:
            try:
                comm = self.stages.comm_init(self.device, bytes(host[:_lib.COMM_ID_BYTES].numpy().tobytes()), self.world,
                                             self.rank)
            except Exception as e:
                err = "rank %d could not join the communicator: %s" % (self.rank, e)
        else:
            err = self.comm_error or "rank 0 could not make an RCCL id"
        ok = torch.tensor([1 if comm is not None else 0], dtype=torch.int32, device=self.device)
        dist.all_reduce(ok, op=dist.ReduceOp.MIN, group=self.group)
        if int(ok.item()) == 1:
            return comm
        if comm is not None:            # some other rank failed: nobody keeps a communicator
            self.stages.comm_destroy(comm)
        self.comm_error = err or "another rank could not join the communicator"
        return None

    def close(self):
        """release the library's communicator (collective-free; safe to call twice)"""
        comm, self._comm = self._comm, None
        if comm is not None:
            self.stages.comm_destroy(comm)

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # views into the workspace -------------------------------------------------------------
    def _section(self, sec, nbytes, dtype):
        o = self._offs[sec]
        return self.ws[o:o + nbytes].view(dtype)

    @property
    def rownorm(self):
        return self._section(_lib.WS_ROWNORM, self.n * 4, torch.float32)

    @property
    def dist(self):
        """The distance block as the kernels store it: tile-major, [row blocks of 128][ld/32 column tiles][128][32]
        fp32 (flat 2-D view [rows padded to 128, ld]).  Use dist_matrix() for a row-major copy."""
        rows = (self.n_local + 127) // 128 * 128
        return self._section(_lib.WS_DIST, rows * self.ld_dist * 4, torch.float32).view(rows, self.ld_dist)

    def dist_matrix(self):
        """Row-major [n_local, n] copy of the distance block (de-tiled; for inspection and tests).  After a single-rank
        step on the split path only the tiles on and above the diagonal are stored; the rest is mirrored in here."""
        if not self._have_dist:
            raise RuntimeError("no distance image: this engine's fused call takes the one-kernel path (n <= 160), which "
                               "keeps D in LDS; build the engine with small=False to get one")
        return untile_distances(self.dist, self.n_local, self.n, upper=self.dist_upper)

    @property
    def hist(self):
        nb = _lib.HIST_LEVELS * 2 * _lib.HIST_BINS * 8
        return self._section(_lib.WS_HIST, nb, torch.int64).view(_lib.HIST_LEVELS, 2, _lib.HIST_BINS)

    @property
    def select_state(self):
        """radix-select state (64 bytes) followed by the speculative-window state (64 bytes)"""
        return self._section(_lib.WS_SELECT, 128, torch.uint8)

    def window_stats(self):
        """(medians recorded since the predictor started, how many of them the speculative window delivered) --
        counted on the device by the library; reading them synchronises with the stream."""
        w = self.select_state[_lib.SPEC_NSTEPS_OFFSET:_lib.SPEC_NHITS_OFFSET + 4].view(torch.int32).cpu()
        return int(w[0]), int(w[1])

    @property
    def spec_section(self):
        o = self._offs[_lib.WS_SPEC]
        return self.ws[o:self._offs[_lib.WS_PLANES]]

    @property
    def spec_table(self):
        """the rank-summed window table (int64 view), behind the slots and the entry buffer of the SPEC section"""
        o = self._offs[_lib.WS_SPEC] + 8 * _lib.SPEC_TABLE_OFFSET_WORDS
        return self.ws[o:o + 8 * _lib.SPEC_TABLE_WORDS].view(torch.int64)

    @property
    def planes(self):
        """split-precision operand planes and scales (None unless x3)"""
        if not self.x3:
            return None
        o = self._offs[_lib.WS_PLANES]
        return self.ws[o:self.ws_bytes]

    # ---------------------------------------------------------------------------------------
    def _radix_levels(self, first_level, need_level0_pass=False):
        """Levels first_level..2 of the radix select: local histogram pass (level 0 normally comes from the distance
        epilogue), all-reduce over the ranks, resolve."""
        st, n, nl, D, ld, hist, sel = self.stages, self.n, self.n_local, self.dist, self.ld_dist, self.hist, self.select_state
        sym = not self.sharded
        for level in range(first_level, _lib.HIST_LEVELS):
            if level > 0 or need_level0_pass:
                st.median_hist_pass(D, ld, nl, n, level, sel, hist, symmetric=sym)
            if self.sharded:
                import torch.distributed as dist
                dist.all_reduce(hist[level], op=dist.ReduceOp.SUM, group=self.group)
            st.median_resolve(hist, level, n, sel, self.h2, self.median)

    def _median_with_window(self, T_all, mark):
        """Several ranks, large blocks: the speculative window of the fused call, with the per-key tallies summed over
        the ranks by one all-reduce.  The host reads the hit flag back (one small synchronising copy per step) to
        decide whether the radix-select passes and their three all-reduces are needed at all."""
        import torch.distributed as dist
        st, n, d, nl = self.stages, self.n, self.d, self.n_local
        D, ld, hist, sel, spec = self.dist, self.ld_dist, self.hist, self.select_state, self.spec_section
        st.spec_begin(hist, sel, spec, n * n)
        mark("distance")
        st.distance_block_spec(T_all, self.rownorm, n, d, self.row0, nl, D, ld, hist[0], sel, spec, planes=self.planes)
        mark("median")
        st.spec_tally(sel, spec)
        dist.all_reduce(self.spec_table, op=dist.ReduceOp.SUM, group=self.group)
        st.spec_pick(sel, spec, n, self.h2, self.median)
        flags = sel[_lib.SPEC_HIT_OFFSET:_lib.SPEC_SKIP_L0_OFFSET + 4].cpu()      # synchronises with the stream
        hit = bool(flags[:4].view(torch.int32).item())
        skip_l0 = bool(flags[-4:].view(torch.int32).item())
        self.window_hit = hit
        if not hit:
            self._radix_levels(0, need_level0_pass=not skip_l0)
        st.spec_update(sel)

    def _sharded_step(self, theta_local, score_local, dK_out, timing=False):
        """The multi-rank step on the HIP stages: one C call per segment between collectives (stein_rank_*), nothing
        else from the host but the collectives themselves.  In the window form the hit flag lands in page-locked memory
        behind an event; the host waits for it while the GPU already builds the score's operand planes, so the read-back
        leaves no bubble on the stream."""
        import torch.distributed as dist
        st, n, d, nl, row0, ws = self.stages, self.n, self.d, self.n_local, self.row0, self.ws
        flags = (self.flags & _lib.FLAG_X3) | (_lib.FLAG_RANK_WINDOW if self.dist_window else 0)
        T_all, G_all, planes = self.T_all, self.G_all, self.planes
        # theta first (collectives of one group run in issue order); the score rows are not needed before the
        # contraction, so their all-gather is asynchronous and runs beside the distance pass
        dist.all_gather_into_tensor(T_all, theta_local, group=self.group)
        gather_g = dist.all_gather_into_tensor(G_all, score_local, group=self.group, async_op=True)
        st.rank_begin(T_all, n, d, row0, nl, ws, flags)

        def score_planes():
            gather_g.wait()            # the launching stream waits for the gathered score rows (the host does not)
            if planes is not None:
                st.x3_prepare(None, G_all, n, d, planes)

        def radix(need_level0_pass):
            if need_level0_pass:
                st.rank_radix(T_all, 0, True, n, d, row0, nl, ws, flags, self.h2, self.median)
            for level in range(_lib.HIST_LEVELS):
                dist.all_reduce(self.hist[level], op=dist.ReduceOp.SUM, group=self.group)
                st.rank_radix(T_all, level, False, n, d, row0, nl, ws, flags, self.h2, self.median)

        if self.dist_window:
            if self._flags_host is None:
                self._flags_host = torch.zeros(8, dtype=torch.int32).pin_memory()
                self._flags_event = torch.cuda.Event()
            dist.all_reduce(self.spec_table, op=dist.ReduceOp.SUM, group=self.group)
            st.rank_pick(T_all, n, d, row0, nl, ws, flags, self.h2, self.median, self._flags_host)
            self._flags_event.record(torch.cuda.current_stream(self.device))
            score_planes()                       # GPU work that does not depend on the flag: covers the host's wake-up
            self._flags_event.synchronize()
            hit, skip_l0 = bool(self._flags_host[0]), bool(self._flags_host[6])
            self.window_hit = hit
            if not hit:
                radix(not skip_l0)
        else:
            radix(False)
            score_planes()
        st.rank_finish(T_all, G_all, n, d, row0, nl, self.h2, self.phi, self.sqnorm, dK_out, ws,
                       flags | (_lib.FLAG_TIMING if timing else 0))
        dist.all_reduce(self.sqnorm, op=dist.ReduceOp.SUM, group=self.group)
        return self.phi

    def compute_phi(self, theta_local, score_local, K_out=None, dK_out=None, mark=None, timing=False):
        """theta_local, score_local: [n_local, d] float32 contiguous device tensors (this rank's rows).

        Returns self.phi ([n_local, d] float32, unclipped).  Afterwards self.h2 holds bandwidth^2
        and self.sqnorm the GLOBAL |phi|_F^2 (fp64), both on device; nothing syncs with the host.

        mark: optional callable(label) invoked between stages on the launching stream (bench.py records
        HIP events there to time individual kernels).  Passing it selects the staged calls: the same kernels
        as the fused call except that the median always takes the radix-select passes (the speculative
        window of the fused call needs state that persists inside one workspace, see stein_common.h).
        timing: let the library record HIP events on the stream at its stage boundaries (_lib.timing_reserve /
        _lib.timing_read): every stage of the fused call on a single rank, the contraction and the finish pass in a
        sharded step.  timing="contract" (single rank): only the two events around the contraction -- every event between two
        kernels costs the step ~3 us, so a loop that is itself being timed should carry as few as it can.
        """
        st, n, d, nl = self.stages, self.n, self.d, self.n_local
        for name, t in (("theta", theta_local), ("score", score_local)):
            if tuple(t.shape) != (nl, d) or t.dtype != self.dtype or not t.is_contiguous():
                raise ValueError("%s must be a contiguous %s [%d, %d] tensor, got %s %s" %
                                 (name, self.dtype, nl, d, tuple(t.shape), t.dtype))
        if not self.sharded and mark is None:
            st.svgd_phi(theta_local, score_local, n, d, self.phi, self.h2, self.sqnorm, K_out, dK_out, self.ws,
                        self.flags | (_lib.FLAG_TIMING if timing else 0) |
                        (_lib.FLAG_TIMING_CONTRACT if timing == "contract" else 0))
            self._have_dist = not self._one_kernel
            self.dist_upper = self.x3 and self._have_dist and not SvgdEngine._full_distance_image
            return self.phi
        if self.sharded and mark is None and K_out is None and self._comm is not None:
            flags = ((self.flags & _lib.FLAG_X3) | (_lib.FLAG_RANK_WINDOW if self.dist_window else 0) |
                     (_lib.FLAG_TIMING if timing else 0))
            self.window_hit = st.rank_step(self._comm, theta_local, score_local, self.T_all, self.G_all, n, d, self.phi,
                                           self.h2, self.median, self.sqnorm, dK_out, self.ws, flags)
            self._have_dist, self.dist_upper = True, False
            return self.phi
        if self.sharded and mark is None and K_out is None and hasattr(st, "rank_begin"):
            self._have_dist, self.dist_upper = True, False
            return self._sharded_step(theta_local, score_local, dK_out, timing)
        if mark is None:
            def mark(label):
                return None

        gather_g = None
        if self.sharded:
            import torch.distributed as dist
            mark("gather")
            # theta first (collectives of one group run in issue order); the score rows are not needed before the
            # contraction, so their all-gather is asynchronous and runs beside the distance pass
            dist.all_gather_into_tensor(self.T_all, theta_local, group=self.group)
            gather_g = dist.all_gather_into_tensor(self.G_all, score_local, group=self.group, async_op=True)
            T_all, G_all = self.T_all, self.G_all
        else:
            T_all, G_all = theta_local, score_local
        D, ld, hist, sel = self.dist, self.ld_dist, self.hist, self.select_state
        mark("rownorms")
        st.rownorms(T_all, n, d, self.rownorm)
        planes = self.planes
        if planes is not None:
            st.x3_prepare(T_all, None if gather_g is not None else G_all, n, d, planes)
        # the distance pass fills the level-0 histogram from its accumulators; a single rank holds the whole
        # symmetric matrix and only computes / counts its upper triangle
        sym = not self.sharded
        if self.dist_window:
            self._median_with_window(T_all, mark)
        else:
            st.median_begin(hist, sel, n * n)
            mark("distance")
            st.distance_block(T_all, self.rownorm, n, d, self.row0, nl, D, ld, hist0=hist[0], symmetric=sym,
                              planes=planes)
            mark("median")
            self._radix_levels(0)
        upper = sym and planes is not None     # what the symmetric distance pass of the split path stores
        if SvgdEngine._full_distance_image:     # A/B scripts against -DSTEIN_NO_UPPER builds of the library only
            upper = False
        self.dist_upper, self._have_dist = upper, True
        if K_out is not None:
            st.kernel_matrix(D, ld, nl, n, self.h2, K_out, upper)
        if gather_g is not None:
            gather_g.wait()            # the launching stream waits for the gathered score rows (the host does not)
            if planes is not None:
                st.x3_prepare(None, G_all, n, d, planes)
        mark("contract")
        st.contract_partial(D, ld, T_all, G_all, n, d, self.row0, nl, self.h2, self.ws, planes, upper)
        mark("finish")
        st.contract_finish(T_all, n, d, self.row0, nl, self.h2, self.phi, self.sqnorm, dK_out, self.ws,
                           self.flags & ~_lib.FLAG_TILE_DISTANCE)
        if self.sharded:
            dist.all_reduce(self.sqnorm, op=dist.ReduceOp.SUM, group=self.group)
        mark("end")
        return self.phi
