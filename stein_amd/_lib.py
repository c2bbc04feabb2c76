"""ctypes binding of libsteinhip.so (the C ABI declared in include/steinhip.h).

There is no CPU fallback: if the shared library is missing or a call fails, an
exception is raised.  Device pointers are taken from PyTorch-ROCm tensors with
``.data_ptr()``; PyTorch is used for device memory and streams only.
"""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libsteinhip.so")

OK, E_BADARG, E_SHAPE, E_WORKSPACE, E_HIP, E_RCCL, E_UNSUPPORTED = 0, -1, -2, -3, -4, -5, -6
F32, BF16, F64 = 0, 1, 2

WS_ROWNORM, WS_DIST, WS_HIST, WS_SELECT, WS_PART_G, WS_PART_T, WS_PART_RS, WS_SQPART, WS_SPEC, WS_PLANES = range(10)
WS_NSECTIONS = 10
WSX_LD_DIST, WSX_SPLIT, WSX_SQ_BLOCKS, WSX_HIST_BINS = range(4)
WSX_N = 4
HIST_BINS, HIST_LEVELS = 2048, 3
STAGE_SYMMETRIC = 1
STAGE_UPPER = 2
STAGE_TILES = 4     # distance pass: always the per-tile kernel
STAGE_PANEL = 8     # distance pass: the panel-resident kernel whenever its restrictions hold
FLAG_X3 = 1
FLAG_TIMING = 4
FLAG_TILED = 8
FLAG_NO_WINDOW = 16
FLAG_RANK_WINDOW = 32
FLAG_TILE_DISTANCE = 64
FLAG_TIMING_CONTRACT = 128
GLM_LINEAR, GLM_LOGISTIC = 0, 1
SPEC_TABLE_WORDS = 65544          # uint64 words of the rank-summed window table ...
SPEC_TABLE_OFFSET_WORDS = 1 << 21  # ... which starts 2^21 words into the SPEC section (slots + entry buffer)
SPEC_HIT_OFFSET, SPEC_SKIP_L0_OFFSET = 64 + 28, 64 + 52   # uint32 state words inside the SELECT section
SPEC_NSTEPS_OFFSET, SPEC_NHITS_OFFSET = 64 + 56, 64 + 60   # medians recorded since the predictor started / window hits
SELECT_BYTES = 192                                        # SelState + SpecState + FuseState
COMM_ID_BYTES = 128                                       # STEIN_COMM_ID_BYTES
T_STAGES = ("prepare", "distance", "median", "contract", "finish")   # STEIN_T_* of include/steinhip.h

_c = ctypes
_vp, _i64, _int, _dbl, _sz = _c.c_void_p, _c.c_int64, _c.c_int, _c.c_double, _c.c_size_t

# name -> argtypes; every function returns int except the two noted below
_SIGNATURES = {
    "stein_workspace_bytes": [_i64, _i64, _i64, _int, _int, _c.POINTER(_sz)],
    "stein_workspace_layout": [_i64, _i64, _i64, _int, _int, _c.POINTER(_sz), _c.POINTER(_i64)],
    "stein_spec_begin": [_vp, _vp, _vp, _i64, _vp],
    "stein_distance_block_spec": [_vp, _vp, _i64, _i64, _i64, _i64, _int, _vp, _i64, _vp, _vp, _int, _vp, _vp, _vp],
    "stein_spec_tally": [_vp, _vp, _vp],
    "stein_spec_pick": [_vp, _vp, _i64, _vp, _vp, _vp],
    "stein_spec_update": [_vp, _vp],
    "stein_score_glm": [_vp, _i64, _i64, _int, _i64, _i64, _i64, _vp, _vp, _i64, _dbl, _dbl, _dbl, _vp, _vp],
    "stein_score_bnn": [_vp, _i64, _i64, _i64, _i64, _c.POINTER(_i64), _vp, _vp, _i64, _dbl, _dbl, _dbl, _vp, _vp],
    "stein_rank_begin": [_vp, _i64, _i64, _i64, _i64, _int, _vp, _sz, _int, _vp],
    "stein_rank_pick": [_i64, _i64, _i64, _i64, _int, _vp, _sz, _int, _vp, _vp, _vp, _vp],
    "stein_rank_radix": [_int, _int, _i64, _i64, _i64, _i64, _int, _vp, _sz, _int, _vp, _vp, _vp],
    "stein_rank_finish": [_vp, _vp, _i64, _i64, _i64, _i64, _int, _vp, _vp, _vp, _vp, _vp, _sz, _int, _vp],
    "stein_comm_unique_id": [_vp, _sz],
    "stein_comm_init": [_vp, _sz, _int, _int, _c.POINTER(_vp)],
    "stein_comm_info": [_vp, _c.POINTER(_int), _c.POINTER(_int)],
    "stein_comm_destroy": [_vp],
    "stein_rank_step": [_vp, _vp, _vp, _vp, _vp, _i64, _i64, _int, _vp, _vp, _vp, _vp, _vp, _vp, _sz, _int,
                        _c.POINTER(_int), _vp],
    "stein_timing_reserve": [_int],
    "stein_timing_read": [_c.POINTER(_c.c_float), _int, _c.POINTER(_int)],
    "stein_svgd_phi": [_vp, _vp, _i64, _i64, _i64, _i64, _int, _vp, _vp, _vp, _vp, _vp, _vp, _sz, _int, _vp],
    "stein_rownorms": [_vp, _i64, _i64, _int, _vp, _vp],
    "stein_distance_block": [_vp, _vp, _i64, _i64, _i64, _i64, _int, _vp, _i64, _vp, _vp, _int, _vp],
    "stein_x3_prepare": [_vp, _vp, _i64, _i64, _int, _vp, _sz, _vp],
    "stein_median_begin": [_vp, _vp, _i64, _vp],
    "stein_median_hist_pass": [_vp, _i64, _i64, _i64, _int, _vp, _vp, _int, _vp],
    "stein_median_resolve": [_vp, _int, _i64, _vp, _vp, _vp, _vp],
    "stein_kernel_matrix": [_vp, _i64, _i64, _i64, _vp, _vp, _i64, _int, _vp],
    "stein_kernel_contract": [_vp, _i64, _vp, _vp, _i64, _i64, _i64, _i64, _int, _vp, _vp, _vp, _vp, _vp, _vp, _sz, _int, _vp],
    "stein_contract_partial": [_vp, _i64, _vp, _vp, _i64, _i64, _i64, _i64, _int, _vp, _vp, _vp, _sz, _int, _vp],
    "stein_contract_finish": [_vp, _i64, _i64, _i64, _i64, _int, _vp, _vp, _vp, _vp, _vp, _sz, _int, _vp],
    "stein_apply_adagrad": [_vp, _vp, _int, _vp, _i64, _int, _vp, _dbl, _dbl, _dbl, _dbl, _dbl, _int, _vp, _vp],
    "stein_apply_adam": [_vp, _vp, _int, _vp, _vp, _i64, _int, _vp, _dbl, _dbl, _dbl, _dbl, _dbl, _dbl, _i64, _vp, _vp],
    "stein_cast_f64_to_f32": [_vp, _vp, _i64, _vp],
    "stein_cast_f32_to_bf16": [_vp, _vp, _i64, _vp],
    "stein_take_device_error": [],
    "stein_debug_hist_all_grid": [_int],
    "stein_debug_hist_all_vblocks": [_int],
    "stein_debug_raise_device_error": [],
}
EXPORTED_SYMBOLS = sorted(list(_SIGNATURES) + ["stein_version", "stein_last_error"])

_lib = None


class SteinHipError(RuntimeError):
    def __init__(self, code, message):
        super().__init__("libsteinhip error %d: %s" % (code, message))
        self.code = code


def load():
    """Load the library once.  Raises if it has not been built (python __graft_entry__.py)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            "libsteinhip.so is not built (%s); run `python -c 'import __graft_entry__ as g; g.build()'`. "
            "stein_amd has no CPU fallback." % LIB_PATH)
    lib = ctypes.CDLL(LIB_PATH)
    for name, args in _SIGNATURES.items():
        fn = getattr(lib, name)
        fn.argtypes, fn.restype = args, _int
    lib.stein_version.argtypes, lib.stein_version.restype = [], _int
    lib.stein_last_error.argtypes, lib.stein_last_error.restype = [], _c.c_char_p
    _lib = lib
    return lib


def check(rc):
    if rc != OK:
        msg = load().stein_last_error().decode("utf-8", "replace")
        if rc in (E_BADARG, E_SHAPE, E_UNSUPPORTED):
            raise ValueError("libsteinhip error %d: %s" % (rc, msg))
        raise SteinHipError(rc, msg)


def call(name, *args):
    check(getattr(load(), name)(*args))


def call_on(device, name, *args):
    """`call` with `device` (a torch.device of type cuda) as the process's current HIP device for its duration.

    The library launches on the caller's stream; stream 0 and kernel attributes belong to the CURRENT device, so a
    tensor on cuda:1 handed over while cuda:0 is current would be launched on the wrong GPU.  The common case (the
    tensor's device is already current) costs one integer comparison.
    """
    import torch
    idx = device.index if device.index is not None else torch.cuda.current_device()
    if idx == torch.cuda.current_device():
        return call(name, *args)
    with torch.cuda.device(idx):
        return call(name, *args)


def workspace_layout(n_local, n, d, dtype=F32, flags=0):
    """-> (total_bytes, offsets[WS_NSECTIONS], extra[WSX_N]).  Pure host arithmetic, no GPU needed."""
    total = _sz(0)
    call("stein_workspace_bytes", n_local, n, d, dtype, flags, ctypes.byref(total))
    offs = (_sz * WS_NSECTIONS)()
    extra = (_i64 * WSX_N)()
    call("stein_workspace_layout", n_local, n, d, dtype, flags, offs, extra)
    return int(total.value), [int(o) for o in offs], [int(e) for e in extra]


def version():
    return load().stein_version()


def timing_reserve(calls):
    """Reserve HIP events for `calls` fused calls made with FLAG_TIMING (and rewind the cursor)."""
    call("stein_timing_reserve", int(calls))


def timing_read(max_calls):
    """-> list of {stage: ms} for the timed fused calls since timing_reserve (waits for their events)."""
    buf = (_c.c_float * (max_calls * len(T_STAGES)))()
    got = _int(0)
    call("stein_timing_read", buf, int(max_calls), _c.byref(got))
    k = len(T_STAGES)
    return [{T_STAGES[j]: float(buf[i * k + j]) for j in range(k)} for i in range(got.value)]
