"""CPU: the C-ABI library loads, exports every symbol the header declares, and its host-side argument
checks work -- no compute calls (there is no GPU here)."""
import ctypes
import os
import re

import pytest

from stein_amd import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def header_functions():
    text = open(os.path.join(ROOT, "include", "steinhip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(stein_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    lib = _lib.load()
    declared = header_functions()
    assert len(declared) >= 19
    for name in declared:
        assert hasattr(lib, name), "libsteinhip.so does not export %s" % name
    assert set(declared) == set(_lib.EXPORTED_SYMBOLS), "ctypes table and header disagree"
    assert _lib.version() == 100


def test_workspace_layout_is_consistent():
    for nl, n, d in [(100, 100, 10), (4096, 4096, 128), (16384, 16384, 256), (2048, 16384, 256), (8192, 8192, 2001),
                     (16384, 131072, 256)]:
        total, offs, extra = _lib.workspace_layout(nl, n, d)
        assert offs == sorted(offs) and offs[-1] <= total
        assert all(o % 256 == 0 for o in offs)
        ld = extra[_lib.WSX_LD_DIST]
        assert ld >= n and ld % 64 == 0
        assert offs[_lib.WS_HIST] - offs[_lib.WS_DIST] >= (nl + 127) // 128 * 128 * ld * 4   # tile-major, rows padded to 128
        assert 1 <= extra[_lib.WSX_SPLIT] <= 16
        assert extra[_lib.WSX_HIST_BINS] == _lib.HIST_BINS
    # C3 fits comfortably: D is 1 GiB, everything else (partials, 16 MiB window buffer, ...) < 128 MiB
    total, _, extra = _lib.workspace_layout(16384, 16384, 256)
    assert (1 << 30) < total < (1 << 30) + (128 << 20)


@pytest.mark.parametrize("args", [(1, 1, 8), (0, 8, 8), (9, 8, 8), (8, 8, 0)])
def test_bad_shapes_are_refused_with_a_message(args):
    with pytest.raises(ValueError) as e:
        _lib.workspace_layout(*args)
    assert "libsteinhip error" in str(e.value)


def test_null_pointers_are_refused_before_any_launch():
    lib = _lib.load()
    null = ctypes.c_void_p(0)
    assert lib.stein_rownorms(null, 8, 8, _lib.F32, null, null) == _lib.E_BADARG
    assert b"NULL" in lib.stein_last_error()
    assert lib.stein_median_hist_pass(null, 8, 8, 8, 0, null, null, 0, null) == _lib.E_BADARG
    assert lib.stein_apply_adam(null, null, _lib.F32, null, null, 8, _lib.F32, null, 1.0, 10.0, 1e-3, 0.9, 0.999, 1e-8, 1,
                                null, null) == _lib.E_BADARG
    total = ctypes.c_size_t(0)
    assert lib.stein_workspace_bytes(8, 8, 8, 7, 0, ctypes.byref(total)) == _lib.E_UNSUPPORTED


def test_product_package_does_not_import_the_oracle():
    """The oracle is test infrastructure; nothing under stein_amd/ may reference it."""
    for base, _, files in os.walk(os.path.join(ROOT, "stein_amd")):
        for f in files:
            if f.endswith(".py"):
                src = open(os.path.join(base, f)).read()
                assert "oracle" not in src, "%s mentions the oracle" % os.path.join(base, f)


def test_stages_refuse_cpu_tensors():
    import torch
    from stein_amd.engine import HipStages
    t = torch.zeros(8, 4)
    with pytest.raises(RuntimeError, match="no CPU path"):
        HipStages().rownorms(t, 8, 4, torch.zeros(8))


def test_calls_run_with_the_tensor_device_current(monkeypatch):
    """_lib.call_on makes the device of the tensors it is handed the current HIP device for the call (stream 0 and
    kernel attributes belong to the CURRENT device): no switch when it already is, a scoped switch otherwise."""
    import torch
    events = []

    class FakeCtx:
        def __init__(self, idx):
            self.idx = idx

        def __enter__(self):
            events.append(("enter", self.idx))

        def __exit__(self, *a):
            events.append(("exit", self.idx))

    monkeypatch.setattr(torch.cuda, "current_device", lambda: 0)
    monkeypatch.setattr(torch.cuda, "device", FakeCtx)
    monkeypatch.setattr(_lib, "call", lambda name, *a: events.append(("call", name)))
    _lib.call_on(torch.device("cuda", 0), "stein_rownorms")
    assert events == [("call", "stein_rownorms")]
    events.clear()
    _lib.call_on(torch.device("cuda", 1), "stein_rownorms")
    assert events == [("enter", 1), ("call", "stein_rownorms"), ("exit", 1)]
    events.clear()
    _lib.call_on(torch.device("cuda"), "stein_rownorms")                # no index: the current device
    assert events == [("call", "stein_rownorms")]


def test_library_reads_no_environment():
    """The C ABI is steered through its flags only (no hidden global state)."""
    import subprocess
    out = subprocess.run(["nm", "-D", "--undefined-only", _lib.LIB_PATH], capture_output=True, text=True).stdout
    assert "getenv" not in out
    for src in ("steinhip.hip", "stein_x3.hip", "stein_small.hip", "stein_score.hip", "stein_common.h"):
        assert "getenv" not in open(os.path.join(ROOT, "stein_amd", "csrc", src)).read()


def test_build_digest_does_not_depend_on_where_the_tree_lives(monkeypatch):
    """The GPU box unpacks the snapshot under another absolute path; a digest that changed with it would rebuild the
    library there in every process (and bench.py's stdout must stay one JSON line: build chatter goes to stderr)."""
    import __graft_entry__ as ge
    a = ge._digest(ge.SRCS + ge.HDRS, "lib")
    monkeypatch.setattr(ge, "HIPCC_FLAGS", [f if not f.startswith("-I") else "-I/some/other/place/include" for f in ge.HIPCC_FLAGS])
    assert ge._digest(ge.SRCS + ge.HDRS, "lib") == a
    src = open(os.path.join(ROOT, "__graft_entry__.py")).read()
    assert 'print("[build]", " ".join(cmd), flush=True)' not in src


def test_communicator_entry_points_check_their_arguments_without_rccl():
    """stein_comm_* / stein_rank_step refuse bad handles and buffers before RCCL or the GPU is touched."""
    lib = _lib.load()
    null = ctypes.c_void_p(0)
    buf = (ctypes.c_ubyte * 64)()
    assert lib.stein_comm_unique_id(buf, 64) == _lib.E_BADARG                 # the id is STEIN_COMM_ID_BYTES = 128 bytes
    assert b"128" in lib.stein_last_error()
    assert lib.stein_comm_unique_id(null, _lib.COMM_ID_BYTES) == _lib.E_BADARG
    out = ctypes.c_void_p(0)
    big = (ctypes.c_ubyte * _lib.COMM_ID_BYTES)()
    assert lib.stein_comm_init(big, _lib.COMM_ID_BYTES, 2, 2, ctypes.byref(out)) == _lib.E_BADARG   # rank 2 of 2
    assert lib.stein_comm_destroy(null) == _lib.E_BADARG
    fake = (ctypes.c_ubyte * 64)()                                            # not a communicator: wrong magic word
    hit = ctypes.c_int(0)
    rc = lib.stein_rank_step(ctypes.cast(fake, ctypes.c_void_p), null, null, null, null, 8, 8, _lib.F32, null, null, null,
                             null, null, null, 0, 0, ctypes.byref(hit), null)
    assert rc == _lib.E_BADARG and b"communicator" in lib.stein_last_error()


def test_c_host_example_builds_as_plain_c11(tmp_path):
    """include/steinhip.h is a C header and every entry point the C host uses links: examples/c_host/svgd_steps.c
    compiles with gcc -std=c11 -Wall -Werror (no GPU needed to build; tests/test_gpu_c_host.py runs it)."""
    import __graft_entry__ as ge
    exe = ge.build_c_host(str(tmp_path / "svgd_steps"))
    assert os.path.exists(exe) and os.access(exe, os.X_OK)


def test_upper_image_flag_is_refused_where_no_upper_image_can_exist():
    """STEIN_STAGE_UPPER (the distance image holds only the tiles on / above the diagonal) is what the split path's
    symmetric single-rank distance pass leaves; a row block or the fp32-MFMA path never has it -- refused before any launch."""
    lib = _lib.load()
    fake = (ctypes.c_ubyte * 64)()
    f = ctypes.cast(fake, ctypes.c_void_p)
    tot, offs, extra = _lib.workspace_layout(128, 256, 8, _lib.F32, _lib.FLAG_X3 | _lib.FLAG_TILED)
    ld = extra[_lib.WSX_LD_DIST]
    # a row block (row0 = 128 of n = 256)
    assert lib.stein_contract_partial(f, ld, f, f, 256, 8, 128, 128, _lib.F32, f, f, f, tot, _lib.STAGE_UPPER, None) == _lib.E_BADARG
    assert b"STEIN_STAGE_UPPER" in lib.stein_last_error()
    # the fp32-MFMA path (no operand planes)
    assert lib.stein_contract_partial(f, ld, f, f, 256, 8, 0, 256, _lib.F32, f, None, f, 1 << 40, _lib.STAGE_UPPER, None) == _lib.E_BADARG
    # an unknown distance flag
    assert lib.stein_contract_partial(f, ld, f, f, 256, 8, 0, 256, _lib.F32, f, f, f, 1 << 40, 64, None) == _lib.E_BADARG
