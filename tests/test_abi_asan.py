"""CPU: the host side of libsteinhip.so (argument checks, workspace layout arithmetic, error strings) under
AddressSanitizer -- SURVEY.md section 5 ("-fsanitize=address host build of the C-ABI shim").  The sanitized library is
built from the same sources (device code unchanged: GPU ASan is not available on this pool) and driven in a child
process with the ASan runtime preloaded; any heap / stack / global overflow in the host layer aborts the child."""
import os
import subprocess
import sys
import textwrap

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

DRIVER = textwrap.dedent(r'''
    import ctypes, sys
    lib = ctypes.CDLL(sys.argv[1])
    lib.stein_last_error.restype = ctypes.c_char_p
    i64, dbl, szt = ctypes.c_int64, ctypes.c_double, ctypes.c_size_t
    null = ctypes.c_void_p(0)
    ok = 0
    shapes = [(100, 100, 10), (4096, 4096, 128), (16384, 16384, 256), (2048, 16384, 256), (8192, 8192, 2001), (16384, 131072, 256),
              (7, 7, 3), (2, 2, 1), (160, 160, 85), (161, 161, 3), (1, 1, 8), (0, 8, 8), (9, 8, 8), (8, 8, 0), (1 << 31, 1 << 31, 4)]
    for nl, n, d in shapes:
        for dtype in (0, 1, 7):
            for flags in (0, 1, 8, 9, 1 | 16, 1 | 32, 64):
                sz = szt(0)
                offs = (szt * 10)(); extra = (i64 * 4)()
                rc = lib.stein_workspace_bytes(i64(nl), i64(n), i64(d), dtype, flags, ctypes.byref(sz))
                rc2 = lib.stein_workspace_layout(i64(nl), i64(n), i64(d), dtype, flags, offs, extra)
                assert (rc == 0) == (rc2 == 0), (nl, n, d, dtype, flags, rc, rc2)
                if rc == 0:
                    ok += 1
                    assert list(offs) == sorted(offs) and offs[9] <= sz.value
                else:
                    assert rc in (-1, -2, -6) and len(lib.stein_last_error()) > 0
    assert lib.stein_workspace_bytes(i64(8), i64(8), i64(8), 0, 0, None) == -1
    # every entry point refuses NULL pointers / bad shapes before it touches the device
    assert lib.stein_rownorms(null, i64(8), i64(8), 0, null, null) == -1
    assert lib.stein_distance_block(null, null, i64(8), i64(8), i64(0), i64(8), 0, null, i64(64), null, null, 0, null) == -1
    assert lib.stein_median_hist_pass(null, i64(8), i64(8), i64(8), 0, null, null, 0, null) == -1
    assert lib.stein_median_resolve(null, 0, i64(8), null, null, null, null) == -1
    assert lib.stein_spec_begin(null, null, null, i64(64), null) == -1
    assert lib.stein_svgd_phi(null, null, i64(8), i64(8), i64(0), i64(8), 0, null, null, null, null, null, null, szt(0), 0, null) == -1
    assert lib.stein_rank_begin(null, i64(8), i64(8), i64(0), i64(8), 0, null, szt(0), 1, null) == -1
    assert lib.stein_rank_finish(null, null, i64(8), i64(8), i64(0), i64(8), 0, null, null, null, null, null, szt(0), 1, null) == -1
    assert lib.stein_apply_adagrad(null, null, 0, null, i64(8), 0, null, dbl(1.0), dbl(10.0), dbl(1e-3), dbl(0.9), dbl(1e-6), 0, null, null) == -1
    assert lib.stein_apply_adam(null, null, 0, null, null, i64(8), 0, null, dbl(1.0), dbl(10.0), dbl(1e-3), dbl(0.9), dbl(0.999), dbl(1e-8), i64(1), null, null) == -1
    buf = (ctypes.c_char * 64)()
    assert lib.stein_apply_adagrad(buf, buf, 2, buf, i64(4), 0, null, dbl(1.0), dbl(10.0), dbl(1e-3), dbl(0.9), dbl(1e-6), 0, null, null) == -6   # fp64 phi needs fp64 state
    assert lib.stein_apply_adagrad(buf, buf, 0, buf, i64(0), 0, null, dbl(1.0), dbl(10.0), dbl(1e-3), dbl(0.9), dbl(1e-6), 0, null, null) == -2
    out = (ctypes.c_float * 8)(); got = ctypes.c_int(0)
    assert lib.stein_timing_read(out, 1, ctypes.byref(got)) == 0 and got.value == 0
    assert lib.stein_timing_reserve(-1) == -1
    print("ASAN-DRIVE-OK", ok)
''')


def test_host_layer_is_clean_under_address_sanitizer(tmp_path):
    sys.path.insert(0, ROOT)
    import __graft_entry__ as ge
    lib, runtime = ge.build_asan()
    assert os.path.exists(lib) and os.path.exists(runtime)
    drv = tmp_path / "drive.py"
    drv.write_text(DRIVER)
    env = dict(os.environ, LD_PRELOAD=runtime, ASAN_OPTIONS="detect_leaks=0:abort_on_error=1")
    r = subprocess.run([sys.executable, str(drv), lib], capture_output=True, text=True, env=env, timeout=600)
    assert r.returncode == 0 and "ASAN-DRIVE-OK" in r.stdout, (r.returncode, r.stdout[-2000:], r.stderr[-4000:])
    assert "AddressSanitizer" not in r.stderr
