"""The C-ABI called directly through ctypes (no Engine wrapper): status codes, error strings, profiling hooks."""
import ctypes

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def test_cabi_status_codes_and_profile_hooks():
    from illico_amd import _lib
    lib = _lib.load()
    vp = ctypes.c_void_p
    ctx = vp()
    assert lib.illico_ctx_create(0, ctypes.byref(ctx)) == 0 and ctx.value
    assert lib.illico_ctx_create(10_000, ctypes.byref(vp())) == _lib.ERR_ARG
    err = lambda: lib.illico_last_error(ctx).decode()

    X = np.arange(40, dtype=np.float32).reshape(10, 4)
    outs = [np.empty((2, 4)) for _ in range(3)]
    po = [o.ctypes.data for o in outs]
    # no groups yet
    rc = lib.illico_run_dense(ctx, X.ctypes.data, _lib.F32, 10, 4, 4, 0, 4, 6, 0, *po, 4)
    assert rc == _lib.ERR_NO_GROUPS and "illico_set_groups" in err()
    enc = np.array([0, 1] * 5, dtype=np.int64)
    cnt = np.array([5, 5], dtype=np.int64)
    idx = np.argsort(enc, kind="stable").astype(np.int64)
    ptr = np.array([0, 5, 10], dtype=np.int64)
    gargs = (enc.ctypes.data, cnt.ctypes.data, idx.ctypes.data, ptr.ctypes.data, 10, 2)
    assert lib.illico_set_groups(ctx, *gargs, 7) == _lib.ERR_ARG                    # reference out of range
    bad_ptr = np.array([0, 4, 10], dtype=np.int64)
    assert lib.illico_set_groups(ctx, enc.ctypes.data, cnt.ctypes.data, idx.ctypes.data, bad_ptr.ctypes.data, 10, 2, 0) == _lib.ERR_ARG
    assert lib.illico_set_groups(ctx, *gargs, 0) == 0
    assert lib.illico_run_dense(ctx, X.ctypes.data, _lib.F32, 9, 4, 4, 0, 4, 6, 0, *po, 4) == _lib.ERR_NO_GROUPS   # row count mismatch
    assert lib.illico_run_dense(ctx, X.ctypes.data, _lib.F32, 10, 4, 4, 1, 5, 6, 0, *po, 4) == _lib.ERR_BOUNDS
    assert "Invalid chunk bounds" in err()
    assert lib.illico_run_dense(ctx, X.ctypes.data, _lib.F32, 10, 4, 4, 0, 4, 6, 5, *po, 4) == _lib.ERR_ALTERNATIVE
    assert lib.illico_run_dense(ctx, X.ctypes.data, 9, 10, 4, 4, 0, 4, 6, 0, *po, 4) == _lib.ERR_DTYPE
    assert lib.illico_run_dense(ctx, None, _lib.F32, 10, 4, 4, 0, 4, 6, 0, *po, 4) == _lib.ERR_ARG
    assert lib.illico_run_dense(ctx, X.ctypes.data, _lib.F32, 10, 4, 4, 0, 4, 6, 0, po[0], po[1], None, 4) == _lib.ERR_ARG
    assert lib.illico_run_dense(ctx, X.ctypes.data, _lib.F32, 10, 4, 4, 0, 4, 6, 0, *po, 3) == _lib.ERR_ARG        # out_ld < width
    assert lib.illico_run_csc(ctx, X.ctypes.data, _lib.F32, X.ctypes.data, X.ctypes.data, 7, 10, 4, 0, 4, 6, 0, *po, 4) == _lib.ERR_DTYPE
    assert lib.illico_ctx_set_option(ctx, b"no_such_option", 1) == _lib.ERR_ARG
    # a good call, with the measurement hooks on
    assert lib.illico_ctx_set_option(ctx, b"profile", 1) == 0
    assert lib.illico_profile_reset(ctx) == 0
    assert lib.illico_run_dense(ctx, X.ctypes.data, _lib.F32, 10, 4, 4, 0, 4, 6, 0, *po, 4) == 0
    assert np.all(outs[0][0] == 1.0) and np.all(outs[1][0] == -1.0)       # reference row
    assert np.all((outs[0][1] > 0) & (outs[0][1] <= 1))
    seen = {}
    for k in range(lib.illico_profile_num_kernels()):
        ms, n = ctypes.c_double(), ctypes.c_int64()
        assert lib.illico_profile_get(ctx, k, ctypes.byref(ms), ctypes.byref(n)) == 0
        if n.value:
            seen[lib.illico_profile_kernel_name(k).decode()] = (ms.value, n.value)
    assert seen and all(ms > 0 and n >= 1 for ms, n in seen.values())
    assert lib.illico_profile_get(ctx, 10_000, None, None) == _lib.ERR_ARG
    assert lib.illico_ctx_synchronize(ctx) == 0
    assert lib.illico_ctx_destroy(ctx) == 0
