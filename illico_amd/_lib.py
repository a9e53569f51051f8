"""ctypes binding of libillico_hip.so (include/illico_hip.h) and the Engine that owns one context.

The product path has no CPU fallback: if the HIP library is missing or no MI355X is visible the
calls below raise -- nothing here imports the CPU oracle.
"""
from __future__ import annotations

import ctypes
import os
import threading
from pathlib import Path

import numpy as np

_SO = Path(__file__).resolve().parent / "csrc" / "libillico_hip.so"

OK = 0
ERR_ARG, ERR_BOUNDS, ERR_ALTERNATIVE, ERR_DTYPE, ERR_NO_GROUPS, ERR_UNSORTED = -1, -2, -3, -4, -5, -6
ERR_HIP, ERR_OOM, ERR_UNSUPPORTED = -10, -11, -12

F32, F64, I32, I64 = 0, 1, 2, 3
IDX_I32, IDX_I64 = 0, 1
ALTERNATIVES = {"two-sided": 0, "less": 1, "greater": 2}
FLAG_LOG1P, FLAG_CONTINUITY, FLAG_TIE_CORRECT, FLAG_INPUT_DEVICE, FLAG_OUTPUT_DEVICE, FLAG_DEFER = 1, 2, 4, 8, 16, 32

_DTYPES = {np.dtype(np.float32): F32, np.dtype(np.float64): F64, np.dtype(np.int32): I32, np.dtype(np.int64): I64}

# every symbol include/illico_hip.h declares
SYMBOLS = [
    "illico_ctx_create", "illico_ctx_destroy", "illico_ctx_set_stream", "illico_ctx_set_option",
    "illico_last_error", "illico_ctx_synchronize", "illico_set_groups", "illico_run_dense", "illico_run_csc",
    "illico_run_csr", "illico_csr_indices_sorted", "illico_rank_statistics", "illico_profile_num_kernels", "illico_profile_kernel_name",
    "illico_profile_get", "illico_profile_reset", "illico_version", "illico_csr_bind", "illico_csc_bind", "illico_run_bound",
    "illico_matrix_release", "illico_matrix_touch", "illico_profile_input_bytes", "illico_planes_to_host",
]

_lib = None
_lock = threading.Lock()


def load() -> ctypes.CDLL:
    """Load libillico_hip.so; raises if it has not been built (python -m illico_amd.csrc.build)."""
    global _lib
    with _lock:
        if _lib is not None:
            return _lib
        if not _SO.exists():
            raise ImportError(
                f"{_SO} is missing: the HIP engine has not been built. Run `python __graft_entry__.py` or "
                "`python illico_amd/csrc/build.py`. There is no CPU fallback.")
        lib = ctypes.CDLL(str(_SO))
        vp, i64, ci = ctypes.c_void_p, ctypes.c_int64, ctypes.c_int
        lib.illico_ctx_create.argtypes = [ci, ctypes.POINTER(vp)]
        lib.illico_ctx_destroy.argtypes = [vp]
        lib.illico_ctx_set_stream.argtypes = [vp, vp]
        lib.illico_ctx_set_option.argtypes = [vp, ctypes.c_char_p, i64]
        lib.illico_last_error.argtypes = [vp]
        lib.illico_last_error.restype = ctypes.c_char_p
        lib.illico_ctx_synchronize.argtypes = [vp]
        lib.illico_set_groups.argtypes = [vp, vp, vp, vp, vp, i64, i64, i64]
        lib.illico_run_dense.argtypes = [vp, vp, ci, i64, i64, i64, i64, i64, ci, ci, vp, vp, vp, i64]
        for f in (lib.illico_run_csc, lib.illico_run_csr):
            f.argtypes = [vp, vp, ci, vp, vp, ci, i64, i64, i64, i64, ci, ci, vp, vp, vp, i64]
        lib.illico_csr_indices_sorted.argtypes = [vp, vp, vp, ci, i64, ci, ctypes.POINTER(ci)]
        lib.illico_rank_statistics.argtypes = [vp, vp, ci, i64, i64, i64, i64, i64, ci, vp, vp, vp]
        lib.illico_profile_num_kernels.argtypes = []
        lib.illico_profile_kernel_name.argtypes = [ci]
        lib.illico_profile_kernel_name.restype = ctypes.c_char_p
        lib.illico_profile_get.argtypes = [vp, ci, ctypes.POINTER(ctypes.c_double), ctypes.POINTER(i64)]
        lib.illico_profile_reset.argtypes = [vp]
        lib.illico_version.restype = ctypes.c_char_p
        for f in (lib.illico_csr_bind, lib.illico_csc_bind):
            f.argtypes = [vp, vp, ci, vp, vp, ci, i64, i64, ci, ctypes.POINTER(vp)]
        lib.illico_run_bound.argtypes = [vp, vp, i64, i64, ci, ci, vp, vp, vp, i64]
        lib.illico_matrix_release.argtypes = [vp, vp]
        lib.illico_matrix_touch.argtypes = [vp, vp]
        lib.illico_profile_input_bytes.argtypes = [vp, ctypes.POINTER(i64)]
        lib.illico_planes_to_host.argtypes = [vp, vp, vp, vp, i64, vp, vp, vp, i64]
        for name in SYMBOLS:  # fail at load time, not at first use, if the library and the header have drifted
            getattr(lib, name)
        _lib = lib
        return lib


def _raise(code: int, msg: str):
    if code in (ERR_ARG, ERR_BOUNDS, ERR_ALTERNATIVE, ERR_NO_GROUPS, ERR_UNSORTED):
        raise ValueError(msg)
    if code == ERR_DTYPE:
        raise KeyError(msg)
    if code == ERR_OOM:
        raise MemoryError(msg)
    if code == ERR_UNSUPPORTED:
        raise NotImplementedError(msg)
    raise RuntimeError(f"illico_hip error {code}: {msg}")


def _is_torch_tensor(x) -> bool:
    return type(x).__module__.startswith("torch") and hasattr(x, "data_ptr")


class _Buf:
    """A host ndarray or a device torch.Tensor seen as (pointer, on_device, keepalive)."""

    def __init__(self, x, dtype=None):
        if _is_torch_tensor(x):
            if not x.is_contiguous():
                raise ValueError("device tensors must be contiguous")
            self.on_device = x.is_cuda
            self.ptr = x.data_ptr()
            self.keep = x
            self.np_dtype = np.dtype(str(x.dtype).replace("torch.", ""))
        else:
            a = np.ascontiguousarray(x, dtype=dtype)
            self.on_device = False
            self.ptr = a.ctypes.data
            self.keep = a
            self.np_dtype = a.dtype


def dtype_code(np_dtype) -> int:
    try:
        return _DTYPES[np.dtype(np_dtype)]
    except KeyError as e:
        raise KeyError(f"Support for element dtype {np_dtype} is not implemented.") from e


def normalize_values(a: np.ndarray) -> np.ndarray:
    """Host arrays of dtypes the engine has no kernel for are widened losslessly (order and equality kept)."""
    dt = a.dtype
    if dt in _DTYPES:
        return a
    if dt == np.float16:
        return a.astype(np.float32)
    if dt.kind == "b" or (dt.kind in "iu" and dt.itemsize < 4):
        return a.astype(np.int32)
    if dt == np.uint32:
        return a.astype(np.int64)
    if dt == np.uint64:
        if a.size and a.max() > np.iinfo(np.int64).max:
            raise KeyError("uint64 values above 2**63-1 are not supported.")
        return a.astype(np.int64)
    raise KeyError(f"Support for element dtype {dt} is not implemented.")


class Engine:
    """One illico_ctx: one device, one stream, device scratch, the current GroupContainer."""

    def __init__(self, device: int | None = None):
        self.lib = load()
        if device is None:
            device = _current_device()
        self.device = int(device)
        h = ctypes.c_void_p()
        rc = self.lib.illico_ctx_create(self.device, ctypes.byref(h))
        if rc != OK or not h.value:
            raise RuntimeError(
                f"illico_ctx_create(device={self.device}) failed with code {rc}: no usable MI355X/HIP device. "
                "The engine has no CPU fallback.")
        self.h = h
        self._groups_key = None
        self._groups_keep = None
        self._stream = None  # None = the context's own non-blocking stream; else the hipStream_t it was bound to

    def close(self):
        if getattr(self, "h", None) is not None and self.h.value:
            self.lib.illico_ctx_destroy(self.h)
            self.h = ctypes.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _check(self, rc: int):
        if rc != OK:
            _raise(rc, (self.lib.illico_last_error(self.h) or b"").decode())

    def set_option(self, key: str, value: int):
        self._check(self.lib.illico_ctx_set_option(self.h, key.encode(), int(value)))

    def set_stream(self, stream_ptr: int):
        self._check(self.lib.illico_ctx_set_stream(self.h, ctypes.c_void_p(stream_ptr)))
        self._stream = int(stream_ptr)

    def _bind_torch_stream(self, *tensors):
        """Device tensors as inputs or outputs: run this call on torch's CURRENT stream of the engine's device.

        The engine's kernels are then ordered after whatever produced the inputs on that stream (``X = torch.log1p(Y)``
        just before the call) and before whatever consumes device-resident output planes on it -- the same contract
        every torch op gives.  Host arrays need nothing: the C side synchronises before it returns them.
        """
        if not any(_is_torch_tensor(t) and t.is_cuda for t in tensors):
            return
        import torch
        s = int(torch.cuda.current_stream(self.device).cuda_stream)
        if s != self._stream:
            self.set_stream(s)

    def synchronize(self):
        self._check(self.lib.illico_ctx_synchronize(self.h))

    # ---- groups (GroupContainer of illico/utils/groups.py:6-15) ----
    def set_groups(self, grpc):
        key = id(grpc)
        if self._groups_key == key and self._groups_keep is grpc:
            return
        enc = np.ascontiguousarray(grpc.encoded_groups, dtype=np.int64)
        cnt = np.ascontiguousarray(grpc.counts, dtype=np.int64)
        idx = np.ascontiguousarray(grpc.indices, dtype=np.int64)
        ptr = np.ascontiguousarray(grpc.indptr, dtype=np.int64)
        self._check(self.lib.illico_set_groups(self.h, enc.ctypes.data, cnt.ctypes.data, idx.ctypes.data,
                                               ptr.ctypes.data, enc.size, cnt.size, int(grpc.encoded_ref_group)))
        self._groups_key, self._groups_keep = key, grpc
        self.n_groups = int(cnt.size)

    @staticmethod
    def _flags(is_log1p, use_continuity, tie_correct):
        return (FLAG_LOG1P if is_log1p else 0) | (FLAG_CONTINUITY if use_continuity else 0) | \
               (FLAG_TIE_CORRECT if tie_correct else 0)

    @staticmethod
    def _alt(alternative):
        try:
            return ALTERNATIVES[alternative]
        except KeyError:
            raise ValueError(f"Unsupported alternative hypothesis: {alternative}") from None

    def _outputs(self, out, G, W, want_device):
        """out: None (allocate host planes), a tuple of three host ndarrays (views with a common row stride are
        fine) or three device tensors."""
        if out is None:
            if want_device:
                import torch
                planes = tuple(torch.empty((G, W), dtype=torch.float64, device=f"cuda:{self.device}") for _ in range(3))
            else:
                planes = tuple(np.empty((G, W), dtype=np.float64) for _ in range(3))
        else:
            planes = tuple(out)
        if W == 0:  # empty chunk (lb == ub is legal, asymptotic_wilcoxon.py:49): nothing to compute
            return planes, None, 0, 1
        ptrs, flag, ld = [], 0, None
        for p in planes:
            if _is_torch_tensor(p):
                if p.dtype != __import__("torch").float64 or p.dim() != 2 or p.stride(1) != 1:
                    raise ValueError("device output planes must be float64 [G, W] with unit column stride")
                l = p.stride(0)
                ptrs.append(p.data_ptr())
                flag = FLAG_OUTPUT_DEVICE if p.is_cuda else 0
            else:
                if p.dtype != np.float64 or p.ndim != 2 or p.strides[1] != 8 or p.shape != (G, W):
                    raise ValueError("output planes must be float64 [G, W] with unit column stride")
                l = p.strides[0] // 8 if G > 1 else max(W, p.strides[0] // 8)
                ptrs.append(p.ctypes.data)
            if ld is None:
                ld = l
            elif ld != l:
                raise ValueError("output planes must share one row stride")
        return planes, ptrs, flag, int(ld if ld else max(W, 1))

    def run_dense(self, X, col_lb, col_ub, *, is_log1p=False, use_continuity=True, tie_correct=True,
                  alternative="two-sided", out=None, device_out=False, defer=False):
        """``defer=True`` (device input and device planes only): return once the pass is enqueued; the planes are complete
        after ``synchronize()`` or the next call on this engine (ILLICO_FLAG_DEFER, include/illico_hip.h)."""
        alt = self._alt(alternative)
        if _is_torch_tensor(X):
            if X.dim() != 2 or X.stride(1) != 1:
                raise ValueError("X must be row-major 2-D")
            buf_ptr, on_dev, keep = X.data_ptr(), X.is_cuda, X
            n_rows, n_cols, ld = X.shape[0], X.shape[1], X.stride(0)
            dt = dtype_code(str(X.dtype).replace("torch.", ""))
        else:
            X = normalize_values(np.asarray(X))
            if X.ndim != 2 or (X.shape[1] > 1 and X.strides[1] != X.itemsize):
                X = np.ascontiguousarray(X)
            buf_ptr, on_dev, keep = X.ctypes.data, False, X
            n_rows, n_cols = X.shape
            ld = X.strides[0] // X.itemsize if n_rows > 1 else n_cols
            dt = dtype_code(X.dtype)
        if col_lb < 0 or col_ub > n_cols or col_lb > col_ub:
            raise ValueError(f"Invalid chunk bounds: {(col_lb, col_ub)} for data with {n_cols} columns.")
        G, W = self.n_groups, col_ub - col_lb
        planes, ptrs, oflag, out_ld = self._outputs(out, G, W, device_out)
        if ptrs is None:
            return planes
        flags = self._flags(is_log1p, use_continuity, tie_correct) | (FLAG_INPUT_DEVICE if on_dev else 0) | oflag | \
            (FLAG_DEFER if defer else 0)
        self._bind_torch_stream(keep, *planes)
        self._check(self.lib.illico_run_dense(self.h, buf_ptr, dt, n_rows, n_cols, ld, col_lb, col_ub, flags, alt,
                                              ptrs[0], ptrs[1], ptrs[2], out_ld))
        del keep
        return planes

    def run_sparse(self, fmt, data, indices, indptr, shape, col_lb, col_ub, *, is_log1p=False, use_continuity=True,
                   tie_correct=True, alternative="two-sided", out=None, device_out=False, defer=False):
        """``defer=True`` (device-resident CSC arrays and device planes only; ignored elsewhere): return once the count-valued
        pass is enqueued; the planes are complete after ``synchronize()`` or the next call on this engine."""
        alt = self._alt(alternative)
        n_rows, n_cols = int(shape[0]), int(shape[1])
        if _is_torch_tensor(data):
            d, i, p = _Buf(data), _Buf(indices), _Buf(indptr)
        else:
            d = _Buf(normalize_values(np.asarray(data)))
            idt = np.int32 if (np.asarray(indices).dtype == np.int32 and np.asarray(indptr).dtype == np.int32) else np.int64
            i, p = _Buf(indices, idt), _Buf(indptr, idt)
        if i.np_dtype != p.np_dtype or i.np_dtype not in (np.dtype(np.int32), np.dtype(np.int64)):
            raise KeyError(f"Support for index dtypes {i.np_dtype}/{p.np_dtype} is not implemented.")
        if not (d.on_device == i.on_device == p.on_device):
            raise ValueError("data, indices and indptr must live on the same side (host or device)")
        if col_lb < 0 or col_ub > n_cols or col_lb > col_ub:
            raise ValueError(f"Invalid chunk bounds: {(col_lb, col_ub)} for data with {n_cols} columns.")
        G, W = self.n_groups, col_ub - col_lb
        planes, ptrs, oflag, out_ld = self._outputs(out, G, W, device_out)
        if ptrs is None:
            return planes
        flags = self._flags(is_log1p, use_continuity, tie_correct) | (FLAG_INPUT_DEVICE if d.on_device else 0) | oflag | \
            (FLAG_DEFER if defer else 0)
        fn = self.lib.illico_run_csc if fmt == "csc" else self.lib.illico_run_csr
        self._bind_torch_stream(d.keep, i.keep, p.keep, *planes)
        self._check(fn(self.h, d.ptr, dtype_code(d.np_dtype), i.ptr, p.ptr, IDX_I32 if i.np_dtype == np.int32 else IDX_I64,
                       n_rows, n_cols, col_lb, col_ub, flags, alt, ptrs[0], ptrs[1], ptrs[2], out_ld))
        return planes

    def bind_sparse(self, fmt, data, indices, indptr, shape):
        """Upload a host CSR / CSC matrix once (or adopt device tensors) -- illico_csr_bind / illico_csc_bind; returns a
        ``BoundMatrix`` whose ``run(col_lb, col_ub, ...)`` computes chunks without moving the matrix again."""
        n_rows, n_cols = int(shape[0]), int(shape[1])
        if _is_torch_tensor(data):
            d, i, p = _Buf(data), _Buf(indices), _Buf(indptr)
        else:
            d = _Buf(normalize_values(np.asarray(data)))
            idt = np.int32 if (np.asarray(indices).dtype == np.int32 and np.asarray(indptr).dtype == np.int32) else np.int64
            i, p = _Buf(indices, idt), _Buf(indptr, idt)
        if i.np_dtype != p.np_dtype or i.np_dtype not in (np.dtype(np.int32), np.dtype(np.int64)):
            raise KeyError(f"Support for index dtypes {i.np_dtype}/{p.np_dtype} is not implemented.")
        if not (d.on_device == i.on_device == p.on_device):
            raise ValueError("data, indices and indptr must live on the same side (host or device)")
        h = ctypes.c_void_p()
        fn = self.lib.illico_csc_bind if fmt == "csc" else self.lib.illico_csr_bind
        self._bind_torch_stream(d.keep, i.keep, p.keep)
        self._check(fn(self.h, d.ptr, dtype_code(d.np_dtype), i.ptr, p.ptr, IDX_I32 if i.np_dtype == np.int32 else IDX_I64,
                       n_rows, n_cols, FLAG_INPUT_DEVICE if d.on_device else 0, ctypes.byref(h)))
        return BoundMatrix(self, h, (n_rows, n_cols), (d.keep, i.keep, p.keep) if d.on_device else None)

    def input_bytes(self) -> int:
        """Matrix bytes copied host -> device by this context so far (illico_profile_input_bytes)."""
        n = ctypes.c_int64(0)
        self._check(self.lib.illico_profile_input_bytes(self.h, ctypes.byref(n)))
        return int(n.value)

    def planes_to_host(self, planes, out=None):
        """A contiguous device tensor ``[3, n_groups, n_cols]`` of float64 planes -> a host ndarray of the same shape, through the
        context's pinned double buffer (include/illico_hip.h: illico_planes_to_host): what the gathering rank of a multi-GPU call does
        ONCE with everything it has received."""
        if not (_is_torch_tensor(planes) and planes.is_cuda and planes.is_contiguous() and planes.dim() == 3 and planes.shape[0] == 3):
            raise ValueError("planes_to_host wants a contiguous CUDA tensor of shape [3, n_groups, n_cols]")
        _, G, W = (int(x) for x in planes.shape)
        import torch
        if planes.dtype != torch.float64:
            raise ValueError(f"planes_to_host wants float64 planes, got {planes.dtype}")
        if G != getattr(self, "n_groups", -1):  # the library copies n_groups rows per plane: the groups last set on this engine
            raise ValueError(f"planes hold {G} groups, the engine's groups are {getattr(self, 'n_groups', None)} (set_groups first)")
        if out is None:
            out = np.empty((3, G, W), dtype=np.float64)
        elif not (isinstance(out, np.ndarray) and out.dtype == np.float64 and out.shape == (3, G, W) and out.strides[2] == 8
                  and out.strides[1] % 8 == 0 and out.strides[1] >= 8 * W and out.flags.writeable):
            raise ValueError("out must be a writeable float64 ndarray [3, n_groups, n_cols] whose rows are contiguous")
        self._bind_torch_stream(planes)
        esz = planes.element_size()
        base = planes.data_ptr()
        self._check(self.lib.illico_planes_to_host(self.h, base, base + G * W * esz, base + 2 * G * W * esz, W,
                                                   out[0].ctypes.data, out[1].ctypes.data, out[2].ctypes.data, out.strides[1] // 8))
        return out

    def rank_statistics(self, X, col_lb, col_ub, *, is_log1p=False):
        """The ranking primitives before finalisation (include/illico_hip.h: illico_rank_statistics):
        ``(two_u int64 [W, G], tie_sum uint64 [W, G], value_sum float64 [W, G])`` for the dense columns [col_lb, col_ub)."""
        if _is_torch_tensor(X):
            if X.dim() != 2 or X.stride(1) != 1:
                raise ValueError("X must be row-major 2-D")
            ptr, on_dev, keep = X.data_ptr(), X.is_cuda, X
            n_rows, n_cols, ld = X.shape[0], X.shape[1], X.stride(0)
            dt = dtype_code(str(X.dtype).replace("torch.", ""))
        else:
            X = np.ascontiguousarray(normalize_values(np.asarray(X)))
            ptr, on_dev, keep = X.ctypes.data, False, X
            (n_rows, n_cols), ld, dt = X.shape, X.shape[1], dtype_code(X.dtype)
        W, G = col_ub - col_lb, self.n_groups
        two_u, tie, vsum = np.zeros((W, G), np.int64), np.zeros((W, G), np.uint64), np.zeros((W, G), np.float64)
        self._bind_torch_stream(keep)
        flags = (FLAG_LOG1P if is_log1p else 0) | (FLAG_INPUT_DEVICE if on_dev else 0)
        self._check(self.lib.illico_rank_statistics(self.h, ptr, dt, n_rows, n_cols, ld, col_lb, col_ub, flags,
                                                    two_u.ctypes.data, tie.ctypes.data, vsum.ctypes.data))
        del keep
        return two_u, tie, vsum

    def csr_indices_sorted(self, indices, indptr, n_rows) -> bool:
        i, p = _Buf(indices), _Buf(indptr)
        if i.np_dtype != p.np_dtype:
            i, p = _Buf(np.asarray(indices), np.int64), _Buf(np.asarray(indptr), np.int64)
        res = ctypes.c_int(0)
        self._bind_torch_stream(i.keep, p.keep)
        self._check(self.lib.illico_csr_indices_sorted(self.h, i.ptr, p.ptr, IDX_I32 if i.np_dtype == np.int32 else IDX_I64,
                                                       int(n_rows), FLAG_INPUT_DEVICE if i.on_device else 0,
                                                       ctypes.byref(res)))
        return bool(res.value)

    # ---- measurement hooks ----
    def profile(self, on: bool = True):
        self.set_option("profile", 1 if on else 0)

    def profile_only(self, kernel_name=None):
        """Restrict event timing to one kernel (by name); ``None`` times every kernel again."""
        kid = -1
        if kernel_name is not None:
            names = [self.lib.illico_profile_kernel_name(k).decode() for k in range(self.lib.illico_profile_num_kernels())]
            kid = names.index(kernel_name)
        self.set_option("profile_only", kid)

    def profile_reset(self):
        self._check(self.lib.illico_profile_reset(self.h))

    def profile_get(self) -> dict:
        out = {}
        for k in range(self.lib.illico_profile_num_kernels()):
            ms, n = ctypes.c_double(0), ctypes.c_int64(0)
            self._check(self.lib.illico_profile_get(self.h, k, ctypes.byref(ms), ctypes.byref(n)))
            if n.value:
                out[self.lib.illico_profile_kernel_name(k).decode()] = {"ms": ms.value, "launches": n.value}
        return out


class BoundMatrix:
    """Handle of illico_csr_bind / illico_csc_bind; ``release()`` (or garbage collection) frees the device copy."""

    def __init__(self, engine, handle, shape, keep):
        self.engine, self.h, self.shape, self._keep = engine, handle, shape, keep

    def run(self, col_lb, col_ub, *, is_log1p=False, use_continuity=True, tie_correct=True, alternative="two-sided", out=None,
            device_out=False, defer=False):
        eng = self.engine
        alt = eng._alt(alternative)
        n_cols = self.shape[1]
        if col_lb < 0 or col_ub > n_cols or col_lb > col_ub:
            raise ValueError(f"Invalid chunk bounds: {(col_lb, col_ub)} for data with {n_cols} columns.")
        G, W = eng.n_groups, col_ub - col_lb
        planes, ptrs, oflag, out_ld = eng._outputs(out, G, W, device_out)
        if ptrs is None:
            return planes
        flags = eng._flags(is_log1p, use_continuity, tie_correct) | oflag | (FLAG_DEFER if defer else 0)
        eng._bind_torch_stream(*planes)
        eng._check(eng.lib.illico_run_bound(eng.h, self.h, col_lb, col_ub, flags, alt, ptrs[0], ptrs[1], ptrs[2], out_ld))
        return planes

    def touch(self):
        """Adopted device arrays were rewritten in place: forget what the context remembers about them (include/illico_hip.h)."""
        self.engine._check(self.engine.lib.illico_matrix_touch(self.engine.h, self.h))

    def release(self):
        if self.h is not None and self.h.value and getattr(self.engine, "h", None) is not None and self.engine.h.value:
            self.engine.lib.illico_matrix_release(self.engine.h, self.h)
        self.h = None

    def __del__(self):
        try:
            self.release()
        except Exception:
            pass


def _current_device() -> int:
    try:
        import torch
        if torch.cuda.is_available():
            return torch.cuda.current_device()
    except Exception:
        pass
    return 0


_engines = threading.local()


def get_engine(device: int | None = None) -> Engine:
    """Per-thread, per-device engine (a context is single-threaded, include/illico_hip.h)."""
    if device is None:
        device = _current_device()
    cache = getattr(_engines, "cache", None)
    if cache is None:
        cache = _engines.cache = {}
    if device not in cache:
        eng = cache[device] = Engine(device)
        if os.environ.get("ILLICO_PREWARM", "1") != "0":
            # The first call on a host-resident dense matrix pins three staging slots and allocates three device windows (~80 ms of a
            # 190-ms first drop-in call at C2 shape): done here, on a thread of its own, while the caller is still encoding groups.  (The
            # context's lock orders it before the first engine call if that comes sooner.)
            def prewarm(e=eng):
                try:
                    e.set_option("prewarm_host_window_bytes", 256 << 20)
                except Exception:
                    pass
            threading.Thread(target=prewarm, name="illico-prewarm", daemon=True).start()
    return cache[device]
