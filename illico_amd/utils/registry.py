"""Dispatch keys and data handlers: the plugin seam the MI355X engine sits behind.

Same public surface as the reference's ``illico/utils/registry.py`` so that its driver code and tests read the
same here: ``Test`` / ``KernelDataFormat`` enums (registry.py:15-23), ``dispatcher_registry.register(test, fmt)`` /
``.get(test, fmt)`` holding the six chunk kernels (registry.py:26-43), ``data_handler_registry.get(X)`` mapping the
Python type of ``X`` onto a handler (registry.py:46-58), and handlers with ``fetch / to_nb / kernel_data_format /
footprint`` (registry.py:67-94).  ``input_signature`` is a Numba notion (HIP code objects are built ahead of time)
and returns ``None``.  Handlers whose ``streams`` flag is set read one gene chunk from storage per ``fetch`` and are
driven through the prefetching chunk pipeline of ``illico_amd.asymptotic_wilcoxon``.
"""
from __future__ import annotations

from collections import namedtuple
from enum import Enum

import numpy as np
from scipy import sparse as _sp

# kernel-side containers of the sparse formats (utils/sparse/csc.py:10, utils/sparse/csr.py:16)
CSCMatrix = namedtuple("CSCMatrix", ["data", "indices", "indptr", "shape"])
CSRMatrix = namedtuple("CSRMatrix", ["data", "indices", "indptr", "shape"])


class Test(Enum):
    OVO = "ovo"
    OVR = "ovr"


class KernelDataFormat(Enum):
    DENSE = "dense"
    CSC = "csc"
    CSR = "csr"


class DispatcherRegistry(dict):
    """``(Test, KernelDataFormat) -> chunk kernel``; enum members or their string values are accepted."""

    @staticmethod
    def _key(test, data_format):
        return Test(test), KernelDataFormat(data_format)

    def register(self, test, data_format):
        key = self._key(test, data_format)

        def _bind(kernel):
            self[key] = kernel
            return kernel

        return _bind

    def get(self, test, data_format):
        key = self._key(test, data_format)
        if key not in self:
            raise KeyError(f"No dispatcher registered for test {test} and data format {data_format}.")
        return self[key]


class DataHandlerRegistry(dict):
    """``type(X) -> DataHandler subclass``; ``get(X)`` instantiates the handler of ``X``'s exact type."""

    def register(self, container_type):
        def _bind(handler_cls):
            self[container_type] = handler_cls
            return handler_cls

        return _bind

    def get(self, X):
        handler_cls = dict.get(self, type(X))
        if handler_cls is None:
            raise KeyError(f"Support for data type {type(X)} is not implemented.")
        return handler_cls(X)


data_handler_registry = DataHandlerRegistry()
dispatcher_registry = DispatcherRegistry()


class DataHandler:
    """Base handler: in-RAM containers hand themselves over whole and let the engine slice on the device."""

    #: True for backed containers: ``fetch`` reads just the requested gene chunk from storage
    streams = False
    fmt: KernelDataFormat = KernelDataFormat.DENSE

    def __init__(self, data):
        self.data = data

    def input_signature(self, *args, **kwargs):
        return None

    def kernel_data_format(self) -> KernelDataFormat:
        return self.fmt

    def fetch(self, lb: int, ub: int) -> tuple:
        """``(container, (lb', ub'))``: the columns to compute inside the returned container (registry.py:97-100)."""
        return self.data, (lb, ub)

    @classmethod
    def to_nb(cls, X):
        return X

    def footprint(self) -> int:
        raise NotImplementedError


InRAMDataHandler = DataHandler  # the reference's name for the same thing


@data_handler_registry.register(np.ndarray)
class DenseDataHandler(DataHandler):
    fmt = KernelDataFormat.DENSE

    def footprint(self) -> int:
        return int(self.data.nbytes)

    @classmethod
    def to_nb(cls, X):
        if not isinstance(X, np.ndarray):
            raise TypeError(f"dense handler got {type(X)}")
        return X


@data_handler_registry.register(np.memmap)
class MemmapDenseDataHandler(DenseDataHandler):
    """Dense matrix backed by a file (``np.load(..., mmap_mode="r")`` / ``np.memmap``): out-of-core like the
    reference's h5py handler (registry.py:162-168) -- one gene chunk is paged in per fetch."""

    streams = True

    def fetch(self, lb: int, ub: int) -> tuple:
        return self.data[:, lb:ub], (0, ub - lb)  # a lazy view: the pages are read when the chunk is staged

    @classmethod
    def to_nb(cls, X):
        return np.asarray(X)


class _CompressedHandler(DataHandler):
    """Shared part of the CSR / CSC handlers: ``(data, indices, indptr, shape)`` goes to the kernel side."""

    container = CSRMatrix

    def footprint(self) -> int:
        m = self.data
        return int(m.data.nbytes + m.indices.nbytes + m.indptr.nbytes)

    @classmethod
    def to_nb(cls, X):
        return cls.container(X.data, X.indices, X.indptr, X.shape)


class CSRDataHandler(_CompressedHandler):
    fmt = KernelDataFormat.CSR
    container = CSRMatrix


class CSCDataHandler(_CompressedHandler):
    fmt = KernelDataFormat.CSC
    container = CSCMatrix


for _name, _handler in (("csr_matrix", CSRDataHandler), ("csc_matrix", CSCDataHandler),
                        ("csr_array", CSRDataHandler), ("csc_array", CSCDataHandler)):
    _type = getattr(_sp, _name, None)
    if _type is not None:
        data_handler_registry[_type] = _handler

try:  # device-resident dense input (a torch.Tensor on the MI355X): no H2D copy inside the call
    import torch as _torch

    @data_handler_registry.register(_torch.Tensor)
    class TorchDenseDataHandler(DataHandler):
        fmt = KernelDataFormat.DENSE

        def footprint(self) -> int:
            return int(self.data.numel() * self.data.element_size())
except Exception:  # pragma: no cover
    pass

# ---- out-of-core handlers of the reference (registry.py:162-188) ------------------------------------------------------
# Duck-typed: anything with ``.shape``, ``.dtype`` and ``[:, lb:ub]`` returning an ndarray (h5py.Dataset) / a scipy CSC matrix
# (anndata's backed ``_CSCDataset``) can be registered under these handlers -- the tests do that with file-backed stand-ins,
# since neither package ships in this image.  ``fetch`` reads ONE gene chunk from storage; the chunk pipeline of
# illico_amd/streaming.py overlaps that read, the upload and the compute of neighbouring chunks.
class H5pyDatasetDataHandler(DenseDataHandler):
    streams = True

    def fetch(self, lb: int, ub: int) -> tuple:
        return self.data[:, lb:ub], (0, ub - lb)

    def footprint(self) -> int:
        return int(np.prod(self.data.shape)) * int(np.dtype(self.data.dtype).itemsize)

    @classmethod
    def to_nb(cls, X):
        return np.asarray(X)


class H5pyBackedCSCDataHandler(CSCDataHandler):
    streams = True

    def footprint(self) -> int:
        d = self.data
        return int(d._data.nbytes + d._indptr.nbytes + d._indices.nbytes)

    def fetch(self, lb: int, ub: int) -> tuple:
        return self.data[:, lb:ub], (0, ub - lb)  # a scipy CSC matrix of the chunk (registry.py:187-188)


try:
    import h5py as _h5py
    data_handler_registry[_h5py.Dataset] = H5pyDatasetDataHandler
except Exception:  # pragma: no cover  (h5py is not in this image)
    pass
try:
    import anndata as _ad
    data_handler_registry[_ad._core.sparse_dataset._CSCDataset] = H5pyBackedCSCDataHandler
except Exception:  # pragma: no cover  (anndata is not in this image)
    pass

# importing the kernel modules registers the six dispatchers (the reference does the same, registry.py:193-202)
from illico_amd import ovo as _ovo  # noqa: E402,F401
from illico_amd import ovr as _ovr  # noqa: E402,F401
