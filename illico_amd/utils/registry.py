"""Dispatch keys and data handlers -- the reference's plugin seam (illico/utils/registry.py).

``dispatcher_registry[(Test, KernelDataFormat)]`` holds the six chunk kernels with the reference's
dispatcher signature (registry.py:26-43); ``data_handler_registry`` maps the Python type of ``X``
onto a handler (registry.py:46-58).  Handlers keep the reference's contract
(``fetch / to_nb / kernel_data_format / footprint``, registry.py:67-94); ``input_signature`` is a
Numba notion and returns ``None`` here (HIP code objects are built ahead of time).
"""
from __future__ import annotations

from abc import ABC, abstractmethod
from collections import namedtuple
from enum import Enum
from typing import Any

import numpy as np
from scipy import sparse as py_sparse

CSCMatrix = namedtuple("CSCMatrix", ["data", "indices", "indptr", "shape"])  # utils/sparse/csc.py:10
CSRMatrix = namedtuple("CSRMatrix", ["data", "indices", "indptr", "shape"])  # utils/sparse/csr.py:16


class Test(Enum):
    OVO = "ovo"
    OVR = "ovr"


class KernelDataFormat(Enum):
    DENSE = "dense"
    CSC = "csc"
    CSR = "csr"


class DispatcherRegistry(dict):
    def register(self, test: Test, data_format: KernelDataFormat):
        test = Test(test)
        data_format = KernelDataFormat(data_format)

        def decorator(obj):
            self[(test, data_format)] = obj
            return obj

        return decorator

    def get(self, test: Test, data_format: KernelDataFormat):
        key = (Test(test), KernelDataFormat(data_format))
        try:
            return self[key]
        except KeyError as e:
            raise KeyError(f"No dispatcher registered for test {test} and data format {data_format}.") from e


class DataHandlerRegistry(dict):
    def register(self, data_format):
        def decorator(obj):
            self[data_format] = obj
            return obj

        return decorator

    def get(self, key):
        try:
            return self[type(key)](key)
        except KeyError as e:
            raise KeyError(f"Support for data type {type(key)} is not implemented.") from e


data_handler_registry = DataHandlerRegistry()
dispatcher_registry = DispatcherRegistry()


class DataHandler(ABC):
    def __init__(self, data):
        self.data = data

    def input_signature(self, *args, **kwargs):
        return None

    @abstractmethod
    def fetch(self, lb: int, ub: int) -> tuple:
        """Return (data, (lb', ub')) -- registry.py:97-100,164-165,187-188."""

    @abstractmethod
    def to_nb(self, X) -> Any:
        """Convert to the kernel-side container."""

    @abstractmethod
    def kernel_data_format(self) -> KernelDataFormat:
        pass

    @abstractmethod
    def footprint(self) -> int:
        pass


class InRAMDataHandler(DataHandler):
    #: backed handlers (h5py / anndata backed / np.memmap) set this: fetch() reads one gene chunk from storage and
    #: the driver streams chunks with a prefetch thread instead of handing the whole range to the engine
    streams = False

    def fetch(self, lb: int, ub: int) -> tuple:
        return self.data, (lb, ub)


@data_handler_registry.register(np.ndarray)
class DenseDataHandler(InRAMDataHandler):
    def kernel_data_format(self) -> KernelDataFormat:
        return KernelDataFormat.DENSE

    def footprint(self) -> int:
        return self.data.nbytes

    @classmethod
    def to_nb(cls, X: np.ndarray) -> np.ndarray:
        assert isinstance(X, np.ndarray)
        return X


@data_handler_registry.register(py_sparse.csr_matrix)
class CSRDataHandler(InRAMDataHandler):
    @classmethod
    def to_nb(cls, X) -> CSRMatrix:
        return CSRMatrix(X.data, X.indices, X.indptr, X.shape)

    def kernel_data_format(self) -> KernelDataFormat:
        return KernelDataFormat.CSR

    def footprint(self) -> int:
        return self.data.data.nbytes + self.data.indptr.nbytes + self.data.indices.nbytes


@data_handler_registry.register(py_sparse.csc_matrix)
class CSCDataHandler(InRAMDataHandler):
    @classmethod
    def to_nb(cls, X) -> CSCMatrix:
        return CSCMatrix(X.data, X.indices, X.indptr, X.shape)

    def kernel_data_format(self) -> KernelDataFormat:
        return KernelDataFormat.CSC

    def footprint(self) -> int:
        return self.data.data.nbytes + self.data.indptr.nbytes + self.data.indices.nbytes


@data_handler_registry.register(np.memmap)
class MemmapDenseDataHandler(DenseDataHandler):
    """Dense matrix backed by a file (``np.load(..., mmap_mode="r")`` / ``np.memmap``): out-of-core like the
    reference's h5py handler (registry.py:162-168) -- one gene chunk is paged in per fetch."""
    streams = True

    def fetch(self, lb: int, ub: int) -> tuple:
        return np.ascontiguousarray(self.data[:, lb:ub]), (0, ub - lb)

    @classmethod
    def to_nb(cls, X) -> np.ndarray:
        return np.asarray(X)


for _name in ("csr_array", "csc_array"):  # scipy's array API twins
    _t = getattr(py_sparse, _name, None)
    if _t is not None:
        data_handler_registry[_t] = CSRDataHandler if _name.startswith("csr") else CSCDataHandler

try:  # device-resident dense input (torch.Tensor on the MI355X): no H2D copy inside the call
    import torch as _torch

    @data_handler_registry.register(_torch.Tensor)
    class TorchDenseDataHandler(InRAMDataHandler):
        def kernel_data_format(self) -> KernelDataFormat:
            return KernelDataFormat.DENSE

        def footprint(self) -> int:
            return self.data.numel() * self.data.element_size()

        @classmethod
        def to_nb(cls, X):
            return X
except Exception:  # pragma: no cover
    pass

try:  # out-of-core handlers (registry.py:162-188) when h5py / anndata are installed
    import h5py as _h5py

    @data_handler_registry.register(_h5py.Dataset)
    class H5pyDatasetDataHandler(DenseDataHandler):
        streams = True

        def fetch(self, lb: int, ub: int) -> tuple:
            return self.data[:, lb:ub], (0, ub - lb)
except Exception:  # pragma: no cover
    pass

try:
    import anndata as _ad

    @data_handler_registry.register(_ad._core.sparse_dataset._CSCDataset)
    class H5pyBackedCSCDataHandler(CSCDataHandler):
        streams = True

        def footprint(self) -> int:
            return self.data._data.nbytes + self.data._indptr.nbytes + self.data._indices.nbytes

        def fetch(self, lb: int, ub: int) -> tuple:
            return self.data[:, lb:ub], (0, ub - lb)
except Exception:  # pragma: no cover
    pass

# import the kernel modules to trigger registration (registry.py:193-202)
from illico_amd import ovo as _ovo  # noqa: E402,F401
from illico_amd import ovr as _ovr  # noqa: E402,F401
