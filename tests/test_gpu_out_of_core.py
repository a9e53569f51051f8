"""Out-of-core inputs (reference illico/utils/registry.py:162-188, tests/test_asymptotic_wilcoxon.py:198-256).

h5py / anndata are not in this image, so the reference's two backed containers are exercised through file-backed stand-ins
with the same duck type -- ``FakeH5Dataset`` ([:, lb:ub] -> ndarray read from a chunked file, like a chunked HDF5 dataset) and
``FakeBackedCSC`` ([:, lb:ub] -> scipy CSC matrix read from three array files, like anndata's ``_CSCDataset``) -- registered
under the SAME handlers a real ``h5py.Dataset`` / ``_CSCDataset`` gets.  Results are checked against the CPU oracle (not
against the in-RAM HIP result), and the host memory of a backed call is bounded by a few chunks, not by the matrix."""
import threading
import time
from pathlib import Path

import numpy as np
import pandas as pd
import pytest
from scipy import sparse

import oracle
from conftest import assert_planes_match, make_counts, make_labels

pytestmark = pytest.mark.gpu

CB = 64  # columns per stored block of the dense stand-in (a chunked HDF5 layout)


class FakeH5Dataset:
    """``h5py.Dataset`` look-alike: ``shape``, ``dtype``, ``nbytes`` and ``ds[:, lb:ub] -> np.ndarray`` read from the file with
    plain reads (no memory mapping: nothing of the file stays resident)."""

    def __init__(self, path, X=None):
        self.path = Path(path)
        if X is not None:
            self.shape, self.dtype = X.shape, X.dtype
            with open(self.path, "wb") as f:   # column blocks of CB genes, each [N, CB] row-major
                for j in range(0, X.shape[1], CB):
                    blk = np.zeros((X.shape[0], CB), X.dtype)
                    blk[:, : min(CB, X.shape[1] - j)] = X[:, j:j + CB]
                    f.write(blk.tobytes())
        self.reads = []

    @property
    def nbytes(self):
        return int(np.prod(self.shape)) * self.dtype.itemsize

    def __getitem__(self, key):
        rows, cols = key
        assert rows == slice(None)
        lb, ub = cols.start, cols.stop
        self.reads.append((lb, ub))
        N = self.shape[0]
        out = np.empty((N, ub - lb), self.dtype)
        for b in range(lb // CB, (ub - 1) // CB + 1):
            blk = np.fromfile(self.path, dtype=self.dtype, count=N * CB, offset=b * N * CB * self.dtype.itemsize).reshape(N, CB)
            c0, c1 = max(lb, b * CB), min(ub, (b + 1) * CB)
            out[:, c0 - lb: c1 - lb] = blk[:, c0 - b * CB: c1 - b * CB]
        return out


class FakeBackedCSC:
    """anndata ``_CSCDataset`` look-alike: ``_data`` / ``_indices`` / ``_indptr`` on disk, ``ds[:, lb:ub] -> scipy.sparse.csc_matrix``."""

    def __init__(self, prefix, M):
        M = sparse.csc_matrix(M)
        self.shape, self.dtype = M.shape, M.data.dtype
        self.prefix = str(prefix)
        M.data.tofile(self.prefix + ".data"); M.indices.tofile(self.prefix + ".indices")
        self._indptr = M.indptr.copy()                      # anndata keeps indptr in memory too
        self._idt = M.indices.dtype
        self._data = np.memmap(self.prefix + ".data", dtype=self.dtype, mode="r")      # only .nbytes is used (footprint)
        self._indices = np.memmap(self.prefix + ".indices", dtype=self._idt, mode="r")
        self.reads = []

    def __getitem__(self, key):
        rows, cols = key
        lb, ub = cols.start, cols.stop
        self.reads.append((lb, ub))
        s, e = int(self._indptr[lb]), int(self._indptr[ub])
        d = np.fromfile(self.prefix + ".data", dtype=self.dtype, count=e - s, offset=s * self.dtype.itemsize)
        i = np.fromfile(self.prefix + ".indices", dtype=self._idt, count=e - s, offset=s * self._idt.itemsize)
        return sparse.csc_matrix((d, i, self._indptr[lb:ub + 1] - s), shape=(self.shape[0], ub - lb))


@pytest.fixture(scope="module", autouse=True)
def _register():
    from illico_amd.utils.registry import H5pyBackedCSCDataHandler, H5pyDatasetDataHandler, data_handler_registry
    data_handler_registry[FakeH5Dataset] = H5pyDatasetDataHandler      # what h5py.Dataset is registered under
    data_handler_registry[FakeBackedCSC] = H5pyBackedCSCDataHandler    # what anndata's _CSCDataset is registered under
    yield
    data_handler_registry.pop(FakeH5Dataset, None)
    data_handler_registry.pop(FakeBackedCSC, None)


def _planes(df, G, M):
    a = df.values.reshape(G, M, 3)
    return a[:, :, 0], a[:, :, 1], a[:, :, 2]


@pytest.mark.parametrize("kind", ["h5-dense", "backed-csc"])
@pytest.mark.parametrize("test", ["ovo", "ovr"])
@pytest.mark.parametrize("values", ["counts", "continuous"])
def test_backed_containers_stream_in_chunks_and_match_the_oracle(tmp_path, monkeypatch, kind, test, values):
    import sys
    from illico_amd import AnnDataLite, asymptotic_wilcoxon
    aw = sys.modules["illico_amd.asymptotic_wilcoxon"]
    X, rng = make_counts(91, 4000, 333, 0.8 if kind == "backed-csc" else 0.5)
    if values == "continuous":
        X = np.log1p(X * rng.uniform(0.5, 1.5, size=X.shape)).astype(np.float32)
    X[:, 11] = rng.poisson(80.0, size=4000) * (rng.rand(4000) < 0.3)   # a gene outside the count tables
    labels = make_labels(rng, 4000, 15, n_ref=300)
    ref = "non-targeting" if test == "ovo" else None
    uniq, g = oracle.encode_and_count_groups(labels, ref)
    want = oracle.run(X, g)
    if kind == "h5-dense":
        ds = FakeH5Dataset(tmp_path / "x.h5like", X)
        monkeypatch.setattr(aw, "STREAM_CHUNK_BYTES", 4000 * 4 * 50)     # 50 genes per chunk -> 7 chunks
    else:
        ds = FakeBackedCSC(tmp_path / "x.csc", X)
        monkeypatch.setattr(aw, "STREAM_CHUNK_BYTES", 4000 * 4 * 50)
    df = asymptotic_wilcoxon(AnnDataLite(ds, obs=pd.DataFrame({"pert": labels})), is_log1p=False, group_keys="pert", reference=ref)
    assert ds.reads == [(lb, min(lb + 50, 333)) for lb in range(0, 333, 50)]     # one read per chunk, in order, nothing twice
    assert_planes_match(_planes(df, len(uniq), 333), want, ref_row=g.encoded_ref_group if ref else None, what=f"{kind} {test} {values}")
    assert list(df.index.get_level_values(0).unique()) == list(uniq)


@pytest.mark.parametrize("kind", ["h5-dense", "backed-csc"])
def test_backed_call_keeps_host_memory_to_a_few_chunks(tmp_path, monkeypatch, kind):
    """The reference bounds the heap of a backed call (tests/test_asymptotic_wilcoxon.py:198-256: < 10 MB backed, > 50 MB eager).
    Here: a 384 MB matrix on disk, 16 MB chunks; the peak resident set of the process grows by less than 8 chunks during the
    backed call (chunk being read + two pinned staging slots + result planes + allocator slack) -- a third of the matrix."""
    import sys
    import psutil
    from illico_amd import AnnDataLite, asymptotic_wilcoxon
    aw = sys.modules["illico_amd.asymptotic_wilcoxon"]
    N, M, G = 32_000, 3_000, 12
    rng = np.random.RandomState(5)
    labels = make_labels(rng, N, G, n_ref=2000)
    obs = pd.DataFrame({"pert": labels})
    means = rng.uniform(0.1, 15, size=M)
    path = tmp_path / "big"
    # written block by block: the full matrix never exists in this process
    if kind == "h5-dense":
        ds = FakeH5Dataset(path)
        ds.shape, ds.dtype = (N, M), np.dtype(np.float32)
        with open(path, "wb") as f:
            for j in range(0, M, CB):
                w = min(CB, M - j)
                blk = np.zeros((N, CB), np.float32)
                blk[:, :w] = rng.poisson(means[j:j + w], size=(N, w)) * (rng.rand(N, w) < 0.5)
                f.write(blk.tobytes())
        matrix_bytes = N * M * 4
    else:
        blocks = [sparse.csc_matrix((rng.poisson(means[j:j + 100], size=(N, 100)) * (rng.rand(N, 100) < 0.5)).astype(np.float32))
                  for j in range(0, M, 100)]
        ds = FakeBackedCSC(path, sparse.hstack(blocks, format="csc"))
        del blocks
        matrix_bytes = ds._data.nbytes + ds._indices.nbytes
    chunk = 16 << 20
    monkeypatch.setattr(aw, "STREAM_CHUNK_BYTES", chunk)
    small = AnnDataLite(np.ascontiguousarray(rng.poisson(2.0, size=(N, 64)).astype(np.float32)), obs=obs)
    asymptotic_wilcoxon(small, is_log1p=False, group_keys="pert", reference="non-targeting")   # runtime, code objects, scratch: warm
    import gc
    gc.collect()
    proc = psutil.Process()
    base = proc.memory_info().rss
    peak = [base]
    stop = threading.Event()

    def watch():
        while not stop.is_set():
            peak[0] = max(peak[0], proc.memory_info().rss)
            time.sleep(0.002)

    th = threading.Thread(target=watch)
    th.start()
    try:
        df = asymptotic_wilcoxon(AnnDataLite(ds, obs=obs), is_log1p=False, group_keys="pert", reference="non-targeting")
    finally:
        stop.set()
        th.join()
    assert len(ds.reads) >= matrix_bytes // chunk // 2 >= 5
    grew = peak[0] - base
    assert grew < 8 * chunk, f"resident set grew by {grew / 2**20:.0f} MiB during a backed call over a {matrix_bytes / 2**20:.0f} MiB matrix"
    assert grew < matrix_bytes / 3
    # and the numbers are right: sampled genes against the oracle
    cols = [0, 1499, M - 1]
    Xs = np.stack([np.asarray(ds[:, c:c + 1].todense()).ravel() if kind == "backed-csc" else ds[:, c:c + 1].ravel() for c in cols], axis=1)
    uniq, g = oracle.encode_and_count_groups(labels, "non-targeting")
    want = oracle.run(np.ascontiguousarray(Xs.astype(np.float32)), g)
    got = _planes(df, G, M)
    assert_planes_match(tuple(a[:, cols] for a in got), want, ref_row=g.encoded_ref_group, what=f"{kind} big")
