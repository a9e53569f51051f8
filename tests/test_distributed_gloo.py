"""world_size-2 gloo test of the gene-sharding + gather logic (CPU): the per-rank compute is injected
(the CPU oracle stands in for the engine), everything else is the product code path of
illico_amd.distributed.asymptotic_wilcoxon_sharded."""
import os
import socket
import sys
from pathlib import Path

import numpy as np
import pandas as pd
import pytest
import torch.multiprocessing as mp

ROOT = Path(__file__).resolve().parent.parent


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, fmt, test, out_path, n_genes=37, n_blocks=3, tail="host"):
    sys.path.insert(0, str(ROOT))
    sys.path.insert(0, str(ROOT / "tests"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import torch.distributed as dist
    from scipy import sparse
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import oracle
    from conftest import make_counts, make_labels
    from illico_amd import AnnDataLite
    from illico_amd.distributed import asymptotic_wilcoxon_sharded

    X, rng = make_counts(3, 600, n_genes, 0.6)
    labels = make_labels(rng, 600, 6, n_ref=60)
    M = {"dense": X, "csc": sparse.csc_matrix(X), "csr": sparse.csr_matrix(X)}[fmt]
    adata = AnnDataLite(M, obs=pd.DataFrame({"pert": labels}))

    def compute(Xm, grpc, lb, ub, **o):
        return oracle.run(Xm, grpc, col_lb=lb, col_ub=ub, **o)

    df = asymptotic_wilcoxon_sharded(adata, False, "pert", "non-targeting" if test == "ovo" else None, n_blocks=n_blocks,
                                     compute_planes=compute, tail=tail)
    if rank == 0:
        if tail == "host":  # the planes live in the shared mapping, whose name is gone already (nothing outlives the processes)
            import glob
            assert not glob.glob("/dev/shm/illico_planes_*"), glob.glob("/dev/shm/illico_planes_*")
            assert not df["p_value"].values.flags.owndata
        df.to_pickle(out_path)
    else:
        assert df is None
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("tail", ["host", "device"])
@pytest.mark.parametrize("fmt,test", [("dense", "ovo"), ("csc", "ovr"), ("csr", "ovo")])
def test_sharded_gather_world2(tmp_path, fmt, test, tail):
    """tail="host": every rank writes its column range of ONE shared host result (no collective on the data path);
    tail="device": the exact-width p2p gather into rank 0."""
    import oracle
    from conftest import make_counts, make_labels
    out = tmp_path / "df.pkl"
    mp.spawn(_worker, args=(2, _free_port(), fmt, test, str(out), 37, 3, tail), nprocs=2, join=True)
    df = pd.read_pickle(out)
    X, rng = make_counts(3, 600, 37, 0.6)
    labels = make_labels(rng, 600, 6, n_ref=60)
    uniq, g = oracle.encode_and_count_groups(labels, "non-targeting" if test == "ovo" else None)
    p, u, fc = oracle.run(X, g)
    got = df.values.reshape(len(uniq), 37, 3)
    np.testing.assert_array_equal(got[:, :, 0], p)
    np.testing.assert_array_equal(got[:, :, 1], u)
    np.testing.assert_array_equal(got[:, :, 2], fc)
    assert list(df.index.get_level_values(0).unique()) == list(uniq)
    assert df.index.names == ["pert", "feature"]


@pytest.mark.parametrize("tail", ["host", "device"])
@pytest.mark.parametrize("world,n_genes,n_blocks", [(2, 5, 4), (3, 7, 4), (2, 1, 2)])
def test_sharded_gather_fewer_genes_than_blocks(tmp_path, world, n_genes, n_blocks, tail):
    """A rank that owns fewer genes than gather blocks (or none) still issues every gather / reaches every barrier: no IndexError, no hang."""
    import oracle
    from conftest import make_counts, make_labels
    out = tmp_path / "df.pkl"
    mp.spawn(_worker, args=(world, _free_port(), "dense", "ovo", str(out), n_genes, n_blocks, tail), nprocs=world, join=True)
    df = pd.read_pickle(out)
    X, rng = make_counts(3, 600, n_genes, 0.6)
    labels = make_labels(rng, 600, 6, n_ref=60)
    uniq, g = oracle.encode_and_count_groups(labels, "non-targeting")
    p, u, fc = oracle.run(X, g)
    got = df.values.reshape(len(uniq), n_genes, 3)
    np.testing.assert_array_equal(got[:, :, 0], p)
    np.testing.assert_array_equal(got[:, :, 1], u)
    np.testing.assert_array_equal(got[:, :, 2], fc)


# ---- nnz-balanced ranges ("for sparse inputs balance by nnz rather than gene count") and the per-rank loader ----
def _skewed_counts(seed, n_cells, n_genes):
    """A count matrix whose genes differ 50-fold in stored entries, densest first (what sorting genes by expression gives)."""
    rng = np.random.RandomState(seed)
    dens = np.geomspace(0.5, 0.01, n_genes)
    X = rng.poisson(3.0, size=(n_cells, n_genes)).astype(np.float32) + 1.0
    X[rng.rand(n_cells, n_genes) >= dens] = 0
    return X, rng


def test_balanced_gene_ranges_by_stored_entries():
    from scipy import sparse
    from illico_amd.distributed import balanced_gene_ranges, gene_ranges_for, rank_gene_range, sparse_gene_weights
    X, _ = _skewed_counts(0, 2000, 400)
    for M in (sparse.csc_matrix(X), sparse.csr_matrix(X)):
        w = sparse_gene_weights(M)
        np.testing.assert_array_equal(w, (X != 0).sum(axis=0))
        for world in (2, 3, 8):
            rg = gene_ranges_for(M, world)
            assert rg[0][0] == 0 and rg[-1][1] == 400 and all(a[1] == b[0] for a, b in zip(rg, rg[1:]))
            per = np.array([w[lb:ub].sum() for lb, ub in rg])
            assert per.max() <= 1.10 * per.mean() and per.min() >= 0.90 * per.mean(), (world, per)
            by_count = np.array([w[slice(*rank_gene_range(400, r, world))].sum() for r in range(world)])
            assert by_count.max() > 1.5 * by_count.mean()  # what the split by gene count would have given
    # dense input: by gene count
    assert gene_ranges_for(X, 3) == [rank_gene_range(400, r, 3) for r in range(3)]
    # degenerate inputs: no genes, no weight, more ranks than genes, one gene holding everything
    assert balanced_gene_ranges([], 3) == [(0, 0)] * 3
    assert balanced_gene_ranges([0, 0, 0, 0], 2) == [(0, 2), (2, 4)]
    rg = balanced_gene_ranges([5.0, 1.0], 4)
    assert rg[0][0] == 0 and rg[-1][1] == 2 and all(a[1] == b[0] and a[0] <= a[1] for a, b in zip(rg, rg[1:]))
    rg = balanced_gene_ranges([0, 0, 100, 0, 0], 3)
    assert sum(1 for lb, ub in rg if lb <= 2 < ub) == 1 and rg[-1][1] == 5


def _worker_loader(rank, world, port, fmt, out_path, log_dir, tail="host"):
    sys.path.insert(0, str(ROOT))
    sys.path.insert(0, str(ROOT / "tests"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import torch.distributed as dist
    from scipy import sparse
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import oracle
    from conftest import make_labels
    from illico_amd.distributed import asymptotic_wilcoxon_sharded, sparse_gene_weights

    X, rng = _skewed_counts(5, 500, 41)
    labels = make_labels(rng, 500, 5, n_ref=50)
    full = {"dense": X, "csc": sparse.csc_matrix(X), "csr": sparse.csr_matrix(X)}[fmt]
    asked = []

    def loader(lb, ub):  # a rank's own columns only (a real loader would read them from storage)
        asked.append((lb, ub))
        return full[:, lb:ub] if fmt == "dense" else full[:, lb:ub].asformat(fmt)

    def compute(Xb, grpc, lb, ub, **o):
        assert Xb.shape[1] <= 41 and (world == 1 or Xb.shape[1] < 41)  # never the whole matrix
        return oracle.run(Xb, grpc, col_lb=lb, col_ub=ub, **o)

    weights = sparse_gene_weights(full) if fmt != "dense" else None
    df = asymptotic_wilcoxon_sharded(None, False, "pert", "non-targeting", n_blocks=3, compute_planes=compute,
                                     column_loader=loader, n_genes=41, groups=labels, gene_weights=weights, tail=tail,
                                     var_names=[f"g{j}" for j in range(41)] if rank == 0 else None)
    with open(os.path.join(log_dir, f"asked_{rank}.txt"), "w") as f:
        f.write(repr(asked))
    if rank == 0:
        df.to_pickle(out_path)
    else:
        assert df is None
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("tail", ["host", "device"])
@pytest.mark.parametrize("fmt", ["dense", "csc", "csr"])
def test_sharded_with_a_per_rank_column_loader(tmp_path, fmt, tail):
    """A rank loads its own gene range once and nothing else; sparse ranges are balanced by stored entries."""
    import oracle
    from scipy import sparse
    from conftest import make_labels
    from illico_amd.distributed import balanced_gene_ranges, rank_gene_range, sparse_gene_weights
    out = tmp_path / "df.pkl"
    mp.spawn(_worker_loader, args=(2, _free_port(), fmt, str(out), str(tmp_path), tail), nprocs=2, join=True)
    df = pd.read_pickle(out)
    X, rng = _skewed_counts(5, 500, 41)
    labels = make_labels(rng, 500, 5, n_ref=50)
    uniq, g = oracle.encode_and_count_groups(labels, "non-targeting")
    p, u, fc = oracle.run(X, g)
    got = df.values.reshape(len(uniq), 41, 3)
    np.testing.assert_array_equal(got[:, :, 0], p)
    np.testing.assert_array_equal(got[:, :, 1], u)
    np.testing.assert_array_equal(got[:, :, 2], fc)
    assert list(df.index.get_level_values(1)[:41]) == [f"g{j}" for j in range(41)]
    want = (balanced_gene_ranges(sparse_gene_weights(sparse.csc_matrix(X)), 2) if fmt != "dense"
            else [rank_gene_range(41, r, 2) for r in range(2)])
    for r in range(2):
        asked = eval((tmp_path / f"asked_{r}.txt").read_text())
        assert asked == [want[r]], (r, asked, want)
    if fmt != "dense":
        assert want[0][1] < 41 // 2  # the dense genes come first: rank 0 takes fewer genes than half


def _worker_shared_planes(rank, world, port, out_path):
    sys.path.insert(0, str(ROOT))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import torch.distributed as dist
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from illico_amd.distributed import SharedHostPlanes, rank_gene_range
    sh = SharedHostPlanes(4, 10, group=None)
    lb, ub = rank_gene_range(10, rank, world)
    for k, w in enumerate(sh.columns(lb, ub)):
        assert w.shape == (4, ub - lb) and w.strides == (80, 8)     # the row pitch is the whole result's: out_ld of the C-ABI
        w[...] = 100 * k + 10 * rank + np.arange(ub - lb)
    sh.barrier()
    if rank == 0:
        np.save(out_path, sh.array)
    dist.barrier()
    dist.destroy_process_group()


def test_shared_host_planes_world3(tmp_path):
    """Three processes map ONE [3][G][M] host result, each writes its own column range, rank 0 reads all of it."""
    import glob
    from illico_amd.distributed import rank_gene_range
    out = str(tmp_path / "planes.npy")
    mp.spawn(_worker_shared_planes, args=(3, _free_port(), out), nprocs=3, join=True)
    got = np.load(out)
    assert got.shape == (3, 4, 10)
    for r in range(3):
        lb, ub = rank_gene_range(10, r, 3)
        for k in range(3):
            np.testing.assert_array_equal(got[k][:, lb:ub], np.broadcast_to(100 * k + 10 * r + np.arange(ub - lb), (4, ub - lb)))
    assert not glob.glob("/dev/shm/illico_planes_*")


@pytest.mark.parametrize("fmt,test,n_dev", [("dense", "ovo", 2), ("csr", "ovr", 3), ("csc", "ovo", 1)])
def test_threads_form_writes_disjoint_column_ranges_of_one_host_result(fmt, test, n_dev):
    """The single-process form: one host thread per device context, each computing its gene range into its columns of one host
    result; no process group.  (CPU compute injected; the engine itself: tests/test_gpu_multi_context.py.)"""
    import oracle
    from scipy import sparse
    from conftest import make_counts, make_labels
    from illico_amd import AnnDataLite
    from illico_amd.distributed import asymptotic_wilcoxon_threads, gene_ranges_for
    X, rng = make_counts(9, 500, 29, 0.6)
    labels = make_labels(rng, 500, 5, n_ref=50)
    M = {"dense": X, "csc": sparse.csc_matrix(X), "csr": sparse.csr_matrix(X)}[fmt]
    seen = []

    def compute(Xm, grpc, lb, ub, out, i, **o):
        seen.append((i, lb, ub))
        for dst, a in zip(out, oracle.run(Xm, grpc, col_lb=lb, col_ub=ub, **o)):
            dst[...] = a

    ref = "non-targeting" if test == "ovo" else None
    df = asymptotic_wilcoxon_threads(AnnDataLite(M, obs=pd.DataFrame({"pert": labels})), False, "pert", ref, devices=[0] * n_dev,
                                     compute_planes=compute, alternative="less")
    uniq, g = oracle.encode_and_count_groups(labels, ref)
    p, u, fc = oracle.run(X, g, alternative="less")
    got = df.values.reshape(len(uniq), 29, 3)
    np.testing.assert_array_equal(got[:, :, 0], p)
    np.testing.assert_array_equal(got[:, :, 1], u)
    np.testing.assert_array_equal(got[:, :, 2], fc)
    assert sorted((lb, ub) for _, lb, ub in seen) == [r for r in gene_ranges_for(M, n_dev) if r[1] > r[0]]
    with pytest.raises(ValueError):
        asymptotic_wilcoxon_threads(AnnDataLite(M, obs=pd.DataFrame({"pert": labels})), False, "pert", "no such label", devices=[0], compute_planes=compute)
