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


def _worker(rank, world, port, fmt, test, out_path, n_genes=37, n_blocks=3):
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
                                     compute_planes=compute)
    if rank == 0:
        df.to_pickle(out_path)
    else:
        assert df is None
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("fmt,test", [("dense", "ovo"), ("csc", "ovr"), ("csr", "ovo")])
def test_sharded_gather_world2(tmp_path, fmt, test):
    import oracle
    from conftest import make_counts, make_labels
    out = tmp_path / "df.pkl"
    mp.spawn(_worker, args=(2, _free_port(), fmt, test, str(out)), nprocs=2, join=True)
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


@pytest.mark.parametrize("world,n_genes,n_blocks", [(2, 5, 4), (3, 7, 4), (2, 1, 2)])
def test_sharded_gather_fewer_genes_than_blocks(tmp_path, world, n_genes, n_blocks):
    """A rank that owns fewer genes than gather blocks (or none) still issues every gather: no IndexError, no hang."""
    import oracle
    from conftest import make_counts, make_labels
    out = tmp_path / "df.pkl"
    mp.spawn(_worker, args=(world, _free_port(), "dense", "ovo", str(out), n_genes, n_blocks), nprocs=world, join=True)
    df = pd.read_pickle(out)
    X, rng = make_counts(3, 600, n_genes, 0.6)
    labels = make_labels(rng, 600, 6, n_ref=60)
    uniq, g = oracle.encode_and_count_groups(labels, "non-targeting")
    p, u, fc = oracle.run(X, g)
    got = df.values.reshape(len(uniq), n_genes, 3)
    np.testing.assert_array_equal(got[:, :, 0], p)
    np.testing.assert_array_equal(got[:, :, 1], u)
    np.testing.assert_array_equal(got[:, :, 2], fc)
