"""Several engine contexts driven by host threads of ONE process, each computing its gene range straight into its column range of
one host result: the single-process multi-GPU form of SURVEY.md 8e (`illico_amd.distributed.asymptotic_wilcoxon_threads`) and the
tail of the sharded drop-in (every rank brings its own planes to the host: `out_ld` = the whole result's width).  A single-GPU box
runs the contexts on the same device -- the code path is the 8-GPU one, device ids apart."""
import numpy as np
import pandas as pd
import pytest
from scipy import sparse

import oracle
from conftest import assert_planes_match, make_counts, make_labels

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("fmt,test,n_ctx", [("dense", "ovo", 2), ("csr", "ovr", 2), ("csc", "ovo", 3), ("dense", "ovr", 4)])
def test_contexts_on_threads_write_disjoint_column_ranges_of_one_host_result(fmt, test, n_ctx):
    from illico_amd import AnnDataLite, asymptotic_wilcoxon
    from illico_amd.distributed import asymptotic_wilcoxon_threads
    X, rng = make_counts(21, 4000, 301, 0.6)
    X[:, 17] = rng.rand(4000).astype(np.float32)          # one continuous gene: not every column takes the same route
    labels = make_labels(rng, 4000, 14, n_ref=400)
    M = {"dense": X, "csc": sparse.csc_matrix(X), "csr": sparse.csr_matrix(X)}[fmt]
    adata = AnnDataLite(M, obs=pd.DataFrame({"pert": labels}))
    ref = "non-targeting" if test == "ovo" else None
    df = asymptotic_wilcoxon_threads(adata, False, "pert", ref, devices=[0] * n_ctx)
    one = asymptotic_wilcoxon(adata, is_log1p=False, group_keys="pert", reference=ref)
    pd.testing.assert_frame_equal(df, one, check_exact=True)
    uniq, g = oracle.encode_and_count_groups(labels, ref)
    got = df.values.reshape(len(uniq), 301, 3)
    assert_planes_match((got[:, :, 0], got[:, :, 1], got[:, :, 2]), oracle.run(X, g), ref_row=g.encoded_ref_group, what=f"threads {fmt} {test}")


def test_two_engines_write_their_halves_of_one_result_concurrently():
    """The raw form: two Engine objects (two illico_ctx) on two threads, host planes = windows of one [3][G][M] array."""
    import threading
    from illico_amd._lib import Engine
    X, rng = make_counts(5, 6000, 256, 0.5)
    labels = make_labels(rng, 6000, 20, n_ref=500)
    _, g = oracle.encode_and_count_groups(labels, "non-targeting")
    G = g.counts.size
    res = np.full((3, G, 256), -7.0)
    errs = []

    def work(lb, ub):
        try:
            eng = Engine(0)
            eng.set_groups(g)
            for _ in range(3):
                eng.run_dense(X, lb, ub, out=tuple(res[k][:, lb:ub] for k in range(3)))
            eng.close()
        except BaseException as e:  # noqa: BLE001
            errs.append(e)

    ts = [threading.Thread(target=work, args=r) for r in ((0, 100), (100, 256))]
    for t in ts:
        t.start()
    for t in ts:
        t.join()
    assert not errs, errs
    assert_planes_match(tuple(res), oracle.run(X, g), ref_row=g.encoded_ref_group, what="two engines, one result")
