"""The shape sweep as a test: unusual group counts / group sizes / densities / value kinds, each (i) exact against the oracle on four
genes and (ii) within 4x of the time its algorithmic bytes take at the rate of the dense continuous OVO pass at C2
grouping, measured in the same session -- the sweeps of tools/shape_sweep*.sh found cliffs of 10 - 100x (groups above 1024 cells on
continuous data, thousands of groups on continuous sparse data, CSC counts with groups above 255 cells); none may come back.

300 000 cells x 2048 genes (a quarter of C2's genes: times scale with the genes, the rate does not); input and planes resident in HBM.
"""
import time

import numpy as np
import pytest

import oracle
from conftest import assert_planes_match

pytestmark = pytest.mark.gpu

N, M = 300_000, 2048


@pytest.fixture(scope="module")
def ctx():
    import torch
    from bench import compress, group_container, make_labels, make_matrix
    from illico_amd._lib import Engine
    dev = torch.device("cuda", 0)
    eng = Engine(0)
    eng.set_stream(torch.cuda.current_stream().cuda_stream)
    cache = {}

    def data(values, sparsity, fmt):
        key = (values, sparsity, fmt)
        if key not in cache:
            cache.clear()   # one matrix at a time
            torch.cuda.empty_cache()
            X = make_matrix(torch, N, M, sparsity, 0, dev, values=values)
            cache[key] = (X if fmt == "dense" else compress(torch, X, fmt), X[:, ::M // 4][:, :4].contiguous().cpu().numpy(), int((X != 0).sum()))
        return cache[key]

    def run(values, sparsity, fmt, G, ovr):
        """(ms per pass, algorithmic bytes, planes of genes 0, M/4, 2M/4, 3M/4, their oracle planes)."""
        Xd, Xs, nnz = data(values, sparsity, fmt)
        codes = make_labels(N, G, 0)
        grpc = group_container(codes, G, ovr)
        eng.set_groups(grpc)
        out = tuple(torch.empty((G, M), dtype=torch.float64, device=dev) for _ in range(3))

        def step():
            if fmt == "dense":
                eng.run_dense(Xd, 0, M, out=out, defer=True)
            else:
                eng.run_sparse(fmt, Xd[0], Xd[1], Xd[2], (N, M), 0, M, out=out, defer=True)
            eng.synchronize()
            torch.cuda.synchronize()

        step()
        best = 1e9
        for _ in range(2):
            t0 = time.perf_counter()
            step()
            best = min(best, (time.perf_counter() - t0) * 1e3)
        alg = (N * M * 4 if fmt == "dense" else nnz * 8 + 4 * ((M if fmt == "csc" else N) + 1)) + 4 * N + 24 * G * M
        cols = list(range(0, M, M // 4))[:4]
        got = tuple(t[:, cols].cpu().numpy() for t in out)
        want = oracle.run(Xs, grpc, batch_size=1, n_threads=4)
        return best, alg, got, want, grpc

    yield {"run": run, "eng": eng}
    eng.close()


@pytest.fixture(scope="module")
def base_rate(ctx):
    """bytes per ms of the dense continuous OVO pass at C2 grouping (2000 groups of ~145 cells): the yardstick of the sweep."""
    ms, alg, got, want, grpc = ctx["run"]("continuous", 0.5, "dense", 2000, False)
    assert_planes_match(got, want, ref_row=grpc.encoded_ref_group, what="yardstick")
    return alg / ms


SHAPES = [
    # values,      sparsity, format,  groups, test
    ("continuous", 0.5, "dense", 10, "ovo"),      # 9 clusters of 32 000 cells: big runs in value buckets, pieces of 256 keys
    ("continuous", 0.5, "dense", 50, "ovo"),      # 153 ms at the start of round 4
    ("continuous", 0.5, "dense", 300, "ovo"),
    ("continuous", 0.5, "dense", 10000, "ovo"),
    ("continuous", 0.5, "dense", 50, "ovr"),
    ("continuous", 0.5, "dense", 2000, "ovr"),
    ("continuous", 0.5, "dense", 10000, "ovr"),
    ("counts", 0.5, "dense", 50, "ovo"),
    ("counts", 0.5, "dense", 300, "ovr"),
    ("counts", 0.5, "dense", 10000, "ovo"),
    ("counts", 0.5, "dense", 10000, "ovr"),
    ("nb", 0.5, "dense", 2000, "ovo"),            # heavy-tailed counts
    ("counts", 0.9, "csc", 300, "ovo"),           # groups above 255 cells: 16-bit cells
    ("counts", 0.9, "csc", 5000, "ovr"),          # windows of groups over the LDS histograms
    ("counts", 0.9, "csc", 10000, "ovo"),
    ("counts", 0.9, "csc", 30000, "ovo"),         # 15 windows of groups over the LDS histograms (eight were the limit: 52.7 ms at full size in round 4)
    ("counts", 0.9, "csc", 30000, "ovr"),
    ("counts", 0.5, "csc", 2000, "ovo"),          # half of the entries stored: 4-bit cells overflow
    ("continuous", 0.9, "csc", 50, "ovo"),        # clusters of thousands of cells on sparse input: 21 ms at the start of round 4
    ("continuous", 0.9, "csc", 300, "ovo"),
    ("continuous", 0.9, "csc", 50, "ovr"),
    ("continuous", 0.9, "csc", 6000, "ovo"),      # 15.6 ms at full size at the start of round 4
    ("continuous", 0.9, "csc", 6000, "ovr"),      # 44 ms
    ("continuous", 0.9, "csc", 10000, "ovr"),     # 100 ms
    ("counts", 0.9, "csr", 2000, "ovo"),          # the group-major single pass
    ("counts", 0.9, "csr", 2000, "ovr"),
    ("counts", 0.9, "csr", 300, "ovo"),           # more big groups than it takes: byte windows
    ("counts", 0.5, "csr", 2000, "ovo"),
    ("continuous", 0.9, "csr", 2000, "ovo"),
    ("nb", 0.9, "csr", 2000, "ovo"),
    ("continuous", 0.7, "csr", 2000, "ovr"),      # columns of 90 000 stored entries: 76 ms at full size before the dense-window branch
    ("continuous", 0.7, "csc", 2000, "ovr"),      # 76 ms
]


@pytest.mark.parametrize("values,sparsity,fmt,G,test", SHAPES)
def test_no_shape_falls_off_a_cliff(ctx, base_rate, values, sparsity, fmt, G, test):
    ms, alg, got, want, grpc = ctx["run"](values, sparsity, fmt, G, test == "ovr")
    assert_planes_match(got, want, ref_row=grpc.encoded_ref_group if test == "ovo" else None, what=f"{values} {fmt} G={G} {test}")
    budget = 4.0 * alg / base_rate + 0.3   # (+ 0.3 ms: launch sequences and host waits do not shrink with the gene count)
    print(f"{values} s={sparsity} {fmt} G={G} {test}: {ms:.2f} ms, budget {budget:.2f} ms")
    assert ms <= budget, f"{values} s={sparsity} {fmt} G={G} {test}: {ms:.2f} ms for {alg / 1e9:.2f} GB, budget {budget:.2f} ms"


# ---- shapes beyond every LDS-resident look-up (round 5): a reference, or a ranked group's run, of more non-zero keys than LDS holds ----
# (gene counts cut to a fraction of the full shapes' -- the rate does not depend on them; full size: tools/shape_sweep.py beyond_lds)
BEYOND_LDS = [
    # name,                 cells,     genes, groups, sparsity, ms at full size the budget is cut from (None: 4x the bytes, like every other shape)
    ("c5 shard, no zeros", 1_000_000, 512, 5000, 0.0, None),     # 33 333 reference keys per gene: 124 ms at full size (3750 genes) -> 44: the reference in two value-range parts
    ("tall, half non-zero", 2_000_000, 256, 2000, 0.5, None),    # reference of 66 667 cells, groups of ~970: 302 ms (1200 genes) -> 25: parts + runs in value buckets
    # ten clusters of 100 000 cells, runs of ~50 000 keys: 235 ms (2400 genes) -> 30: runs dealt through HBM.  Budget: the 45 ms VERDICT r04 asked of
    # the full shape, in proportion -- 4x the bytes would be 34 ms there; k_group_compact runs one workgroup per (cluster, 64 genes), 10 ms of the 30
    ("clusters, half non-zero", 1_000_000, 1200, 10, 0.5, 45.0 * 1200 / 2400),
]


@pytest.mark.parametrize("name,cells,genes,G,sparsity,budget_ms", BEYOND_LDS)
def test_continuous_ovo_beyond_the_lds_resident_lookups(base_rate, name, cells, genes, G, sparsity, budget_ms):
    """Exact on four genes against the oracle, within 4x of the bytes at the yardstick rate (what the packed routes do when neither the
    reference's keys nor a group's run fit LDS)."""
    _atlas_case(base_rate, name, cells, genes, G, sparsity, "continuous", False, budget_ms)


# ---- atlas shapes (round 5, second half): clusters of 100 000 cells and columns of two million cells, every test and value kind ----
ATLAS = [
    # name,                                   cells,     genes, groups, sparsity, values,       test
    ("clusters, counts, ovo",                 1_000_000, 1200, 10, 0.5, "counts", "ovo"),       # 9.1 ms at full size (2400 genes) -> 1.8: group histograms, rows split over the wavefronts
    ("clusters, counts, ovr",                 1_000_000, 1200, 10, 0.5, "counts", "ovr"),       # 8.6 -> 1.9
    ("clusters, a tenth stored, ovr",         1_000_000, 1200, 10, 0.9, "continuous", "ovr"),   # 20.4 -> 7.7: packed rows for the partition, long blocks dealt over the wavefronts
    ("clusters, half non-zero, ovr",          1_000_000, 1200, 10, 0.5, "continuous", "ovr"),   # 33 -> 23
    ("tall, half non-zero, ovr",              2_000_000, 256, 2000, 0.5, "continuous", "ovr"),  # 166 ms at 1200 genes (the general route: columns longer than 128 half-size parts) -> 27
]


@pytest.mark.parametrize("name,cells,genes,G,sparsity,values,test", ATLAS)
def test_atlas_shapes(base_rate, name, cells, genes, G, sparsity, values, test):
    # (continuous OVR of cluster-sized groups half non-zero sits at 3.2x its bytes -- the partition and the rank kernel each move the records
    #  once more --: 6x for that one, so that a slow box does not fail it; everything else: 4x like the other shapes)
    _atlas_case(base_rate, name, cells, genes, G, sparsity, values, test == "ovr", None, times=6.0 if (values == "continuous" and test == "ovr" and sparsity == 0.5 and G == 10) else 4.0)


def _atlas_case(base_rate, name, cells, genes, G, sparsity, values, ovr, budget_ms, times=4.0):
    import torch
    from bench import group_container, make_labels, make_matrix
    from illico_amd._lib import Engine
    dev = torch.device("cuda", 0)
    torch.cuda.empty_cache()
    eng = Engine(0)
    try:
        eng.set_stream(torch.cuda.current_stream().cuda_stream)
        X = make_matrix(torch, cells, genes, sparsity, 0, dev, values=values)
        grpc = group_container(make_labels(cells, G, 0), G, ovr)
        eng.set_groups(grpc)
        out = tuple(torch.empty((G, genes), dtype=torch.float64, device=dev) for _ in range(3))

        def step():
            eng.run_dense(X, 0, genes, out=out, defer=True)
            eng.synchronize()
            torch.cuda.synchronize()

        step()
        best = 1e9
        for _ in range(2):
            t0 = time.perf_counter()
            step()
            best = min(best, (time.perf_counter() - t0) * 1e3)
        cols = list(range(0, genes, genes // 4))[:4]
        got = tuple(t[:, cols].cpu().numpy() for t in out)
        want = oracle.run(X[:, cols].contiguous().cpu().numpy(), grpc, batch_size=1, n_threads=4)
        assert_planes_match(got, want, ref_row=None if ovr else grpc.encoded_ref_group, what=name)
        alg = cells * genes * 4 + 4 * cells + 24 * G * genes
        budget = (times * alg / base_rate if budget_ms is None else budget_ms) + 0.3
        print(f"{name}: {best:.2f} ms, budget {budget:.2f} ms")
        assert best <= budget, f"{name}: {best:.2f} ms for {alg / 1e9:.2f} GB, budget {budget:.2f} ms"
    finally:
        eng.close()
        del X
        torch.cuda.empty_cache()


def test_float64_csr_continuous_within_twice_the_float32_time(ctx):
    """C3 shape as CSR with continuous values in float64 (no float32 holds them: every value times 1 + 2^-30): eight-byte keys halve what
    the per-gene LDS kernels hold, and round 4 sent such a matrix through a 19-GB dense window (14 ms at full size against 6 in float32).
    Now the regrouped runs are ranked by the packed kernel: exact on four genes, within twice the float32 time of the same session."""
    import torch
    from bench import compress, group_container, make_labels, make_matrix
    dev = torch.device("cuda", 0)
    eng = ctx["eng"]
    G = 2000
    X = make_matrix(torch, N, M, 0.9, 0, dev, values="continuous")
    d, i, p = compress(torch, X, "csr")
    cols = list(range(0, M, M // 4))[:4]
    Xs = X[:, cols].contiguous().cpu().numpy().astype(np.float64) * (1.0 + 2.0 ** -30)
    del X
    grpc = group_container(make_labels(N, G, 0), G, False)
    eng.set_groups(grpc)
    out = tuple(torch.empty((G, M), dtype=torch.float64, device=dev) for _ in range(3))
    times = {}
    for name, data in (("f32", d), ("f64", d.double() * (1.0 + 2.0 ** -30))):
        def step():
            eng.run_sparse("csr", data, i, p, (N, M), 0, M, out=out)
            eng.synchronize()
            torch.cuda.synchronize()
        step()
        best = 1e9
        for _ in range(2):
            t0 = time.perf_counter()
            step()
            best = min(best, (time.perf_counter() - t0) * 1e3)
        times[name] = best
    got = tuple(t[:, cols].cpu().numpy() for t in out)   # (the float64 pass's planes)
    assert_planes_match(got, oracle.run(Xs, grpc, batch_size=1, n_threads=4), ref_row=grpc.encoded_ref_group, what="float64 CSR continuous")
    print(f"CSR continuous: float32 {times['f32']:.2f} ms, float64 {times['f64']:.2f} ms")
    assert times["f64"] <= 2.0 * times["f32"] + 0.3, times
