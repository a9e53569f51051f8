"""Full-size (BASELINE.json configs 2-4) checks on the GPU through size-independent properties, plus spot
checks of sampled genes against the CPU oracle.  Data are generated on the device (bench.py recipe)."""
import numpy as np
import pytest

import oracle
from conftest import assert_planes_match

pytestmark = pytest.mark.gpu

N, M, G = 300_000, 8_000, 2_000
ORACLE_THREADS = 16   # the GPU box's CPU share


def one_gene_per_tile(n_genes, tile=64):
    """One oracle-checked gene in every 64-gene tile of the fused kernels, at a position that moves through the tile."""
    return [t * tile + (t * 7) % min(tile, n_genes - t * tile) for t in range((n_genes + tile - 1) // tile)]


@pytest.fixture(scope="module")
def c2():
    import torch
    from bench import group_container, make_labels, make_matrix
    from illico_amd._lib import Engine
    dev = torch.device("cuda", 0)
    codes = make_labels(N, G, 0)
    X = make_matrix(torch, N, M, 0.5, 0, dev)
    eng = Engine(0)
    eng.set_stream(torch.cuda.current_stream().cuda_stream)
    yield {"torch": torch, "X": X, "codes": codes, "eng": eng, "gc": group_container}
    eng.close()


def _planes(c2, grpc, X=None, lb=0, ub=None, **kw):
    torch = c2["torch"]
    X = c2["X"] if X is None else X
    ub = X.shape[1] if ub is None else ub
    c2["eng"].set_groups(grpc)
    out = c2["eng"].run_dense(X, lb, ub, device_out=True, **kw)
    torch.cuda.synchronize()
    return out


def test_c2_ovo_full_pass_properties_and_spot_checks(c2):
    torch = c2["torch"]
    grpc = c2["gc"](c2["codes"], G, False)
    p, u, fc = _planes(c2, grpc)
    counts = torch.from_numpy(grpc.counts).cuda().double()
    n_ref = float(grpc.counts[0])
    # ranges: 0 <= U <= n_ref * n_g, 2U integral, 0 <= p <= 1, reference row = (1, -1)
    assert bool((u[1:] >= 0).all()) and bool((u[1:] <= (n_ref * counts[1:]).unsqueeze(1)).all())
    assert bool(((2 * u[1:]) == torch.round(2 * u[1:])).all())
    assert bool(((p >= 0) & (p <= 1)).all()) and bool((p[0] == 1).all()) and bool((u[0] == -1).all())
    assert bool(torch.isfinite(fc[1:]).all())
    # one gene of every 64-gene tile (125 genes x 2000 groups) against the CPU oracle at full N and G
    cols = sorted(set(one_gene_per_tile(M) + [0, 1, M - 1]))
    assert len(cols) >= 125
    Xs = c2["X"][:, cols].contiguous().cpu().numpy()
    want = oracle.run(Xs, grpc, batch_size=1, n_threads=ORACLE_THREADS)
    got = tuple(a[:, cols].cpu().numpy() for a in (p, u, fc))
    assert_planes_match(got, want, ref_row=0, what="C2 OVO, one gene per tile")


def test_c2_routes_agree_bitwise_on_a_gene_slice(c2):
    """Fused single-pass route == two-pass histogram route == sort route (same integers, same finalisation)."""
    grpc = c2["gc"](c2["codes"], G, False)
    eng = c2["eng"]
    ref = [a.cpu().numpy() for a in _planes(c2, grpc, lb=512, ub=768)]
    for opt in ("no_fused_path", "no_counts_path"):
        eng.set_option(opt, 1)
        try:
            got = [a.cpu().numpy() for a in _planes(c2, grpc, lb=512, ub=768)]
        finally:
            eng.set_option(opt, 0)
        for a, b in zip(got, ref):
            np.testing.assert_array_equal(a, b, err_msg=opt)


def test_c2_ovo_reference_swap_symmetry(c2):
    """U(ref=A, grp=B) + U(ref=B, grp=A) = n_A n_B, and the two-sided p-values agree (a size-independent identity)."""
    codes = c2["codes"]
    gA = c2["gc"](codes, G, False)                      # reference = group 0
    u_ab = _planes(c2, gA, ub=256)[1][7].cpu().numpy()  # U of the reference vs group 7
    p_ab = _planes(c2, gA, ub=256)[0][7].cpu().numpy()
    from illico_amd.utils.groups import GroupContainer
    gB = GroupContainer(gA.encoded_groups, gA.counts, gA.indices, gA.indptr, 7)
    out = _planes(c2, gB, ub=256)
    u_ba, p_ba = out[1][0].cpu().numpy(), out[0][0].cpu().numpy()
    np.testing.assert_array_equal(u_ab + u_ba, float(gA.counts[0]) * float(gA.counts[7]))
    np.testing.assert_allclose(p_ab, p_ba, rtol=1e-12, atol=0)


def test_c4_ovr_rank_sum_checksum_and_spot_checks(c2):
    """OVR: per gene sum_g ranksum_g = N(N+1)/2, with ranksum_g = n_rest n_g + n_g(n_g+1)/2 - U_g (dense_ovr.py:57-61)."""
    torch = c2["torch"]
    grpc = c2["gc"](c2["codes"], G, True)
    p, u, fc = _planes(c2, grpc)
    n_g = torch.from_numpy(grpc.counts).cuda().double().unsqueeze(1)
    ranksum = (N - n_g) * n_g + n_g * (n_g + 1) / 2 - u
    total = ranksum.sum(0)
    assert bool((total == N * (N + 1) / 2).all())
    assert bool(((p >= 0) & (p <= 1)).all())
    cols = sorted(set(one_gene_per_tile(M) + [3, M - 2]))   # one gene of every tile against the oracle
    Xs = c2["X"][:, cols].contiguous().cpu().numpy()
    want = oracle.run(Xs, grpc, batch_size=1, n_threads=ORACLE_THREADS)
    got = tuple(a[:, cols].cpu().numpy() for a in (p, u, fc))
    assert_planes_match(got, want, what="C4 OVR, one gene per tile")


def test_c3_sparse_formats_equal_dense_on_a_gene_slice(c2):
    """CSC and CSR inputs give the same planes as the dense input of the same data (90 % zeros), OVO and OVR -- and twelve genes of each
    against the CPU oracle on the CSC arrays (the reference's own sparse path)."""
    torch = c2["torch"]
    from scipy import sparse
    Xd = c2["X"][:, 1000:1192].contiguous()
    Xd = Xd * (torch.rand(Xd.shape, device=Xd.device) < 0.2)   # ~90 % zeros overall
    Xh = Xd.cpu().numpy()
    eng = c2["eng"]
    for ovr in (False, True):
        grpc = c2["gc"](c2["codes"], G, ovr)
        dense = [a.cpu().numpy() for a in _planes(c2, grpc, X=Xd)]
        for fmt, ctor in (("csc", sparse.csc_matrix), ("csr", sparse.csr_matrix)):
            Ms = ctor(Xh)
            eng.set_groups(grpc)
            got = eng.run_sparse(fmt, Ms.data, Ms.indices, Ms.indptr, Ms.shape, 0, Ms.shape[1])
            # U and the fold change bit for bit.  p: bit for bit for OVO; for OVR the reference's own sparse and dense paths differ in the
            # last bits once n0^3 leaves 53 bits (270 000 zeros here): sparse adds `n0**3 - n0` in float64 (sparse_ovr.py:49,83), dense
            # adds exact integers block by block (ranking.py:30-47) -- the engine follows each (kernels_finalize.h: tie_f64_sparse)
            np.testing.assert_array_equal(got[1], dense[1], err_msg=f"{fmt} ovr={ovr}")
            np.testing.assert_array_equal(got[2], dense[2], err_msg=f"{fmt} ovr={ovr}")
            if ovr: np.testing.assert_allclose(got[0], dense[0], rtol=1e-12, atol=0, err_msg=f"{fmt} ovr={ovr}")
            else: np.testing.assert_array_equal(got[0], dense[0], err_msg=f"{fmt} ovr={ovr}")
            if fmt == "csc": sp_planes = got
        cols = list(range(3, 192, 16))
        want = oracle.run(sparse.csc_matrix(Xh[:, cols]), grpc, batch_size=1, n_threads=len(cols))
        assert_planes_match(tuple(a[:, cols] for a in sp_planes), want, ref_row=None if ovr else 0, what=f"C3 slice vs oracle, ovr={ovr}")
        want_d = oracle.run(np.ascontiguousarray(Xh[:, cols]), grpc, batch_size=1, n_threads=len(cols))
        assert_planes_match(tuple(a[:, cols] for a in dense), want_d, ref_row=None if ovr else 0, what=f"C3 slice (dense) vs oracle, ovr={ovr}")


def _continuous_slice(c2, lb, ub):
    """Normalised-like values on a gene slice of the C2 matrix: log1p(counts * U(0.5, 1.5)), zeros kept (bench.py recipe)."""
    torch = c2["torch"]
    gen = torch.Generator(device=c2["X"].device)
    gen.manual_seed(1234)
    blk = c2["X"][:, lb:ub]
    return torch.log1p(blk * torch.empty_like(blk).uniform_(0.5, 1.5, generator=gen)).contiguous()


def test_c2_continuous_ovo_packed_route_equals_transpose_route(c2):
    """Full-size groups (300k cells, 2000 groups, 10 000 reference cells), continuous values: the packed route (group-wise
    packing + look-ups in the counted bitmap of the reference, kernels_ovo_compact.h) gives the same U and p, bit for bit, as the
    transposition + k_ovo_rank route -- with that route's reference column in value buckets and sorted --; fold changes at
    1e-13 (the group sums are formed by different kernels); sampled genes against the CPU oracle."""
    Xc = _continuous_slice(c2, 2048, 2048 + 192)
    grpc = c2["gc"](c2["codes"], G, False)
    eng = c2["eng"]
    eng.set_option("profile", 1)
    eng.profile_reset()
    try:
        got = [a.cpu().numpy() for a in _planes(c2, grpc, X=Xc)]
        prof = eng.profile_get()
    finally:
        eng.set_option("profile", 0)
    assert "k_group_compact" in prof and "k_ovo_rank_compact" in prof and "k_transpose_permute" not in prof, prof
    for sorted_ref in (0, 1):
        eng.set_option("no_packed_dense", 1)
        eng.set_option("no_ovo_ref_buckets", sorted_ref)
        try:
            want = [a.cpu().numpy() for a in _planes(c2, grpc, X=Xc)]
        finally:
            eng.set_option("no_ovo_ref_buckets", 0)
            eng.set_option("no_packed_dense", 0)
        np.testing.assert_array_equal(got[0], want[0])
        np.testing.assert_array_equal(got[1], want[1])
        np.testing.assert_allclose(got[2], want[2], rtol=1e-13, atol=0, equal_nan=True)
    cols = list(range(0, 192, 12))
    ora = oracle.run(Xc[:, cols].contiguous().cpu().numpy(), grpc, batch_size=1, n_threads=ORACLE_THREADS)
    assert_planes_match(tuple(a[:, cols] for a in got), ora, ref_row=0, what="continuous OVO, 16 genes against the oracle")


def test_c4_continuous_ovr_parts_route_equals_general_route(c2):
    """Full-size OVR on continuous values: the value-range parts route (6 parts per gene here) gives the same U and p,
    bit for bit, as the general route (segmented radix sort + sweeps); rank sums add up to N(N+1)/2 per gene; sampled
    genes against the CPU oracle.  Fold changes are compared at 1e-12 (value sums are accumulated in a different order)."""
    torch = c2["torch"]
    Xc = _continuous_slice(c2, 4096, 4096 + 192)
    grpc = c2["gc"](c2["codes"], G, True)
    eng = c2["eng"]
    eng.set_option("profile", 1)
    eng.profile_reset()
    try:
        gp, gu, gfc = _planes(c2, grpc, X=Xc)
        prof = eng.profile_get()
    finally:
        eng.set_option("profile", 0)
    assert "k_ovr_partition" in prof and "k_ovr_gene" not in prof, prof
    n_g = torch.from_numpy(grpc.counts).cuda().double().unsqueeze(1)
    ranksum = (N - n_g) * n_g + n_g * (n_g + 1) / 2 - gu
    assert bool((ranksum.sum(0) == N * (N + 1) / 2).all())
    eng.set_option("no_ovr_parts_path", 1)
    try:
        wp, wu, wfc = _planes(c2, grpc, X=Xc)
    finally:
        eng.set_option("no_ovr_parts_path", 0)
    np.testing.assert_array_equal(gu.cpu().numpy(), wu.cpu().numpy())
    np.testing.assert_array_equal(gp.cpu().numpy(), wp.cpu().numpy())
    np.testing.assert_allclose(gfc.cpu().numpy(), wfc.cpu().numpy(), rtol=1e-12, atol=0)
    cols = list(range(5, 192, 12))
    ora = oracle.run(Xc[:, cols].contiguous().cpu().numpy(), grpc, batch_size=1, n_threads=ORACLE_THREADS)
    got = tuple(a[:, cols].cpu().numpy() for a in (gp, gu, gfc))
    assert_planes_match(got, ora, what="continuous OVR, 16 genes against the oracle")


def test_c3_continuous_csc_ovr_single_kernel_equals_general_route(c2):
    """C3-shaped CSC slice (90 % zeros, continuous values, ~30 000 stored entries per gene): k_csc_ovr_gene (value
    buckets in LDS) == the general route, bit for bit on U and p; the sorted form of the same kernel as well."""
    torch = c2["torch"]
    from scipy import sparse
    Xc = _continuous_slice(c2, 6000, 6000 + 96)
    Xc = Xc * (torch.rand(Xc.shape, device=Xc.device) < 0.2)
    Ms = sparse.csc_matrix(Xc.cpu().numpy())
    grpc = c2["gc"](c2["codes"], G, True)
    eng = c2["eng"]
    eng.set_groups(grpc)
    run = lambda: eng.run_sparse("csc", Ms.data, Ms.indices, Ms.indptr, Ms.shape, 0, Ms.shape[1])
    got = run()
    for opt in ("no_csc_ovr_gene_path", "csc_ovr_sorted_form"):
        eng.set_option(opt, 1)
        try:
            want = run()
        finally:
            eng.set_option(opt, 0)
        np.testing.assert_array_equal(got[1], want[1], err_msg=opt)
        np.testing.assert_array_equal(got[0], want[0], err_msg=opt)
        np.testing.assert_allclose(got[2], want[2], rtol=1e-12, atol=0, err_msg=opt)
    cols = list(range(2, 96, 8))   # twelve genes against the oracle's own CSC path
    ora = oracle.run(Ms[:, cols].tocsc(), grpc, batch_size=1, n_threads=len(cols))
    assert_planes_match(tuple(a[:, cols] for a in got), ora, what="continuous CSC OVR, 12 genes against the oracle")


# ---- BASELINE configs[2]: the full-size C3 matrix (CSC, 90 % zeros, nnz ~ 2.3e8) against the oracle on CSC input ----
@pytest.fixture(scope="module")
def c3():
    import torch
    from bench import compress, group_container, make_labels, make_matrix
    from illico_amd._lib import Engine
    dev = torch.device("cuda", 0)
    codes = make_labels(N, G, 0)
    X = make_matrix(torch, N, M, 0.9, 0, dev)
    data, indices, indptr = compress(torch, X, "csc")
    del X
    torch.cuda.empty_cache()
    eng = Engine(0)
    yield {"torch": torch, "csc": (data, indices, indptr), "codes": codes, "eng": eng, "gc": group_container}
    eng.close()


def _csc_columns_host(c3, cols):
    """scipy CSC matrix [N, len(cols)] of a few genes of the device-resident C3 matrix."""
    from scipy import sparse
    data, indices, indptr = c3["csc"]
    ip = indptr.cpu().numpy().astype(np.int64)
    d, i, p = [], [], [0]
    for c in cols:
        s, e = int(ip[c]), int(ip[c + 1])
        d.append(data[s:e].cpu().numpy()); i.append(indices[s:e].cpu().numpy()); p.append(p[-1] + e - s)
    return sparse.csc_matrix((np.concatenate(d), np.concatenate(i), np.array(p)), shape=(N, len(cols)))


@pytest.mark.parametrize("test", ["ovo", "ovr"])
def test_c3_full_size_csc_against_the_oracle_on_csc_input(c3, test):
    """The whole M = 8000 CSC matrix through illico_run_csc (k_csc_counts), OVO and OVR; sampled genes are checked against
    the oracle's own CSC path on the same CSC arrays (not against the HIP dense result)."""
    torch = c3["torch"]
    data, indices, indptr = c3["csc"]
    assert data.numel() > 2.0e8
    ovr = test == "ovr"
    grpc = c3["gc"](c3["codes"], G, ovr)
    eng = c3["eng"]
    eng.set_groups(grpc)
    eng.set_option("profile", 1)
    eng.profile_reset()
    try:
        p, u, fc = eng.run_sparse("csc", data, indices, indptr, (N, M), 0, M, device_out=True)
        torch.cuda.synchronize()
        prof = eng.profile_get()
    finally:
        eng.set_option("profile", 0)
    assert "k_csc_counts" in prof, prof
    assert bool(((p >= 0) & (p <= 1)).all())
    n_g = torch.from_numpy(grpc.counts).cuda().double().unsqueeze(1)
    if ovr:   # per gene the rank sums add up to N (N + 1) / 2 (dense_ovr.py:57-61)
        ranksum = (N - n_g) * n_g + n_g * (n_g + 1) / 2 - u
        assert bool((ranksum.sum(0) == N * (N + 1) / 2).all())
    else:
        n_ref = float(grpc.counts[0])
        assert bool((u[1:] >= 0).all()) and bool((u[1:] <= n_ref * n_g[1:]).all()) and bool(((2 * u[1:]) == torch.round(2 * u[1:])).all())
        assert bool((p[0] == 1).all()) and bool((u[0] == -1).all())
    cols = sorted(set(np.linspace(0, M - 1, 32).astype(int).tolist() + [17, 5001]))   # 32+ genes
    want = oracle.run(_csc_columns_host(c3, cols), grpc, batch_size=1, n_threads=ORACLE_THREADS)
    got = tuple(a[:, cols].cpu().numpy() for a in (p, u, fc))
    assert_planes_match(got, want, ref_row=None if ovr else 0, what=f"C3 {test} spot check vs the oracle's CSC path")


# ---- BASELINE configs[4]: a gene slice of one GPU's C5 shard: 1M cells x 5000 groups, n_ref = 33 333 ----
@pytest.fixture(scope="module")
def c5():
    import torch
    from bench import group_container, make_labels, make_matrix
    from illico_amd._lib import Engine
    N5, M5, G5 = 1_000_000, 3_750, 5_000   # the shard one of 8 GPUs owns: 30 000 / 8 genes, 15 GB
    dev = torch.device("cuda", 0)
    codes = make_labels(N5, G5, 0)
    X = make_matrix(torch, N5, M5, 0.5, 5, dev)
    eng = Engine(0)
    yield {"torch": torch, "X": X, "codes": codes, "eng": eng, "gc": group_container, "shape": (N5, M5, G5)}
    eng.close()


@pytest.mark.parametrize("test", ["ovo", "ovr"])
def test_c5_shard_properties_and_oracle_spot_checks(c5, test):
    torch = c5["torch"]
    N5, M5, G5 = c5["shape"]
    ovr = test == "ovr"
    grpc = c5["gc"](c5["codes"], G5, ovr)
    assert int(grpc.counts[0]) == 33_333
    eng = c5["eng"]
    eng.set_groups(grpc)
    p, u, fc = eng.run_dense(c5["X"], 0, M5, device_out=True)
    torch.cuda.synchronize()
    n_g = torch.from_numpy(grpc.counts).cuda().double().unsqueeze(1)
    assert bool(((p >= 0) & (p <= 1)).all())
    if ovr:
        ranksum = (N5 - n_g) * n_g + n_g * (n_g + 1) / 2 - u
        assert bool((ranksum.sum(0) == N5 * (N5 + 1) / 2).all())
        assert bool(torch.isfinite(fc).all())
    else:
        n_ref = 33_333.0
        assert bool((u[1:] >= 0).all()) and bool((u[1:] <= n_ref * n_g[1:]).all()) and bool(((2 * u[1:]) == torch.round(2 * u[1:])).all())
        assert bool((p[0] == 1).all()) and bool((u[0] == -1).all()) and bool(torch.isfinite(fc[1:]).all())
        # U(ref = A, grp = B) + U(ref = B, grp = A) = n_A n_B on a few columns (size-independent identity)
        from illico_amd.utils.groups import GroupContainer
        gB = GroupContainer(grpc.encoded_groups, grpc.counts, grpc.indices, grpc.indptr, 11)
        eng.set_groups(gB)
        u_ba = eng.run_dense(c5["X"], 0, 64, device_out=True)[1][0].cpu().numpy()
        np.testing.assert_array_equal(u[11, :64].cpu().numpy() + u_ba, float(grpc.counts[0]) * float(grpc.counts[11]))
    cols = sorted(set(np.linspace(0, M5 - 1, 32).astype(int).tolist()))   # 32 genes over the shard's 59 tiles
    Xs = c5["X"][:, cols].contiguous().cpu().numpy()
    want = oracle.run(Xs, grpc, batch_size=1, n_threads=ORACLE_THREADS)
    got = tuple(a[:, cols].cpu().numpy() for a in (p, u, fc))
    assert_planes_match(got, want, ref_row=None if ovr else 0, what=f"C5 shard {test} spot check")


# ---- heavy-tailed counts at C2 shape: the fused pass, its 256-value second pass and the two-pass routes in ONE window ----
@pytest.mark.parametrize("test", ["ovo", "ovr"])
def test_c2_heavy_tailed_counts_mix_every_dense_route(test):
    """`bench.py --values nb`: log-normal gene means -- about a fifth of the genes hold counts beyond 63 (second, 256-value pass
    of the fused route), a few per cent beyond 255 (two-pass routes), the rest take the 64-value pass -- what the highly
    expressed genes of a real count matrix do to the route selection (the reference itself distinguishes count from normalised
    data, README.md:99).  Genes of every class are checked against the oracle."""
    import torch
    from bench import group_container, make_labels, make_matrix
    from illico_amd._lib import Engine
    dev = torch.device("cuda", 0)
    ovr = test == "ovr"
    codes = make_labels(N, G, 0)
    grpc = group_container(codes, G, ovr)
    X = make_matrix(torch, N, M, 0.5, 7, dev, values="nb")
    mx = X.max(dim=0).values.cpu().numpy()
    small, mid, big = np.flatnonzero(mx <= 63), np.flatnonzero((mx > 63) & (mx <= 255)), np.flatnonzero(mx > 255)
    assert 0.10 * M < mid.size + big.size < 0.40 * M and big.size > 0.01 * M, (small.size, mid.size, big.size)
    eng = Engine(0)
    try:
        eng.set_stream(torch.cuda.current_stream().cuda_stream)
        eng.set_groups(grpc)
        eng.profile(True)
        eng.profile_reset()
        p, u, fc = eng.run_dense(X, 0, M, device_out=True)
        torch.cuda.synchronize()
        prof = eng.profile_get()
        eng.profile(False)
        assert "k_ovo_fused_wide" in prof, prof   # the 256-value pass ran (OVO and OVR share the id)
        n_g = torch.from_numpy(grpc.counts).cuda().double().unsqueeze(1)
        assert bool(((p >= 0) & (p <= 1)).all())
        if ovr:
            ranksum = (N - n_g) * n_g + n_g * (n_g + 1) / 2 - u
            assert bool((ranksum.sum(0) == N * (N + 1) / 2).all())
        else:
            n_ref = float(grpc.counts[0])
            assert bool((u[1:] >= 0).all()) and bool((u[1:] <= n_ref * n_g[1:]).all()) and bool(((2 * u[1:]) == torch.round(2 * u[1:])).all())
            assert bool((p[0] == 1).all()) and bool((u[0] == -1).all())
        rng = np.random.RandomState(3)
        cols = sorted(set(rng.choice(small, 12, replace=False).tolist() + rng.choice(mid, 12, replace=False).tolist() +
                          rng.choice(big, min(8, big.size), replace=False).tolist()))
        want = oracle.run(X[:, cols].contiguous().cpu().numpy(), grpc, batch_size=1, n_threads=ORACLE_THREADS)
        got = tuple(a[:, cols].cpu().numpy() for a in (p, u, fc))
        assert_planes_match(got, want, ref_row=None if ovr else 0, what=f"heavy-tailed counts {test}")
    finally:
        eng.close()
