// Host-side drivers of the CSC / CSR routes, templated on the value / index types: instantiated in sparse_<type>.hip.
#pragma once

#include "keyed_driver.h"
#include "host_narrow.h"
#include <thread>
// ---- order-independent value sums (kernels_sums.h) ----
template <typename InT, typename IdxT>
static int launch_csc_value_sums(illico_ctx *c, const InT *d_data, const IdxT *d_indices, const IdxT *d_indptr, int64_t kshift, int64_t col0,
                                 const int *d_cols, const int *d_codes, int nb, int dtype, int flags, double *ssum) {
    CscSumsParams P;
    P.data = d_data; P.indices = d_indices; P.indptr = d_indptr; P.kshift = kshift; P.col0 = col0; P.gene_cols = d_cols; P.codes = d_codes; P.codes16 = d_codes ? c->d_codes16 : nullptr;
    P.nb = nb; P.G = (int)c->n_groups; P.dt = dtype; P.is_log1p = (flags & ILLICO_FLAG_LOG1P) ? 1 : 0; P.acc_global = nullptr; P.out_sum = ssum;
    // accumulators in LDS while they fit at all (one workgroup per CU beyond 5000 groups); in HBM through global atomics otherwise:
    // 46 times slower at 10 000 groups (29 ms against 0.65 at C3 shape), which is where the threshold used to sit
    const bool accg = csc_sums_lds_bytes(P.G, false) + 2048 > kMaxLds;
    const size_t lds = csc_sums_lds_bytes(P.G, accg);
    if (accg) {
        void *v;
        int rc = get_scratch(c, "sums_acc", (size_t)nb * 2 * P.G * 8, &v);
        if (rc) return rc;
        P.acc_global = (long long *)v;
        HIPCHK(c, hipMemsetAsync(v, 0, (size_t)nb * 2 * P.G * 8, c->stream));
    }
    ProfScope ps(c, KID_VALUE_SUMS);
    if (accg) {
        auto kern = k_csc_value_sums<InT, IdxT, true>;
        hipLaunchKernelGGL(kern, dim3(nb), dim3(CSUM_NT), lds, c->stream, P);
    } else {
        auto kern = k_csc_value_sums<InT, IdxT, false>;
        HIPCHK(c, hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        hipLaunchKernelGGL(kern, dim3(nb), dim3(CSUM_NT), lds, c->stream, P);
    }
    HIPCHK(c, hipGetLastError());
    return ILLICO_OK;
}
// (run_fused_ovo<uint8_t> / <float> -- the fused kernels on the dense windows of count-valued CSR input -- are instantiated in
// dense_u8.hip / dense_f32.hip: only their declaration, engine.h, is visible here)
static size_t seg_lds_bytes(int G) { return (size_t)((G + 3) & ~3) * 4 + SEG_NT * 4; }

struct SparseBatch {
    int64_t g0, g1;   // gene range (absolute column indices)
    int64_t nnz;      // stored entries in the range
    int64_t max_gene; // largest per-gene nnz in the range
};

// split [col_lb, col_ub) so that each batch's scratch stays under the cap and its nnz below 2^31
static std::vector<SparseBatch> plan_batches(const std::vector<int64_t> &gene_nnz, int64_t col_lb, size_t per_nnz,
                                             size_t per_gene, int64_t gene_batch, size_t cap) {
    std::vector<SparseBatch> out;
    const int64_t W = (int64_t)gene_nnz.size();
    int64_t i = 0;
    while (i < W) {
        SparseBatch b{col_lb + i, col_lb + i, 0, 0};
        size_t bytes = 0;
        while (i < W) {
            int64_t c = gene_nnz[i];
            size_t add = (size_t)c * per_nnz + per_gene;
            bool full = (b.g1 > b.g0) && (bytes + add > cap || b.nnz + c > 0x7FFF0000ll || (gene_batch > 0 && b.g1 - b.g0 >= gene_batch));
            if (full) break;
            bytes += add;
            b.nnz += c;
            b.max_gene = std::max(b.max_gene, c);
            b.g1 += 1;
            ++i;
        }
        out.push_back(b);
    }
    return out;
}

// Single-kernel CSC OVO route over genes [g0, g1): statistics + finalize for every gene it can take; the genes it
// cannot take come back as column runs for the two-kernel route.
// device copy of a column list (absolute indices) for the list-driven kernels / k_finalize's col_map
static int upload_cols(illico_ctx *c, const std::vector<int64_t> &cols, const int **d_cols) {
    void *v;
    int rc = get_scratch(c, "sp_collist", std::max<size_t>(cols.size(), 1) * 4, &v);
    if (rc) return rc;
    std::vector<int> h(cols.begin(), cols.end());
    HIPCHK(c, hipMemcpyAsync(v, h.data(), h.size() * 4, hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream)); // h goes out of scope
    *d_cols = (const int *)v;
    return ILLICO_OK;
}

// Groups per launch of k_csc_counts: all of them while their tables fit LDS the way the kernel likes it (mixed cells: two workgroups
// per CU), else equal windows of about 2000 groups (16-bit cells: about 1100), one launch per window over the same entries.
static int cscc_group_window(int G, bool w16) {
    // (16-bit cells: windows rather than a 32-value table -- genes with a value of 32 .. 63 would leave the route)
    const bool one = w16 ? cscc_lds_bytes16(G, 64) + 8192 <= kMaxLds : cscc_lds_bytes(G, 32) + 8192 <= kMaxLds;
    if (one) return G;
    const int per = w16 ? 1100 : 2000;
    const int k = (G + per - 1) / per;
    return (G + k - 1) / k;
}

// Count-valued CSC genes with small groups: per-group value histograms in LDS (k_csc_counts), OVO and OVR.  `cols` in:
// the genes to compute; out: the genes it could not take.  First the mixed 8- / 4-bit cells (two workgroups per CU); the
// genes where a 4-bit cell overflowed are redone with 8-bit cells; genes with values outside the table are left to the
// general routes.
template <typename InT, typename IdxT, bool MIXED>
static int launch_csc_counts(illico_ctx *c, const CscCountsParams &P, int rt, bool has_big, bool ovr, size_t lds, bool w16 = false) {
    ProfScope ps(c, KID_CSC_COUNTS);
    const bool win = P.G != P.G_total; // a window of the groups (instantiated for 16-bit group codes only: the host's case)
    if (win && !P.codes16) return fail(c, ILLICO_ERR_UNSUPPORTED, "group windows need the 16-bit code table");
    if (w16) { // 16-bit cells for every group (more than CSCC_MAX_BIG groups above 255 cells)
#define CSCC_LAUNCH16(OVRF, RTV, C16F, WINF)                                                                                    \
    do {                                                                                                                   \
        auto kern = k_csc_counts<InT, IdxT, OVRF, RTV, false, false, C16F, CSCC_WT, 0, CSCC_NT, CSCC_LEAN, false, true, WINF>; \
        HIPCHK(c, hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));          \
        hipLaunchKernelGGL(kern, dim3(P.nb), dim3(CSCC_NT), lds, c->stream, P);                                            \
    } while (0)
#define CSCC_LAUNCH16B(OVRF, RTV) do { if (win) CSCC_LAUNCH16(OVRF, RTV, true, true); else if (P.codes16) CSCC_LAUNCH16(OVRF, RTV, true, false); else CSCC_LAUNCH16(OVRF, RTV, false, false); } while (0)
        if (ovr) { if (rt == 64) CSCC_LAUNCH16B(true, 64); else CSCC_LAUNCH16B(true, 32); }
        else { if (rt == 64) CSCC_LAUNCH16B(false, 64); else CSCC_LAUNCH16B(false, 32); }
#undef CSCC_LAUNCH16B
#undef CSCC_LAUNCH16
        HIPCHK(c, hipGetLastError());
        return ILLICO_OK;
    }
#define CSCC_LAUNCH1(OVRF, RTV, BIG, C16F, WINF)                                                                           \
    do {                                                                                                                   \
        auto kern = k_csc_counts<InT, IdxT, OVRF, RTV, BIG, MIXED && RTV == 64, C16F, CSCC_WT, 0, CSCC_NT, CSCC_LEAN, CSCC_PUTB(OVRF), false, WINF>; \
        HIPCHK(c, hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));          \
        hipLaunchKernelGGL(kern, dim3(P.nb), dim3(CSCC_NT), lds, c->stream, P);                                            \
    } while (0)
#define CSCC_LAUNCH(OVRF, RTV, BIG) do { if (win) CSCC_LAUNCH1(OVRF, RTV, BIG, true, true); else if (P.codes16) CSCC_LAUNCH1(OVRF, RTV, BIG, true, false); else CSCC_LAUNCH1(OVRF, RTV, BIG, false, false); } while (0)
#define CSCC_LAUNCH2(OVRF, RTV) do { if (has_big) CSCC_LAUNCH(OVRF, RTV, true); else CSCC_LAUNCH(OVRF, RTV, false); } while (0)
    if (ovr) { if (rt == 64) CSCC_LAUNCH2(true, 64); else CSCC_LAUNCH2(true, 32); }
    else { if (rt == 64) CSCC_LAUNCH2(false, 64); else CSCC_LAUNCH2(false, 32); }
#undef CSCC_LAUNCH2
#undef CSCC_LAUNCH
#undef CSCC_LAUNCH1
    HIPCHK(c, hipGetLastError());
    return ILLICO_OK;
}

template <typename InT, typename IdxT>
static int run_csc_counts_route(illico_ctx *c, const InT *d_data, const IdxT *d_indices, const IdxT *d_indptr, int64_t kshift, const int *d_codes,
                                int64_t n_rows, int64_t col_lb, int flags, int alternative, const OutPlanes &o, std::vector<int64_t> &cols) {
    const int G = (int)c->n_groups;
    const bool ovr = c->ref < 0;
    int rc;
    void *v;
    // groups of more than 255 cells (other than the OVO reference, which has its own table) get 32-bit rows
    std::vector<signed char> h_slot(G, (signed char)-1);
    int n_big = 0;
    for (int g = 0; g < G; ++g)
        if (g != c->ref && c->h_counts[g] > 255) h_slot[g] = (signed char)std::min(n_big++, 127);
    const signed char *d_slot = nullptr;
    if (n_big && n_big <= CSCC_MAX_BIG) {
        if ((rc = get_scratch(c, "cscc_slot", (size_t)G, &v))) return rc;
        HIPCHK(c, hipMemcpyAsync(v, h_slot.data(), (size_t)G, hipMemcpyHostToDevice, c->stream));
        HIPCHK(c, hipStreamSynchronize(c->stream));
        d_slot = (const signed char *)v;
    }
    // more big groups than the side table holds: 16-bit cells for every group, one pass
    // (... or an OVO reference of 30 000 cells or more: the sweep's 32-bit terms -- 3 tS^2 -- would overflow; the 16-bit form's are 64-bit)
    const bool w16 = n_big > CSCC_MAX_BIG || (!ovr && c->h_counts[c->ref] >= 30000);
    if (w16) { n_big = 0; d_slot = nullptr; }
    // more groups than LDS holds tables for: windows of Gw groups, one launch each (CscCountsParams::g_lo)
    const int Gw = cscc_group_window(G, w16);
    const int rt16 = cscc_lds_bytes16(Gw, 64) + 8192 <= kMaxLds ? 64 : 32;
    const int rt8 = w16 ? rt16 : (cscc_lds_bytes(Gw, 64) + 8192 <= kMaxLds ? 64 : 32);
    // the mixed layout pays when two workgroups fit a CU
    // (... or when 64 bytes per group do not fit at all: the mixed table still holds all 63 values where the 8-bit form
    //  would drop to 31)
    const bool try_mixed = !w16 && !c->no_csc_counts_mixed && (2 * (cscc_lds_bytes(Gw, 0) + 4096) <= kMaxLds ||
                                                               (rt8 == 32 && cscc_lds_bytes(Gw, 0) + 8192 <= kMaxLds));
    const u16 *codes16 = d_codes ? c->d_codes16 : nullptr; // (sparse input holds fewer than 65 536 groups: the 16-bit table exists)
    if (d_codes && !codes16) return ILLICO_OK; // every gene stays in `cols`
    std::vector<int64_t> left;
    // pass 0: mixed cells over every gene; pass 1: 8-bit cells over the genes whose 4-bit cells overflowed (or over every
    // gene when the mixed form is not used)
    for (int pass = try_mixed ? 0 : 1; pass < 2 && !cols.empty(); ++pass) {
        const bool mixed = pass == 0;
        const int rt = mixed ? 64 : rt8;
        const size_t lds = w16 ? cscc_lds_bytes16(Gw, rt) : cscc_lds_bytes(Gw, mixed ? 0 : rt);
        const bool contiguous = cols.back() - cols.front() + 1 == (int64_t)cols.size();
        const int *d_cols = nullptr;
        if (!contiguous && (rc = upload_cols(c, cols, &d_cols))) return rc;
        const int64_t n = (int64_t)cols.size();
        const int64_t nb_max = std::max<int64_t>(1, std::min<int64_t>(n, (int64_t)((size_t)(4ll << 30) / ((size_t)G * 24 + 16))));
        if ((rc = get_scratch(c, "stats", (size_t)nb_max * G * 24 + (size_t)nb_max * 8, &v))) return rc;
        long long *s2u = (long long *)v;
        u64 *stie = (u64 *)(s2u + (size_t)nb_max * G);
        double *ssum = (double *)(stie + (size_t)nb_max * G);
        double *gtot = ssum + (size_t)nb_max * G;
        if ((rc = get_scratch(c, "gene_flags", (size_t)nb_max * 4, &v))) return rc;
        u32 *fb = (u32 *)v;
        std::vector<int64_t> redo;
        for (int64_t b0 = 0; b0 < n; b0 += nb_max) {
            const int nb = (int)std::min<int64_t>(nb_max, n - b0);
            HIPCHK(c, hipMemsetAsync(fb, 0, (size_t)nb * 4, c->stream));
            CscCountsParams P;
            P.data = d_data; P.indices = d_indices; P.indptr = d_indptr; P.kshift = kshift; P.col0 = cols[b0];
            P.gene_cols = d_cols ? d_cols + b0 : nullptr; P.nb = nb; P.codes16 = codes16; P.counts = c->d_counts; P.G = G; P.ref = (int)c->ref;
            P.n_cells = n_rows; P.fallback = fb; P.out_2u = s2u; P.out_tie = stie; P.out_sum = ssum; P.big_slot = d_slot;
            P.gene_total = ovr ? gtot : nullptr;
            P.tie_f64 = ovr ? 1 : 0; // (the reference's sparse OVR arithmetic, kernels_finalize.h: tie_f64_sparse)
            P.verdict = nullptr;
            const bool pack16 = n_big == 0 && !w16; // 16-byte statistics while every ranked group has at most 255 cells
            P.pack16 = pack16 ? 1 : 0;
            P.G_total = G;
            for (int g_lo = 0; g_lo < G; g_lo += Gw) { // (one window unless the groups outgrow LDS)
                P.g_lo = g_lo; P.G = std::min(Gw, G - g_lo);
                if (mixed) { if ((rc = launch_csc_counts<InT, IdxT, true>(c, P, rt, n_big > 0, ovr, lds))) return rc; }
                else if ((rc = launch_csc_counts<InT, IdxT, false>(c, P, rt, n_big > 0, ovr, lds, w16))) return rc;
            }
            if (d_cols) { if ((rc = launch_finalize(c, s2u, stie, ssum, ovr ? gtot : nullptr, nb, flags, alternative, o.p, o.u, o.fc, o.ld, -col_lb, d_cols + b0, pack16, ovr))) return rc; }
            else if ((rc = launch_finalize(c, s2u, stie, ssum, ovr ? gtot : nullptr, nb, flags, alternative, o.p, o.u, o.fc, o.ld, cols[b0] - col_lb, nullptr, pack16, ovr))) return rc;
            if (c->pinned_bytes < (size_t)nb * 4) {
                if (c->pinned) hipHostFree(c->pinned);
                c->pinned = nullptr; c->pinned_bytes = 0;
                HIPCHK(c, hipHostMalloc(&c->pinned, (size_t)nb * 4 + 4096, hipHostMallocDefault));
                c->pinned_bytes = (size_t)nb * 4 + 4096;
            }
            HIPCHK(c, hipMemcpyAsync(c->pinned, fb, (size_t)nb * 4, hipMemcpyDeviceToHost, c->stream));
            HIPCHK(c, hipStreamSynchronize(c->stream));
            const u32 *h_fb = (const u32 *)c->pinned;
            for (int64_t j = 0; j < nb; ++j) {
                if (h_fb[j] == 2u) redo.push_back(cols[b0 + j]);
                else if (h_fb[j]) left.push_back(cols[b0 + j]);
            }
        }
        cols.swap(redo);
    }
    std::sort(left.begin(), left.end());
    cols.swap(left);
    return ILLICO_OK;
}

// ILLICO_FLAG_DEFER on device-resident CSC arrays with device planes: the count-valued pass (value sample, k_csc_counts,
// k_finalize) is enqueued and the call returns -- no host wait at all.  Whether the window is count-valued is decided on the
// device from the sample; which genes the pass could not take (values outside the table, 4-bit cells that overflowed) travels
// to pinned memory behind an event and is looked at by the next call on the context / illico_ctx_synchronize
// (resolve_pending_csc), which recomputes exactly those columns through the ordinary routes.
template <typename InT, typename IdxT>
static int run_csc_counts_deferred(illico_ctx *c, const void *data, const void *indices, const void *indptr, int dtype, int idx_dtype,
                                   int64_t n_rows, int64_t n_cols, int64_t col_lb, int64_t col_ub, int flags, int alternative,
                                   const OutPlanes &o) {
    const int G = (int)c->n_groups;
    const bool ovr = c->ref < 0;
    const int64_t W = col_ub - col_lb;
    int rc;
    void *v;
    std::vector<signed char> h_slot(G, (signed char)-1);
    int n_big = 0;
    for (int g = 0; g < G; ++g)
        if (g != c->ref && c->h_counts[g] > 255) h_slot[g] = (signed char)std::min(n_big++, 127);
    const signed char *d_slot = nullptr;
    if (n_big && n_big <= CSCC_MAX_BIG) { // (rare: a one-off upload + wait; a context keeps its groups for many calls)
        if ((rc = get_scratch(c, "cscc_slot", (size_t)G, &v))) return rc;
        HIPCHK(c, hipMemcpyAsync(v, h_slot.data(), (size_t)G, hipMemcpyHostToDevice, c->stream));
        HIPCHK(c, hipStreamSynchronize(c->stream));
        d_slot = (const signed char *)v;
    }
    if ((rc = get_scratch(c, "flag", 16, &v))) return rc;
    u32 *d_cnt = (u32 *)v;
    HIPCHK(c, hipMemsetAsync(d_cnt, 0, 16, c->stream));
    hipLaunchKernelGGL((k_sample_noncount_cols<InT, IdxT>), dim3((1 << 16) / 256), dim3(256), 0, c->stream, (const InT *)data,
                       (const IdxT *)indptr, (long long)col_lb, (long long)col_ub, 1 << 16, CSCC_RT, d_cnt);
    HIPCHK(c, hipGetLastError());
    const bool w16 = n_big > CSCC_MAX_BIG || (!ovr && c->h_counts[c->ref] >= 30000); // 16-bit cells for every group (run_csc_counts_route)
    if (w16) { n_big = 0; d_slot = nullptr; }
    const int Gw = cscc_group_window(G, w16); // windows of groups when they outgrow LDS (run_csc_counts_route)
    const int rt8 = w16 ? (cscc_lds_bytes16(Gw, 64) + 8192 <= kMaxLds ? 64 : 32) : (cscc_lds_bytes(Gw, 64) + 8192 <= kMaxLds ? 64 : 32);
    const bool mixed = !w16 && !c->no_csc_counts_mixed && (2 * (cscc_lds_bytes(Gw, 0) + 4096) <= kMaxLds ||
                                                           (rt8 == 32 && cscc_lds_bytes(Gw, 0) + 8192 <= kMaxLds));
    const int rt = mixed ? 64 : rt8;
    const size_t lds = w16 ? cscc_lds_bytes16(Gw, rt) : cscc_lds_bytes(Gw, mixed ? 0 : rt);
    const int64_t nb_max = std::max<int64_t>(1, std::min<int64_t>(W, (int64_t)((size_t)(4ll << 30) / ((size_t)G * 24 + 16))));
    if ((rc = get_scratch(c, "stats", (size_t)nb_max * G * 24 + (size_t)nb_max * 8, &v))) return rc;
    long long *s2u = (long long *)v;
    u64 *stie = (u64 *)(s2u + (size_t)nb_max * G);
    double *ssum = (double *)(stie + (size_t)nb_max * G);
    double *gtot = ssum + (size_t)nb_max * G;
    if ((rc = get_scratch(c, "sp_defer_flags", (size_t)W * 4, &v))) return rc;
    u32 *fb = (u32 *)v;
    HIPCHK(c, hipMemsetAsync(fb, 0, (size_t)W * 4, c->stream));
    for (int64_t b0 = 0; b0 < W; b0 += nb_max) {
        const int nb = (int)std::min<int64_t>(nb_max, W - b0);
        CscCountsParams P;
        P.data = data; P.indices = indices; P.indptr = indptr; P.kshift = 0; P.col0 = col_lb + b0; P.gene_cols = nullptr; P.nb = nb;
        P.codes16 = c->d_codes16; P.counts = c->d_counts; P.G = G; P.ref = (int)c->ref; P.n_cells = n_rows; P.fallback = fb + b0;
        P.out_2u = s2u; P.out_tie = stie; P.out_sum = ssum; P.big_slot = d_slot; P.gene_total = ovr ? gtot : nullptr; P.verdict = d_cnt; P.tie_f64 = ovr ? 1 : 0;
        const bool pack16 = n_big == 0 && !w16;
        P.pack16 = pack16 ? 1 : 0;
        P.G_total = G;
        for (int g_lo = 0; g_lo < G; g_lo += Gw) {
            P.g_lo = g_lo; P.G = std::min(Gw, G - g_lo);
            if (mixed) { if ((rc = launch_csc_counts<InT, IdxT, true>(c, P, rt, n_big > 0, ovr, lds))) return rc; }
            else if ((rc = launch_csc_counts<InT, IdxT, false>(c, P, rt, n_big > 0, ovr, lds, w16))) return rc;
        }
        if ((rc = launch_finalize(c, s2u, stie, ssum, ovr ? gtot : nullptr, nb, flags, alternative, o.p, o.u, o.fc, o.ld, b0, nullptr, pack16, ovr))) return rc;
    }
    const int slot = c->pend_next;
    void *&pin = c->pend_pinned[slot];
    if (c->pend_pinned_bytes[slot] < (size_t)W * 4) {
        if (pin) hipHostFree(pin);
        pin = nullptr;
        c->pend_pinned_bytes[slot] = 0;
        HIPCHK(c, hipHostMalloc(&pin, (size_t)W * 4 + 4096, hipHostMallocDefault));
        c->pend_pinned_bytes[slot] = (size_t)W * 4 + 4096;
    }
    if (!c->pend_event[slot]) HIPCHK(c, hipEventCreateWithFlags(&c->pend_event[slot], hipEventDisableTiming));
    HIPCHK(c, hipMemcpyAsync(pin, fb, (size_t)W * 4, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipEventRecord(c->pend_event[slot], c->stream));
    c->pend_next ^= 1;
    PendingDense &q = c->pend;
    q = PendingDense();
    q.on = true; q.kind = 1; q.sp_data = data; q.sp_indices = indices; q.sp_indptr = indptr; q.idx_dtype = idx_dtype; q.n_cols = n_cols;
    q.dtype = dtype; q.flags = flags & ~ILLICO_FLAG_DEFER; q.alternative = alternative; q.slot = slot; q.N = n_rows;
    q.col_lb = col_lb; q.col_ub = col_ub; q.out_ld = o.ld; q.p = o.p; q.u = o.u; q.fc = o.fc;
    return ILLICO_OK;
}

// Single-kernel CSC OVO route over the genes in `cols` (in: to compute; out: the genes it could not take, which go to
// the two-kernel route): statistics + finalize per batch.
template <typename InT, typename IdxT, typename KeyT>
static int run_csc_gene_route(illico_ctx *c, const InT *d_data, const IdxT *d_indices, const IdxT *d_indptr, int64_t kshift, const int *d_codes,
                              int dtype, int64_t col_lb, int flags, int alternative, const OutPlanes &o, std::vector<int64_t> &cols,
                              const std::vector<int64_t> *gene_nnz = nullptr /* stored entries of gene col_lb + j */) {
    if (gene_nnz) { // genes with more entries than the kernel's LDS key buffer would only be flagged by it: when that is most of them (eight-byte
        // keys at C3's 30 000 entries per gene) the launch is skipped altogether
        const int runend_cap0 = (int)std::max<int64_t>(1, std::min<int64_t>(c->h_counts[c->ref], 8192));
        const size_t fixed0 = cscg_lds_bytes((int)c->n_groups, 0, runend_cap0, sizeof(KeyT), false);
        const int64_t cap0 = fixed0 < kMaxLds ? (int64_t)((kMaxLds - fixed0) / sizeof(KeyT)) : 0;
        int64_t fit = 0;
        for (int64_t cc : cols) fit += (*gene_nnz)[cc - col_lb] <= cap0 ? 1 : 0;
        if (fit * 4 < (int64_t)cols.size()) return ILLICO_OK; // every gene stays in `cols` for the two-kernel route
    }
    const int64_t g0 = 0, g1 = (int64_t)cols.size();
    std::vector<int64_t> fallback_cols;
    const bool contiguous = cols.back() - cols.front() + 1 == (int64_t)cols.size();
    const int *d_cols = nullptr;
    {
        int rc0;
        if (!contiguous && (rc0 = upload_cols(c, cols, &d_cols))) return rc0;
    }
    const int G = (int)c->n_groups;
    int rc;
    void *v;
    const int runend_cap = (int)std::max<int64_t>(1, std::min<int64_t>(c->h_counts[c->ref], 8192));
    // bucket form of the reference run (no sort, short look-ups): its 16-bit table takes the run-end region
    const bool ref_buckets = !c->no_ovo_ref_buckets && cscg_lds_bytes(G, 0, runend_cap, sizeof(KeyT), true) + 16384 * sizeof(KeyT) <= kMaxLds;
    const size_t fixed = cscg_lds_bytes(G, 0, runend_cap, sizeof(KeyT), ref_buckets);
    if (fixed + 1024 * sizeof(KeyT) > kMaxLds) return ILLICO_OK; // every gene stays in `cols` for the two-kernel route
    const int key_cap = (int)((kMaxLds - fixed) / sizeof(KeyT));
    const size_t lds = cscg_lds_bytes(G, key_cap, runend_cap, sizeof(KeyT), ref_buckets);
    const int64_t nb_max = std::max<int64_t>(1, std::min<int64_t>(g1 - g0, (int64_t)((size_t)(4ll << 30) / ((size_t)G * 24 + 16))));
    if ((rc = get_scratch(c, "stats", (size_t)nb_max * G * 24 + (size_t)nb_max * 8, &v))) return rc;
    long long *s2u = (long long *)v;
    u64 *stie = (u64 *)(s2u + (size_t)nb_max * G);
    double *ssum = (double *)(stie + (size_t)nb_max * G);
    if ((rc = get_scratch(c, "gene_flags", (size_t)nb_max * 4, &v))) return rc;
    u32 *fb = (u32 *)v;
    std::vector<u32> h_fb;
    auto kern = k_csc_gene<InT, IdxT, KeyT>;
    HIPCHK(c, hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    for (int64_t b0 = g0; b0 < g1; b0 += nb_max) {
        const int nb = (int)std::min<int64_t>(nb_max, g1 - b0);
        HIPCHK(c, hipMemsetAsync(fb, 0, (size_t)nb * 4, c->stream));
        CscGeneParams P;
        P.data = d_data; P.indices = d_indices; P.indptr = d_indptr; P.kshift = kshift; P.col0 = cols[b0];
        P.gene_cols = d_cols ? d_cols + b0 : nullptr; P.nb = nb; P.codes = d_codes; P.codes16 = d_codes ? c->d_codes16 : nullptr;
        P.counts = c->d_counts; P.G = G; P.ref = (int)c->ref; P.dt = dtype; P.is_log1p = (flags & ILLICO_FLAG_LOG1P) ? 1 : 0;
        P.key_cap = key_cap; P.runend_cap = runend_cap; P.ref_buckets = ref_buckets ? 1 : 0; P.fallback = fb; P.out_2u = s2u; P.out_tie = stie; P.out_sum = ssum;
        {
            ProfScope ps(c, KID_CSC_GENE);
            hipLaunchKernelGGL(kern, dim3(nb), dim3(CSCG_NT), lds, c->stream, P);
            HIPCHK(c, hipGetLastError());
        }
        // the kernel adds a group's values in the order its LDS regroup happened to leave them: replace its sums by the
        // order-independent ones
        if ((rc = launch_csc_value_sums<InT, IdxT>(c, d_data, d_indices, d_indptr, kshift, cols[b0], d_cols ? d_cols + b0 : nullptr, d_codes, nb, dtype, flags, ssum))) return rc;
        if (d_cols) { if ((rc = launch_finalize(c, s2u, stie, ssum, nullptr, nb, flags, alternative, o.p, o.u, o.fc, o.ld, -col_lb, d_cols + b0))) return rc; }
        else if ((rc = launch_finalize(c, s2u, stie, ssum, nullptr, nb, flags, alternative, o.p, o.u, o.fc, o.ld, cols[b0] - col_lb))) return rc;
        h_fb.resize(nb);
        HIPCHK(c, hipMemcpyAsync(h_fb.data(), fb, (size_t)nb * 4, hipMemcpyDeviceToHost, c->stream));
        HIPCHK(c, hipStreamSynchronize(c->stream));
        for (int64_t j = 0; j < nb; ++j)
            if (h_fb[j]) fallback_cols.push_back(cols[b0 + j]);
    }
    cols.swap(fallback_cols);
    return ILLICO_OK;
}

// Single-kernel CSC OVR route (any values) over the genes in `cols` (in: to compute; out: the genes with more stored
// entries than the LDS key buffer, which go to the general route): statistics + gene totals + finalize per batch.
template <typename InT, typename IdxT, typename KeyT>
static int run_csc_ovr_route(illico_ctx *c, const InT *d_data, const IdxT *d_indices, const IdxT *d_indptr, int64_t kshift, const int *d_codes,
                             int dtype, int64_t n_rows, int64_t col_lb, int64_t max_nnz, int flags, int alternative, const OutPlanes &o,
                             std::vector<int64_t> &cols) {
    const int G = (int)c->n_groups;
    // the group's stored-entry count rides above bit 40 of its doubled rank sum
    if (n_rows >= (1ll << 31) || (double)c->max_nonref * 2.0 * (double)n_rows >= (double)(1ull << CSCO_CNT_SHIFT) || c->max_nonref >= (1ll << 23))
        return ILLICO_OK; // every gene stays in `cols`
    // 16384 buckets when the largest gene still fits beside them, else 8192
    int lg = 14;
    if (csco_key_cap(G, lg, sizeof(KeyT), kMaxLds) < max_nnz) lg = 13;
    int key_cap = csco_key_cap(G, lg, sizeof(KeyT), kMaxLds);
    // thousands of groups: acc[G] takes the key buffer's place -- the accumulators then live in HBM (global atomics) and the keys keep LDS
    bool accg = false;
    int g_lds = G; // groups whose accumulators stay in LDS
    if (key_cap < max_nnz) {
        int lg2 = 14;
        if (csco_key_cap(G, lg2, sizeof(KeyT), kMaxLds, false, true) < max_nnz) lg2 = 13;
        const int cap2 = csco_key_cap(G, lg2, sizeof(KeyT), kMaxLds, false, true);
        if (cap2 > key_cap) {
            accg = true; lg = lg2;
            // what the largest gene's keys leave of LDS holds the accumulators of the first groups; the others' live in HBM
            const size_t need = csco_fixed_lds_bytes(0, lg, false) + ((size_t)std::min<int64_t>(max_nnz, cap2) + 8) * sizeof(KeyT);
            g_lds = need < kMaxLds ? (int)std::min<size_t>((size_t)G, ((kMaxLds - need) / 8) & ~(size_t)1) : 0;
            key_cap = (int)std::min<size_t>((kMaxLds - csco_fixed_lds_bytes(g_lds, lg, false)) / sizeof(KeyT) - 4, 65535 - 4);
        }
    }
    if (key_cap <= 0) return ILLICO_OK;
    // short columns (a matrix of few cells: 2000 stored entries per gene): no more key slots and buckets than the longest column needs --
    // two workgroups per CU then run side by side (one's barriers under the other's passes)
    if (!accg && !c->no_csc_ovr_small_lds && max_nnz + 64 < key_cap) {
        int lg_s = lg;
        while (lg_s > 11 && (1ll << (lg_s - 1)) >= 2 * max_nnz) --lg_s;
        const int cap_s = (int)std::min<int64_t>(csco_key_cap(G, lg_s, sizeof(KeyT), kMaxLds), std::max<int64_t>((max_nnz + 64 + 1023) & ~1023ll, 2048));
        if (cap_s >= max_nnz) { lg = lg_s; key_cap = cap_s; }
    }
    int rc;
    void *v;
    const bool contiguous = cols.back() - cols.front() + 1 == (int64_t)cols.size();
    const int *d_cols = nullptr;
    if (!contiguous && (rc = upload_cols(c, cols, &d_cols))) return rc;
    const size_t lds = csco_fixed_lds_bytes(g_lds, lg, false, accg) + (size_t)(key_cap + 4) * sizeof(KeyT);
    const int64_t n = (int64_t)cols.size();
    const int64_t nb_max = std::max<int64_t>(1, std::min<int64_t>(n, (int64_t)((size_t)(4ll << 30) / ((size_t)G * 24 + 16))));
    if ((rc = get_scratch(c, "stats", (size_t)nb_max * G * 24 + (size_t)nb_max * 8, &v))) return rc;
    long long *s2u = (long long *)v;
    u64 *stie = (u64 *)(s2u + (size_t)nb_max * G);
    double *ssum = (double *)(stie + (size_t)nb_max * G);
    double *gtot = ssum + (size_t)nb_max * G;
    if ((rc = get_scratch(c, "gene_flags", (size_t)nb_max * 4, &v))) return rc;
    u32 *fb = (u32 *)v;
    auto kern = accg ? k_csc_ovr_gene<InT, IdxT, KeyT, true> : k_csc_ovr_gene<InT, IdxT, KeyT, false>;
    HIPCHK(c, hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    u64 *acc_g = nullptr;
    if (accg) {
        if ((rc = get_scratch(c, "csco_acc", (size_t)nb_max * G * 8, &v))) return rc;
        acc_g = (u64 *)v;
    }
    std::vector<int64_t> left;
    for (int64_t b0 = 0; b0 < n; b0 += nb_max) {
        const int nb = (int)std::min<int64_t>(nb_max, n - b0);
        HIPCHK(c, hipMemsetAsync(fb, 0, (size_t)nb * 4, c->stream));
        CscOvrParams P;
        memset(&P, 0, sizeof P);
        P.data = d_data; P.indices = d_indices; P.indptr = d_indptr; P.kshift = kshift; P.col0 = cols[b0];
        P.gene_cols = d_cols ? d_cols + b0 : nullptr; P.nb = nb; P.codes = d_codes; P.codes16 = d_codes ? c->d_codes16 : nullptr; P.counts = c->d_counts; P.G = G; P.dt = dtype;
        P.is_log1p = (flags & ILLICO_FLAG_LOG1P) ? 1 : 0; P.n_cells = n_rows; P.key_cap = key_cap; P.lg_buckets = lg;
        P.force_sorted = c->csc_ovr_sorted_form ? 1 : 0; P.fallback = fb;
        P.out_2u = s2u; P.out_tie = stie; P.tie_f64 = 1; P.acc_global = acc_g; P.g_lds = g_lds;
        if (accg) HIPCHK(c, hipMemsetAsync(acc_g, 0, (size_t)nb * G * 8, c->stream));
        {
            ProfScope ps(c, KID_CSC_OVR);
            hipLaunchKernelGGL(kern, dim3(nb), dim3(CSCO_NT), lds, c->stream, P);
            HIPCHK(c, hipGetLastError());
        }
        if ((rc = launch_csc_value_sums<InT, IdxT>(c, d_data, d_indices, d_indptr, kshift, cols[b0], d_cols ? d_cols + b0 : nullptr, d_codes, nb, dtype, flags, ssum))) return rc;
        if ((rc = launch_gene_totals(c, ssum, G, nb, gtot))) return rc;
        if (d_cols) { if ((rc = launch_finalize(c, s2u, stie, ssum, gtot, nb, flags, alternative, o.p, o.u, o.fc, o.ld, -col_lb, d_cols + b0, false, true))) return rc; }
        else if ((rc = launch_finalize(c, s2u, stie, ssum, gtot, nb, flags, alternative, o.p, o.u, o.fc, o.ld, cols[b0] - col_lb, nullptr, false, true))) return rc;
        if (c->pinned_bytes < (size_t)nb * 4) {
            if (c->pinned) hipHostFree(c->pinned);
            c->pinned = nullptr; c->pinned_bytes = 0;
            HIPCHK(c, hipHostMalloc(&c->pinned, (size_t)nb * 4 + 4096, hipHostMallocDefault));
            c->pinned_bytes = (size_t)nb * 4 + 4096;
        }
        HIPCHK(c, hipMemcpyAsync(c->pinned, fb, (size_t)nb * 4, hipMemcpyDeviceToHost, c->stream));
        HIPCHK(c, hipStreamSynchronize(c->stream));
        const u32 *h_fb = (const u32 *)c->pinned;
        for (int64_t j = 0; j < nb; ++j)
            if (h_fb[j]) left.push_back(cols[b0 + j]);
    }
    cols.swap(left);
    return ILLICO_OK;
}


// ---- CSR, count-valued, rows in order: the group-major single pass (kernels_csr_counts.h) ----
// A sample of the stored values and the order of the rows' column indices are looked at ON THE DEVICE (d_verdict: the kernels leave
// every gene flagged when the matrix is not for them); then the row boundaries of the gene windows, the tables of the reference group
// (OVO) / of the whole column (OVR), the histograms of the groups above 255 cells, k_csr_counts and k_csr_big_sweep: p-values straight
// into the planes.  d_flags (device, [W] + 4 words for the verdict): the genes it could not take.  Nothing here waits for the host.
template <typename InT, typename IdxT>
static int launch_csr_counts_route(illico_ctx *c, const InT *d_data, const IdxT *d_indices, const IdxT *d_indptr, int64_t n_rows, int64_t n_cols,
                                   int64_t col_lb, int64_t col_ub, int flags, int alternative, const OutPlanes &o, u32 *d_flags) {
    const int G = (int)c->n_groups;
    const bool ovr = c->ref < 0;
    const int64_t W = col_ub - col_lb, Wpad = (W + 63) & ~63ll;
    const int n_big = c->csr_n_big, n_chunks = c->csr_n_chunks;
    int rc;
    void *v;
    u32 *d_verdict = d_flags + W;
    HIPCHK(c, hipMemsetAsync(d_flags, 0, (size_t)(W + 4) * 4, c->stream));
    hipLaunchKernelGGL((k_sample_noncount_cols<InT, IdxT>), dim3((1 << 16) / 256), dim3(256), 0, c->stream, d_data, d_indptr, 0ll, (long long)n_rows,
                       1 << 16, CSRC_RT, d_verdict);
    hipLaunchKernelGGL((k_csr_density_verdict<IdxT>), dim3(1), dim3(1), 0, c->stream, d_indptr, (long long)n_rows, (long long)n_cols, 0.3, d_verdict);
    // rows out of order: a call over every column finds them in its entry loops (a row's stretches then cover the row: an entry that
    // is not where the boundaries put it shows up outside its window); a column window asks the whole index array first -- unless the
    // matrix is a bound one, whose order was looked at when it was bound
    if (!(col_lb == 0 && col_ub == n_cols) && !c->cur_sorted_known)
        hipLaunchKernelGGL((k_csr_sorted_check<IdxT>), dim3((unsigned)std::min<int64_t>((n_rows + 3) / 4 + 1, 8192)), dim3(256), 0, c->stream, d_indices,
                           d_indptr, (int)n_rows, (int *)(d_verdict + 3));
    HIPCHK(c, hipGetLastError());
    // windows of about 2048 genes: 72 KB of mixed cells, two workgroups per CU; equal widths
    const int n_win = (int)((W + 2047) / 2048);
    const int Wg = (int)((((W + n_win - 1) / n_win) + 63) & ~63ll);
    const int n_bnd = n_win + 1;
    if ((rc = get_scratch(c, "csrc_bounds", (size_t)n_bnd * n_rows * 4, &v))) return rc;
    u32 *bounds = (u32 *)v;
    const size_t slab = (size_t)Wpad * 64;
    if ((rc = get_scratch(c, "csrc_tables", slab * 4 * (size_t)(2 + n_big) + (size_t)Wpad * (16 + 8), &v))) return rc;
    u32 *hist = (u32 *)v, *tab = hist + slab * (size_t)(1 + n_big);
    uint4 *ginfo = (uint4 *)(tab + slab);
    double *gtot = (double *)(ginfo + Wpad);
    HIPCHK(c, hipMemsetAsync(hist, 0, slab * 4 * (size_t)(1 + n_big), c->stream));
    CsrCountsParams P;
    memset(&P, 0, sizeof P);
    P.data = d_data; P.indices = d_indices; P.indptr = d_indptr; P.perm = c->d_perm; P.pos_ptr = c->d_posptr; P.counts = c->d_counts;
    P.G = G; P.ref = (int)c->ref; P.n_cells = n_rows; P.col_lb = col_lb; P.W = (int)W; P.Wg = Wg; P.bounds = bounds; P.bstep = 1; P.n_bnd = n_bnd;
    P.tab = tab; P.ginfo = ginfo; P.gene_total = gtot; P.Wpad = Wpad; P.gene_flags = d_flags; P.verdict = d_verdict; P.unsorted = d_verdict + 3;
    P.use_continuity = (flags & ILLICO_FLAG_CONTINUITY) ? 1 : 0; P.tie_correct = (flags & ILLICO_FLAG_TIE_CORRECT) ? 1 : 0; P.alternative = alternative;
    P.out_p = o.p; P.out_u = o.u; P.out_fc = o.fc; P.out_ld = o.ld; P.hist = hist;
    P.chunk_p0 = c->d_csr_chunks; P.chunk_n = c->d_csr_chunks + n_chunks; P.chunk_slab = c->d_csr_chunks + 2 * n_chunks;
    P.big_groups = c->d_csr_chunks + 3 * n_chunks; P.n_big = n_big; P.abl = c->csr_counts_abl;
    {
        ProfScope ps(c, KID_SPARSE_SEG);
        const long long tot = (long long)n_rows * n_bnd;
        hipLaunchKernelGGL((k_csr_row_bounds<IdxT>), dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, c->stream, d_indices, d_indptr, (int)n_rows,
                           (long long)n_cols, (long long)col_lb, (long long)col_ub, Wg, n_bnd, bounds, (const u32 *)d_verdict);
        HIPCHK(c, hipGetLastError());
    }
    {
        ProfScope ps(c, KID_FUSED_REF);
        auto kern = k_csr_hist<InT, IdxT>;
        const size_t lds = csrh_lds_bytes(Wg);
        HIPCHK(c, hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        if (n_chunks > 0) hipLaunchKernelGGL(kern, dim3(n_win, n_chunks), dim3(CSRH_NT), lds, c->stream, P); // the reference group (OVO), the big groups
        if (!ovr) hipLaunchKernelGGL((k_csr_tables<false>), dim3((unsigned)((W + 255) / 256)), dim3(256), 0, c->stream, (const u32 *)hist, (long long)Wpad, (int)W, (long long)c->h_counts[c->ref], 0, tab, ginfo, gtot, (const u32 *)d_verdict);
        HIPCHK(c, hipGetLastError());
    }
    const size_t lds = csrc_lds_bytes(Wg);
    if (ovr) {
        if ((rc = get_scratch(c, "csrc_dump", (size_t)G * n_win * CSRC_WPG * Wg * 4, &v))) return rc;
        P.dump = (u32 *)v;
        {
            ProfScope ps(c, KID_CSR_COUNTS);
            auto kern = k_csr_counts<InT, IdxT, true>;
            HIPCHK(c, hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
            hipLaunchKernelGGL(kern, dim3(n_win, G), dim3(CSRC_NT), lds, c->stream, P);
            HIPCHK(c, hipGetLastError());
        }
        {
            ProfScope ps(c, KID_FUSED_REF);
            const int slices = std::max(1, std::min(G, (int)std::min<int64_t>(64, (1 << 20) / std::max<int64_t>(W, 1) + 1))); // ~a million threads
            const int gps = (G + slices - 1) / slices;
            hipLaunchKernelGGL(k_csr_colhist, dim3((unsigned)((W + 255) / 256), (unsigned)((G + gps - 1) / gps)), dim3(256), 0, c->stream, P, n_win, gps);
            hipLaunchKernelGGL((k_csr_tables<true>), dim3((unsigned)((W + 255) / 256)), dim3(256), 0, c->stream, (const u32 *)hist, (long long)Wpad, (int)W, (long long)n_rows, n_big, tab, ginfo, gtot, (const u32 *)d_verdict);
            HIPCHK(c, hipGetLastError());
        }
        ProfScope ps(c, KID_OVR_SCAN);
        hipLaunchKernelGGL(k_csr_ovr_sweep, dim3(n_win, G), dim3(CSRC_NT), 0, c->stream, P);
        if (n_big > 0) hipLaunchKernelGGL((k_csr_big_sweep<true>), dim3((unsigned)((W + 255) / 256), n_big), dim3(256), 0, c->stream, P);
        HIPCHK(c, hipGetLastError());
    } else {
        ProfScope ps(c, KID_CSR_COUNTS);
        auto kern = k_csr_counts<InT, IdxT, false>;
        HIPCHK(c, hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        hipLaunchKernelGGL(kern, dim3(n_win, G), dim3(CSRC_NT), lds, c->stream, P);
        if (n_big > 0) hipLaunchKernelGGL((k_csr_big_sweep<false>), dim3((unsigned)((W + 255) / 256), n_big), dim3(256), 0, c->stream, P);
        HIPCHK(c, hipGetLastError());
    }
    return ILLICO_OK;
}

// Host-resident sparse input whose stored values are counts below 255 (what a raw count matrix holds): the values travel as BYTES -- a
// quarter of a float32 array's share of the link; C3's 0.96 GB of values become 0.24 -- narrowed by host threads (on the NUMA node the array
// lives on) into two pinned 32-MB chunks, chunk k + 1 under the upload of chunk k, and widened again on the device (k_bytes_to_values) into
// the buffer the kernels read: the same values, bit for bit.  A value that is no integer in [0, 255) stops the attempt (*done = false:
// the caller uploads the array as it is).  `d_out` receives entries [k0, k1) of `data`.
#define SPB_CHUNK (32ll << 20)
// `meanwhile` runs on the calling thread while the values are narrowed and sent (the caller's upload of the index array: the link then
// carries both, the narrowing costs nothing on the clock); it runs in every case, exactly once; its status is returned first.
template <typename InT, typename Meanwhile>
static int upload_values_as_bytes(illico_ctx *c, const InT *data, int64_t k0, int64_t k1, InT *d_out, bool *done, Meanwhile &&meanwhile) {
    *done = false;
    const int64_t n = k1 - k0;
    bool look = !c->no_sparse_byte_values && n >= (4ll << 20);
    for (int64_t i = 0; i < 4096 && look; ++i) { // a look first: 4096 evenly spaced values
        uint8_t b;
        narrow_cells<InT>(data + k0 + (n - 1) * i / 4095, &b, 1);
        if (b == 255) look = false;
    }
    if (!look) return meanwhile();
    HostStage *hs = host_stage_of(c);
    for (int j = 0; j < 2; ++j) {
        if (!hs->sp_pin[j]) HIPCHK(c, hipHostMalloc(&hs->sp_pin[j], (size_t)SPB_CHUNK, hipHostMallocDefault));
        if (!hs->sp_up[j]) HIPCHK(c, hipEventCreateWithFlags(&hs->sp_up[j], hipEventDisableTiming));
    }
    void *v;
    int rc;
    if ((rc = get_scratch(c, "sp_bytes", (size_t)n, &v))) return rc;
    uint8_t *d_bytes = (uint8_t *)v;
    const int node = c->no_host_numa ? -1 : numa_node_of_buffer(data + k0, (size_t)n * sizeof(InT));
    const int T = (int)std::max<int64_t>(1, std::min<int64_t>(c->host_fill_threads > 0 ? c->host_fill_threads : 16, 64));
    bool bad = false;
    int hip_err = 0;
    std::thread producer([&] { // (a thread of its own: its CPU mask -- the array's NUMA node -- is inherited by the narrowing threads and ends with it)
        hipSetDevice(c->device);
        numa_confine_this_thread(node);
        int64_t chunk_no = 0;
        for (int64_t o = 0; o < n && !bad && !hip_err; o += SPB_CHUNK, ++chunk_no) {
            const int j = (int)(chunk_no & 1);
            const int64_t m = std::min<int64_t>(SPB_CHUNK, n - o);
            if (chunk_no >= 2 && hipEventSynchronize(hs->sp_up[j]) != hipSuccess) { hip_err = 1; break; }
            uint8_t *dst = (uint8_t *)hs->sp_pin[j];
            std::vector<int> flags((size_t)T, 0);
            auto piece = [&](int t) {
                const int64_t a = m * t / T, b = m * (t + 1) / T;
                narrow_cells<InT>(data + k0 + o + a, dst + a, b - a);
                int f = 0;
                for (int64_t i = a; i < b; ++i) f |= dst[i] == 255 ? 1 : 0;
                flags[(size_t)t] = f;
            };
            std::vector<std::thread> pool;
            for (int t = 1; t < T; ++t) pool.emplace_back(piece, t);
            piece(0);
            for (auto &th : pool) th.join();
            for (int t = 0; t < T; ++t) bad = bad || flags[(size_t)t] != 0;
            if (bad) break;
            if (hipMemcpyAsync(d_bytes + o, dst, (size_t)m, hipMemcpyHostToDevice, c->stream) != hipSuccess ||
                hipEventRecord(hs->sp_up[j], c->stream) != hipSuccess) hip_err = 1;
        }
    });
    const int rc_meanwhile = meanwhile();
    producer.join();
    if (rc_meanwhile) return rc_meanwhile;
    if (hip_err) return fail(c, ILLICO_ERR_HIP, "upload of a sparse matrix's values as bytes failed");
    if (bad) { HIPCHK(c, hipStreamSynchronize(c->stream)); return ILLICO_OK; } // (the pinned chunks are free again; the caller uploads the values as they are)
    hipLaunchKernelGGL((k_bytes_to_values<InT>), dim3(4096), dim3(256), 0, c->stream, (const uint8_t *)d_bytes, (long long)n, d_out);
    HIPCHK(c, hipGetLastError());
    c->h2d_input_bytes += n;
    *done = true;
    return ILLICO_OK;
}

// Sparse OVO with groups whose (gene, group) runs outgrow what k_csc_gene / k_ovo_rank take quickly (clusters of hundreds or
// thousands of cells): regroup, then the packed rank kernel of the dense route (kernels_ovo_compact.h) on the regrouped runs
// (small_groups: groups of at most 256 cells as well -- k_csc_gene takes those in one kernel when a gene's entries fit its LDS key buffer;
//  genes that do not -- eight-byte keys: C3's 30 000 entries per gene -- are ranked by the packed kernel too, not by k_ovo_rank)
static bool sparse_packed_rank_fits(const illico_ctx *c, bool small_groups = false) {
    if (c->ref < 0 || c->no_packed_dense || (c->max_nonref <= 256 && !small_groups) || c->max_nonref > 65535) return false;
    const int64_t n_ref = c->h_counts[c->ref];
    return n_ref >= 1 && n_ref <= 65535;
}

// Average stored entries per column above which a sparse window is written out dense (the dense routes then rank it): what the per-gene
// LDS kernels hold -- 32 768 four-byte keys, half as many eight-byte ones.  OVO with eight-byte keys keeps the four-byte bound: its columns
// are regrouped in HBM and ranked by the packed kernel, whatever their length (C3 shape as CSR in float64: 14.3 ms through the dense
// window -- 19 GB of it --, 24 through k_ovo_rank).
template <typename KeyT> static double long_column(const illico_ctx *c) {
    if (sizeof(KeyT) == 8 && c->ref >= 0 && !c->no_sparse_packed_small && sparse_packed_rank_fits(c, true)) return 32768.0;
    return 32768.0 * 4.0 / (double)sizeof(KeyT);
}

// sizes the group-major CSR pass holds (kernels_csr_counts.h)
static bool csr_counts_route_fits(const illico_ctx *c, int flags, int64_t n_rows) {
    if (c->no_csr_counts_path || c->hold_csr_counts || (flags & ILLICO_FLAG_LOG1P) || c->tap || c->no_counts_path || c->big_n) return false;
    if (c->csr_n_big < 0 || n_rows >= (1ll << 30) || c->n_groups > 65535) return false;
    if (c->ref >= 0 && (c->h_counts[c->ref] < 1 || c->h_counts[c->ref] >= 30000)) return false;
    return true;
}
// what a call learns from the verdict words of the pass: true = the matrix was not for the route at all
static bool csr_counts_verdict_bad(const u32 *vd) {
    return (double)vd[0] > 0.02 * (double)vd[2] || (double)vd[1] > 0.005 * (double)vd[2] || vd[3] != 0u;
}

template <typename InT, typename IdxT, typename KeyT>
int run_sparse_t(illico_ctx *c, bool is_csr, const void *data, const void *indices, const void *indptr, int dtype,
                 int64_t n_rows, int64_t n_cols, int64_t col_lb, int64_t col_ub, int flags, int alternative,
                 const OutPlanes &o, bool allow_dense_window, bool allow_transpose, bool indices_are_codes, bool allow_csr_counts) {
    // indices_are_codes: CSC whose `indices` hold the group code of each stored entry's cell (what the device CSR -> CSC
    // transposition writes: the per-entry lookup codes[row] is an uncoalesced gather the CSC kernels then skip)
    const int *d_codes = indices_are_codes ? nullptr : c->d_codes;
    const int G = (int)c->n_groups;
    const bool ovr = c->ref < 0;
    const bool in_dev = flags & ILLICO_FLAG_INPUT_DEVICE;
    const int64_t W = col_ub - col_lb;
    const int64_t n_ptr = (is_csr ? n_rows : n_cols) + 1;
    // More groups than the regrouping kernels' LDS histogram holds (~40 000): what the count-valued routes below do not take is written out as
    // a dense window in the matrix's own type and takes the dense routes, which know no such limit (the reference has none either:
    // ovr/sparse_ovr.py:23-97, utils/groups.py:18-58).
    const bool many_groups = seg_lds_bytes(G) > kMaxLds;
    int rc;
    void *v;

    // CSC, count-valued, groups of at most 255 cells: per-group histograms in LDS (OVO and OVR) -- when a sample of the window's
    // stored values says they are counts at all
    int n_big_groups = 0;
    int64_t max_ranked = 0;
    for (int g = 0; g < G; ++g)
        if (g != c->ref) { n_big_groups += c->h_counts[g] > 255 ? 1 : 0; max_ranked = std::max<int64_t>(max_ranked, c->h_counts[g]); }
    // (more than CSCC_MAX_BIG groups above 255 cells: 16-bit cells for every group, while those fit LDS)
    // ... and more groups than LDS holds tables for: windows of groups, one launch each over the same entries -- up to 33 of them (65 535 groups:
    // the 16-bit code table's limit); 30 000 groups of ten cells at C3 shape: 52.7 ms through the per-gene sort routes when eight was the limit)
    const bool w16_needed = n_big_groups > CSCC_MAX_BIG || (!ovr && c->h_counts[c->ref] >= 30000); // (64-bit sweep terms for a large reference)
    const int n_windows = (G + cscc_group_window(G, w16_needed) - 1) / std::max(1, cscc_group_window(G, w16_needed));
    const bool cells_fit = (!w16_needed || (max_ranked <= 65535 && !c->no_csc_counts_wide)) && n_windows <= (c->csc_counts_max_windows > 0 ? c->csc_counts_max_windows : 33) &&
                           (n_windows == 1 || (!c->no_csc_counts_windows && c->d_codes16 && !indices_are_codes));
    // (big_n -- OVR over more than 2^21 - 1 cells --: the table kernels' t^3 terms could wrap; the sort-based routes hold)
    const bool counts_route = !is_csr && !c->big_n && !c->no_csc_counts_path && !(flags & ILLICO_FLAG_LOG1P) && cells_fit && n_rows < (1ll << 30) &&
                              (uint64_t)n_rows * std::max(sizeof(InT), sizeof(IdxT)) < (1ull << 32) && // (k_csc_counts forms the byte offsets of a column's entries in 32 bits)
                              (ovr || c->h_counts[c->ref] < (1ll << 30));
    // CSR, count-valued, not too sparse: dense windows + the fused single-pass kernels (below); the same question about the values
    const bool window_route = is_csr && !c->big_n && allow_dense_window && !c->no_dense_window_path && fused_path_allowed(c, flags) &&
                              (size_t)n_rows * 4 * 64 <= (size_t)c->scratch_bytes;
    // CSR, count-valued, small groups: the group-major single pass (kernels_csr_counts.h); the same question about the values
    const bool csr_counts = is_csr && allow_csr_counts && W > 0 && csr_counts_route_fits(c, flags, n_rows);
    u32 h_sample[4] = {0, 0, 0, 0}; // non-integers, integers beyond the table, samples taken
    bool sampled = false;
    if ((flags & ILLICO_FLAG_DEFER) && in_dev && (flags & ILLICO_FLAG_OUTPUT_DEVICE) && !o.staged && counts_route && W > 0 &&
        !indices_are_codes && c->d_codes16 && !c->tap)
        return run_csc_counts_deferred<InT, IdxT>(c, data, indices, indptr, dtype, (int)(sizeof(IdxT) == 4 ? ILLICO_IDX_I32 : ILLICO_IDX_I64),
                                                  n_rows, n_cols, col_lb, col_ub, flags, alternative, o);

    // the group-major CSR pass, deferred: device arrays, device planes -- enqueued as a whole, its flags + verdict travel to pinned memory
    // behind an event (resolve_pending_csc)
    if (csr_counts && (flags & ILLICO_FLAG_DEFER) && in_dev && (flags & ILLICO_FLAG_OUTPUT_DEVICE) && !o.staged) {
        if ((rc = get_scratch(c, "csrc_flags", (size_t)(W + 4) * 4, &v))) return rc;
        u32 *d_flags = (u32 *)v;
        if ((rc = launch_csr_counts_route<InT, IdxT>(c, (const InT *)data, (const IdxT *)indices, (const IdxT *)indptr, n_rows, n_cols, col_lb, col_ub, flags,
                                                     alternative, o, d_flags))) return rc;
        const int slot = c->pend_next;
        void *&pin = c->pend_pinned[slot];
        if (c->pend_pinned_bytes[slot] < (size_t)(W + 4) * 4) {
            if (pin) hipHostFree(pin);
            pin = nullptr;
            c->pend_pinned_bytes[slot] = 0;
            HIPCHK(c, hipHostMalloc(&pin, (size_t)(W + 4) * 4 + 4096, hipHostMallocDefault));
            c->pend_pinned_bytes[slot] = (size_t)(W + 4) * 4 + 4096;
        }
        if (!c->pend_event[slot]) HIPCHK(c, hipEventCreateWithFlags(&c->pend_event[slot], hipEventDisableTiming));
        HIPCHK(c, hipMemcpyAsync(pin, d_flags, (size_t)(W + 4) * 4, hipMemcpyDeviceToHost, c->stream));
        HIPCHK(c, hipEventRecord(c->pend_event[slot], c->stream));
        c->pend_next ^= 1;
        PendingDense &q = c->pend;
        q = PendingDense();
        q.on = true; q.kind = 1; q.is_csr = true; q.sp_data = data; q.sp_indices = indices; q.sp_indptr = indptr;
        q.sorted_known = c->cur_sorted_known;
        q.idx_dtype = (int)(sizeof(IdxT) == 4 ? ILLICO_IDX_I32 : ILLICO_IDX_I64); q.n_cols = n_cols;
        q.dtype = dtype; q.flags = flags & ~ILLICO_FLAG_DEFER; q.alternative = alternative; q.slot = slot; q.N = n_rows;
        q.col_lb = col_lb; q.col_ub = col_ub; q.out_ld = o.ld; q.p = o.p; q.u = o.u; q.fc = o.fc;
        return ILLICO_OK;
    }

    // what the host needs of indptr: all of it for CSC (batch planning), its two ends for CSR (total stored entries)
    std::vector<IdxT> h_indptr;
    IdxT ends[2] = {0, 0};
    if (in_dev) { // on the context's stream: ordered after whatever produced indptr on it (a blocking hipMemcpy runs on the
        // null stream, which non-blocking streams -- torch's side streams -- do not synchronise with)
        if ((counts_route && W > 0) || window_route) { // the sample of the stored values rides along: one wait for both
            if ((rc = get_scratch(c, "flag", 16, &v))) return rc;
            u32 *d_cnt = (u32 *)v;
            HIPCHK(c, hipMemsetAsync(d_cnt, 0, 16, c->stream));
            hipLaunchKernelGGL((k_sample_noncount_cols<InT, IdxT>), dim3((1 << 16) / 256), dim3(256), 0, c->stream, (const InT *)data,
                               (const IdxT *)indptr, (long long)(is_csr ? 0 : col_lb), (long long)(is_csr ? n_rows : col_ub), 1 << 16,
                               is_csr ? FUSED_RT : CSCC_RT, d_cnt);
            HIPCHK(c, hipGetLastError());
            HIPCHK(c, hipMemcpyAsync(h_sample, d_cnt, 16, hipMemcpyDeviceToHost, c->stream));
            sampled = true;
        }
        if (is_csr) {
            HIPCHK(c, hipMemcpyAsync(&ends[0], indptr, sizeof(IdxT), hipMemcpyDeviceToHost, c->stream));
            HIPCHK(c, hipMemcpyAsync(&ends[1], (const IdxT *)indptr + (n_ptr - 1), sizeof(IdxT), hipMemcpyDeviceToHost, c->stream));
        } else {
            h_indptr.resize(n_ptr);
            HIPCHK(c, hipMemcpyAsync(h_indptr.data(), indptr, n_ptr * sizeof(IdxT), hipMemcpyDeviceToHost, c->stream));
        }
        HIPCHK(c, hipStreamSynchronize(c->stream));
        if (!is_csr) { ends[0] = h_indptr[0]; ends[1] = h_indptr[n_ptr - 1]; }
    } else {
        if (!is_csr) h_indptr.assign((const IdxT *)indptr, (const IdxT *)indptr + n_ptr);
        ends[0] = ((const IdxT *)indptr)[0];
        ends[1] = ((const IdxT *)indptr)[n_ptr - 1];
    }
    const int64_t total_nnz = (int64_t)ends[1];
    if (ends[0] != 0) return fail(c, ILLICO_ERR_ARG, "indptr[0] must be 0");

    // device views of the matrix: stored entry k of the caller's arrays is d_data[k - kshift] / d_indices[k - kshift]
    const InT *d_data = (const InT *)data;
    const IdxT *d_indices = (const IdxT *)indices, *d_indptr = (const IdxT *)indptr;
    int64_t kshift = 0;
    if (!in_dev) {
        if ((rc = get_scratch(c, "sp_indptr", n_ptr * sizeof(IdxT), &v))) return rc;
        HIPCHK(c, hipMemcpyAsync(v, indptr, n_ptr * sizeof(IdxT), hipMemcpyHostToDevice, c->stream));
        d_indptr = (const IdxT *)v;
        // CSR rows span every column: the whole matrix goes up; CSC: only the stored entries of the requested window
        const int64_t k0 = is_csr ? 0 : (int64_t)h_indptr[col_lb];
        const int64_t k1 = is_csr ? total_nnz : (int64_t)h_indptr[col_ub];
        const size_t cnt = (size_t)std::max<int64_t>(k1 - k0, 1);
        if ((rc = get_scratch(c, "sp_data", cnt * sizeof(InT), &v))) return rc;
        void *v_idx;
        if ((rc = get_scratch(c, "sp_indices", cnt * sizeof(IdxT), &v_idx))) return rc;
        // the index array goes up on the copy stream while host threads narrow the values (count values below 255 travel as bytes: a quarter
        // of their bytes over the link); the context's stream waits for it
        HostStage *hs = host_stage_of(c);
        if (!hs->copy) HIPCHK(c, hipStreamCreateWithFlags(&hs->copy, hipStreamNonBlocking));
        if (!hs->up[0]) HIPCHK(c, hipEventCreateWithFlags(&hs->up[0], hipEventDisableTiming));
        auto upload_indices = [&]() -> int {
            HIPCHK(c, hipMemcpyAsync(v_idx, (const IdxT *)indices + k0, (size_t)(k1 - k0) * sizeof(IdxT), hipMemcpyHostToDevice, hs->copy));
            HIPCHK(c, hipEventRecord(hs->up[0], hs->copy));
            HIPCHK(c, hipStreamWaitEvent(c->stream, hs->up[0], 0));
            return ILLICO_OK;
        };
        bool as_bytes = false;
        if ((rc = upload_values_as_bytes<InT>(c, (const InT *)data, k0, k1, (InT *)v, &as_bytes, upload_indices))) return rc;
        if (!as_bytes) {
            HIPCHK(c, hipMemcpyAsync(v, (const InT *)data + k0, (size_t)(k1 - k0) * sizeof(InT), hipMemcpyHostToDevice, c->stream));
            c->h2d_input_bytes += (int64_t)((size_t)(k1 - k0) * sizeof(InT));
        }
        d_data = (const InT *)v;
        kshift = k0;
        d_indices = (const IdxT *)v_idx;
        c->h2d_input_bytes += (int64_t)(n_ptr * sizeof(IdxT) + (size_t)(k1 - k0) * sizeof(IdxT));
    }

    auto take_sample = [&]() -> int { // 64k evenly spaced stored values (device-resident arrays: taken with the indptr copy above)
        if (sampled) return ILLICO_OK;
        const int n_samples = (int)std::min<int64_t>(total_nnz, 1 << 16);
        void *vv;
        int rc2;
        if ((rc2 = get_scratch(c, "flag", 16, &vv))) return rc2;
        u32 *d_cnt = (u32 *)vv;
        HIPCHK(c, hipMemsetAsync(d_cnt, 0, 8, c->stream));
        hipLaunchKernelGGL((k_sample_noncount<InT>), dim3((n_samples + 255) / 256), dim3(256), 0, c->stream, d_data, (long long)total_nnz,
                           n_samples, FUSED_RT, d_cnt);
        HIPCHK(c, hipGetLastError());
        HIPCHK(c, hipMemcpyAsync(h_sample, d_cnt, 8, hipMemcpyDeviceToHost, c->stream));
        HIPCHK(c, hipStreamSynchronize(c->stream));
        h_sample[2] = (u32)n_samples;
        sampled = true;
        return ILLICO_OK;
    };
    // ---- CSR, count-valued, rows in order: the group-major single pass (kernels_csr_counts.h); ONE wait, for its flags + verdict ----
    if (csr_counts && total_nnz > 0) {
        if ((rc = get_scratch(c, "csrc_flags", (size_t)(W + 4) * 4, &v))) return rc;
        u32 *d_flags = (u32 *)v;
        if ((rc = launch_csr_counts_route<InT, IdxT>(c, d_data, d_indices, d_indptr, n_rows, n_cols, col_lb, col_ub, flags, alternative, o, d_flags))) return rc;
        std::vector<u32> hf((size_t)W + 4);
        HIPCHK(c, hipMemcpyAsync(hf.data(), d_flags, (size_t)(W + 4) * 4, hipMemcpyDeviceToHost, c->stream));
        HIPCHK(c, hipStreamSynchronize(c->stream));
        int64_t n_flagged = 0;
        for (int64_t j = 0; j < W; ++j) n_flagged += hf[j] ? 1 : 0;
        if (!csr_counts_verdict_bad(hf.data() + W) && n_flagged * 16 > W) // many genes left the pass: all of the window by the other routes, once
            return run_sparse_t<InT, IdxT, KeyT>(c, is_csr, data, indices, indptr, dtype, n_rows, n_cols, col_lb, col_ub, flags, alternative, o, true, true,
                                                 false, false);
        if (!csr_counts_verdict_bad(hf.data() + W)) {
            // runs of flagged genes (closer than 32 genes: one run; the genes in between are recomputed, identically)
            for (int64_t j = 0; j < W;) {
                if (!hf[j]) { ++j; continue; }
                int64_t last = j;
                for (int64_t e = j + 1; e < W && e - last <= 32; ++e) if (hf[e]) last = e;
                OutPlanes o2 = o;
                o2.p += j; o2.u += j; o2.fc += j;
                if ((rc = run_sparse_t<InT, IdxT, KeyT>(c, is_csr, data, indices, indptr, dtype, n_rows, n_cols, col_lb + j, col_lb + last + 1, flags,
                                                        alternative, o2, false, true, false, false))) return rc;
                j = last + 1;
            }
            return ILLICO_OK;
        }
        h_sample[0] = hf[W]; h_sample[1] = hf[W + 1]; h_sample[2] = hf[W + 2]; // (the same sample the dense-window route asks for)
        sampled = true;
    }

    // ---- CSR, count-valued, not too sparse: dense float32 windows + the fused single-pass kernels (k_csr_densify) ----
    const double density = (double)total_nnz / ((double)std::max<int64_t>(n_rows, 1) * (double)std::max<int64_t>(n_cols, 1));
    bool dense_window = window_route && density >= 0.015 && total_nnz > 0;
    if (dense_window) { // worth it only for count-valued data: look at 64k evenly spaced stored values first
        if ((rc = take_sample())) return rc;
        // the genes this route cannot take are redone over the column window that covers them, so it needs nearly all of
        // them to fit: no non-integers, few values beyond the table
        dense_window = (double)h_sample[0] <= 0.02 * (double)h_sample[2] && (double)h_sample[1] <= 0.005 * (double)h_sample[2];
    }
    if (dense_window) {
        // byte cells (the fused kernels only take integers below 64): a quarter of the window's traffic both ways;
        // float32 cells behind "dense_window_f32"
        const bool bytes = !c->dense_window_f32;
        const size_t cell = bytes ? 1 : 4;
        int64_t wmax = (int64_t)((size_t)c->scratch_bytes / ((size_t)n_rows * cell)) & ~63ll;
        wmax = std::min<int64_t>(wmax, (1ll << 29));
        if (c->gene_batch > 0) wmax = std::min<int64_t>(wmax, (c->gene_batch + 63) & ~63ll);
        int64_t bad_lo = -1, bad_hi = -1;
        std::vector<u32> hf;
        for (int64_t w0 = col_lb; w0 < col_ub; w0 += wmax) {
            const int64_t wn = std::min<int64_t>(wmax, col_ub - w0), ldD = (wn + 63) & ~63ll;
            if ((rc = get_scratch(c, "dense_window", (size_t)n_rows * ldD * cell, &v))) return rc;
            {
                ProfScope ps(c, KID_SPARSE_SEG);
                const dim3 grid((unsigned)std::min<int64_t>(n_rows, 1 << 16));
                if (bytes)
                    hipLaunchKernelGGL((k_csr_densify<InT, IdxT, uint8_t>), grid, dim3(DENS_NT), 0, c->stream, d_data, d_indices, d_indptr,
                                       (int)n_rows, (long long)w0, (int)wn, (uint8_t *)v, (long long)ldD);
                else
                    hipLaunchKernelGGL((k_csr_densify<InT, IdxT, float>), grid, dim3(DENS_NT), 0, c->stream, d_data, d_indices, d_indptr,
                                       (int)n_rows, (long long)w0, (int)wn, (float *)v, (long long)ldD);
                HIPCHK(c, hipGetLastError());
            }
            c->fused_tie_sparse = true; // (the window holds CSR input: the reference ranks it by its sparse path)
            if (bytes) rc = run_fused_ovo<uint8_t>(c, v, ldD, 0, (int)wn, flags, alternative, o, w0 - col_lb, hf);
            else rc = run_fused_ovo<float>(c, v, ldD, 0, (int)wn, flags, alternative, o, w0 - col_lb, hf);
            c->fused_tie_sparse = false;
            if (rc) return rc;
            for (int64_t j = 0; j < wn; ++j)
                if (hf[j] == 1u || hf[j] == 3u) { // (2 = taken by the fused route's second, wider pass)
                    if (bad_lo < 0) bad_lo = w0 + j;
                    bad_hi = w0 + j;
                }
        }
        if (bad_lo < 0) return ILLICO_OK;
        // genes the fused kernels could not take (values outside the small-integer table): the exact sparse route
        // over the column window that covers them (it recomputes, identically, the good genes in between)
        OutPlanes o2 = o;
        o2.p += bad_lo - col_lb; o2.u += bad_lo - col_lb; o2.fc += bad_lo - col_lb;
        return run_sparse_t<InT, IdxT, KeyT>(c, is_csr, data, indices, indptr, dtype, n_rows, n_cols, bad_lo, bad_hi + 1, flags,
                                             alternative, o2, false, true, false, false);
    }

    // ---- float64 values that are float32 values throughout (device-resident input, nothing count-valued took it above): the float32
    // kernels give the same bits and hold twice the keys per gene in LDS (C3 shape as CSR, continuous: 14 - 17 ms in float64, 6 in float32) ----
    if constexpr (std::is_same<InT, double>::value) {
        if (is_csr && in_dev && allow_dense_window && allow_transpose && allow_csr_counts && !indices_are_codes && !c->no_f64_narrowing && !c->tap && total_nnz > 0 &&
            !(flags & ILLICO_FLAG_LOG1P)) { // (is_log1p: the float32 kernels form expm1 in float32, the float64 ones and the reference -- utils/sparse/csr.py:282 -- in float64)
            const long long k0 = 0, k1 = (long long)total_nnz;
            if ((rc = get_scratch(c, "flag", 16, &v))) return rc;
            u32 *d_inexact = (u32 *)v;
            HIPCHK(c, hipMemsetAsync(d_inexact, 0, 4, c->stream));
            hipLaunchKernelGGL(k_f64_is_f32, dim3(4096), dim3(256), 0, c->stream, (const double *)d_data + k0, k1 - k0, d_inexact);
            HIPCHK(c, hipGetLastError());
            u32 inexact = 1;
            HIPCHK(c, hipMemcpyAsync(&inexact, d_inexact, 4, hipMemcpyDeviceToHost, c->stream));
            HIPCHK(c, hipStreamSynchronize(c->stream));
            if (!inexact && k1 > k0) {
                // (the copy covers the whole array up to k1, so that entry k of the caller's arrays stays entry k)
                if ((rc = get_scratch(c, "sp_f32", (size_t)k1 * sizeof(float), &v))) return rc;
                hipLaunchKernelGGL(k_f64_to_f32, dim3(4096), dim3(256), 0, c->stream, (const double *)d_data + k0, k1 - k0, (float *)v + k0);
                HIPCHK(c, hipGetLastError());
                return run_sparse_t<float, IdxT, u32>(c, is_csr, v, indices, indptr, ILLICO_F32, n_rows, n_cols, col_lb, col_ub, flags & ~ILLICO_FLAG_DEFER, alternative, o,
                                                      allow_dense_window, allow_transpose, indices_are_codes, false);
            }
        }
    }

    // ---- CSR, any values, columns longer than the per-gene LDS kernels hold (a "sparse" matrix a fifth or more of whose cells are
    // stored): a dense window in the matrix's own type + the dense routes.  The per-gene kernels behind the transposition keep a
    // gene's keys in LDS (~36 000 four-byte keys, half as many eight-byte ones: the bound below); longer columns fall to the general sort routes one by one -- C3 shape with 30 % of the cells
    // stored and continuous values: 76 ms (OVR) / 37 ms (OVO) that way, against 12.6 ms for the same values handed over dense.
    // OVR: the reference accumulates a sparse column's tie sum in float64 (sparse_ovr.py:49,83), the dense routes in exact integers;
    // what separates them is the rounding of n0^3 (n0 zeros), 1.1e-16 of it, which reaches p as z^2 (1 - d)^3 / (6 d) x 1.1e-16 at a
    // fraction d of cells stored: 6e-13 at |z| = 37 (p ~ 1e-300) for d = 0.04, the bound used here; below that the window stays with
    // the sparse routes (kernels_finalize.h: tie_f64_sparse).
    if (is_csr && allow_dense_window && allow_transpose && !c->no_csr_densify_any && !c->tap && !c->big_n && n_rows < (1ll << 31) &&
        (many_groups || (density * (double)n_rows > long_column<KeyT>(c) && (c->ref >= 0 || density >= 0.04))) && (size_t)n_rows * 64 * sizeof(InT) <= (size_t)c->scratch_bytes) {
        int64_t wmax = (int64_t)((size_t)c->scratch_bytes / ((size_t)n_rows * sizeof(InT))) & ~63ll;
        wmax = std::min<int64_t>(wmax, (1ll << 29));
        if (c->gene_batch > 0) wmax = std::min<int64_t>(wmax, (c->gene_batch + 63) & ~63ll);
        for (int64_t w0 = col_lb; w0 < col_ub; w0 += wmax) {
            const int64_t wn = std::min<int64_t>(wmax, col_ub - w0), ldD = (wn + 63) & ~63ll;
            if ((rc = get_scratch(c, "dense_window", (size_t)n_rows * ldD * sizeof(InT), &v))) return rc;
            {
                ProfScope ps(c, KID_DENSIFY);
                const dim3 grid((unsigned)std::min<int64_t>(n_rows, 1 << 16));
                hipLaunchKernelGGL((k_csr_densify<InT, IdxT, InT>), grid, dim3(DENS_NT), 0, c->stream, d_data, d_indices, d_indptr,
                                   (int)n_rows, (long long)w0, (int)wn, (InT *)v, (long long)ldD);
                HIPCHK(c, hipGetLastError());
            }
            OutPlanes o2 = o;
            o2.p += w0 - col_lb; o2.u += w0 - col_lb; o2.fc += w0 - col_lb;
            if ((rc = run_dense_t<InT, KeyT>(c, v, dtype, n_rows, ldD, 0, wn, (flags | ILLICO_FLAG_INPUT_DEVICE) & ~ILLICO_FLAG_DEFER, alternative, o2))) return rc;
        }
        return ILLICO_OK;
    }

    if (many_groups && is_csr) return fail(c, ILLICO_ERR_UNSUPPORTED, "CSR input with %d groups: beyond the regrouping kernels' LDS histogram, and the dense window was not available here", G);
    // ---- CSR, any values: transpose the column window into CSC on the device, then the CSC routes ----
    if (is_csr && allow_transpose && !c->no_csr_transpose_path && n_rows < (1ll << 31)) {
        // sorted column indices (the reference's contract) allow the gather form of pass 2
        int sorted = (c->cur_sorted_known && !c->no_csr_tile_gather) ? 1 : 0; // (a bound matrix: looked at when it was bound)
        if (!sorted && !c->no_csr_tile_gather) {
            if ((rc = get_scratch(c, "flag", 16, &v))) return rc;
            int *d_bad = (int *)v;
            HIPCHK(c, hipMemsetAsync(d_bad, 0, 8, c->stream));
            hipLaunchKernelGGL((k_csr_sorted_check<IdxT>), dim3((unsigned)std::min<int64_t>((n_rows + 3) / 4 + 1, 8192)), dim3(256), 0, c->stream,
                               d_indices, d_indptr, (int)n_rows, d_bad);
            HIPCHK(c, hipGetLastError());
            int bad = 0;
            HIPCHK(c, hipMemcpyAsync(&bad, d_bad, 4, hipMemcpyDeviceToHost, c->stream));
            HIPCHK(c, hipStreamSynchronize(c->stream));
            sorted = bad ? 0 : 1;
        }
        const int cap = 8192; // LDS staging entries of k_csr_tile_gather
        int RB = 256;
        if (sorted) { // expected entries per (row block, 64-column tile) <= cap / 2
            const double per_row = std::max(density * TRG_COLS, 1e-9);
            RB = TRG_NT * TRG_RPT;
            while (RB > 64 && RB * per_row > cap / 2) RB >>= 1;
        }
        const int n_blocks = (int)((n_rows + RB - 1) / RB);
        // few row blocks (20 000 cells: 40): the counting pass takes a block's entries in slices, the gather form a window's tiles in stretches
        const int ny = (n_blocks >= 2048 || c->no_csr_transpose_split) ? 1 : std::min(16, (2048 + n_blocks - 1) / n_blocks);
        // the per-block column tables live in LDS: 16-bit counters in the counting pass (RB <= 512 entries per (block, column)), 32-bit
        // cursors in the scatter form of pass 2 (unsorted rows, or a (block, tile) piece beyond the gather form's staging)
        int64_t wmax = std::min<int64_t>(W, (sorted && RB <= 512) ? 65536 : 32768);
        for (int64_t w0 = col_lb; w0 < col_ub;) {
            const int64_t wn = std::min<int64_t>(wmax, col_ub - w0);
            if ((rc = get_scratch(c, "tr_counts", (size_t)n_blocks * wn * 4, &v))) return rc;
            u32 *counts = (u32 *)v;
            if ((rc = get_scratch(c, "tr_cols", (size_t)(wn + 1) * 8 + 16, &v))) return rc;
            u32 *col_total = (u32 *)v, *col_ptr = col_total + (wn + 1), *d_over = col_ptr + (wn + 1);
            u32 total = 0;
            {
                ProfScope ps(c, KID_SPARSE_SEG);
                HIPCHK(c, hipMemsetAsync(col_total + wn, 0, 4, c->stream));
                HIPCHK(c, hipMemsetAsync(d_over, 0, 4, c->stream));
                HIPCHK(c, hipFuncSetAttribute((const void *)k_csr_block_count<IdxT>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)(((wn + 1) / 2) * 4)));
                if (ny > 1) HIPCHK(c, hipMemsetAsync(counts, 0, (size_t)n_blocks * wn * 4, c->stream));
                hipLaunchKernelGGL((k_csr_block_count<IdxT>), dim3(n_blocks, ny), dim3(TRC_NT), (size_t)((wn + 1) / 2) * 4, c->stream, d_indices, d_indptr,
                                   (int)n_rows, RB, (long long)w0, (int)wn, counts);
                hipLaunchKernelGGL(k_col_block_scan, dim3((unsigned)((wn + 255) / 256)), dim3(256), 0, c->stream, counts, n_blocks, (int)wn, col_total);
                hipLaunchKernelGGL(k_gene_base_scan, dim3(1), dim3(1024), 0, c->stream, (const u32 *)col_total, (int)wn + 1, col_ptr);
                HIPCHK(c, hipGetLastError());
            }
            HIPCHK(c, hipMemcpyAsync(&total, col_ptr + wn, 4, hipMemcpyDeviceToHost, c->stream));
            HIPCHK(c, hipStreamSynchronize(c->stream));
            // (a window whose stored entries reach 2^31 would wrap the 32-bit scan: such windows are halved before they
            //  get here -- 2^31 entries do not fit the scratch cap at 8+ bytes each)
            const size_t need = (size_t)std::max<u32>(total, 1) * (sizeof(InT) + 4);
            if ((need > (size_t)c->scratch_bytes || (double)total_nnz * (double)wn / (double)std::max<int64_t>(n_cols, 1) > 1.5e9) && wn > 64) {
                wmax = std::max<int64_t>(64, wn / 2);
                continue;
            }
            if ((rc = get_scratch(c, "tr_data", (size_t)std::max<u32>(total, 1) * sizeof(InT), &v))) return rc;
            InT *t_data = (InT *)v;
            if ((rc = get_scratch(c, "tr_rows", (size_t)std::max<u32>(total, 1) * 4, &v))) return rc;
            int *t_rows = (int *)v;
            bool done = false;
            if (sorted) {
                ProfScope ps(c, KID_SPARSE_SEG);
                const size_t lds = (size_t)cap * (sizeof(InT) + 4 + 1);
                HIPCHK(c, hipFuncSetAttribute((const void *)k_csr_tile_gather<InT, IdxT>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
                hipLaunchKernelGGL((k_csr_tile_gather<InT, IdxT>), dim3(n_blocks, ny), dim3(TRG_NT), lds, c->stream,
                                   d_data, d_indices, d_indptr, (int)n_rows, RB, (long long)w0, (int)wn, (const u32 *)counts, (const u32 *)col_total,
                                   (const u32 *)col_ptr, cap, (const int *)c->d_codes, t_data, t_rows, d_over);
                HIPCHK(c, hipGetLastError());
                u32 over = 0;
                HIPCHK(c, hipMemcpyAsync(&over, d_over, 4, hipMemcpyDeviceToHost, c->stream));
                HIPCHK(c, hipStreamSynchronize(c->stream));
                done = over == 0; // a (block, tile) piece larger than the LDS staging: redo the window with the scatter form
            }
            if (!done && wn > 32768) { // (the scatter form keeps 32-bit cursors per column in LDS: narrower windows)
                wmax = 32768;
                continue;
            }
            if (!done) {
                ProfScope ps(c, KID_SPARSE_SEG);
                HIPCHK(c, hipFuncSetAttribute((const void *)k_csr_block_scatter<InT, IdxT>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)(wn * 4)));
                hipLaunchKernelGGL((k_csr_block_scatter<InT, IdxT>), dim3(n_blocks), dim3(TR_NT), (size_t)wn * 4, c->stream, d_data, d_indices,
                                   d_indptr, (int)n_rows, RB, (long long)w0, (int)wn, (const u32 *)counts, (const u32 *)col_ptr, (const int *)c->d_codes, t_data, t_rows);
                HIPCHK(c, hipGetLastError());
            }
            OutPlanes o2 = o;
            o2.p += w0 - col_lb; o2.u += w0 - col_lb; o2.fc += w0 - col_lb;
            if ((rc = run_sparse_t<InT, int32_t, KeyT>(c, false, t_data, t_rows, col_ptr, dtype, n_rows, wn, 0, wn,
                                                       flags | ILLICO_FLAG_INPUT_DEVICE, alternative, o2, false, false, true, false)))
                return rc;
            w0 += wn;
        }
        return ILLICO_OK;
    }

    // per-gene stored-entry counts of the requested window
    std::vector<int64_t> gene_nnz(W);
    if (!is_csr) {
        for (int64_t j = 0; j < W; ++j) gene_nnz[j] = (int64_t)h_indptr[col_lb + j + 1] - (int64_t)h_indptr[col_lb + j];
    } else {
        if ((rc = get_scratch(c, "sp_colcnt", std::max<size_t>(W, 1) * 4, &v))) return rc;
        u32 *d_cc = (u32 *)v;
        HIPCHK(c, hipMemsetAsync(d_cc, 0, W * 4, c->stream));
        {
            ProfScope ps(c, KID_SPARSE_SEG);
            hipLaunchKernelGGL((k_csr_col_nnz<IdxT>), dim3(2048), dim3(256), 0, c->stream, d_indices, (long long)total_nnz,
                               (long long)col_lb, (long long)col_ub, d_cc);
            HIPCHK(c, hipGetLastError());
        }
        std::vector<u32> h_cc(W);
        HIPCHK(c, hipMemcpyAsync(h_cc.data(), d_cc, W * 4, hipMemcpyDeviceToHost, c->stream));
        HIPCHK(c, hipStreamSynchronize(c->stream));
        for (int64_t j = 0; j < W; ++j) gene_nnz[j] = h_cc[j];
    }

    // ---- CSC OVO: single-kernel route first; it reports the genes it could not take ----
    // `cols`: the columns the two-kernel route still has to compute.  CSC batches are arbitrary column LISTS (the
    // stragglers of the single-kernel route are batched together); CSR batches are contiguous windows.
    std::vector<int64_t> cols(W);
    for (int64_t j = 0; j < W; ++j) cols[j] = col_lb + j;
    if (counts_route) {
        const int64_t k0 = (int64_t)h_indptr[col_lb], k1 = (int64_t)h_indptr[col_ub];
        bool counts = k1 > k0;
        if (counts && sampled) counts = (double)h_sample[0] <= 0.02 * (double)h_sample[2]; // (taken with the indptr copy above)
        else if (counts) { // count-valued at all?  64k evenly spaced stored values of the window decide
            const int n_samples = (int)std::min<int64_t>(k1 - k0, 1 << 16);
            if ((rc = get_scratch(c, "flag", 16, &v))) return rc;
            u32 *d_cnt = (u32 *)v;
            HIPCHK(c, hipMemsetAsync(d_cnt, 0, 8, c->stream));
            hipLaunchKernelGGL((k_sample_noncount<InT>), dim3((n_samples + 255) / 256), dim3(256), 0, c->stream, d_data + (k0 - kshift),
                               (long long)(k1 - k0), n_samples, CSCC_RT, d_cnt);
            HIPCHK(c, hipGetLastError());
            u32 n_bad[2] = {0, 0};
            HIPCHK(c, hipMemcpyAsync(n_bad, d_cnt, 8, hipMemcpyDeviceToHost, c->stream));
            HIPCHK(c, hipStreamSynchronize(c->stream));
            counts = (double)n_bad[0] <= 0.02 * (double)n_samples; // large integers only take their own genes out (column list)
        }
        if (counts) {
            if ((rc = run_csc_counts_route<InT, IdxT>(c, d_data, d_indices, d_indptr, kshift, d_codes, n_rows, col_lb, flags, alternative, o, cols))) return rc;
            if (cols.empty()) return ILLICO_OK;
        }
    }
    // ---- CSC in float64 whose stored values are float32 values throughout: as for CSR above, behind the histogram route ----
    if constexpr (std::is_same<InT, double>::value) {
        if (!is_csr && in_dev && allow_dense_window && !indices_are_codes && !c->no_f64_narrowing && !c->tap && (int64_t)cols.size() == W && W > 0 &&
            !(flags & ILLICO_FLAG_LOG1P)) { // (utils/sparse/csc.py:207: expm1 of float64 data)
            const long long k0 = (long long)h_indptr[col_lb], k1 = (long long)h_indptr[col_ub];
            if (k1 > k0) {
                if ((rc = get_scratch(c, "flag", 16, &v))) return rc;
                u32 *d_inexact = (u32 *)v;
                HIPCHK(c, hipMemsetAsync(d_inexact, 0, 4, c->stream));
                hipLaunchKernelGGL(k_f64_is_f32, dim3(4096), dim3(256), 0, c->stream, (const double *)d_data + k0, k1 - k0, d_inexact);
                HIPCHK(c, hipGetLastError());
                u32 inexact = 1;
                HIPCHK(c, hipMemcpyAsync(&inexact, d_inexact, 4, hipMemcpyDeviceToHost, c->stream));
                HIPCHK(c, hipStreamSynchronize(c->stream));
                if (!inexact) {
                    if ((rc = get_scratch(c, "sp_f32", (size_t)k1 * sizeof(float), &v))) return rc;
                    hipLaunchKernelGGL(k_f64_to_f32, dim3(4096), dim3(256), 0, c->stream, (const double *)d_data + k0, k1 - k0, (float *)v + k0);
                    HIPCHK(c, hipGetLastError());
                    return run_sparse_t<float, IdxT, u32>(c, false, v, indices, indptr, ILLICO_F32, n_rows, n_cols, col_lb, col_ub, flags & ~ILLICO_FLAG_DEFER, alternative, o,
                                                          allow_dense_window, allow_transpose, indices_are_codes, false);
                }
            }
        }
    }
    // ---- CSC, any values, columns longer than the per-gene LDS kernels hold: a dense window in the matrix's own type + the dense routes
    // (as for CSR above; the columns' row indices must ascend: asked on the device) ----
    if (!is_csr && !indices_are_codes && allow_dense_window && !c->no_csr_densify_any && !c->tap && !c->big_n && n_rows < (1ll << 31) &&
        W > 0 && (many_groups || ((int64_t)cols.size() == W && (double)((int64_t)h_indptr[col_ub] - (int64_t)h_indptr[col_lb]) / (double)W > long_column<KeyT>(c) &&
        (c->ref >= 0 || (double)((int64_t)h_indptr[col_ub] - (int64_t)h_indptr[col_lb]) >= 0.04 * (double)W * (double)n_rows))) &&
        (size_t)n_rows * 64 * sizeof(InT) <= (size_t)c->scratch_bytes) {
        if ((rc = get_scratch(c, "flag", 16, &v))) return rc;
        int *d_bad = (int *)v;
        HIPCHK(c, hipMemsetAsync(d_bad, 0, 8, c->stream));
        // (few, long parcels: the window's entries as one flat run, kernels_sparse.h)
        hipLaunchKernelGGL((k_flat_descents<IdxT>), dim3(4096), dim3(256), 0, c->stream, d_indices, d_indptr + col_lb, (int)W, (long long)kshift, (u32 *)d_bad);
        HIPCHK(c, hipGetLastError());
        u32 h_order[2] = {0u, 0u};
        HIPCHK(c, hipMemcpyAsync(h_order, d_bad, 8, hipMemcpyDeviceToHost, c->stream));
        HIPCHK(c, hipStreamSynchronize(c->stream));
        if (h_order[0] == h_order[1]) {
            int64_t wmax = (int64_t)((size_t)c->scratch_bytes / ((size_t)n_rows * sizeof(InT))) & ~63ll;
            wmax = std::min<int64_t>(wmax, (1ll << 29));
            if (c->gene_batch > 0) wmax = std::min<int64_t>(wmax, (c->gene_batch + 63) & ~63ll);
            constexpr int RC = sizeof(InT) == 4 ? 128 : 64; // (33 KB tiles: four workgroups per CU)
            for (int64_t w0 = col_lb; w0 < col_ub; w0 += wmax) {
                const int64_t wn = std::min<int64_t>(wmax, col_ub - w0), ldD = (wn + 63) & ~63ll;
                if ((rc = get_scratch(c, "dense_window", (size_t)n_rows * ldD * sizeof(InT), &v))) return rc;
                {
                    ProfScope ps(c, KID_DENSIFY);
                    const dim3 grid((unsigned)(ldD / 64), (unsigned)((n_rows + RC * CDN_SUP - 1) / (RC * CDN_SUP)));
                    hipLaunchKernelGGL((k_csc_densify<InT, IdxT, RC>), grid, dim3(CDN_NT), 0, c->stream, d_data, d_indices, d_indptr, (long long)kshift,
                                       (long long)w0, (int)wn, (int)n_rows, (InT *)v, (long long)ldD);
                    HIPCHK(c, hipGetLastError());
                }
                OutPlanes o2 = o;
                o2.p += w0 - col_lb; o2.u += w0 - col_lb; o2.fc += w0 - col_lb;
                if ((rc = run_dense_t<InT, KeyT>(c, v, dtype, n_rows, ldD, 0, wn, (flags | ILLICO_FLAG_INPUT_DEVICE) & ~ILLICO_FLAG_DEFER, alternative, o2))) return rc;
            }
            return ILLICO_OK;
        }
    }
    if (many_groups) return fail(c, ILLICO_ERR_UNSUPPORTED, "CSC input with %d groups: beyond the regrouping kernels' LDS histogram, and the dense window was not available here (row indices out of order?)", G);
    if (!is_csr && !ovr && !c->no_csc_gene_path && !sparse_packed_rank_fits(c)) { // (runs of more than 128 keys leave k_csc_gene, runs of 32 .. 128 are slow in it)
        if ((rc = run_csc_gene_route<InT, IdxT, KeyT>(c, d_data, d_indices, d_indptr, kshift, d_codes, dtype, col_lb, flags, alternative, o, cols, &gene_nnz)))
            return rc;
        if (cols.empty()) return ILLICO_OK;
    }

    if (!is_csr && ovr && !c->no_csc_ovr_gene_path) {
        int64_t max_nnz = 0;
        for (int64_t cc : cols) max_nnz = std::max(max_nnz, gene_nnz[cc - col_lb]);
        if ((rc = run_csc_ovr_route<InT, IdxT, KeyT>(c, d_data, d_indices, d_indptr, kshift, d_codes, dtype, n_rows, col_lb, max_nnz, flags, alternative, o, cols)))
            return rc;
        if (cols.empty()) return ILLICO_OK;
    }

    // ---- two-kernel route (regroup into HBM, then rank) ----
    const bool may_glob = !ovr && !ovo_sort_route_fits<KeyT>(c->h_counts[c->ref], c->max_nonref);
    const size_t per_nnz = sizeof(KeyT) * ((ovr || may_glob) ? 2 : 1) + ((ovr || may_glob) ? 8 : 0);
    const size_t per_gene = (size_t)(G + 1) * 4 * (is_csr ? 2 : 1) + (size_t)G * 24 + 64;
    std::vector<int64_t> list_nnz(cols.size());
    for (size_t j = 0; j < cols.size(); ++j) list_nnz[j] = gene_nnz[cols[j] - col_lb];
    auto batches = plan_batches(list_nnz, 0, per_nnz, per_gene, c->gene_batch, (size_t)c->scratch_bytes); // g0/g1 index `cols`
    {
    for (const SparseBatch &bi : batches) {
        SparseBatch b = bi;
        const int nb = (int)(b.g1 - b.g0);
        const int64_t i0 = b.g0;
        b.g0 = cols[i0];                 // first column (CSR windows are contiguous: cols[i] = col_lb + i)
        b.g1 = cols[i0 + nb - 1] + 1;
        const size_t nnz = (size_t)std::max<int64_t>(b.nnz, 1);
        if ((rc = get_scratch(c, "xt", nnz * sizeof(KeyT), &v))) return rc;
        KeyT *Xs = (KeyT *)v;
        if ((rc = get_scratch(c, "sp_seg", (size_t)nb * (G + 1) * 4, &v))) return rc;
        u32 *seg = (u32 *)v;
        u32 *va = nullptr, *vb = nullptr;
        void *kb = nullptr;
        const int64_t ref_cap_b = ovr ? 0 : std::min<int64_t>(c->h_counts[c->ref], b.max_gene);
        const int64_t grp_cap_b = std::min<int64_t>(c->max_nonref, b.max_gene);
        const bool need_glob = !ovr && !ovo_sort_route_fits<KeyT>(ref_cap_b, grp_cap_b);
        if (ovr || need_glob) {
            if ((rc = get_scratch(c, "ovr_kb", nnz * sizeof(KeyT), &v))) return rc;
            kb = v;
            if ((rc = get_scratch(c, "ovr_va", nnz * 4, &v))) return rc;
            va = (u32 *)v;
            if ((rc = get_scratch(c, "ovr_vb", nnz * 4, &v))) return rc;
            vb = (u32 *)v;
        }
        if ((rc = get_scratch(c, "stats", (size_t)nb * G * 24 + (size_t)nb * 8, &v))) return rc;
        long long *s2u = (long long *)v;
        u64 *stie = (u64 *)(s2u + (size_t)nb * G);
        double *ssum = (double *)(stie + (size_t)nb * G);
        double *gtot = ssum + (size_t)nb * G;
        u32 *gflags = nullptr;
        if (counts_path_allowed(c, flags)) {
            if ((rc = get_scratch(c, "gene_flags", (size_t)nb * 4, &v))) return rc;
            gflags = (u32 *)v;
            HIPCHK(c, hipMemsetAsync(gflags, 0, (size_t)nb * 4, c->stream));
        }
        // CSC: column list of this batch + where each gene's keys start in Xs
        const int *d_cols = nullptr;
        const u32 *d_base = nullptr;
        if (!is_csr) {
            std::vector<int> h_cols(nb);
            std::vector<u32> h_base(nb);
            u32 run = 0;
            for (int j = 0; j < nb; ++j) {
                h_cols[j] = (int)cols[i0 + j];
                h_base[j] = run;
                run += (u32)list_nnz[i0 + j];
            }
            if ((rc = get_scratch(c, "sp_cols", (size_t)nb * 8, &v))) return rc;
            HIPCHK(c, hipMemcpyAsync(v, h_cols.data(), (size_t)nb * 4, hipMemcpyHostToDevice, c->stream));
            HIPCHK(c, hipMemcpyAsync((char *)v + (size_t)nb * 4, h_base.data(), (size_t)nb * 4, hipMemcpyHostToDevice, c->stream));
            HIPCHK(c, hipStreamSynchronize(c->stream)); // the host vectors go out of scope
            d_cols = (const int *)v;
            d_base = (const u32 *)((char *)v + (size_t)nb * 4);
        }
        const int64_t fin_off = is_csr ? b.g0 - col_lb : -col_lb; // CSC: col_map holds absolute columns

        if (!is_csr) {
            ProfScope ps(c, KID_SPARSE_SEG);
            // LDS-staged regroup first (coalesced stores, one read of every entry); genes too large for it are redone by
            // k_csc_segment, which handles any size
            const size_t fixed = (size_t)((G + 1 + 3) & ~3) * 4 + (size_t)CSCG_NT * 4;
            const size_t per_key = sizeof(KeyT);
            const int key_cap = fixed + 4096 < kMaxLds ? (int)std::min<size_t>((kMaxLds - fixed) / per_key, (size_t)CSCG_NT * CSCR_CACHE) : 0;
            u32 *d_fb = nullptr;
            const bool staged = !c->no_csc_regroup_lds && key_cap >= 4096;
            if (staged) {
                if ((rc = get_scratch(c, "sp_fb", (size_t)nb * 4, &v))) return rc;
                d_fb = (u32 *)v;
                HIPCHK(c, hipMemsetAsync(d_fb, 0, (size_t)nb * 4, c->stream));
                CscRegroupParams R;
                R.data = d_data; R.indices = d_indices; R.indptr = d_indptr; R.kshift = kshift; R.col0 = b.g0; R.gene_cols = d_cols;
                R.gene_base = d_base; R.nb = nb; R.codes = d_codes; R.G = G; R.key_cap = key_cap; R.count_limit = ovo_counts_limit(c); R.Xs = Xs;
                R.vals = va; R.seg_ptr = seg; R.gene_flags = gflags; R.fallback = d_fb;
                auto kern = k_csc_regroup<InT, IdxT, KeyT>;
                const size_t lds = fixed + (size_t)key_cap * per_key;
                HIPCHK(c, hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
                hipLaunchKernelGGL(kern, dim3(nb), dim3(CSCG_NT), lds, c->stream, R);
                HIPCHK(c, hipGetLastError());
            }
            auto kern = k_csc_segment<InT, IdxT, KeyT>;
            size_t lds = seg_lds_bytes(G);
            HIPCHK(c, hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
            hipLaunchKernelGGL(kern, dim3(nb), dim3(SEG_NT), lds, c->stream, d_data, d_indices, d_indptr, (long long)b.g0, nb,
                               d_codes, G, Xs, va, seg, gflags, ovo_counts_limit(c), d_cols, d_base, (long long)kshift, (const u32 *)d_fb);
            HIPCHK(c, hipGetLastError());
        } else {
            if ((rc = get_scratch(c, "sp_cursor", (size_t)nb * (G + 1) * 4 + (size_t)nb * 8, &v))) return rc;
            u32 *cursor = (u32 *)v;
            u32 *gene_tot = cursor + (size_t)nb * (G + 1);
            u32 *gene_base = gene_tot + nb;
            HIPCHK(c, hipMemsetAsync(seg, 0, (size_t)nb * (G + 1) * 4, c->stream));
            ProfScope ps(c, KID_SPARSE_SEG);
            const int rows_grid = (int)std::min<int64_t>((n_rows + 3) / 4, 8192);
            hipLaunchKernelGGL((k_csr_count<InT, IdxT>), dim3(rows_grid), dim3(256), 0, c->stream, d_data, d_indices, d_indptr,
                               (int)n_rows, (long long)b.g0, (long long)b.g1, (const int *)c->d_codes, G, seg);
            size_t lds = seg_lds_bytes(G);
            HIPCHK(c, hipFuncSetAttribute((const void *)k_seg_scan, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
            hipLaunchKernelGGL(k_seg_scan, dim3(nb), dim3(SEG_NT), lds, c->stream, seg, G, nb, gene_tot);
            hipLaunchKernelGGL(k_gene_base_scan, dim3(1), dim3(1024), 0, c->stream, (const u32 *)gene_tot, nb, gene_base);
            long long tot = (long long)nb * (G + 1);
            hipLaunchKernelGGL(k_seg_add_base, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, c->stream, seg, cursor,
                               (const u32 *)gene_base, G, nb);
            hipLaunchKernelGGL((k_csr_scatter<InT, IdxT, KeyT>), dim3(rows_grid), dim3(256), 0, c->stream, d_data, d_indices,
                               d_indptr, (int)n_rows, (long long)b.g0, (long long)b.g1, (const int *)c->d_codes, G, cursor, Xs, va,
                               gflags, ovo_counts_limit(c));
            HIPCHK(c, hipGetLastError());
        }

        if (!ovr) {
            OvoParams P;
            P.Xs = Xs; P.gene_stride = 0; P.pos_ptr = c->d_posptr; P.seg_ptr = seg; P.counts = c->d_counts;
            P.G = G; P.ref = (int)c->ref; P.n_genes = nb; P.dt = dtype; P.is_log1p = (flags & ILLICO_FLAG_LOG1P) ? 1 : 0;
            // value sums first (the global-sort fallback permutes Xs), exact whatever order the regroup left the runs in; the
            // rank kernels then leave out_sum alone
            // (CSC: straight from the CSC arrays, accumulators in LDS -- the per-segment kernel spends a wavefront on every (gene, group)
            //  segment, a handful of entries each once there are thousands of groups: 9.4 ms against ~1.5 at C3 shape with 6000 groups)
            if (!is_csr) { if ((rc = launch_csc_value_sums<InT, IdxT>(c, d_data, d_indices, d_indptr, kshift, b.g0, d_cols, d_codes, nb, dtype, flags, ssum))) return rc; }
            else if ((rc = launch_seg_value_sums<KeyT>(c, Xs, seg, nb, dtype, flags, ssum))) return rc;
            P.ref_cap = 0; P.out_2u = s2u; P.out_tie = stie; P.out_sum = nullptr;
            OvoGlobalBufs gb;
            gb.kb = kb; gb.va = va; gb.vb = vb;
            // When the in-LDS sort route holds this batch it serves every gene (its lane-per-group form makes the
            // short runs of a sparse layout cheap for any key type); the histogram route is kept for the sizes it
            // alone can take without the global-sort fallback.
            const u32 *route_flags = need_glob ? gflags : nullptr;
            if (sparse_packed_rank_fits(c, sizeof(KeyT) == 8 && !c->no_sparse_packed_small) && ovo_sort_route_fits<KeyT>(std::min<int64_t>(c->h_counts[c->ref], b.max_gene), 1024)) {
                // groups of hundreds / thousands of cells: the regrouped runs in the packed layout's terms, runs above 256 keys dealt into
                // value buckets, then k_ovo_rank_compact (look-ups in the bucketed reference, pieces of 256 keys); what it leaves -- tie-heavy
                // reference runs -- and the count-valued genes (gflags == 0: k_ovo_counts) go on to launch_ovo
                if ((rc = get_scratch(c, "sp_pk_nnz", (size_t)nb * G * 2 + (size_t)nb * 2 + 64, &v))) return rc;
                u16 *pk_nnz = (u16 *)v, *ref_nnz = pk_nnz + (((size_t)nb * G + 7) & ~(size_t)7);
                if ((rc = get_scratch(c, "sp_pk_gofs", (size_t)nb * G * 4 + (size_t)nb * 4, &v))) return rc;
                u32 *pk_gofs = (u32 *)v, *route = pk_gofs + (size_t)nb * G;
                HIPCHK(c, hipMemsetAsync(route, 0, (size_t)nb * 4, c->stream));
                BigRunFn<KeyT> *big_fn = nullptr;
                {
                    ProfScope ps(c, KID_GROUP_COMPACT);
                    hipLaunchKernelGGL(k_seg_to_packed, dim3((unsigned)(((size_t)nb * G + 255) / 256)), dim3(256), 0, c->stream, (const u32 *)seg, G, nb, (int)c->ref,
                                       pk_nnz, pk_gofs, ref_nnz, route, (const int *)nullptr, (u32 *)nullptr, 0); // (groups of at most 65535 cells here: 16-bit run lengths hold)
                    if (c->pk_nbig > 0) {
                        if ((rc = get_scratch(c, "packed_big_fn", (size_t)nb * c->pk_nbig * sizeof(BigRunFn<KeyT>), &v))) return rc;
                        big_fn = (BigRunFn<KeyT> *)v;
                        const int64_t longest = std::min<int64_t>(c->max_nonref, b.max_gene);
                        int cap = (int)std::min<int64_t>(srt_cap<KeyT>(), (longest + 63) & ~63ll);
                        if (c->big_runs_cap > 0) cap = std::min(cap, std::max(c->big_runs_cap, 512) & ~63);
                        // (runs beyond the LDS slots are dealt through the global sort's second key buffer)
                        if ((rc = launch_bucket_big_runs<KeyT>(c, (void *)Xs, c->no_big_runs_global ? nullptr : kb, 0ll, pk_nnz, pk_gofs, nb, G, cap, big_fn, route, longest, nullptr, nullptr))) return rc;
                    }
                    HIPCHK(c, hipGetLastError());
                }
                {
                    OvoCompactParams C;
                    memset(&C, 0, sizeof C);
                    C.Xs = Xs; C.gene_stride = 0; C.counts = c->d_counts; C.nnz = pk_nnz; C.gofs = pk_gofs; C.ref_out = 0; C.seg_nnz = ref_nnz; C.seg_sum = nullptr;
                    C.out_sum = nullptr; C.G = G; C.ref = (int)c->ref; C.n_genes = nb; C.nseg = 1;
                    packed_ref_sizing<KeyT>(std::min<int64_t>(c->h_counts[c->ref], std::max<int64_t>(b.max_gene, 1)), &C.ref_cap, &C.nbk_lg);
                    if (c->packed_ref_cap > 0) C.ref_cap = std::min(C.ref_cap, std::max(c->packed_ref_cap, 1024));
                    C.out_2u = s2u; C.out_tie = stie; C.route = route; C.big_sorted = c->pk_nbig > 0 ? 1 : 0; C.ref_by_gofs = 1; C.gene_flags = route_flags;
                    C.big_fn = big_fn; C.big_tmp = kb; C.run_cuts = nullptr; C.cand_of = c->pk_nbig > 0 ? c->d_pk_big + c->pk_nbig : nullptr; C.n_cand = c->pk_nbig;
                    const size_t lds = ocr_lds_bytes(C.ref_cap, C.nbk_lg, sizeof(KeyT));
                    const bool eq = c->packed_eq_buckets >= 0 ? c->packed_eq_buckets != 0 : c->h_counts[c->ref] > 16384;
                    // a reference run longer than the kernel's key slots is taken in value-range parts (every part adds its share: the
                    // statistics start from zero; k_ovo_counts writes the count-valued genes' afterwards)
                    C.n_parts = packed_ref_parts<KeyT>(c, std::min<int64_t>(c->h_counts[c->ref], std::max<int64_t>(b.max_gene, 1)), C.ref_cap, C.nbk_lg);
                    const bool parts = C.n_parts > 1;
                    if (parts) {
                        if ((rc = get_scratch(c, "packed_needs_parts", (size_t)nb * 4, &v))) return rc;
                        C.needs_parts = (u32 *)v;
                        HIPCHK(c, hipMemsetAsync(C.needs_parts, 0, (size_t)nb * 4, c->stream));
                        HIPCHK(c, hipMemsetAsync(s2u, 0, (size_t)nb * G * sizeof(long long), c->stream));
                        HIPCHK(c, hipMemsetAsync(stie, 0, (size_t)nb * G * sizeof(u64), c->stream));
                    }
                    auto kern = eq ? k_ovo_rank_compact<KeyT, true> : k_ovo_rank_compact<KeyT, false>;
                    HIPCHK(c, hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
                    ProfScope ps(c, KID_OVO_RANK_COMPACT);
                    hipLaunchKernelGGL(kern, dim3((unsigned)nb), dim3(OCR_NT), lds, c->stream, C);
                    HIPCHK(c, hipGetLastError());
                    if (parts && (rc = launch_rank_parts<KeyT>(c, C, nb, lds))) return rc;
                }
                if ((rc = launch_ovo<KeyT>(c, P, ref_cap_b, grp_cap_b, route_flags, &gb, true, route))) return rc;
            } else
            if ((rc = launch_ovo<KeyT>(c, P, ref_cap_b, grp_cap_b, route_flags, &gb, true))) return rc;
            if ((rc = launch_finalize(c, s2u, stie, ssum, nullptr, nb, flags, alternative, o.p, o.u, o.fc, o.ld, fin_off, d_cols))) return rc;
        } else {
            OvrParams P;
            P.keys_a = Xs; P.keys_b = kb; P.vals_a = va; P.vals_b = vb; P.code_by_pos = nullptr; P.seg_ptr = seg;
            P.stride = 0; P.pos_ptr = nullptr; P.counts = c->d_counts; P.G = G; P.n_genes = nb; P.dt = dtype;
            P.is_log1p = (flags & ILLICO_FLAG_LOG1P) ? 1 : 0; P.n_cells = n_rows; P.ref = -1; P.gene_flags = nullptr;
            P.out_2u = s2u; P.out_tie = stie; P.out_sum = nullptr; P.tie_f64 = 1; // (the sort below permutes Xs: the sums come first)
            if (!is_csr) { if ((rc = launch_csc_value_sums<InT, IdxT>(c, d_data, d_indices, d_indptr, kshift, b.g0, d_cols, d_codes, nb, dtype, flags, ssum))) return rc; }
            else if ((rc = launch_seg_value_sums<KeyT>(c, Xs, seg, nb, dtype, flags, ssum))) return rc;
            if ((rc = launch_ovr_gene<KeyT, true>(c, P))) return rc;
            if ((rc = launch_gene_totals(c, ssum, G, nb, gtot))) return rc;
            if ((rc = launch_finalize(c, s2u, stie, ssum, gtot, nb, flags, alternative, o.p, o.u, o.fc, o.ld, fin_off, d_cols, false, true))) return rc;
        }
    }
    }
    return ILLICO_OK;
}
