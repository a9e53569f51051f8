// Launchers that depend on the KEY type only (u32 for float32 / int32 values, u64 for float64 / int64): instantiated once per key
// type in keyed_u32.hip / keyed_u64.hip; every other translation unit sees explicit-instantiation declarations.
#pragma once
#include "engine.h"
#include <functional>

template <typename KeyT> static size_t ovo_lds_bytes(int ref_cap, bool runend, int nt, bool buckets = false) {
    size_t nw = nt / 64;
    size_t b = ((((size_t)ref_cap + 4) * sizeof(KeyT)) + 15) & ~(size_t)15;
    if (runend) b += ovo_runend_bytes(ref_cap, buckets);
    b += nw * 256 * sizeof(KeyT) + nw * 256 * 4;
    b += nw * 8 * 2 + 16 + 48;
    return b;
}
// Does the in-LDS sort route (k_ovo_rank) hold these sizes?  (reference column in LDS, groups <= 1024 keys)
template <typename KeyT> static bool ovo_sort_route_fits(int64_t max_ref_nnz, int64_t max_grp_nnz) {
    int ref_cap = (int)std::max<int64_t>(max_ref_nnz, 1);
    bool runend = ref_cap <= 65535 && ovo_lds_bytes<KeyT>(ref_cap, true, kOvoThreads) <= kMaxLds;
    return max_grp_nnz <= 1024 && ovo_lds_bytes<KeyT>(ref_cap, runend, kOvoThreads) <= kMaxLds;
}

struct OvoGlobalBufs { // scratch of the global-sort fallback (same element count as the key buffer)
    void *kb = nullptr;
    u32 *va = nullptr, *vb = nullptr;
};

// ---- packed dense OVO route (kernels_ovo_compact.h) ----
// LDS sizing of the packed rank kernel: key slots for the reference's NON-ZERO keys and the bucket count (2^lg, half a byte each).
// A reference whose every cell fits beside 2^17 buckets gets one slot per cell; a larger one gets the slots that fit beside 2^16
// buckets -- an expression matrix is mostly zeros, so the non-zeros of a 33 000-cell reference still fit -- and a gene whose
// non-zeros exceed the slots is left to k_ovo_rank by the kernel (as the tie-heavy ones are).
template <typename KeyT> static void packed_ref_sizing(int64_t n_ref, int *cap, int *lg) {
    if (ocr_lds_bytes((int)n_ref, 17, sizeof(KeyT)) <= kMaxLds) {
        // (a small reference: ~32 buckets per key are as good as 2^17 of them, and a table of 2^14 .. 2^16 buckets is zeroed and scanned in a
        //  quarter of the time -- what a gene of a wide matrix, 120 000 genes x 20 000 cells, mostly costs)
        int l = 14;
        while (l < 17 && (32ll << 0) * n_ref > (1ll << l)) ++l;
        *cap = (int)n_ref; *lg = l;
        return;
    }
    // 2^17 buckets while 55 % of the reference's cells would still fit the slots left beside them, else 2^16 and more slots
    const size_t fixed17 = ocr_lds_bytes(0, 17, sizeof(KeyT));
    const int64_t cap17 = fixed17 < kMaxLds ? (int64_t)((kMaxLds - fixed17) / sizeof(KeyT)) - 8 : 0;
    if (cap17 > 0 && n_ref * 55 <= cap17 * 100) { *lg = 17; *cap = (int)std::min<int64_t>(n_ref, cap17); return; }
    *lg = 16;
    const size_t fixed = ocr_lds_bytes(0, 16, sizeof(KeyT));
    *cap = (int)std::min<int64_t>(n_ref, (int64_t)((kMaxLds - fixed) / sizeof(KeyT)) - 8);
}
// Value-range parts (k_ovo_rank_compact<.., PARTS>): how many a gene of this reference may need -- every cell non-zero, an eighth of
// the slots kept as slack for cuts that fall on cell boundaries -- or 1 when its cells fit the slots anyway (or the table is too small
// to count the 4096 cells in, or "no_ovo_parts").  Genes that would need more than 32 parts are left to the general route by the kernel.
template <typename KeyT> static int packed_ref_parts(const illico_ctx *c, int64_t n_ref, int cap, int lg) {
    if (c->no_ovo_parts || n_ref <= cap || lg < 16 || cap < 1024) return 1;
    const int64_t cap_s = cap - cap / 8;
    return (int)std::min<int64_t>(32, (n_ref + cap_s - 1) / cap_s);
}
// Sizes the route holds.  The reference may be of any size: it
// is packed in 512-row segments, the rank kernel keeps as many of its NON-ZERO keys as LDS holds (packed_ref_sizing) and leaves a gene
// with more -- like the tie-heavy ones -- to k_ovo_rank, or, when that kernel's LDS does not hold the reference either (or groups exceed
// 1024 cells), to the general sort route (run_ovo_packed: redo).  (Until sweep 10 the route asked for a reference of at most 65535
// cells: the control group of a two-million-cell atlas -- 66 667 cells -- sent every gene to the radix sort in HBM, 287 ms for 9.6 GB.)
template <typename KeyT> static bool packed_route_fits(const illico_ctx *c) {
    if (c->ref < 0 || c->no_packed_dense) return false;
    const int64_t n_ref = c->h_counts[c->ref];
    // (ranked groups of any size: a (gene, group) run's 16-bit count saturates, and a run beyond k_bucket_big_runs' LDS slots -- 16 384 or
    //  32 768 keys -- sends its gene to the general route; a cluster of 100 000 cells a tenth of whose values are stored stays here)
    return n_ref >= 1 && n_ref < (1ll << 24) && c->max_nonref < (1ll << 24);
}
// ... and whether what the packed kernel leaves can go to k_ovo_rank over the same layout
template <typename KeyT> static bool packed_leftovers_fit_sort_route(const illico_ctx *c) {
    return c->max_nonref <= 1024 && ovo_sort_route_fits<KeyT>(c->h_counts[c->ref], c->max_nonref);
}


// the two launches that deal the runs of more than 256 keys into value buckets (kernels_ovo_compact.h): through LDS up to `cap` keys, through
// the second key buffer `tmp` beyond (tmp == nullptr: such a run sends its gene to the general route)
template <typename KeyT>
static int launch_bucket_big_runs(illico_ctx *c, void *Xs, void *tmp, long long stride, const u16 *nnz, const u32 *gofs, int nb, int G, int cap, BigRunFn<KeyT> *big_fn,
                                  u32 *route, int64_t longest_run, const u32 *run_n, u32 *run_cuts /* optional [nb][pk_nbig][OCR_CUTS] */) {
    const bool global = tmp != nullptr && longest_run > cap;
    // a workgroup takes a stretch of a gene's candidates when there are many (most hold no run above 256 keys)
    const unsigned gx = (long long)c->pk_nbig * nb <= 32768 ? (unsigned)c->pk_nbig : (unsigned)std::max(1, std::min(c->pk_nbig, (32768 + nb - 1) / nb));
    // runs up to 8192 keys: 256 threads each; longer ones (they take the LDS of half a CU and more): 1024 threads, a launch of their own
    // (10 groups of 30 000 cells half non-zero: 6.9 -> 5.6 ms; at 1024 threads, runs of 3000 keys took 9.1 instead of 8.2)
    const int cap_a = c->no_big_runs_wide ? cap : std::min(cap, 8192);
    {
        auto kern = k_bucket_big_runs<KeyT, SRT_NT>;
        const size_t lds = srt_lds_bytes(sizeof(KeyT), cap_a);
        HIPCHK(c, hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        hipLaunchKernelGGL(kern, dim3(gx, nb), dim3(SRT_NT), lds, c->stream, Xs, stride, nnz, gofs, (const int *)c->d_pk_big, c->pk_nbig, G, cap_a, big_fn, route,
                           (global || cap > cap_a) ? 1 : 0, 64 * OCR_KMAX, run_cuts);
        HIPCHK(c, hipGetLastError());
    }
    if (cap > cap_a) {
        auto kern = k_bucket_big_runs<KeyT, 1024>;
        const size_t lds = srt_lds_bytes(sizeof(KeyT), cap);
        HIPCHK(c, hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        hipLaunchKernelGGL(kern, dim3(gx, nb), dim3(1024), lds, c->stream, Xs, stride, nnz, gofs, (const int *)c->d_pk_big, c->pk_nbig, G, cap, big_fn, route, global ? 1 : 0, cap_a, run_cuts);
        HIPCHK(c, hipGetLastError());
    }
    if (global) {
        auto kg = k_bucket_big_runs_global<KeyT>;
        int lg = 6;
        while (lg < SRT_LG_MAX_G && (4ll << lg) < longest_run) ++lg;
        if (c->big_runs_cap > 0) lg = std::min(lg, 11); // (tests: few counters as well)
        // LDS: the counters, and behind them the key slots of one slice of the output -- everything a CU has (one workgroup of 1024 threads
        // each): a slice costs a read of the run, and ten clusters of 100 000 cells half non-zero took 6.2 ms with 48 KB of slots (two
        // workgroups per CU), 4.9 with 96 KB, 4.1 with 126 KB
        const size_t cnt_bytes = (size_t)4 << lg;
        size_t slice_bytes = c->big_runs_slice_bytes > 0 ? (size_t)c->big_runs_slice_bytes : (size_t)kMaxLds - cnt_bytes - 8192; // (8 KB: the kernel's own static LDS)
        slice_bytes = std::min(std::max(slice_bytes, (size_t)512 * sizeof(KeyT)), (size_t)kMaxLds - cnt_bytes - 8192);
        const int slice_keys = (int)(slice_bytes / sizeof(KeyT));
        const size_t ldsg = cnt_bytes + (size_t)slice_keys * sizeof(KeyT);
        HIPCHK(c, hipFuncSetAttribute((const void *)kg, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsg));
        hipLaunchKernelGGL(kg, dim3(gx, nb), dim3(SRTG_NT), ldsg, c->stream, Xs, tmp, stride, nnz, gofs, (const int *)c->d_pk_big, c->pk_nbig, G, cap, lg, big_fn, route, run_n, slice_keys, run_cuts);
        HIPCHK(c, hipGetLastError());
    }
    return ILLICO_OK;
}


// The PARTS launch of the packed rank kernel with what it needs in front of it (kernels_ovo_compact.h): the cuts of every gene the plain
// kernel handed over (k_ref_cuts), and -- for genes of at most four parts -- their short runs dealt by part (k_deal_runs).
template <typename KeyT>
static int launch_rank_parts(illico_ctx *c, OvoCompactParams C, int nb, size_t lds) {
    void *v;
    int rc;
    if ((rc = get_scratch(c, "packed_cuts", (size_t)nb * sizeof(PartCuts<KeyT>), &v))) return rc;
    PartCuts<KeyT> *cuts = (PartCuts<KeyT> *)v;
    C.cuts = cuts; C.pofs = nullptr;
    hipLaunchKernelGGL((k_ref_cuts<KeyT>), dim3((unsigned)nb), dim3(1024), 0, c->stream, C, cuts);
    HIPCHK(c, hipGetLastError());
    if (!c->no_deal_runs && C.n_parts <= 4) { // (more parts: the masked look-ups of the parts kernel)
        if ((rc = get_scratch(c, "packed_pofs", (size_t)nb * (size_t)C.G * 8, &v))) return rc;
        u16 *pofs = (u16 *)v;
        const unsigned gx = (unsigned)std::max(1, std::min((C.G + DEAL_NT / 64 - 1) / (DEAL_NT / 64), 64));
        hipLaunchKernelGGL((k_deal_runs<KeyT>), dim3(gx, (unsigned)nb), dim3(DEAL_NT), 0, c->stream, C, (const PartCuts<KeyT> *)cuts, pofs);
        HIPCHK(c, hipGetLastError());
        C.pofs = pofs;
    }
    auto kp = k_ovo_rank_compact<KeyT, true, true>;
    HIPCHK(c, hipFuncSetAttribute((const void *)kp, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL(kp, dim3((unsigned)nb * (unsigned)C.n_parts), dim3(OCR_NT), lds, c->stream, C);
    HIPCHK(c, hipGetLastError());
    return ILLICO_OK;
}

constexpr int kOvrThreads = 256; // several small workgroups per CU overlap each other's barriers (1024 measured the same)

struct OvrPackedInput {
    const u16 *nnz;
    const u32 *blk_cnt;
    std::function<int(int, int)> repad;
};

// ---- the launchers (keyed_impl.h; instantiated in keyed_u32.hip / keyed_u64.hip) ----
template <typename KeyT> int launch_ovr_partition_packed_coop(illico_ctx *c, const OvrPartPackedParams &Q, int nb); // (keyed_coop.hip)
template <typename KeyT> int launch_seg_value_sums(illico_ctx *c, const KeyT *Xs, const u32 *seg, int nb, int dtype, int flags, double *ssum);
template <typename KeyT> int launch_group_sums_rows(illico_ctx *c, const KeyT *Xt, int64_t stride, int nb, int dtype, int flags, double *ssum);
// Per-group accumulators in LDS when they fit, else in HBM (one [3*G] u64 block per gene of the batch).
template <typename KeyT, bool SPARSE, bool OVO = false> int launch_ovr_gene(illico_ctx *c, OvrParams P);
// padded = true: Xt is the padded dense layout of k_group_compact (slot codes c->d_pk_code, c->pk_stride slots per gene; the value
// sums are in ssum already)
template <typename KeyT>
int run_ovr_dense_batch(illico_ctx *c, KeyT *Xt, int64_t stride, int nb, int N, int dtype, int flags, long long *s2u, u64 *stie, double *ssum,
                        double *gtot, bool padded = false);
template <typename KeyT>
int run_ovr_dense_parts(illico_ctx *c, KeyT *Xt, int64_t stride, int nb, int N, int dtype, int flags, long long *s2u, u64 *stie, double *ssum,
                        double *gtot, bool *done, bool padded = false, const OvrPackedInput *packed = nullptr);
template <typename KeyT>
int launch_ovo(illico_ctx *c, OvoParams P, int64_t max_ref_nnz, int64_t max_grp_nnz, const u32 *flags, const OvoGlobalBufs *gb, bool sparse,
               const u32 *only = nullptr);

#define ILLICO_KEYED_INSTANCES(X, KeyT)                                                                                                  \
    X template int launch_seg_value_sums<KeyT>(illico_ctx *, const KeyT *, const u32 *, int, int, int, double *);                            \
    X template int launch_group_sums_rows<KeyT>(illico_ctx *, const KeyT *, int64_t, int, int, int, double *);                               \
    X template int launch_ovr_gene<KeyT, true, false>(illico_ctx *, OvrParams);                                                             \
    X template int launch_ovr_gene<KeyT, false, false>(illico_ctx *, OvrParams);                                                            \
    X template int launch_ovr_gene<KeyT, true, true>(illico_ctx *, OvrParams);                                                              \
    X template int launch_ovr_gene<KeyT, false, true>(illico_ctx *, OvrParams);                                                             \
    X template int run_ovr_dense_batch<KeyT>(illico_ctx *, KeyT *, int64_t, int, int, int, int, long long *, u64 *, double *, double *, bool); \
    X template int run_ovr_dense_parts<KeyT>(illico_ctx *, KeyT *, int64_t, int, int, int, int, long long *, u64 *, double *, double *, bool *, bool, const OvrPackedInput *); \
    X template int launch_ovo<KeyT>(illico_ctx *, OvoParams, int64_t, int64_t, const u32 *, const OvoGlobalBufs *, bool, const u32 *);
#ifndef ILLICO_KEYED_IMPL
ILLICO_KEYED_INSTANCES(extern, u32)
ILLICO_KEYED_INSTANCES(extern, u64)
#endif
