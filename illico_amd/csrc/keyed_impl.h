// Definitions of the key-type launchers declared in keyed_driver.h (included by keyed_u32.hip / keyed_u64.hip only).
#pragma once
#define ILLICO_KEYED_IMPL
#include "keyed_driver.h"
template <typename KeyT>
int launch_seg_value_sums(illico_ctx *c, const KeyT *Xs, const u32 *seg, int nb, int dtype, int flags, double *ssum) {
    SegSumsParams P;
    P.Xs = Xs; P.seg_ptr = seg; P.nb = nb; P.G = (int)c->n_groups; P.dt = dtype; P.is_log1p = (flags & ILLICO_FLAG_LOG1P) ? 1 : 0; P.out_sum = ssum;
    ProfScope ps(c, KID_VALUE_SUMS);
    hipLaunchKernelGGL((k_seg_value_sums<KeyT>), dim3(nb), dim3(SUMS_NT), 0, c->stream, P);
    HIPCHK(c, hipGetLastError());
    return ILLICO_OK;
}
template <typename KeyT>
int launch_group_sums_rows(illico_ctx *c, const KeyT *Xt, int64_t stride, int nb, int dtype, int flags, double *ssum) {
    ProfScope ps(c, KID_VALUE_SUMS);
    hipLaunchKernelGGL((k_group_sums_rows<KeyT>), dim3(nb), dim3(SUMS_NT), 0, c->stream, Xt, (long long)stride, nb, (const int *)c->d_posptr,
                       (int)c->n_groups, dtype, (flags & ILLICO_FLAG_LOG1P) ? 1 : 0, ssum);
    HIPCHK(c, hipGetLastError());
    return ILLICO_OK;
}
// Per-group accumulators in LDS when they fit, else in HBM (one [3*G] u64 block per gene of the batch).
template <typename KeyT, bool SPARSE, bool OVO>
int launch_ovr_gene(illico_ctx *c, OvrParams P) {
    constexpr bool DC = !SPARSE && !OVO; // dense OVR: compact the zeros away inside the kernel
    const bool accg = ovr_lds_bytes(P.G, SPARSE || DC, OVO, kOvrThreads, false) > kMaxLds;
    const size_t lds = ovr_lds_bytes(P.G, SPARSE || DC, OVO, kOvrThreads, accg);
    P.acc_global = nullptr;
    P.seg_begin = P.seg_end = nullptr;
    P.seg_n = nullptr;
    if (accg) {
        void *v;
        int rc = get_scratch(c, "ovr_acc", (size_t)P.n_genes * 3 * P.G * 8, &v);
        if (rc) return rc;
        P.acc_global = (u64 *)v;
    }
    // One kernel per gene does everything: per-group sums, zero compaction, a stable LSD radix sort of the (key, group)
    // pairs in HBM (2048-element rounds, one 256-thread workgroup per gene) and the rank sweeps.  (Round 1 sorted with
    // rocPRIM's segmented radix sort here -- 29 G pairs/s against this kernel's 12 -- ; the library is gone from the build:
    // this route only takes what the LDS routes leave -- tie-heavy dense columns outside the 64-value table, CSC genes larger
    // than the LDS key buffer -- and no BASELINE configuration reaches it.)
    ProfScope ps(c, KID_OVR_SCAN);
    if (accg) {
        auto kern = k_ovr_gene<KeyT, SPARSE, OVO, kOvrThreads, true, DC>;
        HIPCHK(c, hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        hipLaunchKernelGGL(kern, dim3(P.n_genes), dim3(kOvrThreads), lds, c->stream, P);
    } else {
        auto kern = k_ovr_gene<KeyT, SPARSE, OVO, kOvrThreads, false, DC>;
        HIPCHK(c, hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        hipLaunchKernelGGL(kern, dim3(P.n_genes), dim3(kOvrThreads), lds, c->stream, P);
    }
    HIPCHK(c, hipGetLastError());
    return ILLICO_OK;
}

// padded = true: Xt is the padded dense layout of k_group_compact (slot codes c->d_pk_code, c->pk_stride slots per gene; the value
// sums are in ssum already)
template <typename KeyT>
int run_ovr_dense_batch(illico_ctx *c, KeyT *Xt, int64_t stride, int nb, int N, int dtype, int flags,
                        long long *s2u, u64 *stie, double *ssum, double *gtot, bool padded) {
    void *v;
    int rc;
    if ((rc = get_scratch(c, "ovr_kb", (size_t)nb * stride * sizeof(KeyT), &v))) return rc;
    void *kb = v;
    if ((rc = get_scratch(c, "ovr_va", (size_t)nb * stride * 4, &v))) return rc;
    u32 *va = (u32 *)v;
    if ((rc = get_scratch(c, "ovr_vb", (size_t)nb * stride * 4, &v))) return rc;
    u32 *vb = (u32 *)v;
    OvrParams P;
    P.keys_a = Xt; P.keys_b = kb; P.vals_a = va; P.vals_b = vb;
    P.code_by_pos = padded ? c->d_pk_code : c->d_code_by_pos; P.seg_ptr = nullptr; P.stride = stride; P.pos_ptr = c->d_posptr;
    P.counts = c->d_counts; P.G = (int)c->n_groups; P.n_genes = nb; P.dt = dtype;
    P.is_log1p = (flags & ILLICO_FLAG_LOG1P) ? 1 : 0; P.n_cells = N; P.scan_len = padded ? c->pk_len : 0; P.ref = -1; P.gene_flags = nullptr;
    P.out_2u = s2u; P.out_tie = stie; P.out_sum = padded ? nullptr : ssum; P.tie_f64 = 0;
    if ((rc = launch_ovr_gene<KeyT, false>(c, P))) return rc;
    return launch_gene_totals(c, ssum, (int)c->n_groups, nb, gtot);
}

// Dense OVR, any values, in value-range parts (kernels_csc_ovr.h): partition each gene's non-zero keys into parts that
// fit the LDS key buffer, rank every part with the bucket / sorted forms of k_csc_ovr_gene, sum the parts up.  Genes whose
// values crowd into one coarse bucket (heavy ties) are recomputed by the general route over the gene range that covers
// them (run by run).  *done = false: the route does not apply (nothing was launched).
// packed = non-null: Xt holds the PACKED rows of k_group_compact (non-zero keys only; nnz / blk_cnt beside them): the partition walks
// those; genes that leave the route are first written again in the padded layout (repad(first gene, count)) for the general route.
template <typename KeyT>
int run_ovr_dense_parts(illico_ctx *c, KeyT *Xt, int64_t stride, int nb, int N, int dtype, int flags,
                        long long *s2u, u64 *stie, double *ssum, double *gtot, bool *done, bool padded,
                        const OvrPackedInput *packed) {
    *done = false;
    const int G = (int)c->n_groups;
    if (c->no_ovr_parts_path || G > 65535 || c->max_nonref >= (1ll << 23) ||
        (double)c->max_nonref * 2.0 * (double)N >= (double)(1ull << CSCO_CNT_SHIFT))
        return ILLICO_OK;
    // Two workgroups of 512 threads per CU, each with half of the LDS (parts half as long, twice as many), while a part still holds 8192
    // keys beside the groups' accumulators: one workgroup's barriers and first loads of a part run under the other's ranking -- the rank
    // kernel 8.0 -> 6.0 ms at C2 shape, the partition (twice the parts to write) 4.8 -> 6.1: 17.5 -> 16.7 ms.  "ovr_rank_whole" = 1: one
    // workgroup of 1024 threads, as before round 5.
    // (columns too long for 128 half-size parts -- two million cells -- take whole-size parts, up to OVRP_PMAX = 255 of them: the general
    //  route instead was 151 ms at 2 000 000 x 1200 continuous, where the parts are 17)
    const bool half = !c->ovr_rank_whole && csco_key_cap(G, 13, sizeof(KeyT), kMaxLds / 2, true) >= 8192 &&
                      (int64_t)(csco_key_cap(G, 13, sizeof(KeyT), kMaxLds / 2, true) & ~1023) * 128 >= N;
    const size_t lds_budget = half ? kMaxLds / 2 : kMaxLds;
    int lg = half ? 13 : 14;
    if (csco_key_cap(G, lg, sizeof(KeyT), lds_budget, true) < 12288 && !half) lg = 13;
    const int key_cap = csco_key_cap(G, lg, sizeof(KeyT), lds_budget, true);
    int cap = key_cap & ~1023; // any part can also take the sorted form (1024-key chunks)
    if (cap < 4096) return ILLICO_OK;
    if (c->ovr_parts_cap > 0) cap = (int)std::min<int64_t>(cap, std::max<int64_t>(1024, c->ovr_parts_cap & ~1023ll)); // tests: many small parts
    if ((int64_t)cap * OVRP_PMAX < N) return ILLICO_OK;
    int rc;
    void *v;
    if ((rc = get_scratch(c, "ovr_kb", (size_t)nb * stride * sizeof(KeyT), &v))) return rc;
    void *pkeys = v;
    if ((rc = get_scratch(c, "ovr_va", (size_t)nb * stride * 4, &v))) return rc;
    u16 *pcodes = (u16 *)v;
    const size_t meta = (size_t)nb * ((OVRP_PMAX + 1) * 4 + 16 + 4) + 64;
    if ((rc = get_scratch(c, "ovr_parts_meta", meta, &v))) return rc;
    u32 *part_start = (u32 *)v;
    u32 *gene_info = part_start + (size_t)nb * (OVRP_PMAX + 1);
    u32 *gflag = gene_info + (size_t)nb * 4;
    u32 *gene_ctr = gflag + nb; // the rank kernel's queue head
    HIPCHK(c, hipMemsetAsync(gflag, 0, (size_t)nb * 4, c->stream));
    HIPCHK(c, hipMemsetAsync(gene_ctr, 0, 4, c->stream));
    // per-group value sums from the group-contiguous key rows, in a fixed order (the parts see a group's values in an order
    // that depends on timing)
    if (!padded && (rc = launch_group_sums_rows<KeyT>(c, Xt, stride, nb, dtype, flags, ssum))) return rc;
    if (packed) {
        OvrPartPackedParams Q;
        Q.Xt = Xt; Q.stride = stride; Q.n_genes = nb; Q.G = G; Q.nblk = c->pk_nblk; Q.nnz = packed->nnz; Q.blk_cnt = packed->blk_cnt;
        Q.blk_g0 = c->d_pk_blk; Q.blk_g1 = c->d_pk_blk + c->pk_nblk; Q.blk_out = c->d_pk_blk + 2 * c->pk_nblk; Q.cap = cap;
        Q.out_keys = pkeys; Q.out_codes = pcodes; Q.part_start = part_start; Q.gene_info = gene_info;
        Q.n_long = c->no_ovr_part_coop ? 0 : c->pk_nlong; Q.long_blk = c->d_pk_long; Q.blk_is_long = c->d_pk_islong;
        ProfScope ps(c, KID_OVR_PART);
        if (Q.n_long > 0) { // (the COOP form lives in keyed_coop.hip)
            if ((rc = launch_ovr_partition_packed_coop<KeyT>(c, Q, nb))) return rc;
        } else {
            auto kern = k_ovr_partition_packed<KeyT, false>;
            HIPCHK(c, hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ovrp_lds_bytes()));
            hipLaunchKernelGGL(kern, dim3(nb), dim3(OVRP_NT), ovrp_lds_bytes(), c->stream, Q);
            HIPCHK(c, hipGetLastError());
        }
    } else {
        OvrPartParams Q;
        Q.Xt = Xt; Q.stride = stride; Q.n_genes = nb; Q.n_cells = padded ? (int)c->pk_len : N; Q.code_by_pos = padded ? c->d_pk_code : c->d_code_by_pos; Q.cap = cap;
        Q.out_keys = pkeys; Q.out_codes = pcodes; Q.part_start = part_start; Q.gene_info = gene_info;
        ProfScope ps(c, KID_OVR_PART);
        auto kern = k_ovr_partition<KeyT>;
        HIPCHK(c, hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ovrp_lds_bytes()));
        hipLaunchKernelGGL(kern, dim3(nb), dim3(OVRP_NT), ovrp_lds_bytes(), c->stream, Q);
        HIPCHK(c, hipGetLastError());
    }
    {
        // one workgroup per gene (drawn from a queue), the gene's parts one after the other: kernels_ovr_parts.h
        OvrRankGeneParams P;
        memset(&P, 0, sizeof P);
        P.pkeys = pkeys; P.pcodes = pcodes; P.pstride = stride; P.part_start = part_start; P.gene_info = gene_info;
        P.gflag = gflag; P.gene_counter = gene_ctr; P.nb = nb; P.G = G; P.counts = c->d_counts; P.n_cells = N;
        P.key_cap = key_cap; P.lg_buckets = lg; P.force_sorted = c->csc_ovr_sorted_form ? 1 : 0;
        P.out_2u = s2u; P.out_tie = stie;
        const size_t lds = csco_fixed_lds_bytes(G, lg, true) + (size_t)(key_cap + 4) * sizeof(KeyT);
        ProfScope ps(c, KID_OVR_RANK_PARTS);
        int n_cu = 256;
        hipDeviceGetAttribute(&n_cu, hipDeviceAttributeMultiprocessorCount, c->device);
        if (half) {
            auto kern = k_ovr_rank_gene_parts<KeyT, 512>;
            HIPCHK(c, hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
            const unsigned grid = (unsigned)std::min<long long>((long long)nb, 2ll * std::max(n_cu, 1)); // two resident workgroups per CU
            hipLaunchKernelGGL(kern, dim3(grid), dim3(512), lds, c->stream, P);
        } else {
            auto kern = k_ovr_rank_gene_parts<KeyT>;
            HIPCHK(c, hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
            const unsigned grid = (unsigned)std::min<long long>((long long)nb, std::max(n_cu, 1)); // one resident workgroup per CU (LDS)
            hipLaunchKernelGGL(kern, dim3(grid), dim3(CSCO_NT), lds, c->stream, P);
        }
        HIPCHK(c, hipGetLastError());
    }
    // genes that left the route (a coarse bucket too full, a part that fits neither form): the general route, over the
    // gene range that covers them
    std::vector<u32> h((size_t)nb * 5);
    HIPCHK(c, hipMemcpyAsync(h.data(), gene_info, (size_t)nb * 16, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipMemcpyAsync(h.data() + (size_t)nb * 4, gflag, (size_t)nb * 4, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    // (runs of flagged genes closer than 16 genes are merged: the genes in between are recomputed, identically)
    for (int j = 0; j < nb;) {
        if (!(h[(size_t)j * 4 + 3] || h[(size_t)nb * 4 + j])) { ++j; continue; }
        int first = j, last = j;
        for (int e = j + 1; e < nb && e - last <= 16; ++e)
            if (h[(size_t)e * 4 + 3] || h[(size_t)nb * 4 + e]) last = e;
        const int sub = last - first + 1;
        if (packed && (rc = packed->repad(first, sub))) return rc;
        if ((rc = run_ovr_dense_batch<KeyT>(c, Xt + (size_t)first * stride, stride, sub, N, dtype, flags, s2u + (size_t)first * G,
                                            stie + (size_t)first * G, ssum + (size_t)first * G, gtot + first, padded)))
            return rc;
        j = last + 1;
    }
    *done = true;
    return launch_gene_totals(c, ssum, G, nb, gtot);
}
template <typename KeyT, int KMAX, bool RUNEND, bool LG>
static int launch_ovo_t(illico_ctx *c, const OvoParams &P, size_t lds, const u32 *flags) {
    auto kern = k_ovo_rank<KeyT, KMAX, RUNEND, kOvoThreads, LG>;
    HIPCHK(c, hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    ProfScope ps(c, KID_OVO_RANK);
    hipLaunchKernelGGL(kern, dim3(P.n_genes), dim3(kOvoThreads), lds, c->stream, P, flags);
    HIPCHK(c, hipGetLastError());
    return ILLICO_OK;
}
// flags: per-gene routing word written by the ingest kernels (0 = count-valued gene -> k_ovo_counts,
// non-zero -> sort route); nullptr routes every gene through the sort route.  The sort route is k_ovo_rank when
// the reference column and the groups fit LDS / registers, else the per-gene global radix sort (k_ovr_gene in
// OVO mode), which has no size limit.
template <typename KeyT>
int launch_ovo(illico_ctx *c, OvoParams P, int64_t max_ref_nnz, int64_t max_grp_nnz, const u32 *flags,
               const OvoGlobalBufs *gb, bool sparse, const u32 *only) {
    // only (non-null; flags null): the genes with a non-zero word alone (what the packed rank kernel left)
    if (only) P.only = only;
    if (flags) { // (the ingest kernels flagged with the same limit: ovo_counts_limit)
        ProfScope ps(c, KID_OVO_COUNTS);
        // few genes (the leftovers of a count matrix): a gene's groups over several workgroups, 512 groups (eight wavefronts x 64) each at least
        const int splits = std::max(1, std::min({(1024 + P.n_genes - 1) / std::max(P.n_genes, 1), (P.G + 511) / 512, 16}));
        if (ovo_counts_limit(c) == COUNTS_R8) hipLaunchKernelGGL((k_ovo_counts<KeyT, COUNTS_R8, 8>), dim3(P.n_genes, splits), dim3(COUNTS_NT), 0, c->stream, P, flags);
        else hipLaunchKernelGGL((k_ovo_counts<KeyT, COUNTS_R, 16>), dim3(P.n_genes, splits), dim3(COUNTS_NT), 0, c->stream, P, flags);
        HIPCHK(c, hipGetLastError());
    }
    if (!ovo_sort_route_fits<KeyT>(max_ref_nnz, max_grp_nnz)) {
        if (!gb || !gb->kb) return fail(c, ILLICO_ERR_UNSUPPORTED, "internal: global-sort scratch missing");
        OvrParams Q;
        Q.keys_a = (void *)P.Xs; Q.keys_b = gb->kb; Q.vals_a = gb->va; Q.vals_b = gb->vb;
        Q.code_by_pos = sparse ? nullptr : c->d_code_by_pos; Q.seg_ptr = P.seg_ptr; Q.stride = P.gene_stride;
        Q.pos_ptr = P.pos_ptr; Q.counts = P.counts; Q.G = P.G; Q.n_genes = P.n_genes; Q.dt = P.dt; Q.is_log1p = P.is_log1p;
        Q.n_cells = c->n_cells; Q.ref = P.ref; Q.gene_flags = only ? only : flags;
        Q.out_2u = P.out_2u; Q.out_tie = P.out_tie; Q.out_sum = P.out_sum; Q.tie_f64 = 0;
        return sparse ? launch_ovr_gene<KeyT, true, true>(c, Q) : launch_ovr_gene<KeyT, false, true>(c, Q);
    }
    int ref_cap = (int)std::max<int64_t>(max_ref_nnz, 1);
    bool runend = ref_cap <= 65535 && ovo_lds_bytes<KeyT>(ref_cap, true, kOvoThreads) <= kMaxLds;
    // bucket form of the reference column (no sort, short look-ups): needs the 16-bit table beside the keys
    const bool buckets = runend && !c->no_ovo_ref_buckets && ref_cap <= 65531 && ovo_lds_bytes<KeyT>(ref_cap, true, kOvoThreads, true) <= kMaxLds;
    size_t lds = ovo_lds_bytes<KeyT>(ref_cap, runend, kOvoThreads, buckets);
    P.ref_cap = ref_cap;
    P.ref_buckets = buckets ? 1 : 0;
    bool big = max_grp_nnz > 256;
    if (sparse) { // sparse layouts get the lane-per-group form as well
        if (big) return runend ? launch_ovo_t<KeyT, 16, true, true>(c, P, lds, flags) : launch_ovo_t<KeyT, 16, false, true>(c, P, lds, flags);
        return runend ? launch_ovo_t<KeyT, 4, true, true>(c, P, lds, flags) : launch_ovo_t<KeyT, 4, false, true>(c, P, lds, flags);
    }
    if (big) return runend ? launch_ovo_t<KeyT, 16, true, false>(c, P, lds, flags) : launch_ovo_t<KeyT, 16, false, false>(c, P, lds, flags);
    return runend ? launch_ovo_t<KeyT, 4, true, false>(c, P, lds, flags) : launch_ovo_t<KeyT, 4, false, false>(c, P, lds, flags);
}
