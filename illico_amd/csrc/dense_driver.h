// Dense-input drivers, templated on the value type: instantiated in dense_<type>.hip.
#pragma once
#include "keyed_driver.h"
#include "host_narrow.h"
#ifndef ILLICO_DENSE_U8_UNIT // the fused kernels on byte windows are instantiated in dense_u8.hip only
extern template int run_fused_ovo<uint8_t>(illico_ctx *, const void *, int64_t, int64_t, int, int, int, const OutPlanes &, int64_t, std::vector<u32> &, int, bool, int64_t, const u32 *);
#endif

// k_group_compact over one gene batch; pack = false: the padded dense layout (every key kept, sums only)
template <typename InT, typename KeyT>
static int launch_group_compact(illico_ctx *c, GroupCompactParams Q, int nb, int flags, bool pack) {
    constexpr int VEC = 16 / (int)sizeof(InT);
    const bool aligned = ((uintptr_t)Q.X % 16 == 0) && (Q.ld % VEC == 0) && (Q.col0 % VEC == 0);
    const bool lg = flags & ILLICO_FLAG_LOG1P;
    // few, long blocks (cluster-sized groups): a workgroup's chain of 64-row chunks is what the launch waits for -- tiles of 32 genes
    // (128-byte row pieces) put twice the workgroups on the same rows
    Q.blk_order = (c->no_compact_order || Q.nblk != c->pk_nblk) ? nullptr : c->d_pk_order; // (null unless the blocks' lengths differ much: set_groups)
    const bool narrow = !c->no_compact_narrow && c->pk_max_block_rows >= c->compact_narrow_rows && (long long)Q.nblk * ((nb + 63) / 64) < c->compact_narrow_wgs;
    const int tw = narrow ? 32 : 64;
    const dim3 grid(((Q.nseg + 7) & ~7) + Q.nblk, (nb + tw - 1) / tw);
    ProfScope ps(c, KID_GROUP_COMPACT);
#define GC_LAUNCH(V, L, K) do { if (narrow) hipLaunchKernelGGL((k_group_compact<InT, KeyT, V, L, K, 32>), grid, dim3(GCMP_NT), 0, c->stream, Q); \
                                else hipLaunchKernelGGL((k_group_compact<InT, KeyT, V, L, K, 64>), grid, dim3(GCMP_NT), 0, c->stream, Q); } while (0)
    if (pack) {
        if (aligned && !lg) GC_LAUNCH(true, false, true); else if (aligned) GC_LAUNCH(true, true, true);
        else if (!lg) GC_LAUNCH(false, false, true); else GC_LAUNCH(false, true, true);
    } else {
        if (aligned && !lg) GC_LAUNCH(true, false, false); else if (aligned) GC_LAUNCH(true, true, false);
        else if (!lg) GC_LAUNCH(false, false, false); else GC_LAUNCH(false, true, false);
    }
#undef GC_LAUNCH
    HIPCHK(c, hipGetLastError());
    return ILLICO_OK;
}

template <typename InT, typename KeyT>
static int run_ovo_packed(illico_ctx *c, const void *X, int64_t ld, int64_t col0, int nb, int N, KeyT *Xt, int64_t stride, int dtype, int flags,
                          long long *s2u, u64 *stie, double *ssum, std::vector<int> *redo /* genes of the batch the route left (groups above 1024 cells) */) {
    const int G = (int)c->n_groups, ref = (int)c->ref;
    const int64_t n_ref = c->h_counts[ref];
    const int nseg = gcmp_ref_segments(n_ref);
    int rc;
    void *v;
    if ((rc = get_scratch(c, "packed_nnz", (size_t)nb * G * 2 + (size_t)nb * nseg * 2 + 64, &v))) return rc;
    u16 *nnz = (u16 *)v;
    u16 *seg_nnz = nnz + (((size_t)nb * G + 7) & ~(size_t)7);
    if ((rc = get_scratch(c, "packed_seg_sum", (size_t)nb * nseg * 8 + (size_t)nb * 4 + (size_t)nb * G * 4, &v))) return rc;
    double *seg_sum = (double *)v;
    u32 *route = (u32 *)(seg_sum + (size_t)nb * nseg);
    u32 *gofs = route + nb;
    HIPCHK(c, hipMemsetAsync(route, 0, (size_t)nb * 4, c->stream));
    const int is_log1p = (flags & ILLICO_FLAG_LOG1P) ? 1 : 0;
    u32 *run_n = nullptr;
    {
        GroupCompactParams Q;
        Q.X = X; Q.ld = ld; Q.col0 = col0; Q.ncols = nb; Q.perm = c->d_perm; Q.pos_ptr = c->d_posptr; Q.G = G; Q.ref = ref; Q.nseg = nseg;
        Q.blk_g0 = c->d_pk_blk; Q.blk_g1 = c->d_pk_blk + c->pk_nblk; Q.blk_out = c->d_pk_blk + 2 * c->pk_nblk; Q.nblk = c->pk_nblk; Q.ref_out = c->pk_ref_out;
        Q.Xt = Xt; Q.xt_stride = stride; Q.nnz = nnz; Q.gofs = gofs; Q.blk_cnt = nullptr; Q.out_sum = ssum; Q.seg_nnz = seg_nnz; Q.seg_sum = seg_sum;
        Q.cand_of = nullptr; Q.run_n = nullptr; Q.n_cand = c->pk_nbig;
        if (c->pk_nbig > 0 && c->max_nonref >= 65535) { // a (gene, group) run can outgrow the 16-bit lengths: exact ones beside them
            if ((rc = get_scratch(c, "packed_run_n", (size_t)nb * c->pk_nbig * 4, &v))) return rc;
            run_n = (u32 *)v;
            Q.cand_of = c->d_pk_big + c->pk_nbig; Q.run_n = run_n;
        }
        if ((rc = launch_group_compact<InT, KeyT>(c, Q, nb, flags, true))) return rc;
    }
    bool parts = false;
    BigRunFn<KeyT> *big_fn = nullptr;
    u32 *run_cuts = nullptr;
    void *big_tmp = nullptr; // groups with more cells than k_bucket_big_runs' LDS slots: a second key buffer, into which such runs are dealt
    if (c->pk_nbig > 0) { // runs of more than 256 non-zero keys are dealt into value buckets in place: the rank kernel walks them in pieces
        if ((rc = get_scratch(c, "packed_big_fn", (size_t)nb * c->pk_nbig * sizeof(BigRunFn<KeyT>), &v))) return rc;
        big_fn = (BigRunFn<KeyT> *)v;
        ProfScope ps(c, KID_GROUP_COMPACT);
        int cap = (int)std::min<int64_t>(srt_cap<KeyT>(), (c->max_nonref + 63) & ~63ll); // (a run holds at most its group's cells)
        if (c->big_runs_cap > 0) cap = std::min(cap, std::max(c->big_runs_cap, 512) & ~63);
        if (c->max_nonref > cap && !c->no_big_runs_global && get_scratch(c, "packed_big_tmp", (size_t)nb * (size_t)stride * sizeof(KeyT), &v) == ILLICO_OK) big_tmp = v;
        if (c->max_nonref > OCR_COOP_MIN && !c->no_coop_runs) { // runs long enough for the rank kernel to walk them with all its wavefronts: the bucket kernels leave cuts
            if ((rc = get_scratch(c, "packed_run_cuts", (size_t)nb * c->pk_nbig * OCR_CUTS * 4, &v))) return rc;
            run_cuts = (u32 *)v;
        }
        if ((rc = launch_bucket_big_runs<KeyT>(c, (void *)Xt, big_tmp, (long long)stride, nnz, gofs, nb, G, cap, big_fn, route, c->max_nonref, run_n, run_cuts))) return rc;
    }
    {
        OvoCompactParams C;
        C.Xs = Xt; C.gene_stride = stride; C.counts = c->d_counts; C.nnz = nnz; C.gofs = gofs; C.ref_out = c->pk_ref_out; C.seg_nnz = seg_nnz; C.seg_sum = seg_sum;
        C.out_sum = ssum; C.G = G; C.ref = ref; C.n_genes = nb; C.nseg = nseg;
        packed_ref_sizing<KeyT>(n_ref, &C.ref_cap, &C.nbk_lg);
        if (c->packed_ref_cap > 0) C.ref_cap = std::min(C.ref_cap, std::max(c->packed_ref_cap, 1024));
        C.out_2u = s2u; C.out_tie = stie; C.route = route; C.big_sorted = c->pk_nbig > 0 ? 1 : 0;
        C.ref_by_gofs = 0; C.gene_flags = nullptr; C.big_fn = big_fn; C.big_tmp = big_tmp; C.run_cuts = run_cuts; C.run_n = run_n; C.cand_of = c->pk_nbig > 0 ? c->d_pk_big + c->pk_nbig : nullptr; C.n_cand = c->pk_nbig;
        // small problems per gene (a reference of at most 2048 cells, fewer than 128 groups, none above 256 cells): workgroups of 256 threads,
        // several per CU
        const bool eq0 = c->packed_eq_buckets >= 0 ? c->packed_eq_buckets != 0 : n_ref > 16384;
        const bool small_wg = n_ref <= 2048 && G < 128 && c->pk_nbig == 0 && !c->no_packed_small_wg && C.nbk_lg <= 16 && !eq0; // (such a reference never needs parts)
        const size_t lds = ocr_lds_bytes(C.ref_cap, C.nbk_lg, sizeof(KeyT), small_wg ? 256 : OCR_NT);
        // large references: the bucket function follows the reference's distribution (a crowded stretch of values would otherwise
        // fill buckets beyond three keys and send whole table words to key-by-key walks); "packed_eq_buckets" = 0 / 1 forces
        const bool eq = c->packed_eq_buckets >= 0 ? c->packed_eq_buckets != 0 : n_ref > 16384;
        // a reference with more cells than the kernel has key slots: its genes may need value-range parts (kernels_ovo_compact.h: PARTS)
        C.n_parts = packed_ref_parts<KeyT>(c, n_ref, C.ref_cap, C.nbk_lg);
        parts = C.n_parts > 1;
        C.needs_parts = nullptr;
        if (parts) { // every part adds its share: the statistics start from zero (the plain kernel, first, stores those of the genes that need no parts)
            if ((rc = get_scratch(c, "packed_needs_parts", (size_t)nb * 4, &v))) return rc;
            C.needs_parts = (u32 *)v;
            HIPCHK(c, hipMemsetAsync(C.needs_parts, 0, (size_t)nb * 4, c->stream));
            HIPCHK(c, hipMemsetAsync(s2u, 0, (size_t)nb * G * sizeof(long long), c->stream));
            HIPCHK(c, hipMemsetAsync(stie, 0, (size_t)nb * G * sizeof(u64), c->stream));
        }
        auto kern = small_wg ? k_ovo_rank_compact<KeyT, false, false, 256> : eq ? k_ovo_rank_compact<KeyT, true> : k_ovo_rank_compact<KeyT, false>;
        HIPCHK(c, hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        ProfScope ps(c, KID_OVO_RANK_COMPACT);
        hipLaunchKernelGGL(kern, dim3((unsigned)nb), dim3(small_wg ? 256 : OCR_NT), lds, c->stream, C);
        HIPCHK(c, hipGetLastError());
        if (parts && (rc = launch_rank_parts<KeyT>(c, C, nb, lds))) return rc;
    }
    // What the packed kernels left.  The plain kernel's genes (route word 1: a tie-heavy reference column, the reference's segments moved
    // together) go to k_ovo_rank over the packed layout when its LDS holds the reference and the groups (<= 1024 keys); everything else --
    // the PARTS kernel's genes (a reason in the word's high bits: their segments lie where they were), a run beyond every bucket kernel
    // (route 2), or sizes k_ovo_rank does not take -- is handed back to the caller: transposition + the general sort route.
    const bool sort_fits = packed_leftovers_fit_sort_route<KeyT>(c);
    if (parts || !sort_fits || c->pk_nbig > 0) { // (runs above 256 keys: a gene with a value bucket above 256 keys leaves the rank kernel late, its segments unmoved)
        std::vector<u32> hr((size_t)nb);
        HIPCHK(c, hipMemcpyAsync(hr.data(), route, (size_t)nb * 4, hipMemcpyDeviceToHost, c->stream));
        HIPCHK(c, hipStreamSynchronize(c->stream));
        bool any_sort = false;
        for (int j = 0; j < nb; ++j) {
            if (hr[j] && (!sort_fits || hr[j] > 255u || (hr[j] & 255u) == 2u)) redo->push_back(j);
            else if (hr[j]) any_sort = true;
        }
        if (c->debug_routes) { // why (PARTS: 1 = more parts than the launch has, 2 = a part beyond the slots, 3 = walks of overfull table words
            // would dominate, 5 = a run above 256 keys not dealt)
            int why[8] = {0, 0, 0, 0, 0, 0, 0, 0}, r2 = 0;
            for (int j = 0; j < nb; ++j) { if ((hr[j] & 255u) == 2u) ++r2; else if (hr[j]) ++why[(hr[j] >> 8) & 7u]; }
            fprintf(stderr, "[illico] packed OVO: %d genes, parts %d, left %zu to the general route (route 2: %d; reasons 0..5: %d %d %d %d %d %d)\n", nb, parts ? 1 : 0,
                    redo->size(), r2, why[0], why[1], why[2], why[3], why[4], why[5]);
        }
        if (!any_sort) return ILLICO_OK;
    }
    // the genes the packed kernel left (tie-heavy reference column, a group of more than 256 non-zeros): k_ovo_rank over the
    // packed layout; its workgroups return at once for every other gene
    OvoParams P;
    P.Xs = Xt; P.gene_stride = stride; P.pos_ptr = c->d_posptr; P.seg_ptr = nullptr; P.counts = c->d_counts;
    P.G = G; P.ref = ref; P.n_genes = nb; P.dt = dtype; P.is_log1p = is_log1p;
    P.ref_cap = 0; P.out_2u = s2u; P.out_tie = stie; P.out_sum = nullptr; P.nnz = nnz; P.gofs = gofs; P.only = route;
    return launch_ovo<KeyT>(c, P, n_ref, c->max_nonref, nullptr, nullptr, false);
}

template <typename InT, typename KeyT>
static int launch_transpose(illico_ctx *c, const void *X, int64_t ld, int64_t col0, int ncols, int N, KeyT *Xt, int64_t stride, u32 *flags,
                            int limit) { // flags[gene] != 0: a value that is no integer in [0, limit)
    ProfScope ps(c, KID_TRANSPOSE);
    dim3 grid((N + 63) / 64, (ncols + 63) / 64);
    constexpr int VEC = 16 / (int)sizeof(InT);
    const bool aligned = ((uintptr_t)X % 16 == 0) && (ld % VEC == 0) && (col0 % VEC == 0) && ((uintptr_t)Xt % 16 == 0) && (stride % 64 == 0);
    if (aligned)
        hipLaunchKernelGGL((k_transpose_permute_vec<InT, KeyT, VEC>), grid, dim3(256), 0, c->stream, (const InT *)X, (long long)ld,
                           (long long)col0, ncols, (const int *)c->d_perm, N, Xt, (long long)stride, flags, limit);
    else
        hipLaunchKernelGGL((k_transpose_permute<InT, KeyT>), grid, dim3(256), 0, c->stream, (const InT *)X, (long long)ld,
                           (long long)col0, ncols, (const int *)c->d_perm, N, Xt, (long long)stride, flags, limit);
    HIPCHK(c, hipGetLastError());
    return ILLICO_OK;
}


static int ensure_pinned(illico_ctx *c, size_t bytes) {
    if (c->pinned_bytes >= bytes) return ILLICO_OK;
    if (c->pinned) hipHostFree(c->pinned);
    c->pinned = nullptr;
    c->pinned_bytes = 0;
    HIPCHK(c, hipHostMalloc(&c->pinned, bytes + 4096, hipHostMallocDefault));
    c->pinned_bytes = bytes + 4096;
    return ILLICO_OK;
}
// Fused single-pass route over genes [b0, b0+nb): writes final planes for every gene it can take and sets
// h_flags[j] != 0 for the others (1 / 3: left to the two-pass routes; 2: done by the 256-value stage).  h_flags[nb] (also word nb of
// the deferred call's pinned flags) != 0: the 256-value stage was left to the host (k_wide_decide; only with max_gather > 0).
// init_flags (host, [nb]): the 256-value stage ALONE, for the genes marked 1 there (run_leftovers: a narrow matrix of gathered columns).
template <typename InT>
int run_fused_ovo(illico_ctx *c, const void *X, int64_t ld, int64_t b0, int nb, int flags, int alternative,
                  const OutPlanes &o, int64_t col_off, std::vector<u32> &h_flags, int defer_slot, bool probe,
                  int64_t max_gather, const u32 *init_flags) {
    constexpr int RT = FUSED_RT;
    const bool ovr = c->ref < 0;
    void *v;
    int rc;
    const size_t nb64 = ((size_t)nb + 63) & ~(size_t)63; // the cumulative tables are stored per 64-gene tile
    size_t bytes = nb64 * (RT + 1) * 4 + (size_t)nb * 8 * 2 + (size_t)nb * 4 + (size_t)nb * RT * 4 + 64;
    if ((rc = get_scratch(c, "fused_tables", bytes, &v))) return rc;
    FusedParams P;
    P.X = X; P.ld = ld; P.col0 = b0; P.ncols = nb; P.perm = c->d_perm; P.pos_ptr = c->d_posptr; P.counts = c->d_counts; P.gconst = c->d_gconst;
    P.G = (int)c->n_groups; P.ref = (int)c->ref;
    P.ref_TA = (u64 *)v;
    P.ref_sum = P.ref_TA + nb;
    P.ref_cum = (u32 *)(P.ref_sum + nb);
    P.gene_flags = P.ref_cum + nb64 * (RT + 1);
    P.hist_all = P.gene_flags + nb; // OVR: the column histograms; OVO: the reference group's
    P.group_hist = nullptr;
    P.wide_tiles = nullptr;
    P.wide_bad = nullptr;
    P.hist_off = nullptr;
    P.hist_words = nullptr;
    P.tie_mode = ovr ? (c->fused_tie_sparse ? 2 : 1) : 0; // (the reference's float64 tie accumulation: dense order, or a CSR window's sparse order)
    P.hist_full = c->ovr_full_dump ? 1 : 0;
    P.hist_total = (long long)c->hist_words;
    u32 *skipw = P.hist_all + (size_t)nb * RT; // (inside the 64 spare bytes of the allocation)
    P.wide_skip = skipw;
    const bool wide_only = init_flags != nullptr;
    P.n_cells = c->n_cells;
    P.rows_per_wg = (int)std::max<int64_t>(1024, (c->n_cells + 31) / 32);
    P.use_continuity = (flags & ILLICO_FLAG_CONTINUITY) ? 1 : 0;
    P.tie_correct = (flags & ILLICO_FLAG_TIE_CORRECT) ? 1 : 0;
    P.alternative = alternative;
    P.out_p = o.p + col_off; P.out_u = o.u + col_off; P.out_fc = o.fc + col_off; P.out_ld = o.ld;
    const int tiles = (nb + 63) / 64;
    int gpw = c->fused_groups_per_wg;
    if (gpw <= 0) { // 8 groups per workgroup (two per wavefront) measured best at C2 (4: +2 %, 16: +1 %, 32: +3 %: shorter
        // workgroups leave a shorter tail at the end of the launch); keep >= ~2048 workgroups on smaller problems
        // OVR (k_ovr_group_hists): a workgroup ends by adding its share of the column histogram to the global one -- same-process A/B
        // at C4 (tools/ab.py): 8 / 16 / 32 groups per workgroup 2.601 / 2.589 / 2.614 ms; 4: +36 %
        gpw = ovr ? 16 : 8;
        while (gpw > 4 && (int64_t)tiles * ((c->n_groups + gpw - 1) / gpw) < 2048) gpw >>= 1;
    }
    P.groups_per_wg = gpw = std::min(gpw, 128); // (k_ovr_group_hists packs a workgroup's cells into 16-bit fields: 128 x 255 < 2^16)
    HIPCHK(c, hipMemsetAsync(skipw, 0, 4, c->stream));
    if (wide_only) HIPCHK(c, hipMemcpyAsync(P.gene_flags, init_flags, (size_t)nb * 4, hipMemcpyHostToDevice, c->stream));
    else HIPCHK(c, hipMemsetAsync(P.gene_flags, 0, (size_t)nb * 4, c->stream));
    // few, large groups (clusters of an atlas): the (group, gene) value histograms first, rows split over as many wavefronts as the launch
    // needs, then the same integers from the histograms (kernels_group_hists.h) -- the fused kernels give a wavefront one GROUP at a time
    // ... and OVR with a group beyond the 16-bit cells of the one-pass form (an atlas whose control group has 66 667 cells): the histograms
    // here are 32 bits wide, one read of X instead of the two-pass form's two (2 000 000 x 1200 x 2000 groups: 6.9 ms)
    const size_t gh_bytes = (size_t)c->n_groups * (size_t)((nb + 63) / 64) * RT * 64 * 4;
    const bool gh_few = (int64_t)((nb + 63) / 64) * ((c->n_groups + 3) / 4) < c->group_hist_max_wgs && gh_bytes <= ((size_t)256 << 20);
    // (OVO with a ranked group beyond 65535 cells: the fused kernel's 32-bit multiplicities take 82 KB of LDS -- one workgroup, four wavefronts, per CU)
    const bool gh_ovr_big = c->max_nonref > 65535 && gh_bytes <= ((size_t)1 << 30);
    // ... and groups of very different sizes (clusters from fifty to tens of thousands of cells): the largest group alone is more than twice
    // an average wavefront's share of the fused launch -- its wavefront is what that launch waits for (100 000 cells x 8192 genes x 30
    // clusters: 0.84 ms with equal groups, 1.56 with a Dirichlet draw of sizes)
    const bool gh_ragged = gh_bytes <= ((size_t)256 << 20) && c->max_nonref > 2 * (c->n_cells * (int64_t)((nb + 63) / 64) / 4096) && c->max_nonref >= 4096;
    const bool hist_route = !wide_only && !c->no_group_hist_route && (gh_few || gh_ovr_big || gh_ragged) && c->n_cells >= c->group_hist_min_cells && c->n_cells <= (1ll << 21);
    if ((probe || hist_route) && !wide_only) { // OVR on device-resident input: which genes are count-valued at all is found on the device (the OVO pass has
        // k_fused_ref, which reads every reference row first)
        ProfScope ps(c, KID_FUSED_REF);
        hipLaunchKernelGGL((k_fused_probe<InT, RT>), dim3(tiles), dim3(FUSED_PROBE_NT), 0, c->stream, P);
        HIPCHK(c, hipGetLastError());
    }
    const dim3 main_grid(tiles, ((int)c->n_groups + P.groups_per_wg - 1) / P.groups_per_wg);
    const size_t lds8 = fused_main_lds_bytes<RT, false, 8>(), lds16 = fused_main_lds_bytes<RT, false, 16>(), lds_ovr = fused_main_lds_bytes<RT, true, 16>();
    (void)lds8; (void)lds16; (void)lds_ovr;
    // is the 256-value stage left to the host?  (decided on the device, after the first pass: nothing waits for it here)
    auto wide_decide = [&]() -> int {
        if (wide_only || max_gather <= 0 || c->no_wide_gather) return ILLICO_OK;
        ProfScope ps(c, KID_FUSED_REF);
        hipLaunchKernelGGL(k_wide_decide, dim3(1), dim3(1024), 0, c->stream, (const u32 *)P.gene_flags, nb, (int)std::min<int64_t>(max_gather, 0x7FFFFFFF), skipw);
        HIPCHK(c, hipGetLastError());
        return ILLICO_OK;
    };
    if (hist_route) {
        const size_t h_bytes = (size_t)c->n_groups * tiles * RT * 64 * 4;
        if ((rc = get_scratch(c, "group_value_hists", h_bytes, &v))) return rc;
        u32 *H = (u32 *)v;
        HIPCHK(c, hipMemsetAsync(H, 0, h_bytes, c->stream));
        constexpr int NWH = GH_NT / 64;
        // positions per wavefront: ~2048 workgroups, at most 4096 positions (16-bit cells)
        int wave_rows = (int)std::min<int64_t>(4096, std::max<int64_t>(256, (c->n_cells * tiles / (2048 * NWH) + 31) & ~31ll));
        const int chunks = (int)((c->n_cells + (int64_t)NWH * wave_rows - 1) / ((int64_t)NWH * wave_rows));
        {
            ProfScope ps(c, KID_GROUP_HISTS);
            hipLaunchKernelGGL((k_group_value_hists<InT, RT>), dim3(tiles, chunks), dim3(GH_NT), 0, c->stream, P, H, wave_rows);
            HIPCHK(c, hipGetLastError());
        }
        ProfScope ps(c, KID_FUSED_REF);
        if (ovr) {
            HIPCHK(c, hipMemsetAsync(P.hist_all, 0, (size_t)nb * RT * 4, c->stream));
            hipLaunchKernelGGL((k_group_hists_to_column<RT, true>), dim3(tiles, ((int)c->n_groups + 63) / 64), dim3(256), 0, c->stream, P, (const u32 *)H);
        } else hipLaunchKernelGGL((k_group_hists_to_column<RT, false>), dim3(tiles), dim3(256), 0, c->stream, P, (const u32 *)H);
        hipLaunchKernelGGL((k_fused_tables_all<RT>), dim3((nb + 255) / 256), dim3(256), 0, c->stream, P);
        const dim3 ge(tiles, ((int)c->n_groups + 3) / 4);
        if (ovr) hipLaunchKernelGGL((k_emit_from_group_hists<RT, true>), ge, dim3(256), 0, c->stream, P, (const u32 *)H);
        else hipLaunchKernelGGL((k_emit_from_group_hists<RT, false>), ge, dim3(256), 0, c->stream, P, (const u32 *)H);
        HIPCHK(c, hipGetLastError());
    } else if (!ovr) {
        if (wide_only) {
        } else if (tiles >= 100) { // one 1024-thread workgroup per tile builds the tables (C2: 125 tiles, 0.074 ms)
            ProfScope ps(c, KID_FUSED_REF);
            auto kern = k_fused_ref<InT, RT>;
            hipLaunchKernelGGL(kern, dim3(tiles), dim3(FUSED_REF_NT), fused_ref_lds_bytes(RT), c->stream, P);
            HIPCHK(c, hipGetLastError());
        } else { // few tiles (a C5 shard: 59): the reference rows split over (tiles, row chunks), then one thread per gene for
            // the tables -- 0.20 -> 0.11 ms there, 0.074 -> 0.083 ms at C2, hence the switch
            ProfScope ps(c, KID_FUSED_REF);
            HIPCHK(c, hipMemsetAsync(P.hist_all, 0, (size_t)nb * RT * 4, c->stream));
            const int64_t n_ref = c->h_counts[c->ref];
            const int want_chunks = std::max(1, 768 / std::max(tiles, 1)); // enough workgroups for 256 CUs, few enough flushes
            P.rows_per_wg = (int)std::max<int64_t>(FUSED_REF_ROWS, (n_ref + want_chunks - 1) / want_chunks);
            const int chunks = (int)std::max<int64_t>(1, (n_ref + P.rows_per_wg - 1) / P.rows_per_wg);
            hipLaunchKernelGGL((k_fused_ref_hist<InT, RT>), dim3(tiles, chunks), dim3(FUSED_NT), 0, c->stream, P);
            hipLaunchKernelGGL((k_fused_tables_all<RT>), dim3((nb + 255) / 256), dim3(256), 0, c->stream, P);
            HIPCHK(c, hipGetLastError());
        }
        if (!wide_only) {
            ProfScope ps(c, KID_OVO_FUSED);
            if (c->max_nonref <= 255) // 8-bit running multiplicities: 34 KB of LDS per workgroup instead of 50 KB
                hipLaunchKernelGGL((k_ovo_fused<InT, RT, false, 8>), main_grid, dim3(FUSED_NT), lds8, c->stream, P);
            else if (c->max_nonref <= 65535) hipLaunchKernelGGL((k_ovo_fused<InT, RT, false, 16>), main_grid, dim3(FUSED_NT), lds16, c->stream, P);
            else { // clusters of more than 65535 cells: 32-bit multiplicities (82 KB: one workgroup per CU)
                auto kern = k_ovo_fused<InT, RT, false, 32>;
                const size_t lds32 = fused_main_lds_bytes<RT, false, 32>();
                HIPCHK(c, hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds32));
                hipLaunchKernelGGL(kern, main_grid, dim3(FUSED_NT), lds32, c->stream, P);
            }
            HIPCHK(c, hipGetLastError());
        }
        // Second pass, 256-value tables, over the tiles that hold genes the first pass flagged (counts of 64 .. 255: highly
        // expressed genes of real count matrices): same kernels, one workgroup per CU (130 KB of LDS), resident workgroups
        // working through the list of such tiles that k_fused_ref<WIDE> builds on the device -- an empty list costs two
        // near-empty launches (0.005 ms at C2).  Flags after it: 1 = the host's two-pass routes, 0 / 2 = done.
        if (c->max_nonref <= 255 && !c->no_fused_wide) {
            if ((rc = wide_decide())) return rc;
            constexpr int WRT = FUSED_WIDE_RT;
            const size_t wbytes = nb64 * (WRT + 1) * 4 + (size_t)nb * 8 * 2 + (size_t)(tiles + 1) * 4 + 64;
            if ((rc = get_scratch(c, "fused_tables_wide", wbytes, &v))) return rc;
            FusedParams Q = P;
            Q.ref_TA = (u64 *)v;
            Q.ref_sum = Q.ref_TA + nb;
            Q.ref_cum = (u32 *)(Q.ref_sum + nb);
            Q.wide_tiles = Q.ref_cum + nb64 * (WRT + 1);
            HIPCHK(c, hipMemsetAsync(Q.wide_tiles, 0, 4, c->stream));
            {
                ProfScope ps(c, KID_FUSED_REF);
                auto kern = k_fused_ref<InT, WRT, true>;
                HIPCHK(c, hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)fused_ref_lds_bytes(WRT)));
                hipLaunchKernelGGL(kern, dim3(tiles), dim3(FUSED_REF_NT), fused_ref_lds_bytes(WRT), c->stream, Q);
                HIPCHK(c, hipGetLastError());
            }
            ProfScope ps(c, KID_OVO_FUSED_WIDE);
            auto kern = k_ovo_fused<InT, WRT, false, 8, FUSED_U, true>;
            const size_t lds = fused_main_lds_bytes<WRT, false, 8>();
            HIPCHK(c, hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
            int n_cu = 256;
            hipDeviceGetAttribute(&n_cu, hipDeviceAttributeMultiprocessorCount, c->device);
            hipLaunchKernelGGL(kern, dim3((unsigned)std::max(n_cu, 1)), dim3(FUSED_NT), lds, c->stream, Q); // resident workgroups over the listed tiles
            HIPCHK(c, hipGetLastError());
        }
    } else if (!wide_only) {
        HIPCHK(c, hipMemsetAsync(P.hist_all, 0, (size_t)nb * RT * 4, c->stream));
        // one pass over X when the per-(group, gene) histograms fit the scratch cap (64 or 128 bytes each)
        // 8-bit cells when no group is larger than 255 cells, else the width per group (0): a few large groups do not double
        // the histogram bytes of all the small ones
        const int cbits = c->max_nonref <= 255 ? 8 : 0;
        const size_t hist_bytes = cbits ? (size_t)c->n_groups * tiles * (RT * cbits / 32) * 64 * 4 : (size_t)c->hist_words * tiles * 64 * 4;
        if (!c->no_ovr_one_pass && c->max_nonref <= 65535 && hist_bytes <= (size_t)c->scratch_bytes) { // (16-bit cells: no group beyond 65535 cells)
            if ((rc = get_scratch(c, "group_hist", hist_bytes, &v))) return rc;
            P.group_hist = (u32 *)v;
            if ((rc = get_scratch(c, "group_hist_words", (size_t)c->n_groups * tiles, &v))) return rc;
            P.hist_words = (unsigned char *)v;
            P.hist_off = c->d_hist_off;
            {
                ProfScope ps(c, KID_OVR_FUSED);
                if (cbits == 8) hipLaunchKernelGGL((k_ovr_group_hists<InT, RT, 8>), main_grid, dim3(FUSED_NT), 0, c->stream, P);
                else hipLaunchKernelGGL((k_ovr_group_hists<InT, RT, 0>), main_grid, dim3(FUSED_NT), 0, c->stream, P);
                HIPCHK(c, hipGetLastError());
            }
            ProfScope ps(c, KID_FUSED_REF);
            hipLaunchKernelGGL((k_fused_tables_all<RT>), dim3((nb + 255) / 256), dim3(256), 0, c->stream, P);
            // the rank-sum kernel keeps a 64-entry table per lane in registers: more groups per workgroup amortise its fill
            FusedParams P2 = P;
            P2.groups_per_wg = c->ovr_hist_groups_per_wg > 0 ? c->ovr_hist_groups_per_wg : 32;
            while (P2.groups_per_wg > 8 && (int64_t)tiles * ((c->n_groups + P2.groups_per_wg - 1) / P2.groups_per_wg) < 2048) P2.groups_per_wg >>= 1;
            const dim3 grid2(tiles, ((int)c->n_groups + P2.groups_per_wg - 1) / P2.groups_per_wg);
            const bool np3 = c->n_cells < (1ll << 23); // s < 2^24: three byte planes
            if (cbits == 8 && np3) hipLaunchKernelGGL((k_ovr_from_hists<RT, 8, 3>), grid2, dim3(FUSED_NT), 0, c->stream, P2);
            else if (cbits == 8) hipLaunchKernelGGL((k_ovr_from_hists<RT, 8, 4>), grid2, dim3(FUSED_NT), 0, c->stream, P2);
            else if (np3) hipLaunchKernelGGL((k_ovr_from_hists<RT, 0, 3>), grid2, dim3(FUSED_NT), 0, c->stream, P2);
            else hipLaunchKernelGGL((k_ovr_from_hists<RT, 0, 4>), grid2, dim3(FUSED_NT), 0, c->stream, P2);
            HIPCHK(c, hipGetLastError());
        } else {
            {
                ProfScope ps(c, KID_FUSED_REF);
                const int chunks = (int)((c->n_cells + P.rows_per_wg - 1) / P.rows_per_wg);
                hipLaunchKernelGGL((k_fused_hist_all<InT, RT>), dim3(tiles, chunks), dim3(FUSED_NT), fused_ref_lds_bytes(RT), c->stream, P);
                hipLaunchKernelGGL((k_fused_tables_all<RT>), dim3((nb + 255) / 256), dim3(256), 0, c->stream, P);
                HIPCHK(c, hipGetLastError());
            }
            ProfScope ps(c, KID_OVR_FUSED);
            hipLaunchKernelGGL((k_ovo_fused<InT, RT, true, 16>), main_grid, dim3(FUSED_NT), lds_ovr, c->stream, P);
            HIPCHK(c, hipGetLastError());
        }
    }
    // OVR second stage, 256-value tables, over the tiles that hold genes the 64-value pass flagged (counts of 64 .. 255): the
    // two-pass form -- column histograms of those tiles (k_fused_hist_all<WIDE>: every row, so a candidate is known to fit),
    // tables, then k_ovo_fused<OVR, WIDE> with resident workgroups over the listed tiles.  No per-group state: 67 KB of LDS.
    // C4 shape with gene means up to 40: 97 ms (those genes through the general sort route) -> see DESIGN.md.
    if (ovr && !c->no_fused_wide) {
        if ((rc = wide_decide())) return rc;
        constexpr int WRT = FUSED_WIDE_RT;
        const size_t wbytes = nb64 * (WRT + 1) * 4 + (size_t)nb * 8 * 2 + (size_t)nb * WRT * 4 + (size_t)(nb + tiles) * 4 + (size_t)(tiles + 1) * 4 + 64;
        if ((rc = get_scratch(c, "fused_tables_wide", wbytes, &v))) return rc;
        FusedParams Q = P;
        Q.ref_TA = (u64 *)v;
        Q.ref_sum = Q.ref_TA + nb;
        Q.ref_cum = (u32 *)(Q.ref_sum + nb);
        Q.hist_all = Q.ref_cum + nb64 * (WRT + 1);
        Q.wide_bad = Q.hist_all + (size_t)nb * WRT;          // [nb] + [tiles] tile marks
        Q.wide_tiles = Q.wide_bad + nb + tiles;
        HIPCHK(c, hipMemsetAsync(Q.hist_all, 0, ((size_t)nb * WRT + nb + tiles + 1) * 4, c->stream));
        {
            ProfScope ps(c, KID_FUSED_REF);
            const int chunks = (int)((c->n_cells + P.rows_per_wg - 1) / P.rows_per_wg);
            auto kh = k_fused_hist_all<InT, WRT, true>;
            HIPCHK(c, hipFuncSetAttribute((const void *)kh, hipFuncAttributeMaxDynamicSharedMemorySize, (int)fused_ref_lds_bytes(WRT)));
            hipLaunchKernelGGL(kh, dim3(tiles, chunks), dim3(FUSED_NT), fused_ref_lds_bytes(WRT), c->stream, Q);
            hipLaunchKernelGGL((k_fused_tables_all<WRT, true>), dim3((nb + 255) / 256), dim3(256), 0, c->stream, Q);
            HIPCHK(c, hipGetLastError());
        }
        ProfScope ps(c, KID_OVO_FUSED_WIDE);
        auto kern = k_ovo_fused<InT, WRT, true, 16, FUSED_U, true>;
        const size_t lds = fused_main_lds_bytes<WRT, true, 16>();
        HIPCHK(c, hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        int n_cu = 256;
        hipDeviceGetAttribute(&n_cu, hipDeviceAttributeMultiprocessorCount, c->device);
        hipLaunchKernelGGL(kern, dim3((unsigned)std::max(2 * n_cu, 1)), dim3(FUSED_NT), lds, c->stream, Q); // resident workgroups over the listed tiles
        HIPCHK(c, hipGetLastError());
    }
    // route flags back through a pinned staging buffer (a pageable destination makes the copy a blocking, staged one)
    if (defer_slot >= 0) { // deferred: the copy is enqueued, an event marks it, nobody waits here (resolve_pending does)
        void *&pin = c->pend_pinned[defer_slot];
        if (c->pend_pinned_bytes[defer_slot] < (size_t)nb * 4 + 4) {
            if (pin) hipHostFree(pin);
            pin = nullptr;
            c->pend_pinned_bytes[defer_slot] = 0;
            HIPCHK(c, hipHostMalloc(&pin, (size_t)nb * 4 + 4096, hipHostMallocDefault));
            c->pend_pinned_bytes[defer_slot] = (size_t)nb * 4 + 4096;
        }
        if (!c->pend_event[defer_slot]) HIPCHK(c, hipEventCreateWithFlags(&c->pend_event[defer_slot], hipEventDisableTiming));
        HIPCHK(c, hipMemcpyAsync(pin, P.gene_flags, (size_t)nb * 4, hipMemcpyDeviceToHost, c->stream));
        HIPCHK(c, hipMemcpyAsync((u32 *)pin + nb, skipw, 4, hipMemcpyDeviceToHost, c->stream));
        HIPCHK(c, hipEventRecord(c->pend_event[defer_slot], c->stream));
        return ILLICO_OK;
    }
    if ((rc = ensure_pinned(c, (size_t)nb * 4 + 4))) return rc;
    HIPCHK(c, hipMemcpyAsync(c->pinned, P.gene_flags, (size_t)nb * 4, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipMemcpyAsync((u32 *)c->pinned + nb, skipw, 4, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    h_flags.assign((const u32 *)c->pinned, (const u32 *)c->pinned + nb + 1);
    return ILLICO_OK;
}
// Of 64k evenly spaced cells of a HOST matrix window: is it count-valued at all?  (The fused route over a host matrix copies
// the window up; on normalised data that copy would be made twice, once for nothing.)
template <typename InT> static bool host_window_is_count_valued(const InT *X, int64_t ld, int64_t col_lb, int64_t N, int64_t W, bool *light_tails = nullptr) {
    const int64_t n_samples = std::min<int64_t>(N * W, 1 << 16);
    int64_t bad = 0, big = 0;
    for (int64_t i = 0; i < n_samples; ++i) {
        const int64_t k = (int64_t)((double)i * (double)(N * W) / (double)n_samples);
        const int64_t r = k / W, j = k - r * W;
        const InT v = X[r * ld + col_lb + j];
        if (!(v >= (InT)0 && v < (InT)(1 << 24) && (InT)(int)v == v)) ++bad;
        else if (v >= (InT)255) ++big;
    }
    // light tails: (nearly) no sampled cell of 255 or more -- the byte windows of host_windows_pipeline_narrow then hold (nearly) every
    // gene; a heavy-tailed count matrix keeps the float32 windows, whose flagged genes are gathered on the device instead of going up again
    if (light_tails) *light_tails = (double)(bad + big) * 2048.0 <= (double)n_samples;
    return (double)bad <= 0.02 * (double)n_samples;
}
// ---- host-resident dense input: a three-stage pipeline over column windows ----------------------------------------------
// A pageable 2-D copy of the whole window (what this path did before) moves 9.6 GB at ~43 GB/s through the driver's own
// staging and nothing overlaps it.  Here: (1) HS_THREADS host threads copy window k + 1's row pieces out of the caller's
// matrix into a PINNED slot, (2) the copy stream moves window k's slot to the device at the link's rate, (3) the context's
// stream runs the fused pass on window k - 1 -- all three at once, three slots deep.  Slot j serves the windows k = j mod 3: its
// pinned half is free once its upload has completed, its device half once the pass over it has (events both ways).
#define HS_THREADS 12
#define HS_THREADS_NARROW 16 // (the byte pipeline: the fill -- 9.6 GB of host reads at C2 -- is what must keep up with a quarter-size upload)
struct HostLeftovers { // the flagged genes' columns, gathered on the device while their window is still there
    void *xl = nullptr;    // [N][cap] values
    int64_t cap = 0, n = 0;
    int *d_dst = nullptr;  // [n] output column (relative to the call's planes) of gathered column j
};
template <typename InT>
static int host_windows_pipeline(illico_ctx *c, const InT *X, int64_t ld, int64_t N, int64_t col_lb, int64_t col_ub, int flags, int alternative,
                                 const OutPlanes &o, std::vector<std::pair<int64_t, int64_t>> &runs, HostLeftovers &left) {
    int rc;
    void *v;
    // windows of ~256 MB (a multiple of 64 genes): long enough for the link's rate, short enough that the first pass starts early
    int64_t wmax = std::max<int64_t>(64, (int64_t)(((size_t)256 << 20) / ((size_t)N * sizeof(InT))) & ~63ll);
    wmax = std::min<int64_t>(wmax, std::max<int64_t>(64, (int64_t)((size_t)c->scratch_bytes / HS_SLOTS / ((size_t)N * sizeof(InT))) & ~63ll));
    if (c->gene_batch > 0) wmax = std::min<int64_t>(wmax, std::max<int64_t>(1, c->gene_batch));
    const int64_t n_win = (col_ub - col_lb + wmax - 1) / wmax;
    const size_t slot_bytes = (size_t)wmax * (size_t)N * sizeof(InT);
    HostStage *hs = host_stage_of(c);
    if (!hs->copy) HIPCHK(c, hipStreamCreateWithFlags(&hs->copy, hipStreamNonBlocking));
    for (int j = 0; j < HS_SLOTS; ++j) {
        if (!hs->up[j]) HIPCHK(c, hipEventCreateWithFlags(&hs->up[j], hipEventDisableTiming));
        if (!hs->done[j]) HIPCHK(c, hipEventCreateWithFlags(&hs->done[j], hipEventDisableTiming));
    }
    if (hs->pin_bytes < slot_bytes) {
        for (int j = 0; j < HS_SLOTS; ++j) { if (hs->pin[j]) hipHostFree(hs->pin[j]); hs->pin[j] = nullptr; }
        hs->pin_bytes = 0;
        for (int j = 0; j < HS_SLOTS; ++j) HIPCHK(c, hipHostMalloc(&hs->pin[j], slot_bytes, hipHostMallocDefault));
        hs->pin_bytes = slot_bytes;
    }
    // room for the genes the fused pass flags (a count matrix: few): they are gathered out of their window while it is on the device,
    // so that no window travels twice
    left.cap = c->no_leftover_gather ? 0 : std::min<int64_t>(((col_ub - col_lb) / 4 + 63) & ~63ll, (int64_t)((size_t)c->scratch_bytes / 4 / ((size_t)N * sizeof(InT))) & ~63ll);
    int *d_src = nullptr;
    if (left.cap >= 64) {
        if ((rc = get_scratch(c, "xleft", (size_t)N * (size_t)left.cap * sizeof(InT), &v))) return rc;
        left.xl = v;
        HIPCHK(c, hipMemsetAsync(left.xl, 0, (size_t)N * (size_t)left.cap * sizeof(InT), c->stream));
        if ((rc = get_scratch(c, "xleft_cols", (size_t)left.cap * 8, &v))) return rc;
        d_src = (int *)v; left.d_dst = d_src + left.cap;
        if (hs->lists_ints < (size_t)left.cap * 2) {
            if (hs->lists) hipHostFree(hs->lists);
            hs->lists = nullptr; hs->lists_ints = 0;
            HIPCHK(c, hipHostMalloc((void **)&hs->lists, (size_t)left.cap * 8, hipHostMallocDefault));
            hs->lists_ints = (size_t)left.cap * 2;
        }
    } else left.cap = 0;
    InT *dev[HS_SLOTS];
    static const char *names[HS_SLOTS] = {"xin0", "xin1", "xin2"};
    for (int j = 0; j < HS_SLOTS; ++j) {
        if ((rc = get_scratch(c, names[j], slot_bytes, &v))) return rc;
        dev[j] = (InT *)v;
    }
    // producer: fills and uploads the slots; the calling thread consumes them.  `ready` = windows whose upload is enqueued.
    std::mutex mu;
    std::condition_variable cv;
    int64_t ready = 0, consumed = 0;
    int err = 0; // hipError_t of the producer, if any
    double t_fill = 0.0, t_wait = 0.0; // (ILLICO_HS_DEBUG=1 prints them: seconds the producer spent filling slots / the consumer waiting for one)
    const int device = c->device;
    hipStream_t compute = c->stream;
    const int x_node = c->no_host_numa ? -1 : numa_node_of_buffer(X, (size_t)N * (size_t)ld * sizeof(InT));
    std::thread producer([&] {
        hipSetDevice(device);
        numa_confine_this_thread(x_node); // (the fill threads started below inherit the mask; this thread ends with the call)
        const int T = (int)std::max<int64_t>(1, std::min<int64_t>(c->host_fill_threads > 0 ? c->host_fill_threads : HS_THREADS, N / 4096 + 1));
        for (int64_t k = 0; k < n_win; ++k) {
            const int j = (int)(k % HS_SLOTS);
            const int64_t w0 = col_lb + k * wmax, wn = std::min<int64_t>(wmax, col_ub - w0);
            if (k >= HS_SLOTS) { // slot j still belongs to window k - HS_SLOTS until the pass over it is done
                std::unique_lock<std::mutex> lk(mu);
                cv.wait(lk, [&] { return consumed > k - HS_SLOTS || err; });
                if (err) return;
                lk.unlock();
                if (hipEventSynchronize(hs->done[j]) != hipSuccess) { std::lock_guard<std::mutex> g(mu); err = 1; cv.notify_all(); return; }
            }
            InT *dst = (InT *)hs->pin[j];
            const size_t piece = (size_t)wn * sizeof(InT);
            const auto t_a = std::chrono::steady_clock::now();
            std::vector<std::thread> pool;
            auto rows = [=](int64_t r0, int64_t r1) { // (the piece eight rows ahead is prefetched by hand: see host_windows_pipeline_narrow)
                for (int64_t r = r0; r < r1; ++r) {
                    if (r + 8 < r1) {
                        const char *pf = (const char *)(X + (size_t)(r + 8) * ld + w0);
                        for (size_t b = 0; b < piece; b += 64) __builtin_prefetch(pf + b, 0, 1);
                    }
                    memcpy(dst + (size_t)r * wn, X + (size_t)r * ld + w0, piece);
                }
            };
            for (int t = 1; t < T; ++t) pool.emplace_back(rows, N * t / T, N * (t + 1) / T);
            rows(0, N / T);
            for (auto &th : pool) th.join();
            t_fill += std::chrono::duration<double>(std::chrono::steady_clock::now() - t_a).count();
            hipError_t e = hipMemcpyAsync(dev[j], dst, piece * (size_t)N, hipMemcpyHostToDevice, hs->copy);
            if (e == hipSuccess) e = hipEventRecord(hs->up[j], hs->copy);
            std::lock_guard<std::mutex> g(mu);
            if (e != hipSuccess) err = 1;
            ready = k + 1;
            cv.notify_all();
            if (err) return;
        }
    });
    std::vector<u32> hf;
    rc = ILLICO_OK;
    for (int64_t k = 0; k < n_win && !rc; ++k) {
        const int j = (int)(k % HS_SLOTS);
        const int64_t w0 = col_lb + k * wmax, wn = std::min<int64_t>(wmax, col_ub - w0);
        {
            const auto t_a = std::chrono::steady_clock::now();
            std::unique_lock<std::mutex> lk(mu);
            cv.wait(lk, [&] { return ready > k || err; });
            t_wait += std::chrono::duration<double>(std::chrono::steady_clock::now() - t_a).count();
            if (err) { rc = fail(c, ILLICO_ERR_HIP, "staging a host window failed"); break; }
        }
        if (hipStreamWaitEvent(compute, hs->up[j], 0) != hipSuccess) { rc = fail(c, ILLICO_ERR_HIP, "hipStreamWaitEvent failed"); break; }
        rc = run_fused_ovo<InT>(c, dev[j], wn, 0, (int)wn, flags, alternative, o, w0 - col_lb, hf);
        if (!rc) { // this window's flagged genes: out of the device window into the leftover matrix (else: column runs, uploaded again later)
            int cnt = 0;
            for (int64_t q = 0; q < wn; ++q) cnt += (hf[q] == 1u || hf[q] == 3u) ? 1 : 0;
            if (cnt && left.n + cnt <= left.cap) {
                int *ls = hs->lists + left.n, *ld_ = hs->lists + left.cap + left.n;
                int e = 0;
                for (int64_t q = 0; q < wn; ++q)
                    if (hf[q] == 1u || hf[q] == 3u) { ls[e] = (int)q; ld_[e] = (int)(w0 - col_lb + q); ++e; }
                if (hipMemcpyAsync(d_src + left.n, ls, (size_t)cnt * 4, hipMemcpyHostToDevice, compute) != hipSuccess ||
                    hipMemcpyAsync(left.d_dst + left.n, ld_, (size_t)cnt * 4, hipMemcpyHostToDevice, compute) != hipSuccess)
                    rc = fail(c, ILLICO_ERR_HIP, "uploading a column list failed");
                if (!rc) {
                    ProfScope ps(c, KID_GATHER_COLS);
                    hipLaunchKernelGGL((k_gather_columns<InT>), dim3((unsigned)((N + 63) / 64)), dim3(256), 0, compute, (const InT *)dev[j], (long long)wn,
                                       (int)N, (const int *)(d_src + left.n), cnt, cnt, (InT *)left.xl, (long long)left.cap, (long long)left.n);
                    if (hipGetLastError() != hipSuccess) rc = fail(c, ILLICO_ERR_HIP, "k_gather_columns launch failed");
                }
                left.n += cnt;
            } else if (cnt) flagged_runs(hf.data(), wn, w0, runs);
        }
        if (!rc && hipEventRecord(hs->done[j], compute) != hipSuccess) rc = fail(c, ILLICO_ERR_HIP, "hipEventRecord failed");
        c->h2d_input_bytes += (int64_t)((size_t)wn * sizeof(InT) * (size_t)N);
        std::lock_guard<std::mutex> g(mu);
        consumed = k + 1;
        if (rc) err = 1;
        cv.notify_all();
    }
    {
        std::lock_guard<std::mutex> g(mu);
        if (rc) err = 1;
        consumed = n_win + HS_SLOTS;
        cv.notify_all();
    }
    producer.join();
    hipStreamSynchronize(hs->copy);
    if (getenv("ILLICO_HS_DEBUG"))
        fprintf(stderr, "[illico] host windows: %lld x %lld genes, slot fill %.1f ms, consumer waited %.1f ms for uploads\n", (long long)n_win,
                (long long)wmax, t_fill * 1e3, t_wait * 1e3);
    return rc;
}

// ---- the same pipeline with BYTE windows (host_narrow.h): a count matrix in host memory goes up as a quarter of its float32 bytes ----
// The threads that fill a pinned slot convert as they copy (cell = the value when it is an integer in [0, 255), else 255); the copy
// stream moves N x wn bytes; the context's stream runs the fused kernels on the byte window -- k_ovo_fused<uint8_t> /
// k_ovr_group_hists<uint8_t>, the forms count-valued CSR windows take, 64-value pass and 256-value second pass alike.  A gene they
// flag (a 255 cell: a value of 255 or more, a fraction, a negative) comes back as a column run and goes up again, in its own type,
// through the two-pass routes (run_dense_twopass on the host matrix): few genes on count data.  Windows are four times as wide as the
// float32 pipeline's for the same pinned memory: each row piece is a longer contiguous read of the caller's matrix.
template <typename InT>
static int host_windows_pipeline_narrow(illico_ctx *c, const InT *X, int64_t ld, int64_t N, int64_t col_lb, int64_t col_ub, int flags, int alternative,
                                        const OutPlanes &o, std::vector<std::pair<int64_t, int64_t>> &runs) {
    int rc;
    void *v;
    int64_t wmax = std::max<int64_t>(64, (int64_t)(((size_t)256 << 20) / (size_t)N) & ~63ll);
    wmax = std::min<int64_t>(wmax, std::max<int64_t>(64, (int64_t)((size_t)c->scratch_bytes / HS_SLOTS / (size_t)N) & ~63ll));
    if (c->gene_batch > 0) wmax = std::min<int64_t>(wmax, std::max<int64_t>(64, (c->gene_batch + 63) & ~63ll));
    const int64_t n_win = (col_ub - col_lb + wmax - 1) / wmax;
    const size_t slot_bytes = (size_t)wmax * (size_t)N; // (wmax is a multiple of 64: every window's row pitch, its width rounded up to 64, fits)
    HostStage *hs = host_stage_of(c);
    if (!hs->copy) HIPCHK(c, hipStreamCreateWithFlags(&hs->copy, hipStreamNonBlocking));
    for (int j = 0; j < HS_SLOTS; ++j) {
        if (!hs->up[j]) HIPCHK(c, hipEventCreateWithFlags(&hs->up[j], hipEventDisableTiming));
        if (!hs->done[j]) HIPCHK(c, hipEventCreateWithFlags(&hs->done[j], hipEventDisableTiming));
    }
    if (hs->pin_bytes < slot_bytes) {
        for (int j = 0; j < HS_SLOTS; ++j) { if (hs->pin[j]) hipHostFree(hs->pin[j]); hs->pin[j] = nullptr; }
        hs->pin_bytes = 0;
        for (int j = 0; j < HS_SLOTS; ++j) HIPCHK(c, hipHostMalloc(&hs->pin[j], slot_bytes, hipHostMallocDefault));
        hs->pin_bytes = slot_bytes;
    }
    uint8_t *dev[HS_SLOTS];
    static const char *names[HS_SLOTS] = {"xin0", "xin1", "xin2"};
    for (int j = 0; j < HS_SLOTS; ++j) {
        if ((rc = get_scratch(c, names[j], slot_bytes, &v))) return rc;
        dev[j] = (uint8_t *)v;
    }
    std::mutex mu;
    std::condition_variable cv;
    int64_t ready = 0, consumed = 0;
    int err = 0;
    double t_fill = 0.0, t_wait = 0.0;
    const int device = c->device;
    hipStream_t compute = c->stream;
    const int x_node = c->no_host_numa ? -1 : numa_node_of_buffer(X, (size_t)N * (size_t)ld * sizeof(InT));
    std::thread producer([&] {
        hipSetDevice(device);
        numa_confine_this_thread(x_node); // (the fill threads started below inherit the mask; this thread ends with the call)
        const int T = (int)std::max<int64_t>(1, std::min<int64_t>(c->host_fill_threads > 0 ? c->host_fill_threads : HS_THREADS_NARROW, N / 4096 + 1));
        for (int64_t k = 0; k < n_win; ++k) {
            const int j = (int)(k % HS_SLOTS);
            const int64_t w0 = col_lb + k * wmax, wn = std::min<int64_t>(wmax, col_ub - w0), pitch = (wn + 63) & ~63ll;
            if (k >= HS_SLOTS) { // slot j still belongs to window k - HS_SLOTS until the pass over it is done
                std::unique_lock<std::mutex> lk(mu);
                cv.wait(lk, [&] { return consumed > k - HS_SLOTS || err; });
                if (err) return;
                lk.unlock();
                if (hipEventSynchronize(hs->done[j]) != hipSuccess) { std::lock_guard<std::mutex> g(mu); err = 1; cv.notify_all(); return; }
            }
            uint8_t *dst = (uint8_t *)hs->pin[j];
            const auto t_a = std::chrono::steady_clock::now();
            auto rows = [=](int64_t r0, int64_t r1) {
                for (int64_t r = r0; r < r1; ++r) {
                    // a row piece is a few KB, the next one a whole matrix row further on: the hardware prefetchers do not follow; ask for
                    // the piece eight rows ahead by hand (a thread is otherwise held to its handful of outstanding cache misses)
                    if (r + 8 < r1) {
                        const char *pf = (const char *)(X + (size_t)(r + 8) * ld + w0);
                        for (size_t b = 0; b < (size_t)wn * sizeof(InT); b += 64) __builtin_prefetch(pf + b, 0, 1);
                    }
                    narrow_cells<InT>(X + (size_t)r * ld + w0, dst + (size_t)r * pitch, wn);
                    if (pitch > wn) memset(dst + (size_t)r * pitch + wn, 0, (size_t)(pitch - wn));
                }
            };
            std::vector<std::thread> pool;
            for (int t = 1; t < T; ++t) pool.emplace_back(rows, N * t / T, N * (t + 1) / T);
            rows(0, N / T);
            for (auto &th : pool) th.join();
            t_fill += std::chrono::duration<double>(std::chrono::steady_clock::now() - t_a).count();
            hipError_t e = hipMemcpyAsync(dev[j], dst, (size_t)pitch * (size_t)N, hipMemcpyHostToDevice, hs->copy);
            if (e == hipSuccess) e = hipEventRecord(hs->up[j], hs->copy);
            std::lock_guard<std::mutex> g(mu);
            if (e != hipSuccess) err = 1;
            ready = k + 1;
            cv.notify_all();
            if (err) return;
        }
    });
    std::vector<u32> hf;
    rc = ILLICO_OK;
    for (int64_t k = 0; k < n_win && !rc; ++k) {
        const int j = (int)(k % HS_SLOTS);
        const int64_t w0 = col_lb + k * wmax, wn = std::min<int64_t>(wmax, col_ub - w0), pitch = (wn + 63) & ~63ll;
        {
            const auto t_a = std::chrono::steady_clock::now();
            std::unique_lock<std::mutex> lk(mu);
            cv.wait(lk, [&] { return ready > k || err; });
            t_wait += std::chrono::duration<double>(std::chrono::steady_clock::now() - t_a).count();
            if (err) { rc = fail(c, ILLICO_ERR_HIP, "staging a host window failed"); break; }
        }
        if (hipStreamWaitEvent(compute, hs->up[j], 0) != hipSuccess) { rc = fail(c, ILLICO_ERR_HIP, "hipStreamWaitEvent failed"); break; }
        rc = run_fused_ovo<uint8_t>(c, dev[j], pitch, 0, (int)wn, flags, alternative, o, w0 - col_lb, hf);
        if (!rc) flagged_runs(hf.data(), wn, w0, runs); // these genes go up again in the matrix's own type (the two-pass routes)
        if (!rc && hipEventRecord(hs->done[j], compute) != hipSuccess) rc = fail(c, ILLICO_ERR_HIP, "hipEventRecord failed");
        c->h2d_input_bytes += (int64_t)((size_t)pitch * (size_t)N);
        std::lock_guard<std::mutex> g(mu);
        consumed = k + 1;
        if (rc) err = 1;
        cv.notify_all();
    }
    {
        std::lock_guard<std::mutex> g(mu);
        if (rc) err = 1;
        consumed = n_win + HS_SLOTS;
        cv.notify_all();
    }
    producer.join();
    hipStreamSynchronize(hs->copy);
    if (getenv("ILLICO_HS_DEBUG"))
        fprintf(stderr, "[illico] host byte windows: %lld x %lld genes, slot fill %.1f ms, consumer waited %.1f ms for uploads, matrix on NUMA node %d\n", (long long)n_win,
                (long long)wmax, t_fill * 1e3, t_wait * 1e3, x_node);
    return rc;
}

template <typename InT, typename KeyT>
static int run_dense_twopass(illico_ctx *c, const void *X, int dtype, int64_t N, int64_t ld, int64_t col_lb, int64_t col_ub, int flags,
                             int alternative, const OutPlanes &o, std::vector<std::pair<int64_t, int64_t>> runs, const int *col_map = nullptr,
                             bool prefer_counts = false, bool allow_packed = true);

template <typename InT, typename KeyT>
int run_leftovers(illico_ctx *c, const void *X, int dtype, int64_t N, int64_t ld, int64_t col_lb, int64_t col_ub, int flags, int alternative,
                  const OutPlanes &o, const u32 *hf, bool wide_skipped, const int *outer);

// A narrow device matrix xl [N][n_pad] of n flagged genes (gathered by run_leftovers) computed as one window: init[j] = 1: the 256-value stage first (wide_skipped: it was left to us), 3: not worth it;
// dst[j]: gene j's column of the caller's planes.
template <typename InT, typename KeyT>
static int leftovers_on_narrow(illico_ctx *c, InT *xl, int dtype, int64_t N, int64_t n, int64_t n_pad, int flags, int alternative, const OutPlanes &o,
                               const std::vector<u32> &init, const std::vector<int> &dst, bool wide_skipped, bool is_outer) {
    const int G = (int)c->n_groups;
    int rc;
    void *v;
    if ((rc = get_scratch(c, is_outer ? "xleft2_cols" : "xleft_cols", (size_t)n * 8, &v))) return rc;
    int *d_dst = (int *)v;
    u32 *d_flags2 = (u32 *)(d_dst + n);
    HIPCHK(c, hipMemcpyAsync(d_dst, dst.data(), (size_t)n * 4, hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream)); // (the host list may go out of scope)
    const int lflags = (flags | ILLICO_FLAG_INPUT_DEVICE) & ~ILLICO_FLAG_DEFER;
    if (wide_skipped) {
        std::vector<u32> hf2;
        bool any = false;
        for (int64_t j = 0; j < n; ++j) any = any || init[j] == 1u;
        if (any) {
            if ((rc = get_scratch(c, "wide_tmp", (size_t)3 * G * (size_t)n_pad * 8, &v))) return rc;
            double *tp = (double *)v;
            const OutPlanes ot{tp, tp + (size_t)G * n_pad, tp + (size_t)2 * G * n_pad, n_pad, false};
            if ((rc = run_fused_ovo<InT>(c, xl, n_pad, 0, (int)n, lflags, alternative, ot, 0, hf2, -1, false, 0, init.data()))) return rc;
            HIPCHK(c, hipMemcpyAsync(d_flags2, hf2.data(), (size_t)n * 4, hipMemcpyHostToDevice, c->stream));
            {
                ProfScope ps(c, KID_GATHER_COLS);
                const dim3 grid((unsigned)((n + 255) / 256), (unsigned)std::min(G, 1024));
                hipLaunchKernelGGL(k_scatter_planes, grid, dim3(256), 0, c->stream, (const double *)ot.p, (const double *)ot.u, (const double *)ot.fc, (long long)n_pad,
                                   (const int *)d_dst, (const u32 *)d_flags2, 2u, (int)n, G, o.p, o.u, o.fc, (long long)o.ld);
                HIPCHK(c, hipGetLastError());
            }
            HIPCHK(c, hipStreamSynchronize(c->stream));
            std::vector<u32> hf3((size_t)n);
            for (int64_t j = 0; j < n; ++j) hf3[j] = hf2[j] == 2u ? 0u : 1u;
            return run_leftovers<InT, KeyT>(c, xl, dtype, N, n_pad, 0, n, lflags, alternative, o, hf3.data(), false, dst.data());
        }
    }
    std::vector<std::pair<int64_t, int64_t>> runs{{0, n}};
    return run_dense_twopass<InT, KeyT>(c, xl, dtype, N, n_pad, 0, n, lflags, alternative, o, runs, d_dst, true);
}

// The genes the fused passes of a DEVICE-resident window [col_lb, col_ub) left behind (hf[j] = 1 / 3).  Few and scattered (a count
// matrix's highly expressed genes): gathered into a narrow matrix of their own and computed as ONE window whose results
// k_finalize scatters back through a column map (kernels_leftover.h).  Many (normalised data: every gene): the column runs, as before.
// wide_skipped: the device left the 256-value stage to us (k_wide_decide): it runs on the narrow matrix first (the genes flagged 1),
// its finished columns are copied into the caller's planes, and what it leaves is gathered once more out of the narrow matrix.
// outer (host, [W]): the window is itself such a narrow matrix -- column j of it is column outer[j] of the caller's planes.
template <typename InT, typename KeyT>
int run_leftovers(illico_ctx *c, const void *X, int dtype, int64_t N, int64_t ld, int64_t col_lb, int64_t col_ub, int flags, int alternative,
                  const OutPlanes &o, const u32 *hf, bool wide_skipped, const int *outer) {
    const int64_t W = col_ub - col_lb;
    const int G = (int)c->n_groups;
    int rc;
    void *v;
    std::vector<int> src, dst;
    for (int64_t j = 0; j < W; ++j)
        if (hf[j] == 1u || hf[j] == 3u) { src.push_back((int)(col_lb + j)); dst.push_back(outer ? outer[j] : (int)j); }
    if (src.empty()) return ILLICO_OK;
    const int64_t n = (int64_t)src.size(), n_pad = (n + 63) & ~63ll;
    const bool can_gather = (flags & ILLICO_FLAG_INPUT_DEVICE) && !c->tap && !c->no_leftover_gather && n * 2 <= W && col_ub <= 0x7FFFFFFFll &&
                            (size_t)N * (size_t)n_pad * sizeof(InT) <= (size_t)c->scratch_bytes;
    if (!can_gather) {
        std::vector<u32> merged(hf, hf + W);
        if (wide_skipped) { // (k_wide_decide only leaves the stage to us when the gather is possible; an option changed in between)
            std::vector<u32> init((size_t)W), hf2;
            for (int64_t j = 0; j < W; ++j) init[j] = hf[j] == 1u ? 1u : 3u;
            if ((rc = run_fused_ovo<InT>(c, X, ld, col_lb, (int)W, flags & ~ILLICO_FLAG_DEFER, alternative, o, 0, hf2, -1, false, 0, init.data()))) return rc;
            for (int64_t j = 0; j < W; ++j) merged[j] = ((hf[j] == 1u || hf[j] == 3u) && hf2[j] != 2u) ? 1u : 0u;
        }
        if (outer) { // a narrow matrix whose leftovers cannot be gathered again: all of it as one window, through the map
            if ((rc = get_scratch(c, "xleft_outer", (size_t)W * 4, &v))) return rc;
            HIPCHK(c, hipMemcpyAsync(v, outer, (size_t)W * 4, hipMemcpyHostToDevice, c->stream));
            HIPCHK(c, hipStreamSynchronize(c->stream));
            std::vector<std::pair<int64_t, int64_t>> all{{col_lb, col_ub}};
            return run_dense_twopass<InT, KeyT>(c, X, dtype, N, ld, col_lb, col_ub, flags, alternative, o, all, (const int *)v, true);
        }
        std::vector<std::pair<int64_t, int64_t>> runs;
        flagged_runs(merged.data(), W, col_lb, runs);
        return run_dense_twopass<InT, KeyT>(c, X, dtype, N, ld, col_lb, col_ub, flags, alternative, o, runs);
    }
    if ((rc = get_scratch(c, outer ? "xleft2" : "xleft", (size_t)N * (size_t)n_pad * sizeof(InT), &v))) return rc;
    InT *xl = (InT *)v;
    if ((rc = get_scratch(c, outer ? "xleft2_src" : "xleft_src", (size_t)n * 4, &v))) return rc;
    int *d_src = (int *)v;
    HIPCHK(c, hipMemcpyAsync(d_src, src.data(), (size_t)n * 4, hipMemcpyHostToDevice, c->stream));
    {
        ProfScope ps(c, KID_GATHER_COLS);
        hipLaunchKernelGGL((k_gather_columns<InT>), dim3((unsigned)((N + 63) / 64)), dim3(256), 0, c->stream, (const InT *)X, (long long)ld, (int)N,
                           (const int *)d_src, (int)n, (int)n_pad, xl, (long long)n_pad, 0ll);
        HIPCHK(c, hipGetLastError());
    }
    HIPCHK(c, hipStreamSynchronize(c->stream)); // (the host list goes out of scope)
    std::vector<u32> init((size_t)n);
    for (int64_t j = 0; j < n; ++j) init[j] = hf[src[j] - col_lb] == 1u ? 1u : 3u;
    return leftovers_on_narrow<InT, KeyT>(c, xl, dtype, N, n, n_pad, flags, alternative, o, init, dst, wide_skipped, outer != nullptr);
}

template <typename InT, typename KeyT>
int run_dense_t(illico_ctx *c, const void *X, int dtype, int64_t N, int64_t ld, int64_t col_lb, int64_t col_ub, int flags,
                int alternative, const OutPlanes &o) {
    const int64_t W = col_ub - col_lb;
    const bool ovr = c->ref < 0;
    const bool in_dev = flags & ILLICO_FLAG_INPUT_DEVICE;
    int rc;

    // ---- route 1 (dense, count-valued genes): fused single pass; it reports the genes it could not take ----
    // Which genes are count-valued is found by the kernels themselves (k_fused_ref reads the reference rows first, k_fused_probe
    // a few hundred rows of every gene): flagged tiles are skipped on the device, so there is no host-side route decision, no
    // sampling round trip and nothing cached between calls.
    std::vector<std::pair<int64_t, int64_t>> runs; // column ranges still to be computed by the two-pass routes
    bool try_fused = fused_path_allowed(c, flags) && (uint64_t)ld * sizeof(InT) < (1ull << 32); // row pitch: 32-bit byte offsets
    if (c->tap) try_fused = false; // the fused kernels go from values to p-values without leaving statistics behind
    if (in_dev && try_fused && N > 0 && W > 0) {
        // ILLICO_FLAG_DEFER (device planes only): enqueue and return; the flags are looked at by resolve_pending
        const bool defer = (flags & ILLICO_FLAG_DEFER) && (flags & ILLICO_FLAG_OUTPUT_DEVICE) && !o.staged;
        std::vector<u32> hf;
        // how many flagged columns run_leftovers could gather (0: it could not) -- the bound under which the device may leave the
        // 256-value stage to it (k_wide_decide)
        const int64_t max_gather = (c->no_leftover_gather || col_ub > 0x7FFFFFFFll) ? 0 : (int64_t)((size_t)c->scratch_bytes / ((size_t)N * sizeof(InT))) & ~63ll;
        if (defer) {
            const int slot = c->pend_next;
            if ((rc = run_fused_ovo<InT>(c, X, ld, col_lb, (int)W, flags, alternative, o, 0, hf, slot, ovr, max_gather))) return rc;
            c->pend_next ^= 1;
            PendingDense &q = c->pend;
            q.on = true; q.kind = 0; q.X = X; q.dtype = dtype; q.flags = flags & ~ILLICO_FLAG_DEFER; q.alternative = alternative; q.slot = slot;
            q.N = N; q.ld = ld; q.col_lb = col_lb; q.col_ub = col_ub; q.out_ld = o.ld; q.p = o.p; q.u = o.u; q.fc = o.fc;
            return ILLICO_OK;
        }
        if ((rc = run_fused_ovo<InT>(c, X, ld, col_lb, (int)W, flags, alternative, o, 0, hf, -1, ovr, max_gather))) return rc;
        return run_leftovers<InT, KeyT>(c, X, dtype, N, ld, col_lb, col_ub, flags, alternative, o, hf.data(), hf[W] != 0u);
    } else if (bool light = false; !in_dev && try_fused && N > 0 && W > 0 && c->host_narrow >= 0 && c->max_nonref <= 65535 &&
               (c->host_narrow > 0 || (host_window_is_count_valued<InT>((const InT *)X, ld, col_lb, N, W, &light) && light))) {
        // host matrix of counts: byte windows ("host_narrow": 1 forces them, -1 forbids them)
        if ((rc = host_windows_pipeline_narrow<InT>(c, (const InT *)X, ld, N, col_lb, col_ub, flags, alternative, o, runs))) return rc;
        if (runs.empty()) return ILLICO_OK;
    } else if (!in_dev && try_fused && N > 0 && W > 0 && host_window_is_count_valued<InT>((const InT *)X, ld, col_lb, N, W)) {
        // host matrix: column windows travel through pinned staging slots (host_windows_pipeline below) and take the same fused pass
        HostLeftovers left;
        if ((rc = host_windows_pipeline<InT>(c, (const InT *)X, ld, N, col_lb, col_ub, flags, alternative, o, runs, left))) return rc;
        if (left.n > 0) { // the gathered leftovers: one window of a device matrix, results scattered through the column map
            std::vector<std::pair<int64_t, int64_t>> lr{{0, left.n}};
            if ((rc = run_dense_twopass<InT, KeyT>(c, left.xl, dtype, N, left.cap, 0, left.n, flags | ILLICO_FLAG_INPUT_DEVICE, alternative, o, lr,
                                                   left.d_dst, true))) return rc;
        }
        if (runs.empty()) return ILLICO_OK;
    } else {
        runs.push_back({col_lb, col_ub});
    }
    return run_dense_twopass<InT, KeyT>(c, X, dtype, N, ld, col_lb, col_ub, flags, alternative, o, runs);
}
// ---- routes 2-4 over the column runs the fused route left (or over everything) ----
template <typename InT, typename KeyT>
// col_map (device, one entry per column of X's window): the output column of each gene, relative to the planes (the gathered
// leftover columns of a count matrix, kernels_leftover.h); prefer_counts: those genes are count-like -- the plain transposition
// with per-gene histogram routes (k_ovo_counts / k_ovr_counts) first, the routes for continuous values only for what they leave.
static int run_dense_twopass(illico_ctx *c, const void *X, int dtype, int64_t N, int64_t ld, int64_t col_lb, int64_t col_ub, int flags,
                             int alternative, const OutPlanes &o, std::vector<std::pair<int64_t, int64_t>> runs, const int *col_map,
                             bool prefer_counts, bool allow_packed) {
    const int G = (int)c->n_groups;
    const bool ovr = c->ref < 0;
    const bool in_dev = flags & ILLICO_FLAG_INPUT_DEVICE;
    prefer_counts = prefer_counts && !(flags & ILLICO_FLAG_LOG1P) && !c->no_counts_path && (ovr || counts_path_allowed(c, flags));
    // dense OVO: group-wise packing + look-ups (kernels_ovo_compact.h) whenever the sizes allow; it has no histogram side path
    // (count-valued genes reach this function only when the fused route is off, or as gathered leftovers: prefer_counts) and holds
    // ties exactly
    const bool packed = !ovr && !prefer_counts && allow_packed && packed_route_fits<KeyT>(c);
    std::vector<std::pair<int64_t, int64_t>> redo_runs; // genes the packed route left while groups above 1024 cells rule k_ovo_rank out
    // dense OVR: the transposition with the group sums folded in (k_group_compact keeping every key: padded dense layout)
    // (any group sizes: only the PACKED rows below count a (gene, group)'s non-zeros in 16 bits)
    const bool padded = ovr && !prefer_counts && !c->no_packed_dense && c->pk_nblk > 0 && c->pk_stride < (1ll << 31);
    const bool ovr_counts = ovr && prefer_counts && N < (1ll << 31);
    const int64_t stride = (packed || padded) ? c->pk_stride : ((N + 63) & ~63ll);
    int rc;
    void *v;
    // Flagged genes scattered through the window would make one tiny launch sequence each: runs closer than 32 genes
    // are merged (the good genes in between are recomputed, identically, by the two-pass routes).
    if (runs.size() > 1) {
        std::vector<std::pair<int64_t, int64_t>> merged;
        for (auto &r : runs) {
            if (!merged.empty() && r.first - merged.back().second < 32) merged.back().second = r.second;
            else merged.push_back(r);
        }
        runs.swap(merged);
    }
    int64_t widest = 0;
    for (auto &r : runs) widest = std::max(widest, r.second - r.first);

    // ---- routes 2/3: transpose pass + per-gene rank kernels, in gene batches bounded by the scratch cap ----
    const bool need_glob = !ovr && !ovo_sort_route_fits<KeyT>(c->h_counts[c->ref], c->max_nonref);
    const bool pingpong = ovr || need_glob;
    size_t per_gene = (size_t)stride * sizeof(KeyT) * (pingpong ? 2 : 1) + (pingpong ? (size_t)stride * 4 * 2 : 0) +
                      (in_dev ? 0 : (size_t)N * sizeof(InT)) + (size_t)G * 24 + 64;
    int64_t nb_max = c->gene_batch > 0 ? c->gene_batch : std::max<int64_t>(64, (int64_t)(c->scratch_bytes / per_gene));
    nb_max = std::min<int64_t>(nb_max, widest);
    if (nb_max > 64) nb_max &= ~63ll;
    nb_max = std::max<int64_t>(nb_max, 1);

    if ((rc = get_scratch(c, "xt", (size_t)nb_max * stride * sizeof(KeyT), &v))) return rc;
    KeyT *Xt = (KeyT *)v;
    if ((rc = get_scratch(c, "stats", (size_t)nb_max * G * 24 + (size_t)nb_max * 8, &v))) return rc;
    long long *s2u = (long long *)v;
    u64 *stie = (u64 *)(s2u + (size_t)nb_max * G);
    double *ssum = (double *)(stie + (size_t)nb_max * G);
    double *gtot = ssum + (size_t)nb_max * G;
    u32 *gflags = nullptr;
    if ((counts_path_allowed(c, flags) && !packed && !ovr) || ovr_counts) {
        if ((rc = get_scratch(c, "gene_flags", (size_t)nb_max * 4, &v))) return rc;
        gflags = (u32 *)v;
    }
    const int *cmap = col_map; // (finalize: output column of batch gene j = cmap[b0 - col_lb + j])
    OvoGlobalBufs gb;
    if (need_glob) {
        if ((rc = get_scratch(c, "ovr_kb", (size_t)nb_max * stride * sizeof(KeyT), &v))) return rc;
        gb.kb = v;
        if ((rc = get_scratch(c, "ovr_va", (size_t)nb_max * stride * 4, &v))) return rc;
        gb.va = (u32 *)v;
        if ((rc = get_scratch(c, "ovr_vb", (size_t)nb_max * stride * 4, &v))) return rc;
        gb.vb = (u32 *)v;
    }
    InT *xin = nullptr;
    if (!in_dev) {
        if ((rc = get_scratch(c, "xin", (size_t)nb_max * N * sizeof(InT), &v))) return rc;
        xin = (InT *)v;
    }
    for (auto &run : runs)
    for (int64_t b0 = run.first; b0 < run.second; b0 += nb_max) {
        const int nb = (int)std::min<int64_t>(nb_max, run.second - b0);
        const void *src = X;
        int64_t src_ld = ld, src_col0 = b0;
        if (!in_dev) {
            HIPCHK(c, hipMemcpy2DAsync(xin, (size_t)nb * sizeof(InT), (const InT *)X + b0, (size_t)ld * sizeof(InT),
                                       (size_t)nb * sizeof(InT), (size_t)N, hipMemcpyHostToDevice, c->stream));
            src = xin; src_ld = nb; src_col0 = 0;
        }
        if (packed) {
            std::vector<int> redo;
            if ((rc = run_ovo_packed<InT, KeyT>(c, src, src_ld, src_col0, nb, (int)N, Xt, stride, dtype, flags, s2u, stie, ssum, &redo))) return rc;
            for (int j : redo) {
                if (!redo_runs.empty() && redo_runs.back().second == b0 + j) redo_runs.back().second = b0 + j + 1;
                else redo_runs.push_back({b0 + j, b0 + j + 1});
            }
            if (c->tap) {
                const size_t off = (size_t)(b0 - col_lb) * G, cnt = (size_t)nb * G;
                HIPCHK(c, hipMemcpyAsync(c->tap->two_u + off, s2u, cnt * 8, hipMemcpyDeviceToHost, c->stream));
                HIPCHK(c, hipMemcpyAsync(c->tap->tie + off, stie, cnt * 8, hipMemcpyDeviceToHost, c->stream));
                HIPCHK(c, hipMemcpyAsync(c->tap->sum + off, ssum, cnt * 8, hipMemcpyDeviceToHost, c->stream));
                HIPCHK(c, hipStreamSynchronize(c->stream));
                continue;
            }
            if ((rc = launch_finalize(c, s2u, stie, ssum, nullptr, nb, flags, alternative, o.p, o.u, o.fc, o.ld, cmap ? 0 : b0 - col_lb, cmap ? cmap + (b0 - col_lb) : nullptr))) return rc;
            continue;
        }
        OvrPackedInput pki;
        // (groups of any size: a group of 65535 cells and more is the last of its block, whose 16-bit length the partition never looks at)
        const bool ovr_packed = padded && !c->no_ovr_packed_partition && !c->no_ovr_parts_path && G <= 65535 && (c->max_nonref <= 65535 || !c->no_ovr_packed_big);
        if (padded) {
            GroupCompactParams Q;
            memset(&Q, 0, sizeof Q);
            Q.X = src; Q.ld = src_ld; Q.col0 = src_col0; Q.ncols = nb; Q.perm = c->d_perm; Q.pos_ptr = c->d_posptr; Q.G = G; Q.ref = -1; Q.nseg = 0;
            Q.blk_g0 = c->d_pk_blk; Q.blk_g1 = c->d_pk_blk + c->pk_nblk; Q.blk_out = c->d_pk_blk + 2 * c->pk_nblk; Q.nblk = c->pk_nblk;
            Q.Xt = Xt; Q.xt_stride = stride; Q.out_sum = ssum;
            if (ovr_packed) { // packed rows (non-zero keys only) for the partition; flagged genes are written again, padded, below
                if ((rc = get_scratch(c, "packed_nnz", (size_t)nb * G * 2 + 64, &v))) return rc;
                Q.nnz = (u16 *)v;
                if ((rc = get_scratch(c, "packed_seg_sum", (size_t)nb * G * 4 + (size_t)nb * c->pk_nblk * 4 + 64, &v))) return rc;
                Q.gofs = (u32 *)v;
                Q.blk_cnt = Q.gofs + (size_t)nb * G;
                pki.nnz = Q.nnz; pki.blk_cnt = Q.blk_cnt;
                const GroupCompactParams Q0 = Q;
                pki.repad = [c, Q0, Xt, stride, flags, G](int first, int sub) -> int {
                    GroupCompactParams R = Q0;
                    R.col0 = Q0.col0 + first; R.ncols = sub; R.Xt = (KeyT *)Xt + (size_t)first * stride;
                    R.out_sum = Q0.out_sum + (size_t)first * G; R.nnz = nullptr; R.gofs = nullptr; R.blk_cnt = nullptr;
                    return launch_group_compact<InT, KeyT>(c, R, sub, flags, false);
                };
            }
            if ((rc = launch_group_compact<InT, KeyT>(c, Q, nb, flags, ovr_packed))) return rc;
        } else {
        if (gflags) HIPCHK(c, hipMemsetAsync(gflags, 0, (size_t)nb * 4, c->stream));
        if ((rc = launch_transpose<InT, KeyT>(c, src, src_ld, src_col0, nb, (int)N, Xt, stride, gflags, ovr_counts ? OVRC_R : ovo_counts_limit(c)))) return rc;
        }
        if (!ovr) {
            OvoParams P;
            P.Xs = Xt; P.gene_stride = stride; P.pos_ptr = c->d_posptr; P.seg_ptr = nullptr; P.counts = c->d_counts;
            P.G = G; P.ref = (int)c->ref; P.n_genes = nb; P.dt = dtype; P.is_log1p = (flags & ILLICO_FLAG_LOG1P) ? 1 : 0;
            P.ref_cap = 0; P.out_2u = s2u; P.out_tie = stie; P.out_sum = ssum;
            if ((rc = launch_ovo<KeyT>(c, P, c->h_counts[c->ref], c->max_nonref, gflags, &gb, false))) return rc;
            if (c->tap) {
                const size_t off = (size_t)(b0 - col_lb) * G, cnt = (size_t)nb * G;
                HIPCHK(c, hipMemcpyAsync(c->tap->two_u + off, s2u, cnt * 8, hipMemcpyDeviceToHost, c->stream));
                HIPCHK(c, hipMemcpyAsync(c->tap->tie + off, stie, cnt * 8, hipMemcpyDeviceToHost, c->stream));
                HIPCHK(c, hipMemcpyAsync(c->tap->sum + off, ssum, cnt * 8, hipMemcpyDeviceToHost, c->stream));
                HIPCHK(c, hipStreamSynchronize(c->stream));
                continue;
            }
            if ((rc = launch_finalize(c, s2u, stie, ssum, nullptr, nb, flags, alternative, o.p, o.u, o.fc, o.ld, cmap ? 0 : b0 - col_lb, cmap ? cmap + (b0 - col_lb) : nullptr))) return rc;
        } else if (ovr_counts) {
            // count-like leftovers: the column-histogram kernel takes every integer gene below OVRC_R; the value-range parts /
            // the general route only see the runs of genes it flags
            {
                OvrCountsParams Q;
                Q.Xt = Xt; Q.stride = stride; Q.pos_ptr = c->d_posptr; Q.counts = c->d_counts; Q.G = G; Q.n_genes = nb; Q.dt = dtype; Q.n_cells = N;
                Q.gene_flags = gflags; Q.out_2u = s2u; Q.out_tie = stie; Q.out_sum = ssum; Q.gene_total = gtot;
                ProfScope ps(c, KID_OVR_COUNTS);
                auto kern = k_ovr_counts<KeyT>;
                HIPCHK(c, hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, OVRC_R * 4));
                hipLaunchKernelGGL(kern, dim3(nb), dim3(OVRC_NT), OVRC_R * 4, c->stream, Q);
                HIPCHK(c, hipGetLastError());
            }
            std::vector<u32> hg(nb);
            HIPCHK(c, hipMemcpyAsync(hg.data(), gflags, (size_t)nb * 4, hipMemcpyDeviceToHost, c->stream));
            HIPCHK(c, hipStreamSynchronize(c->stream));
            for (int j = 0; j < nb;) {
                if (!hg[j]) { ++j; continue; }
                int e = j;
                while (e < nb && hg[e]) ++e;
                const int sub = e - j;
                bool done = false;
                if ((rc = run_ovr_dense_parts<KeyT>(c, Xt + (size_t)j * stride, stride, sub, (int)N, dtype, flags, s2u + (size_t)j * G, stie + (size_t)j * G,
                                                    ssum + (size_t)j * G, gtot + j, &done, false, nullptr))) return rc;
                if (!done && (rc = run_ovr_dense_batch<KeyT>(c, Xt + (size_t)j * stride, stride, sub, (int)N, dtype, flags, s2u + (size_t)j * G,
                                                             stie + (size_t)j * G, ssum + (size_t)j * G, gtot + j, false))) return rc;
                j = e;
            }
            if ((rc = launch_finalize(c, s2u, stie, ssum, gtot, nb, flags, alternative, o.p, o.u, o.fc, o.ld, cmap ? 0 : b0 - col_lb, cmap ? cmap + (b0 - col_lb) : nullptr))) return rc;
        } else {
            bool done = false;
            if ((rc = run_ovr_dense_parts<KeyT>(c, Xt, stride, nb, (int)N, dtype, flags, s2u, stie, ssum, gtot, &done, padded, ovr_packed ? &pki : nullptr))) return rc;
            if (!done) { // the parts route does not take these sizes: the general route, over padded rows
                if (ovr_packed && (rc = pki.repad(0, nb))) return rc;
                if ((rc = run_ovr_dense_batch<KeyT>(c, Xt, stride, nb, (int)N, dtype, flags, s2u, stie, ssum, gtot, padded))) return rc;
            }
            if (c->tap) {
                const size_t off = (size_t)(b0 - col_lb) * G, cnt = (size_t)nb * G;
                HIPCHK(c, hipMemcpyAsync(c->tap->two_u + off, s2u, cnt * 8, hipMemcpyDeviceToHost, c->stream));
                HIPCHK(c, hipMemcpyAsync(c->tap->tie + off, stie, cnt * 8, hipMemcpyDeviceToHost, c->stream));
                HIPCHK(c, hipMemcpyAsync(c->tap->sum + off, ssum, cnt * 8, hipMemcpyDeviceToHost, c->stream));
                HIPCHK(c, hipStreamSynchronize(c->stream));
                continue;
            }
            if ((rc = launch_finalize(c, s2u, stie, ssum, gtot, nb, flags, alternative, o.p, o.u, o.fc, o.ld, cmap ? 0 : b0 - col_lb, cmap ? cmap + (b0 - col_lb) : nullptr))) return rc;
        }
    }
    if (!redo_runs.empty()) { // (tie-heavy columns of a matrix with groups above 1024 cells: transposition + the general sort route)
        // Few genes scattered over the window (each a run of its own: one transposition, one single-workgroup sort after the other --
        // nine genes of a two-million-cell matrix: 480 ms): gathered into a narrow matrix and computed as ONE batch, side by side.
        int64_t n = 0;
        for (auto &r : redo_runs) n += r.second - r.first;
        const int64_t n_pad = (n + 63) & ~63ll;
        if (redo_runs.size() > 1 && !col_map && (flags & ILLICO_FLAG_INPUT_DEVICE) && !c->tap && !c->no_leftover_gather && n * 2 <= col_ub - col_lb &&
            col_ub <= 0x7FFFFFFFll && (size_t)N * (size_t)n_pad * sizeof(InT) <= (size_t)c->scratch_bytes) {
            std::vector<int> src, dst;
            for (auto &r : redo_runs)
                for (int64_t j = r.first; j < r.second; ++j) { src.push_back((int)j); dst.push_back((int)(j - col_lb)); }
            void *v;
            int rc;
            if ((rc = get_scratch(c, "xredo", (size_t)N * (size_t)n_pad * sizeof(InT), &v))) return rc;
            InT *xl = (InT *)v;
            if ((rc = get_scratch(c, "xredo_cols", (size_t)n * 8, &v))) return rc;
            int *d_src = (int *)v, *d_dst = d_src + n;
            HIPCHK(c, hipMemcpyAsync(d_src, src.data(), (size_t)n * 4, hipMemcpyHostToDevice, c->stream));
            HIPCHK(c, hipMemcpyAsync(d_dst, dst.data(), (size_t)n * 4, hipMemcpyHostToDevice, c->stream));
            {
                ProfScope ps(c, KID_GATHER_COLS);
                hipLaunchKernelGGL((k_gather_columns<InT>), dim3((unsigned)((N + 63) / 64)), dim3(256), 0, c->stream, (const InT *)X, (long long)ld, (int)N,
                                   (const int *)d_src, (int)n, (int)n_pad, xl, (long long)n_pad, 0ll);
                HIPCHK(c, hipGetLastError());
            }
            HIPCHK(c, hipStreamSynchronize(c->stream)); // (the host lists go out of scope)
            std::vector<std::pair<int64_t, int64_t>> all{{0, n}};
            return run_dense_twopass<InT, KeyT>(c, xl, dtype, N, n_pad, 0, n, flags, alternative, o, all, d_dst, prefer_counts, false);
        }
        return run_dense_twopass<InT, KeyT>(c, X, dtype, N, ld, col_lb, col_ub, flags, alternative, o, redo_runs, col_map, prefer_counts, false);
    }
    return ILLICO_OK;
}
