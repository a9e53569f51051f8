// libillico_hip: MI355X (gfx950) engine behind include/illico_hip.h.
// This translation unit: context, device scratch, groups, output staging, the C-ABI entry points and the dispatch on value / index
// types.  The kernels live in the per-type translation units (dense_*.hip, sparse_*.hip, keyed_*.hip).
#include "engine.h"

const char *const kKernelNames[KID_COUNT] = {"k_transpose_permute", "k_ovo_rank", "k_ovo_counts", "k_ovo_fused", "k_ovr_fused",
                                              "k_fused_tables", "k_finalize", "k_ovr_gene", "k_sparse_seg", "k_csc_gene", "k_gene_totals", "k_csc_counts", "k_csc_ovr_gene", "k_ovr_partition", "k_ovr_rank_parts", "k_value_sums", "k_ovo_fused_wide", "k_group_compact", "k_ovo_rank_compact", "k_ovr_counts", "k_gather_columns", "k_csr_counts", "k_densify", "k_group_value_hists"};

// The message of a failed call is kept per calling thread (and in the context, for single-threaded callers): a second
// thread's failure must not replace the text the first is about to read through illico_last_error.
static thread_local std::string t_err;
static thread_local const illico_ctx *t_err_ctx = nullptr;

int fail(illico_ctx *c, int code, const char *fmt, ...) {
    if (c) {
        char buf[512];
        va_list ap;
        va_start(ap, fmt);
        vsnprintf(buf, sizeof buf, fmt, ap);
        va_end(ap);
        CTX_LOCK(c);
        c->err = buf;
        t_err = buf;
        t_err_ctx = c;
    }
    return code;
}
int get_scratch(illico_ctx *c, const char *name, size_t bytes, void **out) {
    auto &s = c->scratch[name];
    if (s.second < bytes) {
        if (s.first) {
            HIPCHK(c, hipStreamSynchronize(c->stream));
            HIPCHK(c, hipFree(s.first));
            s.first = nullptr;
            s.second = 0;
        }
        size_t want = bytes + (bytes >> 4) + 256;
        void *p = nullptr;
        hipError_t e = hipMalloc(&p, want);
        if (e != hipSuccess) return fail(c, ILLICO_ERR_OOM, "hipMalloc(%zu bytes) for scratch '%s' failed: %s", want, name, hipGetErrorString(e));
        s.first = p;
        s.second = want;
    }
    *out = s.first;
    return ILLICO_OK;
}

// Timing events come from a pool (no create / destroy per launch) and carry no system-scope fence: recording one
// must not flush the L2 between kernels of the timed region.
hipEvent_t take_event(illico_ctx *c) {
    if (!c->event_pool.empty()) {
        hipEvent_t e = c->event_pool.back();
        c->event_pool.pop_back();
        return e;
    }
    hipEvent_t e = nullptr;
    if (hipEventCreateWithFlags(&e, hipEventDisableSystemFence) != hipSuccess) hipEventCreate(&e);
    return e;
}
void drain_events(illico_ctx *c) {
    if (c->events.empty()) return;
    hipStreamSynchronize(c->stream);
    for (auto &e : c->events) {
        float ms = 0.f;
        if (hipEventElapsedTime(&ms, e.a, e.b) == hipSuccess) {
            c->prof_ms[e.kid] += ms;
            c->prof_n[e.kid] += 1;
        }
        c->event_pool.push_back(e.a);
        c->event_pool.push_back(e.b);
    }
    c->events.clear();
}


// ============================================================================================
extern "C" {

const char *illico_version(void) { return "illico_hip 0.1 (gfx950)"; }

int illico_ctx_create(int device_id, illico_ctx **out_ctx) {
    if (!out_ctx) return ILLICO_ERR_ARG;
    *out_ctx = nullptr;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return ILLICO_ERR_HIP;
    if (device_id < 0 || device_id >= ndev) return ILLICO_ERR_ARG;
    illico_ctx *c = new illico_ctx();
    c->device = device_id;
    if (hipSetDevice(device_id) != hipSuccess || hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking) != hipSuccess) {
        delete c;
        return ILLICO_ERR_HIP;
    }
    c->own_stream = true;
    { // scratch cap: 64 GiB of the 288 GB an MI355X carries (a C2-shaped continuous OVR pass then runs as one gene batch), a quarter
      // of the device's memory on anything smaller; "scratch_bytes" overrides
        size_t free_b = 0, total_b = 0;
        if (hipMemGetInfo(&free_b, &total_b) == hipSuccess && total_b > 0) c->scratch_bytes = (int64_t)std::min<size_t>((size_t)64 << 30, total_b / 4);
    }
    *out_ctx = c;
    return ILLICO_OK;
}

static void free_groups(illico_ctx *c) {
    for (int **p : {&c->d_codes, &c->d_perm, &c->d_posptr, &c->d_counts, &c->d_code_by_pos, &c->d_pk_blk, &c->d_pk_code, &c->d_pk_big, &c->d_pk_long, &c->d_pk_order}) {
        if (*p) hipFree(*p);
        *p = nullptr;
    }
    if (c->d_codes16) hipFree(c->d_codes16);
    c->d_codes16 = nullptr;
    if (c->d_pk_islong) hipFree(c->d_pk_islong);
    c->d_pk_islong = nullptr;
    c->pk_nlong = 0;
    if (c->d_hist_off) hipFree(c->d_hist_off);
    c->d_hist_off = nullptr;
    if (c->d_gconst) hipFree(c->d_gconst);
    c->d_gconst = nullptr;
    if (c->d_csr_chunks) hipFree(c->d_csr_chunks);
    c->d_csr_chunks = nullptr;
    c->csr_n_chunks = c->csr_n_big = 0;
    c->has_groups = false;
}

int illico_ctx_destroy(illico_ctx *c) {
    if (!c) return ILLICO_ERR_ARG;
    hipSetDevice(c->device);
    hipStreamSynchronize(c->stream);
    drain_events(c);
    for (hipEvent_t e : c->event_pool) hipEventDestroy(e);
    if (c->pinned) hipHostFree(c->pinned);
    for (int k = 0; k < 2; ++k) {
        if (c->out_pin[k]) hipHostFree(c->out_pin[k]);
        if (c->out_ev[k]) hipEventDestroy(c->out_ev[k]);
        if (c->pend_pinned[k]) hipHostFree(c->pend_pinned[k]);
        if (c->pend_event[k]) hipEventDestroy(c->pend_event[k]);
    }
    free_groups(c);
    free_host_stage(c);
    for (auto &a : c->ahead) if (a.planes) hipFree(a.planes);
    for (illico_matrix *m : c->bound) {
        if (m->owns) { hipFree(m->d_data); hipFree(m->d_indices); hipFree(m->d_indptr); }
        delete m;
    }
    for (auto &kv : c->scratch)
        if (kv.second.first) hipFree(kv.second.first);
    if (c->own_stream && c->stream) hipStreamDestroy(c->stream);
    delete c;
    return ILLICO_OK;
}

int illico_ctx_set_stream(illico_ctx *c, void *hip_stream) {
    if (!c) return ILLICO_ERR_ARG;
    CTX_LOCK(c);
    hipSetDevice(c->device);
    int rc = resolve_pending(c);
    if (rc) return rc;
    hipStreamSynchronize(c->stream);
    drain_events(c);
    if (c->own_stream && c->stream) hipStreamDestroy(c->stream);
    c->stream = (hipStream_t)hip_stream;
    c->own_stream = false;
    return ILLICO_OK;
}

int illico_ctx_set_option(illico_ctx *c, const char *key, int64_t value) {
    if (!c || !key) return ILLICO_ERR_ARG;
    CTX_LOCK(c);
    if (!strcmp(key, "gene_batch")) c->gene_batch = value;
    else if (!strcmp(key, "scratch_bytes")) c->scratch_bytes = value;
    else if (!strcmp(key, "profile")) c->profile = value != 0;
    else if (!strcmp(key, "profile_only")) c->profile_only = (value >= 0 && value < KID_COUNT) ? (int)value : -1;
    else if (!strcmp(key, "no_counts_path")) c->no_counts_path = value != 0;
    else if (!strcmp(key, "no_fused_path")) c->no_fused_path = value != 0;
    else if (!strcmp(key, "no_fused_wide")) c->no_fused_wide = value != 0;
    else if (!strcmp(key, "no_leftover_gather")) c->no_leftover_gather = value != 0;
    else if (!strcmp(key, "no_wide_gather")) c->no_wide_gather = value != 0;
    else if (!strcmp(key, "bound_ahead_genes")) c->bound_ahead_genes = value < 0 ? 0 : value;
    else if (!strcmp(key, "no_csc_counts_windows")) c->no_csc_counts_windows = value != 0;
    else if (!strcmp(key, "no_csc_counts_wide")) c->no_csc_counts_wide = value != 0;
    else if (!strcmp(key, "ovr_full_dump")) c->ovr_full_dump = value != 0;
    else if (!strcmp(key, "no_packed_dense")) c->no_packed_dense = value != 0;
    else if (!strcmp(key, "packed_eq_buckets")) c->packed_eq_buckets = (int)value;
    else if (!strcmp(key, "no_ovo_parts")) c->no_ovo_parts = value != 0;
    else if (!strcmp(key, "no_packed_small_wg")) c->no_packed_small_wg = value != 0;
    else if (!strcmp(key, "no_deal_runs")) c->no_deal_runs = value != 0;
    else if (!strcmp(key, "no_coop_runs")) c->no_coop_runs = value != 0;
    else if (!strcmp(key, "no_csc_ovr_small_lds")) c->no_csc_ovr_small_lds = value != 0;
    else if (!strcmp(key, "no_csr_transpose_split")) c->no_csr_transpose_split = value != 0;
    else if (!strcmp(key, "no_group_hist_route")) c->no_group_hist_route = value != 0;
    else if (!strcmp(key, "group_hist_max_wgs")) c->group_hist_max_wgs = value > 0 ? value : 1024;
    else if (!strcmp(key, "group_hist_min_cells")) c->group_hist_min_cells = value > 0 ? value : 32768;
    else if (!strcmp(key, "no_ovr_part_coop")) c->no_ovr_part_coop = value != 0;
    else if (!strcmp(key, "no_ovr_packed_big")) c->no_ovr_packed_big = value != 0;
    else if (!strcmp(key, "big_runs_slice_bytes")) c->big_runs_slice_bytes = (int)value;
    else if (!strcmp(key, "no_big_runs_wide")) c->no_big_runs_wide = value != 0;
    else if (!strcmp(key, "no_compact_order")) c->no_compact_order = value != 0;
    else if (!strcmp(key, "no_compact_narrow")) c->no_compact_narrow = value != 0;
    else if (!strcmp(key, "compact_narrow_wgs")) c->compact_narrow_wgs = value > 0 ? value : 2048;
    else if (!strcmp(key, "compact_narrow_rows")) c->compact_narrow_rows = value > 0 ? value : 8192;
    else if (!strcmp(key, "no_big_runs_global")) c->no_big_runs_global = value != 0;
    else if (!strcmp(key, "packed_ref_cap")) c->packed_ref_cap = (int)value;
    else if (!strcmp(key, "debug_routes")) c->debug_routes = value != 0;
    else if (!strcmp(key, "host_fill_threads")) c->host_fill_threads = (int)value;
    else if (!strcmp(key, "prewarm_host_window_bytes")) { // what a context's FIRST call on a host-resident dense matrix otherwise pays for (~80 ms
        // of a 190-ms first drop-in call at C2): the three pinned slots and the three device windows of the host-window pipelines
        if (value > 0) {
            if (hipSetDevice(c->device) != hipSuccess) return ILLICO_ERR_HIP;
            HostStage *hs = host_stage_of(c);
            const size_t want = (size_t)value;
            if (hs->pin_bytes < want) {
                for (int j = 0; j < HS_SLOTS; ++j) { if (hs->pin[j]) hipHostFree(hs->pin[j]); hs->pin[j] = nullptr; }
                hs->pin_bytes = 0;
                for (int j = 0; j < HS_SLOTS; ++j)
                    if (hipHostMalloc(&hs->pin[j], want, hipHostMallocDefault) != hipSuccess) return fail(c, ILLICO_ERR_OOM, "pinning %zu bytes failed", want);
                hs->pin_bytes = want;
            }
            static const char *names[HS_SLOTS] = {"xin0", "xin1", "xin2"};
            void *v;
            for (int j = 0; j < HS_SLOTS; ++j) { const int rc = get_scratch(c, names[j], want, &v); if (rc) return rc; }
        }
    }
    else if (!strcmp(key, "no_host_numa")) c->no_host_numa = value != 0;
    else if (!strcmp(key, "no_sparse_byte_values")) c->no_sparse_byte_values = value != 0;
    else if (!strcmp(key, "csc_counts_max_windows")) c->csc_counts_max_windows = (int)value;
    else if (!strcmp(key, "no_sparse_packed_small")) c->no_sparse_packed_small = value != 0;
    else if (!strcmp(key, "big_runs_cap")) c->big_runs_cap = (int)value;
    else if (!strcmp(key, "no_ovr_packed_partition")) c->no_ovr_packed_partition = value != 0;
    else if (!strcmp(key, "no_csc_counts_path")) c->no_csc_counts_path = value != 0;
    else if (!strcmp(key, "no_csc_counts_mixed")) c->no_csc_counts_mixed = value != 0;
    else if (!strcmp(key, "no_csc_regroup_lds")) c->no_csc_regroup_lds = value != 0;
    else if (!strcmp(key, "no_csc_gene_path")) c->no_csc_gene_path = value != 0;
    else if (!strcmp(key, "ovr_parts_cap")) c->ovr_parts_cap = value;
    else if (!strcmp(key, "ovr_rank_whole")) c->ovr_rank_whole = value != 0;
    else if (!strcmp(key, "no_ovo_ref_buckets")) c->no_ovo_ref_buckets = value != 0;
    else if (!strcmp(key, "no_ovr_parts_path")) c->no_ovr_parts_path = value != 0;
    else if (!strcmp(key, "no_csc_ovr_gene_path")) c->no_csc_ovr_gene_path = value != 0;
    else if (!strcmp(key, "csc_ovr_sorted_form")) c->csc_ovr_sorted_form = value != 0;
    else if (!strcmp(key, "no_ovr_one_pass")) c->no_ovr_one_pass = value != 0;
    else if (!strcmp(key, "no_ovr_library_sort")) (void)value; // accepted and ignored: there is no library sort any more
    else if (!strcmp(key, "no_csr_tile_gather")) c->no_csr_tile_gather = value != 0;
    else if (!strcmp(key, "no_csr_transpose_path")) c->no_csr_transpose_path = value != 0;
    else if (!strcmp(key, "no_csr_densify_any")) c->no_csr_densify_any = value != 0;
    else if (!strcmp(key, "no_f64_narrowing")) c->no_f64_narrowing = value != 0;
    else if (!strcmp(key, "dense_window_f32")) c->dense_window_f32 = value != 0;
    else if (!strcmp(key, "host_narrow")) c->host_narrow = value > 0 ? 1 : (value < 0 ? -1 : 0);
    else if (!strcmp(key, "no_csr_counts_path")) c->no_csr_counts_path = value != 0;
    else if (!strcmp(key, "csr_counts_abl")) c->csr_counts_abl = (int)value;
    else if (!strcmp(key, "no_dense_window_path")) c->no_dense_window_path = value != 0;
    else if (!strcmp(key, "ovr_hist_groups_per_wg")) c->ovr_hist_groups_per_wg = (int)std::max<int64_t>(0, value);
    else if (!strcmp(key, "fused_groups_per_wg")) c->fused_groups_per_wg = (int)std::max<int64_t>(0, value);
    else return fail(c, ILLICO_ERR_ARG, "unknown option '%s'", key);
    return ILLICO_OK;
}

const char *illico_last_error(const illico_ctx *c) {
    if (!c) return "null context";
    return t_err_ctx == c ? t_err.c_str() : c->err.c_str();
}

int illico_ctx_synchronize(illico_ctx *c) {
    if (!c) return ILLICO_ERR_ARG;
    CTX_LOCK(c);
    HIPCHK(c, hipSetDevice(c->device));
    int rc = resolve_pending(c); // a deferred call's leftover genes are recomputed now
    if (rc) return rc;
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return ILLICO_OK;
}

int illico_profile_num_kernels(void) { return KID_COUNT; }
const char *illico_profile_kernel_name(int k) { return (k >= 0 && k < KID_COUNT) ? kKernelNames[k] : ""; }
int illico_profile_get(illico_ctx *c, int k, double *total_ms, int64_t *launches) {
    if (!c || k < 0 || k >= KID_COUNT) return ILLICO_ERR_ARG;
    CTX_LOCK(c);
    hipSetDevice(c->device);
    drain_events(c);
    if (total_ms) *total_ms = c->prof_ms[k];
    if (launches) *launches = c->prof_n[k];
    return ILLICO_OK;
}
int illico_profile_input_bytes(illico_ctx *c, int64_t *h2d_bytes) {
    if (!c || !h2d_bytes) return ILLICO_ERR_ARG;
    CTX_LOCK(c);
    *h2d_bytes = c->h2d_input_bytes;
    return ILLICO_OK;
}
int illico_profile_reset(illico_ctx *c) {
    if (!c) return ILLICO_ERR_ARG;
    CTX_LOCK(c);
    hipSetDevice(c->device);
    drain_events(c);
    for (int k = 0; k < KID_COUNT; ++k) { c->prof_ms[k] = 0; c->prof_n[k] = 0; }
    return ILLICO_OK;
}

// ---- groups ---------------------------------------------------------------------------------
int illico_set_groups(illico_ctx *c, const int64_t *encoded_groups, const int64_t *counts, const int64_t *indices,
                      const int64_t *indptr, int64_t n_cells, int64_t n_groups, int64_t ref) {
    if (!c) return ILLICO_ERR_ARG;
    CTX_LOCK(c);
    if (!encoded_groups || !counts || !indices || !indptr) return fail(c, ILLICO_ERR_ARG, "null group array");
    if (n_cells <= 0 || n_groups <= 0 || n_cells > 0x7FFFFFF0ll) return fail(c, ILLICO_ERR_ARG, "bad n_cells/n_groups");
    if (ref < -1 || ref >= n_groups) return fail(c, ILLICO_ERR_ARG, "encoded_ref_group out of range");
    HIPCHK(c, hipSetDevice(c->device));
    { int rc0 = resolve_pending(c); if (rc0) return rc0; } // (its leftover genes need the groups it was made with)
    HIPCHK(c, hipStreamSynchronize(c->stream));
    free_groups(c);
    // perm is padded with valid indices: tail chunks of k_ovo_fused read (and discard) up to 8 entries past a group's end
    std::vector<int> codes(n_cells), perm(n_cells + 64, 0), cbp(n_cells), posptr(n_groups + 1), cnt(n_groups);
    int64_t tot = 0, max_nonref = 0;
    for (int64_t g = 0; g < n_groups; ++g) {
        if (counts[g] < 0 || indptr[g] != tot) return fail(c, ILLICO_ERR_ARG, "indptr/counts inconsistent at group %lld", (long long)g);
        cnt[g] = (int)counts[g];
        posptr[g] = (int)tot;
        tot += counts[g];
        if (g != ref) max_nonref = std::max<int64_t>(max_nonref, counts[g]);
    }
    if (tot != n_cells || indptr[n_groups] != n_cells) return fail(c, ILLICO_ERR_ARG, "counts do not sum to n_cells");
    {   // n (n-1) (n+1) and the t^3 tie terms are 64-bit integer products, as in the reference (utils/math.py:95,
        // ranking.py:107): they hold up to n = 2^21 - 1 cells per test (n = n_ref + n_tgt for OVO, every cell for OVR).
        // Beyond that the reference's int64 wraps silently.  OVO: refused.  OVR: the groups are accepted -- SPARSE input then takes the
        // routes whose arithmetic holds (float64 zero block and variance terms, kernels_finalize.h: tie_f64_sparse / pval_nnn) and returns
        // what the reference's formulas give WITHOUT the wrap; dense input is refused at the call (illico_run_dense).
        const int64_t n_test = ref < 0 ? n_cells : counts[ref] + max_nonref;
        if (n_test > 2097151 && ref >= 0)
            return fail(c, ILLICO_ERR_UNSUPPORTED, "%lld cells in one test: n(n-1)(n+1) and the tie sums overflow 64-bit integers beyond 2097151 cells (the reference's int64 arithmetic wraps there, utils/math.py:95)", (long long)n_test);
        c->big_n = n_test > 2097151;
    }
    posptr[n_groups] = (int)n_cells;
    for (int64_t i = 0; i < n_cells; ++i) {
        int64_t g = encoded_groups[i];
        if (g < 0 || g >= n_groups) return fail(c, ILLICO_ERR_ARG, "encoded group out of range at cell %lld", (long long)i);
        codes[i] = (int)g;
    }
    for (int64_t g = 0; g < n_groups; ++g)
        for (int64_t p = indptr[g]; p < indptr[g + 1]; ++p) {
            int64_t cell = indices[p];
            if (cell < 0 || cell >= n_cells || codes[cell] != g) return fail(c, ILLICO_ERR_ARG, "indices[%lld] is not a cell of group %lld", (long long)p, (long long)g);
            perm[p] = (int)cell;
            cbp[p] = (int)g;
        }
    auto up = [&](int **d, const std::vector<int> &h) -> int {
        HIPCHK(c, hipMalloc((void **)d, h.size() * sizeof(int)));
        HIPCHK(c, hipMemcpy(*d, h.data(), h.size() * sizeof(int), hipMemcpyHostToDevice));
        return ILLICO_OK;
    };
    int rc;
    if ((rc = up(&c->d_codes, codes)) || (rc = up(&c->d_perm, perm)) || (rc = up(&c->d_posptr, posptr)) ||
        (rc = up(&c->d_counts, cnt)) || (rc = up(&c->d_code_by_pos, cbp)))
        return rc;
    {
        std::vector<GroupConst> gc(n_groups);
        for (int64_t g = 0; g < n_groups; ++g) {
            const long long n_tgt = cnt[g], n_ref = ref >= 0 ? (long long)cnt[ref] : (long long)n_cells - n_tgt;
            gc[g] = group_const(n_ref, n_tgt, ref >= 0 ? n_ref + n_tgt : (long long)n_cells);
        }
        HIPCHK(c, hipMalloc((void **)&c->d_gconst, gc.size() * sizeof(GroupConst)));
        HIPCHK(c, hipMemcpy(c->d_gconst, gc.data(), gc.size() * sizeof(GroupConst), hipMemcpyHostToDevice));
    }
    {
        std::vector<u32> ho(n_groups + 1, 0u);
        for (int64_t g = 0; g < n_groups; ++g) ho[g + 1] = ho[g] + (counts[g] <= 255 ? 16u : 32u);
        HIPCHK(c, hipMalloc((void **)&c->d_hist_off, ho.size() * sizeof(u32)));
        HIPCHK(c, hipMemcpy(c->d_hist_off, ho.data(), ho.size() * sizeof(u32), hipMemcpyHostToDevice));
        c->hist_words = ho[n_groups];
    }
    c->pk_nblk = 0;
    { // blocks of the packed / padded dense layouts: consecutive groups (never the reference) of >= GCMP_BLOCK_ROWS rows together
        std::vector<int> g0, g1, out;
        int64_t pos = 0, rows = 0;
        bool open = false;
        c->pk_max_block_rows = 0;
        auto close = [&](int64_t end) { g1.push_back((int)end); pos += (rows + 63) & ~63ll; open = false; c->pk_max_block_rows = std::max(c->pk_max_block_rows, rows); };
        for (int64_t g = 0; g < n_groups; ++g) {
            if (g == ref) { if (open) close(g); continue; }
            if (!open) { g0.push_back((int)g); out.push_back((int)pos); rows = 0; open = true; }
            rows += counts[g];
            if (rows >= GCMP_BLOCK_ROWS) close(g + 1);
        }
        if (open) close(n_groups);
        std::vector<int> packed;
        packed.insert(packed.end(), g0.begin(), g0.end());
        packed.insert(packed.end(), g1.begin(), g1.end());
        packed.insert(packed.end(), out.begin(), out.end());
        if (packed.empty()) packed.push_back(0);
        c->pk_nblk = (int)g0.size();
        { // blocks of very different lengths (clusters from fifty to tens of thousands of cells): k_group_compact numbers its workgroups
            // block-major, longest block first, so that a long block's chain of chunks starts with the launch instead of ending it
            std::vector<int64_t> rows_b(g0.size(), 0);
            int64_t tot = 0, mx = 0;
            for (size_t b = 0; b < g0.size(); ++b) {
                for (int g = g0[b]; g < g1[b]; ++g) rows_b[b] += counts[g];
                tot += rows_b[b]; mx = std::max(mx, rows_b[b]);
            }
            if (g0.size() >= 2 && mx * (int64_t)g0.size() >= 4 * tot) {
                std::vector<int> ord(g0.size());
                for (size_t b = 0; b < ord.size(); ++b) ord[b] = (int)b;
                std::stable_sort(ord.begin(), ord.end(), [&](int a, int b) { return rows_b[a] > rows_b[b]; });
                HIPCHK(c, hipMalloc((void **)&c->d_pk_order, ord.size() * sizeof(int)));
                HIPCHK(c, hipMemcpy(c->d_pk_order, ord.data(), ord.size() * sizeof(int), hipMemcpyHostToDevice));
            }
        }
        { // long blocks (a cluster of thousands of cells, the control group of a screen): see k_ovr_partition_packed<COOP>
            std::vector<int> lng;
            std::vector<unsigned char> is_long(g0.size() + 1, 0);
            for (size_t b = 0; b < g0.size(); ++b) {
                int64_t rows_b = 0;
                for (int g = g0[b]; g < g1[b]; ++g) rows_b += counts[g];
                if (rows_b > 4096 /* OVRP_LONG_ROWS */ && g1[b] - g0[b] <= 64) { lng.push_back((int)b); is_long[b] = 1; }
            }
            c->pk_nlong = (int)lng.size();
            if (lng.empty()) lng.push_back(0);
            HIPCHK(c, hipMalloc((void **)&c->d_pk_long, lng.size() * sizeof(int)));
            HIPCHK(c, hipMemcpy(c->d_pk_long, lng.data(), lng.size() * sizeof(int), hipMemcpyHostToDevice));
            HIPCHK(c, hipMalloc((void **)&c->d_pk_islong, is_long.size()));
            HIPCHK(c, hipMemcpy(c->d_pk_islong, is_long.data(), is_long.size(), hipMemcpyHostToDevice));
        }
        c->pk_ref_out = (int)pos;
        c->pk_len = pos;
        c->pk_stride = pos + (ref >= 0 ? ((counts[ref] + 63) & ~63ll) : 0) + 64;
        HIPCHK(c, hipMalloc((void **)&c->d_pk_blk, packed.size() * sizeof(int)));
        HIPCHK(c, hipMemcpy(c->d_pk_blk, packed.data(), packed.size() * sizeof(int), hipMemcpyHostToDevice));
        { // groups whose packed runs can exceed the 256 keys the packed rank kernel looks up at a time
            std::vector<int> big;
            for (int64_t g = 0; g < n_groups; ++g)
                if (g != ref && counts[g] > 256) big.push_back((int)g);
            c->pk_nbig = (int)big.size();
            if (!big.empty()) {
                std::vector<int> both(big);
                both.resize(big.size() + (size_t)n_groups, -1);
                for (size_t k = 0; k < big.size(); ++k) both[big.size() + (size_t)big[k]] = (int)k;
                HIPCHK(c, hipMalloc((void **)&c->d_pk_big, both.size() * sizeof(int)));
                HIPCHK(c, hipMemcpy(c->d_pk_big, both.data(), both.size() * sizeof(int), hipMemcpyHostToDevice));
            }
        }
        if (ref < 0) { // dense OVR walks the padded rows: group code per key slot
            std::vector<int> pc((size_t)c->pk_stride, 0);
            for (size_t b = 0; b < g0.size(); ++b) {
                int64_t o = out[b];
                for (int g = g0[b]; g < g1[b]; ++g)
                    for (int64_t k = 0; k < counts[g]; ++k) pc[(size_t)o++] = g;
            }
            HIPCHK(c, hipMalloc((void **)&c->d_pk_code, pc.size() * sizeof(int)));
            HIPCHK(c, hipMemcpy(c->d_pk_code, pc.data(), pc.size() * sizeof(int), hipMemcpyHostToDevice));
        }
    }
    if (n_groups <= 65535) {
        std::vector<u16> c16(codes.begin(), codes.end());
        HIPCHK(c, hipMalloc((void **)&c->d_codes16, c16.size() * sizeof(u16)));
        HIPCHK(c, hipMemcpy(c->d_codes16, c16.data(), c16.size() * sizeof(u16), hipMemcpyHostToDevice));
    }
    { // row chunks of the group-major CSR pass (kernels_csr_counts.h)
        std::vector<int> p0, nr, slab, big;
        auto chunks = [&](int64_t first, int64_t rows, int s, bool natural) {
            for (int64_t r = 0; r < rows; r += CSRH_ROWS) {
                p0.push_back(natural ? (int)(-1 - (first + r)) : (int)(first + r));
                nr.push_back((int)std::min<int64_t>(CSRH_ROWS, rows - r));
                slab.push_back(s);
            }
        };
        if (ref >= 0) chunks(indptr[ref], counts[ref], 0, false); // (OVR: the column histograms come out of the count pass itself)
        for (int64_t g = 0; g < n_groups; ++g)
            if (g != ref && counts[g] > 255) big.push_back((int)g);
        c->csr_n_big = (int)big.size();
        if (big.size() > CSRC_MAX_BIG) { c->csr_n_big = -1; big.clear(); }
        for (size_t k = 0; k < big.size(); ++k) chunks(indptr[big[k]], counts[big[k]], 1 + (int)k, false);
        c->csr_n_chunks = (int)p0.size();
        std::vector<int> all;
        all.insert(all.end(), p0.begin(), p0.end());
        all.insert(all.end(), nr.begin(), nr.end());
        all.insert(all.end(), slab.begin(), slab.end());
        all.insert(all.end(), big.begin(), big.end());
        if (all.empty()) all.push_back(0);
        HIPCHK(c, hipMalloc((void **)&c->d_csr_chunks, all.size() * sizeof(int)));
        HIPCHK(c, hipMemcpy(c->d_csr_chunks, all.data(), all.size() * sizeof(int), hipMemcpyHostToDevice));
    }
    c->h_counts = cnt;
    c->n_cells = n_cells;
    c->n_groups = n_groups;
    c->ref = ref;
    c->max_nonref = max_nonref;
    c->has_groups = true;
    ++c->groups_gen;
    return ILLICO_OK;
}

} // extern "C"
int launch_gene_totals(illico_ctx *c, const double *ssum, int G, int nb, double *gtot) {
    ProfScope ps(c, KID_GENE_TOTALS);
    hipLaunchKernelGGL(k_gene_totals, dim3((nb + 63) / 64), dim3(256), 0, c->stream, ssum, G, nb, gtot);
    HIPCHK(c, hipGetLastError());
    return ILLICO_OK;
}
// table size of the two-pass histogram route (k_ovo_counts): 4096 values with 8-bit multiplicities while no ranked group exceeds 255
// cells, else 2048 with 16-bit ones; the ingest kernels flag genes against the same limit
int ovo_counts_limit(const illico_ctx *c) { return c->max_nonref <= 255 ? COUNTS_R8 : COUNTS_R; }

// the histogram path needs integer value sums (no expm1) and 16-bit group bins
bool counts_path_allowed(const illico_ctx *c, int flags) {
    return !(flags & ILLICO_FLAG_LOG1P) && c->ref >= 0 && c->max_nonref <= 65535 && !c->no_counts_path;
}
// fused single-pass routes (OVO and OVR): integer value sums (no expm1), 16-bit running multiplicities (OVO) / histogram cells (OVR in one pass),
// 32-bit chunk partial sums (n_cells < 2^25)
bool fused_path_allowed(const illico_ctx *c, int flags) {
    // (OVO: 16-bit running multiplicities per group.  OVR keeps no per-group state in its two-pass form -- a control group of 70 000 cells
    //  among 2 000 000 sent the whole matrix to the general sort route: 115 ms for 9.6 GB -- only its one-pass form counts in 16-bit cells)
    //  OVO with ranked groups above 65 535 cells -- cluster against cluster -- runs the same kernel with 32-bit multiplicities: 158 -> see NOTES_r04)
    return !(flags & ILLICO_FLAG_LOG1P) && c->n_cells < (1ll << 25) && !c->no_counts_path && !c->no_fused_path;
}

int launch_finalize(illico_ctx *c, const long long *s2u, const u64 *stie, const double *ssum, const double *gene_total,
                           int nb, int flags, int alternative, double *out_p, double *out_u, double *out_fc, int64_t out_ld,
                           int64_t col_off, const int *col_map, bool packed, bool tie_f64) {
    FinalizeParams F;
    F.col_map = col_map;
    F.packed = packed ? 1 : 0;
    F.tie_f64 = tie_f64 ? 1 : 0;
    F.in_2u = s2u; F.in_tie = stie; F.in_sum = ssum; F.gene_total = gene_total;
    F.counts = c->d_counts; F.gconst = c->d_gconst; F.G = (int)c->n_groups; F.ref = (int)c->ref; F.nb = nb; F.n_cells = c->n_cells;
    F.use_continuity = (flags & ILLICO_FLAG_CONTINUITY) ? 1 : 0;
    F.tie_correct = (flags & ILLICO_FLAG_TIE_CORRECT) ? 1 : 0;
    F.alternative = alternative;
    F.out_p = out_p + col_off; F.out_u = out_u + col_off; F.out_fc = out_fc + col_off; F.out_ld = out_ld;
    ProfScope ps(c, KID_FINALIZE);
    dim3 grid((nb + 31) / 32, ((int)c->n_groups + 31) / 32);
    hipLaunchKernelGGL(k_finalize, grid, dim3(256), 0, c->stream, F);
    HIPCHK(c, hipGetLastError());
    return ILLICO_OK;
}
static int begin_outputs(illico_ctx *c, int flags, int64_t W, double *out_p, double *out_u, double *out_fc, int64_t out_ld, OutPlanes *o) {
    if (flags & ILLICO_FLAG_OUTPUT_DEVICE) {
        *o = {out_p, out_u, out_fc, out_ld, false};
        return ILLICO_OK;
    }
    void *buf;
    size_t plane = (size_t)c->n_groups * (size_t)W;
    int rc = get_scratch(c, "out_planes", plane * 3 * sizeof(double), &buf);
    if (rc) return rc;
    double *b = (double *)buf;
    *o = {b, b + plane, b + 2 * plane, W, true};
    return ILLICO_OK;
}

// Freshly allocated host planes (np.empty: 384 MB at C2) take their page faults when they are first written -- in end_outputs, after
// the device is done, on the scatter threads: ~35 ms at C2.  PlaneTouch takes them early instead: a few threads touch one byte per
// page of the three destination windows (read and written back: contents are preserved) while the uploads and the kernels run.
struct PlaneTouch {
    std::vector<std::thread> pool;
    void start(double *const planes[3], size_t n_rows, size_t row_bytes, size_t pitch_bytes) {
        if (3 * n_rows * row_bytes < ((size_t)64 << 20)) return;
        const int T = 8;
        double *p0 = planes[0], *p1 = planes[1], *p2 = planes[2];
        const bool dbg = getenv("ILLICO_HS_DEBUG") != nullptr;
        for (int t = 0; t < T; ++t)
            pool.emplace_back([=]() {
                const auto t0 = std::chrono::steady_clock::now();
                double *const pl[3] = {p0, p1, p2};
                for (size_t r = 3 * n_rows * t / T; r < 3 * n_rows * (t + 1) / T; ++r) {
                    char *row = (char *)pl[r / n_rows] + (r % n_rows) * pitch_bytes; // (8-byte aligned: a row of doubles)
                    // (volatile read + write-back: the page is faulted in for writing, its contents stay; an atomic add of 0 is
                    // folded into a load by the compiler.  Nobody else touches the planes before join().)
                    for (size_t b = 0; b < row_bytes; b += 4096) { volatile char *q = row + b; *q = *q; }
                    { volatile char *q = row + row_bytes - 1; *q = *q; }
                }
                if (dbg && t == 0) fprintf(stderr, "[illico] plane touch thread 0: %.1f ms\n", std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count());
            });
    }
    void join() {
        for (auto &th : pool) th.join();
        pool.clear();
    }
    ~PlaneTouch() { join(); }
};

// Host planes: the device staging planes come back through two pinned 32-MB buffers (row blocks of the three planes in turn:
// block i is copied down at the link's rate while block i - 1 is scattered into the caller's planes by a few host threads).  A
// pageable destination made the driver stage the 24 bytes per test itself: 20 - 40 ms for C2's 384 MB, against ~10 ms.
static int end_outputs(illico_ctx *c, const OutPlanes &o, int64_t W, double *out_p, double *out_u, double *out_fc, int64_t out_ld) {
    if (!o.staged) return ILLICO_OK;
    const size_t G = (size_t)c->n_groups, row = (size_t)W * 8;
    const size_t total = 3 * G * row;
    if (total < ((size_t)8 << 20)) { // small results: three strided copies
        HIPCHK(c, hipMemcpy2DAsync(out_p, out_ld * 8, o.p, W * 8, W * 8, G, hipMemcpyDeviceToHost, c->stream));
        HIPCHK(c, hipMemcpy2DAsync(out_u, out_ld * 8, o.u, W * 8, W * 8, G, hipMemcpyDeviceToHost, c->stream));
        HIPCHK(c, hipMemcpy2DAsync(out_fc, out_ld * 8, o.fc, W * 8, W * 8, G, hipMemcpyDeviceToHost, c->stream));
        HIPCHK(c, hipStreamSynchronize(c->stream));
        return ILLICO_OK;
    }
    const size_t buf = (size_t)32 << 20;
    if (c->out_pin_bytes < buf) {
        for (int k = 0; k < 2; ++k) { if (c->out_pin[k]) hipHostFree(c->out_pin[k]); c->out_pin[k] = nullptr; }
        c->out_pin_bytes = 0;
        for (int k = 0; k < 2; ++k) HIPCHK(c, hipHostMalloc(&c->out_pin[k], buf, hipHostMallocDefault));
        for (int k = 0; k < 2; ++k) if (!c->out_ev[k]) HIPCHK(c, hipEventCreateWithFlags(&c->out_ev[k], hipEventDisableTiming));
        c->out_pin_bytes = buf;
    }
    const size_t rows_per = std::max<size_t>(1, buf / row), n_rows = 3 * G; // rows of the three planes, one after the other
    if (row > buf) { // (a window too wide for the buffers: the plain copies)
        HIPCHK(c, hipMemcpy2DAsync(out_p, out_ld * 8, o.p, W * 8, W * 8, G, hipMemcpyDeviceToHost, c->stream));
        HIPCHK(c, hipMemcpy2DAsync(out_u, out_ld * 8, o.u, W * 8, W * 8, G, hipMemcpyDeviceToHost, c->stream));
        HIPCHK(c, hipMemcpy2DAsync(out_fc, out_ld * 8, o.fc, W * 8, W * 8, G, hipMemcpyDeviceToHost, c->stream));
        HIPCHK(c, hipStreamSynchronize(c->stream));
        return ILLICO_OK;
    }
    const double *src[3] = {o.p, o.u, o.fc};
    double *dst[3] = {out_p, out_u, out_fc};
    auto scatter = [&](int k, size_t r0, size_t r1) { // rows [r0, r1) of the concatenated planes, from pinned buffer k
        const char *from = (const char *)c->out_pin[k];
        const int T = (r1 - r0) * row >= ((size_t)4 << 20) ? 4 : 1;
        std::vector<std::thread> pool;
        auto part = [&](int t) {
            for (size_t r = r0 + (r1 - r0) * t / T; r < r0 + (r1 - r0) * (t + 1) / T; ++r)
                memcpy(dst[r / G] + (r % G) * (size_t)out_ld, from + (r - r0) * row, row);
        };
        for (int t = 1; t < T; ++t) pool.emplace_back(part, t);
        part(0);
        for (auto &th : pool) th.join();
    };
    size_t prev0 = 0, prev1 = 0;
    int i = 0;
    for (size_t r0 = 0; r0 < n_rows; r0 += rows_per, ++i) {
        const size_t r1 = std::min(n_rows, r0 + rows_per);
        const int k = i & 1;
        // a block may straddle two planes: one contiguous device range per plane it touches (the staging planes are [G][W], dense)
        for (size_t r = r0; r < r1;) {
            const size_t pl = r / G, e = std::min(r1, (pl + 1) * G);
            HIPCHK(c, hipMemcpyAsync((char *)c->out_pin[k] + (r - r0) * row, src[pl] + (r % G) * (size_t)W, (e - r) * row, hipMemcpyDeviceToHost, c->stream));
            r = e;
        }
        HIPCHK(c, hipEventRecord(c->out_ev[k], c->stream));
        if (i > 0) {
            HIPCHK(c, hipEventSynchronize(c->out_ev[k ^ 1]));
            scatter(k ^ 1, prev0, prev1);
        }
        prev0 = r0; prev1 = r1;
    }
    if (i > 0) {
        HIPCHK(c, hipEventSynchronize(c->out_ev[(i - 1) & 1]));
        scatter((i - 1) & 1, prev0, prev1);
    }
    return ILLICO_OK;
}

static int check_common(illico_ctx *c, int64_t n_rows, int64_t n_cols, int64_t col_lb, int64_t col_ub, int alternative,
                        const void *o1, const void *o2, const void *o3, int64_t out_ld) {
    if (!c) return ILLICO_ERR_ARG;
    if (!c->has_groups) return fail(c, ILLICO_ERR_NO_GROUPS, "illico_set_groups has not been called");
    if (n_rows != c->n_cells) return fail(c, ILLICO_ERR_NO_GROUPS, "X has %lld rows but the groups describe %lld cells", (long long)n_rows, (long long)c->n_cells);
    if (col_lb < 0 || col_ub > n_cols || col_lb > col_ub) return fail(c, ILLICO_ERR_BOUNDS, "Invalid chunk bounds: (%lld, %lld) for data with %lld columns.", (long long)col_lb, (long long)col_ub, (long long)n_cols);
    if (alternative < 0 || alternative > 2) return fail(c, ILLICO_ERR_ALTERNATIVE, "Unsupported alternative hypothesis code %d", alternative);
    if (!o1 || !o2 || !o3) return fail(c, ILLICO_ERR_ARG, "null output plane");
    if (out_ld < col_ub - col_lb) return fail(c, ILLICO_ERR_ARG, "out_ld smaller than the chunk width");
    return ILLICO_OK;
}
// column runs [first, second) of the flagged genes of a window starting at column w0
void flagged_runs(const u32 *hf, int64_t wn, int64_t w0, std::vector<std::pair<int64_t, int64_t>> &runs) {
    // (1 = the gene left the fused route; 2 = taken by its second, wider pass: done)
    for (int64_t j = 0; j < wn;) {
        if (hf[j] != 1u && hf[j] != 3u) { ++j; continue; } // (3: flagged by the probe as no count at all)
        int64_t e = j;
        while (e < wn && (hf[e] == 1u || hf[e] == 3u)) ++e;
        if (!runs.empty() && runs.back().second == w0 + j) runs.back().second = w0 + e;
        else runs.push_back({w0 + j, w0 + e});
        j = e;
    }
}
HostStage *host_stage_of(illico_ctx *c) { // (one per context, freed with it)
    if (!c->host_stage) c->host_stage = new HostStage();
    return c->host_stage;
}
void free_host_stage(illico_ctx *c) {
    HostStage *hs = c->host_stage;
    if (!hs) return;
    if (hs->copy) { hipStreamSynchronize(hs->copy); hipStreamDestroy(hs->copy); }
    if (hs->lists) hipHostFree(hs->lists);
    for (int j = 0; j < 2; ++j) {
        if (hs->sp_pin[j]) hipHostFree(hs->sp_pin[j]);
        if (hs->sp_up[j]) hipEventDestroy(hs->sp_up[j]);
    }
    for (int j = 0; j < HS_SLOTS; ++j) {
        if (hs->pin[j]) hipHostFree(hs->pin[j]);
        if (hs->up[j]) hipEventDestroy(hs->up[j]);
        if (hs->done[j]) hipEventDestroy(hs->done[j]);
    }
    delete hs;
    c->host_stage = nullptr;
}
// Completes a deferred dense call: waits for its route flags and sends the genes the fused pass could not take through the
// two-pass routes.  Every entry point that takes the context runs this first (illico_run_dense may enqueue its own fused pass
// before it, see there), so results are complete after illico_ctx_synchronize or any later call.
static int resolve_pending_csc(illico_ctx *c, const PendingDense &q); // sparse_driver.h
static int resolve_pending(illico_ctx *c, PendingDense q) {
    if (!q.on) return ILLICO_OK;
    HIPCHK(c, hipSetDevice(c->device));
    HIPCHK(c, hipEventSynchronize(c->pend_event[q.slot]));
    if (q.kind == 1) return resolve_pending_csc(c, q);
    const u32 *hf = (const u32 *)c->pend_pinned[q.slot];
    const bool skipped = hf[q.col_ub - q.col_lb] != 0u; // the 256-value stage was left to run_leftovers (k_wide_decide)
    const OutPlanes o{q.p, q.u, q.fc, q.out_ld, false};
    switch (q.dtype) {
    case ILLICO_F32: return run_leftovers<float, u32>(c, q.X, q.dtype, q.N, q.ld, q.col_lb, q.col_ub, q.flags, q.alternative, o, hf, skipped);
#ifndef ILLICO_DEV_F32_ONLY
    case ILLICO_F64: return run_leftovers<double, u64>(c, q.X, q.dtype, q.N, q.ld, q.col_lb, q.col_ub, q.flags, q.alternative, o, hf, skipped);
    case ILLICO_I32: return run_leftovers<int32_t, u32>(c, q.X, q.dtype, q.N, q.ld, q.col_lb, q.col_ub, q.flags, q.alternative, o, hf, skipped);
    default: return run_leftovers<int64_t, u64>(c, q.X, q.dtype, q.N, q.ld, q.col_lb, q.col_ub, q.flags, q.alternative, o, hf, skipped);
#else
    default: return fail(c, ILLICO_ERR_DTYPE, "this development build holds the float32 kernels only");
#endif
    }
}
int resolve_pending(illico_ctx *c) {
    const PendingDense q = c->pend;
    c->pend.on = false;
    return resolve_pending(c, q);
}

static int run_dense_any(illico_ctx *c, const void *X, int dtype, int64_t n_rows, int64_t ld, int64_t col_lb, int64_t col_ub, int flags,
                         int alternative, const OutPlanes &o) {
    switch (dtype) {
    case ILLICO_F32: return run_dense_t<float, u32>(c, X, dtype, n_rows, ld, col_lb, col_ub, flags, alternative, o);
#ifndef ILLICO_DEV_F32_ONLY // development builds (ILLICO_DEV_F32_ONLY=1 python build.py) compile the float32 kernels only: 4x faster to build
    case ILLICO_F64: return run_dense_t<double, u64>(c, X, dtype, n_rows, ld, col_lb, col_ub, flags, alternative, o);
    case ILLICO_I32: return run_dense_t<int32_t, u32>(c, X, dtype, n_rows, ld, col_lb, col_ub, flags, alternative, o);
    default: return run_dense_t<int64_t, u64>(c, X, dtype, n_rows, ld, col_lb, col_ub, flags, alternative, o);
#else
    default: return fail(c, ILLICO_ERR_DTYPE, "this development build holds the float32 kernels only");
#endif
    }
}

extern "C" int illico_run_dense(illico_ctx *c, const void *X, int dtype, int64_t n_rows, int64_t n_cols, int64_t ld,
                                int64_t col_lb, int64_t col_ub, int flags, int alternative, double *out_p, double *out_u,
                                double *out_fc, int64_t out_ld) {
    if (!c) return ILLICO_ERR_ARG;
    CTX_LOCK(c);
    int rc = check_common(c, n_rows, n_cols, col_lb, col_ub, alternative, out_p, out_u, out_fc, out_ld);
    if (rc) return rc;
    if (!X) return fail(c, ILLICO_ERR_ARG, "null X");
    if (ld < n_cols) return fail(c, ILLICO_ERR_ARG, "ld smaller than n_cols");
    if (dtype < 0 || dtype > 3) return fail(c, ILLICO_ERR_DTYPE, "unsupported dtype code %d", dtype);
    if (c->big_n) return fail(c, ILLICO_ERR_UNSUPPORTED, "%lld cells in one test: the dense routes' 64-bit tie sums hold up to 2097151 cells (the reference's int64 arithmetic wraps there, utils/math.py:95); sparse input is taken", (long long)c->n_cells);
    HIPCHK(c, hipSetDevice(c->device));
    const int64_t W = col_ub - col_lb;
    // A deferred call still in flight: when this call is deferred too and writes other planes, its fused pass is enqueued
    // FIRST (the GPU goes from one pass to the next without waiting for the host) and the earlier call is completed after;
    // otherwise the earlier call is completed before anything else happens.
    PendingDense prev = c->pend;
    c->pend.on = false;
    bool later = false;
    if (prev.on && (flags & ILLICO_FLAG_DEFER) && (flags & ILLICO_FLAG_OUTPUT_DEVICE) && (flags & ILLICO_FLAG_INPUT_DEVICE) && W > 0) {
        const size_t span = (size_t)(c->n_groups - 1) * (size_t)out_ld + (size_t)W, pspan = (size_t)(c->n_groups - 1) * (size_t)prev.out_ld + (size_t)(prev.col_ub - prev.col_lb);
        auto apart = [](const double *a, size_t na, const double *b, size_t nb) { return a + na <= b || b + nb <= a; };
        later = true;
        for (const double *a : {out_p, out_u, out_fc})
            for (const double *b : {prev.p, prev.u, prev.fc}) later = later && apart(a, span, b, pspan);
    }
    if (!later && (rc = resolve_pending(c, prev))) return rc;
    if (W == 0) return later ? resolve_pending(c, prev) : ILLICO_OK;
    OutPlanes o;
    if ((rc = begin_outputs(c, flags, W, out_p, out_u, out_fc, out_ld, &o))) { if (later) resolve_pending(c, prev); return rc; }
    PlaneTouch touch; // (joined before the first result is scattered, and on every way out)
    if (o.staged) { double *const dst[3] = {out_p, out_u, out_fc}; touch.start(dst, (size_t)c->n_groups, (size_t)W * 8, (size_t)out_ld * 8); }
    rc = run_dense_any(c, X, dtype, n_rows, ld, col_lb, col_ub, flags, alternative, o);
    if (later) { const int rc2 = resolve_pending(c, prev); if (!rc) rc = rc2; }
    if (rc) return rc;
    touch.join();
    return end_outputs(c, o, W, out_p, out_u, out_fc, out_ld);
}

extern "C" int illico_planes_to_host(illico_ctx *c, const double *dev_p, const double *dev_u, const double *dev_fc, int64_t n_cols, double *out_p,
                                     double *out_u, double *out_fc, int64_t out_ld) {
    if (!c) return ILLICO_ERR_ARG;
    CTX_LOCK(c);
    if (!c->has_groups) return fail(c, ILLICO_ERR_NO_GROUPS, "illico_set_groups has not been called");
    if (!dev_p || !dev_u || !dev_fc || !out_p || !out_u || !out_fc) return fail(c, ILLICO_ERR_ARG, "null plane");
    if (n_cols < 0 || out_ld < n_cols) return fail(c, ILLICO_ERR_ARG, "out_ld smaller than n_cols");
    HIPCHK(c, hipSetDevice(c->device));
    int rc = resolve_pending(c);
    if (rc || n_cols == 0) return rc;
    const OutPlanes o{const_cast<double *>(dev_p), const_cast<double *>(dev_u), const_cast<double *>(dev_fc), n_cols, true};
    PlaneTouch touch;
    { double *const dst[3] = {out_p, out_u, out_fc}; touch.start(dst, (size_t)c->n_groups, (size_t)n_cols * 8, (size_t)out_ld * 8); }
    touch.join();
    return end_outputs(c, o, n_cols, out_p, out_u, out_fc, out_ld);
}

extern "C" int illico_rank_statistics(illico_ctx *c, const void *X, int dtype, int64_t n_rows, int64_t n_cols, int64_t ld, int64_t col_lb,
                                      int64_t col_ub, int flags, int64_t *out_two_u, uint64_t *out_tie_sum, double *out_value_sum) {
    if (!c) return ILLICO_ERR_ARG;
    CTX_LOCK(c);
    int rc = check_common(c, n_rows, n_cols, col_lb, col_ub, 0, out_two_u, out_tie_sum, out_value_sum, col_ub - col_lb);
    if (rc) return rc;
    if (!X) return fail(c, ILLICO_ERR_ARG, "null X");
    if (ld < n_cols) return fail(c, ILLICO_ERR_ARG, "ld smaller than n_cols");
    if (dtype < 0 || dtype > 3) return fail(c, ILLICO_ERR_DTYPE, "unsupported dtype code %d", dtype);
    HIPCHK(c, hipSetDevice(c->device));
    if ((rc = resolve_pending(c))) return rc;
    if (c->big_n) return fail(c, ILLICO_ERR_UNSUPPORTED, "%lld cells in one test: dense input holds up to 2097151", (long long)c->n_cells);
    if (col_ub == col_lb) return ILLICO_OK;
    illico_ctx::StatsTap tap{(long long *)out_two_u, (u64 *)out_tie_sum, out_value_sum};
    c->tap = &tap;
    OutPlanes none{nullptr, nullptr, nullptr, 0, false};
    rc = run_dense_any(c, X, dtype, n_rows, ld, col_lb, col_ub, flags & (ILLICO_FLAG_LOG1P | ILLICO_FLAG_INPUT_DEVICE), 0, none);
    c->tap = nullptr;
    return rc;
}
// dispatch on the value / index types (no argument checks, no deferred-call bookkeeping: run_sparse does both)
int run_sparse_inner(illico_ctx *c, bool is_csr, const void *data, int dtype, const void *indices, const void *indptr,
                            int idx_dtype, int64_t n_rows, int64_t n_cols, int64_t col_lb, int64_t col_ub, int flags, int alternative,
                            const OutPlanes &o) {
    int rc;
#ifndef ILLICO_DEV_F32_ONLY
#define SP_CALL(InT, KeyT)                                                                                                 \
    (idx_dtype == ILLICO_IDX_I32                                                                                           \
         ? run_sparse_t<InT, int32_t, KeyT>(c, is_csr, data, indices, indptr, dtype, n_rows, n_cols, col_lb, col_ub, flags, alternative, o) \
         : run_sparse_t<InT, int64_t, KeyT>(c, is_csr, data, indices, indptr, dtype, n_rows, n_cols, col_lb, col_ub, flags, alternative, o))
    switch (dtype) {
    case ILLICO_F32: rc = SP_CALL(float, u32); break;
    case ILLICO_F64: rc = SP_CALL(double, u64); break;
    case ILLICO_I32: rc = SP_CALL(int32_t, u32); break;
    default: rc = SP_CALL(int64_t, u64); break;
    }
#undef SP_CALL
#else // development build: float32 values, int32 indices only
    if (dtype == ILLICO_F32 && idx_dtype == ILLICO_IDX_I32)
        rc = run_sparse_t<float, int32_t, u32>(c, is_csr, data, indices, indptr, dtype, n_rows, n_cols, col_lb, col_ub, flags, alternative, o);
    else rc = fail(c, ILLICO_ERR_DTYPE, "this development build holds the float32 / int32-index kernels only");
#endif
    return rc;
}

// the columns a deferred count-valued CSC / CSR pass could not take, through the ordinary routes
static int resolve_pending_csc(illico_ctx *c, const PendingDense &q) {
    const u32 *hf = (const u32 *)c->pend_pinned[q.slot];
    const int64_t W = q.col_ub - q.col_lb;
    // what is known about the rows' order belongs to the matrix of the call that is running: while a PENDING call's columns are recomputed
    // it is that call's knowledge that holds (another bound matrix may be the running one: illico_run_bound sets the flag before it gets here)
    struct Sorted { illico_ctx *c; bool was; Sorted(illico_ctx *c_, bool now) : c(c_), was(c_->cur_sorted_known) { c->cur_sorted_known = now; }
                    ~Sorted() { c->cur_sorted_known = was; } } sorted_scope(c, q.is_csr && q.sorted_known);
    if (q.is_csr) { // the group-major CSR pass: flags + 4 verdict words; its leftovers must not come back to it
        struct Hold { illico_ctx *c; bool was; Hold(illico_ctx *c_) : c(c_), was(c_->hold_csr_counts) { c->hold_csr_counts = true; } ~Hold() { c->hold_csr_counts = was; } } hold(c);
        const u32 *vd = hf + W;
        int64_t n_flagged = 0;
        for (int64_t j = 0; j < W; ++j) n_flagged += hf[j] ? 1 : 0;
        if ((double)vd[0] > 0.02 * (double)vd[2] || (double)vd[1] > 0.005 * (double)vd[2] || vd[3] != 0u || n_flagged * 16 > W) { // not a matrix for the route (or many genes left it): all of it
            const OutPlanes o{q.p, q.u, q.fc, q.out_ld, false};
            return run_sparse_inner(c, true, q.sp_data, q.dtype, q.sp_indices, q.sp_indptr, q.idx_dtype, q.N, q.n_cols, q.col_lb, q.col_ub, q.flags, q.alternative, o);
        }
        for (int64_t j = 0; j < W;) { // runs of flagged genes (closer than 32 genes: one run)
            if (!hf[j]) { ++j; continue; }
            int64_t last = j;
            for (int64_t e = j + 1; e < W && e - last <= 32; ++e) if (hf[e]) last = e;
            const OutPlanes o{q.p + j, q.u + j, q.fc + j, q.out_ld, false};
            const int rc = run_sparse_inner(c, true, q.sp_data, q.dtype, q.sp_indices, q.sp_indptr, q.idx_dtype, q.N, q.n_cols, q.col_lb + j, q.col_lb + last + 1,
                                            q.flags, q.alternative, o);
            if (rc) return rc;
            j = last + 1;
        }
        return ILLICO_OK;
    }
    // Many scattered flagged columns (a denser matrix: 4-bit cells overflowing in every other gene) are completed by ONE call over
    // the range that covers them -- the route itself works on column lists and recomputes an unflagged column identically -- not by
    // one call, with its value sample and host waits, per run of flagged columns (4000 runs: 250 ms at C3 shape with half the entries stored).
    {
        int64_t runs = 0, first = -1, last = -1;
        for (int64_t j = 0; j < W; ++j)
            if (hf[j]) { if (j == 0 || !hf[j - 1]) ++runs; if (first < 0) first = j; last = j; }
        if (runs > 8) {
            const OutPlanes o{q.p + first, q.u + first, q.fc + first, q.out_ld, false};
            return run_sparse_inner(c, false, q.sp_data, q.dtype, q.sp_indices, q.sp_indptr, q.idx_dtype, q.N, q.n_cols, q.col_lb + first,
                                    q.col_lb + last + 1, q.flags, q.alternative, o);
        }
    }
    for (int64_t j = 0; j < W;) {
        if (!hf[j]) { ++j; continue; }
        int64_t e = j;
        while (e < W && hf[e]) ++e;
        const OutPlanes o{q.p + j, q.u + j, q.fc + j, q.out_ld, false};
        const int rc = run_sparse_inner(c, false, q.sp_data, q.dtype, q.sp_indices, q.sp_indptr, q.idx_dtype, q.N, q.n_cols, q.col_lb + j,
                                        q.col_lb + e, q.flags, q.alternative, o);
        if (rc) return rc;
        j = e;
    }
    return ILLICO_OK;
}

static int run_sparse(illico_ctx *c, bool is_csr, const void *data, int dtype, const void *indices, const void *indptr,
                      int idx_dtype, int64_t n_rows, int64_t n_cols, int64_t col_lb, int64_t col_ub, int flags, int alternative,
                      double *out_p, double *out_u, double *out_fc, int64_t out_ld) {
    if (!c) return ILLICO_ERR_ARG;
    CTX_LOCK(c);
    int rc = check_common(c, n_rows, n_cols, col_lb, col_ub, alternative, out_p, out_u, out_fc, out_ld);
    if (rc) return rc;
    if (!data || !indices || !indptr) return fail(c, ILLICO_ERR_ARG, "null sparse array");
    if (dtype < 0 || dtype > 3) return fail(c, ILLICO_ERR_DTYPE, "unsupported dtype code %d", dtype);
    if (idx_dtype != ILLICO_IDX_I32 && idx_dtype != ILLICO_IDX_I64) return fail(c, ILLICO_ERR_DTYPE, "unsupported index dtype code %d", idx_dtype);
    HIPCHK(c, hipSetDevice(c->device));
    const int64_t W = col_ub - col_lb;
    // A deferred call still in flight: as in illico_run_dense, a deferred call that writes OTHER planes is enqueued first and
    // the earlier one completed after (the GPU goes from one pass to the next without waiting for the host); otherwise the
    // earlier call is completed before anything else happens.
    PendingDense prev = c->pend;
    c->pend.on = false;
    bool later = false;
    if (prev.on && (flags & ILLICO_FLAG_DEFER) && (flags & ILLICO_FLAG_OUTPUT_DEVICE) && (flags & ILLICO_FLAG_INPUT_DEVICE) && W > 0) {
        const size_t span = (size_t)(c->n_groups - 1) * (size_t)out_ld + (size_t)W, pspan = (size_t)(c->n_groups - 1) * (size_t)prev.out_ld + (size_t)(prev.col_ub - prev.col_lb);
        auto apart = [](const double *a, size_t na, const double *b, size_t nb) { return a + na <= b || b + nb <= a; };
        later = true;
        for (const double *a : {out_p, out_u, out_fc})
            for (const double *b : {prev.p, prev.u, prev.fc}) later = later && apart(a, span, b, pspan);
    }
    if (!later && (rc = resolve_pending(c, prev))) return rc;
    if (W == 0) return later ? resolve_pending(c, prev) : ILLICO_OK;
    OutPlanes o;
    if ((rc = begin_outputs(c, flags, W, out_p, out_u, out_fc, out_ld, &o))) { if (later) resolve_pending(c, prev); return rc; }
    PlaneTouch touch;
    if (o.staged) { double *const dst[3] = {out_p, out_u, out_fc}; touch.start(dst, (size_t)c->n_groups, (size_t)W * 8, (size_t)out_ld * 8); }
    rc = run_sparse_inner(c, is_csr, data, dtype, indices, indptr, idx_dtype, n_rows, n_cols, col_lb, col_ub, flags, alternative, o);
    if (later) { // (the earlier call's leftovers run on the ordinary routes; this call's own pending state must survive them)
        const PendingDense mine = c->pend;
        c->pend.on = false;
        const int rc2 = resolve_pending(c, prev);
        c->pend = mine;
        if (!rc) rc = rc2;
    }
    if (rc) return rc;
    touch.join();
    return end_outputs(c, o, W, out_p, out_u, out_fc, out_ld);
}

extern "C" int illico_run_csc(illico_ctx *c, const void *data, int dtype, const void *indices, const void *indptr,
                              int idx_dtype, int64_t n_rows, int64_t n_cols, int64_t col_lb, int64_t col_ub, int flags,
                              int alternative, double *out_p, double *out_u, double *out_fc, int64_t out_ld) {
    return run_sparse(c, false, data, dtype, indices, indptr, idx_dtype, n_rows, n_cols, col_lb, col_ub, flags, alternative,
                      out_p, out_u, out_fc, out_ld);
}
extern "C" int illico_run_csr(illico_ctx *c, const void *data, int dtype, const void *indices, const void *indptr,
                              int idx_dtype, int64_t n_rows, int64_t n_cols, int64_t col_lb, int64_t col_ub, int flags,
                              int alternative, double *out_p, double *out_u, double *out_fc, int64_t out_ld) {
    return run_sparse(c, true, data, dtype, indices, indptr, idx_dtype, n_rows, n_cols, col_lb, col_ub, flags, alternative,
                      out_p, out_u, out_fc, out_ld);
}

// ---- bound matrices -------------------------------------------------------------------------
// the rows' order of a bound CSR matrix, looked at once: the group-major CSR pass of every later call relies on it
static void look_at_row_order(illico_ctx *c, illico_matrix *m) {
    m->sorted = -1;
    if (!m->is_csr || m->n_rows >= (1ll << 31)) return;
    void *v;
    int bad = 0;
    if (get_scratch(c, "flag", 16, &v) == ILLICO_OK && hipMemsetAsync(v, 0, 4, c->stream) == hipSuccess) {
        const unsigned grid = (unsigned)std::min<int64_t>((m->n_rows + 3) / 4 + 1, 8192);
        if (m->idx_dtype == ILLICO_IDX_I32)
            hipLaunchKernelGGL((k_csr_sorted_check<int32_t>), dim3(grid), dim3(256), 0, c->stream, (const int32_t *)m->d_indices, (const int32_t *)m->d_indptr, (int)m->n_rows, (int *)v);
        else
            hipLaunchKernelGGL((k_csr_sorted_check<int64_t>), dim3(grid), dim3(256), 0, c->stream, (const int64_t *)m->d_indices, (const int64_t *)m->d_indptr, (int)m->n_rows, (int *)v);
        if (hipGetLastError() == hipSuccess && hipMemcpyAsync(&bad, v, 4, hipMemcpyDeviceToHost, c->stream) == hipSuccess &&
            hipStreamSynchronize(c->stream) == hipSuccess)
            m->sorted = bad ? 0 : 1;
    }
}

static int sparse_bind(illico_ctx *c, bool is_csr, const void *data, int dtype, const void *indices, const void *indptr, int idx_dtype,
                       int64_t n_rows, int64_t n_cols, int flags, illico_matrix **out) {
    if (!c || !out) return ILLICO_ERR_ARG;
    *out = nullptr;
    CTX_LOCK(c);
    if (!data || !indices || !indptr) return fail(c, ILLICO_ERR_ARG, "null sparse array");
    if (dtype < 0 || dtype > 3) return fail(c, ILLICO_ERR_DTYPE, "unsupported dtype code %d", dtype);
    if (idx_dtype != ILLICO_IDX_I32 && idx_dtype != ILLICO_IDX_I64) return fail(c, ILLICO_ERR_DTYPE, "unsupported index dtype code %d", idx_dtype);
    if (n_rows <= 0 || n_cols < 0) return fail(c, ILLICO_ERR_ARG, "bad matrix shape");
    HIPCHK(c, hipSetDevice(c->device));
    const size_t isz = idx_dtype == ILLICO_IDX_I32 ? 4 : 8, vsz = dtype_size(dtype);
    const int64_t n_ptr = (is_csr ? n_rows : n_cols) + 1;
    illico_matrix *m = new illico_matrix();
    m->owner = c; m->is_csr = is_csr; m->dtype = dtype; m->idx_dtype = idx_dtype; m->n_rows = n_rows; m->n_cols = n_cols;
    if (flags & ILLICO_FLAG_INPUT_DEVICE) { // adopt: nothing is copied, the caller keeps the arrays alive
        m->d_data = const_cast<void *>(data); m->d_indices = const_cast<void *>(indices); m->d_indptr = const_cast<void *>(indptr);
        m->nnz = -1;
    } else {
        const int64_t nnz = idx_dtype == ILLICO_IDX_I32 ? (int64_t)((const int32_t *)indptr)[n_ptr - 1] : ((const int64_t *)indptr)[n_ptr - 1];
        const int64_t first = idx_dtype == ILLICO_IDX_I32 ? (int64_t)((const int32_t *)indptr)[0] : ((const int64_t *)indptr)[0];
        if (first != 0 || nnz < 0) { delete m; return fail(c, ILLICO_ERR_ARG, "indptr[0] must be 0 and indptr[-1] >= 0"); }
        m->nnz = nnz;
        m->owns = true;
        const size_t cnt = (size_t)std::max<int64_t>(nnz, 1);
        hipError_t e;
        if ((e = hipMalloc(&m->d_data, cnt * vsz)) != hipSuccess || (e = hipMalloc(&m->d_indices, cnt * isz)) != hipSuccess ||
            (e = hipMalloc(&m->d_indptr, (size_t)n_ptr * isz)) != hipSuccess) {
            hipFree(m->d_data); hipFree(m->d_indices); hipFree(m->d_indptr);
            delete m;
            return fail(c, ILLICO_ERR_OOM, "hipMalloc for a bound matrix of %lld stored entries failed: %s", (long long)nnz, hipGetErrorString(e));
        }
        hipError_t e1 = hipMemcpyAsync(m->d_data, data, (size_t)nnz * vsz, hipMemcpyHostToDevice, c->stream);
        hipError_t e2 = hipMemcpyAsync(m->d_indices, indices, (size_t)nnz * isz, hipMemcpyHostToDevice, c->stream);
        hipError_t e3 = hipMemcpyAsync(m->d_indptr, indptr, (size_t)n_ptr * isz, hipMemcpyHostToDevice, c->stream);
        hipError_t e4 = hipStreamSynchronize(c->stream); // the caller's arrays are free to go once bind returns
        if (e1 != hipSuccess || e2 != hipSuccess || e3 != hipSuccess || e4 != hipSuccess) {
            hipFree(m->d_data); hipFree(m->d_indices); hipFree(m->d_indptr);
            delete m;
            return fail(c, ILLICO_ERR_HIP, "upload of a bound matrix failed");
        }
        c->h2d_input_bytes += (int64_t)((size_t)nnz * (vsz + isz) + (size_t)n_ptr * isz);
    }
    look_at_row_order(c, m);
    c->bound.push_back(m);
    *out = m;
    return ILLICO_OK;
}

extern "C" int illico_csr_bind(illico_ctx *c, const void *data, int dtype, const void *indices, const void *indptr, int idx_dtype,
                               int64_t n_rows, int64_t n_cols, int flags, illico_matrix **out) {
    return sparse_bind(c, true, data, dtype, indices, indptr, idx_dtype, n_rows, n_cols, flags, out);
}
extern "C" int illico_csc_bind(illico_ctx *c, const void *data, int dtype, const void *indices, const void *indptr, int idx_dtype,
                               int64_t n_rows, int64_t n_cols, int flags, illico_matrix **out) {
    return sparse_bind(c, false, data, dtype, indices, indptr, idx_dtype, n_rows, n_cols, flags, out);
}
// columns [0, W) of three [G][src_ld] planes that lie `plane` doubles apart -> three planes of pitch dst_ld (run_bound_ahead: a chunk's slice of a window)
static __global__ __launch_bounds__(256) void k_copy_plane_slices(const double *__restrict__ src, long long plane, long long src_ld, double *__restrict__ d0,
                                                                  double *__restrict__ d1, double *__restrict__ d2, long long dst_ld, int W, int G) {
    for (int g = blockIdx.x; g < G; g += gridDim.x)
        for (int j = threadIdx.x; j < W; j += 256) {
            const size_t i = (size_t)g * src_ld + j, o = (size_t)g * dst_ld + j;
            d0[o] = src[i];
            d1[o] = src[plane + i];
            d2[o] = src[2 * plane + i];
        }
}

// "bound_ahead_genes" = A: genes [col_lb, col_ub), fewer than A, of a bound CSR matrix.  The aligned window of A genes that holds them
// (or the A genes from col_lb on, when they straddle a boundary) is computed ONCE into planes of the context's own -- two such windows
// are kept, the older one is replaced -- and every call for genes inside it is a copy of its slice: 32 calls of 256 genes at C3 shape
// cost 7.2 ms as 32 passes over the rows, 2.3 ms this way.  A window belongs to (matrix, groups, flags, alternative).
static int run_bound_ahead(illico_ctx *c, const illico_matrix *m, int64_t col_lb, int64_t col_ub, int flags, int alternative,
                           double *out_p, double *out_u, double *out_fc, int64_t out_ld) {
    const int64_t A = c->bound_ahead_genes, W = col_ub - col_lb;
    const size_t G = (size_t)c->n_groups;
    const int kflags = flags & (ILLICO_FLAG_LOG1P | ILLICO_FLAG_CONTINUITY | ILLICO_FLAG_TIE_CORRECT);
    int rc;
    HIPCHK(c, hipSetDevice(c->device));
    illico_ctx::AheadWindow *w = nullptr;
    for (auto &a : c->ahead)
        if (a.m == m && a.gen == c->groups_gen && a.flags == kflags && a.alternative == alternative && a.lb <= col_lb && col_ub <= a.ub) w = &a;
    if (!w) {
        w = c->ahead[0].stamp <= c->ahead[1].stamp ? &c->ahead[0] : &c->ahead[1];
        for (auto &a : c->ahead) if (!a.m) w = &a; // (a free one first)
        int64_t lb = (col_lb / A) * A, ub = std::min(lb + A, m->n_cols);
        if (col_ub > ub) { lb = col_lb; ub = std::min(lb + A, m->n_cols); }
        const size_t need = 3 * G * (size_t)(ub - lb) * sizeof(double);
        w->m = nullptr;
        if (w->cap < need) {
            if (w->planes) hipFree(w->planes);
            w->planes = nullptr; w->cap = 0;
            HIPCHK(c, hipMalloc((void **)&w->planes, need));
            w->cap = need;
        }
        const size_t plane = G * (size_t)(ub - lb);
        if ((rc = run_sparse(c, true, m->d_data, m->dtype, m->d_indices, m->d_indptr, m->idx_dtype, m->n_rows, m->n_cols, lb, ub,
                             kflags | ILLICO_FLAG_INPUT_DEVICE | ILLICO_FLAG_OUTPUT_DEVICE, alternative, w->planes, w->planes + plane, w->planes + 2 * plane, ub - lb)))
            return rc;
        w->m = m; w->gen = c->groups_gen; w->flags = kflags; w->alternative = alternative; w->lb = lb; w->ub = ub;
    } else { // (a deferred call of another kind may still be in flight: completed first, as every entry point does)
        PendingDense prev = c->pend;
        c->pend.on = false;
        if ((rc = resolve_pending(c, prev))) return rc;
    }
    w->stamp = ++c->ahead_clock;
    OutPlanes o;
    if ((rc = begin_outputs(c, flags, W, out_p, out_u, out_fc, out_ld, &o))) return rc;
    const int64_t cw = w->ub - w->lb;
    const size_t plane = G * (size_t)cw;
    double *const dst[3] = {o.p, o.u, o.fc};
    // one launch for the three slices (three strided copies cost three launches per chunk: 1 ms of the 32 chunks' 2.5)
    hipLaunchKernelGGL(k_copy_plane_slices, dim3((unsigned)std::min<size_t>(G, 65535)), dim3(256), 0, c->stream, (const double *)(w->planes + (col_lb - w->lb)), (long long)plane,
                       (long long)cw, dst[0], dst[1], dst[2], (long long)o.ld, (int)W, (int)G);
    HIPCHK(c, hipGetLastError());
    if (!o.staged) {
        if (!(flags & ILLICO_FLAG_DEFER)) HIPCHK(c, hipStreamSynchronize(c->stream));
        return ILLICO_OK;
    }
    return end_outputs(c, o, W, out_p, out_u, out_fc, out_ld);
}

extern "C" int illico_run_bound(illico_ctx *c, const illico_matrix *m, int64_t col_lb, int64_t col_ub, int flags, int alternative,
                                double *out_p, double *out_u, double *out_fc, int64_t out_ld) {
    if (!c || !m) return ILLICO_ERR_ARG;
    CTX_LOCK(c); // (recursive: held for the whole call, so that illico_matrix_release on another thread cannot free the arrays under it)
    if (m->owner != c || std::find(c->bound.begin(), c->bound.end(), m) == c->bound.end())
        return fail(c, ILLICO_ERR_ARG, "the matrix handle does not belong to this context (or was released)");
    const int keep = ILLICO_FLAG_LOG1P | ILLICO_FLAG_CONTINUITY | ILLICO_FLAG_TIE_CORRECT | ILLICO_FLAG_OUTPUT_DEVICE | ILLICO_FLAG_DEFER;
    // what was learnt about the rows' order when the matrix was bound: in order -- the group-major CSR pass need not ask again; not --
    // it is not for that pass
    const bool hold0 = c->hold_csr_counts;
    c->cur_sorted_known = m->is_csr && m->sorted == 1;
    if (m->is_csr && m->sorted == 0) c->hold_csr_counts = true;
    int rc;
    const int64_t A = c->bound_ahead_genes;
    if (A > 0 && m->is_csr && !c->tap && c->has_groups && col_lb >= 0 && col_lb < col_ub && col_ub <= m->n_cols && col_ub - col_lb < A && col_ub - col_lb < m->n_cols &&
        out_p && out_u && out_fc && out_ld >= col_ub - col_lb)
        rc = run_bound_ahead(c, m, col_lb, col_ub, flags & keep, alternative, out_p, out_u, out_fc, out_ld);
    else
        rc = run_sparse(c, m->is_csr, m->d_data, m->dtype, m->d_indices, m->d_indptr, m->idx_dtype, m->n_rows, m->n_cols, col_lb, col_ub,
                        (flags & keep) | ILLICO_FLAG_INPUT_DEVICE, alternative, out_p, out_u, out_fc, out_ld);
    c->cur_sorted_known = false;
    c->hold_csr_counts = hold0;
    return rc;
}
extern "C" int illico_matrix_release(illico_ctx *c, illico_matrix *m) {
    if (!c || !m) return ILLICO_ERR_ARG;
    CTX_LOCK(c);
    auto it = std::find(c->bound.begin(), c->bound.end(), m);
    if (it == c->bound.end() || m->owner != c) return fail(c, ILLICO_ERR_ARG, "the matrix handle does not belong to this context (or was released)");
    HIPCHK(c, hipSetDevice(c->device));
    int rc = resolve_pending(c); // a deferred call may still read the arrays
    HIPCHK(c, hipStreamSynchronize(c->stream));
    c->bound.erase(it);
    for (auto &a : c->ahead) if (a.m == m) a.m = nullptr;
    if (m->owns) { hipFree(m->d_data); hipFree(m->d_indices); hipFree(m->d_indptr); }
    delete m;
    return rc;
}

extern "C" int illico_matrix_touch(illico_ctx *c, illico_matrix *m) {
    if (!c || !m) return ILLICO_ERR_ARG;
    CTX_LOCK(c);
    if (m->owner != c || std::find(c->bound.begin(), c->bound.end(), m) == c->bound.end())
        return fail(c, ILLICO_ERR_ARG, "the matrix handle does not belong to this context (or was released)");
    HIPCHK(c, hipSetDevice(c->device));
    const int rc = resolve_pending(c); // (a deferred call on the arrays as they were completes first)
    for (auto &a : c->ahead) if (a.m == m) a.m = nullptr;
    look_at_row_order(c, m);
    return rc;
}

template <typename IdxT> static int csr_sorted_host(const IdxT *indices, const IdxT *indptr, int64_t n_rows) {
    for (int64_t r = 0; r < n_rows; ++r)
        for (int64_t k = (int64_t)indptr[r] + 1; k < (int64_t)indptr[r + 1]; ++k)
            if (indices[k] < indices[k - 1]) return 0;
    return 1;
}

extern "C" int illico_csr_indices_sorted(illico_ctx *c, const void *indices, const void *indptr, int idx_dtype,
                                         int64_t n_rows, int flags, int *out_sorted) {
    if (!c) return ILLICO_ERR_ARG;
    CTX_LOCK(c);
    if (!indices || !indptr || !out_sorted || n_rows < 0) return fail(c, ILLICO_ERR_ARG, "bad argument");
    if (idx_dtype != ILLICO_IDX_I32 && idx_dtype != ILLICO_IDX_I64) return fail(c, ILLICO_ERR_DTYPE, "unsupported index dtype code %d", idx_dtype);
    if (!(flags & ILLICO_FLAG_INPUT_DEVICE)) {
        *out_sorted = idx_dtype == ILLICO_IDX_I32 ? csr_sorted_host((const int32_t *)indices, (const int32_t *)indptr, n_rows)
                                                  : csr_sorted_host((const int64_t *)indices, (const int64_t *)indptr, n_rows);
        return ILLICO_OK;
    }
    HIPCHK(c, hipSetDevice(c->device));
    void *v;
    int rc = get_scratch(c, "flag", 16, &v);
    if (rc) return rc;
    int *d_bad = (int *)v;
    HIPCHK(c, hipMemsetAsync(d_bad, 0, 4, c->stream));
    const int grid = (int)std::min<int64_t>((n_rows + 3) / 4 + 1, 8192);
    if (idx_dtype == ILLICO_IDX_I32)
        hipLaunchKernelGGL((k_csr_sorted_check<int32_t>), dim3(grid), dim3(256), 0, c->stream, (const int32_t *)indices, (const int32_t *)indptr, (int)n_rows, d_bad);
    else
        hipLaunchKernelGGL((k_csr_sorted_check<int64_t>), dim3(grid), dim3(256), 0, c->stream, (const int64_t *)indices, (const int64_t *)indptr, (int)n_rows, d_bad);
    HIPCHK(c, hipGetLastError());
    int bad = 0;
    HIPCHK(c, hipMemcpyAsync(&bad, d_bad, 4, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    *out_sorted = bad ? 0 : 1;
    return ILLICO_OK;
}
