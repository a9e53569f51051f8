// libillico_hip: MI355X (gfx950) engine behind include/illico_hip.h.
// Host side: context, device scratch, gene batching, kernel launches, measurement hooks.
#include <cstring>
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <chrono>
#include <condition_variable>
#include <map>
#include <mutex>
#include <thread>
#include <string>
#include <vector>

#include "../../include/illico_hip.h"
#include "common.h"
#include "kernels_finalize.h"
#include "kernels_ovo.h"
#include "kernels_ovo_compact.h"
#include "kernels_ovo_counts.h"
#include "kernels_ovo_fused.h"
#include "kernels_ovr.h"
#include "kernels_sparse.h"
#include "kernels_csc_gene.h"
#include "kernels_csc_counts.h"
#include "kernels_csc_ovr.h"
#include "kernels_ovr_parts.h"
#include "kernels_sums.h"
#include "kernels_leftover.h"

// ---- profiled kernel ids ---------------------------------------------------------------------
enum {
    KID_TRANSPOSE = 0,
    KID_OVO_RANK,
    KID_OVO_COUNTS,
    KID_OVO_FUSED,
    KID_OVR_FUSED,
    KID_FUSED_REF,
    KID_FINALIZE,
    KID_OVR_SCAN,
    KID_SPARSE_SEG,
    KID_CSC_GENE,
    KID_GENE_TOTALS,
    KID_CSC_COUNTS,
    KID_CSC_OVR,
    KID_OVR_PART,
    KID_OVR_RANK_PARTS,
    KID_VALUE_SUMS,
    KID_OVO_FUSED_WIDE,
    KID_GROUP_COMPACT,
    KID_OVO_RANK_COMPACT,
    KID_OVR_COUNTS,
    KID_GATHER_COLS,
    KID_COUNT
};
static const char *kKernelNames[KID_COUNT] = {"k_transpose_permute", "k_ovo_rank", "k_ovo_counts", "k_ovo_fused", "k_ovr_fused",
                                              "k_fused_tables", "k_finalize", "k_ovr_gene", "k_sparse_seg", "k_csc_gene", "k_gene_totals", "k_csc_counts", "k_csc_ovr_gene", "k_ovr_partition", "k_ovr_rank_parts", "k_value_sums", "k_ovo_fused_wide", "k_group_compact", "k_ovo_rank_compact", "k_ovr_counts", "k_gather_columns"};

// A dense call made with ILLICO_FLAG_DEFER whose fused pass is in flight: which genes it could not take is known only once
// its route flags have reached the host; they are then recomputed by the two-pass routes (resolve_pending).
struct PendingDense {
    bool on = false;
    int kind = 0;                 // 0: dense (X, ld), 1: CSC (sp_*: the count-valued CSC pass, sparse_driver.h)
    const void *sp_data = nullptr, *sp_indices = nullptr, *sp_indptr = nullptr;
    int idx_dtype = 0;
    int64_t n_cols = 0;
    const void *X = nullptr;
    int dtype = 0, flags = 0, alternative = 0, slot = 0;
    int64_t N = 0, ld = 0, col_lb = 0, col_ub = 0, out_ld = 0;
    double *p = nullptr, *u = nullptr, *fc = nullptr;
};

// a sparse matrix bound to a context (illico_csr_bind / illico_csc_bind): device arrays, owned or adopted
struct illico_matrix {
    illico_ctx *owner = nullptr;
    bool is_csr = false, owns = false;
    int dtype = 0, idx_dtype = 0;
    int64_t n_rows = 0, n_cols = 0, nnz = 0;
    void *d_data = nullptr, *d_indices = nullptr, *d_indptr = nullptr;
};

struct ProfEvent {
    int kid;
    hipEvent_t a, b;
};

struct illico_ctx {
    // Every entry point that takes the context holds this lock for the whole call: two host threads driving ONE context
    // (the reference's joblib threads share one dispatcher, asymptotic_wilcoxon.py:236-241) are serialised, not raced.
    std::recursive_mutex mu;
    int device = 0;
    hipStream_t stream = nullptr;
    bool own_stream = false;
    std::string err;
    // groups
    bool has_groups = false;
    int64_t n_cells = 0, n_groups = 0, ref = -1;
    std::vector<int> h_counts;
    int64_t max_nonref = 0;
    int *d_codes = nullptr;       // [N] group code of each cell
    int *d_perm = nullptr;        // [N] cell index at group-contiguous position p
    int *d_posptr = nullptr;      // [G+1]
    int *d_pk_blk = nullptr;      // packed dense layout (kernels_ovo_compact.h): [pk_nblk+1] first group of each block, then [pk_nblk] first key slot
    int pk_nblk = 0, pk_ref_out = 0;
    int *d_pk_code = nullptr;     // padded dense layout (dense OVR): group code of every key slot (holes: 0, they hold zero keys)
    int64_t pk_stride = 0;        // keys per gene in the packed layout
    int64_t pk_len = 0;           // ... of which the blocks take the first pk_len (the padded dense layout's row length)
    int *d_counts = nullptr;      // [G]
    int *d_code_by_pos = nullptr; // [N] group code at position p
    u32 *d_hist_off = nullptr;    // [G+1] OVR one-pass histograms: words per lane before group g (16 per group of <= 255 cells, else 32)
    size_t hist_words = 0;        // d_hist_off[G]
    u16 *d_codes16 = nullptr;     // [N] d_codes as 16-bit values when G <= 65535 (half the cache lines per codes[row] gather), else null
    // options
    int64_t gene_batch = 0;
    int64_t scratch_bytes = 24ll << 30; // (illico_ctx_create: min(64 GiB, a quarter of the device's memory))
    bool no_counts_path = false;
    bool no_fused_path = false;
    bool no_ovr_packed_partition = false; // 1: dense OVR partitions the padded rows (every key) instead of the packed ones
    bool no_packed_dense = false;      // 1: dense OVO on continuous values takes the transpose + k_ovo_rank route (no group-wise packing)
    int packed_eq_buckets = -1;        // packed rank kernel: distribution-following bucket function; -1 = for references above 16384 cells
    bool no_csc_counts_windows = false; // 1: count-valued CSC with more groups than LDS holds tables for never takes k_csc_counts (windows of groups)
    bool no_csc_counts_wide = false;   // 1: count-valued CSC with more than 8 groups above 255 cells never takes k_csc_counts (16-bit cells)
    bool ovr_full_dump = false;        // 1: the one-pass OVR route dumps whole group histograms (A/B of the shortened dump)
    bool no_wide_gather = false;       // 1: the 256-value stage always runs over the window as it lies (never left to the host's gather)
    bool no_leftover_gather = false;   // 1: the genes the fused passes leave are recomputed as column runs of the input (no gather into a narrow matrix)
    bool no_fused_wide = false;        // 1: no second, 256-value pass of the fused OVO route (genes beyond 63 go to the two-pass routes)
    bool no_csc_regroup_lds = false;   // 1: the two-kernel CSC route regroups with k_csc_segment only
    bool no_csc_gene_path = false;
    int64_t ovr_parts_cap = 0;         // > 0: keys per part at most in the value-range parts route (tests: many small parts)
    bool no_ovo_ref_buckets = false;   // 1: the OVO sort route always sorts the reference column (no value-bucket form)
    bool no_ovr_parts_path = false;    // 1: dense OVR (any values) never takes the value-range parts route (k_ovr_partition + k_csc_ovr_gene)
    bool no_csc_ovr_gene_path = false; // 1: CSC OVR never takes the single-kernel LDS-sort route (k_csc_ovr_gene)
    bool csc_ovr_sorted_form = false;  // 1: k_csc_ovr_gene sorts every gene's keys in LDS (the form tie-heavy columns take) instead of bucketing them
    bool no_csc_counts_path = false;   // 1: count-valued CSC genes do not take the LDS-histogram kernel (k_csc_counts)
    bool no_csc_counts_mixed = false;  // 1: k_csc_counts with 8-bit cells for every value only (the form 4-bit overflows fall back to)
    bool no_ovr_one_pass = false;      // 1: dense OVR reads X twice (column histogram, then rank sums) instead of once
    bool no_csr_tile_gather = false;    // 1: CSR -> CSC always by the scatter form (k_csr_block_scatter), as for unsorted rows
    bool no_csr_transpose_path = false; // 1: CSR is regrouped by (gene, group) with global atomics instead of being transposed to CSC
    bool dense_window_f32 = false;      // 1: CSR dense windows hold float32 cells instead of bytes
    bool no_dense_window_path = false; // 1: CSR never goes through dense float32 windows + the fused kernels
    int fused_groups_per_wg = 0; // 0 = auto
    int ovr_hist_groups_per_wg = 0; // k_ovr_from_hists; 0 = auto
    bool profile = false;
    int profile_only = -1;        // >= 0: time this kernel id only (the others run without events around them)
    PendingDense pend;            // deferred dense call (ILLICO_FLAG_DEFER), see resolve_pending
    void *pend_pinned[2] = {nullptr, nullptr}; // its route flags arrive here (two buffers: the next call may be enqueued first)
    size_t pend_pinned_bytes[2] = {0, 0};
    hipEvent_t pend_event[2] = {nullptr, nullptr};
    int pend_next = 0;
    void *out_pin[2] = {nullptr, nullptr}; // pinned buffers + events of end_outputs (host planes)
    size_t out_pin_bytes = 0;
    hipEvent_t out_ev[2] = {nullptr, nullptr};
    void *pinned = nullptr;       // pinned host staging for small device -> host results
    size_t pinned_bytes = 0;
    int64_t h2d_input_bytes = 0;  // matrix bytes copied host -> device (illico_profile_input_bytes)
    std::vector<illico_matrix *> bound; // matrices bound to this context and not yet released
    struct HostStage *host_stage = nullptr; // pinned slots / copy stream of the host-window pipeline (dense driver)
    std::vector<ProfEvent> events;
    std::vector<hipEvent_t> event_pool;
    double prof_ms[KID_COUNT] = {0};
    int64_t prof_n[KID_COUNT] = {0};
    // grow-only scratch
    std::map<std::string, std::pair<void *, size_t>> scratch;
    // illico_rank_statistics: host arrays that receive the integer rank statistics of the two-pass routes instead of the
    // finalisation ([W][G] each, W = the call's column window)
    struct StatsTap { long long *two_u; u64 *tie; double *sum; } *tap = nullptr;
};

#define CTX_LOCK(c) std::lock_guard<std::recursive_mutex> ctx_lock__((c)->mu)

// The message of a failed call is kept per calling thread (and in the context, for single-threaded callers): a second
// thread's failure must not replace the text the first is about to read through illico_last_error.
static thread_local std::string t_err;
static thread_local const illico_ctx *t_err_ctx = nullptr;

static int fail(illico_ctx *c, int code, const char *fmt, ...) {
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

#define HIPCHK(ctx, call)                                                                              \
    do {                                                                                               \
        hipError_t e__ = (call);                                                                       \
        if (e__ != hipSuccess)                                                                         \
            return fail(ctx, e__ == hipErrorOutOfMemory ? ILLICO_ERR_OOM : ILLICO_ERR_HIP, "%s failed: %s (%s:%d)", \
                        #call, hipGetErrorString(e__), __FILE__, __LINE__);                            \
    } while (0)

static int get_scratch(illico_ctx *c, const char *name, size_t bytes, void **out) {
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
static hipEvent_t take_event(illico_ctx *c) {
    if (!c->event_pool.empty()) {
        hipEvent_t e = c->event_pool.back();
        c->event_pool.pop_back();
        return e;
    }
    hipEvent_t e = nullptr;
    if (hipEventCreateWithFlags(&e, hipEventDisableSystemFence) != hipSuccess) hipEventCreate(&e);
    return e;
}

struct ProfScope {
    illico_ctx *c;
    int kid;
    hipEvent_t a = nullptr, b = nullptr;
    bool on;
    ProfScope(illico_ctx *c_, int kid_) : c(c_), kid(kid_), on(c_->profile && (c_->profile_only < 0 || c_->profile_only == kid_)) {
        if (on) {
            a = take_event(c);
            b = take_event(c);
            hipEventRecord(a, c->stream);
        }
    }
    ~ProfScope() {
        if (on) {
            hipEventRecord(b, c->stream);
            c->events.push_back({kid, a, b});
        }
    }
};

static void drain_events(illico_ctx *c) {
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

static int resolve_pending(illico_ctx *c); // completes a deferred dense call (defined with the dense driver)
static void free_host_stage(illico_ctx *c); // (dense driver)

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
    for (int **p : {&c->d_codes, &c->d_perm, &c->d_posptr, &c->d_counts, &c->d_code_by_pos, &c->d_pk_blk, &c->d_pk_code}) {
        if (*p) hipFree(*p);
        *p = nullptr;
    }
    if (c->d_codes16) hipFree(c->d_codes16);
    c->d_codes16 = nullptr;
    if (c->d_hist_off) hipFree(c->d_hist_off);
    c->d_hist_off = nullptr;
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
    else if (!strcmp(key, "no_csc_counts_windows")) c->no_csc_counts_windows = value != 0;
    else if (!strcmp(key, "no_csc_counts_wide")) c->no_csc_counts_wide = value != 0;
    else if (!strcmp(key, "ovr_full_dump")) c->ovr_full_dump = value != 0;
    else if (!strcmp(key, "no_packed_dense")) c->no_packed_dense = value != 0;
    else if (!strcmp(key, "packed_eq_buckets")) c->packed_eq_buckets = (int)value;
    else if (!strcmp(key, "no_ovr_packed_partition")) c->no_ovr_packed_partition = value != 0;
    else if (!strcmp(key, "no_csc_counts_path")) c->no_csc_counts_path = value != 0;
    else if (!strcmp(key, "no_csc_counts_mixed")) c->no_csc_counts_mixed = value != 0;
    else if (!strcmp(key, "no_csc_regroup_lds")) c->no_csc_regroup_lds = value != 0;
    else if (!strcmp(key, "no_csc_gene_path")) c->no_csc_gene_path = value != 0;
    else if (!strcmp(key, "ovr_parts_cap")) c->ovr_parts_cap = value;
    else if (!strcmp(key, "no_ovo_ref_buckets")) c->no_ovo_ref_buckets = value != 0;
    else if (!strcmp(key, "no_ovr_parts_path")) c->no_ovr_parts_path = value != 0;
    else if (!strcmp(key, "no_csc_ovr_gene_path")) c->no_csc_ovr_gene_path = value != 0;
    else if (!strcmp(key, "csc_ovr_sorted_form")) c->csc_ovr_sorted_form = value != 0;
    else if (!strcmp(key, "no_ovr_one_pass")) c->no_ovr_one_pass = value != 0;
    else if (!strcmp(key, "no_ovr_library_sort")) (void)value; // accepted and ignored: there is no library sort any more
    else if (!strcmp(key, "no_csr_tile_gather")) c->no_csr_tile_gather = value != 0;
    else if (!strcmp(key, "no_csr_transpose_path")) c->no_csr_transpose_path = value != 0;
    else if (!strcmp(key, "dense_window_f32")) c->dense_window_f32 = value != 0;
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
        // Beyond that the reference's int64 wraps silently; this build refuses instead of returning wrapped values.
        const int64_t n_test = ref < 0 ? n_cells : counts[ref] + max_nonref;
        if (n_test > 2097151)
            return fail(c, ILLICO_ERR_UNSUPPORTED, "%lld cells in one test: n(n-1)(n+1) and the tie sums overflow 64-bit integers beyond 2097151 cells (the reference's int64 arithmetic wraps there, utils/math.py:95)", (long long)n_test);
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
        auto close = [&](int64_t end) { g1.push_back((int)end); pos += (rows + 63) & ~63ll; open = false; };
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
        c->pk_ref_out = (int)pos;
        c->pk_len = pos;
        c->pk_stride = pos + (ref >= 0 ? ((counts[ref] + 63) & ~63ll) : 0) + 64;
        HIPCHK(c, hipMalloc((void **)&c->d_pk_blk, packed.size() * sizeof(int)));
        HIPCHK(c, hipMemcpy(c->d_pk_blk, packed.data(), packed.size() * sizeof(int), hipMemcpyHostToDevice));
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
    c->h_counts = cnt;
    c->n_cells = n_cells;
    c->n_groups = n_groups;
    c->ref = ref;
    c->max_nonref = max_nonref;
    c->has_groups = true;
    return ILLICO_OK;
}

} // extern "C"

// ============================================================================================
// dense driver
// ============================================================================================
static const size_t kMaxLds = 160 * 1024;
static const int kOvoThreads = 512;

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
template <typename KeyT>
static int launch_seg_value_sums(illico_ctx *c, const KeyT *Xs, const u32 *seg, int nb, int dtype, int flags, double *ssum) {
    SegSumsParams P;
    P.Xs = Xs; P.seg_ptr = seg; P.nb = nb; P.G = (int)c->n_groups; P.dt = dtype; P.is_log1p = (flags & ILLICO_FLAG_LOG1P) ? 1 : 0; P.out_sum = ssum;
    ProfScope ps(c, KID_VALUE_SUMS);
    hipLaunchKernelGGL((k_seg_value_sums<KeyT>), dim3(nb), dim3(SUMS_NT), 0, c->stream, P);
    HIPCHK(c, hipGetLastError());
    return ILLICO_OK;
}
template <typename KeyT>
static int launch_group_sums_rows(illico_ctx *c, const KeyT *Xt, int64_t stride, int nb, int dtype, int flags, double *ssum) {
    ProfScope ps(c, KID_VALUE_SUMS);
    hipLaunchKernelGGL((k_group_sums_rows<KeyT>), dim3(nb), dim3(SUMS_NT), 0, c->stream, Xt, (long long)stride, nb, (const int *)c->d_posptr,
                       (int)c->n_groups, dtype, (flags & ILLICO_FLAG_LOG1P) ? 1 : 0, ssum);
    HIPCHK(c, hipGetLastError());
    return ILLICO_OK;
}
#include "ovr_driver.h"

template <typename KeyT> static size_t ovo_lds_bytes(int ref_cap, bool runend, int nt, bool buckets = false) {
    size_t nw = nt / 64;
    size_t b = ((((size_t)ref_cap + 4) * sizeof(KeyT)) + 15) & ~(size_t)15;
    if (runend) b += ovo_runend_bytes(ref_cap, buckets);
    b += nw * 256 * sizeof(KeyT) + nw * 256 * 4;
    b += nw * 8 * 2 + 16 + 48;
    return b;
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

// Does the in-LDS sort route (k_ovo_rank) hold these sizes?  (reference column in LDS, groups <= 1024 keys)
template <typename KeyT> static bool ovo_sort_route_fits(int64_t max_ref_nnz, int64_t max_grp_nnz) {
    int ref_cap = (int)std::max<int64_t>(max_ref_nnz, 1);
    bool runend = ref_cap <= 65535 && ovo_lds_bytes<KeyT>(ref_cap, true, kOvoThreads) <= kMaxLds;
    return max_grp_nnz <= 1024 && ovo_lds_bytes<KeyT>(ref_cap, runend, kOvoThreads) <= kMaxLds;
}

// table size of the two-pass histogram route (k_ovo_counts): 4096 values with 8-bit multiplicities while no ranked group exceeds 255
// cells, else 2048 with 16-bit ones; the ingest kernels flag genes against the same limit
static int ovo_counts_limit(const illico_ctx *c) { return c->max_nonref <= 255 ? COUNTS_R8 : COUNTS_R; }

struct OvoGlobalBufs { // scratch of the global-sort fallback (same element count as the key buffer)
    void *kb = nullptr;
    u32 *va = nullptr, *vb = nullptr;
};

// flags: per-gene routing word written by the ingest kernels (0 = count-valued gene -> k_ovo_counts,
// non-zero -> sort route); nullptr routes every gene through the sort route.  The sort route is k_ovo_rank when
// the reference column and the groups fit LDS / registers, else the per-gene global radix sort (k_ovr_gene in
// OVO mode), which has no size limit.
template <typename KeyT>
static int launch_ovo(illico_ctx *c, OvoParams P, int64_t max_ref_nnz, int64_t max_grp_nnz, const u32 *flags,
                      const OvoGlobalBufs *gb, bool sparse) {
    if (flags) { // (the ingest kernels flagged with the same limit: ovo_counts_limit)
        ProfScope ps(c, KID_OVO_COUNTS);
        if (ovo_counts_limit(c) == COUNTS_R8) hipLaunchKernelGGL((k_ovo_counts<KeyT, COUNTS_R8, 8>), dim3(P.n_genes), dim3(COUNTS_NT), 0, c->stream, P, flags);
        else hipLaunchKernelGGL((k_ovo_counts<KeyT, COUNTS_R, 16>), dim3(P.n_genes), dim3(COUNTS_NT), 0, c->stream, P, flags);
        HIPCHK(c, hipGetLastError());
    }
    if (!ovo_sort_route_fits<KeyT>(max_ref_nnz, max_grp_nnz)) {
        if (!gb || !gb->kb) return fail(c, ILLICO_ERR_UNSUPPORTED, "internal: global-sort scratch missing");
        OvrParams Q;
        Q.keys_a = (void *)P.Xs; Q.keys_b = gb->kb; Q.vals_a = gb->va; Q.vals_b = gb->vb;
        Q.code_by_pos = sparse ? nullptr : c->d_code_by_pos; Q.seg_ptr = P.seg_ptr; Q.stride = P.gene_stride;
        Q.pos_ptr = P.pos_ptr; Q.counts = P.counts; Q.G = P.G; Q.n_genes = P.n_genes; Q.dt = P.dt; Q.is_log1p = P.is_log1p;
        Q.n_cells = c->n_cells; Q.ref = P.ref; Q.gene_flags = flags;
        Q.out_2u = P.out_2u; Q.out_tie = P.out_tie; Q.out_sum = P.out_sum;
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

// ---- packed dense OVO route (kernels_ovo_compact.h) ----
// LDS sizing of the packed rank kernel: key slots for the reference's NON-ZERO keys and the bucket count (2^lg, half a byte each).
// A reference whose every cell fits beside 2^17 buckets gets one slot per cell; a larger one gets the slots that fit beside 2^16
// buckets -- an expression matrix is mostly zeros, so the non-zeros of a 33 000-cell reference still fit -- and a gene whose
// non-zeros exceed the slots is left to k_ovo_rank by the kernel (as the tie-heavy ones are).
template <typename KeyT> static void packed_ref_sizing(int64_t n_ref, int *cap, int *lg) {
    if (ocr_lds_bytes((int)n_ref, 17, sizeof(KeyT)) <= kMaxLds) { *cap = (int)n_ref; *lg = 17; return; }
    // 2^17 buckets while 55 % of the reference's cells would still fit the slots left beside them, else 2^16 and more slots
    const size_t fixed17 = ocr_lds_bytes(0, 17, sizeof(KeyT));
    const int64_t cap17 = fixed17 < kMaxLds ? (int64_t)((kMaxLds - fixed17) / sizeof(KeyT)) - 8 : 0;
    if (cap17 > 0 && n_ref * 55 <= cap17 * 100) { *lg = 17; *cap = (int)std::min<int64_t>(n_ref, cap17); return; }
    *lg = 16;
    const size_t fixed = ocr_lds_bytes(0, 16, sizeof(KeyT));
    *cap = (int)std::min<int64_t>(n_ref, (int64_t)((kMaxLds - fixed) / sizeof(KeyT)) - 8);
}
// Sizes the route holds: reference of at most 65535 cells whose keys fit k_ovo_rank's LDS (it takes the tie-heavy genes and those
// whose non-zeros exceed the packed kernel's key slots), other groups of at most 1024 cells (k_ovo_rank's register form).
template <typename KeyT> static bool packed_route_fits(const illico_ctx *c) {
    if (c->ref < 0 || c->no_packed_dense) return false;
    const int64_t n_ref = c->h_counts[c->ref];
    return n_ref >= 1 && n_ref <= 65535 && c->max_nonref <= 1024 && ovo_sort_route_fits<KeyT>(n_ref, c->max_nonref);
}

// k_group_compact over one gene batch; pack = false: the padded dense layout (every key kept, sums only)
template <typename InT, typename KeyT>
static int launch_group_compact(illico_ctx *c, GroupCompactParams Q, int nb, int flags, bool pack) {
    constexpr int VEC = 16 / (int)sizeof(InT);
    const bool aligned = ((uintptr_t)Q.X % 16 == 0) && (Q.ld % VEC == 0) && (Q.col0 % VEC == 0);
    const bool lg = flags & ILLICO_FLAG_LOG1P;
    const dim3 grid(((Q.nseg + 7) & ~7) + Q.nblk, (nb + 63) / 64);
    ProfScope ps(c, KID_GROUP_COMPACT);
#define GC_LAUNCH(V, L, K) hipLaunchKernelGGL((k_group_compact<InT, KeyT, V, L, K>), grid, dim3(GCMP_NT), 0, c->stream, Q)
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
                          long long *s2u, u64 *stie, double *ssum) {
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
    {
        GroupCompactParams Q;
        Q.X = X; Q.ld = ld; Q.col0 = col0; Q.ncols = nb; Q.perm = c->d_perm; Q.pos_ptr = c->d_posptr; Q.G = G; Q.ref = ref; Q.nseg = nseg;
        Q.blk_g0 = c->d_pk_blk; Q.blk_g1 = c->d_pk_blk + c->pk_nblk; Q.blk_out = c->d_pk_blk + 2 * c->pk_nblk; Q.nblk = c->pk_nblk; Q.ref_out = c->pk_ref_out;
        Q.Xt = Xt; Q.xt_stride = stride; Q.nnz = nnz; Q.gofs = gofs; Q.blk_cnt = nullptr; Q.out_sum = ssum; Q.seg_nnz = seg_nnz; Q.seg_sum = seg_sum;
        if ((rc = launch_group_compact<InT, KeyT>(c, Q, nb, flags, true))) return rc;
    }
    {
        OvoCompactParams C;
        C.Xs = Xt; C.gene_stride = stride; C.counts = c->d_counts; C.nnz = nnz; C.gofs = gofs; C.ref_out = c->pk_ref_out; C.seg_nnz = seg_nnz; C.seg_sum = seg_sum;
        C.out_sum = ssum; C.G = G; C.ref = ref; C.n_genes = nb; C.nseg = nseg;
        packed_ref_sizing<KeyT>(n_ref, &C.ref_cap, &C.nbk_lg);
        C.out_2u = s2u; C.out_tie = stie; C.route = route;
        const size_t lds = ocr_lds_bytes(C.ref_cap, C.nbk_lg, sizeof(KeyT));
        // large references: the bucket function follows the reference's distribution (a crowded stretch of values would otherwise
        // fill buckets beyond three keys and send whole table words to key-by-key walks); "packed_eq_buckets" = 0 / 1 forces
        const bool eq = c->packed_eq_buckets >= 0 ? c->packed_eq_buckets != 0 : n_ref > 16384;
        auto kern = eq ? k_ovo_rank_compact<KeyT, true> : k_ovo_rank_compact<KeyT, false>;
        HIPCHK(c, hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        ProfScope ps(c, KID_OVO_RANK_COMPACT);
        hipLaunchKernelGGL(kern, dim3(nb), dim3(OCR_NT), lds, c->stream, C);
        HIPCHK(c, hipGetLastError());
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

// the histogram path needs integer value sums (no expm1) and 16-bit group bins
static bool counts_path_allowed(const illico_ctx *c, int flags) {
    return !(flags & ILLICO_FLAG_LOG1P) && c->ref >= 0 && c->max_nonref <= 65535 && !c->no_counts_path;
}
// fused single-pass routes (OVO and OVR): integer value sums (no expm1), 16-bit running multiplicities (OVO),
// 32-bit chunk partial sums (n_cells < 2^25)
static bool fused_path_allowed(const illico_ctx *c, int flags) {
    return !(flags & ILLICO_FLAG_LOG1P) && c->max_nonref <= 65535 && c->n_cells < (1ll << 25) && !c->no_counts_path && !c->no_fused_path;
}

static int launch_finalize(illico_ctx *c, const long long *s2u, const u64 *stie, const double *ssum, const double *gene_total,
                           int nb, int flags, int alternative, double *out_p, double *out_u, double *out_fc, int64_t out_ld,
                           int64_t col_off, const int *col_map = nullptr, bool packed = false) {
    FinalizeParams F;
    F.col_map = col_map;
    F.packed = packed ? 1 : 0;
    F.in_2u = s2u; F.in_tie = stie; F.in_sum = ssum; F.gene_total = gene_total;
    F.counts = c->d_counts; F.G = (int)c->n_groups; F.ref = (int)c->ref; F.nb = nb; F.n_cells = c->n_cells;
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

static size_t dtype_size(int dt) { return (dt == ILLICO_F32 || dt == ILLICO_I32) ? 4 : 8; }

struct OutPlanes {
    double *p, *u, *fc; // device
    int64_t ld;
    bool staged;
};

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

#define FUSED_RT 64

// Fused single-pass route over genes [b0, b0+nb): writes final planes for every gene it can take and sets
// h_flags[j] != 0 for the others (1 / 3: left to the two-pass routes; 2: done by the 256-value stage).  h_flags[nb] (also word nb of
// the deferred call's pinned flags) != 0: the 256-value stage was left to the host (k_wide_decide; only with max_gather > 0).
// init_flags (host, [nb]): the 256-value stage ALONE, for the genes marked 1 there (run_leftovers: a narrow matrix of gathered columns).
template <typename InT>
static int run_fused_ovo(illico_ctx *c, const void *X, int64_t ld, int64_t b0, int nb, int flags, int alternative,
                         const OutPlanes &o, int64_t col_off, std::vector<u32> &h_flags, int defer_slot = -1, bool probe = false,
                         int64_t max_gather = 0, const u32 *init_flags = nullptr) {
    constexpr int RT = FUSED_RT;
    const bool ovr = c->ref < 0;
    void *v;
    int rc;
    const size_t nb64 = ((size_t)nb + 63) & ~(size_t)63; // the cumulative tables are stored per 64-gene tile
    size_t bytes = nb64 * (RT + 1) * 4 + (size_t)nb * 8 * 2 + (size_t)nb * 4 + (size_t)nb * RT * 4 + 64;
    if ((rc = get_scratch(c, "fused_tables", bytes, &v))) return rc;
    FusedParams P;
    P.X = X; P.ld = ld; P.col0 = b0; P.ncols = nb; P.perm = c->d_perm; P.pos_ptr = c->d_posptr; P.counts = c->d_counts;
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
    if (probe && !wide_only) { // OVR on device-resident input: which genes are count-valued at all is found on the device (the OVO pass has
        // k_fused_ref, which reads every reference row first)
        ProfScope ps(c, KID_FUSED_REF);
        hipLaunchKernelGGL((k_fused_probe<InT, RT>), dim3(tiles), dim3(256), 0, c->stream, P);
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
    if (!ovr) {
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
            else hipLaunchKernelGGL((k_ovo_fused<InT, RT, false, 16>), main_grid, dim3(FUSED_NT), lds16, c->stream, P);
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
        if (!c->no_ovr_one_pass && hist_bytes <= (size_t)c->scratch_bytes) {
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
    if (c->pinned_bytes < (size_t)nb * 4 + 4) {
        if (c->pinned) hipHostFree(c->pinned);
        c->pinned = nullptr;
        c->pinned_bytes = 0;
        HIPCHK(c, hipHostMalloc(&c->pinned, (size_t)nb * 4 + 4096, hipHostMallocDefault));
        c->pinned_bytes = (size_t)nb * 4 + 4096;
    }
    HIPCHK(c, hipMemcpyAsync(c->pinned, P.gene_flags, (size_t)nb * 4, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipMemcpyAsync((u32 *)c->pinned + nb, skipw, 4, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    h_flags.assign((const u32 *)c->pinned, (const u32 *)c->pinned + nb + 1);
    return ILLICO_OK;
}

// column runs [first, second) of the flagged genes of a window starting at column w0
static void flagged_runs(const u32 *hf, int64_t wn, int64_t w0, std::vector<std::pair<int64_t, int64_t>> &runs) {
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

// Of 64k evenly spaced cells of a HOST matrix window: is it count-valued at all?  (The fused route over a host matrix copies
// the window up; on normalised data that copy would be made twice, once for nothing.)
template <typename InT> static bool host_window_is_count_valued(const InT *X, int64_t ld, int64_t col_lb, int64_t N, int64_t W) {
    const int64_t n_samples = std::min<int64_t>(N * W, 1 << 16);
    int64_t bad = 0;
    for (int64_t i = 0; i < n_samples; ++i) {
        const int64_t k = (int64_t)((double)i * (double)(N * W) / (double)n_samples);
        const int64_t r = k / W, j = k - r * W;
        const InT v = X[r * ld + col_lb + j];
        if (!(v >= (InT)0 && v < (InT)(1 << 24) && (InT)(int)v == v)) ++bad;
    }
    return (double)bad <= 0.02 * (double)n_samples;
}

// ---- host-resident dense input: a three-stage pipeline over column windows ----------------------------------------------
// A pageable 2-D copy of the whole window (what this path did before) moves 9.6 GB at ~43 GB/s through the driver's own
// staging and nothing overlaps it.  Here: (1) HS_THREADS host threads copy window k + 1's row pieces out of the caller's
// matrix into a PINNED slot, (2) the copy stream moves window k's slot to the device at the link's rate, (3) the context's
// stream runs the fused pass on window k - 1 -- all three at once, three slots deep.  Slot j serves the windows k = j mod 3: its
// pinned half is free once its upload has completed, its device half once the pass over it has (events both ways).
#define HS_SLOTS 3
#define HS_THREADS 12
struct HostStage {
    void *pin[HS_SLOTS] = {nullptr, nullptr, nullptr};
    size_t pin_bytes = 0;
    int *lists = nullptr;        // pinned: column lists of the flagged genes, window after window (gathered leftovers)
    size_t lists_ints = 0;
    hipStream_t copy = nullptr;
    hipEvent_t up[HS_SLOTS] = {nullptr, nullptr, nullptr}, done[HS_SLOTS] = {nullptr, nullptr, nullptr};
};
static HostStage *host_stage_of(illico_ctx *c) { // (one per context, freed with it)
    if (!c->host_stage) c->host_stage = new HostStage();
    return c->host_stage;
}
static void free_host_stage(illico_ctx *c) {
    HostStage *hs = c->host_stage;
    if (!hs) return;
    if (hs->copy) { hipStreamSynchronize(hs->copy); hipStreamDestroy(hs->copy); }
    if (hs->lists) hipHostFree(hs->lists);
    for (int j = 0; j < HS_SLOTS; ++j) {
        if (hs->pin[j]) hipHostFree(hs->pin[j]);
        if (hs->up[j]) hipEventDestroy(hs->up[j]);
        if (hs->done[j]) hipEventDestroy(hs->done[j]);
    }
    delete hs;
    c->host_stage = nullptr;
}

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
    std::thread producer([&] {
        hipSetDevice(device);
        const int T = (int)std::max<int64_t>(1, std::min<int64_t>(HS_THREADS, N / 4096 + 1));
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
            for (int t = 1; t < T; ++t)
                pool.emplace_back([=] {
                    for (int64_t r = N * t / T; r < N * (t + 1) / T; ++r) memcpy(dst + (size_t)r * wn, X + (size_t)r * ld + w0, piece);
                });
            for (int64_t r = 0; r < N / T; ++r) memcpy(dst + (size_t)r * wn, X + (size_t)r * ld + w0, piece);
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

template <typename InT, typename KeyT>
static int run_dense_twopass(illico_ctx *c, const void *X, int dtype, int64_t N, int64_t ld, int64_t col_lb, int64_t col_ub, int flags,
                             int alternative, const OutPlanes &o, std::vector<std::pair<int64_t, int64_t>> runs, const int *col_map = nullptr,
                             bool prefer_counts = false);

// The genes the fused passes of a DEVICE-resident window [col_lb, col_ub) left behind (hf[j] = 1 / 3).  Few and scattered (a count
// matrix's highly expressed genes): gathered into a narrow matrix of their own and computed as ONE window whose results
// k_finalize scatters back through a column map (kernels_leftover.h).  Many (normalised data: every gene): the column runs, as before.
// wide_skipped: the device left the 256-value stage to us (k_wide_decide): it runs on the narrow matrix first (the genes flagged 1),
// its finished columns are copied into the caller's planes, and what it leaves is gathered once more out of the narrow matrix.
// outer (host, [W]): the window is itself such a narrow matrix -- column j of it is column outer[j] of the caller's planes.
template <typename InT, typename KeyT>
static int run_leftovers(illico_ctx *c, const void *X, int dtype, int64_t N, int64_t ld, int64_t col_lb, int64_t col_ub, int flags, int alternative,
                         const OutPlanes &o, const u32 *hf, bool wide_skipped = false, const int *outer = nullptr) {
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
    if ((rc = get_scratch(c, outer ? "xleft2_cols" : "xleft_cols", (size_t)n * 12, &v))) return rc;
    int *d_src = (int *)v, *d_dst = d_src + n;
    u32 *d_flags2 = (u32 *)(d_dst + n);
    HIPCHK(c, hipMemcpyAsync(d_src, src.data(), (size_t)n * 4, hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipMemcpyAsync(d_dst, dst.data(), (size_t)n * 4, hipMemcpyHostToDevice, c->stream));
    {
        ProfScope ps(c, KID_GATHER_COLS);
        hipLaunchKernelGGL((k_gather_columns<InT>), dim3((unsigned)((N + 63) / 64)), dim3(256), 0, c->stream, (const InT *)X, (long long)ld, (int)N,
                           (const int *)d_src, (int)n, (int)n_pad, xl, (long long)n_pad, 0ll);
        HIPCHK(c, hipGetLastError());
    }
    HIPCHK(c, hipStreamSynchronize(c->stream)); // (the host lists go out of scope)
    const int lflags = (flags | ILLICO_FLAG_INPUT_DEVICE) & ~ILLICO_FLAG_DEFER;
    if (wide_skipped) {
        std::vector<u32> init((size_t)n), hf2;
        bool any = false;
        for (int64_t j = 0; j < n; ++j) { init[j] = hf[src[j] - col_lb] == 1u ? 1u : 3u; any = any || init[j] == 1u; }
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

template <typename InT, typename KeyT>
static int run_dense_t(illico_ctx *c, const void *X, int dtype, int64_t N, int64_t ld, int64_t col_lb, int64_t col_ub, int flags,
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
                             bool prefer_counts) {
    const int G = (int)c->n_groups;
    const bool ovr = c->ref < 0;
    const bool in_dev = flags & ILLICO_FLAG_INPUT_DEVICE;
    prefer_counts = prefer_counts && !(flags & ILLICO_FLAG_LOG1P) && !c->no_counts_path && (ovr || counts_path_allowed(c, flags));
    // dense OVO: group-wise packing + look-ups (kernels_ovo_compact.h) whenever the sizes allow; it has no histogram side path
    // (count-valued genes reach this function only when the fused route is off, or as gathered leftovers: prefer_counts) and holds
    // ties exactly
    const bool packed = !ovr && !prefer_counts && packed_route_fits<KeyT>(c);
    // dense OVR: the transposition with the group sums folded in (k_group_compact keeping every key: padded dense layout)
    const bool padded = ovr && !prefer_counts && !c->no_packed_dense && c->pk_nblk > 0 && c->max_nonref <= 65535 && c->pk_stride < (1ll << 31);
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
            if ((rc = run_ovo_packed<InT, KeyT>(c, src, src_ld, src_col0, nb, (int)N, Xt, stride, dtype, flags, s2u, stie, ssum))) return rc;
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
        const bool ovr_packed = padded && !c->no_ovr_packed_partition && !c->no_ovr_parts_path && G <= 65535;
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
    return ILLICO_OK;
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
static int resolve_pending(illico_ctx *c) {
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
    if (col_ub == col_lb) return ILLICO_OK;
    illico_ctx::StatsTap tap{(long long *)out_two_u, (u64 *)out_tie_sum, out_value_sum};
    c->tap = &tap;
    OutPlanes none{nullptr, nullptr, nullptr, 0, false};
    rc = run_dense_any(c, X, dtype, n_rows, ld, col_lb, col_ub, flags & (ILLICO_FLAG_LOG1P | ILLICO_FLAG_INPUT_DEVICE), 0, none);
    c->tap = nullptr;
    return rc;
}

#include "sparse_driver.h"
