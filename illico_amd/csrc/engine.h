// Host-side shared definitions of libillico_hip: the context, its helpers, and the declarations that tie the translation
// units together.  core.hip holds the context / C-ABI / dispatch; keyed_*.hip the launchers that depend on the key type only;
// dense_*.hip / sparse_*.hip one value type each (explicit instantiations of dense_driver.h / sparse_driver.h).
#pragma once
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
#include "kernels_group_hists.h"
#include "kernels_ovr.h"
#include "kernels_sparse.h"
#include "kernels_csc_gene.h"
#include "kernels_csc_counts.h"
#include "kernels_csc_ovr.h"
#include "kernels_ovr_parts.h"
#include "kernels_sums.h"
#include "kernels_leftover.h"
#include "kernels_csr_counts.h"

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
    KID_CSR_COUNTS,
    KID_DENSIFY,
    KID_GROUP_HISTS,
    KID_COUNT
};
extern const char *const kKernelNames[KID_COUNT];

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
    bool is_csr = false;          // kind 1: the arrays are CSR (the group-major count pass, kernels_csr_counts.h)
    bool sorted_known = false;    // kind 1, CSR: the matrix was bound and its rows found in order (illico_ctx::cur_sorted_known when the call was made)
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
    int sorted = -1;              // CSR: 1 = every row's column indices ascend (looked at once, when the matrix is bound), 0 = not, -1 = not looked at
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
    bool big_n = false;           // OVR over more than 2^21 - 1 cells: sparse input only, on the routes whose arithmetic does not wrap
    int *d_codes = nullptr;       // [N] group code of each cell
    int *d_perm = nullptr;        // [N] cell index at group-contiguous position p
    int *d_posptr = nullptr;      // [G+1]
    int *d_pk_blk = nullptr;      // packed dense layout (kernels_ovo_compact.h): [pk_nblk+1] first group of each block, then [pk_nblk] first key slot
    int pk_nblk = 0, pk_ref_out = 0;
    int *d_pk_code = nullptr;     // padded dense layout (dense OVR): group code of every key slot (holes: 0, they hold zero keys)
    int64_t pk_stride = 0;        // keys per gene in the packed layout
    int *d_pk_big = nullptr;      // [pk_nbig] the groups (never the reference) of more than 256 cells: their packed runs may need k_bucket_big_runs;
                                  // then [G]: a group's place in that list, or -1
    int pk_nbig = 0;
    int64_t pk_max_block_rows = 0; // rows of the longest block
    int *d_pk_order = nullptr;    // [pk_nblk] the blocks by falling row count, or null when their lengths are alike (k_group_compact starts the longest first)
    int pk_nlong = 0;             // blocks of more than OVRP_LONG_ROWS rows and at most 64 groups (k_ovr_partition_packed deals their units over all wavefronts)
    int *d_pk_long = nullptr;     // [pk_nlong] those blocks
    unsigned char *d_pk_islong = nullptr; // [pk_nblk]
    int64_t pk_len = 0;           // ... of which the blocks take the first pk_len (the padded dense layout's row length)
    int *d_counts = nullptr;      // [G]
    GroupConst *d_gconst = nullptr; // [G] per-group constants of the p-value / fold change for this ref (kernels_finalize.h)
    int *d_code_by_pos = nullptr; // [N] group code at position p
    u32 *d_hist_off = nullptr;    // [G+1] OVR one-pass histograms: words per lane before group g (16 per group of <= 255 cells, else 32)
    size_t hist_words = 0;        // d_hist_off[G]
    u16 *d_codes16 = nullptr;     // [N] d_codes as 16-bit values when G <= 65535 (half the cache lines per codes[row] gather), else null
    // the group-major CSR pass (kernels_csr_counts.h): row chunks of its histogram pre-pass -- slab 0 = the reference group's rows (OVO; OVR:
    // the column histograms come out of the count pass itself), slab 1 + k = the k-th group of more than 255 cells -- as [3][csr_n_chunks] {p0, rows, slab}, then [csr_n_big] group numbers
    int *d_csr_chunks = nullptr;
    int csr_n_chunks = 0, csr_n_big = 0; // csr_n_big < 0: more big groups than the route takes
    bool cur_sorted_known = false;       // the running call is on a bound CSR matrix whose rows were found in order when it was bound
    bool fused_tie_sparse = false;       // the fused OVR kernels run on a window of CSR input: tie sums as the reference's SPARSE path forms them
    int csr_counts_abl = 0;              // timing experiments (CsrCountsParams::abl)
    bool hold_csr_counts = false;        // set while a deferred call's leftover genes are recomputed (they must not come back to the route)
    // options
    int64_t gene_batch = 0;
    int64_t scratch_bytes = 24ll << 30; // (illico_ctx_create: min(64 GiB, a quarter of the device's memory))
    bool no_counts_path = false;
    bool no_fused_path = false;
    bool no_ovr_packed_partition = false; // 1: dense OVR partitions the padded rows (every key) instead of the packed ones
    bool no_packed_dense = false;      // 1: dense OVO on continuous values takes the transpose + k_ovo_rank route (no group-wise packing)
    bool no_sparse_packed_small = false; // sparse OVO, eight-byte keys: genes beyond k_csc_gene's LDS go to k_ovo_rank / the dense window (as before round 5), not to the packed rank kernel
    int csc_counts_max_windows = 0;    // > 0: count-valued CSC with more windows of groups than this leaves the histogram route (8: as before round 5)
    bool no_sparse_byte_values = false; // host-resident sparse input: the stored values always go up in their own type (count values: as bytes otherwise)
    bool no_host_numa = false;         // host-window pipelines: do not confine the fill threads to the NUMA node the caller's matrix lives on
    int host_fill_threads = 0;         // > 0: host threads that fill the pinned slots of the host-window pipelines (default: 12 float32 / 16 byte windows)
    bool debug_routes = false;         // stderr: what the packed OVO rank kernel left to the general routes, and why
    int packed_ref_cap = 0;            // > 0: caps the packed rank kernel's key slots for the reference (tests: value-range parts at small sizes)
    int big_runs_cap = 0;              // > 0: caps k_bucket_big_runs' LDS key slots (tests: the route through HBM at small sizes)
    int big_runs_slice_bytes = 0;      // > 0: the LDS bytes of k_bucket_big_runs_global's slice buffer (default: what the CU's LDS leaves beside the counters)
    bool no_big_runs_wide = false;     // k_bucket_big_runs: 256 threads whatever the runs' length
    bool no_big_runs_global = false;   // packed routes: a (gene, group) run beyond k_bucket_big_runs' LDS slots sends its gene to the general route (as before round 5)
    bool no_compact_order = false;     // k_group_compact: workgroups in grid order whatever the blocks' lengths
    bool no_compact_narrow = false;    // k_group_compact: never the 32-gene tiles for few, long blocks
    int64_t compact_narrow_wgs = 2048;  // ... and the launch would have fewer 64-gene workgroups than this (8 per compute unit)
    int64_t compact_narrow_rows = 8192; // ... from this many rows in the longest block
    bool no_ovr_part_coop = false;     // k_ovr_partition_packed: one wavefront per block whatever the blocks' lengths
    bool no_ovr_packed_big = false;    // dense OVR: groups above 65535 cells take the padded rows (every key, zeros included), as before
    bool no_group_hist_route = false;  // count-valued dense input with few, large groups: the fused kernels (a wavefront per group), as before
    int64_t group_hist_max_wgs = 1024;  // ... while the fused launch would have fewer workgroups than this
    int64_t group_hist_min_cells = 32768; // ... from this many cells (tests lower it)
    bool no_csr_transpose_split = false; // CSR -> CSC on the device: one workgroup per row block whatever their number
    bool no_csc_ovr_small_lds = false; // k_csc_ovr_gene: a CU's whole LDS per workgroup whatever the columns' lengths
    bool no_coop_runs = false;         // packed rank kernel: a long run is walked by one wavefront (as before) instead of all of the workgroup's
    bool no_deal_runs = false;         // packed rank kernel in parts: never deal the short runs by part first (every part then looks every key up, masked)
    bool no_packed_small_wg = false;   // packed rank kernel: never the 256-thread form for small references with few groups
    bool no_ovo_parts = false;         // packed rank kernel: never take a reference in value-range parts (genes beyond the LDS slots go to the general routes, as before round 5)
    int packed_eq_buckets = -1;        // packed rank kernel: distribution-following bucket function; -1 = for references above 16384 cells
    bool no_csc_counts_windows = false; // 1: count-valued CSC with more groups than LDS holds tables for never takes k_csc_counts (windows of groups)
    bool no_csc_counts_wide = false;   // 1: count-valued CSC with more than 8 groups above 255 cells never takes k_csc_counts (16-bit cells)
    bool ovr_full_dump = false;        // 1: the one-pass OVR route dumps whole group histograms (A/B of the shortened dump)
    bool no_wide_gather = false;       // 1: the 256-value stage always runs over the window as it lies (never left to the host's gather)
    bool no_leftover_gather = false;   // 1: the genes the fused passes leave are recomputed as column runs of the input (no gather into a narrow matrix)
    bool no_fused_wide = false;        // 1: no second, 256-value pass of the fused OVO route (genes beyond 63 go to the two-pass routes)
    bool no_csc_regroup_lds = false;   // 1: the two-kernel CSC route regroups with k_csc_segment only
    bool no_csc_gene_path = false;
    bool ovr_rank_whole = false;       // dense OVR rank kernel: one 1024-thread workgroup per CU with all of the LDS (as before round 5) instead of two of 512
    int64_t ovr_parts_cap = 0;         // > 0: keys per part at most in the value-range parts route (tests: many small parts)
    bool no_ovo_ref_buckets = false;   // 1: the OVO sort route always sorts the reference column (no value-bucket form)
    bool no_ovr_parts_path = false;    // 1: dense OVR (any values) never takes the value-range parts route (k_ovr_partition + k_csc_ovr_gene)
    bool no_csc_ovr_gene_path = false; // 1: CSC OVR never takes the single-kernel LDS-sort route (k_csc_ovr_gene)
    bool csc_ovr_sorted_form = false;  // 1: k_csc_ovr_gene sorts every gene's keys in LDS (the form tie-heavy columns take) instead of bucketing them
    bool no_csc_counts_path = false;   // 1: count-valued CSC genes do not take the LDS-histogram kernel (k_csc_counts)
    bool no_csc_counts_mixed = false;  // 1: k_csc_counts with 8-bit cells for every value only (the form 4-bit overflows fall back to)
    bool no_ovr_one_pass = false;      // 1: dense OVR reads X twice (column histogram, then rank sums) instead of once
    bool no_csr_tile_gather = false;    // 1: CSR -> CSC always by the scatter form (k_csr_block_scatter), as for unsorted rows
    bool no_f64_narrowing = false;      // 1: float64 sparse values that are all float32 values stay with the float64 kernels (A/B)
    bool no_csr_densify_any = false;    // 1: CSR windows with long columns of any values are never handed to the dense routes (A/B)
    bool no_csr_transpose_path = false; // 1: CSR is regrouped by (gene, group) with global atomics instead of being transposed to CSC
    bool dense_window_f32 = false;      // 1: CSR dense windows hold float32 cells instead of bytes
    int host_narrow = 0;               // host-resident count matrices as byte windows: 0 = when a value sample says counts, 1 = always, -1 = never
    bool no_csr_counts_path = false;   // 1: count-valued CSR never takes the group-major single pass (k_csr_counts)
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
    // "bound_ahead_genes" > 0: a call for FEWER genes of a bound CSR matrix computes the aligned window of that many genes around them
    // into planes of the context's own and hands out slices -- the reference's driver asks for ~256 genes at a time
    // (asymptotic_wilcoxon.py:213-241), and a CSR call walks every row whatever the width of its window (core.hip: run_bound_ahead)
    int64_t bound_ahead_genes = 0;
    uint64_t groups_gen = 0;      // counts illico_set_groups calls (the windows below belong to one set of groups)
    struct AheadWindow {
        const illico_matrix *m = nullptr;
        uint64_t gen = 0, stamp = 0;
        int flags = 0, alternative = 0;
        int64_t lb = 0, ub = 0;
        double *planes = nullptr;   // device, [3][G][ub - lb]
        size_t cap = 0;             // bytes
    } ahead[2];
    uint64_t ahead_clock = 0;
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

int fail(illico_ctx *c, int code, const char *fmt, ...);

#define HIPCHK(ctx, call)                                                                              \
    do {                                                                                               \
        hipError_t e__ = (call);                                                                       \
        if (e__ != hipSuccess)                                                                         \
            return fail(ctx, e__ == hipErrorOutOfMemory ? ILLICO_ERR_OOM : ILLICO_ERR_HIP, "%s failed: %s (%s:%d)", \
                        #call, hipGetErrorString(e__), __FILE__, __LINE__);                            \
    } while (0)

int get_scratch(illico_ctx *c, const char *name, size_t bytes, void **out);
hipEvent_t take_event(illico_ctx *c);

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

void drain_events(illico_ctx *c);
int resolve_pending(illico_ctx *c); // completes a deferred call (core.hip)

#define FUSED_RT 64 // table size of the fused single-pass routes (values 0 .. 63)
static const size_t kMaxLds = 160 * 1024;
static const int kOvoThreads = 512;

static inline size_t dtype_size(int dt) { return (dt == ILLICO_F32 || dt == ILLICO_I32) ? 4 : 8; }

struct OutPlanes {
    double *p, *u, *fc; // device
    int64_t ld;
    bool staged;
};

// pinned slots / copy stream of the host-window pipeline (dense_driver.h: host_windows_pipeline); one per context, freed with it
#define HS_SLOTS 3
struct HostStage {
    void *pin[HS_SLOTS] = {nullptr, nullptr, nullptr};
    size_t pin_bytes = 0;
    int *lists = nullptr;        // pinned: column lists of the flagged genes, window after window (gathered leftovers)
    size_t lists_ints = 0;
    hipStream_t copy = nullptr;
    hipEvent_t up[HS_SLOTS] = {nullptr, nullptr, nullptr}, done[HS_SLOTS] = {nullptr, nullptr, nullptr};
    void *sp_pin[2] = {nullptr, nullptr}; // pinned byte chunks of a sparse matrix's count values on their way up (sparse_driver.h)
    hipEvent_t sp_up[2] = {nullptr, nullptr};
};
HostStage *host_stage_of(illico_ctx *c);
void free_host_stage(illico_ctx *c);

// ---- non-template host helpers (core.hip) ----
int ovo_counts_limit(const illico_ctx *c);           // table size of the two-pass histogram route (k_ovo_counts)
bool counts_path_allowed(const illico_ctx *c, int flags);
bool fused_path_allowed(const illico_ctx *c, int flags);
int launch_finalize(illico_ctx *c, const long long *s2u, const u64 *stie, const double *ssum, const double *gene_total, int nb, int flags,
                    int alternative, double *out_p, double *out_u, double *out_fc, int64_t out_ld, int64_t col_off, const int *col_map = nullptr,
                    bool packed = false, bool tie_f64 = false);
int launch_gene_totals(illico_ctx *c, const double *ssum, int G, int nb, double *gtot);
void flagged_runs(const u32 *hf, int64_t wn, int64_t w0, std::vector<std::pair<int64_t, int64_t>> &runs);

// ---- per-value-type entry points (dense_driver.h / sparse_driver.h; one explicit instantiation per type, in its own translation unit) ----
template <typename InT, typename KeyT>
int run_dense_t(illico_ctx *c, const void *X, int dtype, int64_t N, int64_t ld, int64_t col_lb, int64_t col_ub, int flags, int alternative,
                const OutPlanes &o);
template <typename InT, typename KeyT>
int run_leftovers(illico_ctx *c, const void *X, int dtype, int64_t N, int64_t ld, int64_t col_lb, int64_t col_ub, int flags, int alternative,
                  const OutPlanes &o, const u32 *hf, bool wide_skipped = false, const int *outer = nullptr);
template <typename InT>
int run_fused_ovo(illico_ctx *c, const void *X, int64_t ld, int64_t b0, int nb, int flags, int alternative, const OutPlanes &o, int64_t col_off,
                  std::vector<u32> &h_flags, int defer_slot = -1, bool probe = false, int64_t max_gather = 0, const u32 *init_flags = nullptr);
template <typename InT, typename IdxT, typename KeyT>
int run_sparse_t(illico_ctx *c, bool is_csr, const void *data, const void *indices, const void *indptr, int dtype, int64_t n_rows, int64_t n_cols,
                 int64_t col_lb, int64_t col_ub, int flags, int alternative, const OutPlanes &o, bool allow_dense_window = true,
                 bool allow_transpose = true, bool indices_are_codes = false, bool allow_csr_counts = true);
// the run_sparse_t of a type given by its codes (core.hip; what a deferred CSC pass's leftovers and the drivers' own re-entries call)
int run_sparse_inner(illico_ctx *c, bool is_csr, const void *data, int dtype, const void *indices, const void *indptr, int idx_dtype, int64_t n_rows,
                     int64_t n_cols, int64_t col_lb, int64_t col_ub, int flags, int alternative, const OutPlanes &o);
