/*
 * illico_hip.h -- C-ABI of libillico_hip.so, the MI355X (gfx950) engine for illico's asymptotic
 * Wilcoxon rank-sum hot path.
 *
 * The reference (remydubois/illico v0.2.0) has no FFI: its seam is the Python operator
 *   dispatcher(X, chunk_lb, chunk_ub, grpc, is_log1p, use_continuity, tie_correct, alternative)
 *       -> (pvalues, statistics, fold_change)          each float64 [n_groups, chunk_ub-chunk_lb]
 * (illico/asymptotic_wilcoxon.py:59-67; Numba signature illico/utils/compile.py:35-47), one
 * implementation per (Test, KernelDataFormat) key (illico/utils/registry.py:15-43).  The entry
 * points below are what a binding for that seam binds: plain pointers and sizes, `int` status
 * returns (0 = ok, negative = error mapped by the host onto the reference's exception types),
 * nothing thrown across the boundary.  INTEGRATION.md shows the ctypes stub.
 *
 * Ownership: the caller owns X, the group arrays and the three output planes; the library owns
 * only device scratch inside the context and never writes to X (the reference's tests assert the
 * input is not mutated, tests/test_asymptotic_wilcoxon.py:187-194).
 * Threading: one context = one HIP stream.  Every entry point that takes a context locks it for the
 * duration of the call, so host threads sharing ONE context (the reference's joblib threads share one
 * dispatcher, illico/asymptotic_wilcoxon.py:236-241) are serialised, never raced; illico_last_error
 * returns the calling thread's own last message.  Threads that want overlap use one context each.
 * illico_ctx_destroy must not race with other calls on the same context.
 */
#ifndef ILLICO_HIP_H
#define ILLICO_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct illico_ctx illico_ctx;

/* status codes; the Python host maps them onto the reference's exceptions */
enum {
    ILLICO_OK = 0,
    ILLICO_ERR_ARG = -1,          /* null pointer / nonsensical size                          -> ValueError */
    ILLICO_ERR_BOUNDS = -2,       /* bad chunk bounds (asymptotic_wilcoxon.py:49-50, csc.py:158-159, csr.py:161-162) -> ValueError */
    ILLICO_ERR_ALTERNATIVE = -3,  /* unknown alternative (utils/math.py:116)                  -> ValueError */
    ILLICO_ERR_DTYPE = -4,        /* unsupported element / index dtype (registry.py:54-58)    -> KeyError   */
    ILLICO_ERR_NO_GROUPS = -5,    /* illico_set_groups not called / inconsistent with n_rows  -> ValueError */
    ILLICO_ERR_UNSORTED = -6,     /* CSR indices not sorted (asymptotic_wilcoxon.py:186-193)  -> ValueError */
    ILLICO_ERR_HIP = -10,         /* a HIP runtime call failed (see illico_last_error)        -> RuntimeError */
    ILLICO_ERR_OOM = -11,         /* device allocation failed                                 -> MemoryError */
    ILLICO_ERR_UNSUPPORTED = -12  /* shape outside what this build handles (see last_error)   -> NotImplementedError */
};

/* element dtypes of X / data */
enum { ILLICO_F32 = 0, ILLICO_F64 = 1, ILLICO_I32 = 2, ILLICO_I64 = 3 };
/* index dtypes of sparse indices/indptr (scipy uses one dtype for both) */
enum { ILLICO_IDX_I32 = 0, ILLICO_IDX_I64 = 1 };
/* alternative hypothesis (utils/math.py:99-116) */
enum { ILLICO_ALT_TWO_SIDED = 0, ILLICO_ALT_LESS = 1, ILLICO_ALT_GREATER = 2 };
/* flags */
enum {
    ILLICO_FLAG_LOG1P = 1,          /* is_log1p        (utils/math.py:212)        */
    ILLICO_FLAG_CONTINUITY = 2,     /* use_continuity  (ovo/dense_ovo.py:58)      */
    ILLICO_FLAG_TIE_CORRECT = 4,    /* tie_correct     (ovo/dense_ovo.py:54)      */
    ILLICO_FLAG_INPUT_DEVICE = 8,   /* X / data / indices / indptr are device pointers on ctx's device */
    ILLICO_FLAG_OUTPUT_DEVICE = 16, /* out_p / out_u / out_fc are device pointers                       */
    /* illico_run_dense with device-resident input AND device planes: enqueue the fused single-pass route and return without
     * waiting for it.  The few genes that route cannot take (values outside its table) are recomputed when their flags have
     * arrived -- by the next call on the context or by illico_ctx_synchronize, after which the planes are complete.  X and
     * the planes must stay valid until then.  A following deferred call that writes OTHER planes is enqueued before the
     * earlier one is completed, so back-to-back passes run without a host round trip in between.
     * illico_run_csc / illico_run_bound on device-resident CSC arrays with device planes honour it too: the count-valued pass
     * (whether the window holds counts at all is decided on the device, from a sample of its stored values) is enqueued and the
     * columns it cannot take are recomputed the same way, later.  Ignored elsewhere (host arrays, host planes, CSR). */
    ILLICO_FLAG_DEFER = 32
};

/* ---- context ---------------------------------------------------------------------------- */
int illico_ctx_create(int device_id, illico_ctx **out_ctx);
int illico_ctx_destroy(illico_ctx *ctx);
/* Use an existing hipStream_t (e.g. torch's current stream) instead of the context's own. */
int illico_ctx_set_stream(illico_ctx *ctx, void *hip_stream);
/* Tunables: "gene_batch" (genes per device pass, 0 = auto), "scratch_bytes" (cap of device scratch; default: 64 GiB or a
 * quarter of the device's memory, whichever is less),
 * "profile" (1 = bracket kernel launches with HIP events on the context's stream), "profile_only" (kernel id: time
 * that kernel only, -1 = all), "fused_groups_per_wg" / "ovr_hist_groups_per_wg" (launch geometry, 0 = auto).
 * Route switches, all 0 by default; every route produces the same integers, the switches exist so that tests and
 * A/B measurements can force each one: "no_fused_path", "no_counts_path" (dense / segmented histogram routes),
 * "no_ovr_one_pass" (dense OVR in two passes over X; "ovr_full_dump" = 1: its one pass writes every word of every group histogram
 * instead of the leading non-zero ones), "no_csc_counts_path" (count-valued CSC on LDS histograms;
 * "no_csc_counts_mixed" = 1: its 8-bit cell form only; "no_csc_counts_wide" = 1: never its 16-bit-cell form, i.e. the route is off when
 * more than 8 ranked groups exceed 255 cells; "no_csc_counts_windows" = 1: never in windows of groups, i.e. off when the groups' tables
 * do not fit LDS at once),
 * "no_packed_dense" (dense two-pass routes: group-wise packing of the non-zero keys + look-ups in a counted bitmap of the reference
 * for OVO, the transposition with the group sums folded in for OVR; 1 = the plain transposition and the kernels behind it;
 * "no_ovr_packed_partition" = 1: dense OVR splits the padded key rows -- every key -- instead of the packed ones;
 * "packed_eq_buckets": the packed OVO kernel's value buckets follow the reference's distribution, 1 = always, 0 = never,
 * -1 = for references of more than 16384 cells, the default; "no_ovo_parts" = 1: a reference whose non-zero keys outgrow the kernel's
 * LDS slots is never taken in value-range parts -- such genes go to the general sort routes; "no_big_runs_global" = 1: a ranked
 * group's run of more keys than LDS holds is not dealt into value buckets through HBM -- its gene goes to the general sort routes),
 * "no_compact_narrow" (k_group_compact never takes its 32-gene tiles for few, long blocks; "compact_narrow_rows": rows of the longest
 * block from which it does, default 8192), "no_big_runs_wide" (runs above 8192 keys get no 1024-thread launch of their own),
 * "big_runs_slice_bytes" (> 0: LDS bytes of k_bucket_big_runs_global's slice buffer), "no_ovr_packed_big" (dense OVR with a group above
 * 65535 cells partitions the padded rows, as before), "no_ovr_part_coop" (the packed partition walks every block with one wavefront),
 * "no_group_hist_route" (count-valued dense input with few large groups, or OVR / OVO with a group above 65535 cells: the fused kernels
 * instead of the (group, gene) value histograms of kernels_group_hists.h; "group_hist_min_cells": cells from which the route is
 * taken, default 32768), "no_csr_transpose_split" (CSR -> CSC on the device: one workgroup per row block whatever their number),
 * "no_csc_ovr_small_lds" (k_csc_ovr_gene takes a CU's whole LDS per workgroup whatever the columns' lengths),
 * "no_ovo_ref_buckets" (OVO sort route: reference column in value buckets instead of sorted), "no_ovr_parts_path" (dense OVR, any values: value-range parts ranked in LDS; "ovr_parts_cap" > 0 caps the keys per part), "no_csc_gene_path" (CSC OVO single-kernel route), "no_csc_ovr_gene_path" (CSC OVR single-kernel route; "csc_ovr_sorted_form" = 1 makes it sort every
 * gene in LDS, the form tie-heavy columns take, instead of bucketing the keys), "no_csc_regroup_lds" (two-kernel CSC route: regroup with scattered
 * stores only),
 * "no_fused_wide" (the 256-value second stage of the fused routes), "no_wide_gather" (that stage always over the window as it lies, never
 * on the gathered columns), "no_leftover_gather" (the genes the fused passes leave are recomputed as column runs of the input instead
 * of being gathered into a narrow matrix),
 * "no_dense_window_path" (CSR through dense windows; "dense_window_f32" = 1: float32 cells instead of bytes), "no_csr_transpose_path" / "no_csr_tile_gather"
 * (CSR -> CSC transposition on the device / its gather form for sorted rows), "no_csr_counts_path" (count-valued CSR by the byte windows
 * instead of the group-major pass), "no_csr_densify_any" (sparse windows with columns of more than 32 768 stored entries stay with the
 * sparse routes instead of being written out dense), "no_f64_narrowing" (float64 sparse values that are all float32 values stay with the
 * float64 kernels).
 * Not route switches: "host_narrow" (-1 automatic / 1 / 0: host-resident count matrices go up as bytes), "bound_ahead_genes" (0 = off:
 * a call for fewer genes of a bound CSR matrix computes the aligned window of that many genes around them once and later calls inside
 * the window are slices of it -- for bindings that keep the reference's 256-gene chunk loop).
 * Unknown keys return ILLICO_ERR_ARG. */
int illico_ctx_set_option(illico_ctx *ctx, const char *key, int64_t value);
const char *illico_last_error(const illico_ctx *ctx);
int illico_ctx_synchronize(illico_ctx *ctx);

/* ---- groups: GroupContainer of illico/utils/groups.py:6-15, all int64 host arrays ------- */
/* encoded_groups[n_cells], counts[n_groups], indices[n_cells] (cells ordered by group),
 * indptr[n_groups+1]; encoded_ref_group == -1 selects OVR (asymptotic_wilcoxon.py:41-44). */
int illico_set_groups(illico_ctx *ctx, const int64_t *encoded_groups, const int64_t *counts,
                      const int64_t *indices, const int64_t *indptr, int64_t n_cells, int64_t n_groups,
                      int64_t encoded_ref_group);

/* ---- the six (Test x KernelDataFormat) dispatchers of registry.py:26-43 -------------------
 * Each computes columns [col_lb, col_ub) and writes three row-major float64 planes
 * out_*[g * out_ld + (j - col_lb)], g < n_groups.  The reference-group row of an OVO call is
 * written as (p = 1.0, U = -1.0) in every format (sparse_ovo.py:140-143).
 */
/* replaces dense_ovo_mwu_kernel_over_contiguous_col_chunk (ovo/dense_ovo.py:65-137) and
 * dense_ovr_mwu_kernel_over_contiguous_col_chunk (ovr/dense_ovr.py:15-80); X is row-major
 * [n_rows, >=n_cols] with leading dimension ld elements (registry.py:105-108). */
int illico_run_dense(illico_ctx *ctx, const void *X, int dtype, int64_t n_rows, int64_t n_cols, int64_t ld,
                     int64_t col_lb, int64_t col_ub, int flags, int alternative, double *out_p, double *out_u,
                     double *out_fc, int64_t out_ld);
/* replaces csc_ovo_mwu_kernel_over_contiguous_col_chunk (ovo/sparse_ovo.py:163-210) and
 * csc_ovr_mwu_kernel_over_contiguous_col_chunk (ovr/sparse_ovr.py:100-155); CSCMatrix(data, indices,
 * indptr, shape) of utils/sparse/csc.py:10. */
int illico_run_csc(illico_ctx *ctx, const void *data, int dtype, const void *indices, const void *indptr,
                   int idx_dtype, int64_t n_rows, int64_t n_cols, int64_t col_lb, int64_t col_ub, int flags,
                   int alternative, double *out_p, double *out_u, double *out_fc, int64_t out_ld);
/* replaces csr_ovo_mwu_kernel_over_contiguous_col_chunk (ovo/sparse_ovo.py:214-260) and
 * csr_ovr_mwu_kernel_over_contiguous_col_chunk (ovr/sparse_ovr.py:158-208); CSRMatrix of
 * utils/sparse/csr.py:16.  Indices must be sorted per row (checked by illico_csr_indices_sorted). */
int illico_run_csr(illico_ctx *ctx, const void *data, int dtype, const void *indices, const void *indptr,
                   int idx_dtype, int64_t n_rows, int64_t n_cols, int64_t col_lb, int64_t col_ub, int flags,
                   int alternative, double *out_p, double *out_u, double *out_fc, int64_t out_ld);
/* ---- a sparse matrix bound once, computed chunk by chunk -------------------------------------
 * The reference's driver calls a dispatcher once per gene chunk with the SAME matrix (illico/asymptotic_wilcoxon.py:236-241:
 * 32 calls at 8000 genes and batch_size 256); CSR rows span every gene, so illico_run_csr on HOST arrays has to move the whole
 * matrix to the device in every call.  illico_csr_bind / illico_csc_bind upload the arrays ONCE (or, with
 * ILLICO_FLAG_INPUT_DEVICE, adopt device arrays without copying) and return a handle; illico_run_bound then computes any
 * column chunk of it like illico_run_csr / illico_run_csc on device-resident arrays (flags: LOG1P / CONTINUITY / TIE_CORRECT /
 * OUTPUT_DEVICE / DEFER).  The caller may free or modify its host arrays as soon as bind returns.  A handle belongs to the
 * context that made it; illico_matrix_release frees the device copy (illico_ctx_destroy releases what is left).  An explicit
 * handle, not a cache keyed on pointers: nothing about a matrix is remembered behind the caller's back. */
typedef struct illico_matrix illico_matrix;
int illico_csr_bind(illico_ctx *ctx, const void *data, int dtype, const void *indices, const void *indptr, int idx_dtype,
                    int64_t n_rows, int64_t n_cols, int flags, illico_matrix **out_matrix);
int illico_csc_bind(illico_ctx *ctx, const void *data, int dtype, const void *indices, const void *indptr, int idx_dtype,
                    int64_t n_rows, int64_t n_cols, int flags, illico_matrix **out_matrix);
int illico_run_bound(illico_ctx *ctx, const illico_matrix *matrix, int64_t col_lb, int64_t col_ub, int flags, int alternative,
                     double *out_p, double *out_u, double *out_fc, int64_t out_ld);
int illico_matrix_release(illico_ctx *ctx, illico_matrix *matrix);
/* Adopted device arrays (ILLICO_FLAG_INPUT_DEVICE) stay the caller's: they must not change while a call on them is in flight, and what
 * the context remembers about a bound matrix -- whether its CSR rows are in order (looked at once, at bind time) and, with the option
 * "bound_ahead_genes", result windows computed ahead of the chunk calls -- describes the arrays as they were.  A caller that rewrites
 * adopted arrays in place (normalise, log1p_) calls illico_matrix_touch afterwards: windows of the matrix are dropped, the row order is
 * looked at again.  (Uploaded matrices are the library's own copy and never need it.) */
int illico_matrix_touch(illico_ctx *ctx, illico_matrix *matrix);

/* replaces check_indices_sorted_per_parcel (utils/ranking.py:245-273); *out_sorted = 1/0. */
int illico_csr_indices_sorted(illico_ctx *ctx, const void *indices, const void *indptr, int idx_dtype,
                              int64_t n_rows, int flags, int *out_sorted);

/* ---- the ranking primitives, before finalisation -----------------------------------------
 * Device counterpart of rank_sum_and_ties_from_sorted (utils/ranking.py:52-158; OVO) and
 * _accumulate_group_ranksums_from_argsort (utils/ranking.py:7-49; OVR) for the dense columns [col_lb, col_ub): the integer
 * statistics the dispatchers feed to compute_pval, as the reference's own primitive tests look at them
 * (tests/utils/test_ranking.py:13-56).  Host arrays [col_ub - col_lb][n_groups]:
 *   out_two_u[j][g]     2 * U1,  U1 = n_ref n_tgt + n_tgt (n_tgt + 1) / 2 - ranksum_g   (dense_ovo.py:48, dense_ovr.py:57-61;
 *                       n_ref = reference-group size, or every other cell for OVR) -- ranksum_g follows exactly;
 *   out_tie_sum[j][g]   sum over tie blocks of t^3 - t  (of reference + group g for OVO, of the whole column for OVR);
 *   out_value_sum[j][g] the group's value sum (expm1'd under ILLICO_FLAG_LOG1P).
 * The reference group's own entries are unspecified in OVO.  Runs the two-pass routes (the fused single-pass kernels
 * never materialise these numbers).  flags: ILLICO_FLAG_LOG1P, ILLICO_FLAG_INPUT_DEVICE. */
int illico_rank_statistics(illico_ctx *ctx, const void *X, int dtype, int64_t n_rows, int64_t n_cols, int64_t ld,
                           int64_t col_lb, int64_t col_ub, int flags, int64_t *out_two_u, uint64_t *out_tie_sum,
                           double *out_value_sum);

/* ---- multi-GPU ----------------------------------------------------------------------------
 * Gene sharding is host-side: one context per GPU and process, each computing its own column range with the entry points
 * above (no input is exchanged); the one collective of the path -- the gather of the planes to rank 0 -- is issued by the
 * host over RCCL (illico_amd/distributed.py: torch.distributed.gather on device planes, backend "nccl").  The C-ABI has no
 * illico_gather entry: a binding that wants several GPUs brings its own process group, as the Python host does.
 *
 * What the C-ABI does offer the gathering rank: the way its planes reach host memory.  Three gathered device planes [n_groups][n_cols]
 * (dense, row pitch n_cols) are copied into the caller's host planes (row pitch out_ld >= n_cols) through the context's two pinned
 * buffers -- block i travels at the link's rate while block i - 1 is scattered by a few host threads -- i.e. the path the results of
 * an ordinary call with host planes take (asymptotic_wilcoxon.py:242-244 copies each chunk's planes into `results`).  Needs groups
 * (n_groups).  */
int illico_planes_to_host(illico_ctx *ctx, const double *dev_p, const double *dev_u, const double *dev_fc, int64_t n_cols,
                          double *out_p, double *out_u, double *out_fc, int64_t out_ld);

/* ---- measurement hooks (bench.py roofline leg) ------------------------------------------- */
int illico_profile_num_kernels(void);
const char *illico_profile_kernel_name(int kernel_id);
/* Sums HIP-event durations of kernel `kernel_id` since the last reset (synchronises the stream). */
int illico_profile_get(illico_ctx *ctx, int kernel_id, double *total_ms, int64_t *launches);
int illico_profile_reset(illico_ctx *ctx);
/* Bytes of INPUT (matrix values / indices / index pointers) copied host -> device since the context was created: what a test
 * of "one upload per matrix" looks at. */
int illico_profile_input_bytes(illico_ctx *ctx, int64_t *h2d_bytes);

const char *illico_version(void);

#ifdef __cplusplus
}
#endif
#endif /* ILLICO_HIP_H */
