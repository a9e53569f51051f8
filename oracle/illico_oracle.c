/*
 * TEST INFRASTRUCTURE ONLY -- CPU oracle for the illico asymptotic Wilcoxon rank-sum hot path.
 *
 * This file is a plain-C restatement of the reference's algorithm (remydubois/illico v0.2.0,
 * mounted at /root/reference in the build container).  Each function cites the reference
 * file:line it follows.  It exists so that
 *   - tests/ can check the HIP engine (and the committed golden vectors) against it,
 *   - __graft_entry__.smoke() can check one tiny GPU invocation,
 *   - bench.py can time a CPU baseline ("cpu_baseline.kind" = "port") on the GPU box's host.
 * Nothing under illico_amd/ imports, links or executes it: the product path is the HIP
 * library and fails loudly when that library is missing.
 *
 * Parity pinning: tests/golden/ holds outputs of the reference itself (imported un-jitted in the
 * build container by tests/golden/make_goldens.py) and tests/test_oracle_vs_golden.py checks
 * this oracle against them; tests also pin it against scipy.stats.mannwhitneyu, the
 * reference's own test oracle (reference tests/test_asymptotic_wilcoxon.py:63-108).
 *
 * Build: see oracle/Makefile (gcc -O2 -ffp-contract=off -fopenmp; strict IEEE, no fast-math).
 */
#define _GNU_SOURCE
#include <math.h>
#include <sched.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define ORACLE_ERR_BOUNDS (-2)
#define ORACLE_ERR_ALTERNATIVE (-3)
#define ORACLE_ERR_DTYPE (-4)

enum { ALT_TWO_SIDED = 0, ALT_LESS = 1, ALT_GREATER = 2 };

/* GroupContainer, illico/utils/groups.py:6-15 (all int64, as the reference builds them) */
typedef struct {
    const int64_t *encoded_groups; /* [n_cells] */
    const int64_t *counts;         /* [n_groups] */
    const int64_t *indices;        /* [n_cells]  argsort of labels */
    const int64_t *indptr;         /* [n_groups+1] */
    int64_t n_cells, n_groups;
    int64_t encoded_ref_group;     /* -1 => OVR */
} oracle_groups;

/* ---- illico/utils/math.py:64-118  compute_pval (fastmath=False) ---- */
double oracle_compute_pval(int64_t n_ref, int64_t n_tgt, int64_t n, double tie_sum, double U, double mu,
                           double contin_corr, int alternative) {
    double tie_corr = 1.0 - tie_sum / (double)(n * (n - 1) * (n + 1));          /* math.py:95 */
    if (tie_corr > 1.0e-9) {                                                     /* :96 */
        double sigma = sqrt((double)(n_ref * n_tgt * (n_ref + n_tgt + 1)) / 12.0 * tie_corr); /* :97 */
        if (alternative == ALT_TWO_SIDED) {                                      /* :99-104 */
            double other = (double)(n_ref * n_tgt) - U;
            U = (U < other) ? U : other;
            double delta = U - mu;
            double sgn = (delta > 0.0) ? 1.0 : ((delta < 0.0) ? -1.0 : 0.0);
            double z = (fabs(delta) + sgn * contin_corr) / sigma;
            return erfc(z / sqrt(2.0));
        } else if (alternative == ALT_GREATER) {                                 /* :105-109 */
            double delta = U - mu;
            double z = (delta - contin_corr) / sigma;
            return 0.5 * erfc(z / sqrt(2.0));
        } else {                                                                 /* :110-114 */
            double delta = U - mu;
            double z = (delta + contin_corr) / sigma;
            return 0.5 * erfc(-z / sqrt(2.0));
        }
    }
    return 1.0;                                                                  /* :117-118 */
}

/* ---- illico/utils/math.py:168-193  fold_change_from_summed_expr ---- */
void oracle_fold_change_from_summed_expr(const double *agg /*[G,w]*/, const oracle_groups *g, int64_t w,
                                         double *fold_change /*[G,w]*/) {
    int64_t G = g->n_groups;
    if (g->encoded_ref_group == -1) {
        int64_t total_count = 0;
        for (int64_t k = 0; k < G; ++k) total_count += g->counts[k];
        for (int64_t j = 0; j < w; ++j) {
            double total = 0.0; /* group_agg_counts.sum(axis=0): rows added in order */
            for (int64_t k = 0; k < G; ++k) total += agg[k * w + j];
            for (int64_t k = 0; k < G; ++k) {
                double mu_tgt = agg[k * w + j] / (double)g->counts[k];
                double mu_ref = (total - agg[k * w + j]) / (double)(total_count - g->counts[k]);
                fold_change[k * w + j] = (mu_ref == 0.0) ? INFINITY : mu_tgt / mu_ref;
            }
        }
    } else {
        int64_t r = g->encoded_ref_group;
        for (int64_t j = 0; j < w; ++j) {
            double mu_ref = agg[r * w + j] / (double)g->counts[r];
            for (int64_t k = 0; k < G; ++k) {
                double mu_tgt = agg[k * w + j] / (double)g->counts[k];
                fold_change[k * w + j] = (mu_ref == 0.0) ? INFINITY : mu_tgt / mu_ref;
            }
        }
    }
}

/* ---- illico/utils/ranking.py:223-273  check_indices_sorted_per_parcel ---- */
int oracle_check_indices_sorted_per_parcel(const int64_t *indices, const int64_t *indptr, int64_t n_parcels) {
    for (int64_t k = 0; k < n_parcels; ++k)
        for (int64_t i = indptr[k] + 1; i < indptr[k + 1]; ++i)
            if (indices[i] < indices[i - 1]) return 0;
    return 1;
}

#define T float
#define SFX(n) n##_f32
#define EXPM1 expm1f
#include "oracle_impl.inc"
#undef T
#undef SFX
#undef EXPM1

#define T double
#define SFX(n) n##_f64
#define EXPM1 expm1
#include "oracle_impl.inc"
#undef T
#undef SFX
#undef EXPM1

/* ======================================================================================
 * Whole-call drivers: the gene-chunk loop of illico/asymptotic_wilcoxon.py:213-249 with an
 * integer batch_size (the "auto" splitter skips columns), one OpenMP thread
 * per chunk in place of joblib's thread pool (asymptotic_wilcoxon.py:236-241).  Outputs are the
 * three [G, n_cols] row-major planes that the driver scatters into (asymptotic_wilcoxon.py:242-244).
 * fmt: 0 dense, 1 csc, 2 csr.  dtype: 0 f32, 1 f64.  test is OVR iff encoded_ref_group == -1
 * (asymptotic_wilcoxon.py:41-44).
 * ====================================================================================== */
/* Thread placement for the timed CPU baseline (bench.py): worker t is pinned to cpus[t % n].  Without it the kernel's
 * scheduler may leave freshly created OpenMP threads stacked on one CPU for hundreds of milliseconds (measured: 4
 * threads, 3 of them on the same CPU, 4x the single-thread wall time), which would understate the baseline.  The list
 * comes from the Python side (one CPU per physical core first, SMT siblings after).  n = 0 (default): no pinning. */
static int g_cpus[4096];
static int g_ncpus = 0;
void oracle_set_cpu_list(const int *cpus, int n) {
    if (n < 0) n = 0;
    if (n > 4096) n = 4096;
    for (int i = 0; i < n; ++i) g_cpus[i] = cpus[i];
    g_ncpus = n;
}

int oracle_run(int fmt, int dtype, const void *data, const int64_t *indices, const int64_t *indptr,
               int64_t n_rows, int64_t n_cols, int64_t ld, int64_t col_lb, int64_t col_ub,
               const int64_t *encoded_groups, const int64_t *counts, const int64_t *grp_indices,
               const int64_t *grp_indptr, int64_t n_groups, int64_t encoded_ref_group, int is_log1p,
               int use_continuity, int tie_correct, int alternative, int64_t batch_size, int n_threads,
               double *out_p, double *out_u, double *out_fc /* each [G, col_ub-col_lb] */) {
    if (col_lb < 0 || col_ub > n_cols || col_lb > col_ub) return ORACLE_ERR_BOUNDS; /* asymptotic_wilcoxon.py:49-50 */
    if (alternative < 0 || alternative > 2) return ORACLE_ERR_ALTERNATIVE;          /* math.py:116 */
    if (dtype != 0 && dtype != 1) return ORACLE_ERR_DTYPE;
    oracle_groups g = {encoded_groups, counts, grp_indices, grp_indptr, n_rows, n_groups, encoded_ref_group};
    int64_t W = col_ub - col_lb;
    if (batch_size <= 0) batch_size = 256;
    int64_t n_chunks = (W + batch_size - 1) / batch_size;
    int rc_all = 0;
    if (n_threads < 1) n_threads = 1;
    cpu_set_t saved_mask;
    int pinned = 0;
#ifdef _OPENMP
    if (g_ncpus > 0 && n_threads > 1 && sched_getaffinity(0, sizeof saved_mask, &saved_mask) == 0) {
        pinned = 1;
#pragma omp parallel num_threads(n_threads)
        {
            cpu_set_t m;
            CPU_ZERO(&m);
            CPU_SET(g_cpus[omp_get_thread_num() % g_ncpus], &m);
            sched_setaffinity(0, sizeof m, &m);
        }
    }
#endif
#pragma omp parallel for schedule(dynamic, 1) num_threads(n_threads)
    for (int64_t c = 0; c < n_chunks; ++c) {
        int64_t lb = col_lb + c * batch_size;
        int64_t ub = lb + batch_size < col_ub ? lb + batch_size : col_ub;
        int64_t w = ub - lb;
        double *p = (double *)malloc(sizeof(double) * (size_t)(n_groups * w));
        double *u = (double *)malloc(sizeof(double) * (size_t)(n_groups * w));
        double *fc = (double *)malloc(sizeof(double) * (size_t)(n_groups * w));
        int rc = 0;
        int ovr = (encoded_ref_group == -1);
        if (fmt == 0) {
            if (dtype == 0)
                rc = ovr ? oracle_dense_ovr_chunk_f32((const float *)data, n_rows, ld, lb, ub, &g, is_log1p, use_continuity, tie_correct, alternative, p, u, fc)
                         : oracle_dense_ovo_chunk_f32((const float *)data, n_rows, ld, lb, ub, &g, is_log1p, use_continuity, tie_correct, alternative, p, u, fc);
            else
                rc = ovr ? oracle_dense_ovr_chunk_f64((const double *)data, n_rows, ld, lb, ub, &g, is_log1p, use_continuity, tie_correct, alternative, p, u, fc)
                         : oracle_dense_ovo_chunk_f64((const double *)data, n_rows, ld, lb, ub, &g, is_log1p, use_continuity, tie_correct, alternative, p, u, fc);
        } else {
            int is_csr = (fmt == 2);
            if (dtype == 0)
                rc = ovr ? oracle_sparse_ovr_chunk_f32(is_csr, (const float *)data, indices, indptr, n_rows, n_cols, lb, ub, &g, is_log1p, use_continuity, tie_correct, alternative, p, u, fc)
                         : oracle_sparse_ovo_chunk_f32(is_csr, (const float *)data, indices, indptr, n_rows, n_cols, lb, ub, &g, is_log1p, use_continuity, tie_correct, alternative, p, u, fc);
            else
                rc = ovr ? oracle_sparse_ovr_chunk_f64(is_csr, (const double *)data, indices, indptr, n_rows, n_cols, lb, ub, &g, is_log1p, use_continuity, tie_correct, alternative, p, u, fc)
                         : oracle_sparse_ovo_chunk_f64(is_csr, (const double *)data, indices, indptr, n_rows, n_cols, lb, ub, &g, is_log1p, use_continuity, tie_correct, alternative, p, u, fc);
        }
        if (rc == 0) {
            for (int64_t k = 0; k < n_groups; ++k) { /* results[:, lb:ub, .] = ... */
                memcpy(out_p + k * W + (lb - col_lb), p + k * w, sizeof(double) * (size_t)w);
                memcpy(out_u + k * W + (lb - col_lb), u + k * w, sizeof(double) * (size_t)w);
                memcpy(out_fc + k * W + (lb - col_lb), fc + k * w, sizeof(double) * (size_t)w);
            }
        } else {
#pragma omp critical
            rc_all = rc;
        }
        free(p); free(u); free(fc);
    }
    if (pinned) sched_setaffinity(0, sizeof saved_mask, &saved_mask); /* the caller's thread gets its mask back */
    return rc_all;
}

int oracle_max_threads(void) {
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}
