// K3  integer rank statistics -> (p-value, U statistic, fold change), transposed to [group][gene].
//
// Restates compute_pval (illico/utils/math.py:64-118, fastmath=False) operation by operation in
// IEEE float64 (this translation unit is built with -ffp-contract=off), and
// fold_change_from_summed_expr (utils/math.py:168-193).
#pragma once
#include "common.h"

struct GroupConst;
struct FinalizeParams {
    const long long *in_2u;   // [nb][G]  2*U
    const u64 *in_tie;        // [nb][G]  tie sum
    const double *in_sum;     // [nb][G]  per-group sum of (expm1'd) values
    const double *gene_total; // [nb] sum over groups, OVR only (utils/math.py:185)
    const int *counts;        // [G]
    const GroupConst *gconst; // [G] (declared below) for this call's test: OVO against ref, or OVR
    int G, ref, nb;           // ref == -1 => OVR
    long long n_cells;
    int use_continuity, tie_correct, alternative;
    double *out_p, *out_u, *out_fc; // [G][out_ld], column offset already applied
    long long out_ld;
    const int *col_map;       // optional: output column of batch gene j (relative to the offset); nullptr = j
    int tie_f64;              // in_tie holds the BITS of a float64: the tie sum as the reference's sparse OVR path accumulates it
                              // (tie_f64_sparse below), not an exact integer
    int packed;               // 16-byte statistics (k_csc_counts): in_2u = value sum << 40 | 2U (40 bits, two's complement: -2 = the OVO reference row); no in_sum
};

// compute_pval with its per-(group) constants handed in -- nnn = (double)(n (n-1) (n+1)) (math.py:95), var0 = (double)(n_ref n_tgt
// (n_ref + n_tgt + 1)) / 12.0 (:97, the part in front of "* tie_corr"), n12 = (double)(n_ref n_tgt) -- so that a kernel which computes
// many genes of one group forms them once: the same operations on the same values, bit for bit
// The reference forms the tie sum in a float64 accumulator.  Its dense paths add exact integers, block by block
// (utils/ranking.py:30-47); its sparse OVR path adds the non-zero blocks first and then `n0**3 - n0` with n0 a float64
// (ovr/sparse_ovr.py:49,83): beyond n0 ~ 2.1e5 zeros n0^3 leaves float64's 53 bits, the two differ in the last bits, and at
// z ~ 36 those bits show in p at the 1e-12 level (tests/test_gpu_tail.py).  t_nonzero: the exact sum over the non-zero blocks.
__device__ __forceinline__ u64 tie_f64_sparse(u64 t_nonzero, long long n_zeros) {
    const double n0 = (double)n_zeros;
    double t = (double)t_nonzero;
    t += n0 * n0 * n0 - n0;
    return (u64)__double_as_longlong(t);
}

__device__ __forceinline__ double pval_device_pre(double nnn, double var0, double n12, double tie_sum, double U, double mu, double cc, int alternative) {
    double tie_corr = 1.0 - tie_sum / nnn;                                         // math.py:95
    if (tie_corr > 1.0e-9) {                                                       // :96
        double sigma = sqrt(var0 * tie_corr);                                      // :97
        if (alternative == 0) {                                                    // :99-104
            double other = n12 - U;
            U = (U < other) ? U : other;
            double delta = U - mu;
            double sgn = (delta > 0.0) ? 1.0 : ((delta < 0.0) ? -1.0 : 0.0);
            double z = (fabs(delta) + sgn * cc) / sigma;
            return erfc(z / sqrt(2.0));
        } else if (alternative == 2) {                                             // greater :105-109
            double z = ((U - mu) - cc) / sigma;
            return 0.5 * erfc(z / sqrt(2.0));
        } else {                                                                   // less :110-114
            double z = ((U - mu) + cc) / sigma;
            return 0.5 * erfc(-z / sqrt(2.0));
        }
    }
    return 1.0;                                                                    // :117-118
}
// n (n - 1) (n + 1) and n_ref n_tgt (n_ref + n_tgt + 1) as float64.  Up to 2^21 - 1 cells the int64 products of the reference
// (utils/math.py:95,97) are exact and converted once -- bit for bit the reference.  Beyond, its int64 WRAPS (the reference is wrong
// there); here the last factor is multiplied in float64 instead: n (n - 1) and n_ref n_tgt are still exact integers below 2^53 / 2^62,
// so the result is the correctly rounded product.  (Sparse OVR only: illico_set_groups refuses larger dense / OVO problems.)
__host__ __device__ inline double pval_nnn(long long n) {
    return n < 2097152ll ? (double)(n * (n - 1) * (n + 1)) : (double)(n * (n - 1)) * (double)(n + 1);
}
__host__ __device__ inline double pval_var0(long long n_ref, long long n_tgt) {
    const long long n1 = n_ref + n_tgt + 1;
    return (n1 <= 2097152ll ? (double)(n_ref * n_tgt * n1) : (double)(n_ref * n_tgt) * (double)n1) / 12.0;
}
__device__ __forceinline__ double pval_device(long long n_ref, long long n_tgt, long long n, double tie_sum, double U,
                                              double mu, double cc, int alternative) {
    return pval_device_pre(pval_nnn(n), pval_var0(n_ref, n_tgt), (double)(n_ref * n_tgt), tie_sum, U, mu, cc, alternative);
}

// What compute_pval and fold_change_from_summed_expr form from a group's SIZES alone (math.py:95,97,100 and :186-188): formed once
// per illico_set_groups (for the context's test: OVO against the reference group, or OVR) instead of once per test -- the same
// operations on the same values, bit for bit.  (The two group means of the fold change stay true divisions: taking them by
// reciprocals of the sizes saves two of the three divisions per test -- k_finalize 0.188 -> 0.161 ms at C3, nothing at C2 / C4 -- but
// moves the fold change by a unit in the last place, and tests/test_gpu_determinism.py holds it to the reference's bits:
// profiles/NOTES_r04.md.)
struct GroupConst { double d_tgt, d_ref, var0, n12, nnn, mu; };
__host__ __device__ inline GroupConst group_const(long long n_ref, long long n_tgt, long long n) {
    GroupConst c;
    c.d_tgt = (double)n_tgt;
    c.d_ref = (double)n_ref;
    c.var0 = pval_var0(n_ref, n_tgt);
    c.n12 = (double)(n_ref * n_tgt);
    c.nnn = pval_nnn(n);
    c.mu = (double)(n_ref * n_tgt) / 2.0;
    return c;
}
// math.py:181-192; ref_part: the reference group's value sum (OVO) / the column total minus the group's (OVR)
__device__ __forceinline__ double fold_change_device(double sum_g, double ref_part, const GroupConst &c) {
    const double mu_tgt = sum_g / c.d_tgt, mu_ref = ref_part / c.d_ref;
    return (mu_ref == 0.0) ? __longlong_as_double(0x7FF0000000000000ll) : mu_tgt / mu_ref;
}

// 32 genes x 32 groups per block; stats are read coalesced along groups, results written coalesced
// along genes.
static __global__ __launch_bounds__(256) void k_finalize(FinalizeParams P) {
    __shared__ double tp[32][33], tu[32][33], tf[32][33];
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5; // ty 0..7
    const int gene0 = blockIdx.x * 32, grp0 = blockIdx.y * 32;
    const bool ovr = P.ref < 0;
    const double cc = P.use_continuity ? 0.5 : 0.0;
    // every load of this thread's four (gene, group) pairs is requested before any of the arithmetic: the kernel moves 48 bytes
    // per test and does not compute much -- what it must not do is wait for memory four times in a row
    const int g = grp0 + tx;
    const bool gok = g < P.G;
    const GroupConst gc = P.gconst[gok ? g : 0]; // this thread's four tests are four genes of ONE group
    // OVO: the reference group's mean is a gene's, not a test's (math.py:183): 32 divisions per block instead of 1024
    __shared__ double s_mref[32];
    if (!ovr && threadIdx.x < 32) {
        const int gene = gene0 + (int)threadIdx.x;
        double rs = 1.0;
        if (gene < P.nb) rs = P.packed ? (double)((u64)P.in_2u[(size_t)gene * P.G + P.ref] >> 40) : P.in_sum[(size_t)gene * P.G + P.ref];
        s_mref[threadIdx.x] = rs / (double)P.counts[P.ref];
    }
    long long in2u[4];
    u64 intie[4];
    double insum[4], inref[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int gene = gene0 + ty + 8 * k;
        const bool ok = gok && gene < P.nb;
        const size_t o = ok ? (size_t)gene * P.G + g : 0;
        in2u[k] = ok ? P.in_2u[o] : 0;
        intie[k] = (ok && P.tie_correct) ? P.in_tie[o] : 0ull;
        if (P.packed) { // (uniform)
            insum[k] = 0.0;
            inref[k] = (ok && ovr) ? P.gene_total[gene] : 1.0;
        } else {
            insum[k] = ok ? P.in_sum[o] : 0.0;
            inref[k] = (ok && ovr) ? P.gene_total[gene] : 1.0;
        }
    }
    if (P.packed) {
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const u64 w = (u64)in2u[k], lo = w & 0xFFFFFFFFFFull;
            insum[k] = (double)(w >> 40);
            in2u[k] = (long long)(lo << 24) >> 24; // the 40-bit field sign-extended: 2U of a ranked group is below 2^39, the OVO reference row carries -2
        }
    }
    __syncthreads(); // s_mref
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int gy = ty + 8 * k, gene = gene0 + gy;
        if (gene < P.nb && gok) {
            double U = 0.5 * (double)in2u[k];
            double tie = !P.tie_correct ? 0.0 : (P.tie_f64 ? __longlong_as_double((long long)intie[k]) : (double)intie[k]);
            double p;
            if (!ovr && g == P.ref) { p = 1.0; U = -1.0; }                         // sparse_ovo.py:140-143
            else p = pval_device_pre(gc.nnn, gc.var0, gc.n12, tie, U, gc.mu, cc, P.alternative);
            const double sum_g = insum[k];
            double fc;
            if (ovr) fc = fold_change_device(sum_g, inref[k] - sum_g, gc);
            else {
                const double mu_ref = s_mref[gy];
                fc = (mu_ref == 0.0) ? __longlong_as_double(0x7FF0000000000000ll) : (sum_g / gc.d_tgt) / mu_ref;
            }
            tp[gy][tx] = p;
            tu[gy][tx] = U;
            tf[gy][tx] = fc;
        }
    }
    __syncthreads();
    for (int gy = ty; gy < 32; gy += 8) {
        int g = grp0 + gy, gene = gene0 + tx;
        if (gene < P.nb && g < P.G) {
            size_t o = (size_t)g * P.out_ld + (P.col_map ? P.col_map[gene] : gene);
            P.out_p[o] = tp[tx][gy];
            P.out_u[o] = tu[tx][gy];
            P.out_fc[o] = tf[tx][gy];
        }
    }
}

// per-gene sum over groups, rows added in group order like group_agg_counts.sum(axis=0) (math.py:185).
// One workgroup per 64 genes: [64 genes][64 groups] tiles of in_sum ([gene][G], so a gene's groups are contiguous) are
// read coalesced along groups and handed through LDS to one thread per gene, which adds them in group order.
static __global__ __launch_bounds__(256) void k_gene_totals(const double *in_sum, int G, int nb, double *gene_total) {
    __shared__ double tile[64][65];
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6; // ty 0..3
    const int gene0 = blockIdx.x * 64;
    double t = 0.0;
    double nx[16]; // the next tile waits in registers while this one is added up (one load latency per tile, not two)
    auto fetch = [&](int g0) {
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const int gene = gene0 + ty + 4 * i, g = g0 + tx;
            nx[i] = (gene < nb && g < G) ? in_sum[(size_t)gene * G + g] : 0.0;
        }
    };
    fetch(0);
    for (int g0 = 0; g0 < G; g0 += 64) {
#pragma unroll
        for (int i = 0; i < 16; ++i) tile[ty + 4 * i][tx] = nx[i];
        __syncthreads();
        if (g0 + 64 < G) fetch(g0 + 64);
        if (ty == 0) {
            const int lim = min(64, G - g0);
            for (int k = 0; k < lim; ++k) t += tile[tx][k];
        }
        __syncthreads();
    }
    if (ty == 0 && gene0 + tx < nb) gene_total[gene0 + tx] = t;
}
