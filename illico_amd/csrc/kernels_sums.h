// Per-(gene, group) value sums for the fold change -- dense_fold_change / csc_fold_change / csr_fold_change
// (illico/utils/math.py:196-221, utils/sparse/csc.py:186-211, utils/sparse/csr.py:261-286) -- wherever the rank kernels
// meet a group's values in an order that depends on timing (LDS / global atomics regroup the stored entries of a sparse
// column).  A float64 sum depends on the order of its additions; the reference adds in cell-index order.  These kernels
// make the result independent of the order instead:
//
//  * EXACT sums (k_csc_value_sums, k_seg_value_sums): every value is turned into an 84-bit fixed-point integer whose unit
//    is 2^-83 of the gene's largest magnitude, split into two 42-bit limbs, and the limbs are added as 64-bit INTEGERS
//    (atomics in LDS / per-lane registers): integer addition is associative, so any order of arrival gives the same two
//    totals, and the pair is rounded to float64 ONCE.  float32 values within 2^-59 of the gene's largest magnitude (float64:
//    2^-30) are represented exactly, smaller ones are truncated at 2^-83 of it.  A (gene, group) sum holds up to 2^21
//    values (the per-test cell limit of illico_set_groups).  The result is the correctly rounded exact sum, i.e. it
//    differs from the reference's sequential float64 sum only by the reference's own rounding error (<= n eps / 2).
//  * fixed-order sums (k_group_sums_rows): group-contiguous key rows (what k_transpose_permute writes) are summed by one
//    wavefront per group, lane-strided, then a fixed butterfly: the same additions in the same order on every run.
#pragma once
#include "common.h"

#define SUMS_NT 256
#define EXS_LIMB 42 // payload bits per limb

// 2^k with k in [-1022, 1023]
__device__ __forceinline__ double exs_pow2(int k) { return __longlong_as_double((long long)(k + 1023) << 52); }

// scale of a gene whose largest magnitude is vmax (finite, > 0): vmax * 2^k lies in [2^83, 2^84).
struct ExsScale {
    int k;         // v -> trunc(v * 2^k)
    double u1, u2; // inverse: x * u1 * u2 (two exact power-of-two steps, so that neither factor leaves the normal range)
};
__device__ __forceinline__ ExsScale exs_scale(double vmax) {
    int e = (int)((__double_as_longlong(vmax) >> 52) & 0x7FF);
    e = (e ? e : 1) - 1023;                       // floor(log2(vmax)) for normal vmax; subnormals count as 2^-1022
    ExsScale S;
    S.k = 2 * EXS_LIMB - 1 - e;                   // in [-940, 1105]
    const int k1 = S.k / 2, k2 = S.k - k1;
    S.u1 = exs_pow2(-k1); S.u2 = exs_pow2(-k2);
    return S;
}
// v -> (l1, l0): trunc(v * 2^k) = l1 * 2^42 + l0, both limbs carry v's sign.  Integer arithmetic on the bits of v (a 53-bit
// significand shifted into place): the float64 form of the same split (multiply, trunc, subtract, two f64 -> i64 conversions)
// made the sums kernels VALU-bound.
__device__ __forceinline__ void exs_split(double v, const ExsScale &S, long long &l0, long long &l1) {
    const u64 b = (u64)__double_as_longlong(v);
    const int ef = (int)((b >> 52) & 0x7FF);
    u64 m = (b & 0x000FFFFFFFFFFFFFull) | (ef ? 0x0010000000000000ull : 0ull); // |v| = m * 2^(max(ef, 1) - 1075)
    int p = (ef ? ef : 1) - 1075 + S.k;                                        // trunc(|v| 2^k) = m * 2^p; p <= 31 as |v| <= vmax
    if (p < 0) { m = p > -64 ? m >> (-p) : 0ull; p = 0; }                      // truncation toward zero: the one inexact step
    const int cut = EXS_LIMB - p;                                              // bits of m that stay in the low limb (11 .. 42)
    u64 hi = m >> cut, lo = (m & ((1ull << cut) - 1ull)) << p;
    const bool neg = (long long)b < 0;
    l1 = neg ? -(long long)hi : (long long)hi;
    l0 = neg ? -(long long)lo : (long long)lo;
}
// (L1 * 2^42 + L0) / sc, rounded to float64 once
__device__ __forceinline__ double exs_combine(long long L0, long long L1, const ExsScale &S) {
    // 128-bit two's complement T = L1 * 2^42 + L0
    u64 lo = (u64)L1 << EXS_LIMB;
    long long hi = L1 >> (64 - EXS_LIMB);
    const u64 lo2 = lo + (u64)L0;
    hi += (L0 >> 63) + (lo2 < lo ? 1 : 0);
    lo = lo2;
    const bool neg = hi < 0;
    if (neg) { // magnitude
        lo = ~lo + 1ull;
        hi = ~hi + (lo == 0ull ? 1 : 0);
    }
    double r;
    if (hi == 0) r = (double)lo; // u64 -> f64 is rounded to nearest even
    else { // keep 64 significant bits, fold everything below into a sticky bit: one rounding, ties decided correctly
        const int s = 64 - __clzll(hi);
        u64 top = ((u64)hi << (64 - s)) | (lo >> s);
        if ((lo << (64 - s)) != 0ull) top |= 1ull;
        r = (double)top * exs_pow2(s);
    }
    r = r * S.u1 * S.u2;
    return neg ? -r : r;
}

template <typename InT> __device__ __forceinline__ double sums_value(InT v, int dt, int is_log1p) {
    return is_log1p ? key_to_expm1(key_of(v), dt) : (double)v;
}
__device__ __forceinline__ double block_max_f64(double x, double *s_red, int tid) { // SUMS_NT threads; NaN-propagating max of |.|
    constexpr int NW = SUMS_NT / 64;
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) {
        const double o = __shfl_xor(x, d);
        x = (o > x || o != o) ? o : x;
    }
    __syncthreads();
    if ((tid & 63) == 0) s_red[tid >> 6] = x;
    __syncthreads();
    double m = s_red[0];
    for (int w = 1; w < NW; ++w) { const double o = s_red[w]; m = (o > m || o != o) ? o : m; }
    __syncthreads();
    return m;
}

// ---- CSC arrays -> out_sum[gene][G]: one workgroup per gene -----------------------------------------------------------------
struct CscSumsParams {
    const void *data, *indices, *indptr; // stored entry k lives at data[k - kshift], indices[k - kshift]
    long long kshift;
    long long col0;                      // first gene of the batch (contiguous batches)
    const int *gene_cols;                // or the batch's genes as a column list; nullptr = contiguous
    const int *codes;                    // [n_cells] group code per cell; nullptr: `indices` already holds group codes
    const u16 *codes16;                  // the same as 16-bit values (fewer cache lines per gather), or nullptr
    int nb, G, dt, is_log1p;
    long long *acc_global;               // ACCG: [nb][2][G] limb totals in HBM (zeroed by the host) when 16 G bytes exceed LDS
    double *out_sum;                     // [nb][G]
};
#define CSUM_NT 512
template <typename InT, typename IdxT, bool ACCG>
__global__ __launch_bounds__(CSUM_NT, 4) void k_csc_value_sums(CscSumsParams P) {
    constexpr int NT = CSUM_NT, UL = 8, NW = NT / 64;
    extern __shared__ __align__(16) unsigned char smem[];
    double *s_red = (double *)smem;                 // [NW]
    long long *L0 = (long long *)(smem + 64);       // [G]
    long long *L1 = L0 + P.G;                       // [G]
    const int tid = threadIdx.x, G = P.G;
    const InT *data = (const InT *)P.data;
    const IdxT *indices = (const IdxT *)P.indices, *indptr = (const IdxT *)P.indptr;
    for (int gene = blockIdx.x; gene < P.nb; gene += gridDim.x) {
        const long long col = P.gene_cols ? (long long)P.gene_cols[gene] : P.col0 + gene;
        const long long k0 = (long long)indptr[col] - P.kshift, k1 = (long long)indptr[col + 1] - P.kshift;
        if constexpr (ACCG) { L0 = P.acc_global + (size_t)gene * 2 * G; L1 = L0 + G; }
        else for (int g = tid; g < G; g += NT) { L0[g] = 0; L1[g] = 0; }
        // ---- the gene's largest magnitude (UL independent loads per thread and round) ----
        double vmax = 0.0;
        for (long long kb = k0; kb < k1; kb += (long long)NT * UL) {
            InT v[UL];
#pragma unroll
            for (int u = 0; u < UL; ++u) { const long long k = kb + u * NT + tid; v[u] = k < k1 ? data[k] : (InT)0; }
#pragma unroll
            for (int u = 0; u < UL; ++u) {
                const double a = fabs(sums_value(v[u], P.dt, P.is_log1p));
                vmax = (a > vmax || a != a) ? a : vmax;
            }
        }
        {   // block max, NaN-propagating (its barriers also order the zeroing above)
#pragma unroll
            for (int d = 32; d > 0; d >>= 1) { const double o = __shfl_xor(vmax, d); vmax = (o > vmax || o != o) ? o : vmax; }
            __syncthreads();
            if ((tid & 63) == 0) s_red[tid >> 6] = vmax;
            __syncthreads();
            vmax = s_red[0];
            for (int w = 1; w < NW; ++w) { const double o = s_red[w]; vmax = (o > vmax || o != o) ? o : vmax; }
            __syncthreads();
        }
        const bool finite = vmax < __longlong_as_double(0x7FF0000000000000ll); // false for inf and NaN
        double *out = P.out_sum + (size_t)gene * G;
        if (vmax == 0.0) { // uniform: nothing but (stored) zeros
            for (int g = tid; g < G; g += NT) out[g] = 0.0;
            __syncthreads();
            continue;
        }
        const bool exact = finite;
        ExsScale S;
        S.k = 0; S.u1 = S.u2 = 1.0;
        if (exact) S = exs_scale(vmax);
        double *F = (double *)L0; // inf / NaN among the values: the sums are inf / NaN whatever the order; plain float64 adds
        // ---- two-stage pipeline over the entries (as k_csc_counts): the values / rows of round i + 1 are requested before
        // round i's codes[row] gather and LDS atomics ----
        auto entry_loop = [&](auto code_of) {
            InT vn[UL];
            IdxT in[UL];
#pragma unroll
            for (int u = 0; u < UL; ++u) {
                const long long k = k0 + u * NT + tid;
                vn[u] = k < k1 ? data[k] : (InT)0;
                in[u] = k < k1 ? indices[k] : (IdxT)0;
            }
            for (long long kb = k0; kb < k1; kb += (long long)NT * UL) {
                InT v[UL];
                int cd[UL];
#pragma unroll
                for (int u = 0; u < UL; ++u) { v[u] = vn[u]; cd[u] = code_of(in[u]); }
                const long long kn = kb + (long long)NT * UL;
                if (kn < k1) { // uniform
#pragma unroll
                    for (int u = 0; u < UL; ++u) {
                        const long long k = kn + u * NT + tid;
                        vn[u] = k < k1 ? data[k] : (InT)0;
                        in[u] = k < k1 ? indices[k] : (IdxT)0;
                    }
                }
#pragma unroll
                for (int u = 0; u < UL; ++u)
                    if (v[u] != (InT)0) {
                        const double x = sums_value(v[u], P.dt, P.is_log1p);
                        if (exact) {
                            long long l0, l1;
                            exs_split(x, S, l0, l1);
                            if (l1) atomicAdd((u64 *)&L1[cd[u]], (u64)l1);
                            if (l0) atomicAdd((u64 *)&L0[cd[u]], (u64)l0);
                        } else atomicAdd(&F[cd[u]], x);
                    }
            }
        };
        if (P.codes16) { const u16 *t = P.codes16; entry_loop([t](IdxT row) { return (int)t[(long long)row]; }); }
        else if (P.codes) { const int *t = P.codes; entry_loop([t](IdxT row) { return t[(long long)row]; }); }
        else entry_loop([](IdxT row) { return (int)row; });
        if constexpr (ACCG) __threadfence();
        __syncthreads();
        for (int g = tid; g < G; g += NT) out[g] = exact ? exs_combine(L0[g], L1[g], S) : F[g];
        __syncthreads();
    }
}
static inline size_t csc_sums_lds_bytes(int G, bool accg) { return 64 + (accg ? 0 : (size_t)G * 16); }

// ---- keys regrouped by (gene, group) in HBM (the two-kernel sparse routes: Xs + seg_ptr) -> out_sum[gene][G] -----------------
struct SegSumsParams {
    const void *Xs;       // keys
    const u32 *seg_ptr;   // [nb][G + 1] run offsets into Xs
    int nb, G, dt, is_log1p;
    double *out_sum;      // [nb][G]
};
template <typename KeyT>
__global__ __launch_bounds__(SUMS_NT) void k_seg_value_sums(SegSumsParams P) {
    constexpr int NT = SUMS_NT, NW = NT / 64;
    __shared__ double s_red[NW];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, G = P.G;
    const KeyT *Xs = (const KeyT *)P.Xs;
    auto val = [&](KeyT k) { return P.is_log1p ? key_to_expm1(k, P.dt) : key_to_double(k, P.dt); };
    for (int gene = blockIdx.x; gene < P.nb; gene += gridDim.x) {
        const u32 *sp = P.seg_ptr + (size_t)gene * (G + 1);
        const u32 a0 = sp[0], a1 = sp[G];
        double vmax = 0.0;
        for (u32 i = a0 + tid; i < a1; i += NT) {
            const double a = fabs(val(Xs[i]));
            vmax = (a > vmax || a != a) ? a : vmax;
        }
        vmax = block_max_f64(vmax, s_red, tid);
        const bool finite = vmax < __longlong_as_double(0x7FF0000000000000ll);
        const bool exact = finite && vmax > 0.0;
        ExsScale S;
        if (exact) S = exs_scale(vmax);
        double *out = P.out_sum + (size_t)gene * G;
        for (int g = wave; g < G; g += NW) {
            const u32 p0 = sp[g], p1 = sp[g + 1];
            if (exact) {
                long long L0 = 0, L1 = 0;
                for (u32 i = p0 + lane; i < p1; i += 64) {
                    long long l0, l1;
                    exs_split(val(Xs[i]), S, l0, l1);
                    L0 += l0; L1 += l1;
                }
                L0 = (long long)wave_sum((u64)L0);
                L1 = (long long)wave_sum((u64)L1);
                if (lane == 0) out[g] = exs_combine(L0, L1, S);
            } else { // all zero, or an inf / NaN among the gene's values (sums are inf / NaN whatever the order)
                double s = 0.0;
                for (u32 i = p0 + lane; i < p1; i += 64) s += val(Xs[i]);
                s = wave_sum(s);
                if (lane == 0) out[g] = s;
            }
        }
    }
}

// ---- group-contiguous key rows -> out_sum[gene][G], fixed order ---------------------------------------------------------------
// One wavefront per 64 groups: for each group, lane l adds elements l, l + 64, ... of the group's run (the next group's first
// 256 keys are requested before this group is summed); the 64 per-lane partial vectors are folded by a transpose-reduce
// (common.h: TrReduce, 63 VALU combines for 64 groups instead of 64 six-step butterflies), after which lane j holds group j's
// sum and the block is stored with one coalesced write.  The order of additions is fixed by the positions alone.
template <typename KeyT, int RR = 4>
__device__ __forceinline__ void group_sums_row(const KeyT *__restrict__ ka, const int *__restrict__ pos_ptr, int G, int dt, int is_log1p,
                                               double *__restrict__ out, int wave, int lane, int n_waves) {
    auto val = [&](KeyT k) { return is_log1p ? key_to_expm1(k, dt) : key_to_double(k, dt); };
    for (int g0 = wave * 64; g0 < G; g0 += n_waves * 64) {
        TrReduce<double> red;
        KeyT nxt[RR];
        int np0 = 0, np1 = 0;
        auto fetch = [&](int g) {
            np0 = np1 = 0;
            if (g < G) { np0 = pos_ptr[g]; np1 = pos_ptr[g + 1]; }
#pragma unroll
            for (int r = 0; r < RR; ++r) {
                const int i = np0 + r * 64 + lane;
                nxt[r] = i < np1 ? ka[i] : (KeyT)0;
            }
        };
        fetch(g0);
        for (int j = 0; j < 64; ++j) { // always 64 pushes so that the transpose-reduce completes
            KeyT cur[RR];
#pragma unroll
            for (int r = 0; r < RR; ++r) cur[r] = nxt[r];
            const int p0 = np0, p1 = np1;
            fetch(g0 + j + 1 < g0 + 64 ? g0 + j + 1 : G);
            double s = 0.0;
#pragma unroll
            for (int r = 0; r < RR; ++r)
                if (p0 + r * 64 + lane < p1) s += val(cur[r]);
            for (int i = p0 + RR * 64 + lane; i < p1; i += 64) s += val(ka[i]);
            red.push(s, j, lane);
        }
        if (g0 + lane < G) out[g0 + lane] = red.result;
    }
}
template <typename KeyT>
__global__ __launch_bounds__(SUMS_NT) void k_group_sums_rows(const KeyT *__restrict__ Xt, long long stride, int n_genes, const int *__restrict__ pos_ptr,
                                                           int G, int dt, int is_log1p, double *__restrict__ out_sum) {
    const int tid = threadIdx.x;
    for (int gene = blockIdx.x; gene < n_genes; gene += gridDim.x)
        group_sums_row<KeyT>(Xt + (size_t)gene * stride, pos_ptr, G, dt, is_log1p, out_sum + (size_t)gene * G, tid >> 6, tid & 63, SUMS_NT / 64);
}
