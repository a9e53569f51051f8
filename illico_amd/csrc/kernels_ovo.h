// Dense/sparse one-versus-reference (OVO) kernels.
//
// Replaces, on device, the per-chunk work of the reference's
//   dense_ovo_mwu_kernel_over_contiguous_col_chunk   (illico/ovo/dense_ovo.py:65-137)
//   multi_group_sparse_ovo_mwu_kernel                (illico/ovo/sparse_ovo.py:103-158)
// i.e. chunk_and_fortranize (utils/math.py:247-278), the per-column sorts (utils/ranking.py:161-172,
// 200-220) and rank_sum_and_ties_from_sorted (utils/ranking.py:52-158).
//
// Not a translation of the two-pointer merge.  With A = reference values, B = one group's values:
//   R_B (rank sum of B in A u B, average ranks) = n_B(n_B+1)/2 + sum_b [ #A<b + (#A==b)/2 ]
//   U_A = n_A n_B + n_B(n_B+1)/2 - R_B = n_A n_B - S2/2,   S2 = sum_b [ 2 #A<b + #A==b ]   (integer)
//   tie_sum = sum_v (tA+tB)^3-(tA+tB) = T_A + sum_{distinct v in B} tB (3 tA (tA+tB) + tB^2 - 1)
// so the reference column is sorted once per gene into LDS, every other group is sorted by one
// wavefront in registers, its distinct values are compacted and each is binary-searched in LDS.
// Everything is integer arithmetic => rank sums / U are bit-exact.
#pragma once
#include "common.h"

// ---------------------------------------------------------------------------------------------
// K1  transpose + row permutation + key conversion
//   X row-major [N, ld] (any supported dtype)  ->  Xt[gene][pos] keys, pos = position in the
//   group-contiguous cell order (perm[pos] = cell index; perm = GroupContainer.indices).
//   64x64 tile through LDS: reads 64 consecutive genes of one row (256 B for f32), writes 64
//   consecutive positions of one gene.
// ---------------------------------------------------------------------------------------------
template <typename InT, typename KeyT>
__global__ __launch_bounds__(256) void k_transpose_permute(const InT *__restrict__ X, long long ld, long long col0,
                                                           int ncols, const int *__restrict__ perm, int N,
                                                           KeyT *__restrict__ Xt, long long xt_stride) {
    __shared__ KeyT tile[64][65];
    const int p0 = blockIdx.x * 64, c0 = blockIdx.y * 64;
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
#pragma unroll 4
    for (int r = ty; r < 64; r += 4) {
        int p = p0 + r, c = c0 + tx;
        KeyT k = KeyInfo<KeyT>::MAXK;
        if (p < N && c < ncols) {
            long long row = perm[p];
            k = key_of(X[row * ld + col0 + c]);
        }
        tile[r][tx] = k;
    }
    __syncthreads();
#pragma unroll 4
    for (int cc = ty; cc < 64; cc += 4) {
        int c = c0 + cc, p = p0 + tx;
        if (c < ncols && p < N) Xt[(long long)c * xt_stride + p] = tile[tx][cc];
    }
}

// ---------------------------------------------------------------------------------------------
// K2  per-gene OVO ranking
// ---------------------------------------------------------------------------------------------
struct OvoParams {
    const void *Xs;          // keys, gene-major, group-contiguous
    long long gene_stride;   // dense layout: keys per gene row
    const int *pos_ptr;      // dense layout: [G+1] first position of each group (GroupContainer.indptr)
    const u32 *seg_ptr;      // sparse layout: [n_genes][G+1] offsets into Xs of each (gene, group) run of non-zeros
    const int *counts;       // [G] cells per group
    int G, ref, n_genes, dt, is_log1p;
    int ref_cap;             // LDS slots reserved for the reference column
    long long *out_2u;       // [n_genes][G]  2*U (U of the reference sample, as scipy's mannwhitneyu(ref, grp))
    u64 *out_tie;            // [n_genes][G]  sum_v (t^3 - t)
    double *out_sum;         // [n_genes][G]  sum of values (expm1'd if is_log1p)
};

template <typename KeyT, int K, bool RUNEND>
__device__ __forceinline__ void ovo_wave_group(const KeyT *__restrict__ seg, int nB, const KeyT *A,
                                               const u16 *runend, u32 nA, u32 topA, u32 zA, KeyT *sk, u32 *sb,
                                               int lane, int dt, int is_log1p, u64 &S2out, u64 &tieout,
                                               double &sumout) {
    constexpr KeyT MAXK = KeyInfo<KeyT>::MAXK;
    constexpr KeyT ZEROK = KeyInfo<KeyT>::ZEROK;
    KeyT v[K];
    double s = 0.0;
#pragma unroll
    for (int r = 0; r < K; ++r) {
        int idx = r * 64 + lane;
        bool ok = idx < nB;
        v[r] = ok ? seg[idx] : MAXK;
        if (ok) s += is_log1p ? key_to_expm1(v[r], dt) : key_to_double(v[r], dt);
    }
    wave_bitonic_sort<KeyT, K>(v, lane);

    // run heads (first element of each run of equal keys), in sorted order i = lane*K + r
    KeyT prev_last = __shfl_up(v[K - 1], 1);
    bool head[K];
    const int BIG = 0x7FFFFFFF;
    int fh = BIG, hc = 0;
#pragma unroll
    for (int r = K - 1; r >= 0; --r) {
        int idx = lane * K + r;
        KeyT prev = (r == 0) ? prev_last : v[r - 1];
        head[r] = (idx < nB) && (idx == 0 || v[r] != prev);
        if (head[r]) { fh = idx; ++hc; }
    }
    // first head strictly after this lane (exclusive suffix-min over lanes), clipped to nB
    int m = fh;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        int o = __shfl_down(m, d);
        if (lane + d < 64) m = min(m, o);
    }
    int carry = __shfl_down(m, 1);
    if (lane == 63) carry = BIG;
    carry = min(carry, nB);
    u32 tB[K];
    {
        int nxt = carry;
#pragma unroll
        for (int r = K - 1; r >= 0; --r) {
            int idx = lane * K + r;
            tB[r] = (u32)(nxt - idx);
            if (head[r]) nxt = idx;
        }
    }
    // rank of each head among the heads (exclusive scan over lanes)
    int incl = hc;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        int o = __shfl_up(incl, d);
        if (lane >= d) incl += o;
    }
    const int H = __shfl(incl, 63);
    const int excl = incl - hc;

    u64 S2 = 0, tie = 0;
    for (int base = 0; base < H; base += 256) {
        int hr = excl - base;
#pragma unroll
        for (int r = 0; r < K; ++r) {
            if (head[r]) {
                if (hr >= 0 && hr < 256) { sk[hr] = v[r]; sb[hr] = tB[r]; }
                ++hr;
            }
        }
        wave_lds_fence();
        const int cnt = min(256, H - base);
        for (int j = 0; j * 64 < cnt; ++j) {
            int slot = j * 64 + lane;
            if (slot < cnt) {
                KeyT q = sk[slot];
                u64 b = sb[slot];
                u32 lb = lower_bound_pow2(A, nA, topA, q);
                u64 a = 0;
                if (lb < nA && A[lb] == q) {
                    if (RUNEND) a = (u32)runend[lb] - lb;
                    else a = upper_bound_pow2(A, nA, topA, q) - lb;
                }
                u64 lt = (u64)lb + ((q > ZEROK) ? (u64)zA : 0ull);  // implicit zeros of A rank below positives
                S2 += b * (2ull * lt + a);
                tie += b * (3ull * a * (a + b) + b * b - 1ull);
            }
        }
        wave_lds_fence();
    }
    S2out = wave_sum(S2);
    tieout = wave_sum(tie);
    sumout = wave_sum(s);
}

template <typename KeyT, int KMAX, bool RUNEND, int NT>
__global__ __launch_bounds__(NT) void k_ovo_rank(OvoParams P) {
    extern __shared__ __align__(16) unsigned char smem[];
    constexpr int NW = NT / 64;
    constexpr KeyT ZEROK = KeyInfo<KeyT>::ZEROK;
    // LDS carve (all offsets multiples of 16 B): A | runend | per-wave compaction scratch | reduction words
    KeyT *A = (KeyT *)smem;
    size_t off = ((size_t)P.ref_cap * sizeof(KeyT) + 15) & ~(size_t)15;
    u16 *runend = (u16 *)(smem + off);
    if (RUNEND) off += ((size_t)P.ref_cap * sizeof(u16) + 15) & ~(size_t)15;
    KeyT *sk_all = (KeyT *)(smem + off);
    off += (size_t)NW * 256 * sizeof(KeyT);
    u32 *sb_all = (u32 *)(smem + off);
    off += (size_t)NW * 256 * sizeof(u32);
    u64 *s_red = (u64 *)(smem + off);      // [NW]
    double *s_redd = (double *)(s_red + NW); // [NW]
    u64 *s_TA = (u64 *)(s_redd + NW);      // [1]
    double *s_refsum = (double *)(s_TA + 1); // [1]

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const KeyT *Xs = (const KeyT *)P.Xs;
    const int G = P.G, ref = P.ref;
    const int n_ref = P.counts[ref];

    for (int gene = blockIdx.x; gene < P.n_genes; gene += gridDim.x) {
        // ---- reference column -> LDS, sorted ----
        long long rstart;
        u32 nA;
        const u32 *sp = nullptr;
        if (P.seg_ptr) {
            sp = P.seg_ptr + (size_t)gene * (G + 1);
            rstart = sp[ref];
            nA = sp[ref + 1] - sp[ref];
        } else {
            rstart = (long long)gene * P.gene_stride + P.pos_ptr[ref];
            nA = (u32)n_ref;
        }
        const u32 zA = (u32)n_ref - nA;
        double rs = 0.0;
        for (u32 i = tid; i < nA; i += NT) {
            KeyT k = Xs[rstart + i];
            A[i] = k;
            rs += P.is_log1p ? key_to_expm1(k, P.dt) : key_to_double(k, P.dt);
        }
        rs = wave_sum(rs);
        if (lane == 0) s_redd[wave] = rs;
        __syncthreads();
        block_bitonic_sort<KeyT, NT>(A, (int)nA, tid);
        const u32 topA = top_pow2(nA);
        // run ends at run heads (the only slots a lower bound can land on) and T_A = sum (tA^3 - tA)
        u64 ta = 0;
        for (u32 i = tid; i < nA; i += NT) {
            KeyT k = A[i];
            if (i == 0 || A[i - 1] != k) {
                u32 e = upper_bound_pow2(A, nA, topA, k);
                if (RUNEND) runend[i] = (u16)e;
                u64 t = e - i;
                ta += t * t * t - t;
            }
        }
        ta = wave_sum(ta);
        if (lane == 0) s_red[wave] = ta;
        __syncthreads();
        if (tid == 0) {
            u64 t = 0;
            double d = 0.0;
            for (int w = 0; w < NW; ++w) { t += s_red[w]; d += s_redd[w]; }
            *s_TA = t;
            *s_refsum = d;
        }
        __syncthreads();
        const u64 T_A = *s_TA;
        const double refsum = *s_refsum;
        const u32 nnegA = P.seg_ptr ? lower_bound_pow2(A, nA, topA, ZEROK) : 0u;

        // ---- every other group: one wavefront each, 64 groups per output block ----
        KeyT *sk = sk_all + wave * 256;
        u32 *sb = sb_all + wave * 256;
        for (int g0 = wave * 64; g0 < G; g0 += NW * 64) {
            long long r2u = 0;
            u64 rtie = 0;
            double rsum = 0.0;
            const int jn = min(64, G - g0);
            for (int j = 0; j < jn; ++j) {
                const int g = g0 + j;
                if (g == ref) {
                    if (lane == j) { r2u = -2; rtie = 0; rsum = refsum; }
                    continue;
                }
                const int n_g = P.counts[g];
                long long bstart;
                int nB;
                if (sp) { bstart = sp[g]; nB = (int)(sp[g + 1] - sp[g]); }
                else { bstart = (long long)gene * P.gene_stride + P.pos_ptr[g]; nB = n_g; }
                const u32 zB = (u32)(n_g - nB);
                u64 S2 = 0, tie = 0;
                double sum = 0.0;
                const KeyT *seg = Xs + bstart;
                if (nB <= 64) ovo_wave_group<KeyT, 1, RUNEND>(seg, nB, A, runend, nA, topA, zA, sk, sb, lane, P.dt, P.is_log1p, S2, tie, sum);
                else if (nB <= 128) ovo_wave_group<KeyT, 2, RUNEND>(seg, nB, A, runend, nA, topA, zA, sk, sb, lane, P.dt, P.is_log1p, S2, tie, sum);
                else if (nB <= 256) ovo_wave_group<KeyT, 4, RUNEND>(seg, nB, A, runend, nA, topA, zA, sk, sb, lane, P.dt, P.is_log1p, S2, tie, sum);
                else if (KMAX >= 8 && nB <= 512) ovo_wave_group<KeyT, (KMAX >= 8 ? 8 : 4), RUNEND>(seg, nB, A, runend, nA, topA, zA, sk, sb, lane, P.dt, P.is_log1p, S2, tie, sum);
                else if (KMAX >= 16 && nB <= 1024) ovo_wave_group<KeyT, (KMAX >= 16 ? 16 : 4), RUNEND>(seg, nB, A, runend, nA, topA, zA, sk, sb, lane, P.dt, P.is_log1p, S2, tie, sum);
                // implicit zeros of B (sparse layout): each ranks above A's negatives and ties with A's zeros
                S2 += (u64)zB * (2ull * nnegA + zA);
                const u64 t0 = (u64)zA + zB;
                const u64 tie_total = T_A + tie + (t0 * t0 * t0 - t0);
                if (lane == j) {
                    r2u = 2ll * (long long)n_ref * (long long)n_g - (long long)S2;
                    rtie = tie_total;
                    rsum = sum;
                }
            }
            if (lane < jn) {
                size_t o = (size_t)gene * G + g0 + lane;
                P.out_2u[o] = r2u;
                P.out_tie[o] = rtie;
                P.out_sum[o] = rsum;
            }
        }
        __syncthreads();
    }
}
