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
// true iff v is an integer in [0, limit): the gene can take the histogram path (kernels_ovo_counts.h)
__device__ __forceinline__ bool count_ok(float v, int limit) { return v >= 0.0f && v < (float)limit && v == truncf(v); }
__device__ __forceinline__ bool count_ok(double v, int limit) { return v >= 0.0 && v < (double)limit && v == trunc(v); }
__device__ __forceinline__ bool count_ok(int32_t v, int limit) { return v >= 0 && v < limit; }
__device__ __forceinline__ bool count_ok(int64_t v, int limit) { return v >= 0 && v < (int64_t)limit; }

template <typename InT, typename KeyT>
__global__ __launch_bounds__(256) void k_transpose_permute(const InT *__restrict__ X, long long ld, long long col0,
                                                           int ncols, const int *__restrict__ perm, int N,
                                                           KeyT *__restrict__ Xt, long long xt_stride,
                                                           u32 *__restrict__ gene_flags, int count_limit) {
    __shared__ KeyT tile[64][65];
    const int p0 = blockIdx.x * 64, c0 = blockIdx.y * 64;
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
#pragma unroll 4
    for (int r = ty; r < 64; r += 4) {
        int p = p0 + r, c = c0 + tx;
        KeyT k = KeyInfo<KeyT>::MAXK;
        if (p < N && c < ncols) {
            long long row = perm[p];
            InT val = X[row * ld + col0 + c];
            k = key_of(val);
            if (gene_flags && !count_ok(val, count_limit) && gene_flags[c] == 0) gene_flags[c] = 1u;
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

// Vectorised variant: 16 B per lane on both sides (VEC = 16 / sizeof(InT) elements).  Requires X + row*ld
// + col0 to be 16-B aligned (checked by the host: pointer, ld and col0 multiples of VEC).  LDS tile is
// gene-major [64 genes][65]: both phases are at most 2-way bank conflicts.
template <typename InT, typename KeyT, int VEC>
__global__ __launch_bounds__(256) void k_transpose_permute_vec(const InT *__restrict__ X, long long ld, long long col0,
                                                               int ncols, const int *__restrict__ perm, int N,
                                                               KeyT *__restrict__ Xt, long long xt_stride,
                                                               u32 *__restrict__ gene_flags, int count_limit) {
    typedef InT __attribute__((ext_vector_type(VEC))) InV;
    typedef KeyT __attribute__((ext_vector_type(VEC))) KeyV;
    __shared__ KeyT tile[64][65];
    constexpr int LPR = 64 / VEC;      // lanes per 64-wide row / column
    constexpr int RPI = 256 / LPR;     // rows (or genes) per iteration
    const int p0 = blockIdx.x * 64, c0 = blockIdx.y * 64;
    const int q = threadIdx.x % LPR, r0 = threadIdx.x / LPR;
    bool viol[VEC];
#pragma unroll
    for (int e = 0; e < VEC; ++e) viol[e] = false;
#pragma unroll
    for (int r = r0; r < 64; r += RPI) {
        const int p = p0 + r, c = c0 + q * VEC;
        KeyT k[VEC];
#pragma unroll
        for (int e = 0; e < VEC; ++e) k[e] = KeyInfo<KeyT>::MAXK;
        if (p < N) {
            const InT *src = X + (long long)perm[p] * ld + col0 + c;
            if (c + VEC <= ncols) {
                InV v = *reinterpret_cast<const InV *>(src);
#pragma unroll
                for (int e = 0; e < VEC; ++e) { k[e] = key_of(v[e]); viol[e] |= !count_ok(v[e], count_limit); }
            } else {
#pragma unroll
                for (int e = 0; e < VEC; ++e)
                    if (c + e < ncols) { k[e] = key_of(src[e]); viol[e] |= !count_ok(src[e], count_limit); }
            }
        }
#pragma unroll
        for (int e = 0; e < VEC; ++e) tile[q * VEC + e][r] = k[e];
    }
    if (gene_flags) { // at most one (checked) store per gene per thread; a stale-L1 miss only repeats the store
#pragma unroll
        for (int e = 0; e < VEC; ++e) {
            const int c = c0 + q * VEC + e;
            if (viol[e] && c < ncols && gene_flags[c] == 0) gene_flags[c] = 1u;
        }
    }
    __syncthreads();
#pragma unroll
    for (int cc = r0; cc < 64; cc += RPI) {
        const int c = c0 + cc, p = p0 + q * VEC;
        if (c < ncols && p < N) { // rows are padded to a multiple of 64 keys, so the 16-B store stays inside the row
            KeyV o;
#pragma unroll
            for (int e = 0; e < VEC; ++e) o[e] = tile[cc][q * VEC + e];
            *reinterpret_cast<KeyV *>(Xt + (long long)c * xt_stride + p) = o;
        }
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
    int ref_buckets;         // 1: the reference column may take the bucket form (LDS sized for it by the host)
    long long *out_2u;       // [n_genes][G]  2*U (U of the reference sample, as scipy's mannwhitneyu(ref, grp))
    u64 *out_tie;            // [n_genes][G]  sum_v (t^3 - t)
    double *out_sum;         // [n_genes][G]  sum of values (expm1'd if is_log1p)
    // packed dense layout (kernels_ovo_compact.h): group g's NON-ZERO keys at gene * gene_stride + gofs[gene][g], nnz[gene][g] of
    // them (the reference's entries included); zeros are implicit, as in the sparse layout.  only: genes to process (flag 1), or null.
    const u16 *nnz = nullptr;
    const u32 *gofs = nullptr;
    const u32 *only = nullptr;
};

// The reference column in LDS, looked up by every value of the other groups.  Two forms, chosen per gene:
//  * sorted (A ascending, runend[i] = end of the run starting at i): lower bound by binary search, ~log2(nA) dependent
//    LDS reads at random addresses;
//  * bucketed (bk.on): the NON-ZERO reference keys are dealt into value buckets ((key - kmin) >> shift, clamped; order
//    inside a bucket does not matter) and bk.tab[b] = one past bucket b; a look-up reads two table entries and the few
//    keys of one bucket: #A<q = bucket start + smaller keys in the bucket, #A==q = equal keys in it.  The reference's
//    zeros are not in the table: callers add them for q above zero (their zA argument counts ALL reference zeros then).
// bytes of the run-end table / bucket table region
__host__ __device__ static inline size_t ovo_runend_bytes(int ref_cap, bool buckets) {
    size_t b = ((size_t)ref_cap * 2 + 15) & ~(size_t)15;
    const size_t t = (size_t)2 << 13; // OVO_REF_BUCKETS_LG
    return buckets && b < t ? t : b;
}
template <typename KeyT> struct RefBk {
    bool on;
    const u16 *tab;   // [n_buckets] (aliases the run-end table: one form per gene)
    KeyT kmin, last;  // last = n_buckets - 1
    int shift;
    u32 zeros;        // reference cells stored as explicit zeros (dense layout): what a look-up of zero itself finds
};
#define OVO_REF_BUCKETS_LG 13
#define OVO_REF_MAX_BUCKET 48 // bucket form only while no bucket holds more reference keys than this

template <typename KeyT> __device__ __forceinline__ u32 refbk_bucket(const RefBk<KeyT> &bk, KeyT q) {
    const KeyT d = q > bk.kmin ? (KeyT)((KeyT)(q - bk.kmin) >> bk.shift) : (KeyT)0;
    return (u32)(d < bk.last ? d : bk.last);
}
// lb = #A < q, a = #A == q
// PAD: 4 slots of the largest key follow the table's keys (the walk may read past the last bucket); without it the walk
// checks every slot against the bucket's end (the reference run sits inside a larger buffer: k_csc_gene).
template <typename KeyT, bool RUNEND, bool PAD = true>
__device__ __forceinline__ void ref_find(const KeyT *A, const u16 *runend, u32 nA, u32 topA, const RefBk<KeyT> &bk, KeyT q, u32 &lb, u32 &a) {
    if (RUNEND && bk.on) { // uniform
        const u32 b = refbk_bucket(bk, q);
        const u32 lo = b ? bk.tab[b - 1] : 0u, hi = bk.tab[b];
        u32 less = 0, eq = 0;
        for (u32 j = lo; j < hi; j += 4) { // keys past the bucket's end are larger than q (later buckets / the MAXK pad)
            const KeyT a0 = A[j], a1 = A[j + 1], a2 = A[j + 2], a3 = A[j + 3];
            if (PAD) {
                less += (a0 < q ? 1u : 0u) + (a1 < q ? 1u : 0u) + (a2 < q ? 1u : 0u) + (a3 < q ? 1u : 0u);
            } else { // later buckets still hold larger keys; only slots past the table's last key (j >= nA) are foreign
                less += (a0 < q ? 1u : 0u) + ((j + 1 < nA && a1 < q) ? 1u : 0u) + ((j + 2 < nA && a2 < q) ? 1u : 0u) + ((j + 3 < nA && a3 < q) ? 1u : 0u);
            }
            if (PAD) {
                eq += (a0 == q ? 1u : 0u) + (a1 == q ? 1u : 0u) + (a2 == q ? 1u : 0u) + (a3 == q ? 1u : 0u);
            } else {
                eq += (a0 == q ? 1u : 0u) + ((j + 1 < nA && a1 == q) ? 1u : 0u) + ((j + 2 < nA && a2 == q) ? 1u : 0u) + ((j + 3 < nA && a3 == q) ? 1u : 0u);
            }
        }
        // a query equal to the largest key (INT_MAX, an all-ones NaN) also matches the MAXK pad slots past the last bucket:
        // there every key of the bucket that is not smaller is equal
        if (PAD && q == KeyInfo<KeyT>::MAXK) eq = (hi - lo) - less;
        lb = lo + less;
        a = q == KeyInfo<KeyT>::ZEROK ? bk.zeros : eq;
    } else {
        lb = lower_bound_pow2(A, nA, topA, q);
        a = 0;
        if (lb < nA && A[lb] == q) {
            if (RUNEND) a = (u32)runend[lb] - lb;
            else a = upper_bound_pow2(A, nA, topA, q) - lb;
        }
    }
}

// One group's values (nB <= 64*K keys) handled by one wavefront.  Returns PER-LANE partial sums
// (S2, tie, value sum); the caller folds 64 groups' partials with a transpose-reduce.
__device__ __forceinline__ u32 bloom_hash1(u32 k) { return (k * 0x9E3779B1u) >> 19; } // 13 bits
__device__ __forceinline__ u32 bloom_hash2(u32 k) { return ((k ^ (k >> 15)) * 0x85EBCA6Bu) >> 19; }
__device__ __forceinline__ u32 bloom_fold(u32 k) { return k; }
__device__ __forceinline__ u32 bloom_fold(u64 k) { return (u32)(k ^ (k >> 32)); }

// sk / sb (256 entries each, per wavefront) must be all-zero on entry and are all-zero again on exit: they serve
// first as two 8192-bit Bloom tables, then as compaction scratch.
template <typename KeyT, int K, int KIN, bool RUNEND, int KB = K>
__device__ __forceinline__ void ovo_wave_group(const KeyT (&vin)[KIN], int nB, const KeyT *A,
                                               const u16 *runend, u32 nA, u32 topA, u32 zA, u32 lbZ, u32 aZ, KeyT *sk, u32 *sb,
                                               int lane, int dt, int is_log1p, u64 &S2out, u64 &tieout,
                                               double &sumout, const RefBk<KeyT> &bk) {
    constexpr KeyT MAXK = KeyInfo<KeyT>::MAXK;
    constexpr KeyT ZEROK = KeyInfo<KeyT>::ZEROK;
    KeyT v[K];
    double s = 0.0;
#pragma unroll
    for (int r = 0; r < K; ++r) { // vin[r] holds element r*64 + lane of the group (MAXK beyond nB)
        v[r] = vin[r];
        if (r < KB && r * 64 + lane < nB) s += is_log1p ? key_to_expm1(v[r], dt) : key_to_double(v[r], dt); // nB <= 64 * KB
    }

    // ---- distinct-values fast path (normalised / continuous data) ----
    // If the group's non-zero keys are pairwise distinct, every non-zero key is a run of length 1 and the zeros are
    // one run: no sort is needed, each key is only looked up in the reference.  Distinctness is PROVEN by a
    // two-table Bloom test in LDS (a key whose bit was still clear in either table cannot equal an earlier key);
    // any possible duplicate sends the whole group to the exact sort path below.
    if constexpr (K <= 4) {
        u32 *bm1 = sb, *bm2 = (u32 *)sk;
        const u64 lt_mask = (lane == 0) ? 0ull : (~0ull >> (64 - lane));
        bool maybe_dup = false;
        u32 zc = 0;
#pragma unroll
        for (int r = 0; r < KB; ++r) {
            const bool valid = r * 64 + lane < nB;
            const bool nz = valid && v[r] != ZEROK;
            zc += (u32)__popcll(__ballot(valid && v[r] == ZEROK));
            if (nz) {
                const u32 f = bloom_fold(v[r]);
                const u32 h1 = bloom_hash1(f), h2 = bloom_hash2(f);
                const u32 o1 = atomicOr(&bm1[h1 >> 5], 1u << (h1 & 31));
                const u32 o2 = atomicOr(&bm2[h2 >> 5], 1u << (h2 & 31));
                maybe_dup |= ((o1 >> (h1 & 31)) & (o2 >> (h2 & 31)) & 1u) != 0;
            }
        }
        const bool any_dup = __ballot(maybe_dup) != 0ull;
#pragma unroll
        for (int r = 0; r < KB; ++r) { // wipe the words this lane touched (LDS is in order within a wavefront)
            if (r * 64 + lane < nB && v[r] != ZEROK) {
                const u32 f = bloom_fold(v[r]);
                bm1[bloom_hash1(f) >> 5] = 0u;
                bm2[bloom_hash2(f) >> 5] = 0u;
            }
        }
        wave_lds_fence();
        if (!any_dup) {
            // compact the non-zero keys through sk, then one lookup per key
            int base = 0;
#pragma unroll
            for (int r = 0; r < KB; ++r) {
                const bool nz = (r * 64 + lane < nB) && v[r] != ZEROK;
                const u64 m = __ballot(nz);
                if (nz) sk[base + __popcll(m & lt_mask)] = v[r];
                base += __popcll(m);
            }
            wave_lds_fence();
            u64 S2 = 0, TT = 0;
            for (int j = 0; j * 64 < base; ++j) {
                const int slot = j * 64 + lane;
                if (slot < base) {
                    const KeyT q = sk[slot];
                    u32 lb, a32;
                    ref_find<KeyT, RUNEND>(A, runend, nA, topA, bk, q, lb, a32);
                    const u64 a = a32;
                    const u64 lt = (u64)lb + ((q > ZEROK) ? (u64)zA : 0ull);
                    S2 += 2ull * lt + a;
                    TT += a * (a + 1ull); // run of length 1: tB (3 a (a+1) + 1 - 1) / 3
                    sk[slot] = (KeyT)0;
                }
            }
            wave_lds_fence();
            u64 tie = 3ull * TT;
            if (lane == 0 && zc) { // the group's explicit zeros: one run of length zc against aZ reference zeros
                const u64 b = zc, a = aZ;
                S2 += b * (2ull * lbZ + a);
                tie += b * (3ull * a * (a + b) + b * b - 1ull);
            }
            S2out = S2;
            tieout = tie;
            sumout = s;
            return;
        }
    }
    wave_bitonic_sort<KeyT, K>(v, lane);

    // Runs of equal keys in sorted order i = lane*K + r.  The LAST element of a run (its tail) represents
    // it: the run length is i - (position of the run's head) + 1, and the head position of every element is
    // an inclusive prefix-max of head positions -- prefix scans are native DPP (row_shr / row_bcast).
    const KeyT prev_last = wave_shr1(v[K - 1], (KeyT)0);
    const KeyT next_first = wave_shl1(v[0], MAXK);
    int lh = -1;          // position of the last head seen in this lane
    int shead[K];         // head position of element r if it lies in this lane, else -1
    bool tail[K];
    int tc = 0;
#pragma unroll
    for (int r = 0; r < K; ++r) {
        const int idx = lane * K + r;
        const KeyT prev = (r == 0) ? prev_last : v[r - 1];
        const KeyT next = (r == K - 1) ? next_first : v[r + 1];
        const bool valid = idx < nB;
        if (valid && (idx == 0 || v[r] != prev)) lh = idx;
        shead[r] = lh;
        tail[r] = valid && (idx == nB - 1 || v[r] != next);
        tc += tail[r] ? 1 : 0;
    }
    const int incl_h = wave_incl_scan_max(lh);
    const int carry_h = (int)wave_shr1((u32)incl_h, 0xFFFFFFFFu); // head position inherited from earlier lanes
    const int incl_t = wave_incl_scan_add(tc);
    const int H = __builtin_amdgcn_readlane(incl_t, 63);
    const int excl_t = incl_t - tc;

    u64 S2 = 0, tie = 0;
    for (int base = 0; base < H; base += 256) {
        int hr = excl_t - base;
#pragma unroll
        for (int r = 0; r < K; ++r) {
            if (tail[r]) {
                if (hr >= 0 && hr < 256) {
                    const int idx = lane * K + r;
                    const int hp = shead[r] >= 0 ? shead[r] : carry_h;
                    sk[hr] = v[r];
                    sb[hr] = (u32)(idx - hp + 1);
                }
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
                u32 lb, a32;
                ref_find<KeyT, RUNEND>(A, runend, nA, topA, bk, q, lb, a32);
                const u64 a = a32;
                u64 lt = (u64)lb + ((q > ZEROK) ? (u64)zA : 0ull);  // implicit zeros of A rank below positives
                S2 += b * (2ull * lt + a);
                tie += b * (3ull * a * (a + b) + b * b - 1ull);
                sk[slot] = (KeyT)0; // leave the scratch zeroed (it doubles as the Bloom tables)
                sb[slot] = 0u;
            }
        }
        wave_lds_fence();
    }
    S2out = S2;
    tieout = tie;
    sumout = s;
}

// 64 SMALL groups at once, one per lane (sparse layouts: a (gene, group) run holds a handful of non-zeros, so a
// wavefront per group would idle most lanes).  Each lane walks its own run; the number of earlier equal values
// `o` is counted against the lane's own history in registers, and per element
//     S2 += 2 #A<q + #A==q,      TT += t (t+1),  t = #A==q + o        (3 sum t(t+1) is the group's tie term)
// -- the same integers as the run-based form of ovo_wave_group.  Returns per-lane (= per-group) totals.
template <typename KeyT, int SMALL, bool RUNEND, bool PAD = true>
__device__ __forceinline__ void ovo_lane_groups(const KeyT *__restrict__ Xs, long long bstart, int n, int nmax, const KeyT *A,
                                                const u16 *runend, u32 nA, u32 topA, u32 zA, int dt, int is_log1p,
                                                u64 &S2out, u64 &TTout, double &sumout, const RefBk<KeyT> &bk) {
    constexpr KeyT MAXK = KeyInfo<KeyT>::MAXK;
    constexpr KeyT ZEROK = KeyInfo<KeyT>::ZEROK;
    KeyT e[SMALL];
#pragma unroll
    for (int j = 0; j < SMALL; ++j) e[j] = (j < n) ? Xs[bstart + j] : MAXK; // all loads in flight together
    u64 S2 = 0, TT = 0;
    double s = 0.0;
#pragma unroll
    for (int j = 0; j < SMALL; ++j) {
        if (j < nmax) { // wave-uniform
            const KeyT q = e[j];
            const bool valid = j < n;
            u32 o = 0;
#pragma unroll
            for (int i = 0; i < j; ++i) o += (e[i] == q) ? 1u : 0u;
            u32 lb, a;
            ref_find<KeyT, RUNEND, PAD>(A, runend, nA, topA, bk, q, lb, a);
            if (valid) {
                const u64 lt = (u64)lb + ((q > ZEROK) ? (u64)zA : 0ull);
                const u64 t = (u64)a + o;
                S2 += 2ull * lt + a;
                TT += t * (t + 1ull);
                s += is_log1p ? key_to_expm1(q, dt) : key_to_double(q, dt);
            }
        }
    }
    S2out = S2;
    TTout = TT;
    sumout = s;
}

// Same as ovo_lane_groups for longer runs (up to a few hundred keys): the lane re-reads its own run from memory
// (LDS in k_csc_gene) instead of keeping a register history.  O(n^2/2) reads per lane: meant for the rare block
// whose longest run exceeds the register form's limit.
template <typename KeyT, bool RUNEND, bool PAD = true>
__device__ __forceinline__ void ovo_lane_groups_mem(const KeyT *Xs, long long bstart, int n, int nmax, const KeyT *A,
                                                    const u16 *runend, u32 nA, u32 topA, u32 zA, int dt, int is_log1p,
                                                    u64 &S2out, u64 &TTout, double &sumout, const RefBk<KeyT> &bk) {
    constexpr KeyT MAXK = KeyInfo<KeyT>::MAXK;
    constexpr KeyT ZEROK = KeyInfo<KeyT>::ZEROK;
    u64 S2 = 0, TT = 0;
    double s = 0.0;
    for (int j = 0; j < nmax; ++j) {
        const bool valid = j < n;
        const KeyT q = valid ? Xs[bstart + j] : MAXK;
        u32 o = 0;
        for (int i = 0; i < j; ++i) o += (valid && Xs[bstart + i] == q) ? 1u : 0u;
        u32 lb, a;
        ref_find<KeyT, RUNEND, PAD>(A, runend, nA, topA, bk, q, lb, a);
        if (valid) {
            const u64 lt = (u64)lb + ((q > ZEROK) ? (u64)zA : 0ull);
            const u64 t = (u64)a + o;
            S2 += 2ull * lt + a;
            TT += t * (t + 1ull);
            s += is_log1p ? key_to_expm1(q, dt) : key_to_double(q, dt);
        }
    }
    S2out = S2;
    TTout = TT;
    sumout = s;
}

// The reference column -> value buckets in LDS (no sort): A = its non-zero keys in bucket order (+ 4 pad slots), the
// run-end region = 16-bit bucket ends.  Returns false (uniform) when a bucket would hold more than OVO_REF_MAX_BUCKET
// keys: the caller sorts the column instead.  Outputs through LDS: s_kr[0] = kmin, s_cnt[0] = explicit zeros, [1] =
// negatives, [3] = shift, *s_TA = tie term of the column, *s_refsum = its value sum.  Kept out of line: it runs once per
// gene and its registers must not weigh on the group loop.
template <typename KeyT, int NT>
__device__ __noinline__ bool ovo_build_ref_buckets(const KeyT *__restrict__ src, u32 nA, KeyT *A, u16 *runend, u32 *scan_tmp, u64 *s_red,
                                                   double *s_redd, u64 *s_TA, double *s_refsum, KeyT *s_kr, u32 *s_cnt, int dt, int is_log1p) {
    constexpr int NBK = 1 << OVO_REF_BUCKETS_LG, NW = NT / 64;
    constexpr KeyT ZEROK = KeyInfo<KeyT>::ZEROK;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    u32 *tab32 = (u32 *)runend;
    u16 *tab16 = (u16 *)runend;
    RefBk<KeyT> bk;
    bk.on = true; bk.tab = runend; bk.last = (KeyT)(NBK - 1); bk.zeros = 0u;
    for (int b = tid; b < NBK / 2; b += NT) tab32[b] = 0u;
    if (tid == 0) { s_kr[0] = KeyInfo<KeyT>::MAXK; s_kr[1] = (KeyT)0; s_cnt[0] = 0u; s_cnt[1] = 0u; s_cnt[2] = 0u; }
    __syncthreads();
    {
        KeyT tmin = KeyInfo<KeyT>::MAXK, tmax = (KeyT)0;
        u32 nz0 = 0, ng = 0;
        double rs = 0.0;
        for (u32 i = tid; i < nA; i += NT) {
            const KeyT k = src[i];
            rs += is_log1p ? key_to_expm1(k, dt) : key_to_double(k, dt);
            if (k != ZEROK) { tmin = k < tmin ? k : tmin; tmax = k > tmax ? k : tmax; ng += k < ZEROK ? 1u : 0u; }
            else ++nz0;
        }
        rs = wave_sum(rs);
        nz0 = (u32)wave_sum((int)nz0);
        ng = (u32)wave_sum((int)ng);
#pragma unroll
        for (int d = 32; d > 0; d >>= 1) {
            const KeyT o1 = __shfl_xor(tmin, d), o2 = __shfl_xor(tmax, d);
            tmin = o1 < tmin ? o1 : tmin;
            tmax = o2 > tmax ? o2 : tmax;
        }
        if (lane == 0) {
            s_redd[wave] = rs;
            atomicMin(&s_kr[0], tmin);
            atomicMax(&s_kr[1], tmax);
            if (nz0) atomicAdd(&s_cnt[0], nz0);
            if (ng) atomicAdd(&s_cnt[1], ng);
        }
    }
    __syncthreads();
    const u32 aZ = s_cnt[0], nAs = nA - aZ;
    const bool have = s_kr[1] >= s_kr[0];
    const KeyT kmax = have ? s_kr[1] : (KeyT)0;
    bk.kmin = have ? s_kr[0] : (KeyT)0;
    bk.shift = kmax == bk.kmin ? 0 : max(0, (int)(sizeof(KeyT) * 8) - (int)(sizeof(KeyT) == 4 ? __clz((u32)(kmax - bk.kmin)) : __clzll((long long)(u64)(kmax - bk.kmin))) - OVO_REF_BUCKETS_LG);
    __syncthreads();
    if (tid == 0) { s_kr[0] = bk.kmin; s_cnt[3] = (u32)bk.shift; }
    for (u32 i = tid; i < nA; i += NT) {
        const KeyT k = src[i];
        if (k != ZEROK) {
            const u32 b = refbk_bucket(bk, k);
            atomicAdd(&tab32[b >> 1], (b & 1u) ? 0x10000u : 1u);
        }
    }
    __syncthreads();
    u32 mxb = 0;
    for (int b = tid; b < NBK; b += NT) mxb = max(mxb, (u32)tab16[b]);
    mxb = (u32)wave_incl_scan_max((int)mxb);
    if (lane == 63) atomicMax(&s_cnt[2], mxb);
    __syncthreads();
    if (s_cnt[2] > (u32)OVO_REF_MAX_BUCKET) return false; // uniform
    { // exclusive scan of the 16-bit counters (scratch: the per-wave Bloom words, left zeroed)
        const int per = NBK / NT, b0 = tid * per;
        u32 sm = 0;
        for (int i = 0; i < per; ++i) sm += tab16[b0 + i];
        scan_tmp[tid] = sm;
        __syncthreads();
        for (int d = 1; d < NT; d <<= 1) {
            const u32 v = (tid >= d) ? scan_tmp[tid - d] : 0u;
            __syncthreads();
            scan_tmp[tid] += v;
            __syncthreads();
        }
        u32 run = scan_tmp[tid] - sm;
        __syncthreads();
        scan_tmp[tid] = 0u;
        for (int i = 0; i < per; ++i) { const u32 cnt = tab16[b0 + i]; tab16[b0 + i] = (u16)run; run += cnt; }
        __syncthreads();
    }
    for (u32 i = tid; i < nA; i += NT) {
        const KeyT k = src[i];
        if (k != ZEROK) {
            const u32 b = refbk_bucket(bk, k);
            const u32 old = atomicAdd(&tab32[b >> 1], (b & 1u) ? 0x10000u : 1u);
            A[(b & 1u) ? (old >> 16) : (old & 0xFFFFu)] = k;
        }
    }
    if (tid < 4) A[nAs + tid] = KeyInfo<KeyT>::MAXK; // the bucket walk reads up to 3 keys past a bucket's end
    __syncthreads();
    u64 ta = 0; // ties inside the column: sum over its non-zero keys of (run length^2 - 1), plus its zero run
    for (u32 i = tid; i < nAs; i += NT) {
        u32 lb, a;
        ref_find<KeyT, true>(A, runend, nAs, 0u, bk, A[i], lb, a);
        ta += (u64)a * a - 1ull;
    }
    ta = wave_sum(ta);
    if (lane == 0) s_red[wave] = ta;
    __syncthreads();
    if (tid == 0) {
        u64 t = (u64)aZ * aZ * aZ - (u64)aZ;
        double d = 0.0;
        for (int w = 0; w < NW; ++w) { t += s_red[w]; d += s_redd[w]; }
        *s_TA = t;
        *s_refsum = d;
    }
    __syncthreads();
    return true;
}

// LG: also compile the lane-per-group form for blocks of 64 short runs (sparse layouts); it needs more registers
// (the lane's history), so the dense instantiations leave it out and keep 4 waves per SIMD.
template <typename KeyT, int KMAX, bool RUNEND, int NT, bool LG>
__global__ __launch_bounds__(NT, ((KMAX <= 4 && !LG) ? 1024 : 512) / NT * (NT / 256)) void k_ovo_rank(OvoParams P, const u32 *__restrict__ gene_flags) {
    extern __shared__ __align__(16) unsigned char smem[];
    constexpr int NW = NT / 64;
    constexpr KeyT ZEROK = KeyInfo<KeyT>::ZEROK;
    // LDS carve (all offsets multiples of 16 B): A | runend | per-wave compaction scratch | reduction words
    KeyT *A = (KeyT *)smem;
    size_t off = (((size_t)P.ref_cap + 4) * sizeof(KeyT) + 15) & ~(size_t)15;
    u16 *runend = (u16 *)(smem + off);
    if (RUNEND) off += ovo_runend_bytes(P.ref_cap, P.ref_buckets != 0);
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
    // the per-wave scratch doubles as Bloom tables: it starts zeroed and every user leaves it zeroed
    for (int i = tid; i < NW * 256; i += NT) { sk_all[i] = (KeyT)0; sb_all[i] = 0u; }
    __syncthreads();

    for (int gene = blockIdx.x; gene < P.n_genes; gene += gridDim.x) {
        if (gene_flags && gene_flags[gene] == 0) continue; // count-valued gene: handled by k_ovo_counts
        if (P.only && (P.only[gene] == 0u || P.only[gene] > 255u)) continue; // (a word above 255: the packed PARTS kernel's gene, not laid out for this kernel)
        // ---- reference column -> LDS, sorted ----
        long long rstart;
        u32 nA;
        const u32 *sp = nullptr;
        if (P.seg_ptr) {
            sp = P.seg_ptr + (size_t)gene * (G + 1);
            rstart = sp[ref];
            nA = sp[ref + 1] - sp[ref];
        } else {
            rstart = (long long)gene * P.gene_stride + (P.nnz ? (long long)P.gofs[(size_t)gene * G + ref] : (long long)P.pos_ptr[ref]);
            nA = P.nnz ? (u32)P.nnz[(size_t)gene * G + ref] : (u32)n_ref;
        }
        const u32 zA_impl = (u32)n_ref - nA; // implicit zeros of the reference (sparse layout)
        RefBk<KeyT> bk;
        bk.on = false;
        bk.tab = runend;
        bk.kmin = (KeyT)0; bk.last = (KeyT)((1u << OVO_REF_BUCKETS_LG) - 1u); bk.shift = 0; bk.zeros = 0u;
        u32 nAs = nA;          // keys held in A: all of the run (sorted form) / its non-zeros (bucket form)
        u32 topA = 0, nnegA = 0, lbZ = 0, aZ = 0;
        u64 T_A = 0;
        double refsum = 0.0;
        if (RUNEND && P.ref_buckets) { // uniform
            KeyT *s_kr = (KeyT *)(s_refsum + 1);  // [2]
            u32 *s_cnt = (u32 *)(s_kr + 2);       // [0] zeros  [1] negatives  [2] largest bucket  [3] shift
            if (ovo_build_ref_buckets<KeyT, NT>(Xs + rstart, nA, A, runend, sb_all, s_red, s_redd, s_TA, s_refsum, s_kr, s_cnt, P.dt, P.is_log1p)) {
                bk.on = true;
                aZ = s_cnt[0];
                bk.zeros = aZ;
                nnegA = lbZ = s_cnt[1];
                nAs = nA - aZ;
                bk.kmin = s_kr[0];
                bk.shift = (int)s_cnt[3];
                T_A = *s_TA;
                refsum = *s_refsum;
            }
        }
        if (!bk.on) {
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
        topA = top_pow2(nA);
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
        T_A = *s_TA;
        refsum = *s_refsum;
        nnegA = (P.seg_ptr || P.nnz) ? lower_bound_pow2(A, nA, topA, ZEROK) : 0u;
        // reference cells below zero / equal to zero (explicit zeros of the dense layout), for a group's zero run
        lbZ = lower_bound_pow2(A, nA, topA, ZEROK);
        aZ = upper_bound_pow2(A, nA, topA, ZEROK) - lbZ;
        }
        // what a look-up adds for q above zero: the implicit zeros, plus (bucket form) the explicit ones, which are not
        // in the table
        const u32 zA = bk.on ? zA_impl + aZ : zA_impl;

        // ---- every other group: one wavefront each, 64 groups per output block ----
        KeyT *sk = sk_all + wave * 256;
        u32 *sb = sb_all + wave * 256;
        for (int g0 = wave * 64; g0 < G; g0 += NW * 64) {
            if (LG && sp) { // sparse layout: are all 64 runs of this block small enough for the lane-per-group form?
                constexpr int SMALL = sizeof(KeyT) == 4 ? 32 : 16;
                const int gl = g0 + lane;
                const bool has = gl < G && gl != ref;
                const long long bs = has ? (long long)sp[gl] : 0;
                const int n = has ? (int)(sp[gl + 1] - sp[gl]) : 0;
                const int nmax = __builtin_amdgcn_readlane(wave_incl_scan_max(n), 63);
                if (nmax <= SMALL) {
                    u64 S2, TT;
                    double sum;
                    ovo_lane_groups<KeyT, SMALL, RUNEND>(Xs, bs, n, nmax, A, runend, nA, topA, zA, P.dt, P.is_log1p, S2, TT, sum, bk);
                    if (gl < G) {
                        size_t o = (size_t)gene * G + gl;
                        if (gl == ref) {
                            P.out_2u[o] = -2;
                            P.out_tie[o] = 0;
                            if (P.out_sum) P.out_sum[o] = refsum;
                        } else {
                            const long long n_g = P.counts[gl];
                            const u64 zB = (u64)(n_g - n);
                            S2 += zB * (2ull * nnegA + zA_impl);
                            const u64 t0 = (u64)zA_impl + zB;
                            P.out_2u[o] = 2ll * (long long)n_ref * n_g - (long long)S2;
                            P.out_tie[o] = T_A + 3ull * TT + (t0 * t0 * t0 - t0);
                            if (P.out_sum) P.out_sum[o] = sum;
                        }
                    }
                    continue;
                }
            }
            TrReduce<u64> rS2;
            u64 tie_out = 0; // lane j: tie term of group g0 + j
            TrReduce<double> rSum;
            // The next group's keys are fetched into registers while the current group is processed: one
            // wavefront walks ~G/NW groups back to back and would otherwise expose a full HBM round trip each.
            KeyT nxt[KMAX];
            int nB_n = 0, zB_n = 0;
            auto fetch = [&](int g) {
                nB_n = 0; zB_n = 0;
#pragma unroll
                for (int r = 0; r < KMAX; ++r) nxt[r] = KeyInfo<KeyT>::MAXK;
                if (g < G && g != ref) {
                    const int n_g = P.counts[g];
                    long long bstart;
                    if (sp) { bstart = sp[g]; nB_n = (int)(sp[g + 1] - sp[g]); }
                    else if (P.nnz) { bstart = (long long)gene * P.gene_stride + P.gofs[(size_t)gene * G + g]; nB_n = (int)P.nnz[(size_t)gene * G + g]; }
                    else { bstart = (long long)gene * P.gene_stride + P.pos_ptr[g]; nB_n = n_g; }
                    zB_n = n_g - nB_n;
                    const KeyT *seg = Xs + bstart;
#pragma unroll
                    for (int r = 0; r < KMAX; ++r) {
                        const int idx = r * 64 + lane;
                        if (idx < nB_n) nxt[r] = seg[idx];
                    }
                }
            };
            fetch(g0);
            for (int j = 0; j < 64; ++j) { // always 64 pushes so that the transpose-reduce completes
                const int g = g0 + j;
                KeyT cur[KMAX];
#pragma unroll
                for (int r = 0; r < KMAX; ++r) cur[r] = nxt[r];
                const int nB = nB_n;
                const u32 zB = (u32)zB_n;
                fetch(g + 1 < g0 + 64 ? g + 1 : G);
                u64 S2 = 0, tie = 0;
                double sum = 0.0;
                if (g < G && g != ref) {
                    if (nB <= 64) ovo_wave_group<KeyT, 1, KMAX, RUNEND>(cur, nB, A, runend, nA, topA, zA, lbZ, aZ, sk, sb, lane, P.dt, P.is_log1p, S2, tie, sum, bk);
                    else if (nB <= 128) ovo_wave_group<KeyT, 2, KMAX, RUNEND>(cur, nB, A, runend, nA, topA, zA, lbZ, aZ, sk, sb, lane, P.dt, P.is_log1p, S2, tie, sum, bk);
                    else if (nB <= 192) ovo_wave_group<KeyT, 4, KMAX, RUNEND, 3>(cur, nB, A, runend, nA, topA, zA, lbZ, aZ, sk, sb, lane, P.dt, P.is_log1p, S2, tie, sum, bk); // 3 rounds of the distinct-values path
                    else if (nB <= 256) ovo_wave_group<KeyT, 4, KMAX, RUNEND>(cur, nB, A, runend, nA, topA, zA, lbZ, aZ, sk, sb, lane, P.dt, P.is_log1p, S2, tie, sum, bk);
                    else if (KMAX >= 8 && nB <= 512) ovo_wave_group<KeyT, (KMAX >= 8 ? 8 : 4), KMAX, RUNEND>(cur, nB, A, runend, nA, topA, zA, lbZ, aZ, sk, sb, lane, P.dt, P.is_log1p, S2, tie, sum, bk);
                    else if (KMAX >= 16 && nB <= 1024) ovo_wave_group<KeyT, (KMAX >= 16 ? 16 : 4), KMAX, RUNEND>(cur, nB, A, runend, nA, topA, zA, lbZ, aZ, sk, sb, lane, P.dt, P.is_log1p, S2, tie, sum, bk);
                    if (lane == 0) {
                        // implicit zeros of B (sparse layout): each ranks above A's negatives and ties with A's zeros
                        S2 += (u64)zB * (2ull * nnegA + zA_impl);
                        const u64 t0 = (u64)zA_impl + zB;
                        tie += T_A + (t0 * t0 * t0 - t0);
                    }
                }
                rS2.push(S2, j, lane);
                { // tie partials are zero in most lanes (continuous data: all but lane 0's zero-run / T_A terms): reduce only
                  // what is there instead of carrying a third 64-bit transpose-reduce
                    const u64 m = __ballot(tie != 0ull);
                    u64 tot = 0;
                    if (m) {
                        if ((m & (m - 1ull)) == 0ull) {
                            const int src = __ffsll((long long)m) - 1;
                            tot = ((u64)(u32)__builtin_amdgcn_readlane((int)(u32)(tie >> 32), src) << 32) | (u64)(u32)__builtin_amdgcn_readlane((int)(u32)tie, src);
                        } else tot = wave_sum(tie);
                    }
                    if (lane == j) tie_out = tot;
                }
                rSum.push(sum, j, lane);
            }
            const int g = g0 + lane; // lane j now holds the totals of group g0 + j
            if (g < G) {
                size_t o = (size_t)gene * G + g;
                if (g == ref) {
                    P.out_2u[o] = -2;
                    P.out_tie[o] = 0;
                    if (P.out_sum) P.out_sum[o] = refsum;
                } else {
                    P.out_2u[o] = 2ll * (long long)n_ref * (long long)P.counts[g] - (long long)rS2.result;
                    P.out_tie[o] = tie_out;
                    if (P.out_sum) P.out_sum[o] = rSum.result;
                }
            }
        }
        __syncthreads();
    }
}
