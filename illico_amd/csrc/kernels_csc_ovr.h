// CSC one-versus-rest for ANY values in ONE kernel per gene, without a sort.  A rank needs, for every stored entry, the
// tie block [s, e) its value occupies in the sorted column.  The gene's stored non-zeros are dealt into LDS buckets by
// value ((key - kmin) >> shift, 8192 or 16384 buckets with 16-bit offsets: a counting pass, a scan, a scattering pass), and every entry then
// counts the smaller and the equal keys INSIDE ITS OWN BUCKET (a few keys): s = bucket start + #smaller, e = s + #equal,
// 2 * avg_rank = s + e + 1 goes to its group's LDS accumulator and e - s (the tie block length t) gives the tie term as
// sum over entries of (t^2 - 1) = sum over blocks of (t^3 - t).  Nothing but the CSC arrays is read and nothing but the
// [gene][G] statistics is written; the general route it replaces regroups the entries in HBM (k_csc_regroup), sorts
// (key, group) pairs with a segmented radix sort in HBM and sweeps them (k_ovr_gene).
//
// Columns whose values crowd into few buckets (heavy ties, an outlier stretching the range) take the second form in the
// same kernel: the keys are sorted in LDS as bare keys (block_sort_hybrid) and every entry looks its value up with two
// binary searches.
//
// Device counterpart of sparse_ovr_mwu_kernel + its CSC entry (illico/ovr/sparse_ovr.py:23-97, :100-155) and
// _accumulate_group_ranksums_from_argsort (utils/ranking.py:7-49) for one gene at a time; the zeros stay implicit
// (sparse_ovr.py:70-83): n0 = N - nnz cells tie at rank n_neg + (n0 + 1) / 2 and shift every positive entry by n0.
//
// A gene with more stored entries than the LDS key buffer sets fallback[gene]; the host sends those genes through the
// general route.
#pragma once
#include "common.h"
#include "kernels_sparse.h"

#define CSCO_NT 1024
#define CSCO_K 16        // keys per lane of the register sort phases (sorted form): 1024-key chunks
#define CSCO_MAX_BUCKET 192 // bucket form only while no bucket holds more keys than this ...
#define CSCO_MAX_AVG 64     // ... and the average entry shares its bucket with at most this many
#define CSCO_CNT_SHIFT 40   // acc word: doubled rank sum below, stored non-zeros of the group above

struct CscOvrParams {
    const void *data, *indices, *indptr; // CSC arrays (device); stored entry k lives at data[k - kshift], indices[k - kshift]
    long long kshift;
    long long col0;                      // first gene of the batch (contiguous batches)
    const int *gene_cols;                // or: the batch's genes as a column list (absolute indices); nullptr = contiguous
    int nb;
    const int *codes;                    // [n_cells] group code per cell; nullptr: `indices` already holds group codes
    const int *counts;                   // [G]
    int G, dt, is_log1p;
    long long n_cells;
    int key_cap;                         // LDS key slots
    int lg_buckets;                      // log2 of the bucket count
    int force_sorted;                    // 1: every gene takes the sorted form (tests / A-B)
    u32 *fallback;                       // [nb] set to 1 for genes this kernel cannot take
    long long *out_2u;                   // [nb][G] 2 U (U of "the rest", dense_ovr.py:57-61)
    u64 *out_tie;                        // [nb][G] sum (t^3 - t), the same for every group of a gene
    double *out_sum;                     // [nb][G] per-group value sums
};

__host__ __device__ static inline size_t csco_fixed_lds_bytes(int G, int lg_buckets) {
    // acc (value sums in pass 1, packed rank sums / counts afterwards) | bucket table | reductions: a multiple of 16
    return (size_t)((G + 1) & ~1) * 8 + ((size_t)2 << lg_buckets) + 256;
}
static inline int csco_key_cap(int G, int lg_buckets, size_t key_size, size_t lds_max) {
    const size_t fixed = csco_fixed_lds_bytes(G, lg_buckets);
    if (fixed + (size_t)CSCO_NT * 4 + 64 > lds_max) return 0; // the scan borrows NT words of the key buffer
    // bucket offsets are 16-bit; 4 slots stay free behind the keys (the bucket walk reads 4 keys at a time)
    return (int)std::min<size_t>((lds_max - fixed) / key_size - 4, 65535 - 4);
}

template <typename KeyT> __device__ __forceinline__ KeyT wave_min_key(KeyT x) {
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) { KeyT o = __shfl_xor(x, d); x = o < x ? o : x; }
    return x;
}
template <typename KeyT> __device__ __forceinline__ KeyT wave_max_key(KeyT x) {
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) { KeyT o = __shfl_xor(x, d); x = o > x ? o : x; }
    return x;
}
__device__ __forceinline__ int key_bits(u32 r) { return r ? 32 - __clz(r) : 0; }
__device__ __forceinline__ int key_bits(u64 r) { return r ? 64 - __clzll((long long)r) : 0; }

// exclusive scan of the 16-bit counters arr[0..n) in place (n a multiple of NT; totals below 2^16).  tmp: [NT] words.
template <int NT> __device__ __forceinline__ void block_excl_scan_u16(u16 *arr, int n, u32 *tmp, int tid) {
    const int per = n / NT, b = tid * per;
    u32 s = 0;
    for (int i = 0; i < per; ++i) s += arr[b + i];
    tmp[tid] = s;
    __syncthreads();
    for (int d = 1; d < NT; d <<= 1) {
        const u32 v = (tid >= d) ? tmp[tid - d] : 0u;
        __syncthreads();
        tmp[tid] += v;
        __syncthreads();
    }
    u32 run = tmp[tid] - s;
    for (int i = 0; i < per; ++i) { const u32 cnt = arr[b + i]; arr[b + i] = (u16)run; run += cnt; }
    __syncthreads();
}

template <typename InT, typename IdxT, typename KeyT>
__global__ __launch_bounds__(CSCO_NT) void k_csc_ovr_gene(CscOvrParams P) {
    constexpr int NT = CSCO_NT, NW = NT / 64, CH = 64 * CSCO_K;
    constexpr KeyT ZEROK = KeyInfo<KeyT>::ZEROK;
    constexpr KeyT MAXK = KeyInfo<KeyT>::MAXK;
    constexpr u64 CNT1 = 1ull << CSCO_CNT_SHIFT, R2MASK = CNT1 - 1ull;
    extern __shared__ __align__(16) unsigned char smem[];
    const int G = P.G, NBKT = 1 << P.lg_buckets;
    u64 *acc = (u64 *)smem;                                   // [G]
    u32 *tab = (u32 *)(smem + (size_t)((G + 1) & ~1) * 8);    // [NBKT / 2] two 16-bit bucket counters / offsets per word
    u16 *tab16 = (u16 *)tab;
    u64 *s_red = (u64 *)(tab + NBKT / 2);                     // [NW]
    KeyT *s_k = (KeyT *)(s_red + NW);                         // [2] smallest / largest non-zero key
    u32 *s_misc = (u32 *)(s_red + NW + 2);                    // [0] stored zeros  [1] negatives  [2] largest bucket
    KeyT *A = (KeyT *)(smem + csco_fixed_lds_bytes(G, P.lg_buckets));

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const InT *data = (const InT *)P.data;
    const IdxT *indices = (const IdxT *)P.indices, *indptr = (const IdxT *)P.indptr;
    constexpr int UL = 8; // independent entries per thread in flight

    for (int gene = blockIdx.x; gene < P.nb; gene += gridDim.x) {
        const long long col = P.gene_cols ? (long long)P.gene_cols[gene] : P.col0 + gene;
        const long long k0 = (long long)indptr[col] - P.kshift, k1 = (long long)indptr[col + 1] - P.kshift;
        const long long ns_ll = k1 - k0;
        if (ns_ll > (long long)P.key_cap) { // uniform: this gene takes the general route
            if (tid == 0) P.fallback[gene] = 1u;
            continue;
        }
        const int ns = (int)ns_ll;
        // ---- 1. per-group value sums, key range, stored zeros, negatives ----
        double *sums = (double *)acc;
        for (int g = tid; g < G; g += NT) sums[g] = 0.0;
        for (int b = tid; b < NBKT / 2; b += NT) tab[b] = 0u;
        if (tid == 0) { s_k[0] = MAXK; s_k[1] = (KeyT)0; s_misc[0] = 0u; s_misc[1] = 0u; s_misc[2] = 0u; }
        __syncthreads();
        {
            u32 my_zero = 0, my_neg = 0;
            KeyT tmin = MAXK, tmax = (KeyT)0;
            for (long long kb = k0; kb < k1; kb += NT * UL) {
                InT v[UL];
                int cd[UL];
#pragma unroll
                for (int u = 0; u < UL; ++u) {
                    const long long k = kb + u * NT + tid;
                    v[u] = k < k1 ? data[k] : (InT)0;
                    cd[u] = k < k1 ? (P.codes ? P.codes[(long long)indices[k]] : (int)indices[k]) : 0;
                }
#pragma unroll
                for (int u = 0; u < UL; ++u) {
                    const long long k = kb + u * NT + tid;
                    if (k < k1) {
                        if (v[u] != (InT)0) {
                            const KeyT key = key_of(v[u]);
                            atomicAdd(&sums[cd[u]], P.is_log1p ? key_to_expm1(key, P.dt) : key_to_double(key, P.dt));
                            tmin = key < tmin ? key : tmin;
                            tmax = key > tmax ? key : tmax;
                            my_neg += key < ZEROK ? 1u : 0u;
                        } else ++my_zero; // a stored zero is an implicit zero
                    }
                }
            }
            tmin = wave_min_key(tmin);
            tmax = wave_max_key(tmax);
            my_zero = (u32)wave_sum((int)my_zero);
            my_neg = (u32)wave_sum((int)my_neg);
            if (lane == 0) {
                atomicMin(&s_k[0], tmin);
                atomicMax(&s_k[1], tmax);
                if (my_zero) atomicAdd(&s_misc[0], my_zero);
                if (my_neg) atomicAdd(&s_misc[1], my_neg);
            }
        }
        __syncthreads();
        const int n = ns - (int)s_misc[0];                   // stored non-zeros
        const long long n0 = P.n_cells - n;                  // zeros of the column
        const long long nneg = (long long)s_misc[1];
        const KeyT kmin = s_k[0], kmax = s_k[1];
        for (int g = tid; g < G; g += NT) P.out_sum[(size_t)gene * G + g] = sums[g];
        __syncthreads();
        for (int g = tid; g < G; g += NT) acc[g] = 0ull;
        u64 tie = 0;
        bool sorted_form = P.force_sorted != 0;
        const int shift = n > 0 ? max(0, key_bits((KeyT)(kmax - kmin)) - P.lg_buckets) : 0;
        if (n > 0 && !sorted_form) {
            // ---- 2. bucket sizes ----
            for (long long kb = k0; kb < k1; kb += NT * UL) {
                InT v[UL];
#pragma unroll
                for (int u = 0; u < UL; ++u) {
                    const long long k = kb + u * NT + tid;
                    v[u] = k < k1 ? data[k] : (InT)0;
                }
#pragma unroll
                for (int u = 0; u < UL; ++u)
                    if (v[u] != (InT)0) {
                        const u32 b = (u32)((KeyT)(key_of(v[u]) - kmin) >> shift);
                        atomicAdd(&tab[b >> 1], (b & 1u) ? 0x10000u : 1u); // no carry: a counter stays below 2^16
                    }
            }
            __syncthreads();
            u64 sq = 0;
            u32 mx = 0;
            for (int b = tid; b < NBKT; b += NT) { const u32 cb = tab16[b]; sq += (u64)cb * cb; mx = max(mx, cb); }
            sq = wave_sum(sq);
            mx = (u32)wave_incl_scan_max((int)mx);
            if (lane == 63) { s_red[wave] = sq; atomicMax(&s_misc[2], mx); }
            __syncthreads();
            u64 sumsq = 0;
            for (int w = 0; w < NW; ++w) sumsq += s_red[w];
            sorted_form = s_misc[2] > (u32)CSCO_MAX_BUCKET || sumsq > (u64)CSCO_MAX_AVG * (u64)n; // uniform
            __syncthreads();
        }
        if (n > 0 && !sorted_form) {
            // ---- 3. bucket offsets, keys into their buckets ----
            block_excl_scan_u16<NT>(tab16, NBKT, (u32 *)A, tid); // the key buffer is still free: scan scratch
            for (long long kb = k0; kb < k1; kb += NT * UL) {
                InT v[UL];
#pragma unroll
                for (int u = 0; u < UL; ++u) {
                    const long long k = kb + u * NT + tid;
                    v[u] = k < k1 ? data[k] : (InT)0;
                }
#pragma unroll
                for (int u = 0; u < UL; ++u)
                    if (v[u] != (InT)0) {
                        const KeyT key = key_of(v[u]);
                        const u32 b = (u32)((KeyT)(key - kmin) >> shift);
                        const u32 old = atomicAdd(&tab[b >> 1], (b & 1u) ? 0x10000u : 1u);
                        A[(b & 1u) ? (old >> 16) : (old & 0xFFFFu)] = key;
                    }
            }
            if (tid < 4) A[n + tid] = MAXK; // the bucket walk below reads up to 3 keys past a bucket's end
            __syncthreads(); // now tab16[b] = one past bucket b; it starts at tab16[b - 1]
            // ---- 4. every stored entry against its own bucket ----
            for (long long kb = k0; kb < k1; kb += NT * UL) {
                InT v[UL];
                int cd[UL];
#pragma unroll
                for (int u = 0; u < UL; ++u) {
                    const long long k = kb + u * NT + tid;
                    v[u] = k < k1 ? data[k] : (InT)0;
                    cd[u] = k < k1 ? (P.codes ? P.codes[(long long)indices[k]] : (int)indices[k]) : 0;
                }
#pragma unroll
                for (int u = 0; u < UL; ++u) {
                    if (v[u] != (InT)0) {
                        const KeyT q = key_of(v[u]);
                        const u32 b = (u32)((KeyT)(q - kmin) >> shift);
                        const u32 lo = b ? tab16[b - 1] : 0u, hi = tab16[b];
                        u32 less = 0, eq = 0;
                        // 4 keys per step; keys past the bucket's end belong to later buckets (larger than q) or are the
                        // MAXK pad, so they count for neither sum
                        for (u32 j = lo; j < hi; j += 4) {
                            const KeyT a0 = A[j], a1 = A[j + 1], a2 = A[j + 2], a3 = A[j + 3];
                            less += (a0 < q ? 1u : 0u) + (a1 < q ? 1u : 0u) + (a2 < q ? 1u : 0u) + (a3 < q ? 1u : 0u);
                            eq += (a0 == q ? 1u : 0u) + (a1 == q ? 1u : 0u) + (a2 == q ? 1u : 0u) + (a3 == q ? 1u : 0u);
                        }
                        const u32 s = lo + less;
                        const u64 add = 2ull * s + eq + 1ull + ((q > ZEROK) ? 2ull * (u64)n0 : 0ull);
                        atomicAdd(&acc[cd[u]], add + CNT1);
                        tie += (u64)eq * eq - 1ull;
                    }
                }
            }
        } else if (n > 0) {
            // ---- sorted form: keys -> LDS, sort, tie blocks, two look-ups per entry ----
            const int ncap = (ns + CH - 1) / CH * CH;
            if (ncap > P.key_cap) { // uniform
                if (tid == 0) P.fallback[gene] = 1u;
                __syncthreads();
                continue;
            }
            for (int i = ns + tid; i < ncap; i += NT) A[i] = MAXK;
            for (long long kb = k0; kb < k1; kb += NT * UL) {
                InT v[UL];
#pragma unroll
                for (int u = 0; u < UL; ++u) {
                    const long long k = kb + u * NT + tid;
                    v[u] = k < k1 ? data[k] : (InT)0;
                }
#pragma unroll
                for (int u = 0; u < UL; ++u) {
                    const long long k = kb + u * NT + tid;
                    if (k < k1) A[k - k0] = v[u] != (InT)0 ? key_of(v[u]) : MAXK; // stored zeros sort past the n keys
                }
            }
            __syncthreads();
            block_sort_hybrid<KeyT, NT, CSCO_K>(A, ncap, tid);
            const u32 un = (u32)n, top = top_pow2(un);
            for (int i = tid; i < n; i += NT) {
                const KeyT k = A[i];
                if ((i == 0 || A[i - 1] != k) && i + 1 < n && A[i + 1] == k) {
                    const u64 t = upper_bound_pow2(A, un, top, k) - (u32)i;
                    tie += t * t * t - t;
                }
            }
            for (long long kb = k0; kb < k1; kb += NT * UL) {
                InT v[UL];
                int cd[UL];
#pragma unroll
                for (int u = 0; u < UL; ++u) {
                    const long long k = kb + u * NT + tid;
                    v[u] = k < k1 ? data[k] : (InT)0;
                    cd[u] = k < k1 ? (P.codes ? P.codes[(long long)indices[k]] : (int)indices[k]) : 0;
                }
#pragma unroll
                for (int u = 0; u < UL; ++u) {
                    if (v[u] != (InT)0) {
                        const KeyT q = key_of(v[u]);
                        const u32 s = lower_bound_pow2(A, un, top, q);
                        u32 e = s + 1;
                        if (e < un && A[e] == q) e = upper_bound_pow2(A, un, top, q);
                        const u64 add = (u64)s + (u64)e + 1ull + ((q > ZEROK) ? 2ull * (u64)n0 : 0ull);
                        atomicAdd(&acc[cd[u]], add + CNT1);
                    }
                }
            }
        }
        tie = wave_sum(tie);
        __syncthreads(); // (also: s_red's readers of step 2 are done)
        if (lane == 0) s_red[wave] = tie;
        __syncthreads();
        u64 tie_total = 0;
        for (int w = 0; w < NW; ++w) tie_total += s_red[w];
        tie_total += (u64)n0 * (u64)n0 * (u64)n0 - (u64)n0;
        for (int g = tid; g < G; g += NT) {
            const long long n_g = P.counts[g];
            const u64 a = acc[g];
            const long long z = n_g - (long long)(a >> CSCO_CNT_SHIFT);
            const u64 r2 = (a & R2MASK) + (u64)z * (u64)(2 * nneg + n0 + 1);
            P.out_2u[(size_t)gene * G + g] = 2ll * (P.n_cells - n_g) * n_g + n_g * (n_g + 1) - (long long)r2;
            P.out_tie[(size_t)gene * G + g] = tie_total;
        }
        __syncthreads();
    }
}
