// CSC one-versus-rest for ANY values in ONE kernel per gene: the gene's stored non-zeros are sorted as bare keys inside
// LDS (no payload: a rank only needs the sorted VALUES), then every stored entry is read a second time, looks its own
// value up in the sorted column (lower / upper bound = the tie block [s, e) it falls in) and adds 2 * avg_rank =
// s + e + 1 to its group's LDS accumulator.  Nothing but the CSC arrays is read and nothing but the [gene][G] statistics
// is written; the general route it replaces regroups the entries in HBM (k_csc_regroup), sorts (key, group) pairs with a
// segmented radix sort in HBM and sweeps them (k_ovr_gene): 23 ms at C3 shape against 3 ms here.
//
// Device counterpart of sparse_ovr_mwu_kernel + its CSC entry (illico/ovr/sparse_ovr.py:23-97, :100-155) and
// _accumulate_group_ranksums_from_argsort (utils/ranking.py:7-49) for one gene at a time: argsort + tie-block walk
// become sort(values) + two binary searches per entry; the zeros stay implicit (sparse_ovr.py:70-83): n0 = N - nnz
// cells tie at rank n_neg + (n0 + 1) / 2 and shift every positive entry by n0.
//
// A gene with more stored entries than the LDS key buffer sets fallback[gene]; the host sends those genes through the
// general route.
#pragma once
#include "common.h"

#define CSCO_NT 1024
#define CSCO_K 16 // keys per lane of the register sort phases: 1024-key chunks

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
    int key_cap;                         // LDS key slots (multiple of 64 * CSCO_K)
    u32 *fallback;                       // [nb] set to 1 for genes this kernel cannot take
    long long *out_2u;                   // [nb][G] 2 U (U of "the rest", dense_ovr.py:57-61)
    u64 *out_tie;                        // [nb][G] sum (t^3 - t), the same for every group of a gene
    double *out_sum;                     // [nb][G] per-group value sums
};

__host__ __device__ static inline size_t csco_fixed_lds_bytes(int G) {
    // acc (value sums in pass 1, doubled rank sums in pass 2) | stored non-zeros per group | reductions
    return (size_t)((G + 1) & ~1) * 8 + (size_t)((G + 3) & ~3) * 4 + 256; // a multiple of 16
}
static inline int csco_key_cap(int G, size_t key_size, size_t lds_max) {
    const size_t fixed = csco_fixed_lds_bytes(G);
    if (fixed + (size_t)64 * CSCO_K * key_size > lds_max) return 0;
    return (int)((lds_max - fixed) / key_size / (64 * CSCO_K)) * (64 * CSCO_K);
}

template <typename InT, typename IdxT, typename KeyT>
__global__ __launch_bounds__(CSCO_NT) void k_csc_ovr_gene(CscOvrParams P) {
    constexpr int NT = CSCO_NT, NW = NT / 64, CH = 64 * CSCO_K;
    constexpr KeyT ZEROK = KeyInfo<KeyT>::ZEROK;
    constexpr KeyT MAXK = KeyInfo<KeyT>::MAXK;
    extern __shared__ __align__(16) unsigned char smem[];
    const int G = P.G;
    u64 *acc = (u64 *)smem;                                   // [G]
    u32 *gcnt = (u32 *)(smem + (size_t)((G + 1) & ~1) * 8);   // [G]
    u64 *s_red = (u64 *)(gcnt + ((G + 3) & ~3));              // [NW]
    u32 *s_misc = (u32 *)(s_red + NW);                        // [4]
    KeyT *A = (KeyT *)(smem + csco_fixed_lds_bytes(G));

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
        const int ncap = (ns + CH - 1) / CH * CH;
        // ---- 1. keys -> LDS, per-group stored non-zeros and value sums ----
        double *sums = (double *)acc;
        for (int g = tid; g < G; g += NT) { sums[g] = 0.0; gcnt[g] = 0u; }
        if (tid == 0) s_misc[0] = 0u;
        for (int i = ns + tid; i < ncap; i += NT) A[i] = MAXK;
        __syncthreads();
        u32 my_zero = 0;
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
                    const bool nz = v[u] != (InT)0;
                    const KeyT key = key_of(v[u]);
                    A[k - k0] = nz ? key : MAXK; // a stored zero is an implicit zero: out of the sorted column
                    if (nz) {
                        atomicAdd(&gcnt[cd[u]], 1u);
                        atomicAdd(&sums[cd[u]], P.is_log1p ? key_to_expm1(key, P.dt) : key_to_double(key, P.dt));
                    } else ++my_zero;
                }
            }
        }
        if (my_zero) atomicAdd(&s_misc[0], my_zero);
        __syncthreads();
        const int n = ns - (int)s_misc[0];                   // stored non-zeros
        const long long n0 = P.n_cells - n;                  // zeros of the column
        for (int g = tid; g < G; g += NT) P.out_sum[(size_t)gene * G + g] = sums[g];
        __syncthreads();
        for (int g = tid; g < G; g += NT) acc[g] = 0ull;
        // ---- 2. sort the column's non-zero values ----
        block_sort_hybrid<KeyT, NT, CSCO_K>(A, ncap, tid); // starts and ends with a barrier of its own phases
        const u32 un = (u32)n, top = top_pow2(un);
        // ---- 3. tie blocks of the non-zeros ----
        u64 tie = 0;
        for (int i = tid; i < n; i += NT) {
            const KeyT k = A[i];
            if ((i == 0 || A[i - 1] != k) && i + 1 < n && A[i + 1] == k) {
                const u64 t = upper_bound_pow2(A, un, top, k) - (u32)i;
                tie += t * t * t - t;
            }
        }
        tie = wave_sum(tie);
        if (lane == 0) s_red[wave] = tie;
        // ---- 4. every stored entry: 2 * avg_rank = s + e + 1 (+ 2 n0 above the zeros) into its group ----
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
                    atomicAdd(&acc[cd[u]], add);
                }
            }
        }
        __syncthreads();
        u64 tie_total = 0;
        for (int w = 0; w < NW; ++w) tie_total += s_red[w];
        tie_total += (u64)n0 * (u64)n0 * (u64)n0 - (u64)n0;
        const long long nneg = (long long)lower_bound_pow2(A, un, top, ZEROK);
        for (int g = tid; g < G; g += NT) {
            const long long n_g = P.counts[g];
            const long long z = n_g - (long long)gcnt[g];
            const u64 r2 = acc[g] + (u64)z * (u64)(2 * nneg + n0 + 1);
            P.out_2u[(size_t)gene * G + g] = 2ll * (P.n_cells - n_g) * n_g + n_g * (n_g + 1) - (long long)r2;
            P.out_tie[(size_t)gene * G + g] = tie_total;
        }
        __syncthreads();
    }
}
