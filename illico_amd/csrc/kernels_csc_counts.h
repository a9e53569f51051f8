// CSC, count-valued genes, small groups: one pass over a gene's stored entries builds the per-group value histograms in
// LDS -- h[group][value], 8-bit cells, 64 (or 32) values: 64 bytes per group, 128 KB for 2000 groups -- with ONE non-returning
// LDS atomic per entry; a sweep with one thread per group then turns histograms into the statistics:
//   OVO:  S2 = sum_{c>=1} tB[c] (2 zA + 2 cumA[c] + tA[c]) + zB zA,      tie = T_A + sum_{c>=1} tB (3 tA (tA+tB) + tB^2 - 1) + (t0^3 - t0)
//   OVR:  r2 = sum_{c>=1} tB[c] (2 n0 + 2 cum[c] + t[c] + 1) + zB (n0 + 1),  tie = sum_{c>=1} (t^3 - t) + (n0^3 - n0)
// (zA, zB, n0 = implicit zeros of the reference / the group / the column, t0 = zA + zB) -- the same integers the sort-based
// CSC kernels produce (kernels_csc_gene.h, kernels_ovr.h; sparse_ovo.py:58-85, sparse_ovr.py:70-83), without sorting,
// regrouping or searching anything.  Nothing but the CSC arrays is read from HBM: 1.9 GB at C3.
//
// Takes a gene only if every stored value is an integer in [1, RT) (stored zeros are dropped: they are zeros) and
// -- host-checked -- at most CSCC_MAX_BIG ranked groups have more than 255 cells (those get 32-bit cells; the others
// 8-bit cells); other genes set fallback[gene] and go to the general CSC routes.
#pragma once
#include "common.h"

#define CSCC_NT 1024
#define CSCC_RT 64 // widest table (values 1 .. 63); the 32-value form is used when 64 bytes per group do not fit LDS

struct CscCountsParams {
    const void *data, *indices, *indptr; // CSC arrays (device); stored entry k lives at data[k - kshift], indices[k - kshift]
    long long kshift;
    long long col0;                      // first gene of the batch (contiguous batches)
    const int *gene_cols;                // or: the batch's genes as a column list (absolute indices); nullptr = contiguous
    int nb;
    const int *codes;                    // [n_cells] group code per cell; nullptr: `indices` already holds group codes
    const int *counts;                   // [G]
    int G, ref;                          // ref == -1: OVR
    long long n_cells;
    const signed char *big_slot;         // [G] -1, or the row of the group in the 32-bit table (groups of more than 255 cells)
    u32 *fallback;                       // [nb] set to 1 for genes this kernel cannot take
    long long *out_2u;
    u64 *out_tie;
    double *out_sum;
};

static inline size_t cscc_lds_bytes(int G, int rt) { return (size_t)G * rt + (((size_t)G + 15) & ~(size_t)15); } // cells + slot bytes

#define CSCC_MAX_BIG 8
template <typename InT, typename IdxT, bool OVR, int RT, bool HAS_BIG>
__global__ __launch_bounds__(CSCC_NT) void k_csc_counts(CscCountsParams P) {
    constexpr int NT = CSCC_NT, WPG = RT / 4; // words per group
    extern __shared__ __align__(16) u32 cscc_h[];             // [WPG][G] words: cell (g, c) = byte c % 4 of word [c / 4][g]
    // (word-major: a group's words are G apart, so lanes = consecutive groups read consecutive words in the sweep and
    //  the increments of random groups spread over all banks)
    __shared__ u32 hsel[RT];   // OVO: histogram of the reference group's stored values; OVR: of the whole column
    __shared__ u32 hbig[HAS_BIG ? CSCC_MAX_BIG * RT : 1]; // 32-bit cells of the few groups with more than 255 cells
    signed char *slot = (signed char *)(cscc_h + (size_t)WPG * P.G); // [G] copy of big_slot (HAS_BIG)
    __shared__ u32 cum[RT + 1]; // cum[c] = # selected stored values < c (c >= 1)
    __shared__ u64 s_T, s_sum;
    __shared__ u32 s_nnz;
    __shared__ int s_bad;
    const int tid = threadIdx.x;
    const int G = P.G, ref = P.ref;
    const InT *data = (const InT *)P.data;
    const IdxT *indices = (const IdxT *)P.indices, *indptr = (const IdxT *)P.indptr;

    for (int gene = blockIdx.x; gene < P.nb; gene += gridDim.x) {
        const long long col = P.gene_cols ? (long long)P.gene_cols[gene] : P.col0 + gene;
        const long long k0 = (long long)indptr[col] - P.kshift, k1 = (long long)indptr[col + 1] - P.kshift;
        constexpr int UL = 8; // independent entries per thread in flight
        // (the first round's loads are in flight while the tables are zeroed)
        InT vn[UL];
        IdxT in[UL];
#pragma unroll
        for (int u = 0; u < UL; ++u) {
            const long long k = k0 + u * NT + tid;
            vn[u] = k < k1 ? data[k] : (InT)0;
            in[u] = k < k1 ? indices[k] : (IdxT)0;
        }
        for (int i = tid; i < G * WPG; i += NT) cscc_h[i] = 0;
        if (tid < RT) hsel[tid] = 0;
        if (HAS_BIG) {
            for (int i = tid; i < CSCC_MAX_BIG * RT; i += NT) hbig[i] = 0;
            for (int i = tid; i < G; i += NT) slot[i] = P.big_slot[i];
        }
        if (tid == 0) s_bad = 0;
        __syncthreads();
        bool bad = false;
        // two-stage pipeline over the gene's entries: the values / row indices of round i + 1 are requested before round
        // i's group codes (a dependent gather) and LDS atomics, so one HBM round trip per round is off the critical path
        for (long long kb = k0; kb < k1; kb += NT * UL) {
            InT v[UL];
            int cd[UL];
#pragma unroll
            for (int u = 0; u < UL; ++u) {
                v[u] = vn[u];
                cd[u] = P.codes ? P.codes[(long long)in[u]] : (int)in[u]; // (entries past k1: row 0, value 0 -> ignored)
            }
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
                    const int c = (v[u] > (InT)0 && v[u] < (InT)RT) ? (int)v[u] : 0;
                    if (c == 0 || (InT)c != v[u]) bad = true; // negative, fractional, NaN or beyond the table
                    else {
                        const int bs = HAS_BIG ? (int)slot[cd[u]] : -1;
                        if (bs >= 0) atomicAdd(&hbig[bs * RT + c], 1u);
                        else if (OVR || cd[u] != ref) atomicAdd(&cscc_h[(c >> 2) * G + cd[u]], 1u << ((c & 3) * 8));
                        if (OVR || cd[u] == ref) atomicAdd(&hsel[c], 1u);
                    }
                }
        }
        if (bad) s_bad = 1;
        __syncthreads();
        if (s_bad) { // uniform: this gene takes the general routes
            if (tid == 0) P.fallback[gene] = 1u;
            __syncthreads();
            continue;
        }
        if (tid == 0) {
            u32 run = 0;
            u64 T = 0, sum = 0;
            cum[0] = 0; cum[1] = 0;
            for (int c = 1; c < RT; ++c) {
                const u64 t = hsel[c];
                cum[c] = run;
                run += (u32)t;
                T += t * t * t - t;
                sum += t * (u64)c;
            }
            cum[RT] = run;
            s_nnz = run;
            s_T = T;
            s_sum = sum;
        }
        __syncthreads();
        const u64 nnz_sel = s_nnz, T_sel = s_T;
        const long long n_ref = OVR ? 0 : P.counts[OVR ? 0 : ref];
        const u64 zsel = (u64)((OVR ? P.n_cells : n_ref) - (long long)nnz_sel); // zA (OVO) or n0 (OVR)
        for (int g = tid; g < G; g += NT) {
            const size_t o = (size_t)gene * G + g;
            if (!OVR && g == ref) {
                P.out_2u[o] = -2;
                P.out_tie[o] = 0;
                P.out_sum[o] = (double)s_sum;
                continue;
            }
            // 32-bit inner terms (host-checked: n_ref < 30000 for OVO, n_cells < 2^30), one 32 x 32 -> 64 multiply-add each;
            // a word whose four cells are empty for every lane of the wavefront (most of the table) is skipped
            u64 acc = 0, tie = 0;
            u32 nnz_g = 0, vsum = 0;
            const int bs = HAS_BIG ? (int)slot[g] : -1;
#pragma unroll
            for (int i = 0; i < WPG; ++i) {
                const u32 w = cscc_h[i * G + g];
                if (__ballot(w != 0 || bs >= 0) == 0ull) continue;
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const int c = i * 4 + k;
                    if (c == 0) continue;
                    const u32 tB = bs >= 0 ? hbig[bs * RT + c] : ((w >> (k * 8)) & 0xFFu);
                    const u32 tS = hsel[c], lo = cum[c];
                    nnz_g += tB;
                    vsum += tB * (u32)c;
                    if (OVR) acc += (u64)tB * (u32)(2u * (u32)zsel + 2u * lo + tS + 1u);
                    else {
                        acc += (u64)tB * (u32)(2u * (u32)zsel + 2u * lo + tS);
                        tie += (u64)tB * (u32)(3u * tS * (tS + tB) + tB * tB - 1u);
                    }
                }
            }
            const long long n_g = P.counts[g];
            const u64 zB = (u64)(n_g - (long long)nnz_g);
            if (OVR) {
                acc += zB * (zsel + 1ull);
                P.out_2u[o] = 2ll * (P.n_cells - n_g) * n_g + n_g * (n_g + 1) - (long long)acc;
                P.out_tie[o] = T_sel + (zsel * zsel * zsel - zsel);
            } else {
                acc += zB * zsel;
                const u64 t0 = zsel + zB;
                P.out_2u[o] = 2ll * n_ref * n_g - (long long)acc;
                P.out_tie[o] = T_sel + tie + (t0 * t0 * t0 - t0);
            }
            P.out_sum[o] = (double)vsum;
        }
        __syncthreads();
    }
}
