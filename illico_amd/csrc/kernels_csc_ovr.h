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
//
// DENSE columns (and whatever else exceeds the LDS key buffer) take the same rank kernel in pieces: k_ovr_partition splits
// a gene's non-zero keys by VALUE into parts of at most key_cap keys (histogram over 8192 value buckets, scan, part =
// cumulative count / quota, (key, group code) records appended per part in HBM), k_ovr_rank_gene_parts (kernels_ovr_parts.h)
// ranks each part in LDS exactly as a CSC column is ranked here -- a key's rank is the number of keys in lower parts plus its
// rank inside its own part -- one workgroup per gene, its parts one after the other.  Dense continuous OVR at C4 shape: 68 ms (segmented radix sort of (key, group) pairs in HBM) -> see DESIGN.
#pragma once
#include "common.h"
#include "kernels_finalize.h"
#include "kernels_sparse.h"

#define CSCO_NT 1024
#define CSCO_K 16        // keys per lane of the register sort phases (sorted form): 1024-key chunks
#define CSCO_MAX_BUCKET 192 // bucket form only while no bucket holds more keys than this ...
#define CSCO_MAX_AVG 64     // ... and the average entry shares its bucket with at most this many
#define CSCO_CNT_SHIFT 40   // acc word: doubled rank sum below, stored non-zeros of the group above
#define OVRP_NT 512         // k_ovr_partition
#define OVRP_LG 13          // its coarse buckets: (key - kmin) >> shift, 8192 of them over the gene's own key range
#define OVRP_PMAX 255       // parts per gene at most (8-bit part ids; 128 until late in round 5: a column of more cells than 128 parts hold left the route)

struct CscOvrParams {
    // CSC source
    const void *data, *indices, *indptr; // CSC arrays (device); stored entry k lives at data[k - kshift], indices[k - kshift]
    long long kshift;
    long long col0;                      // first gene of the batch (contiguous batches)
    const int *gene_cols;                // or: the batch's genes as a column list (absolute indices); nullptr = contiguous
    const int *codes;                    // [n_cells] group code per cell; nullptr: `indices` already holds group codes
    const u16 *codes16;                  // the same as 16-bit values (fewer cache lines per gather), or nullptr
    int nb;
    const int *counts;                   // [G]
    int G, dt, is_log1p;
    long long n_cells;
    int key_cap;                         // LDS key slots
    int lg_buckets;                      // log2 of the bucket count
    int force_sorted;                    // 1: every gene takes the sorted form (tests / A-B)
    u32 *fallback;                       // [nb] set to 1 for genes this kernel cannot take
    long long *out_2u;                   // [nb][G] 2 U (U of "the rest", dense_ovr.py:57-61)
    u64 *out_tie;                        // [nb][G] sum (t^3 - t), the same for every group of a gene
    int tie_f64;                         // out_tie = the bits of the float64 tie sum of the reference's sparse path (tie_f64_sparse)
    u64 *acc_global;                     // ACCG: [nb][G] the accumulators in HBM (zeroed by the host), for more groups than LDS holds beside the keys
    int g_lds;                           // ACCG: the groups [0, g_lds) keep theirs in LDS all the same (what is left beside the keys)
    // (per-group value sums are not formed here: they would be order-dependent float64 atomics.  The host launches
    //  k_csc_value_sums / k_group_sums_rows, kernels_sums.h, whose results do not depend on the order of arrival.)
};

__host__ __device__ static inline size_t csco_fixed_lds_bytes(int G, int lg_buckets, bool parts, bool accg = false) {
    // acc (packed rank sums / counts; with ACCG: G = the groups that keep theirs in LDS, the others live in HBM) | bucket table |
    // reductions: a multiple of 16
    (void)parts; (void)accg;
    return (size_t)((G + 1) & ~1) * 8 + ((size_t)2 << lg_buckets) + 256;
}
static inline int csco_key_cap(int G, int lg_buckets, size_t key_size, size_t lds_max, bool parts = false, bool accg = false) {
    const size_t fixed = csco_fixed_lds_bytes(accg ? 0 : G, lg_buckets, parts, accg);
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

// exclusive scan of the 16-bit counters arr[0..n) in place; n = NT * per, per a multiple of 2; totals below 2^16.  tmp: [NT / 64] words.
template <int NT> __device__ __forceinline__ void block_excl_scan_u16_waves(u16 *arr, int n, u32 *tmp, int tid) {
    constexpr int NW = NT / 64;
    const int per = n / NT, lane = tid & 63, wave = tid >> 6;
    u32 *w = (u32 *)arr + (size_t)tid * (per / 2);
    u32 s = 0;
    for (int i = 0; i < per / 2; ++i) { const u32 x = w[i]; s += (x & 0xFFFFu) + (x >> 16); }
    const u32 incl = (u32)wave_incl_scan_add((int)s);
    if (lane == 63) tmp[wave] = incl;
    __syncthreads();
    if (wave == 0) {
        const u32 t = lane < NW ? tmp[lane] : 0u;
        const u32 ti = (u32)wave_incl_scan_add((int)t);
        if (lane < NW) tmp[lane] = ti - t;
    }
    __syncthreads();
    u32 run = tmp[wave] + incl - s;
    for (int i = 0; i < per / 2; ++i) {
        const u32 x = w[i];
        const u32 lo = run;
        run += x & 0xFFFFu;
        const u32 hi = run;
        run += x >> 16;
        w[i] = (lo & 0xFFFFu) | (hi << 16);
    }
    __syncthreads();
}

// One source entry: CSC stored entry k of the column, or record k of the part.  Raw loads (value / row index or group
// code) are split from what is derived from them (key; group code through the codes[row] gather) so that the entry
// loops can request round i + 1's raw loads before round i's gather and LDS work.
template <typename InT, typename IdxT, typename KeyT, bool PARTS> struct OvrSource {
    typedef typename std::conditional<PARTS, KeyT, InT>::type RawV;
    typedef typename std::conditional<PARTS, u16, IdxT>::type RawI;
    const InT *data;
    const IdxT *indices;
    const int *codes;
    const u16 *codes16;
    const KeyT *pkeys;
    const u16 *pcodes;
    __device__ __forceinline__ RawV raw_v(long long k, bool in) const {
        if constexpr (PARTS) return in ? pkeys[k] : (KeyT)0;
        else return in ? data[k] : (InT)0;
    }
    __device__ __forceinline__ RawI raw_i(long long k, bool in) const {
        if constexpr (PARTS) return in ? pcodes[k] : (u16)0;
        else return in ? indices[k] : (IdxT)0;
    }
    // key of an entry; nz = it takes part in the ranking (a stored zero is an implicit zero)
    __device__ __forceinline__ KeyT key_from(RawV v, bool in, bool &nz) const {
        if constexpr (PARTS) { nz = in; return v; }
        else { nz = v != (InT)0; return key_of(v); }
    }
    __device__ __forceinline__ int code_from(RawI i) const {
        if constexpr (PARTS) return (int)i;
        else return codes16 ? (int)codes16[(long long)i] : (codes ? codes[(long long)i] : (int)i); // (entries past the end carry row 0: a valid, ignored look-up)
    }
};

// for every entry k in [k0, k1), UL per thread and round: body(u, k, key, nz, code).  Two-stage pipeline: the raw loads of the next
// round are issued before this round's code gather and bodies.  No load sits under a per-lane condition (hipcc gives each such load a
// basic block of its own: `s_and_saveexec` + branch): a round's requests are clamped to the column's last entry, full rounds run
// their bodies unconditionally, only the last, partial round checks each entry.
template <bool WANT_CODE, int NT, int UL, typename Src, typename KeyT, typename Body>
__device__ __forceinline__ void ovr_for_entries(const Src &src, long long k0, long long k1, int tid, Body &&body) {
    constexpr int PER = NT * UL;
    const int n = (int)(k1 - k0); // (a column / a part: at most key_cap entries)
    if (n <= 0) return;
    typename Src::RawV vn[UL];
    typename Src::RawI in[UL];
    auto request = [&](int base) {
#pragma unroll
        for (int u = 0; u < UL; ++u) {
            const long long k = k0 + min(base + u * NT + tid, n - 1);
            vn[u] = src.raw_v(k, true);
            if (WANT_CODE) in[u] = src.raw_i(k, true);
        }
    };
    request(0);
    for (int base = 0; base < n; base += PER) {
        KeyT key[UL];
        bool nz[UL];
        int cd[UL];
#pragma unroll
        for (int u = 0; u < UL; ++u) {
            key[u] = src.key_from(vn[u], true, nz[u]);
            cd[u] = WANT_CODE ? src.code_from(in[u]) : 0;
        }
        if (base + PER < n) request(base + PER); // uniform
        if (base + PER <= n) { // uniform: a full round
#pragma unroll
            for (int u = 0; u < UL; ++u) body(u, k0 + base + u * NT + tid, key[u], nz[u], cd[u]);
        } else {
#pragma unroll
            for (int u = 0; u < UL; ++u)
                if (base + u * NT + tid < n) body(u, k0 + base + u * NT + tid, key[u], nz[u], cd[u]);
        }
    }
}

// ACCG: the per-group accumulators in HBM (global atomics, read back at L2) instead of LDS: with thousands of groups acc[G] would
// take the key buffer's place (6000 groups: 48 KB; the C3 gene's 30 000 keys then no longer fit and the gene fell to the HBM sort).
template <typename InT, typename IdxT, typename KeyT, bool ACCG = false>
__global__ __launch_bounds__(CSCO_NT) void k_csc_ovr_gene(CscOvrParams P) {
    constexpr int NT = CSCO_NT, NW = NT / 64, CH = 64 * CSCO_K;
    constexpr KeyT ZEROK = KeyInfo<KeyT>::ZEROK;
    constexpr KeyT MAXK = KeyInfo<KeyT>::MAXK;
    constexpr u64 CNT1 = 1ull << CSCO_CNT_SHIFT, R2MASK = CNT1 - 1ull;
    extern __shared__ __align__(16) unsigned char smem[];
    const int G = P.G, NBKT = 1 << P.lg_buckets;
    const int G_lds = ACCG ? P.g_lds : G;                    // groups whose accumulators are in LDS
    const size_t accb = (size_t)((G_lds + 1) & ~1) * 8;
    u64 *acc_lds = (u64 *)smem;                               // [G_lds]
    u32 *tab = (u32 *)(smem + accb);                          // [NBKT / 2] two 16-bit bucket counters / offsets per word
    u16 *tab16 = (u16 *)tab;
    u64 *s_red = (u64 *)(tab + NBKT / 2);                     // [NW]
    KeyT *s_k = (KeyT *)(s_red + NW);                         // [2] smallest / largest non-zero key
    u32 *s_misc = (u32 *)(s_red + NW + 2);                    // [0] stored zeros  [1] negatives  [2] largest bucket
    KeyT *A = (KeyT *)(smem + csco_fixed_lds_bytes(G_lds, P.lg_buckets, false, ACCG));

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const IdxT *indptr = (const IdxT *)P.indptr;
    constexpr int UL = 8; // independent entries per thread in flight
    // one gene per workgroup
    for (int gene = blockIdx.x; gene < P.nb; gene += gridDim.x) {
        OvrSource<InT, IdxT, KeyT, false> src;
        const long long col = P.gene_cols ? (long long)P.gene_cols[gene] : P.col0 + gene;
        const long long k0 = (long long)indptr[col] - P.kshift, k1 = (long long)indptr[col + 1] - P.kshift;
        src.data = (const InT *)P.data; src.indices = (const IdxT *)P.indices; src.codes = P.codes; src.codes16 = P.codes16;
        const long long ns_ll = k1 - k0;
        if (ns_ll > (long long)P.key_cap) { // uniform: this gene takes the general route
            if (tid == 0) P.fallback[gene] = 1u;
            continue;
        }
        const int ns = (int)ns_ll;
        // ---- 0. key range of the bucket function, from every 8th row of NT entries.  Any monotone bucket function ranks
        // correctly (keys outside the sampled range are clamped into the first / last bucket); the range only balances
        // the buckets. ----
        u64 *acc_hbm = ACCG ? P.acc_global + (size_t)gene * G : nullptr;
        auto acc_add = [&](int g, u64 x) { // one atomic per entry: LDS for the groups that fit there, HBM (L2) for the others
            if (!ACCG || g < G_lds) atomicAdd(&acc_lds[g], x);
            else atomicAdd(&acc_hbm[g], x);
        };
        for (int g = tid; g < G_lds; g += NT) acc_lds[g] = 0ull;
        for (int b = tid; b < NBKT / 2; b += NT) tab[b] = 0u;
        if (tid == 0) { s_k[0] = MAXK; s_k[1] = (KeyT)0; s_misc[0] = 0u; s_misc[1] = 0u; s_misc[2] = 0u; }
        __syncthreads();
        {
            KeyT tmin = MAXK, tmax = (KeyT)0;
            const int row_step = ns > 8 * NT ? 8 : 1;
            for (long long k = k0 + tid; k < k1; k += (long long)NT * row_step) {
                bool nz;
                const KeyT key = src.key_from(src.raw_v(k, true), true, nz);
                if (nz) { tmin = key < tmin ? key : tmin; tmax = key > tmax ? key : tmax; }
            }
            tmin = wave_min_key(tmin);
            tmax = wave_max_key(tmax);
            if (lane == 0) { atomicMin(&s_k[0], tmin); atomicMax(&s_k[1], tmax); }
        }
        __syncthreads();
        const bool have_range = s_k[1] >= s_k[0];
        const KeyT kmin = have_range ? s_k[0] : (KeyT)0, kmax = have_range ? s_k[1] : (KeyT)0;
        const int shift = max(0, key_bits((KeyT)(kmax - kmin)) - P.lg_buckets);
        // table entry of a key: 1 + its bucket (buckets 0 .. NBKT - 2).  Entry 0 stays 0, so that after the scan and the scattering
        // pass bucket b is [entry b, entry b + 1): two neighbouring 16-bit reads, no special case for the first bucket.
        const KeyT last_bucket = (KeyT)(NBKT - 2);
        auto entry_of = [&](KeyT key) -> u32 {
            const KeyT d = key > kmin ? (KeyT)((KeyT)(key - kmin) >> shift) : (KeyT)0;
            return (u32)(d < last_bucket ? d : last_bucket) + 1u;
        };
        bool sorted_form = P.force_sorted != 0;
        // ---- 1. stored zeros, negatives, bucket sizes ----
        {
            u32 my_zero = 0, my_neg = 0;
            ovr_for_entries<false, NT, UL, decltype(src), KeyT>(src, k0, k1, tid, [&](int, long long, KeyT key, bool nz, int) {
                if (nz) {
                    my_neg += key < ZEROK ? 1u : 0u;
                    if (!sorted_form) {
                        const u32 b = entry_of(key);
                        atomicAdd(&tab[b >> 1], (b & 1u) ? 0x10000u : 1u); // no carry: a counter stays below 2^16
                    }
                } else ++my_zero; // a stored zero is an implicit zero
            });
            my_zero = (u32)wave_sum((int)my_zero);
            my_neg = (u32)wave_sum((int)my_neg);
            if (lane == 0) {
                if (my_zero) atomicAdd(&s_misc[0], my_zero);
                if (my_neg) atomicAdd(&s_misc[1], my_neg);
            }
        }
        __syncthreads();
        const int n = ns - (int)s_misc[0];                   // stored non-zeros
        const long long n0 = P.n_cells - n;                  // zeros of the column
        const long long nneg = (long long)s_misc[1];
        u64 tie = 0;
        if (n > 0 && !sorted_form) {
            // ---- 2. how crowded are the buckets? ----
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
            block_excl_scan_u16_waves<NT>(tab16, NBKT, (u32 *)s_red, tid); // (s_red: NW 64-bit slots, the scan wants NW words)
            ovr_for_entries<false, NT, UL, decltype(src), KeyT>(src, k0, k1, tid, [&](int, long long, KeyT key, bool nz, int) {
                if (nz) {
                    const u32 b = entry_of(key);
                    const u32 old = atomicAdd(&tab[b >> 1], (b & 1u) ? 0x10000u : 1u);
                    A[(b & 1u) ? (old >> 16) : (old & 0xFFFFu)] = key;
                }
            });
            if (tid < 4) A[n + tid] = MAXK; // the window below reads up to 3 keys past a bucket's end
            __syncthreads(); // now bucket b = [tab16[b], tab16[b + 1])
            // ---- 4. every stored entry against its bucket: the first four keys in straight-line code (the average bucket holds
            // two); keys past a bucket's end belong to later buckets (larger than q) or are the MAXK pad and count for neither sum ----
            const u64 c_neg = 1ull + CNT1, c_pos = c_neg + 2ull * (u64)n0;
            u32 tie32 = 0; // a thread's share of the column's tie sum: at most 64 keys x 192^2
            ovr_for_entries<true, NT, UL, decltype(src), KeyT>(src, k0, k1, tid, [&](int, long long, KeyT q, bool nz, int cd) {
                if (nz) {
                    const u32 e = entry_of(q);
                    const u32 lo = tab16[e - 1], hi = tab16[e];
                    const KeyT a0 = A[lo], a1 = A[lo + 1], a2 = A[lo + 2], a3 = A[lo + 3];
                    u32 less = (a0 < q ? 1u : 0u) + (a1 < q ? 1u : 0u) + (a2 < q ? 1u : 0u) + (a3 < q ? 1u : 0u);
                    u32 eq = (a0 == q ? 1u : 0u) + (a1 == q ? 1u : 0u) + (a2 == q ? 1u : 0u) + (a3 == q ? 1u : 0u);
                    for (u32 j = lo + 4; j < hi; j += 4) { // rare
                        const KeyT b0 = A[j], b1 = A[j + 1], b2 = A[j + 2], b3 = A[j + 3];
                        less += (b0 < q ? 1u : 0u) + (b1 < q ? 1u : 0u) + (b2 < q ? 1u : 0u) + (b3 < q ? 1u : 0u);
                        eq += (b0 == q ? 1u : 0u) + (b1 == q ? 1u : 0u) + (b2 == q ? 1u : 0u) + (b3 == q ? 1u : 0u);
                    }
                    if (q == MAXK) eq = (hi - lo) - less; // the largest key also matches the pad slots
                    acc_add(cd, ((q > ZEROK) ? c_pos : c_neg) + (u64)(2u * (lo + less) + eq));
                    tie32 += eq * eq - 1u;
                }
            });
            tie += (u64)tie32;
        } else if (n > 0) {
            // ---- sorted form: keys -> LDS, sort, tie blocks, two look-ups per entry ----
            const int ncap = (ns + CH - 1) / CH * CH;
            if (ncap > P.key_cap) { // uniform
                if (tid == 0) P.fallback[gene] = 1u;
                __syncthreads();
                continue;
            }
            for (int i = ns + tid; i < ncap; i += NT) A[i] = MAXK;
            ovr_for_entries<false, NT, UL, decltype(src), KeyT>(src, k0, k1, tid, [&](int, long long k, KeyT key, bool nz, int) {
                A[k - k0] = nz ? key : MAXK; // stored zeros sort past the n keys
            });
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
            ovr_for_entries<true, NT, UL, decltype(src), KeyT>(src, k0, k1, tid, [&](int, long long, KeyT q, bool nz, int cd) {
                if (nz) {
                    const u32 s = lower_bound_pow2(A, un, top, q);
                    u32 e = s + 1;
                    if (e < un && A[e] == q) e = upper_bound_pow2(A, un, top, q);
                    const u64 add = (u64)s + (u64)e + 1ull + ((q > ZEROK) ? 2ull * (u64)n0 : 0ull);
                    acc_add(cd, add + CNT1);
                }
            });
        }
        tie = wave_sum(tie);
        __syncthreads(); // (also: s_red's readers of step 2 are done)
        if (lane == 0) s_red[wave] = tie;
        __syncthreads();
        u64 tie_total = 0;
        for (int w = 0; w < NW; ++w) tie_total += s_red[w];
        if (P.tie_f64) tie_total = tie_f64_sparse(tie_total, n0);
        else tie_total += (u64)n0 * (u64)n0 * (u64)n0 - (u64)n0;
        for (int g = tid; g < G; g += NT) {
            const long long n_g = P.counts[g];
            const u64 a = (ACCG && g >= G_lds) ? atomicAdd(&acc_hbm[g], 0ull) : acc_lds[g]; // (HBM accumulators: read at L2, where the atomics landed)
            const long long z = n_g - (long long)(a >> CSCO_CNT_SHIFT);
            const u64 r2 = (a & R2MASK) + (u64)z * (u64)(2 * nneg + n0 + 1);
            P.out_2u[(size_t)gene * G + g] = 2ll * (P.n_cells - n_g) * n_g + n_g * (n_g + 1) - (long long)r2;
            P.out_tie[(size_t)gene * G + g] = tie_total;
        }
        __syncthreads();
    }
}

// Parts of a gene whose coarse buckets are SKEWED (one holds more than a quarter of `cap`: heavy ties -- log1p of raw counts, counts times a
// constant -- or a crowd of values in a sliver of the range).  A bucket above cap / 2 ("big") gets a part of its own; the small buckets
// are split evenly by their own running count S (the big ones left out) with a quota of cap / 2, so a part of small buckets holds at most
// cap keys:    part(b) = S(b) / (cap / 2) + 2 B(b) + [b is big],   B(b) = big buckets before b
// -- monotone in b, ids with gaps (an unused id is an empty part: its start is the next part's).  A big bucket's part may exceed `cap`: the
// rank kernel takes it in its streaming form when all its keys are equal (one value: nothing to rank, the records are only counted per
// group) and hands the gene to the general route when they are not.  hist: exclusive bucket offsets; big_list: 2 x 128 words of scratch;
// *np_out: the parts, or 0xFFFFFFFF (more than OVRP_PMAX ids, more than 128 big buckets).  Every thread of the workgroup calls it.
// (Until late in round 5 a skewed gene left the route at once: dense OVR on log1p of raw counts ran the per-gene radix sort in HBM, 24 ms
// for 2048 genes against 5 for tie-free values.)
template <int NT>
__device__ __forceinline__ void ovrp_assign_parts_skewed(const u32 *hist, int NB, u32 n, u32 cap, unsigned char *part_of, u32 *pstart, u32 *big_list,
                                                         u32 *np_out, int tid) {
    u32 *bigb = big_list, *bigc = big_list + 128;
    const u32 half = cap / 2u;
    if (tid == 0) *np_out = 0u; // (first: the number of big buckets; then: the largest part id)
    for (int p = tid; p <= OVRP_PMAX; p += NT) pstart[p] = n;
    __syncthreads();
    for (int b = tid; b < NB; b += NT) {
        const u32 c = (b + 1 < NB ? hist[b + 1] : n) - hist[b];
        if (c > half) { const u32 i = atomicAdd(np_out, 1u); if (i < 128u) { bigb[i] = (u32)b; bigc[i] = c; } }
    }
    __syncthreads();
    const u32 nbig = *np_out;
    __syncthreads();
    if (nbig > 128u) { if (tid == 0) *np_out = 0xFFFFFFFFu; __syncthreads(); return; }
    if (tid == 0) *np_out = 0u;
    __syncthreads();
    u32 pmax = 0;
    for (int b = tid; b < NB; b += NT) {
        const u32 lo = hist[b], c = (b + 1 < NB ? hist[b + 1] : n) - lo;
        u32 B = 0, BS = 0;
        for (u32 i = 0; i < nbig; ++i) { const bool before = bigb[i] < (u32)b; B += before ? 1u : 0u; BS += before ? bigc[i] : 0u; }
        const u32 pid = (lo - BS) / half + 2u * B + (c > half ? 1u : 0u);
        part_of[b] = (unsigned char)min(pid, (u32)OVRP_PMAX - 1);
        if (c) { pmax = max(pmax, pid); if (pid < (u32)OVRP_PMAX) atomicMin(&pstart[pid], lo); }
    }
    atomicMax(np_out, pmax);
    __syncthreads();
    const u32 top = *np_out;
    __syncthreads();
    if (top >= (u32)OVRP_PMAX) { if (tid == 0) *np_out = 0xFFFFFFFFu; __syncthreads(); return; }
    if (tid == 0) { // an unused id starts where the next part does
        for (int p = (int)top - 1; p >= 0; --p) pstart[p] = min(pstart[p], pstart[p + 1]);
        pstart[top + 1] = n;
        *np_out = n ? top + 1u : 0u;
    }
    __syncthreads();
}

// ---------------------------------------------------------------------------------------------------------------------
// Dense (gene-major key rows, group-contiguous positions: what k_transpose_permute writes): split a gene's non-zero keys
// by value into parts of at most `cap` keys.
// ---------------------------------------------------------------------------------------------------------------------
struct OvrPartParams {
    const void *Xt;            // [n_genes][stride] keys
    long long stride;
    int n_genes, n_cells;
    const int *code_by_pos;    // [n_cells]
    int cap;                   // keys per part at most
    void *out_keys;            // [n_genes][stride]
    u16 *out_codes;            // [n_genes][stride]
    u32 *part_start;           // [n_genes][OVRP_PMAX + 1]
    u32 *gene_info;            // [n_genes][4]: non-zeros, negatives, parts, flag
};

static inline size_t ovrp_lds_bytes() { return ((size_t)4 << OVRP_LG) + OVRP_NT * 4 + (2 * OVRP_PMAX + 4) * 4 + 16 + ((size_t)1 << OVRP_LG); }

template <typename KeyT>
__global__ __launch_bounds__(OVRP_NT) void k_ovr_partition(OvrPartParams P) {
    constexpr int NT = OVRP_NT, NB = 1 << OVRP_LG;
    constexpr KeyT ZEROK = KeyInfo<KeyT>::ZEROK;
    constexpr KeyT MAXK = KeyInfo<KeyT>::MAXK;
    extern __shared__ __align__(16) unsigned char smem[];
    u32 *hist = (u32 *)smem;                         // [NB] counts, then exclusive offsets in value order
    u32 *tmp = hist + NB;                            // [NT]
    u32 *pstart = tmp + NT;                          // [OVRP_PMAX + 1]
    u32 *pfill = pstart + OVRP_PMAX + 1;             // [OVRP_PMAX]
    u32 *s_mx = pfill + OVRP_PMAX;                   // [0] largest bucket  [1] negatives
    KeyT *s_k = (KeyT *)(s_mx + 2 + ((2 * OVRP_PMAX + 1) & 1)); // [2] smallest / largest non-zero key (8-byte aligned)
    unsigned char *part_of = (unsigned char *)(s_k + 2); // [NB]
    const int tid = threadIdx.x, lane = tid & 63;
    const int N = P.n_cells;
    const u64 lt_mask = (lane == 0) ? 0ull : (~0ull >> (64 - lane));
    constexpr int UL = 8;
    for (int gene = blockIdx.x; gene < P.n_genes; gene += gridDim.x) {
        const KeyT *row = (const KeyT *)P.Xt + (size_t)gene * P.stride;
        for (int b = tid; b < NB; b += NT) hist[b] = 0u;
        if (tid < OVRP_PMAX) pfill[tid] = 0u;
        if (tid == 0) { s_mx[0] = 0u; s_mx[1] = 0u; s_k[0] = MAXK; s_k[1] = (KeyT)0; }
        __syncthreads();
        { // key range for the bucket function, from every 8th row of NT keys (any monotone function partitions correctly:
          // keys outside the sampled range are clamped into the first / last bucket)
            KeyT tmin = MAXK, tmax = (KeyT)0;
            for (int i = tid; i < N; i += NT * 8) {
                const KeyT k = row[i];
                if (k != ZEROK) { tmin = k < tmin ? k : tmin; tmax = k > tmax ? k : tmax; }
            }
            tmin = wave_min_key(tmin);
            tmax = wave_max_key(tmax);
            if (lane == 0) { atomicMin(&s_k[0], tmin); atomicMax(&s_k[1], tmax); }
        }
        __syncthreads();
        const bool have_range = s_k[1] >= s_k[0];
        const KeyT kmin = have_range ? s_k[0] : (KeyT)0, kmax = have_range ? s_k[1] : (KeyT)0;
        const int shift = max(0, key_bits((KeyT)(kmax - kmin)) - OVRP_LG);
        auto bucket_of = [&](KeyT key) -> u32 {
            const KeyT d = key > kmin ? (KeyT)((KeyT)(key - kmin) >> shift) : (KeyT)0;
            return (u32)(d < (KeyT)(NB - 1) ? d : (KeyT)(NB - 1));
        };
        {
            u32 neg = 0;
            for (int i0 = 0; i0 < N; i0 += NT * UL) {
                KeyT k[UL];
#pragma unroll
                for (int u = 0; u < UL; ++u) { const int i = i0 + u * NT + tid; k[u] = i < N ? row[i] : ZEROK; }
#pragma unroll
                for (int u = 0; u < UL; ++u)
                    if (k[u] != ZEROK) {
                        atomicAdd(&hist[bucket_of(k[u])], 1u);
                        neg += k[u] < ZEROK ? 1u : 0u;
                    }
            }
            neg = (u32)wave_sum((int)neg);
            if (lane == 0 && neg) atomicAdd(&s_mx[1], neg);
        }
        __syncthreads();
        u32 mx = 0;
        for (int b = tid; b < NB; b += NT) mx = max(mx, hist[b]);
        mx = (u32)wave_incl_scan_max((int)mx);
        if (lane == 63) atomicMax(&s_mx[0], mx);
        __syncthreads();
        mx = s_mx[0];
        const u32 n = block_excl_scan_inplace<NT>(hist, NB, tmp, tid);
        const u32 nneg = s_mx[1];
        // part of a bucket = keys before it / quota; no bucket is larger than a quarter of cap, so every part gets a bucket
        // boundary and holds fewer than quota + mx = cap keys
        const bool skew = (unsigned long long)mx * 4ull > (unsigned long long)P.cap;
        const u32 quota = skew ? 1u : (u32)P.cap - mx;
        u32 n_parts = n ? (n - 1) / quota + 1 : 0u;
        if (skew) { // (uniform) crowded buckets get parts of their own: ovrp_assign_parts_skewed
            ovrp_assign_parts_skewed<NT>(hist, NB, n, (u32)P.cap, part_of, pstart, tmp, &s_mx[0], tid);
            n_parts = s_mx[0];
        }
        const bool bad = n_parts > (u32)OVRP_PMAX;
        u32 *gi = P.gene_info + (size_t)gene * 4;
        u32 *ps_out = P.part_start + (size_t)gene * (OVRP_PMAX + 1);
        if (bad) { // uniform: this gene goes to the general route
            if (tid == 0) { gi[0] = n; gi[1] = nneg; gi[2] = 0u; gi[3] = 1u; }
            __syncthreads();
            continue;
        }
        if (!skew) {
            for (int b = tid; b < NB; b += NT) {
                const u32 p = hist[b] / quota;
                part_of[b] = (unsigned char)min(p, (u32)OVRP_PMAX - 1); // (empty buckets past the last key may overshoot)
                if (p < n_parts && (b == 0 || hist[b - 1] / quota != p)) pstart[p] = hist[b];
            }
            if (tid == 0) pstart[n_parts] = n;
        }
        const int p_bits = 32 - __clz(n_parts); // bits of the values 0 .. n_parts
        __syncthreads();
        for (int i0 = 0; i0 < N; i0 += NT * UL) {
            KeyT k[UL];
            int cd[UL];
#pragma unroll
            for (int u = 0; u < UL; ++u) {
                const int i = i0 + u * NT + tid;
                k[u] = i < N ? row[i] : ZEROK;
                cd[u] = i < N ? P.code_by_pos[i] : 0;
            }
#pragma unroll
            for (int u = 0; u < UL; ++u) {
                const bool nz = k[u] != ZEROK;
                const int p = nz ? (int)part_of[bucket_of(k[u])] : (int)n_parts; // n_parts = not a record
                // lanes with the same part, without a loop over parts: match the bits of p through ballots (as many bits as
                // n_parts needs: 3 for the 6 parts of a half-empty 300 000-cell column)
                u64 m = ~0ull;
                for (int bit = 0; bit < p_bits; ++bit) {
                    const bool on = (p >> bit) & 1;
                    const u64 bl = __ballot(on);
                    m &= on ? bl : ~bl;
                }
                u32 slot = 0;
                if (nz) { // one LDS atomic per (wavefront, part present in it): the lowest lane of each match set
                    const int leader = __ffsll((long long)m) - 1;
                    u32 b0 = 0;
                    if (lane == leader) b0 = atomicAdd(&pfill[p], (u32)__popcll(m));
                    b0 = (u32)__shfl((int)b0, leader);
                    slot = pstart[p] + b0 + (u32)__popcll(m & lt_mask);
                }
                if (nz) {
                    const size_t o = (size_t)gene * P.stride + slot;
                    ((KeyT *)P.out_keys)[o] = k[u];
                    P.out_codes[o] = (u16)cd[u];
                }
            }
        }
        if (tid <= (int)n_parts) ps_out[tid] = pstart[tid];
        if (tid == 0) { gi[0] = n; gi[1] = nneg; gi[2] = n_parts; gi[3] = 0u; }
        __syncthreads();
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// The same split over the PACKED key rows of k_group_compact (kernels_ovo_compact.h): only the non-zero keys are there -- block
// after block of groups, blk_cnt[gene][block] keys from slot blk_out[block] on, a group's keys back to back with the next
// group's -- so the three walks read 45 % of a half-empty matrix instead of all of it.  One wavefront takes a block at a time;
// a key's group code is the block's first group plus the group ends (running sums of nnz[gene][group]) at or below the key's
// offset: a few compares against scalars.
// ---------------------------------------------------------------------------------------------------------------------
struct OvrPartPackedParams {
    const void *Xt;            // [n_genes][stride] packed keys
    long long stride;
    int n_genes, G, nblk;
    const u16 *nnz;            // [n_genes][G]
    const u32 *blk_cnt;        // [n_genes][nblk]
    const int *blk_g0, *blk_g1, *blk_out; // [nblk] first group, one past the last, first key slot
    int cap;
    void *out_keys;
    u16 *out_codes;
    u32 *part_start, *gene_info; // as OvrPartParams
    // the kernel's COOP form: some blocks are LONG (more than OVRP_LONG_ROWS rows and at most 64 groups: a cluster of thousands of cells, the
    // control group of a screen): their 512-key units are dealt over ALL the wavefronts (unit u of the i-th long block: wavefront (u + i) % NW)
    // instead of one wavefront walking the block alone; the other blocks stay with their wavefront
    int n_long;
    const int *long_blk;            // [n_long] the long blocks
    const unsigned char *blk_is_long; // [nblk]
};
#define OVRP_LONG_ROWS 4096

template <typename KeyT, bool COOP = false>
__global__ __launch_bounds__(OVRP_NT) void k_ovr_partition_packed(OvrPartPackedParams P) {
    constexpr int NT = OVRP_NT, NW = NT / 64, NB = 1 << OVRP_LG;
    constexpr KeyT ZEROK = KeyInfo<KeyT>::ZEROK;
    constexpr KeyT MAXK = KeyInfo<KeyT>::MAXK;
    extern __shared__ __align__(16) unsigned char smem[];
    u32 *hist = (u32 *)smem;                         // [NB] counts, then exclusive offsets in value order
    u32 *tmp = hist + NB;                            // [NT]
    u32 *pstart = tmp + NT;                          // [OVRP_PMAX + 1]
    u32 *pfill = pstart + OVRP_PMAX + 1;             // [OVRP_PMAX]
    u32 *s_mx = pfill + OVRP_PMAX;                   // [0] largest bucket  [1] negatives
    KeyT *s_k = (KeyT *)(s_mx + 2 + ((2 * OVRP_PMAX + 1) & 1)); // [2] smallest / largest non-zero key (8-byte aligned)
    unsigned char *part_of = (unsigned char *)(s_k + 2); // [NB]
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const u64 lt_mask = (lane == 0) ? 0ull : (~0ull >> (64 - lane));
    constexpr int UL = 8; // 64-key pieces of a block in flight per wavefront (a block of ~1024 rows holds ~500 keys)
    const int gene = blockIdx.x;
    const KeyT *row = (const KeyT *)P.Xt + (size_t)gene * P.stride;
    const u32 *bcnt = P.blk_cnt + (size_t)gene * P.nblk;
    const u16 *nnz = P.nnz + (size_t)gene * P.G;
    for (int b = tid; b < NB; b += NT) hist[b] = 0u;
    if (tid < OVRP_PMAX) pfill[tid] = 0u;
    if (tid == 0) { s_mx[0] = 0u; s_mx[1] = 0u; s_k[0] = MAXK; s_k[1] = (KeyT)0; }
    __syncthreads();
    { // key range for the bucket function, from every 8th block
        KeyT tmin = MAXK, tmax = (KeyT)0;
        if (COOP) { // (and every 8th 64-key piece of the long blocks)
            for (int i8 = wave; i8 < P.n_long; i8 += NW) {
                const int b = P.long_blk[i8], n_b = (int)bcnt[b];
                const KeyT *src = row + P.blk_out[b];
                for (int i = lane; i < n_b; i += 64 * 8) { const KeyT k = src[i]; tmin = k < tmin ? k : tmin; tmax = k > tmax ? k : tmax; }
            }
        }
        for (int b = wave * 8; b < P.nblk; b += NW * 8) {
            if (COOP && P.blk_is_long[b]) continue;
            const int n_b = (int)bcnt[b];
            const KeyT *src = row + P.blk_out[b];
            for (int i = lane; i < n_b; i += 64) { const KeyT k = src[i]; tmin = k < tmin ? k : tmin; tmax = k > tmax ? k : tmax; }
        }
        tmin = wave_min_key(tmin);
        tmax = wave_max_key(tmax);
        if (lane == 0) { atomicMin(&s_k[0], tmin); atomicMax(&s_k[1], tmax); }
    }
    __syncthreads();
    const bool have_range = s_k[1] >= s_k[0];
    const KeyT kmin = have_range ? s_k[0] : (KeyT)0, kmax = have_range ? s_k[1] : (KeyT)0;
    const int shift = max(0, key_bits((KeyT)(kmax - kmin)) - OVRP_LG);
    auto bucket_of = [&](KeyT key) -> u32 {
        const KeyT d = key > kmin ? (KeyT)((KeyT)(key - kmin) >> shift) : (KeyT)0;
        return (u32)(d < (KeyT)(NB - 1) ? d : (KeyT)(NB - 1));
    };
    {
        u32 neg = 0;
        // (COOP: the wavefront's own blocks, the long ones left out, then its units of the long blocks)
        const int n_own = (P.nblk - wave + NW - 1) / NW;
        for (int it = 0; it < n_own + (COOP ? P.n_long : 0); ++it) {
            const bool lng = COOP && it >= n_own;
            const int b = lng ? P.long_blk[it - n_own] : wave + it * NW;
            if (COOP && !lng && P.blk_is_long[b]) continue;
            const int n_b = (int)bcnt[b];
            const KeyT *src = row + P.blk_out[b];
            const int o_first = lng ? ((wave - (it - n_own)) & (NW - 1)) * (64 * UL) : 0, o_step = lng ? NW * 64 * UL : 64 * UL;
            for (int o = o_first; o < n_b; o += o_step) {
                KeyT k[UL];
#pragma unroll
                for (int u = 0; u < UL; ++u) { const int i = o + u * 64 + lane; k[u] = i < n_b ? src[i] : ZEROK; }
#pragma unroll
                for (int u = 0; u < UL; ++u)
                    if (k[u] != ZEROK) { // (packed keys are never the zero key: this is the "past the end" mark)
                        atomicAdd(&hist[bucket_of(k[u])], 1u);
                        neg += k[u] < ZEROK ? 1u : 0u;
                    }
            }
        }
        neg = (u32)wave_sum((int)neg);
        if (lane == 0 && neg) atomicAdd(&s_mx[1], neg);
    }
    __syncthreads();
    u32 mx = 0;
    for (int b = tid; b < NB; b += NT) mx = max(mx, hist[b]);
    mx = (u32)wave_incl_scan_max((int)mx);
    if (lane == 63) atomicMax(&s_mx[0], mx);
    __syncthreads();
    mx = s_mx[0];
    const u32 n = block_excl_scan_inplace<NT>(hist, NB, tmp, tid);
    const u32 nneg = s_mx[1];
    const bool skew = (unsigned long long)mx * 4ull > (unsigned long long)P.cap;
    const u32 quota = skew ? 1u : (u32)P.cap - mx;
    u32 n_parts = n ? (n - 1) / quota + 1 : 0u;
    if (skew) { // (uniform) crowded buckets get parts of their own: ovrp_assign_parts_skewed
        ovrp_assign_parts_skewed<NT>(hist, NB, n, (u32)P.cap, part_of, pstart, tmp, &s_mx[0], tid);
        n_parts = s_mx[0];
    }
    const bool bad = n_parts > (u32)OVRP_PMAX;
    u32 *gi = P.gene_info + (size_t)gene * 4;
    u32 *ps_out = P.part_start + (size_t)gene * (OVRP_PMAX + 1);
    if (bad) { // uniform: this gene goes to the general route
        if (tid == 0) { gi[0] = n; gi[1] = nneg; gi[2] = 0u; gi[3] = 1u; }
        return;
    }
    if (!skew) {
        for (int b = tid; b < NB; b += NT) {
            const u32 p = hist[b] / quota;
            part_of[b] = (unsigned char)min(p, (u32)OVRP_PMAX - 1);
            if (p < n_parts && (b == 0 || hist[b - 1] / quota != p)) pstart[p] = hist[b];
        }
        if (tid == 0) pstart[n_parts] = n;
    }
    const int p_bits = 32 - __clz(n_parts);
    __syncthreads();
    static_assert((NW & (NW - 1)) == 0, "units are dealt by (u + b) % NW");
    const int n_own_c = (P.nblk - wave + NW - 1) / NW;
    for (int it = 0; it < n_own_c + (COOP ? P.n_long : 0); ++it) {
        const bool split = COOP && it >= n_own_c; // a long block (at most 64 groups: its group ends sit in a register): this wavefront's units of it
        const int b = split ? P.long_blk[it - n_own_c] : wave + it * NW;
        if (COOP && !split && P.blk_is_long[b]) continue;
        const int n_b = (int)bcnt[b];
        const KeyT *src = row + P.blk_out[b];
        int gcur = P.blk_g0[b];
        const int glast = P.blk_g1[b];
        const int o_first = split ? ((wave - (it - n_own_c)) & (NW - 1)) * (64 * UL) : 0, o_step = split ? NW * 64 * UL : 64 * UL;
        if (COOP && o_first >= n_b) continue;
        int gend = gcur < glast ? (int)nnz[gcur] : 0; // offset (inside the block) where group gcur's keys end
        // A block of at most 64 groups (the usual case: ~7 groups of ~150 cells per 1024 rows) keeps its group ends in ONE register,
        // lane l = the offset where group g0 + l ends: the walks below then read a lane instead of loading nnz[g] from memory -- a
        // chain of dependent loads per 64-key piece otherwise.
        const int g0 = gcur, ng = glast - g0;
        const bool ends_in_lanes = ng <= 64;
        int endv = 0, gi = 0; // gi: gcur - g0
        // (a group of 65535 cells and more is the LAST of its block -- a block closes at 1024 rows -- and a block's last end is never looked
        //  at: the 16-bit lengths may saturate there)
        if (ends_in_lanes) endv = wave_incl_scan_add(lane < ng ? (int)nnz[g0 + lane] : 0);
        for (int o = o_first; o < n_b; o += o_step) {
            if (split) gi = min(ng - 1, (int)__popcll(__ballot(lane < ng && endv <= o))); // the groups that end at or before this unit's first key
            KeyT k[UL];
#pragma unroll
            for (int u = 0; u < UL; ++u) { const int i = o + u * 64 + lane; k[u] = i < n_b ? src[i] : ZEROK; }
#pragma unroll
            for (int u = 0; u < UL; ++u) {
                const int off = o + u * 64 + lane;
                if (o + u * 64 >= n_b) break; // uniform
                // group code: gcur + the group ends at or below this key's offset (empty groups repeat an end: skipped over)
                int cd = gcur;
                if (ends_in_lanes) { // uniform
                    const int piece_end = o + u * 64 + 64;
                    cd = g0 + gi;
                    int j = gi, e = __builtin_amdgcn_readlane(endv, __builtin_amdgcn_readfirstlane(gi));
                    while (e < piece_end && j + 1 < ng) { // uniform
                        cd += off >= e ? 1 : 0;
                        ++j;
                        e = __builtin_amdgcn_readlane(endv, __builtin_amdgcn_readfirstlane(j));
                    }
                    while (gi + 1 < ng && __builtin_amdgcn_readlane(endv, __builtin_amdgcn_readfirstlane(gi)) <= piece_end) ++gi;
                } else {
                    int g = gcur, e = gend;
                    while (e < o + u * 64 + 64 && g + 1 < glast) { // uniform
                        cd += off >= e ? 1 : 0;
                        ++g;
                        e += (int)nnz[g];
                    }
                    // the next piece starts at o + u * 64 + 64: advance past the groups that end at or before it
                    while (gend <= o + u * 64 + 64 && gcur + 1 < glast) { ++gcur; gend += (int)nnz[gcur]; }
                }
                const bool nz = k[u] != ZEROK;
                const int p = nz ? (int)part_of[bucket_of(k[u])] : (int)n_parts; // n_parts = not a record
                u64 m = ~0ull;
                for (int bit = 0; bit < p_bits; ++bit) {
                    const bool on = (p >> bit) & 1;
                    const u64 bl = __ballot(on);
                    m &= on ? bl : ~bl;
                }
                u32 slot = 0;
                if (nz) { // one LDS atomic per (wavefront, part present in it): the lowest lane of each match set
                    const int leader = __ffsll((long long)m) - 1;
                    u32 b0 = 0;
                    if (lane == leader) b0 = atomicAdd(&pfill[p], (u32)__popcll(m));
                    b0 = (u32)__shfl((int)b0, leader);
                    slot = pstart[p] + b0 + (u32)__popcll(m & lt_mask);
                }
                if (nz) {
                    const size_t oo = (size_t)gene * P.stride + slot;
                    ((KeyT *)P.out_keys)[oo] = k[u];
                    P.out_codes[oo] = (u16)cd;
                }
            }
        }
    }
    __syncthreads();
    if (tid <= (int)n_parts) ps_out[tid] = pstart[tid];
    if (tid == 0) { gi[0] = n; gi[1] = nneg; gi[2] = n_parts; gi[3] = 0u; }
}
