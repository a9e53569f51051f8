// Dense one-versus-rest, any values: the rank kernel over value-range parts (k_ovr_partition / k_ovr_partition_packed write the
// parts: (key, group code) records, part after part, each part at most key_cap records).
//
// One workgroup takes a GENE at a time (drawn from a queue) and walks its parts in value order; the per-group accumulators stay in
// LDS for the whole gene and the statistics leave once, as plain stores -- no per-part atomics into HBM, no finishing kernel.
// Per part, the phases of k_csc_ovr_gene (kernels_csc_ovr.h) over the part's records:
//
//   key range (sampled) -> bucket function (key - kmin) >> shift, 2^lg - 1 buckets of 16-bit counters
//   count   one LDS atomic per key
//   scan    exclusive offsets (three barriers: 16 counters per thread, a wavefront scan, the wavefronts' totals)
//   scatter one returning LDS atomic + one LDS store per key: the keys, bucket after bucket
//   rank    per key: its bucket's bounds (two neighbouring table entries), the first four keys of the bucket in straight-line code
//           (average bucket: two keys), the rare longer bucket in a loop; s = base + lo + #smaller, e = s + #equal,
//           acc[group] += 2 s + t + 1 (+ 2 n0 for positive keys), tie += t^2 - 1
//
// Heavy ties (a crowded bucket) take the sorted form of the same part: keys sorted in LDS, two binary searches per key.
// Measured and not kept (profiles/NOTES_r03.md): the part's records held in registers across the phases (28 - 64 per thread, fully
// unrolled): the kernel becomes instruction-fetch bound (every instruction runs once per part) -- 12.8 ms against 8.3 ms.
//
// Device counterpart of illico/ovr/dense_ovr.py:46-75 + _accumulate_group_ranksums_from_argsort (illico/utils/ranking.py:7-49).
#pragma once
#include "common.h"
#include "kernels_csc_ovr.h"

struct OvrRankGeneParams {
    const void *pkeys;        // [nb][pstride] non-zero keys, part after part
    const u16 *pcodes;        // [nb][pstride] their group codes
    long long pstride;
    const u32 *part_start;    // [nb][OVRP_PMAX + 1] first record of each part, relative to the gene
    const u32 *gene_info;     // [nb][4] non-zeros, negatives, parts, flag (1 = the gene left this route)
    u32 *gflag;               // [nb] set to 1 when a part cannot be ranked here
    u32 *gene_counter;        // work queue head (zeroed by the host)
    int nb, G;
    const int *counts;        // [G]
    long long n_cells;
    int key_cap, lg_buckets, force_sorted;
    long long *out_2u;        // [nb][G]
    u64 *out_tie;             // [nb][G]
};

template <typename KeyT, int NT_ = CSCO_NT>
__global__ __launch_bounds__(NT_) void k_ovr_rank_gene_parts(OvrRankGeneParams P) {
    constexpr int NT = NT_, NW = NT / 64, CH = 64 * CSCO_K, UL = 8;
    constexpr KeyT ZEROK = KeyInfo<KeyT>::ZEROK;
    constexpr KeyT MAXK = KeyInfo<KeyT>::MAXK;
    constexpr u64 CNT1 = 1ull << CSCO_CNT_SHIFT, R2MASK = CNT1 - 1ull;
    extern __shared__ __align__(16) unsigned char smem[];
    const int G = P.G, NBKT = 1 << P.lg_buckets;
    const size_t accb = (size_t)((G + 1) & ~1) * 8;
    u64 *acc = (u64 *)smem;                                   // [G] doubled rank sum | stored non-zeros << 40
    u32 *tab = (u32 *)(smem + accb);                          // [NBKT / 2] two 16-bit bucket counters / offsets per word
    u16 *tab16 = (u16 *)tab;
    u64 *s_red = (u64 *)(tab + NBKT / 2);                     // [NW]
    KeyT *s_k = (KeyT *)(s_red + NW);                         // [2] smallest / largest sampled key of the part
    u32 *s_misc = (u32 *)(s_red + NW + 2);                    // [2] largest bucket  [3] queue slot
    KeyT *A = (KeyT *)(smem + csco_fixed_lds_bytes(G, P.lg_buckets, true));
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    typedef OvrSource<KeyT, int, KeyT, true> Src;
    Src src;
    src.pkeys = (const KeyT *)P.pkeys; src.pcodes = P.pcodes;

    for (;;) {
        if (tid == 0) s_misc[3] = atomicAdd(P.gene_counter, 1u);
        __syncthreads();
        const int gene = (int)s_misc[3];
        __syncthreads();
        if (gene >= P.nb) break; // every wavefront leaves here
        const u32 *gi = P.gene_info + (size_t)gene * 4;
        if (gi[3] != 0u) continue; // uniform: the partition sent this gene to the general route
        const int n_parts = (int)gi[2];
        const long long n0 = P.n_cells - (long long)gi[0], nneg = gi[1];
        const u32 *ps = P.part_start + (size_t)gene * (OVRP_PMAX + 1);
        for (int g = tid; g < G; g += NT) acc[g] = 0ull;
        u64 tie = 0;
        bool bad = false;
        for (int part = 0; part < n_parts; ++part) {
            const u32 base = ps[part];
            const int n = (int)(ps[part + 1] - base);
            if (n == 0) continue;
            if (n > (P.key_cap & ~1023)) { // (uniform; the sorted form below rounds a part up to 1024 keys) a part beyond the key slots: a crowded coarse bucket that the partition gave a part of its own
                // (ovrp_assign_parts_skewed).  ONE value -- the ties of log1p'd or scaled counts -- needs no ranking: every record stands
                // at base .. base + n - 1 with n equals; the records are only counted per group.  Different values: the general route's gene.
                if (tid == 0) { s_k[0] = MAXK; s_k[1] = (KeyT)0; }
                __syncthreads();
                const long long k0 = (long long)gene * P.pstride + base, k1 = k0 + n;
                KeyT tmin = MAXK, tmax = (KeyT)0;
                for (long long k = k0 + tid; k < k1; k += NT) {
                    const KeyT key = src.pkeys[k];
                    tmin = key < tmin ? key : tmin;
                    tmax = key > tmax ? key : tmax;
                }
                tmin = wave_min_key(tmin);
                tmax = wave_max_key(tmax);
                if (lane == 0) { atomicMin(&s_k[0], tmin); atomicMax(&s_k[1], tmax); }
                __syncthreads();
                const KeyT q = s_k[0];
                const bool one_value = s_k[1] == q;
                __syncthreads();
                if (!one_value) { bad = true; break; }
                const u64 add = 2ull * (u64)base + 1ull + CNT1 + ((q > ZEROK) ? 2ull * (u64)n0 : 0ull) + (u64)n;
                for (long long k = k0 + tid; k < k1; k += NT) atomicAdd(&acc[src.pcodes[k]], add);
                if (tid == 0) tie += (u64)n * ((u64)n * (u64)n - 1ull);
                __syncthreads();
                continue;
            }
            const long long k0 = (long long)gene * P.pstride + base, k1 = k0 + n;
            for (int b = tid; b < NBKT / 2; b += NT) tab[b] = 0u;
            if (tid == 0) { s_k[0] = MAXK; s_k[1] = (KeyT)0; s_misc[2] = 0u; }
            __syncthreads();
            { // key range of the bucket function, from every 8th row of NT records (any monotone function ranks correctly: keys
              // outside the sampled range are clamped into the first / last bucket; the range only balances the buckets)
                KeyT tmin = MAXK, tmax = (KeyT)0;
                const int row_step = n > 8 * NT ? 8 : 1;
                for (long long k = k0 + tid; k < k1; k += (long long)NT * row_step) {
                    const KeyT key = src.pkeys[k];
                    tmin = key < tmin ? key : tmin;
                    tmax = key > tmax ? key : tmax;
                }
                tmin = wave_min_key(tmin);
                tmax = wave_max_key(tmax);
                if (lane == 0) { atomicMin(&s_k[0], tmin); atomicMax(&s_k[1], tmax); }
            }
            __syncthreads();
            const KeyT kmin = s_k[0], kmax = s_k[1] >= s_k[0] ? s_k[1] : s_k[0];
            const int shift = max(0, key_bits((KeyT)(kmax - kmin)) - P.lg_buckets);
            // table entry of a key: 1 + its bucket (buckets 0 .. NBKT - 2).  Entry 0 stays 0, so that after the scan and the scattering
            // pass bucket b is [entry b, entry b + 1): two neighbouring 16-bit reads, no special case for the first bucket.
            const KeyT last_bucket = (KeyT)(NBKT - 2);
            auto entry_of = [&](KeyT key) -> u32 {
                const KeyT d = key > kmin ? (KeyT)((KeyT)(key - kmin) >> shift) : (KeyT)0;
                return (u32)(d < last_bucket ? d : last_bucket) + 1u;
            };
            bool sorted_form = P.force_sorted != 0;
            if (!sorted_form) {
                // ---- bucket sizes; how crowded are they? ----
                ovr_for_entries<false, NT, UL, Src, KeyT>(src, k0, k1, tid, [&](int, long long, KeyT key, bool, int) {
                    const u32 e = entry_of(key);
                    atomicAdd(&tab[e >> 1], (e & 1u) ? 0x10000u : 1u); // no carry: a counter stays below 2^16
                });
                __syncthreads();
                u64 sq = 0;
                u32 mx = 0;
                for (int b = tid; b < NBKT / 2; b += NT) {
                    const u32 x = tab[b], c0 = x & 0xFFFFu, c1 = x >> 16;
                    sq += (u64)c0 * c0 + (u64)c1 * c1;
                    mx = max(mx, max(c0, c1));
                }
                sq = wave_sum(sq);
                mx = (u32)wave_incl_scan_max((int)mx);
                if (lane == 63) { s_red[wave] = sq; atomicMax(&s_misc[2], mx); }
                __syncthreads();
                u64 sumsq = 0;
                for (int w = 0; w < NW; ++w) sumsq += s_red[w];
                sorted_form = s_misc[2] > (u32)CSCO_MAX_BUCKET || sumsq > (u64)CSCO_MAX_AVG * (u64)n; // uniform
                __syncthreads();
            }
            if (!sorted_form) {
                // ---- bucket offsets, keys into their buckets ----
                block_excl_scan_u16_waves<NT>(tab16, NBKT, (u32 *)s_red, tid); // (s_red: NW 64-bit slots, the scan wants NW words)
                ovr_for_entries<false, NT, UL, Src, KeyT>(src, k0, k1, tid, [&](int, long long, KeyT key, bool, int) {
                    const u32 e = entry_of(key);
                    const u32 old = atomicAdd(&tab[e >> 1], (e & 1u) ? 0x10000u : 1u);
                    A[(e & 1u) ? (old >> 16) : (old & 0xFFFFu)] = key;
                });
                if (tid < 4) A[n + tid] = MAXK; // the window below reads up to 3 keys past a bucket's end
                __syncthreads(); // now bucket b = [tab16[b], tab16[b + 1])
                // ---- every key against its bucket: the first four keys in straight-line code (the average bucket holds two); keys
                // past a bucket's end belong to later buckets (larger than q) or are the MAXK pad and count for neither sum ----
                const u64 c_neg = 2ull * (u64)base + 1ull + CNT1, c_pos = c_neg + 2ull * (u64)n0;
                u32 tie32 = 0; // a thread's share of a part's tie sum: at most 32 keys x 192^2
                ovr_for_entries<true, NT, UL, Src, KeyT>(src, k0, k1, tid, [&](int, long long, KeyT q, bool, int cd) {
                    const u32 e = entry_of(q);
                    const u32 lo = tab16[e - 1], hi = tab16[e];
                    const KeyT a0 = A[lo], a1 = A[lo + 1], a2 = A[lo + 2], a3 = A[lo + 3];
                    u32 less = (a0 < q ? 1u : 0u) + (a1 < q ? 1u : 0u) + (a2 < q ? 1u : 0u) + (a3 < q ? 1u : 0u);
                    u32 eq = (a0 == q ? 1u : 0u) + (a1 == q ? 1u : 0u) + (a2 == q ? 1u : 0u) + (a3 == q ? 1u : 0u);
                    for (u32 j = lo + 4; j < hi; j += 4) { // rare
                        const KeyT c0 = A[j], c1 = A[j + 1], c2 = A[j + 2], c3 = A[j + 3];
                        less += (c0 < q ? 1u : 0u) + (c1 < q ? 1u : 0u) + (c2 < q ? 1u : 0u) + (c3 < q ? 1u : 0u);
                        eq += (c0 == q ? 1u : 0u) + (c1 == q ? 1u : 0u) + (c2 == q ? 1u : 0u) + (c3 == q ? 1u : 0u);
                    }
                    if (q == MAXK) eq = (hi - lo) - less; // the largest key also matches the pad slots
                    atomicAdd(&acc[cd], ((q > ZEROK) ? c_pos : c_neg) + (u64)(2u * (lo + less) + eq));
                    tie32 += eq * eq - 1u;
                });
                tie += (u64)tie32;
            } else {
                // ---- sorted form: keys -> LDS, sort, tie blocks, two look-ups per key ----
                const int ncap = (n + CH - 1) / CH * CH;
                if (ncap > P.key_cap) { bad = true; break; } // uniform
                for (int i = n + tid; i < ncap; i += NT) A[i] = MAXK;
                ovr_for_entries<false, NT, UL, Src, KeyT>(src, k0, k1, tid, [&](int, long long k, KeyT key, bool, int) { A[k - k0] = key; });
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
                ovr_for_entries<true, NT, UL, Src, KeyT>(src, k0, k1, tid, [&](int, long long, KeyT q, bool, int cd) {
                    const u32 s = lower_bound_pow2(A, un, top, q);
                    u32 e = s + 1;
                    if (e < un && A[e] == q) e = upper_bound_pow2(A, un, top, q);
                    const u64 add = 2ull * (u64)base + (u64)s + (u64)e + 1ull + ((q > ZEROK) ? 2ull * (u64)n0 : 0ull);
                    atomicAdd(&acc[cd], add + CNT1);
                });
            }
            __syncthreads(); // the table and the key buffer are free for the next part
        }
        if (bad) { // uniform
            if (tid == 0) P.gflag[gene] = 1u;
            __syncthreads();
            continue;
        }
        tie = wave_sum(tie);
        if (lane == 0) s_red[wave] = tie;
        __syncthreads();
        u64 tie_total = (u64)n0 * (u64)n0 * (u64)n0 - (u64)n0;
        for (int w = 0; w < NW; ++w) tie_total += s_red[w];
        // the zeros of a group rank at n_neg + (n0 + 1) / 2 (sparse_ovr.py:70-83 restated for a dense column); dense_ovr.py:57-61
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
