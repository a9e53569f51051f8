// One-versus-rest (OVR) kernels.
//
// Replaces, on device, dense_ovr_mwu_kernel_over_contiguous_col_chunk (illico/ovr/dense_ovr.py:15-80)
// and sparse_ovr_mwu_kernel (illico/ovr/sparse_ovr.py:23-97): np.argsort of a whole gene column and
// _accumulate_group_ranksums_from_argsort (utils/ranking.py:7-49).
//
// One workgroup per gene: stable LSD radix sort of (key, group code) pairs through HBM ping-pong
// buffers (8-bit digits, passes whose digit is constant over the column are skipped), then two
// streaming sweeps over the sorted column.  For a run of equal keys occupying sorted slots [s, e)
// every member has 2*avg_rank = s + e + 1 (ranking.py:38: avg_rank = 0.5*(i+1+j)); the forward sweep
// adds s+1 and the backward sweep adds e into the member's group accumulator (LDS, 64-bit integer
// atomics), so rank sums are exact integers.  Implicit zeros (sparse inputs) form one analytic tie
// block between the negative and the positive keys (sparse_ovr.py:70-83).
#pragma once
#include "common.h"
#include "kernels_finalize.h"

struct OvrParams {
    void *keys_a;             // [n_genes rows]  in: unsorted keys (destroyed)
    void *keys_b;             // ping-pong
    u32 *vals_a;              // group code per key; may hold garbage when code_by_pos is given
    u32 *vals_b;
    const int *code_by_pos;   // dense layout: group code of position i (payload generated on the fly); else null
    const u32 *seg_ptr;       // sparse layout: [n_genes][G+1] offsets of each (gene, group) run; null => dense (gene*stride, n = N)
    long long stride;         // dense layout
    const int *pos_ptr;       // dense layout: [G+1] group positions, for the per-group sums
    const int *counts;        // [G]
    int G, n_genes, dt, is_log1p;
    long long n_cells;
    long long scan_len = 0;   // dense layout: keys per gene row to walk (0: n_cells); larger in the padded layout (holes hold zero keys)
    int ref;                  // OVO-through-global-sort mode: reference group code
    const u32 *gene_flags;    // OVO mode: process only genes whose flag is non-zero (nullptr = all)
    u64 *acc_global;          // ACCG: per-gene accumulators [n_genes][3*G] in HBM (group counts beyond what LDS holds)
    long long *out_2u;        // [n_genes][G]
    u64 *out_tie;             // [n_genes][G]
    int tie_f64;              // sparse OVR: out_tie = the bits of the float64 tie sum of the reference's sparse path (tie_f64_sparse)
    double *out_sum;          // [n_genes][G]
    // split form (MODE 1 / 2: the sort between them is rocPRIM's segmented radix sort)
    u32 *seg_begin, *seg_end; // [n_genes] element offsets of each gene's (key, code) pairs in the unsorted buffers
    int *seg_n;               // [n_genes] number of pairs
};

#define OVR_NT 1024
#define OVR_E 4

template <int NT> __device__ __forceinline__ int block_incl_scan_max(int x, int carry, int *wtot, int tid, int &prev_out) {
    // inclusive max-scan over the workgroup in thread order; prev_out = scan value of the previous thread (carry for tid 0)
    const int lane = tid & 63, wave = tid >> 6;
    int wi = x;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        int o = __shfl_up(wi, d);
        if (lane >= d) wi = max(wi, o);
    }
    if (lane == 63) wtot[wave] = wi;
    __syncthreads();
    int wp = carry;
    for (int w = 0; w < wave; ++w) wp = max(wp, wtot[w]);
    int up = __shfl_up(wi, 1);
    prev_out = lane > 0 ? max(up, wp) : wp;
    __syncthreads();
    return max(wi, wp);
}

// inclusive min-scan from the last thread towards the first; next_out = scan value of the NEXT thread (carry for the last)
template <int NT> __device__ __forceinline__ int block_incl_scan_min_rev2(int x, int carry, int *wtot, int tid, int &next_out) {
    constexpr int NW = NT / 64;
    const int lane = tid & 63, wave = tid >> 6;
    int wi = x;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        int o = __shfl_down(wi, d);
        if (lane + d < 64) wi = min(wi, o);
    }
    if (lane == 0) wtot[wave] = wi;
    __syncthreads();
    int wp = carry;
    for (int w = NW - 1; w > wave; --w) wp = min(wp, wtot[w]);
    int dn = __shfl_down(wi, 1);
    next_out = lane < 63 ? min(dn, wp) : wp;
    __syncthreads();
    return min(wi, wp);
}

template <int NT> __device__ __forceinline__ int block_incl_scan_min_rev(int x, int carry, int *wtot, int tid) {
    // inclusive min-scan from the last thread towards the first
    constexpr int NW = NT / 64;
    const int lane = tid & 63, wave = tid >> 6;
    int wi = x;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        int o = __shfl_down(wi, d);
        if (lane + d < 64) wi = min(wi, o);
    }
    if (lane == 0) wtot[wave] = wi;
    __syncthreads();
    int wp = carry;
    for (int w = NW - 1; w > wave; --w) wp = min(wp, wtot[w]);
    __syncthreads();
    return min(wi, wp);
}

// exclusive add-scan over the workgroup in thread order; `carry` (total of earlier chunks) is added and updated
template <int NT> __device__ __forceinline__ int block_excl_scan_add(int x, int &carry, int *wtot, int tid) {
    constexpr int NW = NT / 64;
    const int lane = tid & 63, wave = tid >> 6;
    int wi = x;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        int o = __shfl_up(wi, d);
        if (lane >= d) wi += o;
    }
    if (lane == 63) wtot[wave] = wi;
    __syncthreads();
    int before = carry, total = 0;
    for (int w = 0; w < NW; ++w) { if (w < wave) before += wtot[w]; total += wtot[w]; }
    __syncthreads();
    carry += total;
    return before + wi - x;
}
// same from the last thread towards the first: number of flagged threads strictly after this one (+ carry)
template <int NT> __device__ __forceinline__ int block_excl_scan_add_rev(int x, int &carry, int *wtot, int tid) {
    constexpr int NW = NT / 64;
    const int lane = tid & 63, wave = tid >> 6;
    int wi = x;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        int o = __shfl_down(wi, d);
        if (lane + d < 64) wi += o;
    }
    if (lane == 0) wtot[wave] = wi;
    __syncthreads();
    int after = carry, total = 0;
    for (int w = 0; w < NW; ++w) { if (w > wave) after += wtot[w]; total += wtot[w]; }
    __syncthreads();
    carry += total;
    return after + wi - x;
}

// ACCG: the per-group accumulators live in HBM (global 64-bit atomics) instead of LDS -- any number of groups.
// DC (dense layout, OVR): the gene's zeros are compacted away before the sort and ranked as one analytic tie block
// (exactly the sparse layout's semantics), so the sort passes and the sweeps touch only the non-zeros.
// MODE 0: everything in this kernel (per-group sums, zero compaction, LSD radix sort, rank sweeps).
// MODE 1: sums + compaction only, and the extent of the gene's pairs for the sort that follows.
// MODE 2: rank sweeps only, over pairs already sorted (into the B buffers for the compacted dense layout, else into A).
template <typename KeyT, bool SPARSE, bool OVO, int NT, bool ACCG, bool DC, int MODE = 0>
__global__ __launch_bounds__(NT) void k_ovr_gene(OvrParams P) {
    static_assert(MODE == 0 || !OVO, "the split form serves OVR");
    static_assert(!DC || (!SPARSE && !OVO), "zero compaction is the dense OVR variant");
    constexpr bool ZS = SPARSE || DC; // zeros are implicit during the sort and the sweeps
    constexpr int NW = NT / 64, E = (NT >= 1024 ? 4 : 8);
    constexpr KeyT ZEROK = KeyInfo<KeyT>::ZEROK;
    extern __shared__ __align__(16) unsigned char smem[];
    u32 *wcnt = (u32 *)smem;                 // [NW][256]
    u32 *hist = wcnt + NW * 256;             // [256]
    u32 *dbase = hist + 256;                 // [256]
    int *wtot = (int *)(dbase + 256);        // [NW]
    u64 *red = (u64 *)(wtot + NW + (NW & 1)); // [NW]
    int *flag = (int *)(red + NW);           // [4]
    u64 *R2 = (u64 *)(flag + 4);             // [G]  OVR: 2*ranksum;  OVO: S2
    u64 *tieg = R2 + P.G;                    // [G]  OVO only: per-group tie term
    u32 *gcnt = (u32 *)(OVO ? tieg + P.G : tieg); // [G] (SPARSE OVR only)
    (void)R2; (void)tieg; (void)gcnt;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int G = P.G;
    const u64 lt_mask = (lane == 0) ? 0ull : (~0ull >> (64 - lane));

    for (int gene = blockIdx.x; gene < P.n_genes; gene += gridDim.x) {
        if (OVO && P.gene_flags && P.gene_flags[gene] == 0) continue; // handled by the histogram route
        if (ACCG) { // this gene's accumulator block in HBM
            R2 = P.acc_global + (size_t)gene * 3 * P.G;
            tieg = R2 + P.G;
            gcnt = (u32 *)(R2 + 2 * (size_t)P.G);
        }
        long long start;
        int n;
        const u32 *sp = nullptr;
        if (SPARSE) { sp = P.seg_ptr + (size_t)gene * (G + 1); start = sp[0]; n = (int)(sp[G] - sp[0]); }
        else { start = (long long)gene * P.stride; n = (int)(P.scan_len ? P.scan_len : P.n_cells); }
        KeyT *ka = (KeyT *)P.keys_a + start, *kb = (KeyT *)P.keys_b + start;
        u32 *va = P.vals_a + start, *vb = P.vals_b + start;

        // ---- per-group sums of values for the fold change (deterministic order; runs are group-contiguous) ----
        // Four groups per wavefront and iteration, their first 256 keys each requested before any is summed: one load
        // latency per 4 groups instead of one per group (this loop, not the copy of the keys, set the time of the
        // prepare pass).  The order of additions inside a group does not depend on the batching.
        if constexpr (MODE != 2) {
            constexpr int GB = 4, RR = 4;
            for (int g0 = wave * GB; g0 < G; g0 += NW * GB) {
                int p0[GB], p1[GB];
                KeyT kk[GB][RR];
#pragma unroll
                for (int j = 0; j < GB; ++j) {
                    const int g = g0 + j;
                    p0[j] = p1[j] = 0;
                    if (g < G) {
                        if (SPARSE) { p0[j] = (int)(sp[g] - sp[0]); p1[j] = (int)(sp[g + 1] - sp[0]); }
                        else { p0[j] = P.pos_ptr[g]; p1[j] = P.pos_ptr[g + 1]; }
                    }
#pragma unroll
                    for (int r = 0; r < RR; ++r) {
                        const int i = p0[j] + r * 64 + lane;
                        kk[j][r] = i < p1[j] ? ka[i] : (KeyT)0;
                    }
                }
#pragma unroll
                for (int j = 0; j < GB; ++j) {
                    const int g = g0 + j;
                    if (g >= G) break;
                    double s = 0.0;
#pragma unroll
                    for (int r = 0; r < RR; ++r) {
                        const int i = p0[j] + r * 64 + lane;
                        if (i < p1[j]) s += P.is_log1p ? key_to_expm1(kk[j][r], P.dt) : key_to_double(kk[j][r], P.dt);
                    }
                    for (int i = p0[j] + RR * 64 + lane; i < p1[j]; i += 64) s += P.is_log1p ? key_to_expm1(ka[i], P.dt) : key_to_double(ka[i], P.dt);
                    s = wave_sum(s);
                    if (lane == 0 && P.out_sum) P.out_sum[(size_t)gene * G + g] = s;
                }
            }
        }
        __syncthreads();

        // ---- stable LSD radix sort of (key, group) ----
        KeyT *ksrc = ka, *kdst = kb;
        u32 *vsrc = va, *vdst = vb;
        bool have_vals = SPARSE;
        if constexpr (MODE == 2) { // pairs sorted by the library sort: dense-compacted B -> A, sparse A -> B
            n = P.seg_n[gene];
            if (DC) { ksrc = ka; vsrc = va; kdst = kb; vdst = vb; }
            else { ksrc = kb; vsrc = vb; kdst = ka; vdst = va; }
            have_vals = true;
        }
        if (DC && MODE != 2) { // (key, group code) of the non-zeros -> the other buffer, in any order (the sort follows)
            if (tid == 0) flag[3] = 0;
            __syncthreads();
            for (int i0 = 0; i0 < n; i0 += NT) {
                const int i = i0 + tid;
                const KeyT k = i < n ? ka[i] : ZEROK;
                const bool nz = k != ZEROK;
                const u64 m = __ballot(nz);
                int base = 0;
                if (lane == 0 && m) base = atomicAdd(&flag[3], (int)__popcll(m));
                base = __builtin_amdgcn_readfirstlane(base);
                if (nz) {
                    const int pos = base + (int)__popcll(m & lt_mask);
                    kb[pos] = k;
                    vb[pos] = (u32)P.code_by_pos[i];
                }
            }
            __threadfence_block();
            __syncthreads();
            n = flag[3];
            ksrc = kb; kdst = ka; vsrc = vb; vdst = va;
            have_vals = true;
            __syncthreads();
        }
        if constexpr (MODE == 1) {
            if (tid == 0) {
                const long long b = start; // the pairs sit at [start, start + n) of the source buffers
                P.seg_begin[gene] = (u32)b;
                P.seg_end[gene] = (u32)(b + n);
                P.seg_n[gene] = n;
            }
            __syncthreads();
            continue;
        }
        const long long n0 = P.n_cells - n; // implicit zeros
        if constexpr (MODE == 0)
        for (int shift = 0; shift < (int)sizeof(KeyT) * 8; shift += 8) {
            for (int i = tid; i < 256; i += NT) hist[i] = 0;
            if (tid == 0) flag[0] = 0;
            __syncthreads();
            for (int i0 = 0; i0 < n; i0 += NT) {
                int i = i0 + tid;
                bool valid = i < n;
                u32 d = valid ? (u32)((ksrc[i] >> shift) & 0xFF) : 0u;
                u64 peers = __ballot(valid);
#pragma unroll
                for (int b = 0; b < 8; ++b) {
                    u64 bal = __ballot((d >> b) & 1u);
                    peers &= ((d >> b) & 1u) ? bal : ~bal;
                }
                if (valid && (peers & lt_mask) == 0) atomicAdd(&hist[d], (u32)__popcll(peers));
            }
            __syncthreads();
            if (tid < 256 && hist[tid] == (u32)n) flag[0] = 1;
            __syncthreads();
            if (flag[0]) { __syncthreads(); continue; } // digit constant over the column: nothing to move
            if (tid < 256) {
                u32 acc = 0;
                for (int k = 0; k < tid; ++k) acc += hist[k];
                dbase[tid] = acc;
            }
            __syncthreads();
            for (int cbase = 0; cbase < n; cbase += NT * E) {
                for (int k = lane; k < 256; k += 64) wcnt[wave * 256 + k] = 0;
                wave_lds_fence();
                KeyT key[E];
                u32 val[E], off[E];
#pragma unroll
                for (int e = 0; e < E; ++e) {
                    int i = cbase + (wave * E + e) * 64 + lane;
                    bool valid = i < n;
                    key[e] = valid ? ksrc[i] : (KeyT)0;
                    val[e] = valid ? (have_vals ? vsrc[i] : (u32)P.code_by_pos[i]) : 0u;
                    u32 d = (u32)((key[e] >> shift) & 0xFF);
                    u64 peers = __ballot(valid);
#pragma unroll
                    for (int b = 0; b < 8; ++b) {
                        u64 bal = __ballot((d >> b) & 1u);
                        peers &= ((d >> b) & 1u) ? bal : ~bal;
                    }
                    u32 before = valid ? wcnt[wave * 256 + d] : 0u;
                    off[e] = before + (u32)__popcll(peers & lt_mask);
                    wave_lds_fence();
                    if (valid && (peers & lt_mask) == 0) wcnt[wave * 256 + d] = before + (u32)__popcll(peers);
                    wave_lds_fence();
                }
                __syncthreads();
                if (tid < 256) {
                    u32 run = dbase[tid];
                    for (int w = 0; w < NW; ++w) {
                        u32 cnt = wcnt[w * 256 + tid];
                        wcnt[w * 256 + tid] = run;
                        run += cnt;
                    }
                    dbase[tid] = run;
                }
                __syncthreads();
#pragma unroll
                for (int e = 0; e < E; ++e) {
                    int i = cbase + (wave * E + e) * 64 + lane;
                    if (i < n) {
                        u32 d = (u32)((key[e] >> shift) & 0xFF);
                        u32 dst = wcnt[wave * 256 + d] + off[e];
                        kdst[dst] = key[e];
                        vdst[dst] = val[e];
                    }
                }
                __syncthreads();
            }
            // make this pass's global writes visible to the whole workgroup before they are re-read
            __threadfence_block();
            __syncthreads();
            { KeyT *t = ksrc; ksrc = kdst; kdst = t; }
            { u32 *t = vsrc; vsrc = vdst; vdst = t; }
            have_vals = true;
        }
        if (!have_vals) { // every pass skipped (column constant): materialise the payload
            for (int i = tid; i < n; i += NT) vsrc[i] = (u32)P.code_by_pos[i];
            __threadfence_block();
            __syncthreads();
        }
        const KeyT *K = ksrc;
        const u32 *V = vsrc;

        if constexpr (OVO) {
            // ---- one-versus-reference from the globally sorted column (any group / reference size) ----
            // The pairs went into the stable sort group by group, so inside a run of equal keys the members are
            // ordered by group: every (value, group) sub-run is contiguous.  With R(i) = # reference cells at
            // sorted slots < i, a cell b of a value-run [s, e) has #ref<b = R(s), #ref==b = R(e)-R(s):
            //   S2[g]  += R(s) + R(e)                                  (forward sweep adds R(s), backward R(e))
            //   tie[g] += tB (3 a (a+tB) + tB^2 - 1), a = R(e)-R(s)    (at the tail of each (value, group) sub-run)
            //   T_A    += tA^3 - tA                                    (at the tail of each reference sub-run)
            u32 *tmpRs = vdst;            // free ping-pong buffers hold R(s) and the sub-run start
            u32 *tmpSS = (u32 *)kdst;
            const int ref = P.ref;
            const int n_ref = P.counts[ref];
            const int nA = SPARSE ? (int)(sp[ref + 1] - sp[ref]) : n_ref;
            const u64 zA = (u64)(n_ref - nA);
            for (int g = tid; g < G; g += NT) { R2[g] = 0; tieg[g] = 0; }
            __syncthreads();
            int carryR = 0, carryRs = 0, carrySS = 0;
            u64 nneg_l = 0;
            for (int cbase = 0; cbase < n; cbase += NT) {
                const int i = cbase + tid;
                const bool valid = i < n;
                const KeyT k = valid ? K[i] : (KeyT)0;
                const int gcode = valid ? (int)V[i] : -1;
                const bool isref = valid && gcode == ref;
                const int Rex = block_excl_scan_add<NT>(isref ? 1 : 0, carryR, wtot, tid);
                const bool hv = valid && (i == 0 || K[i - 1] != k);
                int dummy;
                const int Rs = block_incl_scan_max<NT>(hv ? Rex : -1, carryRs, wtot, tid, dummy);
                const bool hs = valid && (hv || (int)V[i - 1] != gcode);
                const int ss = block_incl_scan_max<NT>(hs ? i : -1, carrySS, wtot, tid, dummy);
                if (valid) {
                    tmpRs[i] = (u32)Rs;
                    tmpSS[i] = (u32)ss;
                    if (!isref) atomicAdd(&R2[gcode], (u64)Rs + ((SPARSE && k > ZEROK) ? 2ull * zA : 0ull));
                    else if (k < ZEROK) ++nneg_l;
                }
                const int last = min(n - 1 - cbase, NT - 1);
                if (tid == last) { flag[1] = Rs; flag[2] = ss; }
                __syncthreads();
                carryRs = flag[1];
                carrySS = flag[2];
                __syncthreads();
            }
            __threadfence_block();
            __syncthreads();
            int carrySuf = 0, carryRe = nA;
            u64 ta = 0;
            const int nchunks = (n + NT - 1) / NT;
            for (int c = nchunks - 1; c >= 0; --c) {
                const int i = c * NT + tid;
                const bool valid = i < n;
                const KeyT k = valid ? K[i] : (KeyT)0;
                const int gcode = valid ? (int)V[i] : -1;
                const bool isref = valid && gcode == ref;
                const int Rsuf = block_excl_scan_add_rev<NT>(isref ? 1 : 0, carrySuf, wtot, tid);
                const int Rin = nA - Rsuf; // # reference cells at slots <= i
                const bool tv = valid && (i == n - 1 || K[i + 1] != k);
                const int Re = block_incl_scan_min_rev<NT>(tv ? Rin : 0x7FFFFFFF, carryRe, wtot, tid);
                if (valid) {
                    if (!isref) atomicAdd(&R2[gcode], (u64)Re);
                    const bool ts = tv || (int)V[i + 1] != gcode;
                    if (ts) {
                        const u64 tB = (u64)(i - (int)tmpSS[i] + 1);
                        if (isref) ta += tB * tB * tB - tB;
                        else {
                            const u64 a = (u64)(Re - (int)tmpRs[i]);
                            atomicAdd(&tieg[gcode], tB * (3ull * a * (a + tB) + tB * tB - 1ull));
                        }
                    }
                }
                if (tid == 0) flag[1] = Re;
                __syncthreads();
                carryRe = flag[1];
                __syncthreads();
            }
            ta = wave_sum(ta);
            nneg_l = wave_sum(nneg_l);
            if (lane == 0) { red[wave] = ta; ((u64 *)wcnt)[wave] = nneg_l; }
            __syncthreads();
            u64 T_A = 0, nneg = 0;
            for (int w = 0; w < NW; ++w) { T_A += red[w]; nneg += ((u64 *)wcnt)[w]; }
            for (int g = tid; g < G; g += NT) {
                const size_t o = (size_t)gene * G + g;
                if (g == ref) { P.out_2u[o] = -2; P.out_tie[o] = 0; continue; }
                const long long n_g = P.counts[g];
                const u64 zB = SPARSE ? (u64)(n_g - (long long)(sp[g + 1] - sp[g])) : 0ull;
                const u64 s2 = (ACCG ? atomicAdd(&R2[g], 0ull) : R2[g]) + zB * (2ull * nneg + zA);
                const u64 t0 = zA + zB;
                P.out_2u[o] = 2ll * (long long)n_ref * n_g - (long long)s2;
                P.out_tie[o] = T_A + (ACCG ? atomicAdd(&tieg[g], 0ull) : tieg[g]) + (t0 * t0 * t0 - t0);
            }
            __syncthreads();
            continue;
        }
        // ---- sweeps over the sorted column ----
        for (int g = tid; g < G; g += NT) { R2[g] = 0; if (ZS) gcnt[g] = 0; }
        __syncthreads();
        // Each thread owns SE consecutive sorted slots per round: run heads / tails are resolved inside the thread
        // first, one block scan per round links the threads (8x fewer barriers than one slot per thread).
        constexpr int SE = 8;
        u64 tie = 0;
        int carry = 0; // run start of the last slot of the previous round
        for (int cbase = 0; cbase < n; cbase += NT * SE) { // forward: s(i)+1, run lengths
            const int i0 = cbase + tid * SE;
            KeyT k[SE];
            int sl[SE];
            bool hd[SE];
            KeyT prev = (i0 > 0 && i0 < n) ? K[i0 - 1] : (KeyT)0;
            int lh = -1;
#pragma unroll
            for (int e = 0; e < SE; ++e) {
                const int i = i0 + e;
                const bool valid = i < n;
                k[e] = valid ? K[i] : (KeyT)0;
                hd[e] = valid && (i == 0 || k[e] != prev);
                if (hd[e]) lh = i;
                sl[e] = lh;
                prev = k[e];
            }
            int cin; // run start inherited from the slots before this thread's range
            block_incl_scan_max<NT>(lh, carry, wtot, tid, cin);
            int s_last = cin;
#pragma unroll
            for (int e = 0; e < SE; ++e) {
                const int i = i0 + e;
                if (i < n) {
                    const int s = sl[e] >= 0 ? sl[e] : cin;
                    if (hd[e] && i > 0) { u64 t = (u64)(i - s_last); tie += t * t * t - t; }
                    s_last = s;
                    u64 add = (u64)s + 1ull + ((ZS && k[e] > ZEROK) ? 2ull * (u64)n0 : 0ull);
                    atomicAdd(&R2[V[i]], add);
                    if (ZS) atomicAdd(&gcnt[V[i]], 1u);
                }
            }
            const int last_i = min(n, cbase + NT * SE) - 1; // last valid slot of this round
            if (last_i >= i0 && last_i < i0 + SE) flag[1] = s_last;
            __syncthreads();
            carry = flag[1];
            __syncthreads();
        }
        if (tid == 0 && n > 0) { u64 t = (u64)(n - carry); tie += t * t * t - t; }
        int carry_e = n;
        const int nrounds = (n + NT * SE - 1) / (NT * SE);
        for (int c = nrounds - 1; c >= 0; --c) { // backward: e(i)
            const int i0 = c * NT * SE + tid * SE;
            KeyT k[SE];
            int el[SE];
#pragma unroll
            for (int e = 0; e < SE; ++e) k[e] = (i0 + e < n) ? K[i0 + e] : (KeyT)0;
            KeyT next = (i0 + SE < n) ? K[i0 + SE] : (KeyT)0;
            int le = 0x7FFFFFFF;
#pragma unroll
            for (int e = SE - 1; e >= 0; --e) {
                const int i = i0 + e;
                const bool valid = i < n;
                const bool tail = valid && (i == n - 1 || k[e] != next);
                if (tail) le = i + 1;
                el[e] = le;
                if (valid) next = k[e];
            }
            int cnext; // run end inherited from the slots after this thread's range
            block_incl_scan_min_rev2<NT>(le, carry_e, wtot, tid, cnext);
            int e_first = cnext;
#pragma unroll
            for (int e = SE - 1; e >= 0; --e) {
                const int i = i0 + e;
                if (i < n) {
                    const int ee = el[e] != 0x7FFFFFFF ? el[e] : cnext;
                    atomicAdd(&R2[V[i]], (u64)ee);
                    e_first = ee;
                }
            }
            if (tid == 0) flag[1] = e_first;
            __syncthreads();
            carry_e = flag[1];
            __syncthreads();
        }
        // tie sum: block reduce
        tie = wave_sum(tie);
        if (lane == 0) red[wave] = tie;
        if (tid == 0) { // number of negative keys (sparse layout only)
            int P0 = 0;
            if (ZS && n > 0) {
                int lo = 0, hi = n;
                while (lo < hi) { int mid = (lo + hi) >> 1; if (K[mid] < ZEROK) lo = mid + 1; else hi = mid; }
                P0 = lo;
            }
            flag[2] = P0;
        }
        __syncthreads();
        u64 tie_total = 0;
        for (int w = 0; w < NW; ++w) tie_total += red[w];
        if (P.tie_f64) tie_total = tie_f64_sparse(tie_total, (long long)n0);
        else tie_total += (u64)n0 * (u64)n0 * (u64)n0 - (u64)n0;
        const long long nneg = flag[2];
        for (int g = tid; g < G; g += NT) {
            long long n_g = P.counts[g];
            u64 r2 = ACCG ? atomicAdd(&R2[g], 0ull) : R2[g]; // HBM accumulators: read at L2, where the atomics landed
            if (ZS) {
                long long z = n_g - (long long)(ACCG ? atomicAdd(&gcnt[g], 0u) : gcnt[g]);
                r2 += (u64)z * (u64)(2 * nneg + n0 + 1);
            }
            long long two_u = 2ll * (P.n_cells - n_g) * n_g + n_g * (n_g + 1) - (long long)r2;
            P.out_2u[(size_t)gene * G + g] = two_u;
            P.out_tie[(size_t)gene * G + g] = tie_total;
        }
        __syncthreads();
    }
}

static inline size_t ovr_lds_bytes(int G, bool sparse /* or zero-compacted dense */, bool ovo, int nt, bool acc_global) {
    const int NW = nt / 64;
    size_t b = (size_t)NW * 256 * 4 + 256 * 4 + 256 * 4 + (NW + (NW & 1)) * 4 + NW * 8 + 16;
    if (!acc_global) b += (size_t)G * 8 + (ovo ? (size_t)G * 8 : 0) + (sparse ? (size_t)G * 4 : 0);
    return b;
}
