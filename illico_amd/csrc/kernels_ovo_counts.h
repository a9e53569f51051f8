// OVO ranking for count-valued genes: a gene whose values are all integers in [0, COUNTS_R) needs no
// sort.  The reference column becomes a histogram in LDS (cumA[v] = #ref < v, cntA[v] = #ref == v);
// each group is walked by one wavefront (lane = cell): a fetch-and-add on the wavefront's LDS counter table gives
// o = number of earlier cells of the group with the same value (zeros are ranked by ballot), and per cell
//     S2 += cumA[c] + cumA[c+1]                 TT += t (t+1),  t = cntA[c] + o      (tie term = 3 sum TT)
// -- the same integers as the run-based form of the sort path (kernels_ovo.h), bit-exact.
// Group and reference sizes are unbounded here (nothing has to fit in registers or LDS but the tables).
//
// Genes are routed per gene: k_transpose_permute* (dense) and the sparse ingest kernels set
// gene_flags[g] != 0 for any gene with a value outside the table (negative, fractional, >= COUNTS_R,
// NaN); this kernel skips flagged genes and k_ovo_rank skips unflagged ones.
#pragma once
#include "common.h"
#include "kernels_ovo.h"

#define COUNTS_R 2048      // table size: values 0 .. COUNTS_R-1 (16-bit running multiplicities: groups of any size up to 65535 cells)
#define COUNTS_R8 4096     // ... with 8-bit multiplicities (every ranked group at most 255 cells): twice the values in less LDS
#define COUNTS_NT 512

// integer value of a key known to encode an integer in [0, COUNTS_R)
__device__ __forceinline__ u32 count_of_key(u32 k, int dt) {
    return dt == DT_F32 ? (u32)f32_of_key(k) : (k ^ 0x80000000u);
}
__device__ __forceinline__ u32 count_of_key(u64 k, int dt) {
    return dt == DT_F64 ? (u32)f64_of_key(k) : (u32)(k ^ 0x8000000000000000ull);
}

// R: table size; CB: bits of a group's running multiplicity (8: four bins per word; 16: two)
template <typename KeyT, int R = COUNTS_R, int CB = 16>
__global__ __launch_bounds__(COUNTS_NT, CB == 8 ? 3 : 4) void k_ovo_counts(OvoParams P, const u32 *__restrict__ gene_flags) {
    constexpr int NT = COUNTS_NT, NW = NT / 64, PWB = 32 / CB, LGP = CB == 8 ? 2 : 1;
    __shared__ u32 cumA[R + 1];        // cumA[v] = # reference values < v (non-zeros only in the sparse layout)
    __shared__ u32 hB[NW][R / PWB];    // per-wave group histogram, PWB bins of CB bits per word
    __shared__ u64 s_red[NW];
    __shared__ u64 s_red2[NW];
    __shared__ u32 s_scan[NT];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const KeyT *Xs = (const KeyT *)P.Xs;
    const int G = P.G, ref = P.ref;
    const int n_ref = P.counts[ref];
    u32 *hw = hB[wave];

    for (int gene = blockIdx.x; gene < P.n_genes; gene += gridDim.x) {
        if (gene_flags[gene] != 0) continue; // uniform: not a count-valued gene
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

        // ---- reference histogram -> cumA ----
        for (int i = tid; i <= R; i += NT) cumA[i] = 0;
        for (int i = tid; i < R / PWB * NW; i += NT) (&hB[0][0])[i] = 0;
        __syncthreads();
        u64 rsum = 0;
        for (u32 i0 = 0; i0 < nA; i0 += NT) {
            u32 i = i0 + tid;
            bool valid = i < nA;
            u32 c = valid ? count_of_key(Xs[rstart + i], P.dt) : 0u;
            rsum += c;
            // zeros are the most common value: count them with a ballot instead of 64 same-address atomics
            u64 zb = __ballot(valid && c == 0);
            if (valid && c != 0) atomicAdd(&cumA[c + 1], 1u);
            if (lane == 0 && zb) atomicAdd(&cumA[1], (u32)__popcll(zb));
        }
        __syncthreads();
        // cumA[v+1] currently holds cntA[v]; T_A and the inclusive scan over R bins (NT threads x R/NT bins)
        constexpr int PER = R / NT;
        u32 loc[PER];
        u32 lsum = 0;
        u64 ta = 0;
#pragma unroll
        for (int e = 0; e < PER; ++e) {
            u32 c = cumA[1 + tid * PER + e];
            loc[e] = c;
            lsum += c;
            ta += (u64)c * c * c - c;
        }
        s_scan[tid] = lsum;
        rsum = wave_sum(rsum);
        ta = wave_sum(ta);
        if (lane == 0) { s_red[wave] = ta; s_red2[wave] = rsum; }
        __syncthreads();
        for (int d = 1; d < NT; d <<= 1) {
            u32 o = tid >= d ? s_scan[tid - d] : 0u;
            __syncthreads();
            s_scan[tid] += o;
            __syncthreads();
        }
        {
            u32 run = s_scan[tid] - lsum;
#pragma unroll
            for (int e = 0; e < PER; ++e) { run += loc[e]; cumA[1 + tid * PER + e] = run; }
        }
        u64 T_A = 0, refsum_i = 0;
        for (int w = 0; w < NW; ++w) { T_A += s_red[w]; refsum_i += s_red2[w]; }
        __syncthreads();
        // now cumA[v] = # reference values < v for v in [0, R], cntA[v] = cumA[v+1] - cumA[v]

        // ---- groups: one wavefront each; lane = cell.  Per cell with value c (a = cntA[c], o = earlier cells of
        // the group with the same value, from a fetch-and-add on the wavefront's counter table; zeros are ranked by
        // ballot instead of 64 same-address atomics):  S2 += cum[c] + cum[c+1],  TT += t (t+1), t = a + o.
        // (gridDim.y > 1: a gene's groups are dealt to that many workgroups, each with the reference tables of its own -- a few hundred genes
        //  are one workgroup per CU or less, and a workgroup walks 250 groups per wavefront one after the other)
        const int g_chunk = (((G + (int)gridDim.y - 1) / (int)gridDim.y) + 63) & ~63;
        const int g_lo = (int)blockIdx.y * g_chunk, g_hi = min(G, g_lo + g_chunk);
        for (int g0 = g_lo + wave * 64; g0 < g_hi; g0 += NW * 64) {
            TrReduce<u64> rS2, rTie, rSum;
            // next group's first 256 keys are fetched while the current group is processed
            KeyT nxt[4];
            int nB_n = 0, zB_n = 0;
            long long bs_n = 0;
            auto fetch = [&](int g) {
                nB_n = 0; zB_n = 0; bs_n = 0;
#pragma unroll
                for (int r = 0; r < 4; ++r) nxt[r] = (KeyT)0;
                if (g < G && g != ref) {
                    const int n_g = P.counts[g];
                    if (sp) { bs_n = sp[g]; nB_n = (int)(sp[g + 1] - sp[g]); }
                    else { bs_n = (long long)gene * P.gene_stride + P.pos_ptr[g]; nB_n = n_g; }
                    zB_n = n_g - nB_n;
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int i = r * 64 + lane;
                        if (i < nB_n) nxt[r] = Xs[bs_n + i];
                    }
                }
            };
            fetch(g0);
            const u64 lt_mask = (lane == 0) ? 0ull : (~0ull >> (64 - lane));
            for (int j = 0; j < 64; ++j) {
                const int g = g0 + j;
                KeyT cur[4];
#pragma unroll
                for (int r = 0; r < 4; ++r) cur[r] = nxt[r];
                const int nB = nB_n;
                const u32 zB = (u32)zB_n;
                const long long bstart = bs_n;
                fetch(g + 1 < g0 + 64 ? g + 1 : G);
                u64 S2 = 0, TT = 0, sum = 0;
                if (g < G && g != ref) {
                    const u64 a0 = cumA[1]; // cntA[0]
                    u32 zseen = 0;          // zeros of this group seen so far (wave-uniform)
                    for (int i0 = 0; i0 < nB; i0 += 256) {
                        u32 c[4];
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            const int i = i0 + r * 64 + lane;
                            const KeyT k = (i0 == 0) ? cur[r] : (i < nB ? Xs[bstart + i] : (KeyT)0);
                            c[r] = i < nB ? count_of_key(k, P.dt) : 0u;
                        }
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            const int i = i0 + r * 64 + lane;
                            const bool valid = i < nB;
                            const u64 zb = __ballot(valid && c[r] == 0);
                            if (valid) {
                                u32 o, a, lo, hi;
                                if (c[r] == 0) {
                                    o = zseen + (u32)__popcll(zb & lt_mask);
                                    lo = 0; hi = (u32)a0;
                                } else {
                                    const u32 sh = (c[r] & (u32)(PWB - 1)) * CB;
                                    o = __builtin_amdgcn_ubfe(atomicAdd(&hw[c[r] >> LGP], 1u << sh), sh, CB);
                                    lo = cumA[c[r]]; hi = cumA[c[r] + 1];
                                }
                                a = hi - lo;
                                const u64 t = (u64)a + o;
                                S2 += (u64)lo + hi + 2ull * (c[r] ? zA : 0u); // implicit zeros of A rank below c > 0
                                TT += t * (t + 1ull);
                                sum += c[r];
                            }
                            zseen += (u32)__popcll(zb);
                        }
                    }
                    wave_lds_fence();
                    // wipe the counter words the group touched (plain stores; LDS is in order within a wavefront)
                    for (int i0 = 0; i0 < nB; i0 += 256) {
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            const int i = i0 + r * 64 + lane;
                            if (i < nB) {
                                const KeyT k = (i0 == 0) ? cur[r] : Xs[bstart + i];
                                const u32 cc = count_of_key(k, P.dt);
                                if (cc) hw[cc >> LGP] = 0u;
                            }
                        }
                    }
                    wave_lds_fence();
                    if (lane == 0) {
                        // implicit zeros of both sides (sparse layout) form one block, as in kernels_ovo.h
                        S2 += (u64)zB * zA;
                    }
                }
                u64 tie = 3ull * TT;
                if (lane == 0 && g < G && g != ref) {
                    const u64 t0 = (u64)zA + zB;
                    tie += T_A + (t0 * t0 * t0 - t0);
                }
                rS2.push(S2, j, lane);
                rTie.push(tie, j, lane);
                rSum.push(sum, j, lane);
            }
            const int g = g0 + lane;
            if (g < G) {
                size_t o = (size_t)gene * G + g;
                if (g == ref) {
                    P.out_2u[o] = -2;
                    P.out_tie[o] = 0;
                    if (P.out_sum) P.out_sum[o] = (double)refsum_i;
                } else {
                    P.out_2u[o] = 2ll * (long long)n_ref * (long long)P.counts[g] - (long long)rS2.result;
                    P.out_tie[o] = rTie.result;
                    if (P.out_sum) P.out_sum[o] = (double)rSum.result;
                }
            }
        }
        __syncthreads();
    }
}
