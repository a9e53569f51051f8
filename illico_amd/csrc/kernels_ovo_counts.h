// OVO ranking for count-valued genes: a gene whose values are all integers in [0, COUNTS_R) needs no
// sort.  The reference column becomes a histogram in LDS (cumA[v] = #ref < v, cntA[v] = #ref == v);
// each group is histogrammed by one wavefront (LDS integer atomics, zeros counted by ballot), and every
// occupied bin v with multiplicity tB contributes
//     S2  += tB * (2*cumA[v] + cntA[v])        tie += tB * (3*tA*(tA+tB) + tB^2 - 1),  tA = cntA[v]
// exactly as the sort path does for a run of equal keys (kernels_ovo.h) -- same integers, bit-exact.
// Group and reference sizes are unbounded here (nothing has to fit in registers or LDS but the tables).
//
// Genes are routed per gene: k_transpose_permute* (dense) and the sparse ingest kernels set
// gene_flags[g] != 0 for any gene with a value outside the table (negative, fractional, >= COUNTS_R,
// NaN); this kernel skips flagged genes and k_ovo_rank skips unflagged ones.
#pragma once
#include "common.h"
#include "kernels_ovo.h"

#define COUNTS_R 2048      // table size: values 0 .. COUNTS_R-1
#define COUNTS_NT 512

// integer value of a key known to encode an integer in [0, COUNTS_R)
__device__ __forceinline__ u32 count_of_key(u32 k, int dt) {
    return dt == DT_F32 ? (u32)f32_of_key(k) : (k ^ 0x80000000u);
}
__device__ __forceinline__ u32 count_of_key(u64 k, int dt) {
    return dt == DT_F64 ? (u32)f64_of_key(k) : (u32)(k ^ 0x8000000000000000ull);
}

template <typename KeyT>
__global__ __launch_bounds__(COUNTS_NT, 4) void k_ovo_counts(OvoParams P, const u32 *__restrict__ gene_flags) {
    constexpr int NT = COUNTS_NT, NW = NT / 64, R = COUNTS_R;
    __shared__ u32 cumA[R + 1];        // cumA[v] = # reference values < v (non-zeros only in the sparse layout)
    __shared__ u32 hB[NW][R / 2];      // per-wave group histogram, two 16-bit bins per word
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
        for (int i = tid; i < R / 2 * NW; i += NT) (&hB[0][0])[i] = 0;
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

        // ---- groups: one wavefront each ----
        for (int g0 = wave * 64; g0 < G; g0 += NW * 64) {
            TrReduce<u64> rS2, rTie, rSum;
            for (int j = 0; j < 64; ++j) {
                const int g = g0 + j;
                u64 S2 = 0, tie = 0, sum = 0;
                if (g < G && g != ref) {
                    const int n_g = P.counts[g];
                    long long bstart;
                    int nB;
                    if (sp) { bstart = sp[g]; nB = (int)(sp[g + 1] - sp[g]); }
                    else { bstart = (long long)gene * P.gene_stride + P.pos_ptr[g]; nB = n_g; }
                    const u32 zB = (u32)(n_g - nB);
                    const KeyT *seg = Xs + bstart;
                    u32 nzero = 0, vmax = 0;
                    for (int i0 = 0; i0 < nB; i0 += 256) { // 4 keys per lane per trip
                        u32 c[4];
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            int i = i0 + r * 64 + lane;
                            c[r] = i < nB ? count_of_key(seg[i], P.dt) : 0u;
                        }
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            int i = i0 + r * 64 + lane;
                            bool valid = i < nB;
                            nzero += (u32)__popcll(__ballot(valid && c[r] == 0));
                            if (valid && c[r] != 0) atomicAdd(&hw[c[r] >> 1], (c[r] & 1u) ? 0x10000u : 1u);
                            vmax = max(vmax, c[r]);
                        }
                    }
                    // wave max of the values seen (bins to visit)
                    vmax = (u32)wave_incl_scan_max((int)vmax);
                    vmax = (u32)__builtin_amdgcn_readlane((int)vmax, 63);
                    wave_lds_fence();
                    for (u32 w0 = 0; w0 * 2 <= vmax; w0 += 64) {
                        const u32 w = w0 + lane;
                        u32 word = (w * 2 <= vmax) ? hw[w] : 0u;
                        if (word) {
                            hw[w] = 0; // leave the table clean for the next group
#pragma unroll
                            for (int h = 0; h < 2; ++h) {
                                const u32 v = 2 * w + h;
                                const u64 b = h ? (word >> 16) : (word & 0xFFFFu);
                                if (b) {
                                    const u64 lt = (u64)cumA[v] + zA;        // implicit zeros of A rank below v > 0
                                    const u64 a = cumA[v + 1] - cumA[v];
                                    S2 += b * (2ull * lt + a);
                                    tie += b * (3ull * a * (a + b) + b * b - 1ull);
                                    sum += b * v;
                                }
                            }
                        }
                    }
                    wave_lds_fence();
                    if (lane == 0) {
                        // value 0: explicit zeros of B (dense layout) tie with the reference's explicit zeros;
                        // implicit zeros of both sides (sparse layout) form one block, as in kernels_ovo.h
                        const u64 b0 = nzero, a0 = cumA[1]; // cntA[0]
                        if (b0) {
                            S2 += b0 * a0; // lt = 0
                            tie += b0 * (3ull * a0 * (a0 + b0) + b0 * b0 - 1ull);
                        }
                        S2 += (u64)zB * zA;
                        const u64 t0 = (u64)zA + zB;
                        tie += T_A + (t0 * t0 * t0 - t0);
                    }
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
                    P.out_sum[o] = (double)refsum_i;
                } else {
                    P.out_2u[o] = 2ll * (long long)n_ref * (long long)P.counts[g] - (long long)rS2.result;
                    P.out_tie[o] = rTie.result;
                    P.out_sum[o] = (double)rSum.result;
                }
            }
        }
        __syncthreads();
    }
}
