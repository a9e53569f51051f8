// Count-valued dense input with FEW, LARGE groups (rank_genes_groups on the clusters of an atlas: ten groups of 100 000 cells): the
// fused kernels give a wavefront one group at a time -- 38 tiles x 10 groups are 114 workgroups for 9.6 GB, 9 ms -- because the tie
// term of OVO is accumulated cell by cell (running multiplicities).  But like the rank sums it is a function of the (group, gene) value
// HISTOGRAM alone:
//     2 U-part   S2  = sum_c h[c] (cum[c] + cum[c+1])                       (cum: cumulative counts of the reference group / the column)
//     tie part   TT  = sum_c sum_{o < h[c]} (a_c + o)(a_c + o + 1) = sum_c F(a_c + h[c] - 1) - F(a_c - 1),  F(t) = t (t+1) (t+2) / 3
//     value sum       = sum_c c h[c]
// so the rows of a group can be split over as many wavefronts as the launch needs: each counts its stretch of positions into a
// lane-private LDS histogram and adds it to H[group][tile][value][lane] (global integer atomics, value-major: coalesced); the tables
// come from H (the reference group's histogram, or the sum over groups: k_fused_tables_all as for the other forms); one wavefront per
// (group, tile) then evaluates the same integers and the same p-value code as k_ovo_fused.  Bit-identical to it.
// Reference: illico/ovo/dense_ovo.py:70-132, illico/ovr/dense_ovr.py:46-75, illico/utils/ranking.py:31-47, 52-158.
#pragma once
#include "kernels_ovo_fused.h"

#define GH_NT 256

template <typename InT, int RT, int UU, bool PRED, int NV>
__device__ __forceinline__ void consume_hist(const InT (&v)[NV], int p, int p1, unsigned short *cl, bool &inexact) {
#pragma unroll
    for (int u = 0; u < UU; ++u) {
        bool exact;
        const u32 c = clamp_count<InT, RT>(v[u], exact);
        const bool valid = !PRED || (p + u < p1); // wave-uniform
        inexact |= valid && !exact;
        cl[c * 64] = (unsigned short)(cl[c * 64] + (valid ? 1u : 0u));
    }
}

// grid (tiles, stretches of NW * wave_rows positions); a wavefront takes wave_rows consecutive positions (<= 65535: 16-bit cells) of the
// group-contiguous order, group after group: a flush per group it meets.
template <typename InT, int RT>
__global__ __launch_bounds__(GH_NT) void k_group_value_hists(FusedParams P, u32 *__restrict__ H, int wave_rows) {
    constexpr int NW = GH_NT / 64, U = FUSED_U;
    __shared__ unsigned short cells[NW][RT * 64];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int tile = blockIdx.x, tiles = gridDim.x, gene0 = tile * 64, gene = gene0 + lane;
    const bool act = gene < P.ncols;
    if (__all(!act || P.gene_flags[gene] != 0u)) return; // (per wavefront; no barrier below) every gene of the tile already left this route
    const int N = (int)P.n_cells;
    const int q0 = ((int)blockIdx.y * NW + wave) * wave_rows, q1 = min(N, q0 + wave_rows);
    if (q0 >= N) return;
    unsigned short *cl = &cells[wave][lane];
    for (int c = 0; c < RT; ++c) cl[c * 64] = 0;
    const int lane_c = act ? lane : 0;
    const char *Xb = (const char *)((const InT *)P.X + P.col0 + gene0);
    const u32 row_bytes = (u32)P.ld * (u32)sizeof(InT), col_bytes = (u32)lane_c * (u32)sizeof(InT);
    const const_int_p permc = (const_int_p)P.perm;
    int g = 0;
    { // the group that holds position q0: the last g with pos_ptr[g] <= q0 (uniform)
        int lo = 0, hi = P.G;
        while (hi - lo > 1) { const int mid = (lo + hi) >> 1; if (P.pos_ptr[mid] <= q0) lo = mid; else hi = mid; }
        g = lo;
    }
    bool bad = false;
    int p = q0;
    while (p < q1) { // (uniform)
        while (P.pos_ptr[g + 1] <= p) ++g; // (groups without cells)
        const int e = min(q1, __builtin_amdgcn_readfirstlane(P.pos_ptr[g + 1]));
        InT v[U];
        auto chunk = [&](auto uu, auto pred) {
            constexpr int UU = decltype(uu)::value;
            constexpr bool PRED = decltype(pred)::value;
            gather_rows<InT, UU, PRED>(Xb, row_bytes, permc, p, e, col_bytes, v);
            consume_hist<InT, RT, UU, PRED>(v, p, e, cl, bad);
            p += UU;
        };
        typedef std::integral_constant<bool, false> full_t;
        typedef std::integral_constant<bool, true> pred_t;
        while (p + U <= e) chunk(std::integral_constant<int, U>(), full_t());
        if constexpr (U > 16) { if (p + 16 <= e) chunk(std::integral_constant<int, 16>(), full_t()); }
        if (p + 8 <= e) chunk(std::integral_constant<int, 8>(), full_t());
        while (p < e) chunk(std::integral_constant<int, 8>(), pred_t());
        p = e;
        u32 *hg = H + ((size_t)g * tiles + tile) * (RT * 64) + lane;
        for (int c = 0; c < RT; ++c) {
            const u32 n = cl[c * 64];
            if (n) { atomicAdd(&hg[c * 64], n); cl[c * 64] = 0; }
        }
    }
    if (act && bad && P.gene_flags[gene] != 3u) P.gene_flags[gene] = 1u;
}

// hist_all[gene][c] (what k_fused_tables_all reads): OVR: the column's histogram = the sum over groups (grid.y stretches of 64 groups,
// added with atomics: hist_all zeroed by the host); OVO: the reference group's (grid.y = 1)
template <int RT, bool OVR>
__global__ __launch_bounds__(256) void k_group_hists_to_column(FusedParams P, const u32 *__restrict__ H) {
    const int tile = blockIdx.x, tiles = gridDim.x;
    const int g0 = OVR ? (int)blockIdx.y * 64 : P.ref, g1 = OVR ? min(P.G, g0 + 64) : P.ref + 1;
    for (int i = threadIdx.x; i < RT * 64; i += 256) {
        const int c = i >> 6, l = i & 63, gene = tile * 64 + l;
        if (gene >= P.ncols) continue;
        u32 s = 0;
        for (int g = g0; g < g1; ++g) s += H[((size_t)g * tiles + tile) * (RT * 64) + i];
        if (OVR) { if (s) atomicAdd(&P.hist_all[(size_t)gene * RT + c], s); }
        else P.hist_all[(size_t)gene * RT + c] = s;
    }
}

// grid (tiles, ceil(G / 4)): one wavefront per (group, tile), lane = gene
template <int RT, bool OVR>
__global__ __launch_bounds__(256) void k_emit_from_group_hists(FusedParams P, const u32 *__restrict__ H) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int tile = blockIdx.x, tiles = gridDim.x, gene = tile * 64 + lane;
    const int g = (int)blockIdx.y * 4 + wave;
    if (g >= P.G || (!OVR && g == P.ref)) return;
    if (gene >= P.ncols || P.gene_flags[gene] != 0u) return; // (flagged genes are recomputed by the other routes)
    const u32 *cum = P.ref_cum + (size_t)tile * (64 * (RT + 1)) + lane;
    const u32 *h = H + ((size_t)g * tiles + tile) * (RT * 64) + lane;
    u64 S2 = 0, TT = 0;
    u32 vsum = 0;
    u32 lo = cum[0];
    for (int c = 0; c < RT; ++c) {
        const u32 hi = cum[(c + 1) * 64], n = h[c * 64];
        if (n) {
            S2 += (u64)n * (u64)(lo + hi);
            if (!OVR) { // sum over t = a .. a + n - 1 of t (t + 1)
                const u64 a = hi - lo, t1 = a + n - 1;
                const u64 f1 = t1 * (t1 + 1) * (t1 + 2) / 3, f0 = a ? (a - 1) * a * (a + 1) / 3 : 0ull;
                TT += f1 - f0;
            }
            vsum += (u32)c * n;
        }
        lo = hi;
    }
    const long long n_tgt = P.counts[g];
    const u64 T_A = P.ref_TA[gene];
    const double ref_sum = (double)P.ref_sum[gene];
    const double cc = P.use_continuity ? 0.5 : 0.0;
    const GroupConst gc = P.gconst[g];
    double pv, Ustat, fc;
    if (OVR) { // as k_ovo_fused's emit (dense_ovr.py:57-75)
        const long long n_rest = P.n_cells - n_tgt;
        const long long two_u = 2ll * n_rest * n_tgt + n_tgt * (n_tgt + 1) - ((long long)S2 + n_tgt);
        Ustat = 0.5 * (double)two_u;
        const double tie = !P.tie_correct ? 0.0 : (P.tie_mode ? __longlong_as_double((long long)T_A) : (double)T_A);
        pv = pval_device_pre(gc.nnn, gc.var0, gc.n12, tie, Ustat, gc.mu, cc, P.alternative);
        fc = fold_change_device((double)vsum, ref_sum - (double)vsum, gc);
    } else {
        const long long n_ref = P.counts[P.ref];
        const double mu_ref_ovo = ref_sum / (double)n_ref;
        const u64 tie_i = T_A + 3ull * TT;
        const long long two_u = 2ll * n_ref * n_tgt - (long long)S2;
        Ustat = 0.5 * (double)two_u;
        const double tie = P.tie_correct ? (double)tie_i : 0.0;
        pv = pval_device_pre(gc.nnn, gc.var0, gc.n12, tie, Ustat, gc.mu, cc, P.alternative);
        fc = (mu_ref_ovo == 0.0) ? __longlong_as_double(0x7FF0000000000000ll) : ((double)vsum / gc.d_tgt) / mu_ref_ovo;
    }
    const size_t o = (size_t)g * P.out_ld + gene;
    P.out_p[o] = pv;
    P.out_u[o] = Ustat;
    P.out_fc[o] = fc;
}
