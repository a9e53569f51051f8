// The genes a count matrix's fused passes leave behind (values beyond the 64- / 256-value tables: the highly expressed genes of
// a real count matrix) are few and scattered.  Recomputing the column runs that cover them -- runs closer than 32 genes are
// merged, so 4 % flagged genes meant the whole matrix -- through the routes built for continuous values cost 130 ms (OVO) /
// 350 ms (OVR) at C2 shape with log-normal gene means, against 2 ms for the pass itself.  Instead:
//
//   k_gather_columns   copies the flagged columns into a narrow row-major matrix of their own (one pass over the rows; the
//                      flagged genes then form ONE contiguous window for the two-pass routes, whose results k_finalize scatters
//                      back through a column map);
//   k_scatter_planes   the 256-value stage run on that narrow matrix (run_leftovers, when k_wide_decide left the stage to the host) writes
//                      planes of its own; the finished columns go from there into the caller's planes;
//   k_ovr_counts       dense OVR for integer-valued genes below OVRC_R = 32768: the column's histogram in LDS (one 128-KB table,
//                      one workgroup per gene), turned in place into the doubled-rank table 2 cum[v] + t[v] + 1, then one
//                      wavefront per group sums the table entries of the group's cells.  No sort, no per-group histogram:
//                      OVR's tie term and ranks are properties of the column alone (illico/utils/ranking.py:31-47;
//                      illico/ovr/dense_ovr.py:46-75).  Count-like columns used to crowd the value buckets of the parts
//                      route and fall through to the per-gene radix sort in HBM (0.64 ms per gene).
//
// OVO keeps k_ovo_counts (kernels_ovo_counts.h: integers below 2048, or 4096 with groups of at most 255 cells) for these genes; what lies beyond takes the sort routes.
#pragma once
#include "common.h"
#include "kernels_ovo_counts.h"

// dst[r][dst_col0 + j] = src[r][cols[j]] for j < n; columns n .. n_pad - 1 of dst are zeroed.  grid.x = row blocks of 64 rows.
template <typename InT>
__global__ __launch_bounds__(256) void k_gather_columns(const InT *__restrict__ src, long long ld, int n_rows, const int *__restrict__ cols, int n,
                                                        int n_pad, InT *__restrict__ dst, long long dst_ld, long long dst_col0) {
    const int r0 = blockIdx.x * 64;
    const int r1 = min(r0 + 64, n_rows);
    for (int j0 = 0; j0 < n_pad; j0 += 256) {
        const int j = j0 + (int)threadIdx.x;
        const long long col = j < n ? (long long)cols[j] : -1;
        if (j < n_pad)
            for (int r = r0; r < r1; r += 8) { // eight rows' requests in flight per thread (one at a time the pass ran on latency)
                InT v[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) v[u] = (col >= 0 && r + u < r1) ? src[(size_t)(r + u) * ld + col] : (InT)0;
#pragma unroll
                for (int u = 0; u < 8; ++u)
                    if (r + u < r1) dst[(size_t)(r + u) * dst_ld + dst_col0 + j] = v[u];
            }
    }
}

// columns j < n of three [G][ld] planes whose flags[j] == want -> column map[j] of the caller's planes.  grid (column blocks, group blocks)
static __global__ __launch_bounds__(256) void k_scatter_planes(const double *__restrict__ p, const double *__restrict__ u, const double *__restrict__ fc, long long ld,
                                                        const int *__restrict__ map, const u32 *__restrict__ flags, u32 want, int n, int G,
                                                        double *op, double *ou, double *ofc, long long out_ld) {
    const int j = blockIdx.x * 256 + threadIdx.x;
    if (j >= n || flags[j] != want) return;
    const long long dst = map[j];
    for (int g = blockIdx.y; g < G; g += gridDim.y) {
        const size_t i = (size_t)g * ld + j, o = (size_t)g * out_ld + dst;
        op[o] = p[i];
        ou[o] = u[i];
        ofc[o] = fc[i];
    }
}

#define OVRC_R 32768
#define OVRC_NT 1024

struct OvrCountsParams {
    const void *Xt;           // [n_genes][stride] keys, positions group-contiguous (k_transpose_permute)
    long long stride;
    const int *pos_ptr;       // [G+1]
    const int *counts;        // [G]
    int G, n_genes, dt;
    long long n_cells;
    const u32 *gene_flags;    // 0 = every value an integer in [0, OVRC_R): this kernel's gene; else left to the other routes
    long long *out_2u;        // [n_genes][G]
    u64 *out_tie;             // [n_genes][G]
    double *out_sum;          // [n_genes][G]
    double *gene_total;       // [n_genes]
};

template <typename KeyT>
__global__ __launch_bounds__(OVRC_NT) void k_ovr_counts(OvrCountsParams P) {
    constexpr int NT = OVRC_NT, NW = NT / 64, R = OVRC_R, PER = R / NT;
    extern __shared__ __align__(16) u32 ovrc_tab[]; // [R]: histogram, then 2 cum[v] + t[v] + 1
    __shared__ u32 s_part[NT];
    __shared__ u64 s_tie[NW], s_tot[NW];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int N = (int)P.n_cells, G = P.G;
    for (int gene = blockIdx.x; gene < P.n_genes; gene += gridDim.x) {
        if (P.gene_flags[gene] != 0) continue; // uniform
        const KeyT *x = (const KeyT *)P.Xt + (size_t)gene * P.stride;
        for (int i = tid; i < R; i += NT) ovrc_tab[i] = 0;
        __syncthreads();
        // ---- the column's histogram (zeros, the most common value, by ballot) ----
        for (int i0 = 0; i0 < N; i0 += NT) {
            const int i = i0 + tid;
            const bool valid = i < N;
            const u32 c = valid ? count_of_key(x[i], P.dt) : 0u;
            const u64 zb = __ballot(valid && c == 0);
            if (valid && c != 0) atomicAdd(&ovrc_tab[c], 1u);
            if (lane == 0 && zb) atomicAdd(&ovrc_tab[0], (u32)__popcll(zb));
        }
        __syncthreads();
        // ---- exclusive scan over the values (PER consecutive bins per thread), tie sum, value total; the table in place ----
        u32 t[PER];
        u32 lsum = 0;
        u64 tie = 0, tot = 0;
#pragma unroll
        for (int e = 0; e < PER; ++e) {
            t[e] = ovrc_tab[tid * PER + e];
            lsum += t[e];
            const u64 tt = t[e];
            tie += tt * tt * tt - tt;
            tot += tt * (u64)(tid * PER + e);
        }
        s_part[tid] = lsum;
        tie = wave_sum<u64>(tie);
        tot = wave_sum<u64>(tot);
        if (lane == 0) { s_tie[wave] = tie; s_tot[wave] = tot; }
        __syncthreads();
        if (wave == 0) { // exclusive scan of the NT partial sums by one wavefront: NT / 64 per lane
            constexpr int PL = NT / 64;
            u32 loc[PL];
            u32 run = 0;
#pragma unroll
            for (int e = 0; e < PL; ++e) { loc[e] = s_part[lane * PL + e]; run += loc[e]; }
            u32 inc = run;
#pragma unroll
            for (int d = 1; d < 64; d <<= 1) {
                const u32 o = (u32)__shfl_up((int)inc, d);
                if (lane >= d) inc += o;
            }
            u32 base = inc - run;
#pragma unroll
            for (int e = 0; e < PL; ++e) { s_part[lane * PL + e] = base; base += loc[e]; }
        }
        __syncthreads();
        {
            u32 cum = s_part[tid];
#pragma unroll
            for (int e = 0; e < PER; ++e) {
                ovrc_tab[tid * PER + e] = 2u * cum + t[e] + 1u; // 2 rank(v) of a value with cum smaller cells and t equal ones
                cum += t[e];
            }
        }
        u64 tie_all = 0, tot_all = 0;
#pragma unroll
        for (int w = 0; w < NW; ++w) { tie_all += s_tie[w]; tot_all += s_tot[w]; }
        __syncthreads();
        if (tid == 0) P.gene_total[gene] = (double)tot_all; // integer sums: exact whatever the order of addition
        // ---- one wavefront per group: the doubled rank sum of its cells ----
        for (int g = wave; g < G; g += NW) {
            const int p0 = P.pos_ptr[g], p1 = P.pos_ptr[g + 1];
            u64 acc = 0, vs = 0;
            for (int p = p0 + lane; p < p1; p += 64) {
                const u32 c = count_of_key(x[p], P.dt);
                acc += ovrc_tab[c];
                vs += c;
            }
            acc = wave_sum<u64>(acc);
            vs = wave_sum<u64>(vs);
            if (lane == 0) {
                const long long n_g = p1 - p0;
                const size_t o = (size_t)gene * G + g;
                P.out_2u[o] = 2ll * (P.n_cells - n_g) * n_g + n_g * (n_g + 1) - (long long)acc; // dense_ovr.py:57-61
                P.out_tie[o] = tie_all;
                P.out_sum[o] = (double)vs;
            }
        }
        __syncthreads();
    }
}
