// Fused single-pass dense OVO for small-count genes: reads X once, straight from its row-major layout.
//
// Replaces chunk_and_fortranize + sort + rank_sum_and_ties_from_sorted + compute_pval + dense_fold_change
// (illico/ovo/dense_ovo.py:65-137, utils/math.py:247-278, utils/ranking.py:52-158, utils/math.py:64-118,196-221)
// for genes whose values are all integers in [0, RT).
//
// lane = gene (a wavefront owns 64 consecutive genes of one group's rows: every row read is one coalesced
// 256-B segment), rows are gathered through GroupContainer.indices, FUSED_U rows in flight per wavefront.
// Per element with value c of gene `lane`:
//     S2  += cum[c] + cum[c+1]                    (= 2 #ref<c + #ref==c; cum = cumulative reference histogram)
//     o    = (number of earlier cells of this group with the same value)   -- LDS fetch-and-add, lane-private
//     tie += 3 a^2 + 3 a (2o+1) + 3 o (o+1),  a = cum[c+1]-cum[c]
// Summed over a group, sum_o (2o+1) = tB^2 and sum_o (3o^2+3o+1) = tB^3, so this is exactly
// T_A + sum_v tB (3 tA (tA+tB) + tB^2 - 1) of kernels_ovo.h -- the same integers, bit-exact -- without any
// sort, merge or per-group histogram scan.  The wavefront then evaluates U, p and fold change for its 64
// (group, gene) pairs with all lanes active and writes 512-B output segments: no transpose pass, no
// intermediate statistics.
//
// A gene that shows a value outside the table anywhere sets gene_flags[gene]; the host re-runs flagged genes
// through the two-pass routes (k_ovo_counts / k_ovo_rank), which overwrite the columns.
#pragma once
#include "common.h"
#include "kernels_finalize.h"

#define FUSED_NT 256
#define FUSED_U 16

struct FusedParams {
    const void *X;
    long long ld, col0;       // genes [col0, col0 + ncols) of X
    int ncols;
    const int *perm;          // [N] GroupContainer.indices
    const int *pos_ptr;       // [G+1]
    const int *counts;        // [G]
    int G, ref;
    u32 *ref_hist;            // [ncols][RT]   reference histogram (zeroed by the host)
    u32 *ref_cum;             // [ncols][RT+1] cumulative
    u64 *ref_TA;              // [ncols] sum_v (tA^3 - tA)
    u64 *ref_sum;             // [ncols] sum of reference values
    u32 *gene_flags;          // [ncols] set to 1 when the gene cannot take this route
    int use_continuity, tie_correct, alternative;
    double *out_p, *out_u, *out_fc; // [G][out_ld], already offset to column col0's slot
    long long out_ld;
    int groups_per_wg;
    int ref_rows_per_wg;
};

template <typename InT> __device__ __forceinline__ u32 small_count(InT v, int RT, bool &ok);
template <> __device__ __forceinline__ u32 small_count<float>(float v, int RT, bool &ok) {
    const bool inr = v >= 0.0f && v < (float)RT; // false for NaN
    const u32 c = inr ? (u32)v : 0u;
    ok = inr && (float)c == v;
    return c;
}
template <> __device__ __forceinline__ u32 small_count<double>(double v, int RT, bool &ok) {
    const bool inr = v >= 0.0 && v < (double)RT;
    const u32 c = inr ? (u32)v : 0u;
    ok = inr && (double)c == v;
    return c;
}
template <> __device__ __forceinline__ u32 small_count<int32_t>(int32_t v, int RT, bool &ok) {
    ok = v >= 0 && v < RT;
    return ok ? (u32)v : 0u;
}
template <> __device__ __forceinline__ u32 small_count<int64_t>(int64_t v, int RT, bool &ok) {
    ok = v >= 0 && v < (int64_t)RT;
    return ok ? (u32)v : 0u;
}

// ---- reference histogram: grid (tiles, row chunks); lane = gene; per-wave LDS histogram, flushed with atomics ----
template <typename InT, int RT>
__global__ __launch_bounds__(FUSED_NT) void k_fused_ref_hist(FusedParams P) {
    constexpr int NW = FUSED_NT / 64, STR = RT / 2 + 1; // two 16-bit bins per word, odd stride: conflict-free
    __shared__ u32 h[NW][64 * STR];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int gene0 = blockIdx.x * 64, gene = gene0 + lane;
    const bool act = gene < P.ncols;
    u32 *hw = h[wave] + lane * STR;
    for (int i = 0; i < STR; ++i) hw[i] = 0;
    const int p0 = P.pos_ptr[P.ref], p1 = P.pos_ptr[P.ref + 1];
    const int chunk = P.ref_rows_per_wg;                 // <= 65535 * NW rows per workgroup
    const int wb = p0 + blockIdx.y * chunk, we = min(wb + chunk, p1);
    const InT *X = (const InT *)P.X;
    bool bad = false;
    for (int p = wb + wave; p < we; p += NW) {
        const long long row = P.perm[p];
        if (act) {
            bool ok;
            u32 c = small_count<InT>(X[row * P.ld + P.col0 + gene], RT, ok);
            bad |= !ok;
            c = min(c, (u32)(RT - 1));
            hw[c >> 1] += (c & 1u) ? 0x10000u : 1u;      // lane-private: plain read-modify-write
        }
    }
    if (act) {
        if (bad) P.gene_flags[gene] = 1u;
        for (int w = 0; w < RT / 2; ++w) {
            u32 word = hw[w];
            if (word & 0xFFFFu) atomicAdd(&P.ref_hist[(size_t)gene * RT + 2 * w], word & 0xFFFFu);
            if (word >> 16) atomicAdd(&P.ref_hist[(size_t)gene * RT + 2 * w + 1], word >> 16);
        }
    }
}

// ---- per gene: histogram -> cumulative table, T_A, reference sum; also writes the reference group's row ----
template <int RT> __global__ void k_fused_ref_scan(FusedParams P) {
    const int gene = blockIdx.x * blockDim.x + threadIdx.x;
    if (gene >= P.ncols) return;
    const u32 *h = P.ref_hist + (size_t)gene * RT;
    u32 *cum = P.ref_cum + (size_t)gene * (RT + 1);
    u32 run = 0;
    u64 ta = 0, sum = 0;
    cum[0] = 0;
    for (int c = 0; c < RT; ++c) {
        u64 t = h[c];
        run += (u32)t;
        cum[c + 1] = run;
        ta += t * t * t - t;
        sum += t * (u64)c;
    }
    P.ref_TA[gene] = ta;
    P.ref_sum[gene] = sum;
    const size_t o = (size_t)P.ref * P.out_ld + gene;
    P.out_p[o] = 1.0;                                                            // sparse_ovo.py:140-143
    P.out_u[o] = -1.0;
    P.out_fc[o] = (sum == 0) ? __longlong_as_double(0x7FF0000000000000ll) : 1.0; // math.py:190-192 with mu_tgt == mu_ref
}

// ---- main pass: grid (tiles, group chunks); 4 wavefronts per workgroup, one group at a time per wavefront ----
template <typename InT, int RT>
__global__ __launch_bounds__(FUSED_NT) void k_ovo_fused(FusedParams P) {
    constexpr int NW = FUSED_NT / 64, CSTR = RT + 1, BSTR = RT / 2 + 1, U = FUSED_U;
    __shared__ u32 cumA[64 * CSTR];        // cumA[lane*CSTR + c] = # reference cells of gene `lane` with value < c
    __shared__ u32 cntB[NW][64 * BSTR];    // per wavefront, per gene: running multiplicity of each value (16-bit pairs)
    __shared__ int s_skip;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int gene0 = blockIdx.x * 64, gene = gene0 + lane;
    const bool act = gene < P.ncols;
    // every gene of this tile already sent to the slow routes? then there is nothing to do here
    if (wave == 0) {
        const bool flagged = !act || P.gene_flags[gene] != 0;
        const bool all = __all(flagged);
        if (lane == 0) s_skip = all ? 1 : 0;
    }
    __syncthreads();
    if (s_skip) return;
    for (int i = tid; i < 64 * CSTR; i += FUSED_NT) {
        const int l = i / CSTR, c = i - l * CSTR;
        cumA[i] = (gene0 + l < P.ncols) ? P.ref_cum[(size_t)(gene0 + l) * CSTR + c] : 0u;
    }
    u32 *cb = cntB[wave] + lane * BSTR;
    for (int i = 0; i < BSTR; ++i) cb[i] = 0;
    __syncthreads();
    const u32 *ca = cumA + lane * CSTR;
    const InT *X = (const InT *)P.X;
    const long long n_ref = P.counts[P.ref];
    const u64 T_A = act ? P.ref_TA[gene] : 0ull;
    const double ref_sum = act ? (double)P.ref_sum[gene] : 0.0;
    const double cc = P.use_continuity ? 0.5 : 0.0;
    bool bad = false;

    const int gbeg = blockIdx.y * P.groups_per_wg, gend = min(gbeg + P.groups_per_wg, P.G);
    for (int g = gbeg + wave; g < gend; g += NW) {
        if (g == P.ref) continue;
        const int p0 = __builtin_amdgcn_readfirstlane(P.pos_ptr[g]);
        const int p1 = __builtin_amdgcn_readfirstlane(P.pos_ptr[g + 1]);
        u64 S2 = 0, A2 = 0, AB = 0, OO = 0;
        u32 vsum = 0;
        for (int p = p0; p < p1; p += U) {
            InT v[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                v[u] = (InT)0;
                if (p + u < p1) { // wave-uniform
                    const long long row = __builtin_amdgcn_readfirstlane(P.perm[p + u]);
                    const InT *rp = X + row * P.ld + P.col0 + gene0;
                    if (act) v[u] = rp[lane];
                }
            }
#pragma unroll
            for (int u = 0; u < U; ++u) {
                if (p + u < p1) {
                    bool ok;
                    u32 c = small_count<InT>(v[u], RT, ok);
                    bad |= !ok;
                    c = min(c, (u32)(RT - 1));
                    const u32 lo = ca[c], hi = ca[c + 1];
                    const u32 old = atomicAdd(&cb[c >> 1], (c & 1u) ? 0x10000u : 1u); // lane-private word: fetch-and-add
                    const u32 o = (c & 1u) ? (old >> 16) : (old & 0xFFFFu);
                    const u32 a = hi - lo;
                    S2 += (u64)lo + hi;
                    A2 += (u64)a * a;
                    AB += (u64)a * (2u * o + 1u);
                    OO += (u64)o * (o + 1u);
                    vsum += c;
                }
            }
        }
        // ---- this lane's (group, gene) result ----
        const long long n_tgt = p1 - p0;
        if (act) {
            const u64 tie_i = T_A + 3ull * (A2 + AB + OO);
            const long long two_u = 2ll * n_ref * n_tgt - (long long)S2;
            const double Ustat = 0.5 * (double)two_u;
            const double tie = P.tie_correct ? (double)tie_i : 0.0;
            const double mu = (double)(n_ref * n_tgt) / 2.0;
            const double pv = pval_device(n_ref, n_tgt, n_ref + n_tgt, tie, Ustat, mu, cc, P.alternative);
            const double mu_tgt = (double)vsum / (double)n_tgt;
            const double mu_ref = ref_sum / (double)n_ref;
            const double fc = (mu_ref == 0.0) ? __longlong_as_double(0x7FF0000000000000ll) : mu_tgt / mu_ref;
            const size_t o = (size_t)g * P.out_ld + gene;
            P.out_p[o] = pv;
            P.out_u[o] = Ustat;
            P.out_fc[o] = fc;
        }
        for (int i = 0; i < BSTR; ++i) cb[i] = 0; // lane-private, in-order LDS: no barrier needed
    }
    if (act && bad) P.gene_flags[gene] = 1u;
}
