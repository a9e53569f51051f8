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
//     tie += 3 a^2 + 3 a (2o+1) + 3 o (o+1) = 3 t (t+1),  a = cum[c+1]-cum[c],  t = a + o
// Summed over a group, sum_o (2o+1) = tB^2 and sum_o (3o^2+3o+1) = tB^3, so this is exactly
// T_A + sum_v tB (3 tA (tA+tB) + tB^2 - 1) of kernels_ovo.h -- the same integers, bit-exact -- without any
// sort, merge or per-group histogram scan.  The wavefront then evaluates U, p and fold change for its 64
// (group, gene) pairs with all lanes active and writes 512-B output segments: no transpose pass, no
// intermediate statistics.
//
// A gene that shows a value outside the table anywhere sets gene_flags[gene]; the host re-runs flagged genes
// through the two-pass routes (k_ovo_counts / k_ovo_rank), which overwrite the columns.
#pragma once
#include <type_traits>
#include "common.h"
#include "kernels_finalize.h"

#define FUSED_NT 256
#ifndef FUSED_U
#define FUSED_U 32
#endif

struct FusedParams {
    const void *X;
    long long ld, col0;       // genes [col0, col0 + ncols) of X
    int ncols;
    const int *perm;          // [N] GroupContainer.indices
    const int *pos_ptr;       // [G+1]
    const int *counts;        // [G]
    const GroupConst *gconst; // [G] what compute_pval forms from the group sizes alone (kernels_finalize.h)
    int G, ref;
    u32 *ref_cum;             // [tile][RT+1][64] cumulative counts, tile = 64 consecutive genes (the LDS image of k_ovo_fused)
    u64 *ref_TA;              // [ncols] sum_v (tA^3 - tA); OVR with tie_mode != 0: the BITS of the float64 the reference's accumulator holds
    int tie_mode;             // OVR: 1 = the reference's dense path (exact integers added block by block, ascending: utils/ranking.py:30-47),
                              // 2 = its sparse path (non-zero blocks, then n0**3 - n0 in float64: ovr/sparse_ovr.py:49,83) -- what a CSR
                              // window is held to.  The same value as the exact sum while n^3 fits 53 bits; beyond, the bits the reference has
    u64 *ref_sum;             // [ncols] sum of reference values
    u32 *hist_all;            // [ncols][RT] whole-column histogram (OVR; zeroed by the host)
    long long n_cells;
    int rows_per_wg;          // OVR histogram pass: rows per workgroup
    u32 *gene_flags;          // [ncols] set to 1 when the gene cannot take this route
    int use_continuity, tie_correct, alternative;
    double *out_p, *out_u, *out_fc; // [G][out_ld], already offset to column col0's slot
    long long out_ld;
    int groups_per_wg;
    u32 *group_hist;          // OVR one-pass form: [tiles][G][RT * CB / 32][64] words, per-(group, gene) value histograms
    u32 *wide_tiles;          // WIDE: [0] = number of tiles with candidates, [1 ..] = those tiles (k_fused_ref<WIDE> appends)
    u32 *wide_bad;            // OVR second stage: [ncols] set when a column shows a value outside the 256-value table too
    long long hist_total;      // OVR one pass, width per group: words per lane of ALL groups (hist_off[G])
    int hist_full;             // 1: k_ovr_group_hists writes every word of every histogram (option "ovr_full_dump", A/B)
    unsigned char *hist_words; // OVR one pass: [G][tiles] words of a (group, tile) histogram that were written (the rest are zero)
    const u32 *wide_skip;     // WIDE: *wide_skip != 0 (k_wide_decide): the 256-value stage is left to the host (every WIDE kernel returns at once)
    const u32 *hist_off;      // OVR one-pass form, mixed cell widths: [G + 1] words per lane before group g (16 for a group of <= 255 cells, else 32)
};

// Table index of a value, clamped into [0, RT-1], and whether the value IS that integer (else the gene leaves
// this route).  One v_med3 replaces the range compares: out-of-range, fractional and NaN all fail `exact`.
template <typename InT, int RT> __device__ __forceinline__ u32 clamp_count(InT v, bool &exact) {
    if constexpr (std::is_same<InT, float>::value) {
        const float m = __builtin_amdgcn_fmed3f(v, 0.0f, (float)(RT - 1));
        const u32 c = (u32)m;
        exact = (float)c == v;
        return c;
    } else if constexpr (std::is_same<InT, double>::value) {
        const double m = fmin(fmax(v, 0.0), (double)(RT - 1)); // NaN -> 0
        const u32 c = (u32)m;
        exact = (double)c == v;
        return c;
    } else if constexpr (std::is_same<InT, uint8_t>::value) { // (byte windows written by k_csr_densify: 255 marks a value the window cannot hold)
        const u32 c = min((u32)v, (u32)(RT - 1));
        exact = c == (u32)v && (u32)v != 255u;
        return c;
    } else {
        const InT cl = min(max(v, (InT)0), (InT)(RT - 1));
        exact = cl == v;
        return (u32)cl;
    }
}

// One chunk = UU rows of one group for the wavefront's 64 genes, in two straight-line halves: gather_rows requests
// the UU row segments back to back, consume_* works through them in order behind counted vmcnt waits.
//  * row indices come from scalar loads (the perm array is read through the constant address space, so a uniform
//    index gives s_load_dwordx8); perm is padded so that a chunk may read past the group's end;
//  * the row base is a scalar 64-bit address (row * row_bytes: one s_mul_i32 + one s_mul_hi_u32), the lane adds a
//    32-bit column offset: global_load_dword v, v_off, s[base] -- no vector address arithmetic (the row pitch must
//    fit 32 bits; the host sends wider matrices through the two-pass routes);
//  * the running multiplicities are plain CB-bit LDS cells updated by read + write (the table is lane-private and the
//    LDS executes a wavefront's operations in order, so no atomic is needed): no shift / mask / bit-field extract.
//    CB = 16 (groups up to 65535 cells) or 8 (groups up to 255 cells: half the LDS, one more workgroup per CU).
//    Tables are laid out [lane][value] with an odd lane stride: equal values never conflict, different values collide
//    at random (the conflict-free [value][lane] layout measured 0.8 % slower: one more address instruction per access).
// History of the OVO loop at C2 (same-process A/B): vector-loaded indices + v_readlane + vector addresses + packed
// fetch-and-add counters 2.33 ms -> this form 2.07 ms.
typedef const __attribute__((address_space(4))) int *const_int_p;
template <int CB> struct CntCell;
template <> struct CntCell<8> { typedef unsigned char type; };
template <> struct CntCell<16> { typedef unsigned short type; };
template <> struct CntCell<32> { typedef unsigned int type; };   // groups above 65535 cells (clusters of an atlas): 82 KB of LDS, one workgroup per CU
// UU rows of the wavefront's 64-gene tile, requested back to back: v[u] = X[row_u][gene0 + lane], row_u = perm[p + u].
// p, p1 are wave-uniform.  PRED: positions at or past p1 (the group's end) re-read the group's last row -- a cache hit,
// no HBM traffic -- and are masked out by the consumer.
template <typename InT, int UU, bool PRED, int NV>
__device__ __forceinline__ void gather_rows(const char *__restrict__ Xg, u32 row_bytes, const_int_p perm, int p, int p1, u32 col_bytes,
                                            InT (&v)[NV]) {
    static_assert(UU <= NV, "chunk larger than the value array");
    p = __builtin_amdgcn_readfirstlane(p); // no-ops when the compiler already knows these are uniform
    p1 = __builtin_amdgcn_readfirstlane(p1);
    const int last = PRED ? perm[max(p1 - 1, 0)] : 0;
    u32 coff = col_bytes;
    asm volatile("" : "+v"(coff)); // keeps the 32-bit -> 64-bit extension of the lane offset next to the loads (saddr form)
#pragma unroll
    for (int u = 0; u < UU; ++u) {
        int row = perm[p + u]; // perm is padded: reading past the group's (or the array's) end is safe
        if (PRED) row = (p + u < p1) ? row : last;
        // scalar row base = Xg + row * row_bytes, spelled out: left to itself the compiler moves the row index to a
        // VGPR and forms every address with v_mad_u64_u32 / v_lshl_add_u64 instead of using the saddr form of the load
        u32 blo, bhi;
        asm("s_mul_hi_u32 %1, %2, %3\n\ts_mul_i32 %0, %2, %3\n\ts_add_u32 %0, %0, %4\n\ts_addc_u32 %1, %1, %5"
            : "=&s"(blo), "=&s"(bhi)
            : "s"(row), "s"(row_bytes), "s"((u32)(uintptr_t)Xg), "s"((u32)((uintptr_t)Xg >> 32))
            : "scc");
        typedef const __attribute__((address_space(1))) char *gchar_p;
        typedef const __attribute__((address_space(1))) InT *gval_p;
        const gchar_p rp = (gchar_p)(((u64)bhi << 32) | blo); // uniform row base
        v[u] = *(gval_p)(rp + coff);
    }
}

// The arithmetic of one gathered chunk.  Per element (value c, reference multiplicity a = cum[c+1]-cum[c], o = earlier
// cells of the group with value c):   S2 += cum[c] + cum[c+1];   TT += t (t+1),  t = a + o   [= a^2 + a(2o+1) + o(o+1)]
// PRED: positions at or past p1 are masked out.
template <typename InT, int RT, int UU, bool PRED, int CB, int LS, int NV>
__device__ __forceinline__ void consume_rmw(const InT (&v)[NV], int p, int p1, const u32 *ca, typename CntCell<CB>::type *cb, u64 &S2,
                                            u64 &TT, u32 &vsum, bool &inexact) {
    u32 s2c = 0;
#pragma unroll
    for (int u = 0; u < UU; ++u) {
        bool exact;
        const u32 c = clamp_count<InT, RT>(v[u], exact);
        const u32 lo = ca[c * LS], hi = ca[c * LS + LS];
        const u32 old = cb[c * LS];
        if (PRED) {
            const bool valid = p + u < p1; // wave-uniform
            inexact |= valid && !exact;
            cb[c * LS] = (typename CntCell<CB>::type)(old + (valid ? 1u : 0u));
            const u32 t = valid ? (hi - lo) + old : 0u;
            s2c += valid ? lo + hi : 0u;
            TT += (u64)t * (t + 1u);
            vsum += valid ? c : 0u;
        } else {
            inexact |= !exact;
            cb[c * LS] = (typename CntCell<CB>::type)(old + 1u);
            const u32 t = (hi - lo) + old;
            s2c += lo + hi;
            TT += (u64)t * (t + 1u);
            vsum += c;
        }
    }
    S2 += s2c;
}
// OVR: R2 += cum[c] + cum[c+1] (= 2 #cells<c + #cells==c), value sum.
template <typename InT, int RT, int UU, bool PRED, int LS, int NV>
__device__ __forceinline__ void consume_ovr(const InT (&v)[NV], int p, int p1, const u32 *ca, u64 &R2, u32 &vsum) {
    u32 r2c = 0; // <= UU * 2 * n_cells: fits 32 bits for n_cells < 2^25 per chunk of 32
#pragma unroll
    for (int u = 0; u < UU; ++u) {
        bool exact;
        const u32 c = clamp_count<InT, RT>(v[u], exact);
        const u32 lo = ca[c * LS], hi = ca[c * LS + LS];
        const bool valid = !PRED || (p + u < p1);
        r2c += valid ? lo + hi : 0u;
        vsum += valid ? c : 0u;
    }
    R2 += r2c;
}

// Of n_samples evenly spaced cells of the window: n_bad[0] = not a non-negative integer, n_bad[1] = an integer of `limit`
// or more (as k_sample_noncount; route choice only)
template <typename InT>
__global__ void k_sample_noncount_dense(const InT *__restrict__ X, long long ld, long long col0, long long n_rows, long long W, int n_samples,
                                        int limit, u32 *__restrict__ n_bad) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_samples) return;
    const long long k = (long long)((double)i * (double)(n_rows * W) / (double)n_samples);
    const long long r = k / W, j = k - r * W;
    const InT v = X[r * ld + col0 + j];
    const bool integer = v >= (InT)0 && v < (InT)(1 << 24) && (InT)(int)v == v;
    // one atomic per wavefront and counter: on normalised data every sample would otherwise hit the same address
    const u64 b0 = __ballot(!integer), b1 = __ballot(integer && v >= (InT)limit);
    if ((threadIdx.x & 63) == 0) {
        if (b0) atomicAdd(n_bad, (u32)__popcll(b0));
        if (b1) atomicAdd(n_bad + 1, (u32)__popcll(b1));
    }
}

// Route probe, on the device and for the device: PROBE_ROWS evenly spaced rows of every gene of the window; a gene that shows
// a value outside the table is flagged here, before the main pass, whose workgroups leave at once when every gene of their
// tile is flagged.  On normalised (continuous) data that is every gene, so the pass over X costs nothing and no host round
// trip is needed to choose the route; a gene this probe misses is flagged by the main pass itself.
// (1024 rows, 16 wavefronts: with 256 a gene stored in 1 % of its cells -- a lowly expressed gene of a normalised matrix -- showed the probe
//  nothing but zeros one time in ten, and its whole tile was then read by the passes behind: 1.2 ms of 8.9 on ten clusters of 100 000 cells)
#define FUSED_PROBE_ROWS 1024
#define FUSED_PROBE_NT 1024
template <typename InT, int RT>
__global__ __launch_bounds__(FUSED_PROBE_NT) void k_fused_probe(FusedParams P) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int gene = blockIdx.x * 64 + lane;
    if (gene >= P.ncols) return;
    const InT *Xg = (const InT *)P.X + P.col0 + gene;
    bool bad = false, hopeless = false; // hopeless: not a count below 256 either: the wider second stage need not look at this gene
    constexpr int PER = FUSED_PROBE_ROWS / (FUSED_PROBE_NT / 64);
    InT v[8];
    for (int i0 = 0; i0 < PER; i0 += 8) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const long long r = ((long long)(wave * PER + i0 + u) * P.n_cells) / FUSED_PROBE_ROWS;
            v[u] = Xg[r * P.ld];
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            bool exact, exact_wide;
            clamp_count<InT, RT>(v[u], exact);
            clamp_count<InT, 256>(v[u], exact_wide);
            bad |= !exact;
            hopeless |= !exact_wide;
        }
    }
    if (bad) atomicMax(&P.gene_flags[gene], hopeless ? 3u : 1u); // 3: as 1 (the host's two-pass routes), and skipped by the 256-value stage
}

// After the 64-value pass: is the 256-value stage worth running over the window as it lies?  It reads every row of every tile that holds
// a flagged gene again, at one or two workgroups per CU.  When most tiles hold one (a heavy-tailed count matrix: 20 % of the genes beyond
// 63 put one in EVERY tile, and the stage re-reads the whole matrix at 2 TB/s) and the flagged genes are few enough to be gathered, the
// stage is left to the host, which gathers the flagged columns into a narrow matrix and runs it there (run_leftovers).  One workgroup.
static __global__ __launch_bounds__(1024) void k_wide_decide(const u32 *__restrict__ gene_flags, int ncols, int max_gather, u32 *skip) {
    __shared__ u32 s_tiles, s_genes;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, tiles = (ncols + 63) / 64;
    if (tid == 0) { s_tiles = 0; s_genes = 0; }
    __syncthreads();
    u32 t_cnt = 0, g_cnt = 0;
    for (int t = wave; t < tiles; t += 16) {
        const int gene = t * 64 + lane;
        const u32 f = gene < ncols ? gene_flags[gene] : 0u;
        t_cnt += __any(f == 1u) ? 1u : 0u;
        g_cnt += (u32)__popcll(__ballot(f == 1u || f == 3u));
    }
    if (lane == 0) { atomicAdd(&s_tiles, t_cnt); atomicAdd(&s_genes, g_cnt); }
    __syncthreads();
    if (tid == 0) *skip = (s_tiles >= 8u && s_tiles * 4u > (u32)tiles && s_genes * 2u <= (u32)ncols && s_genes <= (u32)max_gather) ? 1u : 0u;
}

// ---- reference tables: one 1024-thread workgroup per 64-gene tile; lane = gene.  All 16 wavefronts add into one
// LDS histogram (columns are lane-private, so the only contention is between wavefronts), then wavefront 0 scans
// each gene's bins into the cumulative table, T_A and the reference sum, and writes the reference group's row.
// WIDE: the second, wider table (RT = 256) for the genes the 64-value pass flagged (gene_flags == 1): tiles without such a gene
// leave at once; a flagged gene whose reference values all fit becomes a candidate (gene_flags = 2) for k_ovo_fused<WIDE>.
#define FUSED_REF_NT 1024
#define FUSED_WIDE_RT 256
static inline size_t fused_ref_lds_bytes(int rt) { return (size_t)64 * (rt + 1) * 4; }
template <typename InT, int RT, bool WIDE = false>
__global__ __launch_bounds__(FUSED_REF_NT) void k_fused_ref(FusedParams P) {
    constexpr int NW = FUSED_REF_NT / 64, STR = RT + 1, UR = 16;
    extern __shared__ __align__(16) u32 h[]; // [64 * STR]
    __shared__ int s_bad[64];
    __shared__ int s_skip;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int gene0 = blockIdx.x * 64, gene = gene0 + lane;
    const bool act = gene < P.ncols;
    if (WIDE) {
        if (P.wide_skip && *P.wide_skip) return; // uniform
        if (wave == 0) {
            const bool want = act && P.gene_flags[gene] == 1u;
            const bool any = __any(want);
            if (lane == 0) s_skip = any ? 0 : 1;
        }
        __syncthreads();
        if (s_skip) return;
    }
    for (int i = tid; i < 64 * STR; i += FUSED_REF_NT) h[i] = 0;
    if (tid < 64) s_bad[tid] = 0;
    __syncthreads();
    const int p0 = P.pos_ptr[P.ref], p1 = P.pos_ptr[P.ref + 1];
    const InT *Xg = (const InT *)P.X + P.col0 + gene0;
    const int lane_c = act ? lane : 0;
    u32 *hl = h + lane * STR;
    bool bad = false, hopeless = false; // hopeless (64-value pass only): a reference value that is no count below 256 either
    for (int p = p0 + wave * UR; p < p1; p += NW * UR) {
        InT v[UR];
        // rows past p1 are other groups' (or the padding's): valid memory, masked below
        gather_rows<InT, UR, false>((const char *)Xg, (u32)P.ld * (u32)sizeof(InT), (const_int_p)P.perm, p, p1, (u32)lane_c * (u32)sizeof(InT), v);
#pragma unroll
        for (int u = 0; u < UR; ++u) {
            const bool valid = p + u < p1;
            bool exact;
            const u32 c = clamp_count<InT, RT>(v[u], exact);
            bad |= valid && !exact;
            if (!WIDE && RT < FUSED_WIDE_RT) { bool ew; clamp_count<InT, FUSED_WIDE_RT>(v[u], ew); hopeless |= valid && !ew; }
            atomicAdd(&hl[c], valid ? 1u : 0u);
        }
        // every gene of the tile already known to be no count at all (continuous data: after the first rows): nothing this
        // wavefront would still add is used -- the genes are flagged 3 and recomputed, reference row included, by the two-pass routes
        if (!WIDE && RT < FUSED_WIDE_RT && __all(hopeless || !act)) break;
    }
    if (bad) atomicMax(&s_bad[lane], hopeless ? 2 : 1);
    __syncthreads();
    if (wave == 0) { // counts -> cumulative counts, in place: h[lane][c] = # reference cells < c, c = 0 .. RT
        u32 run = 0;
        u64 ta = 0, sum = 0;
        for (int c = 0; c < RT; ++c) {
            const u64 t = hl[c];
            hl[c] = run;
            run += (u32)t;
            ta += t * t * t - t;
            sum += t * (u64)c;
        }
        hl[RT] = run;
        if (act && (!WIDE || P.gene_flags[gene] == 1u)) {
        P.ref_TA[gene] = ta;
        P.ref_sum[gene] = sum;
        if (WIDE) { if (!s_bad[lane]) { P.gene_flags[gene] = 2u; s_skip = 2; } } // candidate for the wide main pass (else it stays flagged); s_skip = 2: this tile has one
        else if (s_bad[lane] && P.gene_flags[gene] != 3u) P.gene_flags[gene] = s_bad[lane] == 2 ? 3u : 1u; // 3: the 256-value stage need not look
        const size_t o = (size_t)P.ref * P.out_ld + gene;
        P.out_p[o] = 1.0;                                                            // sparse_ovo.py:140-143
        P.out_u[o] = -1.0;
        P.out_fc[o] = (sum == 0) ? __longlong_as_double(0x7FF0000000000000ll) : 1.0; // math.py:190-192 with mu_tgt == mu_ref
        }
    }
    __syncthreads();
    if (WIDE) { // the tile joins the list the wide main pass works through (resident workgroups: nothing listed, nothing launched)
        if (s_skip != 2) return; // uniform
        if (tid == 0) P.wide_tiles[1 + atomicAdd(&P.wide_tiles[0], 1u)] = blockIdx.x;
    }
    // copy-out as the [value][lane] image k_ovo_fused keeps in LDS: coalesced stores, conflict-free LDS reads
    u32 *dst = P.ref_cum + (size_t)blockIdx.x * (64 * STR);
    for (int i = tid; i < 64 * STR; i += FUSED_REF_NT) dst[i] = h[(i & 63) * STR + (i >> 6)];
}

// ---- reference tables in two steps (the form the 64-value pass uses): k_fused_ref is one workgroup per tile -- 125 workgroups
// at C2, 59 at a C5 shard, on 256 CUs.  Here the reference rows are split over grid (tiles, row chunks); every workgroup adds
// its rows into an LDS histogram and flushes the non-empty bins with global integer atomics into hist_all;
// k_fused_tables_all then forms the cumulative tables, T_A, the reference sums and the reference group's row.
#define FUSED_REF_ROWS 1024 // reference rows per workgroup at least (the host aims at ~768 workgroups: FusedParams.rows_per_wg)
template <typename InT, int RT>
__global__ __launch_bounds__(FUSED_NT) void k_fused_ref_hist(FusedParams P) {
    constexpr int NW = FUSED_NT / 64, STR = RT + 1, UR = 16;
    __shared__ u32 h[64 * STR];
    __shared__ int s_bad[64];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int gene0 = blockIdx.x * 64, gene = gene0 + lane;
    const bool act = gene < P.ncols;
    for (int i = tid; i < 64 * STR; i += FUSED_NT) h[i] = 0;
    if (tid < 64) s_bad[tid] = 0;
    __syncthreads();
    const int pr0 = P.pos_ptr[P.ref], pr1 = P.pos_ptr[P.ref + 1];
    const int p0 = pr0 + (int)blockIdx.y * P.rows_per_wg, p1 = min(p0 + P.rows_per_wg, pr1);
    const InT *Xg = (const InT *)P.X + P.col0 + gene0;
    const int lane_c = act ? lane : 0;
    u32 *hl = h + lane * STR;
    bool bad = false;
    for (int p = p0 + wave * UR; p < p1; p += NW * UR) {
        InT v[UR];
        // rows past p1 are other cells' (or the padding's): valid memory, masked below
        gather_rows<InT, UR, false>((const char *)Xg, (u32)P.ld * (u32)sizeof(InT), (const_int_p)P.perm, p, p1, (u32)lane_c * (u32)sizeof(InT), v);
#pragma unroll
        for (int u = 0; u < UR; ++u) {
            const bool valid = p + u < p1;
            bool exact;
            const u32 c = clamp_count<InT, RT>(v[u], exact);
            bad |= valid && !exact;
            atomicAdd(&hl[c], valid ? 1u : 0u);
        }
    }
    if (bad) s_bad[lane] = 1;
    __syncthreads();
    for (int i = tid; i < 64 * RT; i += FUSED_NT) {
        const int l = i / RT, c = i - l * RT;
        const u32 cnt = h[l * STR + c];
        if (cnt && gene0 + l < P.ncols) atomicAdd(&P.hist_all[(size_t)(gene0 + l) * RT + c], cnt);
    }
    if (tid < 64 && act && s_bad[tid] && P.gene_flags[gene] != 3u) P.gene_flags[gene] = 1u;
}

// ---- OVR tables.  For one-versus-rest every cell is ranked against the whole column, so the table is the
// histogram of ALL cells: rank of value c = cum[c] + (cnt[c]+1)/2, and the tie term sum_v (t^3 - t) is a
// property of the column alone (ranking.py:31-47) -- no per-group multiplicities are needed.
// Pass A: grid (tiles, row chunks); the workgroup's wavefronts add into one LDS histogram, flushed with global
// integer atomics (non-empty bins only).
// WIDE (OVR second stage, RT = 256): only tiles that hold a gene the 64-value pass flagged (gene_flags == 1); a column that
// leaves this table too is marked in wide_bad (every row chunk of a column must agree before it becomes a candidate).
template <typename InT, int RT, bool WIDE = false>
__global__ __launch_bounds__(FUSED_NT) void k_fused_hist_all(FusedParams P) {
    constexpr int NW = FUSED_NT / 64, STR = RT + 1, UR = 32;
    extern __shared__ __align__(16) u32 h[]; // [64 * STR]
    __shared__ int s_bad[64];
    __shared__ int s_skip;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int gene0 = blockIdx.x * 64, gene = gene0 + lane;
    const bool act = gene < P.ncols;
    if (WIDE) {
        if (P.wide_skip && *P.wide_skip) return; // uniform
        if (wave == 0) {
            const bool want = act && P.gene_flags[gene] == 1u;
            const bool any = __any(want);
            if (lane == 0) s_skip = any ? 0 : 1;
        }
        __syncthreads();
        if (s_skip) return;
    } else { // every gene of the tile already flagged (the row probe on continuous data: all of them): nothing to count
        if (wave == 0) {
            const bool all = __all(!act || P.gene_flags[gene] != 0u);
            if (lane == 0) s_skip = all ? 1 : 0;
        }
        __syncthreads();
        if (s_skip) return;
    }
    for (int i = tid; i < 64 * STR; i += FUSED_NT) h[i] = 0;
    if (tid < 64) s_bad[tid] = 0;
    __syncthreads();
    const long long r0 = (long long)blockIdx.y * P.rows_per_wg, r1 = min(r0 + P.rows_per_wg, P.n_cells);
    const InT *Xg = (const InT *)P.X + P.col0 + gene0;
    const int lane_c = act ? lane : 0;
    u32 *hl = h + lane * STR;
    bool bad = false;
    for (long long r = r0 + wave * UR; r < r1; r += NW * UR) {
        InT v[UR];
#pragma unroll
        for (int u = 0; u < UR; ++u) v[u] = Xg[min(r + u, r1 - 1) * P.ld + lane_c];
#pragma unroll
        for (int u = 0; u < UR; ++u) {
            const bool valid = r + u < r1;
            bool exact;
            const u32 c = clamp_count<InT, RT>(v[u], exact);
            bad |= valid && !exact;
            atomicAdd(&hl[c], valid ? 1u : 0u);
        }
    }
    if (bad) s_bad[lane] = 1;
    __syncthreads();
    for (int i = tid; i < 64 * RT; i += FUSED_NT) {
        const int l = i / RT, c = i - l * RT;
        const u32 cnt = h[l * STR + c];
        if (cnt && gene0 + l < P.ncols) atomicAdd(&P.hist_all[(size_t)(gene0 + l) * RT + c], cnt);
    }
    if (tid < 64 && act && s_bad[tid]) { if (WIDE) P.wide_bad[gene] = 1u; else if (P.gene_flags[gene] != 3u) P.gene_flags[gene] = 1u; }
}
// per gene: histogram -> cumulative table, column tie sum, column total.  WIDE (OVR second stage): only for the genes the first
// pass flagged and whose every row fits the wider table; they become candidates (gene_flags = 2) and their tile joins the list
// the wide main pass works through (wide_tiles; tile_mark keeps a tile from being listed twice).
template <int RT, bool WIDE = false> __global__ void k_fused_tables_all(FusedParams P) {
    const int gene = blockIdx.x * blockDim.x + threadIdx.x;
    if (gene >= P.ncols) return;
    if (WIDE) {
        if (P.wide_skip && *P.wide_skip) return;
        if (P.gene_flags[gene] != 1u || P.wide_bad[gene] != 0u) return;
        P.gene_flags[gene] = 2u;
        u32 *tile_mark = P.wide_bad + P.ncols; // [tiles]
        if (atomicExch(&tile_mark[gene >> 6], 1u) == 0u) P.wide_tiles[1 + atomicAdd(&P.wide_tiles[0], 1u)] = (u32)(gene >> 6);
    }
    const u32 *h = P.hist_all + (size_t)gene * RT;
    u32 *cum = P.ref_cum + (size_t)(gene >> 6) * (64 * (RT + 1)) + (gene & 63); // [tile][value][lane]
    u32 run = 0;
    u64 ta = 0, sum = 0;
    cum[0] = 0;
    for (int c = 0; c < RT; ++c) {
        const u64 t = h[c];
        run += (u32)t;
        cum[(c + 1) * 64] = run;
        ta += t * t * t - t;
        sum += t * (u64)c;
    }
    if (P.ref < 0 && P.tie_mode) { // the float64 tie sum as the reference accumulates it
        if (P.tie_mode == 2) ta = tie_f64_sparse(ta - ((u64)h[0] * h[0] * h[0] - (u64)h[0]), (long long)h[0]);
        else {
            double td = 0.0;
            for (int c = 0; c < RT; ++c) { const u64 t = h[c]; td += (double)(t * t * t - t); }
            ta = (u64)__double_as_longlong(td);
        }
    }
    P.ref_TA[gene] = ta;
    P.ref_sum[gene] = sum;
    if (P.ref >= 0) { // OVO (tables of the reference group, k_fused_ref_hist): the reference group's own row
        const size_t o = (size_t)P.ref * P.out_ld + gene;
        P.out_p[o] = 1.0;                                                            // sparse_ovo.py:140-143
        P.out_u[o] = -1.0;
        P.out_fc[o] = (sum == 0) ? __longlong_as_double(0x7FF0000000000000ll) : 1.0; // math.py:190-192 with mu_tgt == mu_ref
    }
}

// ---- main pass: grid (tiles, group chunks); 4 wavefronts per workgroup, one group at a time per wavefront ----
// (Building the reference tables inside this kernel, per workgroup, instead of reading k_fused_ref's was measured at
// C2: no gain -- the 0.08 ms of k_fused_ref are matched by the redundant per-workgroup work.)
// WIDE (RT = 256, OVO): the second pass over the tiles that hold candidates of the wider table (gene_flags == 2, set by
// k_fused_ref<WIDE>); only those lanes are active, a candidate that shows a value beyond this table too goes back to
// gene_flags = 1 (the host's two-pass routes).  130 KB of LDS: one workgroup per CU -- it only runs where the first pass left genes.
template <int RT, bool OVR, int CB> static inline size_t fused_main_lds_bytes() {
    return (size_t)(RT + 1) * 64 * 4 + (size_t)(FUSED_NT / 64) * (OVR ? 1 : RT * CB / 32) * 64 * 4;
}
template <typename InT, int RT, bool OVR, int CB, int U = FUSED_U, bool WIDE = false>
__global__ __launch_bounds__(FUSED_NT, WIDE ? (OVR ? 2 : 1) : ((OVR || CB == 8) ? 4 : 3)) void k_ovo_fused(FusedParams P) {
    constexpr int NT = FUSED_NT, NW = NT / 64, CSTR = RT + 1, BW = OVR ? 1 : RT * CB / 32;
    // Both tables are laid out [value][lane]: the LDS bank of a lookup is set by the lane alone, whatever the values
    // (32-bit cells: conflict-free; 8- / 16-bit cells: four / two neighbouring lanes share a bank).  With [lane][value]
    // rows and an odd lane stride the data-dependent lookups collided at random (4.8 extra LDS cycles per instruction,
    // SQ_LDS_BANK_CONFLICT); same-process A/B: this layout 0.4 % faster.
    constexpr int LS = 64;                 // cell stride between consecutive values
    extern __shared__ __align__(16) u32 fused_lds[];
    u32 *cumA = fused_lds;                 // [CSTR * 64]: cumA[c][lane] = # reference cells of gene `lane` with value < c
    u32 *cntB_all = fused_lds + CSTR * 64; // [NW][BW * 64] per wavefront: running multiplicities [value][lane], CB bits each
    __shared__ int s_skip;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    // WIDE: resident workgroups (one per CU: 130 KB of LDS) work through (listed tile, group chunk) items; with nothing listed
    // every workgroup leaves at once.  A grid of one workgroup per item cost 0.08 ms at C2 for nothing: 31 250 workgroups
    // that each wait for a whole CU's LDS.
    const int n_chunks = (P.G + P.groups_per_wg - 1) / P.groups_per_wg;
    if (WIDE && P.wide_skip && *P.wide_skip) return; // uniform
    const int n_items = WIDE ? (int)P.wide_tiles[0] * n_chunks : 1;
    for (int item = WIDE ? (int)blockIdx.x : 0; item < n_items; item += WIDE ? (int)gridDim.x : 1) {
    const int tile = WIDE ? __builtin_amdgcn_readfirstlane((int)P.wide_tiles[1 + item / n_chunks]) : (int)blockIdx.x; // (uniform: scalar row bases below)
    const int gchunk = WIDE ? item % n_chunks : (int)blockIdx.y;
    if (WIDE) __syncthreads(); // the previous item's tables and s_skip are done with
    const int gene0 = tile * 64, gene = gene0 + lane;
    const bool act = gene < P.ncols && (!WIDE || P.gene_flags[gene] == 2u);
    const int lane_c = act ? lane : 0; // inactive lanes (tile wider than the batch) re-read a valid column
    const InT *Xg = (const InT *)P.X + P.col0 + gene0;
    const char *Xb = (const char *)Xg;
    const u32 row_bytes = (u32)P.ld * (u32)sizeof(InT), col_bytes = (u32)lane_c * (u32)sizeof(InT);
    const const_int_p permc = (const_int_p)P.perm;
    bool bad = false;
    // every gene of this tile already sent to the slow routes (WIDE: no candidate in it)? then there is nothing to do here
    if (wave == 0) {
        const bool flagged = !act || (!WIDE && P.gene_flags[gene] != 0);
        const bool all = __all(flagged);
        if (lane == 0) s_skip = all ? 1 : 0;
    }
    u32 *cbw = cntB_all + wave * (BW * 64); // the wavefront's counter block as words: lane zeroes words lane, lane + 64, ..
    for (int i = 0; i < BW; ++i) cbw[i * 64 + lane] = 0;
    __syncthreads();
    if (s_skip) { if (WIDE) continue; else return; }
    for (int i = tid; i < 64 * CSTR; i += NT) cumA[i] = P.ref_cum[(size_t)tile * (64 * CSTR) + i];
    __syncthreads();
    const u64 T_A = act ? P.ref_TA[gene] : 0ull;
    const double ref_sum = act ? (double)P.ref_sum[gene] : 0.0;
    const u32 *ca = cumA + lane;
    typedef typename CntCell<CB>::type cell_t;
    cell_t *cb = (cell_t *)cbw + lane;
    const long long n_ref = OVR ? 0 : P.counts[OVR ? 0 : P.ref];
    const double cc = P.use_continuity ? 0.5 : 0.0;
    const double mu_ref_ovo = OVR ? 0.0 : ref_sum / (double)n_ref; // (math.py:183: the reference group's mean is this lane's gene's, not a test's)

    const int gbeg = gchunk * P.groups_per_wg, gend = min(gbeg + P.groups_per_wg, P.G);

    // ---- this lane's (group, gene) result from the group's integer statistics ----
    auto emit = [&](int g, long long n_tgt, u64 S2, u64 TT, u32 vsum) {
        if (!act) return;
        double pv, Ustat, fc;
        const GroupConst gc = P.gconst[g]; // (uniform address: one scalar load per group)
        if (OVR) { // dense_ovr.py:57-75: the "reference" of group g is every other cell
            const long long n_rest = P.n_cells - n_tgt;
            // 2*ranksum = S2 + n_tgt (2 rank = 2 #less + #equal + 1);  U = n_rest n_tgt + n_tgt(n_tgt+1)/2 - ranksum
            const long long two_u = 2ll * n_rest * n_tgt + n_tgt * (n_tgt + 1) - ((long long)S2 + n_tgt);
            Ustat = 0.5 * (double)two_u;
            const double tie = !P.tie_correct ? 0.0 : (P.tie_mode ? __longlong_as_double((long long)T_A) : (double)T_A);
            pv = pval_device_pre(gc.nnn, gc.var0, gc.n12, tie, Ustat, gc.mu, cc, P.alternative);
            fc = fold_change_device((double)vsum, ref_sum - (double)vsum, gc); // math.py:185-188
        } else {
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
    };

    // Scalar row addressing and read/write counters (gather_rows / consume_rmw).  The p-values are evaluated here:
    // splitting them into a second pass (tie sums parked in the p plane) and requesting the next group's first
    // rows before the evaluation were both measured and did not pay.
    for (int g = gbeg + wave; g < gend; g += NW) {
        if (!OVR && g == P.ref) continue;
        const int p0 = __builtin_amdgcn_readfirstlane(P.pos_ptr[g]);
        const int p1 = __builtin_amdgcn_readfirstlane(P.pos_ptr[g + 1]);
        u64 S2 = 0, TT = 0;
        u32 vsum = 0;
        int p = p0;
        InT v[U];
        auto chunk = [&](auto uu, auto pred) {
            constexpr int UU = decltype(uu)::value;
            constexpr bool PRED = decltype(pred)::value;
            gather_rows<InT, UU, PRED>(Xb, row_bytes, permc, p, p1, col_bytes, v);
            if (OVR) consume_ovr<InT, RT, UU, PRED, LS>(v, p, p1, ca, S2, vsum);
            else consume_rmw<InT, RT, UU, PRED, CB, LS>(v, p, p1, ca, cb, S2, TT, vsum, bad);
            p += UU;
        };
        typedef std::integral_constant<bool, false> full_t;
        typedef std::integral_constant<bool, true> pred_t;
        while (p + U <= p1) chunk(std::integral_constant<int, U>(), full_t());
        // the remainder (< U rows): halving full chunks, then one predicated chunk of 8
        if constexpr (U > 32) { if (p + 32 <= p1) chunk(std::integral_constant<int, 32>(), full_t()); }
        if constexpr (U > 16) { if (p + 16 <= p1) chunk(std::integral_constant<int, 16>(), full_t()); }
        if (p + 8 <= p1) chunk(std::integral_constant<int, 8>(), full_t());
        if (p < p1) chunk(std::integral_constant<int, 8>(), pred_t());
        emit(g, p1 - p0, S2, TT, vsum);
        if (!OVR)
            for (int i = 0; i < BW; ++i) cbw[i * 64 + lane] = 0; // the wavefront's own block, in-order LDS: no barrier needed
    }
    if (act && bad && P.gene_flags[gene] != 3u) P.gene_flags[gene] = 1u;
    } // items
}

// ============================================================================================
// OVR from ONE pass over X.  k_fused_hist_all + k_ovo_fused<OVR> read X twice (column histogram, then rank sums).
// Here the single pass leaves, per (group, gene), the histogram of the group's values -- RT cells of CB bits, 64 B at
// CB = 8: 1.0 GB for C4's 2000 x 8000 pairs against 9.6 GB for a second read of X -- plus the column histogram;
// a second, small kernel turns histograms into rank sums:  2 ranksum(g) = sum_c h_g[c] (cum[c] + cum[c+1]) + n_g.
// Same integers as the two-pass form (dense_ovr.py:57-75, ranking.py:31-47), bit-exact.
//
// Cell (value c, gene lane) lives in the lane-PRIVATE word (c / PW) * 64 + lane of the wavefront's block, PW = 32 / CB
// cells per word: the dump to HBM is BW coalesced 256-B stores, the second kernel reads its genes' words the same
// way, and the LDS bank of a cell is the lane (conflict-free).
// CB = 0: the cell width is chosen per group -- 8 bits for a group of at most 255 cells, 16 bits above -- and a group's
// block starts at hist_off[g] words per lane: one 10 000-cell group among 2000 no longer doubles the histogram bytes of the
// 1999 small ones (C4: 2.05 GB of histograms written and read back -> 1.03 GB).
template <typename InT, int RT, int CB>
__global__ __launch_bounds__(FUSED_NT, 4) void k_ovr_group_hists(FusedParams P) {
    constexpr int NT = FUSED_NT, NW = NT / 64, BWMAX = RT * (CB ? CB : 16) / 32, U = FUSED_U;
    // The column histogram (every cell, whatever its group) is not counted per element: when a group is done its 8-bit cells are
    // added, two at a time as 16-bit fields, into hpack -- 32 LDS atomics per group and lane instead of one per cell (a
    // workgroup sees at most 64 groups x 255 cells: the fields cannot overflow); a group with 16-bit cells (more than 255
    // cells: a handful per data set) adds its cells to the global histogram directly.
    __shared__ u32 hpack[(RT / 2) * 64];   // [pair][lane]: word 2 i -> values 4 i (low half), 4 i + 2 (high half); word 2 i + 1 -> 4 i + 1, 4 i + 3
    __shared__ u32 cntB[NW][BWMAX * 64];   // per wavefront: the current group's cells
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int gene0 = blockIdx.x * 64, gene = gene0 + lane;
    const bool act = gene < P.ncols;
    const int lane_c = act ? lane : 0;
    const char *Xb = (const char *)((const InT *)P.X + P.col0 + gene0);
    const u32 row_bytes = (u32)P.ld * (u32)sizeof(InT), col_bytes = (u32)lane_c * (u32)sizeof(InT);
    const const_int_p permc = (const_int_p)P.perm;
    u32 *cbw = cntB[wave];
    // every gene of this tile already sent to the slower routes (k_fused_probe)? then there is nothing to do here.  One wavefront
    // decides for all (other workgroups may flag genes meanwhile); the answer travels through hpack[0] before that is zeroed: a
    // word of its own would cost a resident workgroup (4 x 40 KB fill the CU's 160 KB of LDS)
    if (wave == 0) {
        const bool all = __all(!act || P.gene_flags[gene] != 0);
        if (lane == 0) hpack[0] = all ? 1u : 0u;
    }
    __syncthreads();
    const u32 skip = hpack[0];
    __syncthreads();
    if (skip) return;
    for (int i = tid; i < (RT / 2) * 64; i += NT) hpack[i] = 0;
    for (int i = 0; i < BWMAX; ++i) cbw[i * 64 + lane] = 0;
    __syncthreads();
    bool bad = false;
    const int gbeg = blockIdx.y * P.groups_per_wg, gend = min(gbeg + P.groups_per_wg, P.G);
    for (int g = gbeg + wave; g < gend; g += NW) {
        const int p0 = __builtin_amdgcn_readfirstlane(P.pos_ptr[g]);
        const int p1 = __builtin_amdgcn_readfirstlane(P.pos_ptr[g + 1]);
        auto one_group = [&](auto cbt) {
            constexpr int CBG = decltype(cbt)::value, BW = RT * CBG / 32, PW = 32 / CBG;
            typedef typename CntCell<CBG>::type cell_t;
            cell_t *cells = (cell_t *)cbw + lane * PW; // cell c: cells[(c / PW) * 64 * PW + c % PW]
            int p = p0;
            InT v[U];
            auto chunk = [&](auto uu, auto pred) {
                constexpr int UU = decltype(uu)::value;
                constexpr bool PRED = decltype(pred)::value;
                gather_rows<InT, UU, PRED>(Xb, row_bytes, permc, p, p1, col_bytes, v);
#pragma unroll
                for (int u = 0; u < UU; ++u) {
                    const bool valid = !PRED || (p + u < p1);
                    bool exact;
                    const u32 c = clamp_count<InT, RT>(v[u], exact);
                    bad |= valid && !exact;
                    cell_t *cell = cells + (c / PW) * (64 * PW) + (c % PW);
                    *cell = (cell_t)(*cell + (valid ? 1u : 0u));
                }
                p += UU;
            };
            typedef std::integral_constant<bool, false> full_t;
            typedef std::integral_constant<bool, true> pred_t;
            while (p + U <= p1) chunk(std::integral_constant<int, U>(), full_t());
            if constexpr (U > 16) { if (p + 16 <= p1) chunk(std::integral_constant<int, 16>(), full_t()); }
            if (p + 8 <= p1) chunk(std::integral_constant<int, 8>(), full_t());
            if (p < p1) chunk(std::integral_constant<int, 8>(), pred_t());
            // layout [tile][group][word][lane]: the workgroup's wavefronts (consecutive groups) write neighbouring slots, and
            // k_ovr_from_hists walks a tile's groups through consecutive memory
            const size_t base = CB ? ((size_t)blockIdx.x * P.G + g) * (BW * 64) : ((size_t)blockIdx.x * P.hist_total + P.hist_off[g]) * 64;
            u32 *dst = P.group_hist + base + lane;
            // Only the words up to the last one that is non-zero in ANY of the tile's 64 genes leave (in fours): a group of ~150 cells
            // with Poisson means up to 15 reaches value ~30, half of the 64-value table -- the dump was 0.42 ms of this kernel's
            // 2.14 at C4 (1.03 GB), the words beyond hold zeros that k_ovr_from_hists need not read either.  hist_words[g][tile] says
            // how many were written.
            u32 wv[BW];
            u32 nzm = 0;
#pragma unroll
            for (int i = 0; i < BW; ++i) { wv[i] = cbw[i * 64 + lane]; nzm |= wv[i] ? (1u << i) : 0u; }
            const int nwl = nzm ? 32 - __clz(nzm) : 0;
            const int nw = P.hist_full ? BW : (__builtin_amdgcn_readlane(wave_incl_scan_max(nwl), 63) + 3) & ~3; // (hist_full: A/B switch)
#pragma unroll
            for (int q = 0; q < BW; q += 4)
                if (q < nw) { // uniform
#pragma unroll
                    for (int i = q; i < q + 4; ++i) dst[i * 64] = wv[i];
                }
            if (lane == 0) P.hist_words[(size_t)blockIdx.x * P.G + g] = (unsigned char)nw;
#pragma unroll
            for (int i = 0; i < BW; ++i) {
                const u32 w = wv[i];
                cbw[i * 64 + lane] = 0;
                if (CBG == 8) { // four 8-bit cells: values 4 i .. 4 i + 3
                    const u32 x0 = w & 0x00FF00FFu, x1 = (w >> 8) & 0x00FF00FFu;
                    if (x0) atomicAdd(&hpack[(2 * i) * 64 + lane], x0);
                    if (x1) atomicAdd(&hpack[(2 * i + 1) * 64 + lane], x1);
                } else if (act) { // two 16-bit cells: values 2 i, 2 i + 1 (a large group: straight to the global histogram)
                    if (w & 0xFFFFu) atomicAdd(&P.hist_all[(size_t)gene * RT + 2 * i], w & 0xFFFFu);
                    if (w >> 16) atomicAdd(&P.hist_all[(size_t)gene * RT + 2 * i + 1], w >> 16);
                }
            }
        };
        if constexpr (CB != 0) one_group(std::integral_constant<int, CB>());
        else if (p1 - p0 <= 255) one_group(std::integral_constant<int, 8>()); // uniform
        else one_group(std::integral_constant<int, 16>());
    }
    __syncthreads();
    for (int i = tid; i < (RT / 2) * 64; i += NT) {
        const int pr = i >> 6, l = i & 63;
        const u32 wv = hpack[i];
        const int c0 = (pr >> 1) * 4 + (pr & 1); // low half: value c0, high half: c0 + 2
        if (gene0 + l < P.ncols) {
            if (wv & 0xFFFFu) atomicAdd(&P.hist_all[(size_t)(gene0 + l) * RT + c0], wv & 0xFFFFu);
            if (wv >> 16) atomicAdd(&P.hist_all[(size_t)(gene0 + l) * RT + c0 + 2], wv >> 16);
        }
    }
    if (act && bad && P.gene_flags[gene] != 3u) P.gene_flags[gene] = 1u;
}

// histograms -> rank sums, U, p, fold change.  grid (tiles, group chunks); lane = gene.  s[c] = cum[c] + cum[c+1] sits in
// registers for the workgroup's lifetime as BYTE PLANES (four values per register): a histogram word of four 8-bit cells meets each
// plane in ONE v_dot4_u32_u8 -- 3 (NPL = 3: s < 2^24, i.e. fewer than 2^23 cells) or 4 dot products per word; the value sum is one
// more dot product against the constant bytes (4 i .. 4 i + 3).  A plane's accumulator stays below 255 x 255 x 64.  The next group's
// 16 histogram words are requested BEFORE this group's p-value is evaluated (~900 instructions with nothing in flight otherwise).
// CB == 0: histogram width per group; a group of more than 255 cells (16-bit cells, 32 words) is the rare case and reads s[c] back
// from the cumulative table in global memory instead of keeping a second register copy.
template <int RT, int CB, int NPL = 4>
__global__ __launch_bounds__(FUSED_NT, NPL == 3 ? 3 : 2) void k_ovr_from_hists(FusedParams P) {
    static_assert(CB == 8 || CB == 0, "8-bit cells throughout, or the width per group");
    constexpr int NW = FUSED_NT / 64, CSTR = RT + 1, BW8 = RT / 4;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int gene = blockIdx.x * 64 + lane;
    const bool act = gene < P.ncols;
    const int gbeg = blockIdx.y * P.groups_per_wg, gend = min(gbeg + P.groups_per_wg, P.G);
    // how many words each of this wavefront's groups left (k_ovr_group_hists): lane j <-> its j-th group, read once -- by every lane,
    // before the lanes of flagged genes leave (the values are read back with v_readlane)
    const int my_g = gbeg + wave + lane * NW;
    int nwv = my_g < gend ? (int)P.hist_words[(size_t)blockIdx.x * P.G + my_g] : 0;
    asm volatile("" : "+v"(nwv)); // (the load stays in front of the exit below: the compiler may otherwise sink it to the lanes that remain)
    if (!act || P.gene_flags[gene] != 0) return; // flagged genes are recomputed by the slower routes
    const u32 *cum = P.ref_cum + (size_t)blockIdx.x * (64 * CSTR) + lane;
    u32 bp[NPL][BW8]; // bp[k][i]: byte k of s[4 i .. 4 i + 3]
    {
        u32 prev = cum[0];
#pragma unroll
        for (int i = 0; i < BW8; ++i) {
            u32 sv[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const u32 nxt = cum[(4 * i + j + 1) * 64];
                sv[j] = prev + nxt;
                prev = nxt;
            }
#pragma unroll
            for (int k = 0; k < NPL; ++k)
                bp[k][i] = ((sv[0] >> (8 * k)) & 0xFFu) | (((sv[1] >> (8 * k)) & 0xFFu) << 8) | (((sv[2] >> (8 * k)) & 0xFFu) << 16) |
                           (((sv[3] >> (8 * k)) & 0xFFu) << 24);
        }
    }
    const u64 T_A = P.ref_TA[gene];
    const double total = (double)P.ref_sum[gene], cc = P.use_continuity ? 0.5 : 0.0;
    const double tie = !P.tie_correct ? 0.0 : (P.tie_mode ? __longlong_as_double((long long)T_A) : (double)T_A);
    auto narrow = [&](int g) { return CB == 8 || P.counts[g] <= 255; }; // uniform
    auto hist_of = [&](int g, int bw) {
        const size_t base = CB ? ((size_t)blockIdx.x * P.G + g) * (BW8 * 64) : ((size_t)blockIdx.x * P.hist_total + P.hist_off[g]) * 64;
        (void)bw;
        return P.group_hist + base + lane;
    };
    auto words_of = [&](int g) {
        const int j = (g - gbeg - wave) / NW; // uniform
        return j < 64 ? __builtin_amdgcn_readlane(nwv, __builtin_amdgcn_readfirstlane(j)) : (int)P.hist_words[(size_t)blockIdx.x * P.G + g];
    };
    u32 wpre[BW8];
    auto prefetch = [&](int g) {
        if (g < gend && narrow(g)) {
            const u32 *h = hist_of(g, BW8);
            const int nw = words_of(g);
#pragma unroll
            for (int q = 0; q < BW8; q += 4) {
                if (q < nw) { // uniform
#pragma unroll
                    for (int i = q; i < q + 4; ++i) wpre[i] = h[i * 64];
                } else {
#pragma unroll
                    for (int i = q; i < q + 4; ++i) wpre[i] = 0u;
                }
            }
        }
    };
    prefetch(gbeg + wave);
    for (int g = gbeg + wave; g < gend; g += NW) {
        const long long n_tgt = P.counts[g];
        u64 R2 = 0;
        u32 vsum = 0;
        if (narrow(g)) {
            u32 w[BW8];
#pragma unroll
            for (int i = 0; i < BW8; ++i) w[i] = wpre[i];
            prefetch(g + NW);
            u32 acc[NPL];
#pragma unroll
            for (int k = 0; k < NPL; ++k) acc[k] = 0;
#pragma unroll
            for (int i = 0; i < BW8; ++i) {
#pragma unroll
                for (int k = 0; k < NPL; ++k) acc[k] = __builtin_amdgcn_udot4(w[i], bp[k][i], acc[k], false);
                vsum = __builtin_amdgcn_udot4(w[i], (u32)(4 * i) * 0x01010101u + 0x03020100u, vsum, false);
            }
#pragma unroll
            for (int k = 0; k < NPL; ++k) R2 += (u64)acc[k] << (8 * k);
        } else {
            prefetch(g + NW);
            const u32 *h = hist_of(g, RT / 2);
            const int nw = words_of(g);
            u32 prev = cum[0];
#pragma unroll 4
            for (int i = 0; i < RT / 2; ++i) {
                const u32 wv = i < nw ? h[i * 64] : 0u;
#pragma unroll
                for (int k = 0; k < 2; ++k) {
                    const u32 nxt = cum[(2 * i + k + 1) * 64];
                    const u32 cnt = __builtin_amdgcn_ubfe(wv, k * 16, 16);
                    R2 += (u64)cnt * (prev + nxt);
                    vsum += cnt * (u32)(2 * i + k);
                    prev = nxt;
                }
            }
        }
        // dense_ovr.py:57-75, as in k_ovo_fused<OVR>
        const long long n_rest = P.n_cells - n_tgt;
        const long long two_u = 2ll * n_rest * n_tgt + n_tgt * (n_tgt + 1) - ((long long)R2 + n_tgt);
        const double Ustat = 0.5 * (double)two_u;
        const GroupConst gc = P.gconst[g]; // (uniform address)
        const double pv = pval_device_pre(gc.nnn, gc.var0, gc.n12, tie, Ustat, gc.mu, cc, P.alternative);
        const double fc = fold_change_device((double)vsum, total - (double)vsum, gc); // math.py:185-188
        const size_t o = (size_t)g * P.out_ld + gene;
        P.out_p[o] = pv;
        P.out_u[o] = Ustat;
        P.out_fc[o] = fc;
    }
}
