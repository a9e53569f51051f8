// Dense input, any values: group-wise packing of the key rows (OVO and OVR) + look-ups in a counted bitmap of the reference (OVO).
//
// The two-pass route of kernels_ovo.h transposes the whole window (zeros included) and lets one wavefront prove every
// group's keys distinct, compact them and look them up.  Expression matrices are mostly zeros, and a group's zeros need
// no look-up at all: they are one run whose rank follows from three counts.  So here
//
//   k_group_compact     one workgroup per (block of consecutive groups, 64 genes): reads the block's rows of X once (256-byte
//                       row segments, 64 rows at a time through LDS, two chunks in flight), converts to keys and writes, per
//                       gene, the groups' NON-ZERO keys back to back into the block's region of the gene's key row, as whole
//                       256-byte stores out of an LDS staging piece; beside them, per (gene, group): the number of non-zeros
//                       (16 bit), where they start, and the group's value sum (float64, fixed order: deterministic).  HBM:
//                       X once in, the non-zeros once out.  PACK = false keeps every key (dense OVR's transposition with the
//                       group sums folded in); dense OVR's partition (kernels_csc_ovr.h) walks either form.
//   k_ovo_rank_compact  one workgroup per gene: the reference's non-zero keys are dealt into value buckets in LDS -- 2^17
//                       buckets at half a byte each: per 16 buckets one 64-bit word of 2-bit counters + 16-bit prefix --, then
//                       one wavefront per group looks every non-zero key up (one table word, three keys, six compares):
//                       S2 += 2 #A<q + #A==q.  Ties inside a group are found, not assumed away: a one-word-per-key Bloom
//                       table flags keys that MAY repeat an earlier key of the group; each flagged key is then compared with
//                       all keys of the group (ballots), which yields its exact multiplicity.  No sort, no compaction, no
//                       second Bloom table; integer arithmetic only => bit-exact statistics.  Genes whose reference keys crowd
//                       (heavy ties) are left, flagged, to k_ovo_rank over the same packed layout.  For large references
//                       (EQ = true) the bucket of a key is not a plain shift of key - kmin but follows the reference's
//                       distribution: 256 coarse cells, each with a power-of-two share of the buckets in proportion to its keys.
//
// Replaces the same reference code as kernels_ovo.h: dense_ovo_mwu_kernel_over_contiguous_col_chunk (illico/ovo/dense_ovo.py:
// 65-137), i.e. the per-column sorts (utils/ranking.py:161-172) and rank_sum_and_ties_from_sorted (utils/ranking.py:52-158),
// and the group sums of utils/math.py:27-39.  Formulas as in kernels_ovo.h: with t = #A==q + (earlier equal keys of the group),
//   S2 = sum_b [2 #A<b + #A==b],   tie_sum = T_A(non-zeros) + 3 sum_b t (t + 1) + (z_A + z_B)^3 - (z_A + z_B).
#pragma once
#include "common.h"

#define GCMP_NT 512
#ifndef OCR_NT
#define OCR_NT 1024
#endif
#define OCR_KMAX 4          // 64-key rounds per group: groups of up to 256 non-zero keys
#ifndef OCR_PAIR
#define OCR_PAIR 1          // groups whose keys are requested ahead of their look-ups.  2 (requests two look-up rounds ahead) measured the same at C2 / C5
                            // shapes and 20 % slower in the 256-thread form: what the wavefronts wait on is the chain table word -> keys in LDS, not HBM
#endif
#ifndef OCR_BLOOM_WORDS
#define OCR_BLOOM_WORDS 256 // per wavefront: 8192 bits
#endif

#define OCR_COOP_MIN 8192   // k_ovo_rank_compact: a ranked group's run of more keys than this is walked by all the wavefronts of the gene's workgroup
#define OCR_CUTS 15         // ... between the cuts the bucket kernels leave for such a run (sixteen stretches)
#define GCMP_SEG_ROWS 512    // the reference group is packed in independent segments of this many rows (one workgroup each)
#ifndef GCMP_BLOCK_ROWS
#define GCMP_BLOCK_ROWS 1024
#endif // other groups: consecutive groups of at least this many rows together share one workgroup
static inline int gcmp_ref_segments(long long n_ref) { return (int)((n_ref + GCMP_SEG_ROWS - 1) / GCMP_SEG_ROWS); }

// Packed key layout of one gene (pk_stride keys): [block 0 | block 1 | ... | reference segments].  A block = consecutive groups
// (never the reference); its region starts at a multiple of 64 keys and holds the groups' non-zero keys back to back, group
// after group: one workgroup writes them front to back, so its stores of successive 64-row chunks land next to each other
// and meet in L2 as whole lines.  gofs[gene][g] = where group g's keys start, nnz[gene][g] = how many.
struct GroupCompactParams {
    const void *X;          // row-major [N, ld]
    long long ld, col0;
    int ncols;
    const int *perm;        // cells in group-contiguous order (GroupContainer.indices)
    const int *pos_ptr;     // [G+1] first position of each group in that order
    const int *blk_g0;      // [nblk] first group of each block (blocks never contain the reference)
    const int *blk_g1;      // [nblk] one past its last group
    const int *blk_out;     // [nblk] first key slot of each block
    int G, ref, nseg, nblk; // nseg = gcmp_ref_segments(reference size); grid.x = pad8(nseg) + nblk
    int ref_out;            // first key slot of the reference's segments (segment s at ref_out + s * GCMP_SEG_ROWS)
    void *Xt;               // keys, gene-major, xt_stride keys per gene
    long long xt_stride;
    u16 *nnz;               // [ncols][G] non-zero keys per (gene, group); the reference's entry is not written
    u32 *gofs;              // [ncols][G] first key slot of (gene, group); the reference's entry is not written
    u32 *blk_cnt;           // optional [ncols][nblk]: keys written per (gene, block) (dense OVR walks the packed rows block by block)
    double *out_sum;        // [ncols][G] value sums (expm1'd if LOG1P); the reference's entry is not written
    u16 *seg_nnz;           // [ncols][nseg] non-zero keys per segment of the reference
    double *seg_sum;        // [ncols][nseg]
    const int *cand_of;     // optional [G]: a group's place among the n_cand groups of more than 256 cells, or -1 ...
    u32 *run_n;             // ... and [ncols][n_cand]: the exact length of each such (gene, group) run (nnz saturates at 65535)
    int n_cand;
    const int *blk_order;   // optional [nblk]: the blocks by falling row count -- the launch then starts every tile of the longest blocks first (a long
                            // block's chain of chunks is what a launch with blocks of very different lengths waits for)
};

template <typename InT> __device__ __forceinline__ double gcmp_value(InT v, int is_log1p);
template <> __device__ __forceinline__ double gcmp_value<float>(float v, int is_log1p) { return is_log1p ? (double)expm1f(v) : (double)v; }
template <> __device__ __forceinline__ double gcmp_value<double>(double v, int is_log1p) { return is_log1p ? expm1(v) : v; }
template <> __device__ __forceinline__ double gcmp_value<int32_t>(int32_t v, int is_log1p) { return is_log1p ? expm1((double)v) : (double)v; }
template <> __device__ __forceinline__ double gcmp_value<int64_t>(int64_t v, int is_log1p) { return is_log1p ? expm1((double)v) : (double)v; }

// One workgroup: one block of groups (or one segment of the reference) x 64 genes.  The block's rows go by in 64-row chunks,
// group after group (a group's last chunk is padded with zero rows): thread (q, r0) loads VEC genes of rows r0, r0 + RPI, ...;
// keys go through a gene-major LDS tile; wavefront w then packs genes 16 w ... 16 w + 15: lanes = rows, ballot -> consecutive
// output slots after the keys already written.  The next chunk's rows (and the row indices of the chunk after it) are in
// flight meanwhile, across group boundaries.  Value sums: per-thread partials over the thread's own loads, combined through LDS
// in a fixed order at each group's end.
// PACK = false (dense OVR): every key of the block's rows is kept, zeros included -- the "padded dense" layout: the key rows of
// kernels_ovo.h with each block starting at a multiple of 64 keys (holes hold the zero key) -- and only the sums are written
// beside the keys: the transposition with the group sums folded in.
template <typename InT, typename KeyT, bool VECLOAD, bool LOG1P, bool PACK = true, int TW = 64>
__global__ __launch_bounds__(GCMP_NT) void k_group_compact(GroupCompactParams P) {
    constexpr int VEC = 16 / (int)sizeof(InT);
    constexpr int LPR = TW / VEC;      // lanes per TW-gene row segment
    constexpr int RPI = GCMP_NT / LPR; // rows per load iteration
    constexpr int NLD = 64 / RPI;      // loads per thread per 64-row chunk
    constexpr KeyT ZEROK = KeyInfo<KeyT>::ZEROK;
    typedef InT __attribute__((ext_vector_type(VEC))) InV;
    constexpr int TILE_KEYS = (TW * 65 * (int)sizeof(KeyT) >= (int)sizeof(double) * RPI * TW) ? TW * 65 : (int)(sizeof(double) * RPI * TW / sizeof(KeyT));
    __shared__ KeyT tile_mem[TILE_KEYS]; // [TW][65] keys; doubles as the sum scratch ([RPI][TW] doubles)
    KeyT (*tile)[65] = reinterpret_cast<KeyT (*)[65]>(tile_mem);

    const int nseg_pad = (P.nseg + 7) & ~7;
    int gA, gB, out0, seg = -1, seg_row0 = 0, seg_n = 0, blk = 0;
    // (virtual) grid position: x = segment / block, y = gene tile.  With blk_order the workgroups are numbered block-major, longest block first
    int vx = (int)blockIdx.x, vy = (int)blockIdx.y;
    if (P.blk_order) {
        const int tiles = (int)gridDim.y, L = (int)blockIdx.y * (int)gridDim.x + (int)blockIdx.x, nbt = P.nblk * tiles;
        if (L < nbt) { vx = nseg_pad + P.blk_order[L / tiles]; vy = L - (L / tiles) * tiles; }
        else { vx = (L - nbt) / tiles; vy = (L - nbt) - vx * tiles; }
    }
    if (vx < nseg_pad) {
        seg = vx;
        if (seg >= P.nseg) return;
        gA = P.ref; gB = P.ref + 1;
        seg_row0 = P.pos_ptr[P.ref] + seg * GCMP_SEG_ROWS;
        seg_n = min(GCMP_SEG_ROWS, P.pos_ptr[P.ref + 1] - seg_row0);
        out0 = P.ref_out + seg * GCMP_SEG_ROWS;
    } else {
        const int b = vx - nseg_pad;
        gA = P.blk_g0[b]; gB = P.blk_g1[b];
        out0 = P.blk_out[b];
        blk = b;
    }
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int c0 = vy * TW;
    const int q = tid % LPR, r0 = tid / LPR;
    const InT *X = (const InT *)P.X;
    KeyT *Xt = (KeyT *)P.Xt;
    const int cq = c0 + q * VEC;
    const bool colv = cq + VEC <= P.ncols;
    const int geneW = c0 + wave * (TW / (GCMP_NT / 64)) + lane; // lanes 0..GPW-1: the gene whose counts this lane keeps

    // chunk cursors (uniform scalars): group, chunk inside it, the group's rows and first position -- one for the loads being
    // issued (i*), one for the chunk being packed (c*)
    // a block of fewer than 64 groups (the usual case) keeps its group boundaries in one register, lane l = pos_ptr[gA + l]: moving to
    // the next group then reads two lanes instead of waiting for two loads
    const int ngb = gB - gA;
    const bool pp_in_lanes = seg < 0 && ngb < 64;
    const int ppv = pp_in_lanes ? P.pos_ptr[gA + min(lane, ngb)] : 0;
    auto grp_rows = [&](int g, int &n, int &row0) {
        if (g >= gB) { n = 0; row0 = 0; }
        else if (seg >= 0) { row0 = seg_row0; n = seg_n; }
        else if (pp_in_lanes) {
            const int l = __builtin_amdgcn_readfirstlane(g - gA);
            row0 = __builtin_amdgcn_readlane(ppv, l);
            n = __builtin_amdgcn_readlane(ppv, l + 1) - row0;
        } else { row0 = P.pos_ptr[g]; n = P.pos_ptr[g + 1] - row0; }
    };
#define GCMP_CUR_SKIP(g, c, n, row0) while (g < gB && n == 0) { ++g; c = 0; grp_rows(g, n, row0); }
#define GCMP_CUR_NEXT(g, c, n, row0) { ++c; if (c * 64 >= n) { ++g; c = 0; grp_rows(g, n, row0); GCMP_CUR_SKIP(g, c, n, row0) } }
    if (seg < 0) { // groups without cells have no chunk: their outputs here
        for (int g = gA; g < gB; ++g)
            if (P.pos_ptr[g + 1] == P.pos_ptr[g] && tid < TW && c0 + tid < P.ncols) {
                if (PACK) {
                    P.nnz[(size_t)(c0 + tid) * P.G + g] = 0;
                    P.gofs[(size_t)(c0 + tid) * P.G + g] = (u32)out0;
                }
                P.out_sum[(size_t)(c0 + tid) * P.G + g] = 0.0;
            }
    }

    // staging of packed keys (4-byte keys): per gene a 128-key ring piece in LDS; whenever 64 keys are there, one 256-byte
    // aligned store of full lines goes out.  (Pieces of a line written one store at a time make the L2 fetch the line first.)
    constexpr bool STAGE = sizeof(KeyT) == 4;
    constexpr int NWV = GCMP_NT / 64, GPW = TW / NWV; // wavefronts, genes packed by each
    __shared__ KeyT stage[STAGE ? NWV * GPW * 128 : 1];
    KeyT *st = stage + (STAGE ? wave * GPW * 128 : 0);

    int rows[NLD]; // row indices of the next chunk to load
    InV bufA[NLD], bufB[NLD];
    auto load_rows = [&](int c, int n, int row0) {
#pragma unroll
        for (int i = 0; i < NLD; ++i) {
            const int p = c * 64 + r0 + i * RPI;
            rows[i] = p < n ? P.perm[row0 + p] : -1;
        }
    };
    auto load_chunk = [&](InV (&buf)[NLD]) { // from rows[]
#pragma unroll
        for (int i = 0; i < NLD; ++i) {
            InV v;
#pragma unroll
            for (int e = 0; e < VEC; ++e) v[e] = (InT)0;
            if (rows[i] >= 0) {
                const InT *src = X + (long long)rows[i] * P.ld + P.col0 + cq;
                if (VECLOAD && colv) v = *reinterpret_cast<const InV *>(src);
                else {
#pragma unroll
                    for (int e = 0; e < VEC; ++e)
                        if (cq + e < P.ncols) v[e] = src[e];
                }
            }
            buf[i] = v;
        }
    };

    int cntv = 0, gstartv = 0; // lane i < GPW: keys written so far for gene GPW * wave + i; ... when the current group began
    double sum[VEC];           // per-thread partial value sums of its VEC genes, current group
#pragma unroll
    for (int e = 0; e < VEC; ++e) sum[e] = 0.0;
    const u64 lt_mask = (lane == 0) ? 0ull : (~0ull >> (64 - lane));
    int ig = gA, ic = 0, in_, irow0;
    grp_rows(ig, in_, irow0);
    GCMP_CUR_SKIP(ig, ic, in_, irow0)
    int cg = ig, cc = 0, cn = in_, crow0 = irow0;
    bool rows_ready = false;
    // two chunks of rows in flight (bufA: the chunk packed next, bufB: the one after), row indices of a third.  (Four in flight with the
    // 32-gene tiles: 6.1 -> 8.8 ms on ten clusters of 100 000 cells -- a step's own chain of waits, not the gather, is what a chunk takes.)
    if (ig < gB) { load_rows(ic, in_, irow0); GCMP_CUR_NEXT(ig, ic, in_, irow0) load_chunk(bufA); }
    if (ig < gB) { load_rows(ic, in_, irow0); GCMP_CUR_NEXT(ig, ic, in_, irow0) load_chunk(bufB); }
    if (ig < gB) { load_rows(ic, in_, irow0); GCMP_CUR_NEXT(ig, ic, in_, irow0) rows_ready = true; }
    auto step = [&](InV (&buf)[NLD]) { // pack the chunk in buf, then refill buf with the chunk rows[] describes
#pragma unroll
        for (int i = 0; i < NLD; ++i) {
            const int r = r0 + i * RPI;
#pragma unroll
            for (int e = 0; e < VEC; ++e) {
                const InT v = buf[i][e];
                tile[q * VEC + e][r] = key_of(v);
                sum[e] += gcmp_value<InT>(v, LOG1P ? 1 : 0); // rows past the group's end were loaded as zeros
            }
        }
        if (rows_ready) { load_chunk(buf); rows_ready = false; }
        if (ig < gB) { load_rows(ic, in_, irow0); GCMP_CUR_NEXT(ig, ic, in_, irow0) rows_ready = true; }
        __syncthreads();
#pragma unroll 4
        for (int i = 0; i < GPW; ++i) {
            const int gi = wave * GPW + i;
            const KeyT k = tile[gi][lane];
            const bool nz = PACK ? k != ZEROK : lane < cn - cc * 64; // (rows past the group's end were loaded as zeros)
            const u64 m = __ballot(nz);
            const int ci = __builtin_amdgcn_readlane(cntv, i);
            KeyT *dst = Xt + (long long)(c0 + gi) * P.xt_stride + out0; // uniform
            const int pos = (int)__popcll(m & lt_mask), nk = (int)__popcll(m);
            if constexpr (STAGE) {
                const int fill = ci & 63; // out0 is a multiple of 64: the staged keys start a 256-byte piece
                KeyT *sg = st + i * 128;
                if (nz) sg[fill + pos] = k;
                if (fill + nk >= 64) { // uniform
                    wave_lds_fence();
                    const KeyT full = sg[lane], over = sg[64 + lane];
                    if (c0 + gi < P.ncols) dst[(ci - fill) + lane] = full;
                    wave_lds_fence();
                    if (lane < fill + nk - 64) sg[lane] = over;
                }
            } else {
                if (nz && c0 + gi < P.ncols) dst[ci + pos] = k;
            }
            cntv += lane == i ? nk : 0;
        }
        __syncthreads();
        if ((cc + 1) * 64 >= cn) { // the group's last chunk: its sums ([RPI row slots][64 genes] partials, added in row-slot order), counts, start
            double *part = (double *)tile_mem;
#pragma unroll
            for (int e = 0; e < VEC; ++e) { part[r0 * TW + q * VEC + e] = sum[e]; sum[e] = 0.0; }
            __syncthreads();
            if (tid < TW && c0 + tid < P.ncols) {
                double tot = 0.0;
#pragma unroll 4
                for (int r = 0; r < RPI; ++r) tot += part[r * TW + tid];
                if (seg >= 0) P.seg_sum[(size_t)(c0 + tid) * P.nseg + seg] = tot;
                else P.out_sum[(size_t)(c0 + tid) * P.G + cg] = tot;
            }
            if (PACK && lane < GPW && geneW < P.ncols) {
                if (seg >= 0) P.seg_nnz[(size_t)geneW * P.nseg + seg] = (u16)cntv;
                else {
                    P.nnz[(size_t)geneW * P.G + cg] = (u16)min(cntv - gstartv, 65535); // (saturating: the exact length of a long run is in run_n)
                    P.gofs[(size_t)geneW * P.G + cg] = (u32)(out0 + gstartv);
                    if (P.run_n) { // (uniform: cg is the workgroup's current group)
                        const int cd = P.cand_of[cg];
                        if (cd >= 0) P.run_n[(size_t)geneW * P.n_cand + cd] = (u32)(cntv - gstartv);
                    }
                }
            }
            gstartv = cntv;
            __syncthreads();
        }
        GCMP_CUR_NEXT(cg, cc, cn, crow0)
    };
    while (cg < gB) {
        step(bufA);
        if (cg < gB) step(bufB);
    }
    if (PACK && seg < 0 && P.blk_cnt && lane < GPW && geneW < P.ncols) P.blk_cnt[(size_t)geneW * P.nblk + blk] = (u32)cntv;
    if constexpr (STAGE) { // what is left in the staging pieces: one partial store per gene
        wave_lds_fence();
#pragma unroll 4
        for (int i = 0; i < GPW; ++i) {
            const int gi = wave * GPW + i;
            const int ci = __builtin_amdgcn_readlane(cntv, i), fill = ci & 63;
            if ((PACK ? lane < fill : fill > 0) && c0 + gi < P.ncols) // (padded dense: the hole up to the next block holds zeros)
                Xt[(long long)(c0 + gi) * P.xt_stride + out0 + (ci - fill) + lane] = lane < fill ? st[i * 128 + lane] : ZEROK;
        }
    }
    if constexpr (!STAGE && !PACK) { // padded dense, keys stored directly: the hole up to the next block holds zeros
#pragma unroll 4
        for (int i = 0; i < GPW; ++i) {
            const int gi = wave * GPW + i;
            const int ci = __builtin_amdgcn_readlane(cntv, i), hole = (64 - (ci & 63)) & 63;
            if (lane < hole && c0 + gi < P.ncols) Xt[(long long)(c0 + gi) * P.xt_stride + out0 + ci + lane] = ZEROK;
        }
    }
#undef GCMP_CUR_SKIP
#undef GCMP_CUR_NEXT
}

// ---------------------------------------------------------------------------------------------
// per-gene ranking over the packed layout
// ---------------------------------------------------------------------------------------------
struct OvoCompactParams {
    void *Xs;                // packed keys (k_group_compact)
    long long gene_stride;
    const int *counts;       // [G]
    u16 *nnz;                // [n_genes][G]
    u32 *gofs;               // [n_genes][G] first key slot of each group
    int ref_out;             // first key slot of the reference's segments
    const u16 *seg_nnz;      // [n_genes][nseg] the reference's segments (k_group_compact)
    const double *seg_sum;   // [n_genes][nseg]
    double *out_sum;         // [n_genes][G]: the reference's entry is written here (sum of its segments, in order)
    int G, ref, n_genes, nseg;
    int ref_cap;             // LDS key slots for the reference's non-zeros (a gene with more of them is left to k_ovo_rank)
    int nbk_lg;              // log2(value buckets), 14 .. 17
    long long *out_2u;       // [n_genes][G]
    u64 *out_tie;            // [n_genes][G]
    int big_sorted;          // 1: the runs of more than 256 non-zero keys were dealt into value buckets (k_bucket_big_runs): they are walked in
                             // pieces of at most 256 keys cut where the bucket number changes (equal keys never straddle two pieces, so the
                             // pieces' terms add up); 0: a gene with such a group leaves the kernel
    int ref_by_gofs;         // 1 (sparse input, regrouped by k_csc_regroup / k_csc_segment): the reference's keys are ONE run at gofs[gene][ref]
                             // (seg_nnz: [n_genes] its length, nseg = 1; no seg_sum: value sums are formed elsewhere); gene_stride = 0
    const u32 *gene_flags;   // optional [n_genes]: a gene whose word is 0 is somebody else's (count-valued: the histogram kernel's)
    const void *big_fn;      // [n_genes][n_cand] BigRunFn<KeyT>: each such run's bucket function
    const u32 *run_n;        // optional [n_genes][n_cand]: exact lengths of the runs of the groups above 256 cells (nnz saturates at 65535)
    const u32 *run_cuts;     // optional [n_genes][n_cand][OCR_CUTS]: for runs above OCR_COOP_MIN keys, bucket boundaries at or behind w / 16 of the run (the bucket kernels)
    const void *big_tmp;     // the second key buffer (laid out like Xs) that holds the runs k_bucket_big_runs_global dealt
    const int *cand_of;      // [G] a group's place among the n_cand groups of more than 256 cells, or -1
    int n_cand;
    u32 *needs_parts;        // optional [n_genes], zeroed by the host.  The plain kernel sets the word of a gene whose reference has more non-zero keys
                             // than slots and leaves the gene alone; the PARTS kernel, launched behind it, takes exactly those genes
    const void *cuts;        // PARTS: [n_genes] PartCuts<KeyT> (k_ref_cuts): the parts of each gene that needs them
    const u16 *pofs;         // PARTS, optional: [n_genes][G][4] where parts 1 .. 3 of a dealt (gene, group) run begin (k_deal_runs; genes of at most 4 parts)
    int n_parts;             // PARTS kernels: value-range parts a gene's reference may be taken in (grid.x = n_genes * n_parts; out_2u / out_tie
                             // zeroed by the host: every part ADDS its terms)
    u32 *route;              // [n_genes], zeroed by the host: set to 1 for the genes this kernel leaves to k_ovo_rank (packed
                             // mode): crowded value buckets (a tie-heavy column: it wants the sorted reference and the sort form
                             // of the group loop) or a group of more than 256 non-zeros.  For those the reference's segments are
                             // moved together and nnz / gofs[gene][ref] are set.
};
// a gene leaves this kernel when one table word (16 buckets) holds more than OCR_MAX_WORD reference keys, or when more than half
// of the reference's keys sit in words with an overfull bucket (such words are walked key by key: exact, but slow)
#define OCR_MAX_WORD 64

// The reference's non-zero keys in LDS.  Value buckets (key - kmin) >> shift, 2^nbk_lg of them; keys are stored in bucket order
// (any order inside a bucket).  The table costs HALF A BYTE per bucket: one 64-bit word per 16 buckets,
//     low half  : 16 two-bit counters = keys in each of the 16 buckets (0 .. 3)
//     high half : keys in all earlier words (16 bit); bit 31 = some bucket of this word holds more than 3 keys ("overfull word")
// so a bucket's first key is at  prefix + (sum of the counters below it)  = two popcounts, and with ~30 buckets per key nearly
// every bucket holds 0 or 1 keys: a look-up reads one table word and two keys.  Overfull words (and a bucket of exactly 3, and
// keys that tie with the reference) take an exact per-lane walk over the word's / bucket's keys.
__host__ __device__ static inline size_t ocr_lds_bytes(int ref_cap, int nbk_lg, size_t key_size, int nt = OCR_NT) {
    size_t b = (((size_t)ref_cap + 4) * key_size + 15) & ~(size_t)15;
    b += (((size_t)1 << nbk_lg) / 16 + 2) * 8;
    b += (size_t)(nt / 64) * OCR_BLOOM_WORDS * 4;
    b += 2048; // coarse cells of the distribution-following bucket function: table + counters
    b += 256; // reduction words
    return b;
}

__device__ __forceinline__ u32 ocr_hash(u32 k) { return k ^ (k >> 10); }
__device__ __forceinline__ u32 ocr_hash(u64 k) { const u32 f = (u32)(k ^ (k >> 32)); return f ^ (f >> 10); }

// EQ: the bucket function follows the reference's distribution.  256 coarse cells from the high bits of key - kmin; cell c owns
// 2^e[c] consecutive fine buckets, e[c] chosen from the cell's key count so that a crowded stretch of values is spread over as many
// buckets per key as a sparse one (ctab[c] = first fine bucket | (cshift - e[c]) << 24).  Monotone, like the plain shift: nothing
// downstream changes.  Costs one more 4-byte LDS read per look-up; used for large references, where the plain function would leave
// the crowded stretch with buckets of four and more keys (whole table words then fall back to key-by-key walks).
template <typename KeyT, bool EQ> struct OcrRef {
    const KeyT *A;
    const u32 *tab; // word W: tab[2 W] counters, tab[2 W + 1] prefix | overfull flag; one sentinel word (prefix = all keys)
    KeyT kmin;
    int shift;
    u32 last; // buckets - 1
    u32 nA;   // keys
    const u32 *ctab; // EQ: [256]
    int cshift;      // EQ
    KeyT cmask;      // EQ: (1 << cshift) - 1
};
template <typename KeyT, bool EQ> __device__ __forceinline__ u32 ocr_bucket(const OcrRef<KeyT, EQ> &R, KeyT q) {
    const KeyT d = q > R.kmin ? (KeyT)(q - R.kmin) : (KeyT)0;
    if constexpr (EQ) {
        const KeyT c = d >> R.cshift;
        if (c > (KeyT)255) return R.last; // above every reference key
        const u32 t = R.ctab[(u32)c];
        const u32 f = (t & 0xFFFFFFu) + (u32)((KeyT)(d & R.cmask) >> (t >> 24));
        return f < R.last ? f : R.last;
    } else {
        const KeyT b = d >> R.shift;
        return b < (KeyT)R.last ? (u32)b : R.last;
    }
}
// keys in the buckets below position sh (= 2 * bucket-in-word) of a counter word: each 2-bit value = b0 + 2 b1 = (b0 + b1) + b1
__device__ __forceinline__ u32 ocr_below(u32 w, u32 sh) {
    const u32 x = w & ((1u << sh) - 1u);
    return (u32)__popc(x) + (u32)__popc(x & 0xAAAAAAAAu);
}
// exact look-up: the keys that can be < q without being counted by the word's prefix lie in [lo, hi): the bucket, or the
// whole word when it is overfull (the order of its keys is arbitrary then)
template <typename KeyT, bool EQ> __device__ __forceinline__ void ocr_find_exact(const OcrRef<KeyT, EQ> &R, KeyT q, u32 &less, u32 &eq) {
    const u32 b = ocr_bucket(R, q), W = b >> 4, sh = (b & 15u) << 1;
    const u32 w = R.tab[2 * W], h = R.tab[2 * W + 1];
    u32 lo, hi;
    if (h >> 31) { lo = h & 0xFFFFu; hi = R.tab[2 * W + 3] & 0xFFFFu; }
    else { lo = h + ocr_below(w, sh); hi = lo + ((w >> sh) & 3u); }
    u32 l = lo, a = 0;
    for (u32 t = lo; t < hi; ++t) { const KeyT k = R.A[t]; l += k < q ? 1u : 0u; a += k == q ? 1u : 0u; }
    less = l;
    eq = a;
}

// One group of nB non-zero keys, NR = ceil(nB / 64) rounds of 64 (cur[r] = key r * 64 + lane; lanes past the last key hold
// ZEROK).  Straight-line per round: Bloom insert (one returning LDS atomic), one table word, 2 keys, 4 compares.
// Per-lane partial results: less = sum of #A<q (non-zero reference keys), eqs = sum of #A==q, TT = sum t (t + 1); negs
// (uniform) = keys below zero.  bloom[] is all-zero on entry and on exit.
// MASKED (the reference taken in value-range parts: a group's keys outside the part were replaced by ZEROK, which is no packed key):
// a lane is valid where its key is not ZEROK, in every round; else the lanes past the last key (of the last round only) are the invalid ones.
template <typename KeyT, int NR, bool EQ, bool MASKED = false>
__device__ __forceinline__ void ocr_group(const KeyT (&cur)[OCR_KMAX], int nB, const OcrRef<KeyT, EQ> &R, u32 *bloom, int lane, u64 lt_mask,
                                          u32 &less_out, u32 &eq_out, u64 &TT_out, u32 &negs_out) {
    constexpr KeyT ZEROK = KeyInfo<KeyT>::ZEROK;
    const int rem = nB - 64 * (NR - 1);               // keys of the last round, 1..64
    const u64 vlast = rem >= 64 ? ~0ull : ((1ull << rem) - 1ull);
    const bool vl = (vlast >> lane) & 1ull;
    bool val[NR];
#pragma unroll
    for (int r = 0; r < NR; ++r) val[r] = MASKED ? cur[r] != ZEROK : (r < NR - 1 || vl);
    u32 wofs[NR], lessr[NR];
    u64 fm[NR], eqr[NR], ovr_[NR], eqm = 0, ovm = 0;
#pragma unroll
    for (int r = 0; r < NR; ++r) { // Bloom inserts of every round first: a key's flag says "may repeat an EARLIER key"
        const KeyT q = cur[r];
        const u32 h = ocr_hash(q);
        wofs[r] = (h >> 5) & (OCR_BLOOM_WORDS - 1);
        bool flag = false;
        if (val[r]) {
            const u32 old = atomicOr(&bloom[wofs[r]], 1u << (h & 31));
            flag = (old >> (h & 31)) & 1u;
        }
        fm[r] = __ballot(flag);
    }
#pragma unroll
    for (int r = 0; r < NR; ++r) {
        const KeyT q = cur[r];
        const u32 b = ocr_bucket(R, q), sh = (b & 15u) << 1;
        const u32 *pw = R.tab + 2 * (b >> 4);
        const u32 w = pw[0], h = pw[1];
        const u32 lo = min((h + ocr_below(w, sh)) & 0xFFFFu, R.nA); // meaningless in an overfull word (bit 31 of h): redone below
        const KeyT a0 = R.A[lo], a1 = R.A[lo + 1], a2 = R.A[lo + 2]; // a bucket holds at most 3; past it: later buckets / the pad: > q or == MAXK
        if (val[r]) bloom[wofs[r]] = 0u;       // wipe (LDS operations of one wavefront execute in order)
        u32 l = lo + (a0 < q ? 1u : 0u) + (a1 < q ? 1u : 0u) + (a2 < q ? 1u : 0u);
        u64 e4 = __ballot(a0 == q) | __ballot(a1 == q) | __ballot(a2 == q), o4 = __ballot((int)h < 0), n4 = __ballot(q < ZEROK);
        if (MASKED) { const u64 vm = __ballot(val[r]); l = val[r] ? l : 0u; e4 &= vm; o4 &= vm; n4 &= vm; }
        else if (r == NR - 1) { l = vl ? l : 0u; e4 &= vlast; o4 &= vlast; n4 &= vlast; }
        lessr[r] = l;
        eqr[r] = e4; ovr_[r] = o4;
        eqm |= e4; ovm |= o4;
        negs_out += (u32)__popcll(n4);
    }
    u32 a[NR], eqs = 0, less = 0;
    u64 TT = 0;
#pragma unroll
    for (int r = 0; r < NR; ++r) a[r] = 0u;
    if (eqm) { // a key that ties with the reference somewhere in the group: count, bounded by the bucket (the pad's MAXK never counts)
#pragma unroll
        for (int r = 0; r < NR; ++r) {
            if (eqr[r]) { // uniform; the round's table word and keys are read again here rather than kept in registers all along
                const KeyT q = cur[r];
                const u32 b = ocr_bucket(R, q), sh = (b & 15u) << 1;
                const u32 w = R.tab[2 * (b >> 4)], h = R.tab[2 * (b >> 4) + 1], c = (w >> sh) & 3u;
                const u32 lo = min((h + ocr_below(w, sh)) & 0xFFFFu, R.nA);
                const u32 e = ((c > 0u && R.A[lo] == q) ? 1u : 0u) + ((c > 1u && R.A[lo + 1] == q) ? 1u : 0u) + ((c > 2u && R.A[lo + 2] == q) ? 1u : 0u);
                a[r] = val[r] ? e : 0u;
            }
        }
    }
    if (ovm) { // a key in an overfull word (its keys lie in [wlo, whi) in any order): the first 8 in line, the rest lane by lane
#pragma unroll
        for (int r = 0; r < NR; ++r) {
            if (!ovr_[r]) continue; // uniform
            const KeyT q = cur[r];
            const u32 W = ocr_bucket(R, q) >> 4;
            const u32 h = R.tab[2 * W + 1];
            const bool ov = val[r] && (int)h < 0;
            const u32 wlo = min(h & 0xFFFFu, R.nA), whi = ov ? (R.tab[2 * W + 3] & 0xFFFFu) : wlo;
            u32 l = wlo, e = 0;
#pragma unroll
            for (u32 t = 0; t < 8; ++t) {
                const KeyT k = R.A[wlo + t]; // (the pad is 4 keys: reads past it stay inside the table's LDS and are masked out)
                l += (wlo + t < whi && k < q) ? 1u : 0u;
                e += (wlo + t < whi && k == q) ? 1u : 0u;
            }
            for (u32 t = wlo + 8u; t < whi; ++t) { const KeyT k = R.A[t]; l += k < q ? 1u : 0u; e += k == q ? 1u : 0u; }
            if (ov) { lessr[r] = l; a[r] = e; }
        }
    }
#pragma unroll
    for (int r = 0; r < NR; ++r) less += lessr[r];
    if (eqm | ovm) {
#pragma unroll
        for (int r = 0; r < NR; ++r) {
            eqs += a[r];
            TT += (u64)a[r] * ((u64)a[r] + 1ull); // (a <= 65535)
        }
    }
    u64 anyf = 0;
#pragma unroll
    for (int r = 0; r < NR; ++r) anyf |= fm[r];
    if (anyf) { // keys that may repeat an earlier key of the group: exact multiplicities by comparison
#pragma unroll
        for (int r = 0; r < NR; ++r) {
            while (fm[r]) {
                const int sl = __ffsll((long long)fm[r]) - 1;
                KeyT q;
                if constexpr (sizeof(KeyT) == 8) {
                    q = ((u64)(u32)__builtin_amdgcn_readlane((int)(u32)(cur[r] >> 32), sl) << 32) | (u64)(u32)__builtin_amdgcn_readlane((int)(u32)cur[r], sl);
                } else q = (KeyT)__builtin_amdgcn_readlane((int)cur[r], sl);
                u64 m[NR];
                int c = 0;
#pragma unroll
                for (int s = 0; s < NR; ++s) {
                    m[s] = __ballot(cur[s] == q); // q is a non-zero key: the ZEROK lanes past the end never match
                    c += (int)__popcll(m[s]);
                }
                if (c > 1) {
                    int before = 0;
#pragma unroll
                    for (int s = 0; s < NR; ++s) {
                        if (cur[s] == q) {
                            const u64 o = (u64)(before + (int)__popcll(m[s] & lt_mask));
                            TT += o * (2ull * a[s] + o + 1ull); // (a + o)(a + o + 1) - a (a + 1)
                        }
                        before += (int)__popcll(m[s]);
                    }
                }
#pragma unroll
                for (int s = 0; s < NR; ++s) fm[s] &= ~m[s];
            }
        }
    }
    less_out = less;
    eq_out = eqs;
    TT_out = TT;
}


// ---- groups of more than 256 non-zero keys -----------------------------------------------------------------------------------
// The rank kernel below looks a group's keys up 256 at a time, and what it has to know about a group beyond the look-ups -- which of
// its keys repeat, how often -- it finds among those 256 keys.  A larger group (a cluster of thousands of cells: the common
// non-perturbation use; the reference's cost per element does not depend on the group size, dense_ovo.py:118-132) is therefore dealt
// into VALUE BUCKETS first, in place, one workgroup per (gene, group) run, through LDS: bucket = (key - kmin) >> shift over the run's
// own key range, about four keys per bucket, one counting-sort pass (histogram, scan, scatter).  Equal keys share a bucket, buckets
// come out in ascending order, and the rank kernel cuts its pieces where the bucket number changes (big_fn[gene][k] = {kmin, shift}
// lets it recompute the number): no value straddles two pieces, so the pieces' terms simply add up.  No sort: inside a bucket the order
// is arbitrary, which the piece-wise duplicate search does not mind.  Runs of at most 256 keys are left alone; a run longer than
// the LDS buffer sets route[gene] = 2 (the gene takes the general sort route), as a bucket of more than 256 keys does in the rank kernel.
#define SRT_NT 256
#define SRT_LG_MAX 12 // at most 4096 buckets
template <typename KeyT> __host__ __device__ constexpr int srt_cap() { return sizeof(KeyT) == 4 ? 32768 : 16384; } // at most 128 KB of keys in LDS
// LDS for runs of at most `cap` keys (the host passes the largest candidate group's cell count, capped): keys + bucket counters
static inline int srt_lg_of(int n) { int lg = 6; while (lg < SRT_LG_MAX && (4 << lg) < n) ++lg; return lg; }
static inline size_t srt_lds_bytes(size_t key_size, int cap) { return (size_t)cap * key_size + ((size_t)4 << srt_lg_of(cap)) + 64; }
#define BIG_RUN_IN_TMP (1 << 30) // in BigRunFn::shift: the dealt run lies in the second key buffer (k_bucket_big_runs_global), not in place
template <typename KeyT> struct BigRunFn { KeyT kmin; int shift; };
template <typename KeyT> __device__ __forceinline__ u32 big_bucket(const BigRunFn<KeyT> &f, KeyT k) { return (u32)((KeyT)(k - f.kmin) >> f.shift); }

// every key of a run (and its index) to f, NT threads: 16-byte loads (the run starts anywhere: up to three keys in front of the first whole piece and
// behind the last go one by one), four of them in flight per thread
template <typename KeyT, int NT, typename F> __device__ __forceinline__ void run_for_keys(const KeyT *__restrict__ run, int n, int tid, F f) {
    constexpr int V = 16 / (int)sizeof(KeyT);
    typedef KeyT __attribute__((ext_vector_type(V))) KV;
    const int head = min(n, (int)(((16u - (unsigned)((uintptr_t)run & 15u)) & 15u) / (unsigned)sizeof(KeyT)));
    if (tid < head) f(run[tid], tid);
    const KV *rv = reinterpret_cast<const KV *>(run + head);
    const int nv = (n - head) / V;
    for (int i = tid; i < nv; i += 4 * NT) {
        KV k[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) k[u] = rv[min(i + u * NT, nv - 1)];
#pragma unroll
        for (int u = 0; u < 4; ++u)
            if (i + u * NT < nv) {
#pragma unroll
                for (int e = 0; e < V; ++e) f(k[u][e], head + (i + u * NT) * V + e);
            }
    }
    const int t0 = head + nv * V;
    if (tid < n - t0) f(run[t0 + tid], t0 + tid);
}

// A run longer than the LDS buffer (half of a 100 000-cell cluster non-zero: 50 000 keys) is dealt through HBM instead, by a kernel of
// its own (k_bucket_big_runs leaves such runs alone when one follows): range and bucket counts from two reads of the run, the counters
// -- up to 2^13 in LDS, ~6 keys per bucket at 50 000 -- scanned, the keys dealt slice by slice of the output through LDS key slots behind
// the counters (all of a CU's LDS: one workgroup of 1024 threads) into the run's place in a second key buffer `tmp` (laid out like
// Xs), where the rank kernel reads it (BIG_RUN_IN_TMP in the run's bucket function: no copy back).
#define SRT_LG_MAX_G 13
#define SRTG_NT 1024
template <typename KeyT>
__global__ __launch_bounds__(SRTG_NT) void k_bucket_big_runs_global(void *Xs, void *tmp, long long gene_stride, const u16 *__restrict__ nnz, const u32 *__restrict__ gofs,
                                                                    const int *__restrict__ cand, int n_cand, int G, int cap /* runs up to here are k_bucket_big_runs' */,
                                                                    int lg_max /* log2 of the counters the launch's LDS holds */, BigRunFn<KeyT> *__restrict__ big_fn,
                                                                    u32 *__restrict__ route, const u32 *__restrict__ run_n /* exact run lengths, or null */, int slice_keys /* LDS key slots behind the counters */,
                                                                    u32 *__restrict__ run_cuts /* optional: OCR_CUTS bucket boundaries per run for the rank kernel */) {
    extern __shared__ __align__(16) unsigned char srtg_smem[];
    constexpr KeyT MAXK = KeyInfo<KeyT>::MAXK;
    constexpr int NT = SRTG_NT;
    u32 *cnt = (u32 *)srtg_smem;
    __shared__ KeyT g_min, g_max;
    __shared__ u32 g_part[NT / 64];
    __shared__ int s_list[NT], s_nl;
    const int gene = blockIdx.y, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int per = (n_cand + (int)gridDim.x - 1) / (int)gridDim.x, c_end = min(n_cand, ((int)blockIdx.x + 1) * per);
    for (int c0 = (int)blockIdx.x * per; c0 < c_end; c0 += NT) { // (the workgroup's stretch of the gene's candidates, as in k_bucket_big_runs)
      if (tid == 0) s_nl = 0;
      __syncthreads();
      if (c0 + tid < c_end && (int)nnz[(size_t)gene * G + cand[c0 + tid]] > cap) s_list[atomicAdd(&s_nl, 1)] = c0 + tid;
      __syncthreads();
      const int nl = s_nl;
      for (int li = 0; li < nl; ++li) {
        const int cd = s_list[li], g = cand[cd];
        int n = (int)nnz[(size_t)gene * G + g];
        if (n >= 65535) { // the 16-bit run length saturated
            if (!run_n) { if (tid == 0) route[gene] = 2u; continue; }
            n = (int)run_n[(size_t)gene * n_cand + cd];
        }
        const size_t off = (size_t)((long long)gene * gene_stride) + gofs[(size_t)gene * G + g];
        KeyT *run = (KeyT *)Xs + off, *out = (KeyT *)tmp + off;
        if (tid == 0) { g_min = MAXK; g_max = (KeyT)0; }
        __syncthreads();
        KeyT lo = MAXK, hi = (KeyT)0;
        run_for_keys<KeyT, NT>(run, n, tid, [&](KeyT k, int) { lo = k < lo ? k : lo; hi = k > hi ? k : hi; });
    #pragma unroll
        for (int d = 32; d > 0; d >>= 1) { const KeyT a = __shfl_xor(lo, d), b = __shfl_xor(hi, d); lo = a < lo ? a : lo; hi = b > hi ? b : hi; }
        if (lane == 0) { atomicMin(&g_min, lo); atomicMax(&g_max, hi); }
        __syncthreads();
        int lg = 6;
        while (lg < lg_max && (4 << lg) < n) ++lg;
        const int B = 1 << lg;
        const KeyT range = (KeyT)(g_max - g_min);
        const int bits = range ? (int)(sizeof(KeyT) * 8) - (sizeof(KeyT) == 8 ? __clzll((long long)range) : __clz((int)range)) : 0;
        BigRunFn<KeyT> f;
        f.kmin = g_min;
        f.shift = bits > lg ? bits - lg : 0;
        if (tid == 0) { BigRunFn<KeyT> fo = f; fo.shift |= BIG_RUN_IN_TMP; big_fn[(size_t)gene * n_cand + cd] = fo; } // (the rank kernel reads this run from `tmp`)
        for (int b = tid; b < B; b += NT) cnt[b] = 0u;
        __syncthreads();
        run_for_keys<KeyT, NT>(run, n, tid, [&](KeyT k, int) { atomicAdd(&cnt[big_bucket(f, k)], 1u); });
        __syncthreads();
        { // exclusive scan of the B counters: a thread owns B / 1024 consecutive ones
            const int per = (B + NT - 1) / NT;
            u32 sum = 0;
            for (int e = 0; e < per; ++e) { const int b = tid * per + e; sum += b < B ? cnt[b] : 0u; }
            const u32 inc = (u32)wave_incl_scan_add((int)sum);
            if (lane == 63) g_part[wave] = inc;
            __syncthreads();
            u32 base = inc - sum;
            for (int w = 0; w < wave; ++w) base += g_part[w];
            for (int e = 0; e < per; ++e) { const int b = tid * per + e; if (b < B) { const u32 c = cnt[b]; cnt[b] = base; base += c; } }
        }
        if (run_cuts) { // the rank kernel walks a long run with all its wavefronts: where stretch w of 16 begins (the first bucket that starts at or behind w n / 16)
            __syncthreads();
            if (tid >= 1 && tid <= OCR_CUTS) {
                const u32 t = (u32)((long long)n * tid / (OCR_CUTS + 1));
                int lo_b = 0, hi_b = B;
                while (lo_b < hi_b) { const int mid = (lo_b + hi_b) >> 1; if (cnt[mid] >= t) hi_b = mid; else lo_b = mid + 1; }
                run_cuts[((size_t)gene * n_cand + cd) * OCR_CUTS + (tid - 1)] = lo_b < B ? cnt[lo_b] : (u32)n;
            }
        }
    // the scatter, in SLICES of the output through LDS: a slice = the buckets that start inside [s T, (s + 1) T) of the output, T = the slice
        // buffer less 256 keys (a bucket of more than 256 keys -- its gene leaves the rank kernel anyway -- may overhang: those keys go straight
        // out); the run is read once per slice (16-byte loads, from L2), its keys of the slice take their places in the buffer, the buffer goes
        // out in whole lines.  (Scattered from here, every 4-byte store was an L2 request of its own: 1.2 G of them for ten clusters of 100 000
        // cells half non-zero, 13 of the kernel's 16 ms.)
        KeyT *buf = (KeyT *)(srtg_smem + ((size_t)4 << lg_max));
        const int T = max(slice_keys - 256, 64);
        __syncthreads();
        for (int b_lo = 0, b_hi; b_lo < B; b_lo = b_hi) { // (uniform)
            const u32 base = cnt[b_lo]; // (buckets from b_lo on: their starts, untouched so far)
            if (base >= (u32)n) break;
            { // the first bucket behind b_lo that starts at or after base + T
                int lo_b = b_lo + 1, hi_b = B;
                const u32 t = base + (u32)T;
                while (lo_b < hi_b) { const int mid = (lo_b + hi_b) >> 1; if (cnt[mid] >= t) hi_b = mid; else lo_b = mid + 1; }
                b_hi = lo_b;
            }
            const u32 end = b_hi < B ? cnt[b_hi] : (u32)n;
            __syncthreads(); // (every thread has read the starts before the first atomic moves one)
            run_for_keys<KeyT, NT>(run, n, tid, [&](KeyT k, int) {
                const u32 b = big_bucket(f, k);
                if (b >= (u32)b_lo && b < (u32)b_hi) {
                    const u32 pos = atomicAdd(&cnt[b], 1u), loc = pos - base;
                    if (loc < (u32)slice_keys) buf[loc] = k; else out[pos] = k;
                }
            });
            __syncthreads();
            const u32 m = min(end - base, (u32)slice_keys);
            for (u32 i = tid; i < m; i += NT) out[base + i] = buf[i];
            __syncthreads();
        }
        __syncthreads(); // (the next run reuses the counters and the range)
      }
      __syncthreads(); // (the list is rebuilt)
    }
}

template <typename KeyT, int NT_ = SRT_NT>
__global__ __launch_bounds__(NT_) void k_bucket_big_runs(void *Xs, long long gene_stride, const u16 *__restrict__ nnz, const u32 *__restrict__ gofs,
                                                            const int *__restrict__ cand, int n_cand, int G, int cap /* LDS key slots */,
                                                            BigRunFn<KeyT> *__restrict__ big_fn, u32 *__restrict__ route, int global_follows /* another launch takes the longer runs */,
                                                            int n_min /* runs up to here are left alone (64 * OCR_KMAX, or another launch's) */,
                                                            u32 *__restrict__ run_cuts /* optional: OCR_CUTS bucket boundaries per run above OCR_COOP_MIN keys */) {
    extern __shared__ __align__(16) unsigned char srt_smem[];
    constexpr int NTK = NT_; // (256 threads; 1024 in a launch of its own for runs above 8192 keys: the LDS they take leaves room for two workgroups per CU)
    constexpr KeyT MAXK = KeyInfo<KeyT>::MAXK;
    KeyT *K = (KeyT *)srt_smem;
    __shared__ KeyT s_min, s_max;
    __shared__ u32 s_part[NTK / 64];
    __shared__ int s_list[NTK], s_nl;
    const int gene = blockIdx.y, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    u32 *cnt = (u32 *)(srt_smem + (size_t)cap * sizeof(KeyT)); // [buckets]
    // the workgroup's stretch of the gene's candidate groups: their run lengths are looked at NTK at a time, the runs above 256 keys
    // listed, the list walked (300 groups of 333 cells half non-zero are 1.5 M candidates none of which holds 256 keys)
    const int per = (n_cand + (int)gridDim.x - 1) / (int)gridDim.x, c_end = min(n_cand, ((int)blockIdx.x + 1) * per);
    for (int c0 = (int)blockIdx.x * per; c0 < c_end; c0 += NTK) {
      if (tid == 0) s_nl = 0;
      __syncthreads();
      if (c0 + tid < c_end && (int)nnz[(size_t)gene * G + cand[c0 + tid]] > n_min) s_list[atomicAdd(&s_nl, 1)] = c0 + tid;
      __syncthreads();
      const int nl = s_nl;
      for (int li = 0; li < nl; ++li) {
        const int cd = s_list[li], g = cand[cd];
        const int n = (int)nnz[(size_t)gene * G + g];
        KeyT *run = (KeyT *)Xs + (long long)gene * gene_stride + gofs[(size_t)gene * G + g];
        if (n > cap) { if (tid == 0 && !global_follows) route[gene] = 2u; continue; } // (uniform)
        if (tid == 0) { s_min = MAXK; s_max = (KeyT)0; }
        __syncthreads();
        KeyT lo = MAXK, hi = (KeyT)0;
        run_for_keys<KeyT, NTK>(run, n, tid, [&](KeyT k, int i) { K[i] = k; lo = k < lo ? k : lo; hi = k > hi ? k : hi; });
#pragma unroll
        for (int d = 32; d > 0; d >>= 1) { const KeyT a = __shfl_xor(lo, d), b = __shfl_xor(hi, d); lo = a < lo ? a : lo; hi = b > hi ? b : hi; }
        if (lane == 0) { atomicMin(&s_min, lo); atomicMax(&s_max, hi); }
        __syncthreads();
        // about four keys per bucket
        int lg = 6;
        while (lg < SRT_LG_MAX && (4 << lg) < n) ++lg;
        const int B = 1 << lg;
        const KeyT range = (KeyT)(s_max - s_min);
        const int bits = range ? (int)(sizeof(KeyT) * 8) - (sizeof(KeyT) == 8 ? __clzll((long long)range) : __clz((int)range)) : 0;
        BigRunFn<KeyT> f;
        f.kmin = s_min;
        f.shift = bits > lg ? bits - lg : 0;
        if (tid == 0) big_fn[(size_t)gene * n_cand + cd] = f;
        for (int b = tid; b < B; b += NTK) cnt[b] = 0u;
        __syncthreads();
        for (int i = tid; i < n; i += NTK) atomicAdd(&cnt[big_bucket(f, K[i])], 1u);
        __syncthreads();
        { // exclusive scan of the B counters (B <= 4096 = 16 per thread)
            const int per = (B + NTK - 1) / NTK;
            u32 loc[(1 << SRT_LG_MAX) / NTK], sum = 0;
#pragma unroll
            for (int e = 0; e < (1 << SRT_LG_MAX) / NTK; ++e) { const int b = tid * per + e; loc[e] = (e < per && b < B) ? cnt[b] : 0u; sum += loc[e]; }
            const u32 inc = (u32)wave_incl_scan_add((int)sum);
            if (lane == 63) s_part[wave] = inc;
            __syncthreads();
            u32 base = inc - sum;
            for (int w = 0; w < wave; ++w) base += s_part[w];
#pragma unroll
            for (int e = 0; e < (1 << SRT_LG_MAX) / NTK; ++e) { const int b = tid * per + e; if (e < per && b < B) { cnt[b] = base; base += loc[e]; } }
        }
        __syncthreads();
        if (run_cuts && n > OCR_COOP_MIN && tid >= 1 && tid <= OCR_CUTS) { // (as k_bucket_big_runs_global: the starts are still untouched)
            const u32 t = (u32)((long long)n * tid / (OCR_CUTS + 1));
            int lo_b = 0, hi_b = B;
            while (lo_b < hi_b) { const int mid = (lo_b + hi_b) >> 1; if (cnt[mid] >= t) hi_b = mid; else lo_b = mid + 1; }
            run_cuts[((size_t)gene * n_cand + cd) * OCR_CUTS + (tid - 1)] = lo_b < B ? cnt[lo_b] : (u32)n;
        }
        __syncthreads();
        // (dealt into a second LDS buffer first and written back in whole lines: no faster on dense input -- 11.05 vs 11.06 ms, 10 groups of
        //  30 000 cells -- and the LDS it takes halves the workgroups per CU: 1.43 -> 2.59 ms on CSC input with 50 groups)
        for (int i = tid; i < n; i += NTK) {
            const KeyT k = K[i];
            run[atomicAdd(&cnt[big_bucket(f, k)], 1u)] = k;
        }
        __syncthreads(); // (the next run reuses K, the counters and the range)
      }
      __syncthreads(); // (the list is rebuilt)
    }
}

// sparse input regrouped into Xs + seg ([n_genes][G + 1] offsets of the (gene, group) runs): the same runs in the packed layout's terms
static __global__ __launch_bounds__(256) void k_seg_to_packed(const u32 *__restrict__ seg, int G, int nb, int ref, u16 *__restrict__ nnz, u32 *__restrict__ gofs,
                                                             u16 *__restrict__ ref_nnz, u32 *__restrict__ route, const int *__restrict__ cand_of, u32 *__restrict__ run_n,
                                                             int n_cand) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (long long)nb * G) return;
    const int gene = (int)(i / G), g = (int)(i - (long long)gene * G);
    const u32 a = seg[(size_t)gene * (G + 1) + g], n = seg[(size_t)gene * (G + 1) + g + 1] - a;
    const int cd = (run_n && g != ref) ? cand_of[g] : -1;
    if (cd >= 0) run_n[(size_t)gene * n_cand + cd] = n;
    if (n > 65535u && (cd < 0 || g == ref)) route[gene] = 2u; // (16-bit run lengths; a ranked group's long run has its exact length in run_n)
    nnz[i] = (u16)(n > 65535u ? 65535u : n);
    gofs[i] = a;
    if (g == ref) ref_nnz[gene] = (u16)(n > 65535u ? 65535u : n);
}

// ---- value-range parts of a reference that outgrows the rank kernel's key slots ---------------------------------------------------
#define OCR_CELL_LG 12
#define OCR_PMAX 32
template <typename KeyT> struct PartCuts {
    u32 n_parts;             // 0: the gene needs more parts than OCR_PMAX / the launch has (flagged for the general route)
    u32 n_all;               // the reference's non-zero keys
    KeyT kmin, kmax;         // their range
    KeyT lo[OCR_PMAX];       // lo[j]: the smallest key of part j (j >= 1; part 0 starts at 0, the last part ends at the largest key)
    u32 n_low[OCR_PMAX + 1]; // n_low[j]: reference keys below part j; n_low[n_parts] = n_all
};

// One workgroup per gene that the plain rank kernel handed over (needs_parts): 4096 cells over the reference's key range are counted
// and cut where the running count passes j / P of the keys (P = ceil(keys / (7/8 of the slots)): the cuts fall on cell boundaries, an
// eighth of the slots is slack; a part that still outgrows the slots -- a cell that holds a crowd -- is caught by the rank kernel).
template <typename KeyT>
__global__ __launch_bounds__(1024) void k_ref_cuts(OvoCompactParams P, PartCuts<KeyT> *__restrict__ cuts) {
    constexpr int NT = 1024, NW = NT / 64, NC = 1 << OCR_CELL_LG, CPT = NC / NT;
    constexpr KeyT MAXK = KeyInfo<KeyT>::MAXK;
    __shared__ u32 cells[NC];
    __shared__ KeyT s_kr[2];
    __shared__ u32 s_n, s_scan[NW];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, gene = blockIdx.x, G = P.G;
    if (P.needs_parts[gene] == 0u) return; // (uniform)
    const KeyT *Xg = (const KeyT *)P.Xs + (long long)gene * P.gene_stride;
    const KeyT *src = P.ref_by_gofs ? Xg + P.gofs[(size_t)gene * G + P.ref] : Xg + P.ref_out;
    const u16 *seg_nnz = P.seg_nnz + (size_t)gene * P.nseg;
    auto for_ref = [&](auto f) {
        for (int sg = wave; sg < P.nseg; sg += NW) {
            const int c = (int)seg_nnz[sg];
            const KeyT *sp = src + (size_t)sg * GCMP_SEG_ROWS;
            for (int i = lane; i < c; i += 64) f(sp[i]);
        }
    };
    for (int i = tid; i < NC; i += NT) cells[i] = 0u;
    if (tid == 0) { s_kr[0] = MAXK; s_kr[1] = (KeyT)0; s_n = 0u; }
    __syncthreads();
    for (int sg = tid; sg < P.nseg; sg += NT) atomicAdd(&s_n, (u32)seg_nnz[sg]);
    {
        KeyT tmin = MAXK, tmax = (KeyT)0;
        for_ref([&](KeyT k) { tmin = k < tmin ? k : tmin; tmax = k > tmax ? k : tmax; });
#pragma unroll
        for (int d = 32; d > 0; d >>= 1) {
            const KeyT o1 = __shfl_xor(tmin, d), o2 = __shfl_xor(tmax, d);
            tmin = o1 < tmin ? o1 : tmin;
            tmax = o2 > tmax ? o2 : tmax;
        }
        if (lane == 0) { atomicMin(&s_kr[0], tmin); atomicMax(&s_kr[1], tmax); }
    }
    __syncthreads();
    const u32 n_all = s_n, cap_s = (u32)P.ref_cap - (u32)P.ref_cap / 8u;
    const int n_parts = (int)((n_all + cap_s - 1u) / cap_s);
    PartCuts<KeyT> &C = cuts[gene];
    if (n_parts > P.n_parts || n_parts > OCR_PMAX) { // (uniform) the general route's gene (dense layout: a reason in the word's high bits keeps k_ovo_rank away)
        if (tid == 0) { C.n_parts = 0u; P.route[gene] = P.ref_by_gofs ? 1u : (1u | (1u << 8)); }
        return;
    }
    const KeyT kmin = s_kr[0], range = (KeyT)(s_kr[1] - s_kr[0]);
    const int bits = range ? (int)(sizeof(KeyT) * 8) - (sizeof(KeyT) == 8 ? __clzll((long long)range) : __clz((int)range)) : 0;
    const int cs0 = bits > OCR_CELL_LG ? bits - OCR_CELL_LG : 0;
    for_ref([&](KeyT k) { atomicAdd(&cells[(u32)((KeyT)(k - kmin) >> cs0)], 1u); });
    __syncthreads();
    u32 c4[CPT], sum = 0;
#pragma unroll
    for (int e = 0; e < CPT; ++e) { c4[e] = cells[tid * CPT + e]; sum += c4[e]; }
    const u32 inc = (u32)wave_incl_scan_add((int)sum);
    if (lane == 63) s_scan[wave] = inc;
    __syncthreads();
    u32 ex = inc - sum;
    for (int w = 0; w < wave; ++w) ex += s_scan[w];
    if (tid == 0) { C.n_parts = (u32)n_parts; C.n_all = n_all; C.kmin = kmin; C.kmax = s_kr[1]; C.lo[0] = (KeyT)0; C.n_low[0] = 0u; C.n_low[n_parts] = n_all; }
    // the cut in front of part j: the cell in which the running count passes j / P of the keys (the cell itself goes to part j)
#pragma unroll
    for (int e = 0; e < CPT; ++e) {
        if (c4[e]) {
            for (int j = 1; j < n_parts; ++j) {
                const u32 t = (u32)((u64)n_all * (u64)j / (u64)n_parts);
                if (ex <= t && t < ex + c4[e]) { C.lo[j] = (KeyT)(kmin + ((KeyT)(u32)(tid * CPT + e) << cs0)); C.n_low[j] = ex; }
            }
        }
        ex += c4[e];
    }
}

// The runs of at most 256 keys of a gene that is taken in 2 .. 4 parts, DEALT by part in place: one wavefront loads a (gene, group) run,
// finds each key's part (at most three compares against the cuts) and writes the keys back part after part; pofs[gene][group][j] = where
// part j begins (j = 1 .. 3; part 0 begins at 0, the last part ends with the run).  The parts kernel then reads, for its part, that
// stretch only: a look-up per key instead of one per key and part (masked), and 1 / P of the bytes.  Longer runs are left alone: they
// lie in value-bucket order (k_bucket_big_runs), the parts kernel skips their pieces outside its range.
#define DEAL_NT 256
template <typename KeyT>
__global__ __launch_bounds__(DEAL_NT) void k_deal_runs(OvoCompactParams P, const PartCuts<KeyT> *__restrict__ cuts, u16 *__restrict__ pofs) {
    constexpr KeyT ZEROK = KeyInfo<KeyT>::ZEROK;
    constexpr int KMAX = OCR_KMAX, NW = DEAL_NT / 64;
    const int gene = blockIdx.y, lane = threadIdx.x & 63, wave = threadIdx.x >> 6, G = P.G;
    if (P.needs_parts[gene] == 0u) return; // (uniform)
    const PartCuts<KeyT> &C = cuts[gene];
    const int np = (int)C.n_parts;
    if (np < 2 || np > 4) return;          // (uniform)
    const KeyT c1 = C.lo[1], c2 = np > 2 ? C.lo[2] : KeyInfo<KeyT>::MAXK, c3 = np > 3 ? C.lo[3] : KeyInfo<KeyT>::MAXK;
    const bool two = np > 2, three = np > 3;
    KeyT *Xg = (KeyT *)P.Xs + (long long)gene * P.gene_stride;
    const u64 lt_mask = (lane == 0) ? 0ull : (~0ull >> (64 - lane));
    // a run is a kilobyte: the next group's keys are requested before this group's are dealt (the pass would otherwise run on latency)
    const int step = (int)gridDim.x * NW;
    auto meta = [&](int g, int &n, long long &off) {
        n = 0; off = 0;
        if (g < G && g != P.ref) { const size_t o = (size_t)gene * G + g; n = (int)P.nnz[o]; off = (long long)P.gofs[o]; }
    };
    auto load = [&](int n, long long off, KeyT (&k)[KMAX]) {
#pragma unroll
        for (int r = 0; r < KMAX; ++r) k[r] = (n <= 64 * KMAX && r * 64 + lane < n) ? Xg[off + r * 64 + lane] : ZEROK;
    };
    int g = (int)blockIdx.x * NW + wave, n, n2;
    long long off, off2;
    KeyT cur[KMAX], nxt[KMAX];
    meta(g, n, off);
    load(n, off, cur);
    for (; g < G; g += step) {
        meta(g + step, n2, off2);
        load(n2, off2, nxt);
        if (g != P.ref) {
            u16 *po = pofs + ((size_t)gene * G + g) * 4;
            if (n == 0 || n > 64 * KMAX) { if (lane < 4) po[lane] = 0; } // (not dealt: the parts kernel reads pofs only of runs of 1 .. 256 keys)
            else {
                KeyT *seg = Xg + off;
                int pt[KMAX];
#pragma unroll
                for (int r = 0; r < KMAX; ++r)
                    pt[r] = (r * 64 + lane < n) ? (int)(cur[r] >= c1) + (int)(two && cur[r] >= c2) + (int)(three && cur[r] >= c3) : 7;
                int base = 0;
                for (int q = 0; q < np; ++q) { // (uniform) part q's keys behind the keys of the parts before it
                    if (q > 0 && lane == 0) po[q] = (u16)base;
#pragma unroll
                    for (int r = 0; r < KMAX; ++r) {
                        if (r * 64 >= n) break; // (uniform)
                        const u64 m = __ballot(pt[r] == q);
                        if (pt[r] == q) seg[base + (int)__popcll(m & lt_mask)] = cur[r];
                        base += (int)__popcll(m);
                    }
                }
                if (lane == 0) { po[0] = 1; for (int q = np; q < 4; ++q) po[q] = (u16)n; } // po[0] = 1: dealt
            }
        }
#pragma unroll
        for (int r = 0; r < KMAX; ++r) cur[r] = nxt[r];
        n = n2; off = off2;
    }
}

// PARTS: a reference whose non-zero keys outgrow the LDS slots is taken in VALUE-RANGE PARTS, one workgroup per (gene, part): 4096 cells
// over the reference's key range are counted, cut where the running count passes j / P of the keys, and the workgroup of part j keeps the
// keys of its cells only -- table, look-ups and tie search as before, a group's keys outside the part masked out (every part reads every
// group's packed keys: P reads of the packed rows, from L2 / MALL when the parts of a gene run side by side).  Equal keys share a
// cell, hence a part: S2 = sum over parts of [2 (keys of the reference below the part) + 2 #A_part<b + #A_part==b] over the part's b, and
// the tie terms add up likewise.  Every part ADDS its share into out_2u / out_tie (zeroed by the host); part 0 adds the terms of
// the zeros.  (Replaces, for references of any size, what rank_sum_and_ties_from_sorted does by merging: utils/ranking.py:52-158.)
// NT_ = 256: small references and few groups (a wide matrix: 120 000 genes x 20 000 cells): a gene is a few microseconds of work behind a
// dozen barriers -- several small workgroups per CU overlap them (one of 1024 threads per CU: 11.4 ms at that shape).
template <typename KeyT, bool EQ, bool PARTS = false, int NT_ = OCR_NT>
__global__ __launch_bounds__(NT_) void k_ovo_rank_compact(OvoCompactParams P) {
    extern __shared__ __align__(16) unsigned char smem[];
    constexpr int NT = NT_, NW = NT / 64, KMAX = OCR_KMAX;
    constexpr KeyT ZEROK = KeyInfo<KeyT>::ZEROK, MAXK = KeyInfo<KeyT>::MAXK;
    const int NWD = (1 << P.nbk_lg) >> 4; // table words
    KeyT *A = (KeyT *)smem;
    size_t off = (((size_t)P.ref_cap + 4) * sizeof(KeyT) + 15) & ~(size_t)15;
    u32 *tab = (u32 *)(smem + off);
    off += ((size_t)NWD + 2) * 8;
    u32 *bloom_all = (u32 *)(smem + off);
    off += (size_t)NW * OCR_BLOOM_WORDS * 4;
    u32 *ctab = (u32 *)(smem + off); // [256] coarse cells (EQ)
    u32 *ccnt = ctab + 256;          // [256] their key counts
    off += 2048;
    u64 *s_red = (u64 *)(smem + off); // [NW]
    KeyT *s_kr = (KeyT *)(s_red + NW); // [2] min, max
    u32 *s_cnt = (u32 *)(s_kr + 2);    // [0] negatives  [1] non-zero keys  [2] fullest overfull word  [3] largest group  [4] keys in overfull words
    u32 *s_scan = s_cnt + 8;           // [NW]
    u32 *s_b = s_scan + NW;            // PARTS: [0] first cell of the part  [1] reference keys below it  [2] one past its last cell  [3] keys below that

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    // (part-major: the workgroups of part 0 first.  Gene-major would put the parts a gene does NOT need -- workgroups that return at once --
    //  on every n_parts-th workgroup slot, i.e. on fixed XCDs: with two parts, half of the chip idle whenever the genes need one.)
    const int gene = PARTS ? (int)blockIdx.x % P.n_genes : (int)blockIdx.x;
    const int part = PARTS ? (int)blockIdx.x / P.n_genes : 0;
    if (P.gene_flags && P.gene_flags[gene] == 0u) return; // (uniform)
    if (P.big_sorted && (P.route[gene] & 255u) == 2u) return; // (uniform) k_bucket_big_runs met a run beyond its slots: the general route's gene
    const int G = P.G, ref = P.ref;
    const int n_ref = P.counts[ref];
    u16 *nnz = P.nnz + (size_t)gene * G;
    KeyT *Xg = (KeyT *)P.Xs + (long long)gene * P.gene_stride;
    KeyT *src = P.ref_by_gofs ? Xg + P.gofs[(size_t)gene * G + ref] : Xg + P.ref_out;
    const u16 *seg_nnz = P.seg_nnz + (size_t)gene * P.nseg;
    // the reference's non-zero keys lie in nseg packed segments: wavefront w walks segments w, w + NW, ...
    auto for_ref = [&](auto f) {
        for (int sg = wave; sg < P.nseg; sg += NW) {
            const int c = (int)seg_nnz[sg];
            const KeyT *sp = src + (size_t)sg * GCMP_SEG_ROWS;
            for (int i = lane; i < c; i += 64) f(sp[i]);
        }
    };

    // ---- the reference's non-zero keys -> value buckets ----
    for (int i = tid; i < NW * OCR_BLOOM_WORDS; i += NT) bloom_all[i] = 0u;
    {
        uint4 *t4 = (uint4 *)tab; // (the table starts on a 16-byte boundary and holds an even number of 64-bit words)
        for (int i = tid; i < (NWD + 2) / 2; i += NT) t4[i] = make_uint4(0u, 0u, 0u, 0u);
    }
    if (EQ && tid < 256) ccnt[tid] = 0u;
    if (tid == 0) { s_kr[0] = MAXK; s_kr[1] = (KeyT)0; s_cnt[0] = 0u; s_cnt[1] = 0u; s_cnt[2] = 0u; s_cnt[3] = 0u; s_cnt[4] = 0u; s_red[0] = 0ull; }
    __syncthreads();
    for (int sg = tid; sg < P.nseg; sg += NT) atomicAdd(&s_cnt[1], (u32)seg_nnz[sg]); // (a reference of any size: nseg = n_ref / 512)
    int n_parts_gene = 1;
    if constexpr (!PARTS) {
        if (P.needs_parts) { // (uniform) a PARTS launch follows: the genes beyond the slots are its own
            __syncthreads();
            if (s_cnt[1] > (u32)P.ref_cap) { if (tid == 0) P.needs_parts[gene] = 1u; return; }
        }
    }
    if constexpr (PARTS) { // how many parts this gene's reference needs (an eighth of the slots is slack: the cuts fall on cell boundaries)
        if (P.needs_parts[gene] == 0u) return; // (uniform) the plain kernel's gene
        n_parts_gene = (int)((const PartCuts<KeyT> *)P.cuts)[gene].n_parts; // (0: more parts than the launch has -- k_ref_cuts flagged the gene)
        if (part >= n_parts_gene) return;       // (uniform)
    }
    {
        KeyT tmin = MAXK, tmax = (KeyT)0;
        u32 ng = 0;
        for_ref([&](KeyT k) {
            tmin = k < tmin ? k : tmin;
            tmax = k > tmax ? k : tmax;
            ng += k < ZEROK ? 1u : 0u;
        });
        ng = (u32)wave_sum((int)ng);
#pragma unroll
        for (int d = 32; d > 0; d >>= 1) {
            const KeyT o1 = __shfl_xor(tmin, d), o2 = __shfl_xor(tmax, d);
            tmin = o1 < tmin ? o1 : tmin;
            tmax = o2 > tmax ? o2 : tmax;
        }
        if (lane == 0) {
            atomicMin(&s_kr[0], tmin);
            atomicMax(&s_kr[1], tmax);
            if (ng) atomicAdd(&s_cnt[0], ng);
        }
        u32 gmax = 0;
        for (int gq = tid; gq < G; gq += NT) gmax = max(gmax, gq == ref ? 0u : (u32)nnz[gq]);
        gmax = (u32)wave_incl_scan_max((int)gmax);
        if (lane == 63 && gmax) atomicMax(&s_cnt[3], gmax);
        if (tid == NT - 1 && P.seg_sum && part == 0) { // the reference's value sum: its segments' sums in order
            double t = 0.0;
            for (int sg = 0; sg < P.nseg; ++sg) t += P.seg_sum[(size_t)gene * P.nseg + sg];
            P.out_sum[(size_t)gene * G + ref] = t;
        }
    }
    __syncthreads();
    const u32 nA_all = s_cnt[1];
    const u32 aZ = (u32)n_ref - nA_all;
    u32 nA = nA_all, n_low = 0;               // the part's reference keys; the reference's non-zero keys below the part
    KeyT p_lo = (KeyT)0, p_hi = MAXK;         // the part's key range, both ends included (part 0 starts at 0, the last part ends at MAXK)
    KeyT kmin_p = s_kr[0], kmax_p = s_kr[1];  // the range its value buckets cover
    if constexpr (PARTS) { // this part's key range, the reference's keys below it and inside it: k_ref_cuts has them
        const PartCuts<KeyT> &C = ((const PartCuts<KeyT> *)P.cuts)[gene];
        n_low = C.n_low[part];
        nA = C.n_low[part + 1] - n_low;
        if (part > 0) { p_lo = C.lo[part]; kmin_p = p_lo; }
        if (part + 1 < n_parts_gene) { p_hi = (KeyT)(C.lo[part + 1] - (KeyT)1); kmax_p = p_hi; }
    }
    auto for_part = [&](auto f) { // the reference's keys of this part
        if constexpr (PARTS) for_ref([&](KeyT k) { if (k >= p_lo && k <= p_hi) f(k); });
        else for_ref(f);
    };
    for (u32 i = tid; i < min(nA, (u32)P.ref_cap) + 4u; i += NT) A[i] = MAXK; // empty slots (the scatter claims them by compare-and-swap) and the pad
    OcrRef<KeyT, EQ> R;
    R.A = A; R.tab = tab; R.last = (1u << P.nbk_lg) - 1u;
    R.kmin = nA ? kmin_p : (KeyT)0;
    R.ctab = ctab;
    {
        const KeyT range = nA ? (KeyT)(kmax_p - kmin_p) : (KeyT)0;
        const int bits = range ? (int)(sizeof(KeyT) * 8) - (sizeof(KeyT) == 8 ? __clzll((long long)range) : __clz((int)range)) : 0;
        R.shift = bits > P.nbk_lg ? bits - P.nbk_lg : 0;
        R.cshift = bits > 8 ? bits - 8 : 0;
        R.cmask = (KeyT)(((KeyT)1 << R.cshift) - (KeyT)1);
    }
    R.nA = nA;
    if constexpr (EQ) { // key counts of the 256 coarse cells -> each cell's share of the fine buckets
        for_part([&](KeyT k) { atomicAdd(&ccnt[(u32)((KeyT)(k - R.kmin) >> R.cshift)], 1u); });
        __syncthreads();
        if (wave == 0) { // four cells per lane.  A cell of cnt keys asks for cnt (NB - 256) / (2 n) buckets, rounded up to a power of
            // two (less than twice that) and at least one: the shares add up to at most NB
            const u64 nb_free = (u64)(1u << P.nbk_lg) - 256ull, den = 2ull * (u64)max(nA, 1u);
            u32 e4[4], sz = 0;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const u64 want = (u64)ccnt[lane * 4 + j] * nb_free / den;
                u32 e = want <= 1ull ? 0u : (u32)(64 - __clzll((long long)(want - 1ull)));
                e = min(e, (u32)R.cshift);
                e4[j] = e;
                sz += 1u << e;
            }
            u32 base = (u32)wave_incl_scan_add((int)sz) - sz;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                ctab[lane * 4 + j] = base | (((u32)R.cshift - e4[j]) << 24);
                base += 1u << e4[j];
            }
        }
        __syncthreads();
    }
    const u32 nneg = s_cnt[0];
    // counters: +1 on the bucket's two bits; a bucket already at 3 takes the increment back and marks its word overfull (the
    // carry it sent into the next field in between is removed by the subtraction; whatever the fields of such a word end up
    // holding is never used).  The word's key count is kept in the high half meanwhile.
    for_part([&](KeyT k) {
        const u32 b = ocr_bucket(R, k), sh = (b & 15u) << 1;
        u32 *pw = tab + 2 * (b >> 4);
        atomicAdd(&pw[1], 1u);
        const u32 old = atomicAdd(&pw[0], 1u << sh);
        if (((old >> sh) & 3u) == 3u) {
            atomicSub(&pw[0], 1u << sh);
            atomicOr(&pw[1], 0x80000000u);
        }
    });
    __syncthreads();
    { // exclusive scan of the words' key counts.  Each wavefront owns a contiguous slice and walks it 64 words at a time.
        const int per_wave = NWD / NW, iters = per_wave / 64; // NWD >= NW * 64
        u32 *slice = tab + 2 * (wave * per_wave);
        u32 tot = 0, fmax = 0, ftot = 0;
        u64 fsq = 0;
        for (int it = 0; it < iters; ++it) {
            const u32 h = slice[2 * (it * 64 + lane) + 1], c = h & 0xFFFFu;
            tot += c;
            if (h >> 31) { fmax = max(fmax, c); ftot += c; fsq += (u64)c * c; }
        }
        tot = (u32)wave_sum((int)tot);
        ftot = (u32)wave_sum((int)ftot);
        fmax = (u32)wave_incl_scan_max((int)fmax);
        if constexpr (PARTS) fsq = wave_sum<u64>(fsq);
        if (lane == 0) { s_scan[wave] = tot; if (ftot) atomicAdd(&s_cnt[4], ftot); if (PARTS && fsq) atomicAdd((unsigned long long *)&s_red[0], (unsigned long long)fsq); }
        if (lane == 63 && fmax) atomicMax(&s_cnt[2], fmax);
        __syncthreads();
        // Crowded tables.  What such a gene falls back to is k_ovo_rank (30 us a gene) without parts -- the thresholds are tuned for that --
        // but the general route with them (a per-gene radix sort in HBM: 50 ms for a column of two million cells): there the gene stays
        // unless the key-by-key walks would cost more than that -- a look-up that meets an overfull word of c keys walks c keys, and
        // meets it with probability ~ c / nA: the expected walk is sum c^2 / nA keys per look-up; 32 is where the walks double the kernel.
        const bool crowded = PARTS ? s_red[0] > 32ull * (u64)nA : (s_cnt[2] > OCR_MAX_WORD || s_cnt[4] * 2u > nA);
        if (nA > (u32)P.ref_cap || crowded || (s_cnt[3] > 64u * OCR_KMAX && !P.big_sorted)) { // uniform: this gene goes to k_ovo_rank
            if constexpr (PARTS) { // (the other parts of the gene are reading the segments: nothing is moved; the gene takes the general route)
                // (dense layout: a reason in the high bits keeps k_ovo_rank away; regrouped sparse input: the reference is one run, that kernel can take it)
                if (tid == 0) P.route[gene] = P.ref_by_gofs ? 1u : (1u | ((nA > (u32)P.ref_cap ? 2u : crowded ? 3u : 5u) << 8));
#ifdef OCR_DEBUG_PRINT
                if (tid == 0) printf("left: gene %d part %d/%d nA %u of %u, fullest word %u, in overfull words %u, kmin %llx kmax %llx, part range %llx .. %llx, n_low %u\n", gene, part,
                                     n_parts_gene, nA, nA_all, s_cnt[2], s_cnt[4], (unsigned long long)s_kr[0], (unsigned long long)s_kr[1], (unsigned long long)kmin_p, (unsigned long long)kmax_p, n_low);
#endif
                return;
            }
            u32 dst = P.nseg ? (u32)seg_nnz[0] : 0u;
            for (int sg = 1; sg < P.nseg; ++sg) { // move the reference's segments together (a segment holds at most GCMP_SEG_ROWS keys: NT keys a step)
                const u32 c = (u32)seg_nnz[sg];
                for (u32 o = 0; o < c; o += (u32)NT) { // (the destination ends below the source's start: a step's reads come before its writes, steps in order)
                    KeyT k = (KeyT)0;
                    if (o + (u32)tid < c) k = src[(size_t)sg * GCMP_SEG_ROWS + o + tid];
                    __syncthreads();
                    if (o + (u32)tid < c) src[dst + o + tid] = k;
                    __syncthreads();
                }
                dst += c;
            }
            if (tid == 0) { nnz[ref] = (u16)nA; if (!P.ref_by_gofs) P.gofs[(size_t)gene * G + ref] = (u32)P.ref_out; P.route[gene] = 1u; }
            return;
        }
        u32 base = 0;
        for (int w = 0; w < wave; ++w) base += s_scan[w];
        for (int it = 0; it < iters; ++it) {
            const u32 h = slice[2 * (it * 64 + lane) + 1], c = h & 0xFFFFu;
            const u32 inc = (u32)wave_incl_scan_add((int)c);
            slice[2 * (it * 64 + lane) + 1] = (base + inc - c) | (h & 0x80000000u);
            base += (u32)__builtin_amdgcn_readlane((int)inc, 63);
        }
        if (tid == 0) { tab[2 * NWD + 1] = nA; tab[2 * NWD + 3] = nA; } // sentinel words: where the last word's keys end
    }
    __syncthreads();
    // scatter: a key's bucket owns the slots [lo, hi) (an overfull word: the word's slots); the first empty one is claimed
    for_part([&](KeyT k) {
        const u32 b = ocr_bucket(R, k), W = b >> 4, sh = (b & 15u) << 1;
        const u32 w = tab[2 * W], h = tab[2 * W + 1];
        u32 lo, hi;
        if (h >> 31) { lo = h & 0xFFFFu; hi = tab[2 * W + 3] & 0xFFFFu; }
        else { lo = h + ocr_below(w, sh); hi = lo + ((w >> sh) & 3u); }
        if (k != MAXK) // (a key equal to the empty marker is in place already)
            for (u32 t = lo; t < hi; ++t)
                if (atomicCAS(&A[t], MAXK, k) == MAXK) break;
    });
    __syncthreads();
    u64 T_A = 0; // ties among the reference's non-zero keys: sum over keys of (run length^2 - 1)
    {
        u64 ta = 0;
        for (u32 i = tid; i < nA; i += NT) {
            u32 l, a;
            ocr_find_exact(R, A[i], l, a);
            ta += (u64)a * a - 1ull;
        }
        ta = wave_sum(ta);
        if (lane == 0) s_red[wave] = ta;
        __syncthreads();
        for (int w = 0; w < NW; ++w) T_A += s_red[w];
    }

    // ---- every other group: one wavefront each, 64 groups per output block ----
    u32 *bloom = bloom_all + wave * OCR_BLOOM_WORDS;
    const u64 lt_mask = (lane == 0) ? 0ull : (~0ull >> (64 - lane));
    // Few groups (a cluster-level comparison, a small screen: fewer than 32 groups per wavefront): the blocks of 64 groups below would leave all
    // but G / 64 wavefronts idle, each walking its 64 groups one after the other (20 000 cells x 120 000 genes x 100 groups: 31 ms, 66 us a gene,
    // of which the two working wavefronts' 64 look-ups in a row are nearly all).  Here the groups are dealt round robin, one wavefront-sum each.
    const bool few_groups = G < NW * 32;
    if (few_groups) {
        for (int g = wave; g < G; g += NW) {
            const size_t o = (size_t)gene * G + g;
            if (g == ref) { if (part == 0 && lane == 0) { P.out_2u[o] = -2; P.out_tie[o] = 0; } continue; }
            const int n_run = (int)nnz[g];   // the group's non-zero keys (all parts)
            if (n_run > 64 * KMAX) continue; // (big_sorted: walked piece by piece below)
            const KeyT *seg = Xg + P.gofs[o];
            int nB = n_run;
            if constexpr (PARTS) { // a dealt run (k_deal_runs): this part's stretch of it
                if (P.pofs && n_run) {
                    const u16 *po = P.pofs + o * 4;
                    if (po[0] == 1) {
                        const int beg = part ? (int)po[part] : 0, end = part + 1 < n_parts_gene ? (int)po[part + 1] : n_run;
                        seg += beg;
                        nB = end - beg;
                    }
                }
            }
            KeyT cur[KMAX];
            u32 nv = 0;
#pragma unroll
            for (int r = 0; r < KMAX; ++r) {
                KeyT k = (r * 64 + lane < nB) ? seg[r * 64 + lane] : ZEROK;
                if constexpr (PARTS) k = (k >= p_lo && k <= p_hi) ? k : ZEROK;
                cur[r] = k;
                if constexpr (PARTS) nv += (u32)__popcll(__ballot(k != ZEROK));
            }
            u32 less = 0, eqs = 0, negs = 0;
            u64 TT = 0;
            if (PARTS ? nv != 0u : nB != 0) {
                if (nB <= 64) ocr_group<KeyT, 1, EQ, PARTS>(cur, nB, R, bloom, lane, lt_mask, less, eqs, TT, negs);
                else if (nB <= 128) ocr_group<KeyT, 2, EQ, PARTS>(cur, nB, R, bloom, lane, lt_mask, less, eqs, TT, negs);
                else if (nB <= 192) ocr_group<KeyT, 3, EQ, PARTS>(cur, nB, R, bloom, lane, lt_mask, less, eqs, TT, negs);
                else ocr_group<KeyT, 4, EQ, PARTS>(cur, nB, R, bloom, lane, lt_mask, less, eqs, TT, negs);
            }
            const u64 s2_sum = wave_sum<u64>((u64)(2u * less + eqs));
            const u64 tt_sum = __ballot(TT != 0ull) ? wave_sum<u64>(TT) : 0ull;
            if (lane == 0) {
                const long long n_g = P.counts[g];
                const u64 zc = (u64)(n_g - (long long)n_run), t0 = (u64)aZ + zc;
                if constexpr (PARTS) {
                    u64 S2 = s2_sum + 2ull * n_low * (u64)nv + 2ull * aZ * (u64)(nv - negs);
                    u64 tie = T_A + 3ull * tt_sum;
                    long long two_u = 0;
                    if (part == 0) { S2 += zc * (2ull * nneg + aZ); tie += t0 * t0 * t0 - t0; two_u = 2ll * (long long)n_ref * n_g; }
                    atomicAdd((unsigned long long *)&P.out_2u[o], (unsigned long long)(two_u - (long long)S2));
                    atomicAdd((unsigned long long *)&P.out_tie[o], (unsigned long long)tie);
                } else {
                    const u64 S2 = s2_sum + 2ull * aZ * (u64)((u32)n_run - negs) + zc * (2ull * nneg + aZ);
                    P.out_2u[o] = 2ll * (long long)n_ref * n_g - (long long)S2;
                    P.out_tie[o] = T_A + 3ull * tt_sum + (t0 * t0 * t0 - t0);
                }
            }
        }
    }
    for (int g0 = wave * 64; g0 < G && !few_groups; g0 += NW * 64) {
        const int gl = g0 + lane;
        const bool has = gl < G && gl != ref;
        const u32 my_n = has ? (u32)nnz[gl] : 0u;
        int my_pos = has ? (int)P.gofs[(size_t)gene * G + gl] : 0;
        u32 my_len = my_n; // keys to look up: the whole run, or -- a dealt run (k_deal_runs) -- this part's stretch of it
        if constexpr (PARTS) {
            if (P.pofs && my_n && my_n <= 64u * KMAX) {
                const u16 *po = P.pofs + ((size_t)gene * G + gl) * 4;
                if (po[0] == 1) {
                    const u32 beg = part ? (u32)po[part] : 0u, end = part + 1 < n_parts_gene ? (u32)po[part + 1] : my_n;
                    my_pos += (int)beg;
                    my_len = end - beg;
                }
            }
        }
        TrReduce<u32> rS2;
        u64 tt_out = 0;   // lane j: sum t (t + 1) over group g0 + j's keys (non-zero only where keys tie)
        u32 neg_out = 0;  // lane j: group g0 + j's keys below zero
        u32 nv_out = 0;   // PARTS, lane j: group g0 + j's keys inside the part
        // OCR_PAIR groups' keys are requested ahead and looked up back to back
        constexpr int PF = OCR_PAIR;
        KeyT nxt[PF][KMAX];
        int nB_n[PF];
        auto fetch = [&](int j, int h) {
            int nb = j < 64 ? (int)__builtin_amdgcn_readlane((int)my_len, j & 63) : 0;
            if (nb > 64 * KMAX) nb = 0; // (big_sorted: such a group is walked piece by piece below)
            nB_n[h] = nb;
            const KeyT *seg = Xg + __builtin_amdgcn_readlane(my_pos, j & 63);
#pragma unroll
            for (int r = 0; r < KMAX; ++r)
                if (r * 64 < nb) {
                    KeyT k = (r * 64 + lane < nb) ? seg[r * 64 + lane] : ZEROK;
                    if constexpr (PARTS) k = (k >= p_lo && k <= p_hi) ? k : ZEROK; // (keys of other parts: masked out)
                    nxt[h][r] = k;
                }
        };
#pragma unroll
        for (int h = 0; h < PF; ++h) fetch(h, h);
        for (int j0 = 0; j0 < 64; j0 += PF) { // always 64 pushes so that the transpose-reduce completes
            KeyT cur2[PF][KMAX];
            int nB2[PF];
#pragma unroll
            for (int h = 0; h < PF; ++h) {
#pragma unroll
                for (int r = 0; r < KMAX; ++r) cur2[h][r] = nxt[h][r];
                nB2[h] = nB_n[h];
            }
#pragma unroll
            for (int h = 0; h < PF; ++h) fetch(j0 + PF + h, h);
#pragma unroll
            for (int h = 0; h < PF; ++h) {
            const int j = j0 + h;
            const KeyT (&cur)[KMAX] = cur2[h];
            const int nB = nB2[h];
            u32 S2 = 0;
            u32 nv = 0;
            if constexpr (PARTS) {
                if (nB) {
#pragma unroll
                    for (int r = 0; r < KMAX; ++r)
                        if (r * 64 < nB) nv += (u32)__popcll(__ballot(cur[r] != ZEROK));
                    if (lane == j) nv_out = nv;
                }
            }
            if (PARTS ? nv != 0u : nB != 0) { // uniform
                u32 less = 0, eqs = 0, negs = 0;
                u64 TT = 0;
                if (nB <= 64) ocr_group<KeyT, 1, EQ, PARTS>(cur, nB, R, bloom, lane, lt_mask, less, eqs, TT, negs);
                else if (nB <= 128) ocr_group<KeyT, 2, EQ, PARTS>(cur, nB, R, bloom, lane, lt_mask, less, eqs, TT, negs);
                else if (nB <= 192) ocr_group<KeyT, 3, EQ, PARTS>(cur, nB, R, bloom, lane, lt_mask, less, eqs, TT, negs);
                else ocr_group<KeyT, 4, EQ, PARTS>(cur, nB, R, bloom, lane, lt_mask, less, eqs, TT, negs);
                S2 = 2u * less + eqs;
                const u64 tm = __ballot(TT != 0ull);
                if (tm) { // ties are few: mostly one lane holds the whole term
                    u64 tot;
                    if ((tm & (tm - 1ull)) == 0ull) {
                        const int sl = __ffsll((long long)tm) - 1;
                        tot = ((u64)(u32)__builtin_amdgcn_readlane((int)(u32)(TT >> 32), sl) << 32) | (u64)(u32)__builtin_amdgcn_readlane((int)(u32)TT, sl);
                    } else {
                        tot = TT;
                        tot += xor_lanes<1>(tot, lane); tot += xor_lanes<2>(tot, lane); tot += xor_lanes<4>(tot, lane);
                        tot += xor_lanes<8>(tot, lane); tot += xor_lanes<16>(tot, lane); tot += xor_lanes<32>(tot, lane);
                    }
                    if (lane == j) tt_out = tot;
                }
                if (negs && lane == j) neg_out = negs;
            }
            rS2.push(S2, j, lane);
            }
        }
        if (gl < G && my_n <= 64u * KMAX) { // lane j now holds the totals of group g0 + j
            const size_t o = (size_t)gene * G + gl;
            if (gl == ref) {
                if (part == 0) { P.out_2u[o] = -2; P.out_tie[o] = 0; }
            } else if constexpr (PARTS) { // this part's share, added: its keys against the reference's keys below the part, inside it, and the zeros
                const long long n_g = P.counts[gl];
                u64 S2 = (u64)rS2.result + 2ull * n_low * (u64)nv_out + 2ull * aZ * (u64)(nv_out - neg_out);
                u64 tie = T_A + 3ull * tt_out;
                long long two_u = 0;
                if (part == 0) { // the group's zeros, the constant term
                    const u64 zc = (u64)(n_g - (long long)my_n), t0 = (u64)aZ + zc;
                    S2 += zc * (2ull * nneg + aZ);
                    tie += t0 * t0 * t0 - t0;
                    two_u = 2ll * (long long)n_ref * n_g;
                }
                atomicAdd((unsigned long long *)&P.out_2u[o], (unsigned long long)(two_u - (long long)S2));
                atomicAdd((unsigned long long *)&P.out_tie[o], (unsigned long long)tie);
            } else {
                const long long n_g = P.counts[gl];
                const u64 zc = (u64)(n_g - (long long)my_n);       // the group's zeros: one run against aZ reference zeros
                const u64 S2 = (u64)rS2.result + 2ull * aZ * (u64)(my_n - neg_out) + zc * (2ull * nneg + aZ);
                const u64 t0 = (u64)aZ + zc;
                P.out_2u[o] = 2ll * (long long)n_ref * n_g - (long long)S2;
                P.out_tie[o] = T_A + 3ull * tt_out + (t0 * t0 * t0 - t0);
            }
        }
    }
    // ---- groups of more than 256 non-zero keys (dealt into value buckets by k_bucket_big_runs): one wavefront each, in pieces of at most
    // 256 keys cut at bucket boundaries -- every piece is a group of its own to ocr_group, and S2, the tie term and the negatives add up ----
    // A LONG run (a cluster of tens of thousands of cells: more than OCR_COOP_MIN keys) is walked by ALL the wavefronts together: the bucket
    // kernels left NW - 1 cuts per such run (bucket boundaries at or behind w / NW of the run: run_cuts), wavefront w walks the pieces between
    // cut w and cut w + 1, the stretches' sums meet in LDS.  (One wavefront per run left seven of sixteen idle on ten clusters, fifteen on a
    // single cluster against the reference.)
    if (P.big_sorted && s_cnt[3] > 64u * KMAX) {
        u32 *lr_list = ccnt;                 // (the coarse cells' counters are done with) [0] long runs listed, [1 ..] their groups
        u64 *lr_acc = (u64 *)(ccnt + 64);    // [0] S2 part  [1] tie part  [2] negatives  [3] keys inside the part  [4] a bucket above 256 keys
        if (tid == 0) lr_list[0] = 0u;
        __syncthreads();
        // the pieces of seg[i_begin, i_end) (both bucket boundaries): sums into the accumulators; false: a bucket of more than 256 keys
        auto walk = [&](const KeyT *seg, const BigRunFn<KeyT> &fn, int i_begin, int i_end, u64 &s2_acc, u64 &tt_acc, u32 &neg_acc, u32 &nv_acc) -> bool {
            const int n = i_end;
            bool bad = false;
            for (int s0 = i_begin; s0 < n && !bad;) {
                    const int win = min(64 * KMAX, n - s0);
                    KeyT cur[KMAX];
#pragma unroll
                    for (int r = 0; r < KMAX; ++r) cur[r] = (r * 64 + lane < win) ? seg[s0 + r * 64 + lane] : ZEROK;
                    int nB = win;
                    if (s0 + win < n) { // does the window's last bucket go on beyond it?  then the piece ends where that bucket starts
                        const u32 bn = big_bucket(fn, seg[s0 + win]); // (uniform address)
                        int same = 0;
#pragma unroll
                        for (int r = 0; r < KMAX; ++r) same += (int)__popcll(__ballot(r * 64 + lane < win && big_bucket(fn, cur[r]) == bn));
                        nB = win - same;
                    }
                    if (nB == 0) { bad = true; break; } // a bucket of more than 256 keys: a tie-heavy column, not for this kernel
#pragma unroll
                    for (int r = 0; r < KMAX; ++r) cur[r] = (r * 64 + lane < nB) ? cur[r] : ZEROK;
                    if constexpr (PARTS) { // the run is in ascending bucket order: most pieces lie wholly inside or outside the part
                        u32 nv = 0;
#pragma unroll
                        for (int r = 0; r < KMAX; ++r) {
                            cur[r] = (cur[r] >= p_lo && cur[r] <= p_hi) ? cur[r] : ZEROK;
                            nv += (u32)__popcll(__ballot(cur[r] != ZEROK));
                        }
                        nv_acc += nv;
                        if (nv == 0u) { s0 += nB; continue; } // (uniform)
                    }
                    u32 less = 0, eqs = 0, negs = 0;
                    u64 TT = 0;
                    if (nB <= 64) ocr_group<KeyT, 1, EQ, PARTS>(cur, nB, R, bloom, lane, lt_mask, less, eqs, TT, negs);
                    else if (nB <= 128) ocr_group<KeyT, 2, EQ, PARTS>(cur, nB, R, bloom, lane, lt_mask, less, eqs, TT, negs);
                    else if (nB <= 192) ocr_group<KeyT, 3, EQ, PARTS>(cur, nB, R, bloom, lane, lt_mask, less, eqs, TT, negs);
                    else ocr_group<KeyT, 4, EQ, PARTS>(cur, nB, R, bloom, lane, lt_mask, less, eqs, TT, negs);
                    s2_acc += 2ull * less + eqs;
                    tt_acc += TT;
                    neg_acc += negs;
                    s0 += nB;
            }
            return !bad;
        };
        // a big group's statistics from its run's sums (one lane)
        auto emit_big = [&](int gb, int n, u64 s2_acc, u64 tt_acc, u32 neg_acc, u32 nv_acc) {
            const size_t o = (size_t)gene * G + gb;
            const long long n_g = P.counts[gb];
            if constexpr (PARTS) {
                u64 S2 = s2_acc + 2ull * n_low * (u64)nv_acc + 2ull * aZ * (u64)(nv_acc - neg_acc);
                u64 tie = T_A + 3ull * tt_acc;
                long long two_u = 0;
                if (part == 0) {
                    const u64 zc = (u64)(n_g - (long long)n), t0 = (u64)aZ + zc;
                    S2 += zc * (2ull * nneg + aZ);
                    tie += t0 * t0 * t0 - t0;
                    two_u = 2ll * (long long)n_ref * n_g;
                }
                atomicAdd((unsigned long long *)&P.out_2u[o], (unsigned long long)(two_u - (long long)S2));
                atomicAdd((unsigned long long *)&P.out_tie[o], (unsigned long long)tie);
            } else {
                const u64 zc = (u64)(n_g - (long long)n);
                const u64 S2 = s2_acc + 2ull * aZ * (u64)((u32)n - neg_acc) + zc * (2ull * nneg + aZ);
                const u64 t0 = (u64)aZ + zc;
                P.out_2u[o] = 2ll * (long long)n_ref * n_g - (long long)S2;
                P.out_tie[o] = T_A + 3ull * tt_acc + (t0 * t0 * t0 - t0);
            }
        };
        for (int gb = wave; gb < G; gb += NW) { // (uniform per wavefront; few groups are big)
            if (gb == ref) continue;
            int n = (int)nnz[gb];
            if (n <= 64 * KMAX) continue;
            if (n >= 65535 && P.run_n) n = (int)P.run_n[(size_t)gene * P.n_cand + P.cand_of[gb]]; // (the 16-bit length saturated)
            if (n > OCR_COOP_MIN && P.run_cuts) { // (at most 63 listed: the others are walked by their wavefront alone, below)
                u32 slot = 0;
                if (lane == 0) slot = atomicAdd(&lr_list[0], 1u);
                slot = (u32)__builtin_amdgcn_readfirstlane((int)slot);
                if (slot < 63u) { if (lane == 0) lr_list[1 + slot] = (u32)gb; continue; }
            }
            BigRunFn<KeyT> fn = ((const BigRunFn<KeyT> *)P.big_fn)[(size_t)gene * P.n_cand + P.cand_of[gb]];
            const KeyT *seg = ((fn.shift & BIG_RUN_IN_TMP) ? (const KeyT *)P.big_tmp + (long long)gene * P.gene_stride : (const KeyT *)Xg) + P.gofs[(size_t)gene * G + gb];
            fn.shift &= ~BIG_RUN_IN_TMP;
            u64 s2_acc = 0, tt_acc = 0; // per-lane partial sums
            u32 neg_acc = 0;            // (uniform)
            u32 nv_acc = 0;             // (uniform) PARTS: keys inside the part
            const bool bad = !walk(seg, fn, 0, n, s2_acc, tt_acc, neg_acc, nv_acc);
            // (dense layout: the reference's segments lie where they were -- not k_ovo_rank's gene: a word above 255 sends it to the general route)
            if (bad) { if (lane == 0) P.route[gene] = P.ref_by_gofs ? 2u : (2u | (6u << 8)); continue; }
            s2_acc = wave_sum<u64>(s2_acc);
            tt_acc = wave_sum<u64>(tt_acc);
            if (lane == 0) emit_big(gb, n, s2_acc, tt_acc, neg_acc, nv_acc);
        }
        // ---- the long runs, every wavefront on each ----
        __syncthreads();
        const int n_long = (int)min(lr_list[0], 63u);
        for (int li = 0; li < n_long; ++li) { // (uniform)
            const int gb = (int)lr_list[1 + li];
            if (tid < 5) lr_acc[tid] = 0ull;
            __syncthreads();
            int n = (int)nnz[gb];
            if (n >= 65535 && P.run_n) n = (int)P.run_n[(size_t)gene * P.n_cand + P.cand_of[gb]];
            const size_t ci = (size_t)gene * P.n_cand + P.cand_of[gb];
            BigRunFn<KeyT> fn = ((const BigRunFn<KeyT> *)P.big_fn)[ci];
            const KeyT *seg = ((fn.shift & BIG_RUN_IN_TMP) ? (const KeyT *)P.big_tmp + (long long)gene * P.gene_stride : (const KeyT *)Xg) + P.gofs[(size_t)gene * G + gb];
            fn.shift &= ~BIG_RUN_IN_TMP;
            const u32 *cuts = P.run_cuts + ci * OCR_CUTS; // cut w (w = 1 .. OCR_CUTS): the bucket boundary at or behind w / (OCR_CUTS + 1) of the run
            // NW wavefronts over OCR_CUTS + 1 = 16 stretches: wavefront w takes stretches w * 16 / NW ... (NW = 16: one each; NW = 4: four each)
            const int st0 = wave * (OCR_CUTS + 1) / NW, st1 = (wave + 1) * (OCR_CUTS + 1) / NW;
            const int i0 = st0 == 0 ? 0 : min((int)cuts[st0 - 1], n), i1 = st1 > OCR_CUTS ? n : min((int)cuts[st1 - 1], n);
            u64 s2_acc = 0, tt_acc = 0;
            u32 neg_acc = 0, nv_acc = 0;
            bool ok = true;
            if (i0 < i1) ok = walk(seg, fn, i0, i1, s2_acc, tt_acc, neg_acc, nv_acc);
            s2_acc = wave_sum<u64>(s2_acc);
            tt_acc = wave_sum<u64>(tt_acc);
            if (lane == 0) {
                if (s2_acc) atomicAdd((unsigned long long *)&lr_acc[0], (unsigned long long)s2_acc);
                if (tt_acc) atomicAdd((unsigned long long *)&lr_acc[1], (unsigned long long)tt_acc);
                if (neg_acc) atomicAdd((unsigned long long *)&lr_acc[2], (unsigned long long)neg_acc);
                if (nv_acc) atomicAdd((unsigned long long *)&lr_acc[3], (unsigned long long)nv_acc);
                if (!ok) lr_acc[4] = 1ull;
            }
            __syncthreads();
            if (tid == 0) {
                if (lr_acc[4]) P.route[gene] = P.ref_by_gofs ? 2u : (2u | (6u << 8));
                else emit_big(gb, n, lr_acc[0], lr_acc[1], (u32)lr_acc[2], (u32)lr_acc[3]);
            }
            __syncthreads();
        }
    }
}
