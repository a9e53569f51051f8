// CSC / CSR ingest: one gene batch of a sparse matrix -> the engine's segmented layout
//   Xs   keys of the stored non-zeros, gene-major, and inside a gene grouped by group code
//   vals group code of each key (only materialised for OVR)
//   seg_ptr[gene][0..G]  offsets into Xs of each (gene, group) run
// which is what k_ovo_rank / k_ovr_gene consume (zeros stay implicit and are ranked analytically,
// as the reference does: ovo/sparse_ovo.py:58-85, ovr/sparse_ovr.py:70-83).
//
// This is the device counterpart of the reference's per-chunk counting-sort helpers
//   csc_get_contig_cols_into_csr + csr_get_rows_into_csc   (utils/sparse/csc.py:139-183, csr.py:103-141)
//   csr_get_contig_cols_into_csr / _into_csc               (utils/sparse/csr.py:144-257)
// -- only the requested gene batch is regrouped, the matrix is never converted as a whole.
// Explicitly stored zeros are dropped (they are zeros), so results equal the dense path's.
#pragma once
#include "common.h"
#include "kernels_ovo.h"

// exclusive scan of arr[0..n) in place by the whole workgroup; returns the total.  tmp: [NT] words of LDS.
template <int NT> __device__ __forceinline__ u32 block_excl_scan_inplace(u32 *arr, int n, u32 *tmp, int tid) {
    const int per = (n + NT - 1) / NT;
    const int b = min(tid * per, n), e = min(b + per, n);
    u32 s = 0;
    for (int i = b; i < e; ++i) s += arr[i];
    tmp[tid] = s;
    __syncthreads();
    // Hillis-Steele over NT partial sums
    for (int d = 1; d < NT; d <<= 1) {
        u32 v = (tid >= d) ? tmp[tid - d] : 0u;
        __syncthreads();
        tmp[tid] += v;
        __syncthreads();
    }
    u32 total = tmp[NT - 1];
    u32 run = tmp[tid] - s;
    for (int i = b; i < e; ++i) { u32 c = arr[i]; arr[i] = run; run += c; }
    __syncthreads();
    return total;
}

#define SEG_NT 256

// CSC: one workgroup per gene of the batch.  LDS: hist[G] + tmp[NT].
template <typename InT, typename IdxT, typename KeyT>
__global__ __launch_bounds__(SEG_NT) void k_csc_segment(const InT *__restrict__ data, const IdxT *__restrict__ indices,
                                                        const IdxT *__restrict__ indptr, long long col0, int nb,
                                                        const int *__restrict__ codes, int G, KeyT *__restrict__ Xs,
                                                        u32 *__restrict__ vals, u32 *__restrict__ seg_ptr,
                                                        u32 *__restrict__ gene_flags, int count_limit,
                                                        const int *__restrict__ gene_cols, const u32 *__restrict__ gene_base,
                                                        long long kshift, const u32 *__restrict__ only_flagged) {
    extern __shared__ __align__(16) unsigned char smem[];
    u32 *hist = (u32 *)smem;
    u32 *tmp = hist + ((G + 3) & ~3);
    const int tid = threadIdx.x;
    // batch = columns col0 .. col0+nb-1, or (gene_cols != nullptr) an arbitrary list of columns whose keys are
    // packed at gene_base[gene].  Stored entry k of the caller's matrix lives at data[k - kshift] / indices[k - kshift]
    // (a host-resident matrix is staged from its first needed entry on).
    const long long base0 = (long long)indptr[col0];
    for (int gene = blockIdx.x; gene < nb; gene += gridDim.x) {
        if (only_flagged && only_flagged[gene] == 0) continue; // already regrouped by k_csc_regroup
        const long long col = gene_cols ? (long long)gene_cols[gene] : col0 + gene;
        const long long k0 = (long long)indptr[col], k1 = (long long)indptr[col + 1];
        const u32 gbase = gene_cols ? gene_base[gene] : (u32)(k0 - base0);
        for (int g = tid; g < G; g += SEG_NT) hist[g] = 0;
        __syncthreads();
        constexpr int UL = 8; // independent entries per thread in flight (value, row -> group code)
        for (long long kb = k0 - kshift; kb < k1 - kshift; kb += SEG_NT * UL) {
            InT v[UL];
            int cd[UL];
#pragma unroll
            for (int u = 0; u < UL; ++u) {
                const long long k = kb + u * SEG_NT + tid;
                const bool in = k < k1 - kshift;
                v[u] = in ? data[k] : (InT)0;
                cd[u] = in ? (codes ? codes[(long long)indices[k]] : (int)indices[k]) : 0;
            }
#pragma unroll
            for (int u = 0; u < UL; ++u)
                if (v[u] != (InT)0) atomicAdd(&hist[cd[u]], 1u);
        }
        __syncthreads();
        u32 total = block_excl_scan_inplace<SEG_NT>(hist, G, tmp, tid);
        u32 *sp = seg_ptr + (size_t)gene * (G + 1);
        for (int g = tid; g < G; g += SEG_NT) sp[g] = gbase + hist[g];
        if (tid == 0) sp[G] = gbase + total;
        __syncthreads();
        bool viol = false;
        for (long long kb = k0 - kshift; kb < k1 - kshift; kb += SEG_NT * UL) {
            InT v[UL];
            int cd[UL];
#pragma unroll
            for (int u = 0; u < UL; ++u) {
                const long long k = kb + u * SEG_NT + tid;
                const bool in = k < k1 - kshift;
                v[u] = in ? data[k] : (InT)0;
                cd[u] = in ? (codes ? codes[(long long)indices[k]] : (int)indices[k]) : 0;
            }
#pragma unroll
            for (int u = 0; u < UL; ++u)
                if (v[u] != (InT)0) {
                    const u32 pos = gbase + atomicAdd(&hist[cd[u]], 1u);
                    Xs[pos] = key_of(v[u]);
                    if (vals) vals[pos] = (u32)cd[u];
                    viol |= !count_ok(v[u], count_limit);
                }
        }
        if (gene_flags && viol) gene_flags[gene] = 1u;
        __syncthreads();
    }
}

// CSR pass 1/3: count non-zeros per (gene, group) of the batch columns [c0, c1).  One wavefront per row;
// the row's sorted column indices are binary-searched for the window (as csr.py:172,226 do).
template <typename InT, typename IdxT>
__global__ __launch_bounds__(256) void k_csr_count(const InT *__restrict__ data, const IdxT *__restrict__ indices,
                                                   const IdxT *__restrict__ indptr, int n_rows, long long c0, long long c1,
                                                   const int *__restrict__ codes, int G, u32 *__restrict__ cnt) {
    const int lane = threadIdx.x & 63;
    const int wave_global = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const int n_waves = (gridDim.x * blockDim.x) >> 6;
    for (int row = wave_global; row < n_rows; row += n_waves) {
        long long s = (long long)indptr[row], e = (long long)indptr[row + 1];
        long long lo = s, hi = e;
        while (lo < hi) { long long m = (lo + hi) >> 1; if ((long long)indices[m] < c0) lo = m + 1; else hi = m; }
        long long a = lo;
        hi = e;
        while (lo < hi) { long long m = (lo + hi) >> 1; if ((long long)indices[m] < c1) lo = m + 1; else hi = m; }
        long long b = lo;
        const int c = codes[row];
        for (long long k = a + lane; k < b; k += 64)
            if (data[k] != (InT)0) atomicAdd(&cnt[((long long)indices[k] - c0) * (G + 1) + c], 1u);
    }
}

// CSR pass 2/3: per gene exclusive scan over groups (in place) + gene totals
static __global__ __launch_bounds__(SEG_NT) void k_seg_scan(u32 *__restrict__ seg, int G, int nb, u32 *__restrict__ gene_tot) {
    extern __shared__ __align__(16) unsigned char smem[];
    u32 *hist = (u32 *)smem;
    u32 *tmp = hist + ((G + 3) & ~3);
    const int tid = threadIdx.x;
    for (int gene = blockIdx.x; gene < nb; gene += gridDim.x) {
        u32 *sp = seg + (size_t)gene * (G + 1);
        for (int g = tid; g < G; g += SEG_NT) hist[g] = sp[g];
        __syncthreads();
        u32 total = block_excl_scan_inplace<SEG_NT>(hist, G, tmp, tid);
        for (int g = tid; g < G; g += SEG_NT) sp[g] = hist[g];
        if (tid == 0) { sp[G] = total; gene_tot[gene] = total; }
        __syncthreads();
    }
}
// exclusive scan of the gene totals by one workgroup (nb is at most a few thousand)
static __global__ __launch_bounds__(1024) void k_gene_base_scan(const u32 *__restrict__ gene_tot, int nb, u32 *__restrict__ gene_base) {
    __shared__ u32 tmp[1024];
    __shared__ u32 carry;
    const int tid = threadIdx.x;
    if (tid == 0) carry = 0;
    __syncthreads();
    for (int b0 = 0; b0 < nb; b0 += 1024) {
        int i = b0 + tid;
        u32 v = i < nb ? gene_tot[i] : 0u;
        tmp[tid] = v;
        __syncthreads();
        for (int d = 1; d < 1024; d <<= 1) {
            u32 o = tid >= d ? tmp[tid - d] : 0u;
            __syncthreads();
            tmp[tid] += o;
            __syncthreads();
        }
        if (i < nb) gene_base[i] = carry + tmp[tid] - v;
        __syncthreads();
        if (tid == 1023) carry += tmp[1023];
        __syncthreads();
    }
}
// seg[gene][g] += gene_base[gene]; cursor = copy
static __global__ void k_seg_add_base(u32 *__restrict__ seg, u32 *__restrict__ cursor, const u32 *__restrict__ gene_base, int G, int nb) {
    long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    long long tot = (long long)nb * (G + 1);
    if (i >= tot) return;
    int gene = (int)(i / (G + 1));
    u32 v = seg[i] + gene_base[gene];
    seg[i] = v;
    cursor[i] = v;
}

// CSR pass 3/3: scatter keys (and group codes) to their runs
template <typename InT, typename IdxT, typename KeyT>
__global__ __launch_bounds__(256) void k_csr_scatter(const InT *__restrict__ data, const IdxT *__restrict__ indices,
                                                     const IdxT *__restrict__ indptr, int n_rows, long long c0, long long c1,
                                                     const int *__restrict__ codes, int G, u32 *__restrict__ cursor,
                                                     KeyT *__restrict__ Xs, u32 *__restrict__ vals,
                                                     u32 *__restrict__ gene_flags, int count_limit) {
    const int lane = threadIdx.x & 63;
    const int wave_global = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const int n_waves = (gridDim.x * blockDim.x) >> 6;
    for (int row = wave_global; row < n_rows; row += n_waves) {
        long long s = (long long)indptr[row], e = (long long)indptr[row + 1];
        long long lo = s, hi = e;
        while (lo < hi) { long long m = (lo + hi) >> 1; if ((long long)indices[m] < c0) lo = m + 1; else hi = m; }
        long long a = lo;
        hi = e;
        while (lo < hi) { long long m = (lo + hi) >> 1; if ((long long)indices[m] < c1) lo = m + 1; else hi = m; }
        long long b = lo;
        const int c = codes[row];
        for (long long k = a + lane; k < b; k += 64) {
            InT v = data[k];
            if (v != (InT)0) {
                const long long gcol = (long long)indices[k] - c0;
                u32 pos = atomicAdd(&cursor[gcol * (G + 1) + c], 1u);
                Xs[pos] = key_of(v);
                if (vals) vals[pos] = (u32)c;
                if (gene_flags && !count_ok(v, count_limit) && gene_flags[gcol] == 0) gene_flags[gcol] = 1u;
            }
        }
    }
}

// non-zeros per column of a CSR matrix restricted to [c0, c1)  (batch planning)
template <typename IdxT>
__global__ void k_csr_col_nnz(const IdxT *__restrict__ indices, long long nnz, long long c0, long long c1, u32 *__restrict__ col_cnt) {
    long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    long long stride = (long long)gridDim.x * blockDim.x;
    for (; i < nnz; i += stride) {
        long long c = (long long)indices[i];
        if (c >= c0 && c < c1) atomicAdd(&col_cnt[c - c0], 1u);
    }
}

// Of n_samples evenly spaced stored values: n_bad[0] = how many are not non-negative integers (normalised / log1p data:
// no count route applies), n_bad[1] = how many are integers of `limit` or more (their genes leave the table-based
// kernels one by one).  Route choice only: every route is exact.
template <typename InT>
__global__ void k_sample_noncount(const InT *__restrict__ data, long long nnz, int n_samples, int limit, u32 *__restrict__ n_bad) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_samples) return;
    const long long k = (long long)((double)i * (double)nnz / (double)n_samples);
    const InT v = data[k < nnz ? k : nnz - 1];
    const bool integer = v >= (InT)0 && v < (InT)(1 << 24) && (InT)(int)v == v;
    // one atomic per wavefront and counter: on normalised data every sample would otherwise hit the same address
    const u64 b0 = __ballot(!integer), b1 = __ballot(integer && v >= (InT)limit);
    if ((threadIdx.x & 63) == 0) {
        if (b0) atomicAdd(n_bad, (u32)__popcll(b0));
        if (b1) atomicAdd(n_bad + 1, (u32)__popcll(b1));
    }
}

// The same over the stored entries of the CSC columns [lb, ub), with the window's bounds read from indptr ON THE DEVICE: the
// driver enqueues it together with its own copy of indptr and waits once for both (n_bad[2] = the samples taken).
template <typename InT, typename IdxT>
__global__ void k_sample_noncount_cols(const InT *__restrict__ data, const IdxT *__restrict__ indptr, long long lb, long long ub, int max_samples,
                                       int limit, u32 *__restrict__ n_bad) {
    const long long k0 = (long long)indptr[lb], nnz = (long long)indptr[ub] - k0;
    const int n_samples = (int)(nnz < (long long)max_samples ? nnz : (long long)max_samples);
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i == 0) n_bad[2] = (u32)n_samples;
    if (i >= n_samples) return;
    const long long k = (long long)((double)i * (double)nnz / (double)n_samples);
    const InT v = data[k0 + (k < nnz ? k : nnz - 1)];
    const bool integer = v >= (InT)0 && v < (InT)(1 << 24) && (InT)(int)v == v;
    const u64 b0 = __ballot(!integer), b1 = __ballot(integer && v >= (InT)limit);
    if ((threadIdx.x & 63) == 0) {
        if (b0) atomicAdd(n_bad, (u32)__popcll(b0));
        if (b1) atomicAdd(n_bad + 1, (u32)__popcll(b1));
    }
}

// ---- CSR -> CSC on the device, for a window of W columns [c0, c0 + W): a two-pass blocked transposition.  Row blocks
// of TR_RB rows are contiguous runs of the CSR arrays, so both passes read them coalesced; per block an LDS table over
// the window's columns counts (pass 1) or hands out positions (pass 2), and an entry's destination lies in a run of its
// (block, column) pair -- ~190 neighbouring entries at C3 -- instead of anywhere in the column.  The columns come out
// with their rows grouped by block (ascending) but unordered inside a block; the CSC kernels do not need row order.
// Device counterpart of csr_get_contig_cols_into_csc (utils/sparse/csr.py:19-100).
#define TR_NT 256
#define TRC_NT 1024 // the counting pass: a row block is one workgroup, and there are only n_rows / 512 of them -- 16 wavefronts each
// (16-bit counters, two columns per LDS word -- a (block, column) pair holds at most RB <= 512 entries: 65 536 columns in 128 KB, so a
//  matrix of 33 000 - 65 000 genes is ONE window, not two sequential passes over every stored entry)
template <typename IdxT>
__global__ __launch_bounds__(TRC_NT) void k_csr_block_count(const IdxT *__restrict__ indices, const IdxT *__restrict__ indptr, int n_rows,
                                                          int RB, long long c0, int W, u32 *__restrict__ counts) {
    extern __shared__ u32 tr_cnt[]; // [(W + 1) / 2]: column 2 i in the low half of word i, 2 i + 1 in the high half
    const int tid = threadIdx.x, W2 = (W + 1) >> 1;
    for (int i = tid; i < W2; i += TRC_NT) tr_cnt[i] = 0;
    __syncthreads();
    const int r0 = blockIdx.x * RB, r1 = min(r0 + RB, n_rows);
    long long k0 = (long long)indptr[r0], k1 = (long long)indptr[r1];
    if (gridDim.y > 1) { // few row blocks (a matrix of few cells and many genes): the block's entries in gridDim.y slices, the counts added up in `counts` (zeroed by the host)
        const long long per = (k1 - k0 + gridDim.y - 1) / gridDim.y;
        k0 += per * blockIdx.y;
        k1 = min(k1, k0 + per);
    }
    // eight requests in flight per thread (one at a time, this pass ran at 1.2 TB/s: 0.77 ms at C3 shape for 0.9 GB of column indices)
    for (long long kb = k0; kb < k1; kb += TRC_NT * 8) {
        long long col[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const long long k = kb + u * TRC_NT + tid;
            col[u] = k < k1 ? (long long)indices[k] - c0 : -1;
        }
#pragma unroll
        for (int u = 0; u < 8; ++u)
            if (col[u] >= 0 && col[u] < W) atomicAdd(&tr_cnt[col[u] >> 1], (col[u] & 1) ? 0x10000u : 1u);
    }
    __syncthreads();
    u32 *dst = counts + (size_t)blockIdx.x * W;
    if (gridDim.y > 1) {
        for (int i = tid; i < W; i += TRC_NT) { const u32 n = (tr_cnt[i >> 1] >> ((i & 1) * 16)) & 0xFFFFu; if (n) atomicAdd(&dst[i], n); }
    } else {
        for (int i = tid; i < W; i += TRC_NT) dst[i] = (tr_cnt[i >> 1] >> ((i & 1) * 16)) & 0xFFFFu;
    }
}
// per column: exclusive scan of the block counts (in place) and the column's total
static __global__ void k_col_block_scan(u32 *__restrict__ counts, int n_blocks, int W, u32 *__restrict__ col_total) {
    const int col = blockIdx.x * blockDim.x + threadIdx.x;
    if (col >= W) return;
    u32 run = 0;
    int b = 0;
    for (; b + 8 <= n_blocks; b += 8) {
        u32 t[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) t[j] = counts[(size_t)(b + j) * W + col];
#pragma unroll
        for (int j = 0; j < 8; ++j) { counts[(size_t)(b + j) * W + col] = run; run += t[j]; }
    }
    for (; b < n_blocks; ++b) { const u32 t = counts[(size_t)b * W + col]; counts[(size_t)b * W + col] = run; run += t; }
    col_total[col] = run;
}
template <typename InT, typename IdxT>
__global__ __launch_bounds__(TR_NT) void k_csr_block_scatter(const InT *__restrict__ data, const IdxT *__restrict__ indices,
                                                            const IdxT *__restrict__ indptr, int n_rows, int RB, long long c0, int W,
                                                            const u32 *__restrict__ offsets, const u32 *__restrict__ col_ptr,
                                                            const int *__restrict__ row_codes, InT *__restrict__ out_data,
                                                            int *__restrict__ out_rows) {
    extern __shared__ u32 tr_cur[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const u32 *off = offsets + (size_t)blockIdx.x * W;
    for (int i = tid; i < W; i += TR_NT) tr_cur[i] = col_ptr[i] + off[i];
    __syncthreads();
    const int r0 = blockIdx.x * RB, r1 = min(r0 + RB, n_rows);
    for (int r = r0 + wave; r < r1; r += TR_NT / 64) {
        const int tag = row_codes ? row_codes[r] : r; // what the CSC kernels need of a row is its group code
        const long long k1 = (long long)indptr[r + 1];
        for (long long k = (long long)indptr[r] + lane; k < k1; k += 64) {
            const long long col = (long long)indices[k] - c0;
            const InT v = data[k];
            if (col >= 0 && col < W) {
                const u32 pos = atomicAdd(&tr_cur[col], 1u);
                out_data[pos] = v;
                out_rows[pos] = tag;
            }
        }
    }
}

// Pass 2 for rows with SORTED column indices (the reference requires them, tests/test_asymptotic_wilcoxon.py:259-273):
// one workgroup per row block sweeps the window left to right in tiles of TRG_COLS columns.  Every thread owns RPT
// rows and keeps a cursor into each: the entries of a row that fall into the current tile are the next few after the
// cursor (no search, and a row's cache lines are walked once).  The workgroup counting-sorts the tile's entries by
// column in LDS and writes every (block, column) run -- RB * density entries -- as one contiguous copy.
// k_csr_block_scatter above sends each entry straight to its final position instead: 8000 open write streams per
// workgroup, far more than L2 can merge (17 ms at C3 shape).
#define TRG_COLS 64
#define TRG_RPT 1
#define TRG_NT 512 // one row per thread, 512 rows per block (two rows per thread and 256 threads: half the wavefronts for the same LDS)
#ifndef TRG_WIN
#define TRG_WIN 16 // entries of a row held in registers: 8 / 16 / 32 -> 2.19 / 1.96 / 3.39 ms at C3 shape (32: 168 registers); requesting the
#endif              // next refill ahead of time changes nothing (8: 2.22, 16: 1.97): the pass is not waiting for its refills
template <typename T, int N> struct __attribute__((packed, aligned(4))) PackedRun { T v[N]; }; // 4-byte aligned multi-dword load
template <typename InT, typename IdxT>
__global__ __launch_bounds__(TRG_NT) void k_csr_tile_gather(const InT *__restrict__ data, const IdxT *__restrict__ indices,
                                                          const IdxT *__restrict__ indptr, int n_rows, int RB, long long c0, int W,
                                                          const u32 *__restrict__ offsets, const u32 *__restrict__ col_total,
                                                          const u32 *__restrict__ col_ptr, int cap, const int *__restrict__ row_codes,
                                                          InT *__restrict__ out_data, int *__restrict__ out_rows,
                                                          u32 *__restrict__ overflow) {
    extern __shared__ __align__(16) unsigned char trg_smem[];
    __shared__ u32 start[TRG_COLS + 1], cur[TRG_COLS], gbase[TRG_COLS];
    InT *sval = (InT *)trg_smem;
    int *srow = (int *)(trg_smem + (size_t)cap * sizeof(InT));
    unsigned char *scol = (unsigned char *)(srow + cap);
    constexpr int RPT = TRG_RPT, WN = TRG_WIN;
    const int tid = threadIdx.x;
    const int r0 = blockIdx.x * RB, r1 = min(r0 + RB, n_rows);
    // Per row: a register window of the next WN stored entries (column relative to c0, value), refilled with multi-dword
    // loads when it runs dry.  A thread walks ITS row, so its loads are uncoalesced across lanes (64 cache lines per
    // instruction): fetching 8 entries with 2 + 2 wide loads instead of 16 narrow ones is what keeps the texture
    // addresser from being the limiter (it was: 7 of 8 ms).
    // gridDim.y > 1 (few row blocks): the window's 64-column tiles in gridDim.y stretches, one workgroup each
    const int tiles_all = (W + TRG_COLS - 1) / TRG_COLS, tiles_per = (tiles_all + (int)gridDim.y - 1) / (int)gridDim.y;
    const int cb_lo = (int)blockIdx.y * tiles_per * TRG_COLS, cb_hi = min(W, cb_lo + tiles_per * TRG_COLS);
    long long knext[RPT], kend[RPT];
    int wcol[RPT][WN], wp[RPT], wn[RPT], tag[RPT]; // tag: what is stored for the row -- its group code, or the row index
    InT wval[RPT][WN];
    auto refill = [&](int j) { // the next WN entries of row j become its window; false: the row is finished
        const long long k = knext[j], left = kend[j] - k;
        if (left <= 0) return false;
        if (left >= WN) {
            const PackedRun<IdxT, WN> pi = *(const PackedRun<IdxT, WN> *)(indices + k);
            const PackedRun<InT, WN> pv = *(const PackedRun<InT, WN> *)(data + k);
#pragma unroll
            for (int i = 0; i < WN; ++i) { wcol[j][i] = (int)((long long)pi.v[i] - c0); wval[j][i] = pv.v[i]; }
        } else {
#pragma unroll
            for (int i = 0; i < WN; ++i) {
                const bool ok = i < left;
                wcol[j][i] = ok ? (int)((long long)indices[k + i] - c0) : 0x7FFFFFFF;
                wval[j][i] = ok ? data[k + i] : (InT)0;
            }
        }
        wp[j] = 0;
        wn[j] = (int)(left < WN ? left : WN);
        knext[j] = k + wn[j];
        return true;
    };
#pragma unroll
    for (int j = 0; j < RPT; ++j) {
        const int r = r0 + j * TRG_NT + tid;
        knext[j] = kend[j] = 0;
        wp[j] = wn[j] = 0;
        tag[j] = r;
        if (r < r1) {
            if (row_codes) tag[j] = row_codes[r];
            long long a = (long long)indptr[r], e = (long long)indptr[r + 1], b = e;
            const long long first = c0 + cb_lo; // (the first column of this workgroup's stretch of the window)
            while (a < b) { const long long m = (a + b) >> 1; if ((long long)indices[m] < first) a = m + 1; else b = m; } // window start
            knext[j] = a;
            kend[j] = e;
        }
    }
    // the (block, column) counts of pass 1 survive as differences of the scanned table: this block's offset against the
    // next block's (the column total for the last block)
    const u32 *off = offsets + (size_t)blockIdx.x * W;
    const u32 *off_next = (blockIdx.x + 1 < gridDim.x) ? off + W : col_total;
    for (int cb = cb_lo; cb < cb_hi; cb += TRG_COLS) {
        const int ncols = min(TRG_COLS, W - cb);
        const int chi = cb + ncols; // columns are kept relative to c0
        if (tid < 64) {
            const bool in = tid < ncols;
            const u32 o0 = in ? off[cb + tid] : 0u, o1 = in ? off_next[cb + tid] : 0u, cp = in ? col_ptr[cb + tid] : 0u;
            const u32 cv = o1 - o0;
            const u32 incl = (u32)wave_incl_scan_add((int)cv);
            start[tid] = incl - cv;
            cur[tid] = incl - cv;
            gbase[tid] = cp + o0 - (incl - cv); // destination of staged entry i of this column: gbase + i
            if (tid == 63) start[TRG_COLS] = incl;
        }
        __syncthreads();
        const u32 total = start[TRG_COLS];
        if (total > (u32)cap) { // uniform: the host redoes this window with the scatter kernel
            if (tid == 0) *overflow = 1u;
            return;
        }
#pragma unroll
        for (int j = 0; j < RPT; ++j) {
            const int r = r0 + j * TRG_NT + tid;
            bool more = true;
            while (more) {
#pragma unroll
                for (int i = 0; i < WN; ++i)
                    if (i >= wp[j] && i < wn[j]) {
                        if (more && wcol[j][i] < chi) {
                            const int col = wcol[j][i] - cb;
                            const u32 p = atomicAdd(&cur[col], 1u);
                            sval[p] = wval[j][i];
                            srow[p] = tag[j];
                            scol[p] = (unsigned char)col;
                            wp[j] = i + 1;
                        } else more = false;
                    }
                if (more) more = refill(j); // window consumed: the next entries, or the row is finished
            }
        }
        __syncthreads();
        for (u32 i = tid; i < total; i += TRG_NT) {
            const u32 dst = gbase[scol[i]] + i;
            out_data[dst] = sval[i];
            out_rows[dst] = srow[i];
        }
        __syncthreads();
    }
}

// CSR rows -> a dense float32 window D[n_rows][ldD] of columns [c0, c0 + W): the dense fused single-pass kernels then
// read the window once.  One workgroup builds one row at a time in LDS (zero, scatter the row's stored entries, copy
// out): HBM sees only coalesced 16-byte stores, never a partial-sector scatter.  A stored value that float32 cannot
// hold exactly becomes NaN, which makes the fused kernel hand that gene to the exact sparse route.
// Device counterpart of csr_get_contig_cols_into_csc (utils/sparse/csr.py:19-100) for count-valued windows: the
// reference regroups the chunk's non-zeros by column on the CPU; at the 5-10 % density of expression matrices a dense
// window (N x W x 4 bytes, written and read once at HBM speed) is cheaper on this machine than regrouping 8-byte
// (value, row) pairs with scattered stores.
#define DENS_NT 256
#define DENS_WB 8192 // columns per LDS row block
// OutT = float (the window holds any float32-exact value) or uint8_t (count windows for the fused kernels, which only take
// integers below 64 anyway: a quarter of the bytes written here and read there; 255 = not an integer in [0, 255)).
template <typename OutT, typename InT> __device__ __forceinline__ OutT dens_cell(InT v);
template <> __device__ __forceinline__ float dens_cell<float, float>(float v) { return v; }
template <> __device__ __forceinline__ float dens_cell<float, double>(double v) { float f = (float)v; return (double)f == v ? f : __int_as_float(0x7FC00000); }
template <> __device__ __forceinline__ float dens_cell<float, int32_t>(int32_t v) { float f = (float)v; return (v > -(1 << 24) && v < (1 << 24)) ? f : __int_as_float(0x7FC00000); }
template <> __device__ __forceinline__ float dens_cell<float, int64_t>(int64_t v) { float f = (float)v; return (v > -(1ll << 24) && v < (1ll << 24)) ? f : __int_as_float(0x7FC00000); }
template <> __device__ __forceinline__ uint8_t dens_cell<uint8_t, float>(float v) { const float m = __builtin_amdgcn_fmed3f(v, 0.0f, 255.0f); const u32 c = (u32)m; return ((float)c == v && c < 255u) ? (uint8_t)c : (uint8_t)255; }
template <> __device__ __forceinline__ uint8_t dens_cell<uint8_t, double>(double v) { const double m = fmin(fmax(v, 0.0), 255.0); const u32 c = (u32)m; return ((double)c == v && c < 255u) ? (uint8_t)c : (uint8_t)255; }
template <> __device__ __forceinline__ uint8_t dens_cell<uint8_t, int32_t>(int32_t v) { return (v >= 0 && v < 255) ? (uint8_t)v : (uint8_t)255; }
template <> __device__ __forceinline__ uint8_t dens_cell<uint8_t, int64_t>(int64_t v) { return (v >= 0 && v < 255) ? (uint8_t)v : (uint8_t)255; }

// (the window in the matrix's own type, for the dense routes that take any values: run_sparse_t's dense-ish continuous CSR branch)
template <> __device__ __forceinline__ double dens_cell<double, double>(double v) { return v; }
template <> __device__ __forceinline__ int32_t dens_cell<int32_t, int32_t>(int32_t v) { return v; }
template <> __device__ __forceinline__ int64_t dens_cell<int64_t, int64_t>(int64_t v) { return v; }

template <typename InT, typename IdxT, typename OutT>
__global__ __launch_bounds__(DENS_NT) void k_csr_densify(const InT *__restrict__ data, const IdxT *__restrict__ indices,
                                                         const IdxT *__restrict__ indptr, int n_rows, long long c0, int W,
                                                         OutT *__restrict__ D, long long ldD) {
    __shared__ __align__(16) OutT row[DENS_WB];
    constexpr int UL = 4; // stored entries per thread requested together (column and value loads are independent)
    constexpr int VW = 16 / (int)sizeof(OutT); // cells per 16-byte store
    const int tid = threadIdx.x;
    long long col[UL]; // column (relative to c0) of the entries in flight; < 0: none
    InT v[UL];
    auto fetch = [&](long long k0, long long e) {
#pragma unroll
        for (int j = 0; j < UL; ++j) {
            const long long k = k0 + j * DENS_NT + tid;
            col[j] = k < e ? (long long)indices[k] - c0 : -1;
            v[j] = k < e ? data[k] : (InT)0;
        }
    };
    auto scatter = [&](int cb, int wb) {
#pragma unroll
        for (int j = 0; j < UL; ++j) {
            const long long cj = col[j] - cb;
            if (col[j] >= 0 && cj >= 0 && cj < wb) row[cj] = dens_cell<OutT, InT>(v[j]); // not representable: sends the gene elsewhere
        }
    };
    int r = blockIdx.x;
    long long s = 0, e = 0;
    if (r < n_rows) { s = (long long)indptr[r]; e = (long long)indptr[r + 1]; fetch(s, e); }
    for (; r < n_rows; r += gridDim.x) {
        const int rn = r + gridDim.x;
        long long sn = 0, en = 0;
        if (rn < n_rows) { sn = (long long)indptr[rn]; en = (long long)indptr[rn + 1]; }
        for (int cb = 0; cb < W; cb += DENS_WB) {
            const int wb = min(DENS_WB, W - cb), wbv = (wb + VW - 1) / VW * VW;
            for (int i = tid * VW; i < wbv; i += DENS_NT * VW) *(uint4 *)&row[i] = make_uint4(0u, 0u, 0u, 0u);
            __syncthreads();
            if (cb > 0) fetch(s, e); // wide windows: the row is walked once per column block
            scatter(cb, wb);
            for (long long k0 = s + DENS_NT * UL; k0 < e; k0 += DENS_NT * UL) { fetch(k0, e); scatter(cb, wb); }
            __syncthreads();
            // the next row's first entries are requested before this row is copied out: their latency hides behind the stores
            if (cb + DENS_WB >= W) fetch(sn, en);
            OutT *dst = D + (long long)r * ldD + cb; // ldD and cb are multiples of 64 cells: 16-byte aligned
            for (int i = tid * VW; i < wbv; i += DENS_NT * VW) {
                if (i + VW <= wb) *(uint4 *)&dst[i] = *(const uint4 *)&row[i];
                else for (int j = i; j < wb; ++j) dst[j] = row[j];
            }
            __syncthreads();
        }
        s = sn; e = en;
    }
}

// CSC columns [c0, c0 + W) -> a dense row-major window D[n_rows][ldD] in the matrix's own type (run_sparse_t: CSC windows whose columns
// are longer than the per-gene LDS kernels hold take the dense routes).  The columns' row indices must ascend (checked by the caller:
// k_csr_sorted_check over the CSC arrays).  Workgroup = (64 columns, CDN_SUP row chunks of CDN_RC rows): a lane finds its column's first
// entry of the stretch once (binary search), then the columns' cursors walk on chunk by chunk -- a chunk's entries are scattered into an
// LDS tile [row][column] (one pad word per row: a column's rows fall into different banks) and the tile leaves as full 64-cell rows.
#define CDN_NT 256
#define CDN_SUP 8
template <typename InT, typename IdxT, int RC>
__global__ __launch_bounds__(CDN_NT) void k_csc_densify(const InT *__restrict__ data, const IdxT *__restrict__ indices, const IdxT *__restrict__ indptr,
                                                       long long kshift, long long c0, int W, int n_rows, InT *__restrict__ D, long long ldD) {
    __shared__ InT tile[RC][65];
    __shared__ long long cur[64], kend[64];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int cb = blockIdx.x * 64;
    const long long row_first = (long long)blockIdx.y * (RC * CDN_SUP);
    if (tid < 64) {
        long long a = 0, e = 0;
        if (cb + tid < W) {
            a = (long long)indptr[c0 + cb + tid] - kshift;
            e = (long long)indptr[c0 + cb + tid + 1] - kshift;
            long long lo = a, hi = e;
            while (lo < hi) { const long long m = (lo + hi) >> 1; if ((long long)indices[m] < row_first) lo = m + 1; else hi = m; }
            a = lo;
        }
        cur[tid] = a;
        kend[tid] = e;
    }
    for (int ch = 0; ch < CDN_SUP; ++ch) {
        const long long r0 = row_first + (long long)ch * RC, r1 = min(r0 + RC, (long long)n_rows);
        if (r0 >= n_rows) break; // (uniform)
        for (int i = tid; i < RC * 65; i += CDN_NT) (&tile[0][0])[i] = (InT)0;
        __syncthreads();
        // a wavefront takes its sixteen columns at once, 64 entries of each per step: their entries of this chunk are the next ones behind
        // the cursors (rows and values of all of them requested together; 128 per step read the arrays 3.4 times over at 30 % stored)
        constexpr int CQ = 16;
        for (int jb = wave * 16; jb < wave * 16 + 16; jb += CQ) {
            long long k[CQ], e[CQ];
#pragma unroll
            for (int q = 0; q < CQ; ++q) { k[q] = cur[jb + q]; e[q] = kend[jb + q]; }
            bool more = true;
            while (more) { // (uniform per wavefront)
                IdxT row[CQ];
                bool have[CQ];
                InT val[CQ]; // (requested with the rows, not behind the test that needs the row: a round trip per column otherwise)
#pragma unroll
                for (int q = 0; q < CQ; ++q) {
                    const long long kk = k[q] + lane;
                    have[q] = kk < e[q];
                    row[q] = have[q] ? indices[kk] : (IdxT)0;
                    val[q] = have[q] ? data[kk] : (InT)0;
                }
                more = false;
#pragma unroll
                for (int q = 0; q < CQ; ++q) {
                    const bool in = have[q] && (long long)row[q] < r1;
                    if (in) tile[(long long)row[q] - r0][jb + q] = val[q];
                    const int n_in = (int)__popcll(__ballot(in));
                    k[q] += n_in;
                    more = more || n_in == 64;
                }
            }
#pragma unroll
            for (int q = 0; q < CQ; ++q)
                if (lane == q) cur[jb + q] = k[q];
        }
        __syncthreads();
        for (int r = wave; r < (int)(r1 - r0); r += CDN_NT / 64)
            if (cb + lane < (int)ldD) D[(size_t)(r0 + r) * ldD + cb + lane] = tile[r][lane];
        __syncthreads();
    }
}

// The same question for FEW, LONG parcels (the columns of a CSC window with tens of thousands of stored entries each, where a wavefront
// per parcel leaves most of the chip idle): the stored entries [k0, k1) as one flat run -- order[0] += the positions whose index is
// not larger than the one before it, order[1] += the parcels (after the first) that start with such a step.  In order <=> the two agree.
template <typename IdxT>
__global__ __launch_bounds__(256) void k_flat_descents(const IdxT *__restrict__ indices, const IdxT *__restrict__ indptr, int n_parcels, long long kshift,
                                                       u32 *__restrict__ order) {
    const long long k0 = (long long)indptr[0] - kshift, k1 = (long long)indptr[n_parcels] - kshift;
    const long long stride = (long long)gridDim.x * blockDim.x;
    u32 d = 0, r = 0;
    // (<=: an index stored twice in a parcel counts as out of order too -- a dense window would keep one of the two entries)
    for (long long k = k0 + 1 + (long long)blockIdx.x * blockDim.x + threadIdx.x; k < k1; k += stride) d += indices[k] <= indices[k - 1] ? 1u : 0u;
    for (long long j = 1 + (long long)blockIdx.x * blockDim.x + threadIdx.x; j < n_parcels; j += stride) {
        const long long s = (long long)indptr[j] - kshift, e = (long long)indptr[j + 1] - kshift;
        if (s < e && s > k0 && indices[s] <= indices[s - 1]) ++r;
    }
    d = (u32)wave_sum((int)d);
    r = (u32)wave_sum((int)r);
    if ((threadIdx.x & 63) == 0) {
        if (d) atomicAdd(&order[0], d);
        if (r) atomicAdd(&order[1], r);
    }
}

// float64 values that are all float32 values (a float32 matrix widened somewhere on its way): *inexact != 0 if one is not.  Such a matrix
// ranks, sums and divides to the same bits through the float32 kernels -- whose per-gene LDS buffers hold twice the keys.
static __global__ __launch_bounds__(256) void k_f64_is_f32(const double *__restrict__ v, long long n, u32 *__restrict__ inexact) {
    bool bad = false;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
        const double x = v[i];
        bad |= !((double)(float)x == x); // (NaN: not taken)
    }
    if (__any(bad) && (threadIdx.x & 63) == 0) *inexact = 1u;
}
static __global__ __launch_bounds__(256) void k_f64_to_f32(const double *__restrict__ v, long long n, float *__restrict__ out) {
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) out[i] = (float)v[i];
}

// stored count values that travelled to the device as bytes (sparse_driver.h: upload_values_as_bytes) back in the matrix's own type
template <typename InT>
__global__ __launch_bounds__(256) void k_bytes_to_values(const uint8_t *__restrict__ b, long long n, InT *__restrict__ out) {
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) out[i] = (InT)b[i];
}

// replaces check_indices_sorted_per_parcel (utils/ranking.py:245-273) for device-resident CSR
template <typename IdxT>
__global__ void k_csr_sorted_check(const IdxT *__restrict__ indices, const IdxT *__restrict__ indptr, int n_rows, int *__restrict__ bad) {
    const int lane = threadIdx.x & 63;
    const int wave_global = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const int n_waves = (gridDim.x * blockDim.x) >> 6;
    // four rows of a wavefront in flight at once (a row holds a few hundred entries: one row at a time, most of the pass is waiting)
    for (int row0 = wave_global * 4; row0 < n_rows; row0 += n_waves * 4) {
        long long s[4], e[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int row = min(row0 + j, n_rows - 1);
            s[j] = (long long)indptr[row];
            e[j] = row0 + j < n_rows ? (long long)indptr[row + 1] : s[j];
        }
        long long span = 0;
#pragma unroll
        for (int j = 0; j < 4; ++j) span = max(span, e[j] - s[j]);
        bool out_of_order = false;
        for (long long i = 1 + lane; i < span; i += 64) { // (uniform trip count)
            IdxT a[4], b[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const bool ok = s[j] + i < e[j];
                a[j] = ok ? indices[s[j] + i] : (IdxT)0;
                b[j] = ok ? indices[s[j] + i - 1] : (IdxT)0;
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) out_of_order |= a[j] < b[j];
        }
        if (out_of_order) *bad = 1;
    }
}
