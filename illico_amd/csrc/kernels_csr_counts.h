// CSR, count-valued, small groups: the whole test in ONE pass over the CSR arrays, group-major.
//
// A CSR row is a cell, and a cell has ONE group code: what costs the CSC kernel its time -- a codes[row] gather per stored entry,
// and a finishing kernel that transposes [gene][group] statistics into [group][gene] planes -- does not exist in this orientation.
// A workgroup takes (group g, window of Wg genes): it reads the window's stretch of each of the group's ~150 rows (rows found
// through `perm`, the stretch through per-row window boundaries: both ends of a stretch are known, every request is issued at
// once, a stretch is a contiguous run of ~200 entries), builds the group's value histogram of every gene of the window in LDS --
// h[word][gene], the mixed 8- / 4-bit cells of k_csc_counts, 36 bytes per gene, plain LDS atomics, no branch per entry -- and then sweeps
// the window with one thread per gene: the statistics of (g, gene) against a per-gene table of the reference group (OVO) or of
// the whole column (OVR) read from L2, the p-value and the fold change, written straight into the planes as 512-byte row pieces.
//   OVO:  S2 = sum_{c>=1} tB[c] (2 zA + 2 cumA[c] + tA[c]) + zB zA,         tie = T_A + sum_{c>=1} tB (3 tA (tA+tB) + tB^2 - 1) + (t0^3 - t0)
//   OVR:  r2 = sum_{c>=1} tB[c] (2 n0 + 2 cum[c] + t[c] + 1) + zB (n0 + 1), tie = sum_{c>=1} (t^3 - t) + (n0^3 - n0)
// (kernels_csc_counts.h; sparse_ovo.py:58-85, sparse_ovr.py:70-83).  HBM traffic: the CSR arrays once (OVR: twice, the column
// histograms need a pass of their own) + the planes once.
//
// The tables come from a pre-pass with the same entry loop (k_csr_hist: chunks of <= 255 rows x gene windows, 8-bit cells that
// cannot overflow, non-zero cells added to hist[value][gene] with lane-contiguous global atomics -- the reference group's rows for
// OVO, every row for OVR) and k_csr_tables (one thread per gene).
//
// A gene is taken only if every stored value of it is an integer in [1, 64): a value outside sets gene_flags[gene] = 1, a 4-bit
// cell that would overflow (16 cells of one group with one value >= 8) sets 2; the host recomputes flagged genes by the general
// sparse routes.  Host-checked: rows with sorted column indices (the reference's contract for CSR, tests/test_asymptotic_wilcoxon.py:
// 259-273), ranked groups of at most 255 cells, an OVO reference below 30 000 cells (15-bit multiplicities, 32-bit tie terms).
// Replaces csr_get_contig_cols_into_csc + the per-group CSR -> CSC slicing + the merge of sparse_ovo.py:214-260 / sparse_ovr.py:158-208.
#pragma once
#include "common.h"
#include "kernels_finalize.h"

#define CSRC_NT 512   // main kernel: two workgroups per CU (72 KB of cells each at 2048 genes)
#define CSRC_RT 64
#define CSRC_WPG 9    // words per gene of the mixed layout: words 0, 1 = 8-bit cells of the values 0 .. 7, words 2 .. 8 = eight 4-bit cells each
#define CSRC_U 4      // 64-entry chunks of a row requested together
#define CSRH_NT 1024  // pre-pass: one workgroup per CU (128 KB of 8-bit cells at 2048 genes)
#define CSRH_WPG 16
#define CSRH_ROWS 255
#define CSRC_MAX_BIG 16 // groups of more than 255 cells the route takes (a 2-MB histogram slab each at 8000 genes)

struct CsrCountsParams {
    const void *data, *indices, *indptr; // CSR arrays (device)
    const int *perm;                     // [n_cells] rows in group order
    const int *pos_ptr;                  // [G + 1]
    const int *counts;                   // [G]
    int G, ref;                          // ref == -1: OVR
    long long n_cells;
    long long col_lb;                    // first column of the call's window
    int W;                               // columns of the call's window
    int Wg;                              // genes per workgroup (a multiple of 64)
    const u32 *bounds;                   // [n_bnd][n_cells] row-relative offset of the first entry at or beyond column col_lb + b * Wf
    int bstep;                           // fine windows (Wf genes) per workgroup window
    int n_bnd;                           // boundaries (fine windows + 1)
    const u32 *tab;                      // [64][Wpad]: OVO (A << 15) | tA,  A = 2 zA + 2 cumA[c] + tA[c];  OVR A = 2 n0 + 2 cum[c] + t[c] + 1
    const uint4 *ginfo;                  // [Wpad]: OVO {T_A lo, T_A hi, zA, value sum of the reference}; OVR {T lo, T hi, n0, -}, T = the bits of the float64 tie sum
    const double *gene_total;            // [Wpad] OVR: the column's value sum; OVO: the reference group's mean (mu_ref)
    long long Wpad;
    u32 *gene_flags;                     // [W]
    const u32 *verdict;                  // {non-integer samples, samples beyond the table, samples, rows out of order}: decided on the device
    u32 *unsorted;                       // = verdict + 3: the entry loops set it when they meet an entry outside its window
    int use_continuity, tie_correct, alternative;
    double *out_p, *out_u, *out_fc;      // [G][out_ld], the call's column offset applied
    long long out_ld;
    // pre-pass (k_csr_hist): chunks of <= 255 rows, each added to one histogram slab
    const int *chunk_p0;                 // [chunks] first position in `perm` of the chunk's rows; -1 - r: the rows r, r + 1, ... themselves
    const int *chunk_n;                  // [chunks] rows
    const int *chunk_slab;               // [chunks] slab 0: the reference group (OVO) / every row (OVR); slab 1 + k: big group k
    u32 *hist;                           // [slabs][64][Wpad]
    u32 *dump;                           // OVR: [G][windows][9][Wg] the cells of every (group, window), as the count pass left them in LDS
    // groups of more than 255 cells (k_csr_big_sweep)
    const int *big_groups;               // [n_big] group numbers
    int n_big;
    int abl;                             // timing experiments only (tools/ab.py csr_counts_abl): 1 no entry loop, 2 no sweep, 4 no LDS atomics, 8 no p-values
};

// {non-integer samples, samples beyond the table, samples, rows out of order}: more than 2 % / 0.5 % of the samples, or any row out of
// order: not for this route (flagged genes are recomputed gene run by gene run: nearly all of them must fit)
__device__ __forceinline__ bool csrc_verdict_bad(const u32 *verdict) {
    return verdict && ((double)verdict[0] > 0.02 * (double)verdict[2] || (double)verdict[1] > 0.005 * (double)verdict[2] || verdict[3] != 0u);
}

// a matrix with more than `max_density` of its cells stored is left to the dense byte windows (the group-major pass costs per stored
// entry; 4-bit cells overflow in many genes): verdict[3] |= 2.  One thread.
template <typename IdxT>
__global__ void k_csr_density_verdict(const IdxT *__restrict__ indptr, long long n_rows, long long n_cols, double max_density, u32 *__restrict__ verdict) {
    const double nnz = (double)((long long)indptr[n_rows] - (long long)indptr[0]);
    if (nnz > max_density * (double)n_rows * (double)n_cols) atomicOr(verdict + 3, 2u);
}

// Row-relative offsets of the window boundaries of every row: bounds[b][r] = number of entries of row r with a column below
// col_lb + b * Wf (the last boundary: col_ub).  One thread per (row, boundary): a guess from the column's place in the row (stored
// columns are spread about evenly), a gallop in steps of 16, 32, ..., then a binary search inside the bracket -- two or three cache
// lines per boundary where a plain binary search over a row of 800 entries touches six.
template <typename IdxT>
__global__ __launch_bounds__(256) void k_csr_row_bounds(const IdxT *__restrict__ indices, const IdxT *__restrict__ indptr, int n_rows, long long n_cols,
                                                       long long col_lb, long long col_ub, int Wf, int n_bnd, u32 *__restrict__ bounds,
                                                       const u32 *__restrict__ verdict = nullptr) {
    if (csrc_verdict_bad(verdict)) return; // (not a matrix for the route: 0.09 ms of searches at C3 shape that nobody would read)
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (long long)n_rows * n_bnd) return;
    const int b = (int)(i / n_rows), r = (int)(i - (long long)b * n_rows);
    long long c = col_lb + (long long)b * Wf;
    if (c > col_ub) c = col_ub;
    const long long s = (long long)indptr[r], e = (long long)indptr[r + 1];
    const u32 len = (u32)(e - s);
    u32 res;
    if (c <= 0 || len == 0) res = 0;
    else if (c >= n_cols) res = len;
    else {
        const IdxT *row = indices + s;
        u32 p = (u32)((double)c / (double)n_cols * (double)len);
        if (p >= len) p = len - 1;
        u32 lo, hi; // every entry below lo has a column < c, every entry from hi on a column >= c
        if ((long long)row[p] < c) {
            lo = p + 1; hi = len;
            u32 step = 16;
            while (lo + step <= len) {
                if ((long long)row[lo + step - 1] < c) { lo += step; step <<= 1; }
                else { hi = lo + step - 1; break; }
            }
        } else {
            hi = p; lo = 0;
            u32 step = 16;
            while (hi >= step) {
                if ((long long)row[hi - step] >= c) { hi -= step; step <<= 1; }
                else { lo = hi - step + 1; break; }
            }
        }
        while (lo < hi) { const u32 m = (lo + hi) >> 1; if ((long long)row[m] < c) lo = m + 1; else hi = m; }
        res = lo;
    }
    bounds[i] = res;
}

// a stored value as a table index: 1 .. 63, 0 for a stored zero (it is a zero), -1 when it is no count the tables hold
__device__ __forceinline__ int csrc_code(float v) {
    const int c = (int)v; // (saturating; NaN -> 0)
    return v == 0.0f ? 0 : (((float)c == v && (u32)(c - 1) < (u32)(CSRC_RT - 1)) ? c : -1);
}
__device__ __forceinline__ int csrc_code(double v) {
    const int c = (int)v;
    return v == 0.0 ? 0 : (((double)c == v && (u32)(c - 1) < (u32)(CSRC_RT - 1)) ? c : -1);
}
__device__ __forceinline__ int csrc_code(int32_t v) { return v == 0 ? 0 : ((u32)(v - 1) < (u32)(CSRC_RT - 1) ? v : -1); }
__device__ __forceinline__ int csrc_code(int64_t v) { return v == 0 ? 0 : ((u64)(v - 1) < (u64)(CSRC_RT - 1) ? (int)v : -1); }

__device__ __forceinline__ long long csrc_uniform(long long x) { // a wavefront-uniform 64-bit value into scalar registers
    const u32 lo = (u32)__builtin_amdgcn_readfirstlane((int)(u32)x), hi = (u32)__builtin_amdgcn_readfirstlane((int)(u32)((u64)x >> 32));
    return (long long)(((u64)hi << 32) | lo);
}

// The entry loop both kernels share.  Wavefront `wave` of NW takes the rows wave, wave + NW, ... of the workgroup's row list (LDS:
// row_k = first entry of the row's stretch, row_n = its length); a row's stretch is requested as CSRC_U chunks of 64 entries off a
// SCALAR base (one lane offset per request; a lane beyond the stretch asks for its last entry again and counts nothing), the next
// row's while this row's entries go into the tables -- no branch per entry, no LDS round trip: every table update is a plain LDS
// atomic.  MIXED: the 8- / 4-bit cells; an entry with a value of 8 or more is also counted in byte 0 of the gene's word 0 (the cell of
// the value 0, which no entry has), so that the sweep can tell a 4-bit cell that overflowed: a carry only loses entries, the gene's
// 4-bit cells then add up to fewer than were counted.  Else 8-bit cells for every value.  An entry outside the window it was found in
// means the row's columns are not in order: *unsorted.
template <typename InT, typename IdxT, bool MIXED, int U = CSRC_U>
__device__ __forceinline__ void csrc_entries(const InT *__restrict__ data, const IdxT *__restrict__ indices, int n_rows_wg, const long long *row_k,
                                             const u32 *row_n, int wave, int NW, int lane, long long cbase, int wcols, int Wg, u32 *h,
                                             u32 *__restrict__ gene_flags /* of the window's first gene */, u32 *__restrict__ unsorted) {
    IdxT ni[U];
    InT nv[U];
    long long nk = 0;
    u32 nn = 0;
    auto load = [&](long long k0, u32 j0, u32 n, IdxT *di, InT *dv) { // (n > j0)
        const IdxT *ip = indices + k0 + j0; // (uniform)
        const InT *vp = data + k0 + j0;
        const u32 left = n - j0;
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const u32 e = (u32)(u * 64 + lane), ec = min(e, left - 1u);
            di[u] = ip[ec];
            const InT v = vp[ec];
            dv[u] = e < left ? v : (InT)0; // (a lane without an entry holds a zero: nothing is counted for it)
        }
    };
    auto put = [&](const IdxT *ci, const InT *cv) {
        bool rare = false;
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int c = csrc_code(cv[u]);
            const long long col64 = (long long)ci[u] - cbase;
            const int col = (int)col64;
            const bool inr = sizeof(IdxT) == 4 ? (u32)col < (u32)wcols : (unsigned long long)col64 < (unsigned long long)wcols;
            rare |= c != 0 && (!inr || c < 0);
            if (c > 0 && inr) {
                if (MIXED) {
                    const u32 q = (u32)c + min((u32)c, 8u); // nibble number inside the gene's words: bytes for 1 .. 7, nibbles from 8 on
                    atomicAdd(&h[__umul24(q >> 3, (u32)Wg) + (u32)col], 1u << ((q & 7u) * 4u));
                    if (c >= 8) atomicAdd(&h[col], 1u);
                } else atomicAdd(&h[__umul24((u32)c >> 2, (u32)Wg) + (u32)col], 1u << (((u32)c & 3u) * 8u));
            }
        }
        if (rare) { // a value the tables do not hold, or an entry that is not where the row's order puts it
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int c = csrc_code(cv[u]);
                const long long col64 = (long long)ci[u] - cbase;
                if (c != 0) {
                    if ((unsigned long long)col64 >= (unsigned long long)wcols) *unsorted = 1u;
                    else if (c < 0) gene_flags[col64] = 1u;
                }
            }
        }
    };
    auto zero = [&](IdxT *di, InT *dv) {
#pragma unroll
        for (int u = 0; u < U; ++u) { di[u] = (IdxT)0; dv[u] = (InT)0; }
    };
    int row = wave;
    if (row < n_rows_wg) {
        nk = csrc_uniform(row_k[row]); nn = (u32)__builtin_amdgcn_readfirstlane((int)row_n[row]);
        if (nn) load(nk, 0, nn, ni, nv); else zero(ni, nv);
    }
    for (; row < n_rows_wg; row += NW) {
        IdxT ci[U];
        InT cv[U];
#pragma unroll
        for (int u = 0; u < U; ++u) { ci[u] = ni[u]; cv[u] = nv[u]; }
        const long long ck = nk;
        const u32 cn = nn;
        const int nrow = row + NW;
        if (nrow < n_rows_wg) {
            nk = csrc_uniform(row_k[nrow]); nn = (u32)__builtin_amdgcn_readfirstlane((int)row_n[nrow]);
            if (nn) load(nk, 0, nn, ni, nv); else zero(ni, nv);
        }
        put(ci, cv);
        for (u32 j0 = 64 * U; j0 < cn; j0 += 64 * U) { // (a stretch of more than 64 U entries: the rest, round by round)
            load(ck, j0, cn, ci, cv);
            put(ci, cv);
        }
    }
}

// The same loop for NARROW column windows (the reference's driver asks for 256 genes at a time: a row's stretch is a few dozen entries).
// One 64-entry chunk per row, R rows of the wavefront's list in flight at once (and the next R requested meanwhile) instead of R chunks
// of one row: the same number of requests in flight, no request wasted on entries that are not there.
template <typename InT, typename IdxT, bool MIXED, int R = CSRC_U>
__device__ __forceinline__ void csrc_entries_narrow(const InT *__restrict__ data, const IdxT *__restrict__ indices, int n_rows_wg, const long long *row_k,
                                                    const u32 *row_n, int wave, int NW, int lane, long long cbase, int wcols, int Wg, u32 *h,
                                                    u32 *__restrict__ gene_flags, u32 *__restrict__ unsorted) {
    IdxT ni[R];
    InT nv[R];
    long long nk[R];
    u32 nn[R];
    auto load1 = [&](long long k0, u32 j0, u32 n, IdxT &di, InT &dv) { // one chunk of a stretch; n > j0
        const u32 left = n - j0, ec = min((u32)lane, left - 1u);
        di = (indices + k0 + j0)[ec];
        const InT v = (data + k0 + j0)[ec];
        dv = (u32)lane < left ? v : (InT)0;
    };
    auto put1 = [&](IdxT ci, InT cv) {
        const int c = csrc_code(cv);
        const long long col64 = (long long)ci - cbase;
        const int col = (int)col64;
        const bool inr = sizeof(IdxT) == 4 ? (u32)col < (u32)wcols : (unsigned long long)col64 < (unsigned long long)wcols;
        if (c > 0 && inr) {
            if (MIXED) {
                const u32 q = (u32)c + min((u32)c, 8u);
                atomicAdd(&h[__umul24(q >> 3, (u32)Wg) + (u32)col], 1u << ((q & 7u) * 4u));
                if (c >= 8) atomicAdd(&h[col], 1u);
            } else atomicAdd(&h[__umul24((u32)c >> 2, (u32)Wg) + (u32)col], 1u << (((u32)c & 3u) * 8u));
        } else if (c != 0) {
            if (!inr) *unsorted = 1u;
            else gene_flags[col64] = 1u;
        }
    };
    auto fetch = [&](int row0) { // the stretches of rows row0, row0 + NW, ... (R of them)
#pragma unroll
        for (int u = 0; u < R; ++u) {
            const int r = row0 + u * NW;
            nk[u] = 0; nn[u] = 0; ni[u] = (IdxT)0; nv[u] = (InT)0;
            if (r < n_rows_wg) { // (uniform)
                nk[u] = csrc_uniform(row_k[r]);
                nn[u] = (u32)__builtin_amdgcn_readfirstlane((int)row_n[r]);
                if (nn[u]) load1(nk[u], 0, nn[u], ni[u], nv[u]);
            }
        }
    };
    fetch(wave);
    for (int row0 = wave; row0 < n_rows_wg; row0 += NW * R) {
        IdxT ci[R];
        InT cv[R];
        long long ck[R];
        u32 cn[R];
#pragma unroll
        for (int u = 0; u < R; ++u) { ci[u] = ni[u]; cv[u] = nv[u]; ck[u] = nk[u]; cn[u] = nn[u]; }
        if (row0 + NW * R < n_rows_wg) fetch(row0 + NW * R);
#pragma unroll
        for (int u = 0; u < R; ++u) put1(ci[u], cv[u]);
#pragma unroll
        for (int u = 0; u < R; ++u)
            for (u32 j0 = 64; j0 < cn[u]; j0 += 64) { // (a stretch of more than 64 entries: rare here)
                IdxT xi; InT xv;
                load1(ck[u], j0, cn[u], xi, xv);
                put1(xi, xv);
            }
    }
}

// the workgroup's row list into LDS: first entry and length of the stretch [boundary b0, boundary b1) of each row
template <typename IdxT>
__device__ __forceinline__ void csrc_row_list(const CsrCountsParams &P, int p0, int n_rows_wg, int b0, int b1, int tid, int nt, long long *row_k, u32 *row_n) {
    const IdxT *indptr = (const IdxT *)P.indptr;
    for (int i = tid; i < n_rows_wg; i += nt) {
        const int r = p0 >= 0 ? P.perm[p0 + i] : (-1 - p0) + i;
        const u32 a = P.bounds[(size_t)b0 * P.n_cells + r], b = P.bounds[(size_t)b1 * P.n_cells + r];
        row_k[i] = (long long)indptr[r] + a;
        row_n[i] = b > a ? b - a : 0u;
    }
}

// ---- pre-pass: hist[c][gene] += the 8-bit cells of (chunk of <= 255 rows, window of Wg genes) ------------------------------------
// grid (windows, chunks)
template <typename InT, typename IdxT>
__global__ __launch_bounds__(CSRH_NT) void k_csr_hist(CsrCountsParams P) {
    extern __shared__ __align__(16) u32 csrh_h[]; // [16][Wg]
    __shared__ long long row_k[CSRH_ROWS + 1];
    __shared__ u32 row_n[CSRH_ROWS + 1];
    if (csrc_verdict_bad(P.verdict)) return;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int Wg = P.Wg, w = blockIdx.x;
    const int p0 = P.chunk_p0[blockIdx.y], n_rows_wg = P.chunk_n[blockIdx.y];
    u32 *hist = P.hist + (size_t)P.chunk_slab[blockIdx.y] * CSRC_RT * P.Wpad;
    const int wcols = min(Wg, P.W - w * Wg);
    const int b0 = w * P.bstep, b1 = min(b0 + P.bstep, P.n_bnd - 1);
    csrc_row_list<IdxT>(P, p0, n_rows_wg, b0, b1, tid, CSRH_NT, row_k, row_n);
    {
        uint4 *h4 = (uint4 *)csrh_h;
        for (int i = tid; i < (Wg * CSRH_WPG) >> 2; i += CSRH_NT) h4[i] = make_uint4(0u, 0u, 0u, 0u);
    }
    __syncthreads();
    csrc_entries<InT, IdxT, false>((const InT *)P.data, (const IdxT *)P.indices, n_rows_wg, row_k, row_n, wave, CSRH_NT / 64, lane,
                                   P.col_lb + (long long)w * Wg, wcols, Wg, csrh_h, P.gene_flags + (size_t)w * Wg, P.unsorted);
    __syncthreads();
    for (int j = tid; j < wcols; j += CSRH_NT) {
        u32 *dst = hist + (size_t)w * Wg + j;
#pragma unroll 1
        for (int i = 0; i < CSRH_WPG; ++i) {
            const u32 wd = csrh_h[i * Wg + j];
            if (__ballot(wd != 0u) == 0ull) continue;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const u32 t = (wd >> (k * 8)) & 0xFFu;
                if (t) atomicAdd(dst + (size_t)(i * 4 + k) * P.Wpad, t);
            }
        }
    }
}

// ---- tables: one thread per gene ----------------------------------------------------------------------------------------------
template <bool OVR>
__global__ __launch_bounds__(256) void k_csr_tables(const u32 *__restrict__ hist, long long Wpad, int W, long long n_sel /* reference cells (OVO) / all cells (OVR) */,
                                                   int n_add /* OVR: slabs 1 .. n_add (the big groups) belong to the column as well */,
                                                   u32 *__restrict__ tab, uint4 *__restrict__ ginfo, double *__restrict__ gene_total,
                                                   const u32 *__restrict__ verdict = nullptr) {
    if (csrc_verdict_bad(verdict)) return;
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= W) return;
    auto cell = [&](int c) {
        u64 t = hist[(size_t)c * Wpad + j];
        for (int k = 1; k <= n_add; ++k) t += hist[((size_t)k * CSRC_RT + c) * Wpad + j];
        return t;
    };
    u64 nnz = 0;
    for (int c = 1; c < CSRC_RT; ++c) nnz += cell(c);
    const u64 z = (u64)n_sel - nnz; // zA / n0
    u64 cum = 0, T = 0, sum = 0;
    for (int c = 1; c < CSRC_RT; ++c) {
        const u64 t = cell(c);
        const u64 A = 2ull * z + 2ull * cum + t + (OVR ? 1ull : 0ull);
        tab[(size_t)c * Wpad + j] = OVR ? (u32)A : (u32)((A << 15) | t);
        cum += t;
        T += t * t * t - t;
        sum += t * (u64)c;
    }
    if (OVR) {
        T = tie_f64_sparse(T, (long long)z); // the BITS of the float64 tie sum, as the reference's sparse OVR path forms it (kernels_finalize.h)
        gene_total[j] = (double)sum; // integer sums: exact whatever the order of addition
        ginfo[j] = make_uint4((u32)T, (u32)(T >> 32), (u32)z, 0u);
    } else {
        ginfo[j] = make_uint4((u32)T, (u32)(T >> 32), (u32)z, (u32)sum);
        gene_total[j] = (double)sum / (double)n_sel; // mu_ref of fold_change_from_summed_expr (math.py:183), the same for every group
    }
}

// ---- main pass: grid (windows, groups) ----------------------------------------------------------------------------------------
static inline size_t csrc_lds_bytes(int Wg) { return (size_t)Wg * CSRC_WPG * 4; }
static inline size_t csrh_lds_bytes(int Wg) { return (size_t)Wg * CSRH_WPG * 4; }

// The sweep of one (group, gene window): one thread per gene.  word(i, j, live) = word i of gene j's cells -- from LDS behind the entry
// loop (OVO), or from the dump of the OVR count pass.
template <bool OVR, bool PRELOAD = false, typename WordFn>
__device__ __forceinline__ void csrc_sweep(const CsrCountsParams &P, int g, int n_g, int w, int wcols, bool is_ref, int tid, WordFn word) {
    const int Wg = P.Wg;
    const double cc = P.use_continuity ? 0.5 : 0.0;
    const long long n_tgt = n_g;
    const long long n_refc = OVR ? 0 : (long long)P.counts[OVR ? 0 : P.ref];
    const long long n_ref = OVR ? (P.n_cells - n_tgt) : n_refc;
    const long long n = OVR ? P.n_cells : (n_ref + n_tgt);
    // what compute_pval forms from the group's sizes alone, once per workgroup (pval_device_pre)
    const GroupConst gc = group_const(n_ref, n_tgt, n);
    const double mu = gc.mu, n12 = gc.n12, nnn = gc.nnn, var0 = gc.var0;
    const double inf = __longlong_as_double(0x7FF0000000000000ll);
    for (int j0 = 0; j0 < wcols; j0 += CSRC_NT) { // (uniform trip count: the word skips below are wavefront-wide votes)
        const int j = j0 + tid;
        const bool live = j < wcols;
        const u32 jc = (u32)(w * Wg + (live ? j : 0)); // the gene's place in the call's window: the lane offset of every table read
        const uint4 gi = P.ginfo[jc];
        const double gt = P.gene_total[jc]; // OVO: the reference group's mean; OVR: the column's value sum
        const u64 T_sel = (u64)gi.x | ((u64)gi.y << 32), zsel = gi.z;
        double p, U, fc;
        if (is_ref) { // sparse_ovo.py:140-143
            p = 1.0; U = -1.0;
            fc = (gt == 0.0) ? inf : gt / gt;
        } else {
            u64 acc64 = 0, tie = 0;
            u32 acc32 = 0, nnz_g = 0, vsum = 0;
            u32 pre[CSRC_WPG]; // (PRELOAD: the words come from HBM -- all nine requested before the first is looked at)
            if (PRELOAD) {
#pragma unroll
                for (int i = 0; i < CSRC_WPG; ++i) pre[i] = word(i, j, live);
            }
            auto cell = [&](int c, u32 tB, u32 t) {
                nnz_g += tB;
                vsum = __umul24(tB, (u32)c) + vsum;
                if (OVR) acc64 += (u64)tB * t;
                else { // (n_ref < 30 000, tB <= 255: every 32-bit term holds)
                    const u32 A = t >> 15, tS = t & 0x7FFFu;
                    acc32 += tB * A;
                    tie += (u64)tB * (u32)(3u * (tS * tS) + tB * (3u * tS + tB) - 1u); // tB (3 tS (tS + tB) + tB^2 - 1)
                }
            };
            u32 cnt8 = 0; // entries of the gene with a value of 8 or more, as the entry loop counted them (byte 0 of word 0)
#pragma unroll
            for (int i = 0; i < 2; ++i) { // the 8-bit cells of the values 1 .. 7
                const u32 wd = PRELOAD ? pre[i] : word(i, j, live);
                if (__ballot(wd != 0u) == 0ull) continue;
                if (i == 0) cnt8 = wd & 0xFFu;
                u32 t[4];
#pragma unroll
                for (int k = 0; k < 4; ++k) t[k] = (i * 4 + k) ? (P.tab + (size_t)(i * 4 + k) * P.Wpad)[jc] : 0u; // (scalar row base + lane offset)
#pragma unroll
                for (int k = 0; k < 4; ++k) if (i * 4 + k) cell(i * 4 + k, (wd >> (k * 8)) & 0xFFu, t[k]);
            }
            const u32 nnz_low = nnz_g;
#pragma unroll 1
            for (int i = 2; i < CSRC_WPG; ++i) { // eight 4-bit cells per word
                u32 wd;
                if (PRELOAD) { // (a register array indexed by the loop counter: selects, not scratch)
                    wd = pre[2];
#pragma unroll
                    for (int q = 3; q < CSRC_WPG; ++q) wd = i == q ? pre[q] : wd;
                } else wd = word(i, j, live);
                if (__ballot(wd != 0u) == 0ull) continue;
                const u32 *row = P.tab + (size_t)(8 + (i - 2) * 8) * P.Wpad; // (uniform)
                u32 t[8];
#pragma unroll
                for (int k = 0; k < 8; ++k) t[k] = (row + (size_t)k * P.Wpad)[jc];
#pragma unroll
                for (int k = 0; k < 8; ++k) cell(8 + (i - 2) * 8 + k, (wd >> (k * 4)) & 0xFu, t[k]);
            }
            if (live && nnz_g - nnz_low != cnt8) P.gene_flags[jc] = 2u; // a 4-bit cell overflowed: the gene is recomputed elsewhere
            const u64 zB = (u64)(n_tgt - (long long)nnz_g);
            long long two_u;
            u64 tie_sum;
            if (OVR) {
                const u64 acc = acc64 + zB * (zsel + 1ull);
                two_u = 2ll * (P.n_cells - n_tgt) * n_tgt + n_tgt * (n_tgt + 1) - (long long)acc;
                tie_sum = T_sel;
            } else {
                const u64 acc = (u64)acc32 + zB * zsel;
                const u64 t0 = zsel + zB;
                two_u = 2ll * n_refc * n_tgt - (long long)acc;
                tie_sum = T_sel + tie + (t0 * t0 * t0 - t0);
            }
            U = 0.5 * (double)two_u;
            const double tie_d = !P.tie_correct ? 0.0 : (OVR ? __longlong_as_double((long long)tie_sum) : (double)tie_sum);
            p = (P.abl & 8) ? tie_d : pval_device_pre(nnn, var0, n12, tie_d, U, mu, cc, P.alternative);
            // fold change, math.py:181-192 (integer value sums: exact)
            const double sum_g = (double)vsum;
            if (OVR) fc = fold_change_device(sum_g, gt - sum_g, gc);
            else { // (gt: the reference group's mean, formed once per gene by k_csr_tables)
                const double mu_tgt = sum_g / gc.d_tgt;
                fc = (gt == 0.0) ? inf : mu_tgt / gt;
            }
        }
        if (live) {
            const size_t o = (size_t)g * P.out_ld + jc;
            P.out_p[o] = p;
            P.out_u[o] = U;
            P.out_fc[o] = fc;
        }
    }
}

// OVO: entries + sweep in one kernel.  OVR: the ranks need the histogram of the whole column, which is complete only when every group's
// entries have been counted -- so the OVR count pass (this kernel) leaves each (group, window)'s cells in HBM (dump: a straight copy of
// the LDS image, 36 bytes per test); k_csr_colhist adds them up over the groups into the column histograms (hist slab 0),
// k_csr_tables forms the rank tables and k_csr_ovr_sweep ranks every group from the dump.  The CSR arrays are read ONCE.  (Adding the
// non-zero cells to the column histograms from here, with global atomics -- 1.3e8 of them at C3 --, cost 0.5 ms; the dump is read
// twice instead: 0.58 GB each time.)
template <typename InT, typename IdxT, bool OVR>
__global__ __launch_bounds__(CSRC_NT, 4) void k_csr_counts(CsrCountsParams P) {
    extern __shared__ __align__(16) u32 csrc_h[]; // [9][Wg]
    __shared__ long long row_k[256];
    __shared__ u32 row_n[256];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int Wg = P.Wg, w = blockIdx.x, g = blockIdx.y;
    const int wcols = min(Wg, P.W - w * Wg);
    if (csrc_verdict_bad(P.verdict)) { // uniform: not a count matrix (or rows out of order): every gene is left to the general routes
        if (g == 0) for (int j = tid; j < wcols; j += CSRC_NT) P.gene_flags[(size_t)w * Wg + j] = 1u;
        return;
    }
    const int p0 = P.pos_ptr[g], n_g = P.pos_ptr[g + 1] - p0;
    const bool is_ref = !OVR && g == P.ref;
    if (n_g > 255 && !is_ref) return; // (uniform) a big group: its histograms are built chunk by chunk (k_csr_hist), k_csr_big_sweep ranks it
    if (!is_ref) {
        const int b0 = w * P.bstep, b1 = min(b0 + P.bstep, P.n_bnd - 1);
        csrc_row_list<IdxT>(P, p0, n_g, b0, b1, tid, CSRC_NT, row_k, row_n);
        uint4 *h4 = (uint4 *)csrc_h;
        for (int i = tid; i < (Wg * CSRC_WPG) >> 2; i += CSRC_NT) h4[i] = make_uint4(0u, 0u, 0u, 0u);
        __syncthreads();
        if (P.abl & 1) {}
        else if (Wg > 640) // (uniform) a row's stretch of ~2000 genes: four chunks of 64 entries requested together
            csrc_entries<InT, IdxT, true, CSRC_U>((const InT *)P.data, (const IdxT *)P.indices, n_g, row_k, row_n, wave, CSRC_NT / 64, lane,
                                                  P.col_lb + (long long)w * Wg, wcols, Wg, csrc_h, P.gene_flags + (size_t)w * Wg, P.unsorted);
        else // a narrow column window (the reference's driver asks for 256 genes at a time): stretches of a few dozen entries, one chunk
            csrc_entries_narrow<InT, IdxT, true>((const InT *)P.data, (const IdxT *)P.indices, n_g, row_k, row_n, wave, CSRC_NT / 64, lane,
                                                 P.col_lb + (long long)w * Wg, wcols, Wg, csrc_h, P.gene_flags + (size_t)w * Wg, P.unsorted);
        __syncthreads();
    }
    if (P.abl & 2) return;
    if (OVR) {
        // the cells as they lie, 16 bytes per lane and store
        uint4 *dst = (uint4 *)(P.dump + ((size_t)g * gridDim.x + w) * CSRC_WPG * Wg);
        const uint4 *h4 = (const uint4 *)csrc_h;
        for (int i = tid; i < (Wg * CSRC_WPG) >> 2; i += CSRC_NT) dst[i] = h4[i];
        return;
    }
    csrc_sweep<OVR>(P, g, n_g, w, wcols, is_ref, tid, [&](int i, int j, bool live) { return live ? csrc_h[i * Wg + j] : 0u; });
}

// OVR, second pass: every (group, window) from its dumped cells.  grid (windows, groups)
static __global__ __launch_bounds__(CSRC_NT) void k_csr_ovr_sweep(CsrCountsParams P) {
    if (csrc_verdict_bad(P.verdict)) return;
    const int Wg = P.Wg, w = blockIdx.x, g = blockIdx.y;
    const int wcols = min(Wg, P.W - w * Wg);
    const int n_g = P.pos_ptr[g + 1] - P.pos_ptr[g];
    if (n_g > 255) return; // (k_csr_big_sweep)
    const u32 *src = P.dump + ((size_t)g * gridDim.x + w) * CSRC_WPG * Wg;
    csrc_sweep<true, true>(P, g, n_g, w, wcols, false, threadIdx.x, [&](int i, int j, bool live) { return live ? src[i * Wg + j] : 0u; });
}

// OVR: the column histograms from the dump -- one thread per (gene, slice of the groups): the cells of its groups added up in registers,
// then added to hist slab 0 (63 lane-contiguous atomics per thread).  grid (gene blocks of 256, slices)
static __global__ __launch_bounds__(256) void k_csr_colhist(CsrCountsParams P, int n_win, int groups_per_slice) {
    if (csrc_verdict_bad(P.verdict)) return;
    const int jc = blockIdx.x * blockDim.x + threadIdx.x;
    if (jc >= P.W) return;
    const int w = jc / P.Wg, j = jc - w * P.Wg;
    const int g0 = blockIdx.y * groups_per_slice, g1 = min(P.G, g0 + groups_per_slice);
    u32 cnt[CSRC_RT];
#pragma unroll
    for (int c = 0; c < CSRC_RT; ++c) cnt[c] = 0;
    for (int g = g0; g < g1; ++g) {
        if (P.pos_ptr[g + 1] - P.pos_ptr[g] > 255) continue; // (uniform: a big group has a slab of its own)
        const u32 *src = P.dump + ((size_t)g * n_win + w) * CSRC_WPG * P.Wg + j;
        u32 wd[CSRC_WPG];
#pragma unroll
        for (int i = 0; i < CSRC_WPG; ++i) wd[i] = src[(size_t)i * P.Wg];
#pragma unroll
        for (int c = 1; c < 8; ++c) cnt[c] += (wd[c >> 2] >> ((c & 3) * 8)) & 0xFFu;
#pragma unroll
        for (int c = 8; c < CSRC_RT; ++c) cnt[c] += (wd[1 + (c >> 3)] >> ((c & 7) * 4)) & 0xFu;
    }
#pragma unroll
    for (int c = 1; c < CSRC_RT; ++c)
        if (cnt[c]) atomicAdd(P.hist + (size_t)c * P.Wpad + jc, cnt[c]);
}

// ---- groups of more than 255 cells: their histograms were added up chunk by chunk (k_csr_hist, slab 1 + k); one thread per (big
// group, gene) turns them into the statistics with 64-bit terms.  grid (gene blocks of 256, big groups)
template <bool OVR>
__global__ __launch_bounds__(256) void k_csr_big_sweep(CsrCountsParams P) {
    if (csrc_verdict_bad(P.verdict)) return;
    const int jc = blockIdx.x * blockDim.x + threadIdx.x;
    if (jc >= P.W) return;
    const int g = P.big_groups[blockIdx.y];
    const u32 *hb = P.hist + (size_t)(1 + blockIdx.y) * CSRC_RT * P.Wpad + jc;
    const u32 *tb = P.tab + jc;
    const uint4 gi = P.ginfo[jc];
    const u64 T_sel = (u64)gi.x | ((u64)gi.y << 32), zsel = gi.z;
    const long long n_tgt = P.counts[g];
    const long long n_refc = OVR ? 0 : (long long)P.counts[OVR ? 0 : P.ref];
    const long long n_ref = OVR ? (P.n_cells - n_tgt) : n_refc;
    const long long n = OVR ? P.n_cells : (n_ref + n_tgt);
    u64 acc = 0, tie = 0, nnz_g = 0, vsum = 0;
#pragma unroll 4
    for (int c = 1; c < CSRC_RT; ++c) {
        const u64 tB = hb[(size_t)c * P.Wpad];
        const u32 t = tb[(size_t)c * P.Wpad];
        nnz_g += tB;
        vsum += tB * (u64)c;
        if (OVR) acc += tB * (u64)t;
        else {
            const u64 A = t >> 15, tS = t & 0x7FFFu;
            acc += tB * A;
            tie += tB * (3ull * tS * (tS + tB) + tB * tB - 1ull);
        }
    }
    const u64 zB = (u64)n_tgt - nnz_g;
    long long two_u;
    u64 tie_sum;
    if (OVR) {
        acc += zB * (zsel + 1ull);
        two_u = 2ll * (P.n_cells - n_tgt) * n_tgt + n_tgt * (n_tgt + 1) - (long long)acc;
        tie_sum = T_sel;
    } else {
        acc += zB * zsel;
        const u64 t0 = zsel + zB;
        two_u = 2ll * n_refc * n_tgt - (long long)acc;
        tie_sum = T_sel + tie + (t0 * t0 * t0 - t0);
    }
    const double cc = P.use_continuity ? 0.5 : 0.0;
    const GroupConst gc = group_const(n_ref, n_tgt, n);
    const double U = 0.5 * (double)two_u;
    const double p = pval_device_pre(gc.nnn, gc.var0, gc.n12, !P.tie_correct ? 0.0 : (OVR ? __longlong_as_double((long long)tie_sum) : (double)tie_sum), U, gc.mu, cc, P.alternative);
    const double sum_g = (double)vsum, gt = P.gene_total[jc];
    double fc;
    if (OVR) fc = fold_change_device(sum_g, gt - sum_g, gc);
    else fc = (gt == 0.0) ? __longlong_as_double(0x7FF0000000000000ll) : (sum_g / gc.d_tgt) / gt;
    const size_t o = (size_t)g * P.out_ld + jc;
    P.out_p[o] = p;
    P.out_u[o] = U;
    P.out_fc[o] = fc;
}
