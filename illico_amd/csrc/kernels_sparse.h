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
                                                        long long kshift) {
    extern __shared__ __align__(16) unsigned char smem[];
    u32 *hist = (u32 *)smem;
    u32 *tmp = hist + ((G + 3) & ~3);
    const int tid = threadIdx.x;
    // batch = columns col0 .. col0+nb-1, or (gene_cols != nullptr) an arbitrary list of columns whose keys are
    // packed at gene_base[gene].  Stored entry k of the caller's matrix lives at data[k - kshift] / indices[k - kshift]
    // (a host-resident matrix is staged from its first needed entry on).
    const long long base0 = (long long)indptr[col0];
    for (int gene = blockIdx.x; gene < nb; gene += gridDim.x) {
        const long long col = gene_cols ? (long long)gene_cols[gene] : col0 + gene;
        const long long k0 = (long long)indptr[col], k1 = (long long)indptr[col + 1];
        const u32 gbase = gene_cols ? gene_base[gene] : (u32)(k0 - base0);
        for (int g = tid; g < G; g += SEG_NT) hist[g] = 0;
        __syncthreads();
        for (long long k = k0 - kshift + tid; k < k1 - kshift; k += SEG_NT)
            if (data[k] != (InT)0) atomicAdd(&hist[codes[(long long)indices[k]]], 1u);
        __syncthreads();
        u32 total = block_excl_scan_inplace<SEG_NT>(hist, G, tmp, tid);
        u32 *sp = seg_ptr + (size_t)gene * (G + 1);
        for (int g = tid; g < G; g += SEG_NT) sp[g] = gbase + hist[g];
        if (tid == 0) sp[G] = gbase + total;
        __syncthreads();
        bool viol = false;
        for (long long k = k0 - kshift + tid; k < k1 - kshift; k += SEG_NT) {
            InT v = data[k];
            if (v != (InT)0) {
                int c = codes[(long long)indices[k]];
                u32 pos = gbase + atomicAdd(&hist[c], 1u);
                Xs[pos] = key_of(v);
                if (vals) vals[pos] = (u32)c;
                viol |= !count_ok(v, count_limit);
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
__global__ __launch_bounds__(SEG_NT) void k_seg_scan(u32 *__restrict__ seg, int G, int nb, u32 *__restrict__ gene_tot) {
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
__global__ __launch_bounds__(1024) void k_gene_base_scan(const u32 *__restrict__ gene_tot, int nb, u32 *__restrict__ gene_base) {
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
__global__ void k_seg_add_base(u32 *__restrict__ seg, u32 *__restrict__ cursor, const u32 *__restrict__ gene_base, int G, int nb) {
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
template <typename InT, typename IdxT>
__global__ __launch_bounds__(DENS_NT) void k_csr_densify(const InT *__restrict__ data, const IdxT *__restrict__ indices,
                                                         const IdxT *__restrict__ indptr, int n_rows, long long c0, int W,
                                                         float *__restrict__ D, long long ldD) {
    __shared__ __align__(16) float row[DENS_WB];
    constexpr int UL = 4; // stored entries per thread requested together (column and value loads are independent)
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
            if (col[j] >= 0 && cj >= 0 && cj < wb) {
                float f = (float)v[j];
                if (!((InT)f == v[j])) f = __int_as_float(0x7FC00000); // not representable (or NaN): send the gene elsewhere
                row[cj] = f;
            }
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
            const int wb = min(DENS_WB, W - cb), wb4 = (wb + 3) & ~3;
            for (int i = tid * 4; i < wb4; i += DENS_NT * 4) *(float4 *)&row[i] = make_float4(0.f, 0.f, 0.f, 0.f);
            __syncthreads();
            if (cb > 0) fetch(s, e); // wide windows: the row is walked once per column block
            scatter(cb, wb);
            for (long long k0 = s + DENS_NT * UL; k0 < e; k0 += DENS_NT * UL) { fetch(k0, e); scatter(cb, wb); }
            __syncthreads();
            // the next row's first entries are requested before this row is copied out: their latency hides behind the stores
            if (cb + DENS_WB >= W) fetch(sn, en);
            float *dst = D + (long long)r * ldD + cb; // ldD and cb are multiples of 4: 16-byte aligned
            for (int i = tid * 4; i < wb4; i += DENS_NT * 4) {
                if (i + 4 <= wb) *(float4 *)&dst[i] = *(const float4 *)&row[i];
                else for (int j = i; j < wb; ++j) dst[j] = row[j];
            }
            __syncthreads();
        }
        s = sn; e = en;
    }
}

// replaces check_indices_sorted_per_parcel (utils/ranking.py:245-273) for device-resident CSR
template <typename IdxT>
__global__ void k_csr_sorted_check(const IdxT *__restrict__ indices, const IdxT *__restrict__ indptr, int n_rows, int *__restrict__ bad) {
    const int lane = threadIdx.x & 63;
    const int wave_global = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const int n_waves = (gridDim.x * blockDim.x) >> 6;
    for (int row = wave_global; row < n_rows; row += n_waves) {
        long long s = (long long)indptr[row], e = (long long)indptr[row + 1];
        for (long long k = s + 1 + lane; k < e; k += 64)
            if (indices[k] < indices[k - 1]) *bad = 1;
    }
}
