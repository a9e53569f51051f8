// CSC one-versus-reference in ONE kernel per gene: the gene's stored non-zeros are regrouped by group code inside
// LDS (counting sort: LDS histogram -> scan -> LDS scatter), the reference run is sorted in place, and the
// lane-per-group rank code (kernels_ovo.h: ovo_lane_groups) reads its runs straight from LDS.  Nothing but the
// CSC arrays is read from HBM and nothing but the [gene][G] statistics is written: the two-kernel route
// (k_csc_segment + k_ovo_rank) writes 7.4 GB of scattered 4-byte stores for 0.9 GB of keys at C3.
//
// Device counterpart of csc_get_contig_cols_into_csr + csr_get_rows_into_csc + _sort_csc_columns_inplace +
// single_group_sparse_ovo_mwu_kernel (utils/sparse/csc.py:139-183, csr.py:103-141, ranking.py:161-172,
// ovo/sparse_ovo.py:22-100) for one gene at a time.
//
// A gene that does not fit (more stored values than the LDS key buffer, a non-reference group with more than 128
// stored values, a reference run longer than the run-end table) sets fallback[gene]; the host sends those genes
// through the two-kernel route.
#pragma once
#include "common.h"
#include "kernels_ovo.h"
#include "kernels_sparse.h"

#define CSCG_NT 1024
#define CSCG_SMALL 32   // longest run the register-history lane-per-group form takes
#define CSCG_MEDIUM 128 // longest run the LDS-history form takes; beyond: the gene leaves this kernel

struct CscGeneParams {
    const void *data, *indices, *indptr; // CSC arrays (device); stored entry k lives at data[k - kshift], indices[k - kshift]
    long long kshift;
    long long col0;                      // first gene of the batch (contiguous batches)
    const int *gene_cols;                // or: the batch's genes as a column list (absolute indices); nullptr = contiguous
    int nb;
    const int *codes;                    // [n_cells] group code per cell; nullptr: `indices` already holds group codes
    const u16 *codes16;                  // the same as 16-bit values (fewer cache lines per gather), or nullptr
    const int *counts;                   // [G]
    int G, ref, dt, is_log1p;
    int key_cap;                         // LDS key slots
    int runend_cap;                      // LDS run-end slots (reference run length limit)
    int ref_buckets;                     // 1: the reference run may take the bucket form (run-end region >= 16 KB, runend_cap <= 8192)
    u32 *fallback;                       // [nb] set to 1 for genes this kernel cannot take
    long long *out_2u;
    u64 *out_tie;
    double *out_sum;
};

static inline size_t cscg_lds_bytes(int G, int key_cap, int runend_cap, size_t key_size, bool buckets = false) {
    size_t b = (size_t)((G + 1 + 3) & ~3) * 4;          // ends
    b += (size_t)CSCG_NT * 4;                            // scan scratch
    b += ovo_runend_bytes(runend_cap, buckets);          // run ends / bucket table
    b += 256;                                            // reductions
    b += (size_t)key_cap * key_size;
    return b;
}

// for every stored entry of [k0, k1), UL per thread and round: body(value, group code).  Two-stage pipeline: the values / row
// indices of round i + 1 are requested before round i's codes[row] gather and LDS work (one workgroup per CU holds the
// LDS, so nothing else hides that round trip).
template <int NT, int UL, typename InT, typename IdxT, typename Body>
__device__ __forceinline__ void csc_for_entries(const InT *__restrict__ data, const IdxT *__restrict__ indices, const int *__restrict__ codes,
                                                const u16 *__restrict__ codes16, long long k0, long long k1, int tid, Body &&body) {
    InT vn[UL];
    IdxT in[UL];
#pragma unroll
    for (int u = 0; u < UL; ++u) {
        const long long k = k0 + u * NT + tid;
        vn[u] = k < k1 ? data[k] : (InT)0;
        in[u] = k < k1 ? indices[k] : (IdxT)0;
    }
    for (long long kb = k0; kb < k1; kb += (long long)NT * UL) {
        InT v[UL];
        int cd[UL];
#pragma unroll
        for (int u = 0; u < UL; ++u) {
            v[u] = vn[u];
            cd[u] = codes16 ? (int)codes16[(long long)in[u]] : (codes ? codes[(long long)in[u]] : (int)in[u]); // (entries past k1: row 0, value 0 -> ignored by the bodies)
        }
        const long long kn = kb + (long long)NT * UL;
        if (kn < k1) { // uniform
#pragma unroll
            for (int u = 0; u < UL; ++u) {
                const long long k = kn + u * NT + tid;
                vn[u] = k < k1 ? data[k] : (InT)0;
                in[u] = k < k1 ? indices[k] : (IdxT)0;
            }
        }
#pragma unroll
        for (int u = 0; u < UL; ++u)
            if (v[u] != (InT)0) body(v[u], cd[u]);
    }
}

template <typename InT, typename IdxT, typename KeyT>
__global__ __launch_bounds__(CSCG_NT) void k_csc_gene(CscGeneParams P) {
    constexpr int NT = CSCG_NT, NW = NT / 64;
    constexpr KeyT ZEROK = KeyInfo<KeyT>::ZEROK;
    extern __shared__ __align__(16) unsigned char smem[];
    u32 *ends = (u32 *)smem;
    size_t off = (size_t)((P.G + 1 + 3) & ~3) * 4;
    u32 *tmp = (u32 *)(smem + off);
    off += (size_t)NT * 4;
    u16 *runend = (u16 *)(smem + off);
    off += ovo_runend_bytes(P.runend_cap, P.ref_buckets != 0);
    u64 *s_red = (u64 *)(smem + off);          // [NW]
    double *s_redd = (double *)(s_red + NW);   // [NW]
    u32 *s_misc = (u32 *)(s_redd + NW);        // [8]
    off += 256;
    KeyT *keybuf = (KeyT *)(smem + off);

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int G = P.G, ref = P.ref;
    const InT *data = (const InT *)P.data;
    const IdxT *indices = (const IdxT *)P.indices, *indptr = (const IdxT *)P.indptr;
    const int n_ref = P.counts[ref];

    for (int gene = blockIdx.x; gene < P.nb; gene += gridDim.x) {
        const long long col = P.gene_cols ? (long long)P.gene_cols[gene] : P.col0 + gene;
        const long long k0 = (long long)indptr[col] - P.kshift, k1 = (long long)indptr[col + 1] - P.kshift;
        // ---- 1. stored non-zeros per group ----
        for (int g = tid; g <= G; g += NT) ends[g] = 0;
        __syncthreads();
        constexpr int UL = 8; // independent entries per thread in flight
        csc_for_entries<NT, UL>(data, indices, P.codes, P.codes16, k0, k1, tid, [&](InT, int cd) { atomicAdd(&ends[cd], 1u); });
        __syncthreads();
        const u32 nA = ends[ref];
        u32 mx = 0;
        for (int g = tid; g < G; g += NT)
            if (g != ref) mx = max(mx, ends[g]);
        mx = (u32)wave_incl_scan_max((int)mx);
        if (lane == 63) tmp[wave] = mx;
        __syncthreads();
        u32 maxg = 0;
        for (int w = 0; w < NW; ++w) maxg = max(maxg, tmp[w]);
        __syncthreads();
        // ---- 2. run offsets ----
        const u32 total = block_excl_scan_inplace<NT>(ends, G, tmp, tid);
        if (total > (u32)P.key_cap || maxg > (u32)CSCG_MEDIUM || nA > (u32)P.runend_cap) { // uniform: this gene takes the two-kernel route
            if (tid == 0) P.fallback[gene] = 1u;
            __syncthreads();
            continue;
        }
        // ---- 3. regroup the keys in LDS ----
        csc_for_entries<NT, UL>(data, indices, P.codes, P.codes16, k0, k1, tid, [&](InT v, int cd) { keybuf[atomicAdd(&ends[cd], 1u)] = key_of(v); });
        __syncthreads();
        // now ends[g] = one past the last key of group g; its run starts at ends[g-1] (0 for g = 0)
        // ---- 4. reference run: sort in place, run ends, T_A, sum ----
        const u32 rstart = ref ? ends[ref - 1] : 0u;
        KeyT *A = keybuf + rstart;
        const u32 zA = (u32)n_ref - nA;
        double rs = 0.0;
        for (u32 i = tid; i < nA; i += NT) rs += P.is_log1p ? key_to_expm1(A[i], P.dt) : key_to_double(A[i], P.dt);
        rs = wave_sum(rs);
        if (lane == 0) s_redd[wave] = rs;
        __syncthreads();
        RefBk<KeyT> bk;
        bk.on = false; bk.tab = runend; bk.kmin = (KeyT)0; bk.last = (KeyT)((1u << OVO_REF_BUCKETS_LG) - 1u); bk.shift = 0; bk.zeros = 0u;
        u32 topA = 0, nnegA = 0;
        u64 T_A = 0;
        double refsum = 0.0;
        if (P.ref_buckets && nA > 0) { // uniform
            // ---- 4a. bucket form: the reference run's keys (no zeros among them) go through registers into value
            // buckets, in place; nA <= runend_cap <= 8 * NT ----
            constexpr int NBK = 1 << OVO_REF_BUCKETS_LG, RK = 8;
            u32 *tab32 = (u32 *)runend;
            u16 *tab16 = (u16 *)runend;
            for (int w = 0; w < NW; ++w) refsum += s_redd[w];
            KeyT rk[RK];
            KeyT tmin = KeyInfo<KeyT>::MAXK, tmax = (KeyT)0;
            u32 ng = 0;
#pragma unroll
            for (int r = 0; r < RK; ++r) {
                const u32 i = (u32)r * NT + tid;
                rk[r] = i < nA ? A[i] : KeyInfo<KeyT>::MAXK;
                if (i < nA) { tmin = rk[r] < tmin ? rk[r] : tmin; tmax = rk[r] > tmax ? rk[r] : tmax; ng += rk[r] < ZEROK ? 1u : 0u; }
            }
            for (int b2 = tid; b2 < NBK / 2; b2 += NT) tab32[b2] = 0u;
#pragma unroll
            for (int d = 32; d > 0; d >>= 1) {
                const KeyT o1 = __shfl_xor(tmin, d), o2 = __shfl_xor(tmax, d);
                tmin = o1 < tmin ? o1 : tmin;
                tmax = o2 > tmax ? o2 : tmax;
            }
            ng = (u32)wave_sum((int)ng);
            __syncthreads(); // (s_redd read above by everyone)
            if (lane == 0) { s_red[wave] = (u64)tmin; s_redd[wave] = __longlong_as_double((long long)(u64)tmax); tmp[wave] = ng; }
            __syncthreads();
            KeyT kmin = KeyInfo<KeyT>::MAXK, kmax = (KeyT)0;
            for (int w = 0; w < NW; ++w) {
                const KeyT m1 = (KeyT)s_red[w], m2 = (KeyT)(u64)__double_as_longlong(s_redd[w]);
                kmin = m1 < kmin ? m1 : kmin;
                kmax = m2 > kmax ? m2 : kmax;
                nnegA += tmp[w];
            }
            bk.kmin = kmin;
            bk.shift = kmax == kmin ? 0 : max(0, (int)(sizeof(KeyT) * 8) - (int)(sizeof(KeyT) == 4 ? __clz((u32)(kmax - kmin)) : __clzll((long long)(u64)(kmax - kmin))) - OVO_REF_BUCKETS_LG);
            __syncthreads();
#pragma unroll
            for (int r = 0; r < RK; ++r) {
                if ((u32)r * NT + tid < nA) {
                    const u32 b2 = refbk_bucket(bk, rk[r]);
                    atomicAdd(&tab32[b2 >> 1], (b2 & 1u) ? 0x10000u : 1u);
                }
            }
            __syncthreads();
            u32 mxb = 0;
            for (int b2 = tid; b2 < NBK; b2 += NT) mxb = max(mxb, (u32)tab16[b2]);
            mxb = (u32)wave_incl_scan_max((int)mxb);
            if (lane == 63) tmp[wave] = mxb;
            __syncthreads();
            mxb = 0;
            for (int w = 0; w < NW; ++w) mxb = max(mxb, tmp[w]);
            __syncthreads();
            if (mxb <= (u32)OVO_REF_MAX_BUCKET) { // uniform
                const int per = NBK / NT, b0 = tid * per;
                u32 sm = 0;
                for (int i = 0; i < per; ++i) sm += tab16[b0 + i];
                tmp[tid] = sm;
                __syncthreads();
                for (int d = 1; d < NT; d <<= 1) {
                    const u32 v2 = (tid >= d) ? tmp[tid - d] : 0u;
                    __syncthreads();
                    tmp[tid] += v2;
                    __syncthreads();
                }
                u32 run = tmp[tid] - sm;
                for (int i = 0; i < per; ++i) { const u32 cnt = tab16[b0 + i]; tab16[b0 + i] = (u16)run; run += cnt; }
                __syncthreads();
#pragma unroll
                for (int r = 0; r < RK; ++r) {
                    if ((u32)r * NT + tid < nA) {
                        const u32 b2 = refbk_bucket(bk, rk[r]);
                        const u32 old = atomicAdd(&tab32[b2 >> 1], (b2 & 1u) ? 0x10000u : 1u);
                        A[(b2 & 1u) ? (old >> 16) : (old & 0xFFFFu)] = rk[r];
                    }
                }
                __syncthreads();
                bk.on = true;
                u64 ta = 0;
#pragma unroll
                for (int r = 0; r < RK; ++r) {
                    if ((u32)r * NT + tid < nA) {
                        u32 lb, a;
                        ref_find<KeyT, true, false>(A, runend, nA, 0u, bk, rk[r], lb, a);
                        ta += (u64)a * a - 1ull;
                    }
                }
                ta = wave_sum(ta);
                if (lane == 0) s_red[wave] = ta;
                __syncthreads();
                for (int w = 0; w < NW; ++w) T_A += s_red[w];
            } else {
                nnegA = 0;
            }
        }
        if (!bk.on) {
            if (!(P.ref_buckets && nA > 0))
                for (int w = 0; w < NW; ++w) refsum += s_redd[w];
            __syncthreads();
            block_bitonic_sort<KeyT, NT>(A, (int)nA, tid);
            topA = top_pow2(nA);
            u64 ta = 0;
            for (u32 i = tid; i < nA; i += NT) {
                const KeyT k = A[i];
                if (i == 0 || A[i - 1] != k) {
                    const u32 e = upper_bound_pow2(A, nA, topA, k);
                    runend[i] = (u16)e;
                    const u64 t = e - i;
                    ta += t * t * t - t;
                }
            }
            ta = wave_sum(ta);
            if (lane == 0) s_red[wave] = ta;
            __syncthreads();
            for (int w = 0; w < NW; ++w) T_A += s_red[w];
            nnegA = lower_bound_pow2(A, nA, topA, ZEROK);
        }
        // ---- 5. every other group: 64 runs per wavefront, one per lane ----
        for (int g0 = wave * 64; g0 < G; g0 += NW * 64) {
            const int gl = g0 + lane;
            const bool has = gl < G && gl != ref;
            const u32 start = (has && gl > 0) ? ends[gl - 1] : 0u;
            const int n = has ? (int)(ends[gl] - start) : 0;
            const int nmax = __builtin_amdgcn_readlane(wave_incl_scan_max(n), 63);
            u64 S2, TT;
            double sum;
            if (nmax <= CSCG_SMALL) ovo_lane_groups<KeyT, CSCG_SMALL, true, false>(keybuf, (long long)start, n, nmax, A, runend, nA, topA, zA, P.dt, P.is_log1p, S2, TT, sum, bk);
            else ovo_lane_groups_mem<KeyT, true, false>(keybuf, (long long)start, n, nmax, A, runend, nA, topA, zA, P.dt, P.is_log1p, S2, TT, sum, bk);
            if (gl < G) {
                const size_t o = (size_t)gene * G + gl;
                if (gl == ref) {
                    P.out_2u[o] = -2;
                    P.out_tie[o] = 0;
                    P.out_sum[o] = refsum;
                } else {
                    const long long n_g = P.counts[gl];
                    const u64 zB = (u64)(n_g - n);
                    S2 += zB * (2ull * nnegA + zA);
                    const u64 t0 = (u64)zA + zB;
                    P.out_2u[o] = 2ll * (long long)n_ref * n_g - (long long)S2;
                    P.out_tie[o] = T_A + 3ull * TT + (t0 * t0 * t0 - t0);
                    P.out_sum[o] = sum;
                }
            }
        }
        __syncthreads();
    }
}

// ---- regroup only: the first half of k_csc_gene (count per group, offsets, regroup the keys in LDS) followed by a
// coalesced copy-out of the gene's keys, their group codes and the (gene, group) offsets -- the layout k_csc_segment
// produces for the rank kernels of the two-kernel route, without its two uncoalesced passes (a scattered 4-byte store
// and a codes[row] lookup per entry each: 64 cache lines per wave instruction, which made that kernel address-path
// bound at 8.5 ms for C3).  The value / group code of every entry is read once and kept in registers between the
// counting and the scattering pass.  Genes with more stored values than the LDS key buffer, or than CSCR_CACHE entries
// per thread, set fallback[gene] and are redone by k_csc_segment.
#define CSCR_CACHE 40 // entries per thread kept in registers (1024 threads: 40960 stored values per gene)
struct CscRegroupParams {
    const void *data, *indices, *indptr;
    long long kshift, col0;
    const int *gene_cols;   // optional column list (absolute), else col0 + gene
    const u32 *gene_base;   // with gene_cols: where each gene's keys start in Xs; else k0 - indptr[col0]
    int nb;
    const int *codes;       // nullptr: indices already hold group codes
    int G, key_cap, count_limit;
    void *Xs;               // out: keys, group-contiguous per gene
    u32 *vals;              // out (optional): group code per key
    u32 *seg_ptr;           // out: [nb][G+1]
    u32 *gene_flags;        // out (optional): a value outside the count table
    u32 *fallback;          // out: gene not handled here
};
template <typename InT, typename IdxT, typename KeyT>
__global__ __launch_bounds__(CSCG_NT) void k_csc_regroup(CscRegroupParams P) {
    constexpr int NT = CSCG_NT, UL = 8, NB = CSCR_CACHE / UL;
    extern __shared__ __align__(16) unsigned char smem[];
    u32 *ends = (u32 *)smem;
    u32 *tmp = ends + ((P.G + 1 + 3) & ~3);
    KeyT *keybuf = (KeyT *)(tmp + NT);
    const int tid = threadIdx.x;
    const int G = P.G;
    const InT *data = (const InT *)P.data;
    const IdxT *indices = (const IdxT *)P.indices, *indptr = (const IdxT *)P.indptr;
    const long long base0 = (long long)indptr[P.col0];
    for (int gene = blockIdx.x; gene < P.nb; gene += gridDim.x) {
        const long long col = P.gene_cols ? (long long)P.gene_cols[gene] : P.col0 + gene;
        const long long k0 = (long long)indptr[col] - P.kshift, k1 = (long long)indptr[col + 1] - P.kshift;
        const u32 gbase = P.gene_cols ? P.gene_base[gene] : (u32)(k0 + P.kshift - base0);
        if (k1 - k0 > (long long)NT * CSCR_CACHE || k1 - k0 > (long long)P.key_cap) { // uniform
            if (tid == 0) P.fallback[gene] = 1u;
            continue;
        }
        for (int g = tid; g <= G; g += NT) ends[g] = 0;
        __syncthreads();
        InT v[NB][UL];
        int cd[NB][UL];
        bool viol = false;
#pragma unroll
        for (int b = 0; b < NB; ++b) {
            const long long kb = k0 + (long long)b * NT * UL;
            if (kb < k1) { // uniform
#pragma unroll
                for (int u = 0; u < UL; ++u) {
                    const long long k = kb + u * NT + tid;
                    v[b][u] = k < k1 ? data[k] : (InT)0;
                    cd[b][u] = k < k1 ? (P.codes ? P.codes[(long long)indices[k]] : (int)indices[k]) : 0;
                }
#pragma unroll
                for (int u = 0; u < UL; ++u)
                    if (v[b][u] != (InT)0) { atomicAdd(&ends[cd[b][u]], 1u); viol |= !count_ok(v[b][u], P.count_limit); }
            } else {
#pragma unroll
                for (int u = 0; u < UL; ++u) { v[b][u] = (InT)0; cd[b][u] = 0; }
            }
        }
        __syncthreads();
        const u32 total = block_excl_scan_inplace<NT>(ends, G, tmp, tid);
        u32 *sp = P.seg_ptr + (size_t)gene * (G + 1);
        for (int g = tid; g < G; g += NT) sp[g] = gbase + ends[g];
        if (tid == 0) sp[G] = gbase + total;
        __syncthreads();
#pragma unroll
        for (int b = 0; b < NB; ++b)
#pragma unroll
            for (int u = 0; u < UL; ++u)
                if (v[b][u] != (InT)0) {
                    const u32 p = atomicAdd(&ends[cd[b][u]], 1u);
                    keybuf[p] = key_of(v[b][u]);
                }
        __syncthreads();
        KeyT *Xs = (KeyT *)P.Xs + gbase;
        // after the scatter ends[g] = one past group g's run: the group of staged slot i is the first g with ends[g] > i
        for (u32 i = tid; i < total; i += NT) {
            Xs[i] = keybuf[i];
            if (P.vals) {
                int lo = 0, hi = G - 1;
                while (lo < hi) { const int mid = (lo + hi) >> 1; if (ends[mid] > i) hi = mid; else lo = mid + 1; }
                P.vals[gbase + i] = (u32)lo;
            }
        }
        if (P.gene_flags && viol) P.gene_flags[gene] = 1u;
        __syncthreads();
    }
}

