// What the vector-memory front end charges for the access shapes of the count-valued CSC kernel (kernels_csc_counts.h), in
// isolation: a C3-shaped CSC index array (300k rows, 8000 columns, ~10 % stored, rows ascending inside a column) is walked
// by 512-thread workgroups, one column each, with
//   idx1 / idx2 / idx4     the row indices alone, 1 / 2 / 4 entries per lane and load instruction
//   idx1+val1, idx4+val4   row indices and values
//   g1 / g2 / g4           row indices (1 / 2 / 4 per lane) + one 16-bit code gather per entry
//   g1+val1, g1+val4       the kernel's shape today, and with the values as one 16-byte load per four entries
//   lds                    codes of the wavefront's row span staged into LDS by coalesced 16-byte loads, then read from LDS
// Every variant xors what it loads into one word per thread (stored once), so nothing is optimised away.
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -I illico_amd/csrc -o tools/micro/ta_gather tools/micro/ta_gather.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include <algorithm>
#include "common.h"

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

__device__ __forceinline__ u32 mix(u32 a, u32 b) {
    u32 h = a * 0x9E3779B1u ^ (b + 0x7F4A7C15u) * 0x85EBCA6Bu;
    h ^= h >> 15; h *= 0x2C1B3C6Du; h ^= h >> 12; h *= 0x297A2D39u; h ^= h >> 15;
    return h;
}
__global__ void k_count(int N, u32 thr, int *cnt) {
    const int gene = blockIdx.x;
    int c = 0;
    for (int r = threadIdx.x; r < N; r += blockDim.x) c += mix(gene, r) < thr ? 1 : 0;
    c = wave_sum(c);
    if ((threadIdx.x & 63) == 0) atomicAdd(&cnt[gene], c);
}
__global__ void k_fill(int N, u32 thr, const int *indptr, float *data, int *indices) {
    const int gene = blockIdx.x, lane = threadIdx.x;
    int pos = indptr[gene];
    for (int r0 = 0; r0 < N; r0 += 64) {
        const int r = r0 + lane;
        const bool st = r < N && mix(gene, r) < thr;
        const u64 m = __ballot(st);
        if (st) {
            const int p = pos + (int)__popcll(m & ((1ull << lane) - 1ull));
            data[p] = (float)(1 + (mix(gene * 7919u + 13u, r) & 15u));
            indices[p] = r;
        }
        pos += (int)__popcll(m);
    }
}

template <typename T, int N> struct alignas(sizeof(T) * N) Vec { T v[N]; };

// LW entries per lane and index load; VW: 0 = no values, else entries per lane and value load; GATHER: code gather per entry
template <int NT, int UL, int LW, int VW, bool GATHER>
__global__ __launch_bounds__(NT) void k_walk(const int *indptr, const int *indices, const float *data, const u16 *codes16, int M, u32 *out) {
    const int tid = threadIdx.x;
    u32 acc = 0;
    for (int gene = blockIdx.x; gene < M; gene += gridDim.x) {
        const int k0 = indptr[gene], k1 = indptr[gene + 1];
        const int ka = k0 & ~3; // 16-byte aligned start (entries before k0 belong to the previous column: harmless here)
        for (int kb = ka; kb < k1; kb += NT * UL) {
            int in[UL];
            float v[UL];
#pragma unroll
            for (int u = 0; u < UL / LW; ++u) {
                const int k = kb + (u * NT + tid) * LW;
                if (k + LW <= k1) {
                    const Vec<int, LW> x = *(const Vec<int, LW> *)&indices[k];
#pragma unroll
                    for (int j = 0; j < LW; ++j) in[u * LW + j] = x.v[j];
                } else {
#pragma unroll
                    for (int j = 0; j < LW; ++j) in[u * LW + j] = 0;
                }
            }
            if (VW) {
                constexpr int VWN = VW ? VW : 1;
#pragma unroll
                for (int u = 0; u < UL / VWN; ++u) {
                    const int k = kb + (u * NT + tid) * VWN;
                    if (k + VWN <= k1) {
                        const Vec<float, VWN> x = *(const Vec<float, VWN> *)&data[k];
#pragma unroll
                        for (int j = 0; j < VWN; ++j) v[u * VWN + j] = x.v[j];
                    } else {
#pragma unroll
                        for (int j = 0; j < VWN; ++j) v[u * VWN + j] = 0.f;
                    }
                }
            }
#pragma unroll
            for (int e = 0; e < UL; ++e) {
                u32 c = GATHER ? (u32)codes16[in[e]] : (u32)in[e];
                if (VW) c ^= __float_as_uint(v[e]);
                acc ^= c;
            }
        }
    }
    out[blockIdx.x * NT + tid] = acc;
}

// codes of each wavefront's row span staged through LDS: per 64 consecutive entries, rows [r_lo, r_hi] -> coalesced 16-byte
// loads of codes16[r_lo & ~7 ...] into the wavefront's 2-KB LDS slot, then one ds_read_u16 per entry (spans beyond 1024 rows
// fall back to the gather for the lanes past the slot)
template <int NT, int UL>
__global__ __launch_bounds__(NT) void k_walk_lds(const int *indptr, const int *indices, const u16 *codes16, int n_rows, int M, u32 *out) {
    __shared__ __align__(16) u16 slot[NT / 64][1024];
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    u32 acc = 0;
    for (int gene = blockIdx.x; gene < M; gene += gridDim.x) {
        const int k0 = indptr[gene], k1 = indptr[gene + 1];
        for (int kb = k0; kb < k1; kb += NT * UL) {
            int in[UL];
#pragma unroll
            for (int u = 0; u < UL; ++u) {
                const int k = kb + u * NT + tid;
                in[u] = k < k1 ? indices[k] : 0x7FFFFFFF;
            }
#pragma unroll
            for (int u = 0; u < UL; ++u) {
                const int r_lo = __builtin_amdgcn_readfirstlane(in[u]) & ~7;
                if (r_lo == (0x7FFFFFFF & ~7)) continue; // (uniform: the whole wavefront is past the column's end)
                // two 16-byte loads per lane cover 1024 rows
                const int ra = r_lo + lane * 8, rb = ra + 512;
                uint4 a = make_uint4(0, 0, 0, 0), b = a;
                if (ra + 8 <= n_rows) a = *(const uint4 *)&codes16[ra];
                if (rb + 8 <= n_rows) b = *(const uint4 *)&codes16[rb];
                *(uint4 *)&slot[w][lane * 8] = a;
                *(uint4 *)&slot[w][512 + lane * 8] = b;
                const int d = in[u] - r_lo;
                u32 c;
                if (in[u] == 0x7FFFFFFF) c = 0;
                else if (d < 1024 && in[u] + 8 <= n_rows) c = slot[w][d];
                else c = codes16[in[u]];
                acc ^= c;
            }
        }
    }
    out[blockIdx.x * NT + tid] = acc;
}

int main(int argc, char **argv) {
    const int N = 300000, M = 8000, G = 2000;
    const u32 thr = (u32)(0.1 * 4294967296.0);
    int *d_cnt; CK(hipMalloc(&d_cnt, M * 4)); CK(hipMemset(d_cnt, 0, M * 4));
    k_count<<<M, 256>>>(N, thr, d_cnt);
    std::vector<int> cnt(M), indptr(M + 1, 0);
    CK(hipMemcpy(cnt.data(), d_cnt, M * 4, hipMemcpyDeviceToHost));
    for (int j = 0; j < M; ++j) indptr[j + 1] = indptr[j] + cnt[j];
    const long long nnz = indptr[M];
    int *d_indptr, *d_indices; float *d_data;
    CK(hipMalloc(&d_indptr, (M + 1) * 4)); CK(hipMalloc(&d_indices, nnz * 4 + 64)); CK(hipMalloc(&d_data, nnz * 4 + 64));
    CK(hipMemcpy(d_indptr, indptr.data(), (M + 1) * 4, hipMemcpyHostToDevice));
    k_fill<<<M, 64>>>(N, thr, d_indptr, d_data, d_indices);
    std::vector<u16> codes16(N + 64);
    srand(1);
    for (int i = 0; i < N; ++i) codes16[i] = (u16)(rand() % G);
    u16 *d_codes16; CK(hipMalloc(&d_codes16, (N + 64) * 2)); CK(hipMemcpy(d_codes16, codes16.data(), (N + 64) * 2, hipMemcpyHostToDevice));
    u32 *d_out; CK(hipMalloc(&d_out, (size_t)M * 1024 * 4));
    CK(hipDeviceSynchronize());
    printf("nnz %lld, index bytes %.3f GB, index + value bytes %.3f GB\n", nnz, nnz * 4e-9, nnz * 8e-9);
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    auto time_it = [&](const char *name, auto launch, double gb) {
        if (argc > 1) { bool hit = false; for (int i = 1; i < argc; ++i) hit |= strstr(name, argv[i]) != nullptr; if (!hit) return; }
        launch(); CK(hipDeviceSynchronize());
        CK(hipEventRecord(a));
        for (int i = 0; i < 5; ++i) launch();
        CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
        float ms; CK(hipEventElapsedTime(&ms, a, b)); ms /= 5;
        u32 h[64]; CK(hipMemcpy(h, d_out + 12345, sizeof h, hipMemcpyDeviceToHost));
        u32 cs = 0; for (u32 x : h) cs = cs * 31 + x;
        printf("%-34s %.3f ms  %.2f TB/s of stream bytes  checksum %08x\n", name, ms, gb / ms, cs);
    };
    const double gi = nnz * 4e-9, giv = nnz * 8e-9;
#define W(NAME, NT, UL, LW, VW, GATH, GB) time_it(NAME, [&] { hipLaunchKernelGGL((k_walk<NT, UL, LW, VW, GATH>), dim3(M), dim3(NT), 0, 0, d_indptr, d_indices, d_data, d_codes16, M, d_out); }, GB)
    W("idx1       512x16", 512, 16, 1, 0, false, gi);
    W("idx2       512x16", 512, 16, 2, 0, false, gi);
    W("idx4       512x16", 512, 16, 4, 0, false, gi);
    W("idx1+val1  512x16", 512, 16, 1, 1, false, giv);
    W("idx4+val4  512x16", 512, 16, 4, 4, false, giv);
    W("g1         512x16", 512, 16, 1, 0, true, gi);
    W("g2         512x16", 512, 16, 2, 0, true, gi);
    W("g4         512x16", 512, 16, 4, 0, true, gi);
    W("g1+val1    512x16", 512, 16, 1, 1, true, giv);
    W("g1+val4    512x16", 512, 16, 1, 4, true, giv);
    W("g2+val2    512x16", 512, 16, 2, 2, true, giv);
    W("g2+val4    512x16", 512, 16, 2, 4, true, giv);
    W("g4+val4    512x16", 512, 16, 4, 4, true, giv);
    W("g1+val1    1024x8", 1024, 8, 1, 1, true, giv);
    W("g1+val4    1024x8", 1024, 8, 1, 4, true, giv);
    W("g2+val2    1024x8", 1024, 8, 2, 2, true, giv);
    W("g1+val1    256x16", 256, 16, 1, 1, true, giv);
    W("g1+val1    512x8", 512, 8, 1, 1, true, giv);
    W("g1+val1    512x32", 512, 32, 1, 1, true, giv);
    time_it("lds-staged 512x16", [&] { hipLaunchKernelGGL((k_walk_lds<512, 16>), dim3(M), dim3(512), 0, 0, d_indptr, d_indices, d_codes16, N, M, d_out); }, gi);
    time_it("lds-staged 1024x8", [&] { hipLaunchKernelGGL((k_walk_lds<1024, 8>), dim3(M), dim3(1024), 0, 0, d_indptr, d_indices, d_codes16, N, M, d_out); }, gi);
    return 0;
}
