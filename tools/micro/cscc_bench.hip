// Microbenchmark harness for the count-valued CSC kernel (kernels_csc_counts.h) on a C3-shaped synthetic matrix built on the
// device: 300k cells x 8k genes, ~10 % stored entries, Poisson-like small integers, 2000 groups (one of 10 000 cells).
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -I illico_amd/csrc -o tools/micro/cscc_bench tools/micro/cscc_bench.hip
// Run:   tools/micro/cscc_bench [variant ...]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include <algorithm>
#include "common.h"
#include "kernels_csc_counts.h"
#include "kernels_finalize.h"

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

__device__ __forceinline__ u32 mix(u32 a, u32 b) {
    u32 h = a * 0x9E3779B1u ^ (b + 0x7F4A7C15u) * 0x85EBCA6Bu;
    h ^= h >> 15; h *= 0x2C1B3C6Du; h ^= h >> 12; h *= 0x297A2D39u; h ^= h >> 15;
    return h;
}
// stored iff mix(gene, row) < density * 2^32
__global__ void k_count(int N, int M, u32 thr, int *cnt) {
    const int gene = blockIdx.x;
    int c = 0;
    for (int r = threadIdx.x; r < N; r += blockDim.x) c += mix(gene, r) < thr ? 1 : 0;
    c = wave_sum(c);
    if ((threadIdx.x & 63) == 0) atomicAdd(&cnt[gene], c);
}
__global__ void k_fill(int N, int M, u32 thr, const int *indptr, float *data, int *indices) {
    // one workgroup of 64 threads per gene: ordered fill by ballots
    const int gene = blockIdx.x, lane = threadIdx.x;
    int pos = indptr[gene];
    const u32 mean = 1 + (mix(gene, 0xABCDEFu) % 15u);
    for (int r0 = 0; r0 < N; r0 += 64) {
        const int r = r0 + lane;
        const bool st = r < N && mix(gene, r) < thr;
        const u64 m = __ballot(st);
        if (st) {
            const int p = pos + (int)__popcll(m & ((1ull << lane) - 1ull));
            const u32 h = mix(gene * 7919u + 13u, r);
            // crude Poisson-like value around `mean`: sum of 4 uniform draws
            u32 v = ((h & 0xFF) + ((h >> 8) & 0xFF) + ((h >> 16) & 0xFF) + (h >> 24)) * (2 * mean) / 1024;
            v = v < 1 ? 1 : (v > 63 ? 63 : v);
            data[p] = (float)v;
            indices[p] = r;
        }
        pos += (int)__popcll(m);
    }
}

int main(int argc, char **argv) {
    const int N = 300000, M = 8000, G = 2000, n_ref = 10000;
    const double density = 0.1;
    const u32 thr = (u32)(density * 4294967296.0);
    int *d_cnt; CK(hipMalloc(&d_cnt, M * 4)); CK(hipMemset(d_cnt, 0, M * 4));
    k_count<<<M, 256>>>(N, M, thr, d_cnt);
    std::vector<int> cnt(M), indptr(M + 1, 0);
    CK(hipMemcpy(cnt.data(), d_cnt, M * 4, hipMemcpyDeviceToHost));
    for (int j = 0; j < M; ++j) indptr[j + 1] = indptr[j] + cnt[j];
    const long long nnz = indptr[M];
    int *d_indptr, *d_indices; float *d_data;
    CK(hipMalloc(&d_indptr, (M + 1) * 4)); CK(hipMalloc(&d_indices, nnz * 4)); CK(hipMalloc(&d_data, nnz * 4));
    CK(hipMemcpy(d_indptr, indptr.data(), (M + 1) * 4, hipMemcpyHostToDevice));
    k_fill<<<M, 64>>>(N, M, thr, d_indptr, d_data, d_indices);
    // groups
    std::vector<int> codes(N), counts(G, 0);
    srand(1);
    for (int i = 0; i < N; ++i) codes[i] = i < n_ref ? 0 : 1 + rand() % (G - 1);
    std::random_shuffle(codes.begin(), codes.end());
    for (int i = 0; i < N; ++i) counts[codes[i]]++;
    int *d_codes, *d_counts; CK(hipMalloc(&d_codes, N * 4)); CK(hipMalloc(&d_counts, G * 4));
    CK(hipMemcpy(d_codes, codes.data(), N * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(d_counts, counts.data(), G * 4, hipMemcpyHostToDevice));
    std::vector<u16> codes16(codes.begin(), codes.end());
    u16 *d_codes16; CK(hipMalloc(&d_codes16, N * 2)); CK(hipMemcpy(d_codes16, codes16.data(), N * 2, hipMemcpyHostToDevice));
    long long *s2u; u64 *stie; double *ssum; u32 *fb;
    CK(hipMalloc(&s2u, (size_t)M * G * 8)); CK(hipMalloc(&stie, (size_t)M * G * 8)); CK(hipMalloc(&ssum, (size_t)M * G * 8)); CK(hipMalloc(&fb, M * 4));
    CK(hipMemset(fb, 0, M * 4));
    CK(hipDeviceSynchronize());
    printf("nnz %lld (%.1f per gene), algorithmic bytes %.3f GB\n", nnz, (double)nnz / M, (nnz * 8.0 + (M + 1) * 4 + 4.0 * N + 24.0 * G * M) / 1e9);

    CscCountsParams P;
    P.g_lo = 0; P.G_total = G; // (no group windows here)
    P.data = d_data; P.indices = d_indices; P.indptr = d_indptr; P.kshift = 0; P.col0 = 0; P.gene_cols = nullptr; P.nb = M; P.codes16 = d_codes16;
    P.counts = d_counts; P.G = G; P.n_cells = N; P.big_slot = nullptr; P.fallback = fb; P.out_2u = s2u; P.out_tie = stie; P.out_sum = ssum; P.gene_total = nullptr; P.verdict = nullptr; P.pack16 = 0;
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    auto time_it = [&](const char *name, auto launch) {
        launch(); CK(hipDeviceSynchronize());
        hipEventRecord(a);
        for (int i = 0; i < 5; ++i) launch();
        hipEventRecord(b); hipEventSynchronize(b);
        float ms; hipEventElapsedTime(&ms, a, b); ms /= 5;
        // checksum of the statistics (variants must agree)
        std::vector<long long> h(4096);
        CK(hipMemcpy(h.data(), s2u + (size_t)1234 * G, 4096 * 8, hipMemcpyDeviceToHost));
        long long cs = 0; for (auto x : h) cs = cs * 31 + x;
        std::vector<u64> ht(4096);
        CK(hipMemcpy(ht.data(), stie + (size_t)4321 * G, 4096 * 8, hipMemcpyDeviceToHost));
        u64 ct = 0; for (auto x : ht) ct = ct * 31 + x;
        u32 nfb = 0; std::vector<u32> hf(M); CK(hipMemcpy(hf.data(), fb, M * 4, hipMemcpyDeviceToHost)); for (auto x : hf) nfb += x;
        printf("%-28s %.3f ms   checksum %016llx %016llx  fallback genes %u\n", name, ms, (unsigned long long)cs, (unsigned long long)ct, nfb);
    };
    auto want = [&](const char *name) {
        if (argc <= 1) return true;
        for (int i = 1; i < argc; ++i) if (strstr(name, argv[i])) return true;
        return false;
    };
    for (int ovr = 0; ovr < 2; ++ovr) {
        P.ref = ovr ? -1 : 0;
        if (ovr) { // the 10 000-cell group would need the 32-bit rows (HAS_BIG): uniform groups for the OVR timing
            for (int i = 0; i < N; ++i) codes[i] = i % G;
            std::fill(counts.begin(), counts.end(), 0);
            for (int i = 0; i < N; ++i) counts[codes[i]]++;
            for (int i = 0; i < N; ++i) codes16[i] = (u16)codes[i];
            CK(hipMemcpy(d_codes, codes.data(), N * 4, hipMemcpyHostToDevice));
            CK(hipMemcpy(d_counts, counts.data(), G * 4, hipMemcpyHostToDevice));
            CK(hipMemcpy(d_codes16, codes16.data(), N * 2, hipMemcpyHostToDevice));
        }
        const size_t lds = cscc_lds_bytes(G, 0);
#define VAR(NAME, OVRF, WTF, ABLF)                                                                                          \
        if (want(NAME)) {                                                                                                       \
            auto kern = k_csc_counts<float, int, OVRF, 64, false, true, true, WTF, ABLF>;                                        \
            CK(hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));                   \
            time_it(ovr ? NAME " ovr" : NAME " ovo", [&] { hipLaunchKernelGGL(kern, dim3(M), dim3(CSCC_NT), lds, 0, P); });       \
        }
#define VARS(OVRF)                                                                                                          \
        VAR("base", OVRF, false, 0) VAR("wt", OVRF, true, 0) VAR("base-nogather", OVRF, false, 1) VAR("base-noatomic", OVRF, false, 2)   \
        VAR("base-nogather-noatomic", OVRF, false, 3) VAR("base-nosweep", OVRF, false, 4) VAR("base-nostore", OVRF, false, 8)              \
        VAR("base-nosweep-nostore", OVRF, false, 12) VAR("base-noentries", OVRF, false, 16) VAR("base-entries-only", OVRF, false, 12 | 32)     \
        VAR("wt-noentries", OVRF, true, 16) VAR("wt-nostore", OVRF, true, 8)
        if (ovr) { VARS(true) } else { VARS(false) }
#define VARN(NAME, OVRF, WTF)                                                                                               \
        if (want(NAME)) {                                                                                                       \
            auto kern = k_csc_counts<float, int, OVRF, 64, false, true, true, WTF, 0, 1024>;                              \
            CK(hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));                   \
            time_it(ovr ? NAME " ovr" : NAME " ovo", [&] { hipLaunchKernelGGL(kern, dim3(M), dim3(1024), lds, 0, P); });          \
        }
if (ovr) { VAR("stagger32-wt", true, true, 64) VAR("stagger96-wt", true, true, 128) } else { VAR("stagger32-wt", false, true, 64) VAR("stagger96-wt", false, true, 128) }
#define VARL(NAME, OVRF, WTF, ABLF)                                                                                         \
        if (want(NAME)) {                                                                                                       \
            auto kern = k_csc_counts<float, int, OVRF, 64, false, true, true, WTF, ABLF, CSCC_NT, true>;                         \
            CK(hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));                   \
            time_it(ovr ? NAME " ovr" : NAME " ovo", [&] { hipLaunchKernelGGL(kern, dim3(M), dim3(CSCC_NT), lds, 0, P); });       \
        }
#define VARLB(NAME, OVRF, WTF)                                                                                              \
        if (want(NAME)) {                                                                                                       \
            auto kern = k_csc_counts<float, int, OVRF, 64, false, true, true, WTF, 0, CSCC_NT, true, false>;                     \
            CK(hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));                   \
            time_it(ovr ? NAME " ovr" : NAME " ovo", [&] { hipLaunchKernelGGL(kern, dim3(M), dim3(CSCC_NT), lds, 0, P); });       \
        }
        if (ovr) { VARLB("leanload-wt", true, true) VARLB("leanload-base", true, false) } else { VARLB("leanload-wt", false, true) VARLB("leanload-base", false, false) }
        if (ovr) { VARL("lean-wt", true, true, 0) VARL("lean-base", true, false, 0) VARL("lean-wt-nogather", true, true, 1) } else { VARL("lean-wt", false, true, 0) VARL("lean-base", false, false, 0) VARL("lean-wt-entries-only", false, true, 12 | 32) VARL("lean-wt-nogather", false, true, 1) VARL("lean-wt-noatomic", false, true, 2) VARL("lean-wt-noentries", false, true, 16) }
        if (ovr) { VARN("nt1024-wt", true, true) } else { VARN("nt1024-wt", false, true) VARN("nt1024-base", false, false) }
        if (!ovr && want("finalize")) { // k_finalize on the statistics, alone and under the next batch's k_csc_counts
            double *op, *ou, *ofc; int *d_cnt32 = d_counts;
            CK(hipMalloc(&op, (size_t)M * G * 8)); CK(hipMalloc(&ou, (size_t)M * G * 8)); CK(hipMalloc(&ofc, (size_t)M * G * 8));
            FinalizeParams F;
            F.in_2u = s2u; F.in_tie = stie; F.in_sum = ssum; F.gene_total = nullptr; F.counts = d_cnt32; F.G = G; F.ref = 0; F.nb = M; F.n_cells = N;
            F.use_continuity = 1; F.tie_correct = 1; F.alternative = 0; F.packed = 0; F.out_p = op; F.out_u = ou; F.out_fc = ofc; F.out_ld = M; F.col_map = nullptr;
            time_it("k_finalize alone", [&] { hipLaunchKernelGGL(k_finalize, dim3((M + 31) / 32, (G + 31) / 32), dim3(256), 0, 0, F); });
            auto kern = k_csc_counts<float, int, false, 64, false, true, true, true, 0, CSCC_NT, true, false>;
            CK(hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
            time_it("counts+finalize serial", [&] {
                hipLaunchKernelGGL(kern, dim3(M), dim3(CSCC_NT), lds, 0, P);
                hipLaunchKernelGGL(k_finalize, dim3((M + 31) / 32, (G + 31) / 32), dim3(256), 0, 0, F);
            });
            { // 16-byte statistics
                CscCountsParams Pp = P; Pp.pack16 = 1;
                FinalizeParams Fp = F; Fp.packed = 1;
                time_it("counts+finalize, 16-byte statistics", [&] {
                    hipLaunchKernelGGL(kern, dim3(M), dim3(CSCC_NT), lds, 0, Pp);
                    hipLaunchKernelGGL(k_finalize, dim3((M + 31) / 32, (G + 31) / 32), dim3(256), 0, 0, Fp);
                });
            }
            hipStream_t s2; CK(hipStreamCreateWithFlags(&s2, hipStreamNonBlocking));
            hipStream_t s1; CK(hipStreamCreateWithFlags(&s1, hipStreamNonBlocking));
            for (int NBATCH : {2}) {
                std::vector<hipEvent_t> ev(NBATCH), evf(NBATCH);
                for (auto &evt : ev) CK(hipEventCreateWithFlags(&evt, hipEventDisableTiming));
                for (auto &evt : evf) CK(hipEventCreateWithFlags(&evt, hipEventDisableTiming));
                char nm[64]; snprintf(nm, sizeof nm, "counts||finalize %d batches", NBATCH);
                time_it(nm, [&] {
                    // default stream work of time_it's events brackets both streams: fork from / join into stream 0
                    hipEvent_t fork = ev[0];
                    hipEventRecord(fork, 0); hipStreamWaitEvent(s1, fork, 0); hipStreamWaitEvent(s2, fork, 0);
                    for (int b = 0; b < NBATCH; ++b) {
                        const int g0 = (int)((long long)M * b / NBATCH), g1 = (int)((long long)M * (b + 1) / NBATCH);
                        CscCountsParams Pb = P; Pb.col0 = g0; Pb.nb = g1 - g0; Pb.out_2u = s2u + (size_t)g0 * G; Pb.out_tie = stie + (size_t)g0 * G;
                        Pb.out_sum = ssum + (size_t)g0 * G; Pb.fallback = fb + g0;
                        hipLaunchKernelGGL(kern, dim3(g1 - g0), dim3(CSCC_NT), lds, s1, Pb);
                        hipEventRecord(evf[b], s1);
                        hipStreamWaitEvent(s2, evf[b], 0);
                        FinalizeParams Fb = F; Fb.in_2u = s2u + (size_t)g0 * G; Fb.in_tie = stie + (size_t)g0 * G; Fb.in_sum = ssum + (size_t)g0 * G;
                        Fb.nb = g1 - g0; Fb.out_p = op + g0; Fb.out_u = ou + g0; Fb.out_fc = ofc + g0;
                        hipLaunchKernelGGL(k_finalize, dim3((g1 - g0 + 31) / 32, (G + 31) / 32), dim3(256), 0, s2, Fb);
                    }
                    hipEventRecord(evf[0], s2); hipStreamWaitEvent(0, evf[0], 0);
                    hipEventRecord(evf[1 % NBATCH], s1); hipStreamWaitEvent(0, evf[1 % NBATCH], 0);
                });
            }
        }
    }
    return 0;
}
