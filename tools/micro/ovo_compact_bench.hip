// Microbenchmark + cross-check of the packed dense OVO route (kernels_ovo_compact.h) against the transpose + k_ovo_rank route
// (kernels_ovo.h) on a C2-shaped synthetic matrix built on the device: 300k cells x M genes, 2000 groups (reference = 10 000
// cells), continuous values log1p(count * U(0.5, 1.5)) with half the entries zeroed (mode 0) or plain counts (mode 1: ties).
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -I illico_amd/csrc -o tools/micro/ovo_compact_bench tools/micro/ovo_compact_bench.hip
// Run:   tools/micro/ovo_compact_bench [genes] [mode] [nbk_lg] [key slots] [eq buckets]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <cmath>
#include <vector>
#include <algorithm>
#include <numeric>
#include "common.h"
#include "kernels_ovo.h"
#include "kernels_ovo_compact.h"

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

__device__ __forceinline__ u32 mix(u32 a, u32 b) {
    u32 h = a * 0x9E3779B1u ^ (b + 0x7F4A7C15u) * 0x85EBCA6Bu;
    h ^= h >> 15; h *= 0x2C1B3C6Du; h ^= h >> 12; h *= 0x297A2D39u; h ^= h >> 15;
    return h;
}
__global__ void k_fill(float *X, int N, int M, int mode) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (long long)N * M) return;
    const int r = (int)(i / M), c = (int)(i % M);
    const u32 h1 = mix(r, c), h2 = mix(c * 7919u + 1u, r), h3 = mix(r * 31u + 7u, c + 12345u);
    const float mean = 0.1f + 14.9f * (float)(mix(c, 0xABCDu) >> 8) / 16777216.0f;
    // Poisson(mean) by multiplication of uniforms (Knuth), as bench.py's torch.poisson
    const float L = expf(-mean);
    float pr = 1.0f, cnt = -1.0f;
    u32 hs = h1;
    do {
        hs = mix(hs, 0x51ED27u + (u32)cnt);
        pr *= ((float)(hs >> 8) + 0.5f) / 16777216.0f;
        cnt += 1.0f;
    } while (pr > L);
    const bool keep = (h2 & 1u) != 0;
    const float u = 0.5f + (float)(h3 >> 8) / 16777216.0f;
    float v = mode == 0 ? log1pf(cnt * u) : cnt;
    if (mode == 2 && (h3 & 7u) == 0) v = -v; // some negative values
    X[i] = keep ? v : 0.0f;
}

static size_t old_lds(int ref_cap, int nt) {
    size_t nw = nt / 64;
    size_t b = ((((size_t)ref_cap + 4) * 4) + 15) & ~(size_t)15;
    b += ovo_runend_bytes(ref_cap, true);
    b += nw * 256 * 4 + nw * 256 * 4;
    b += nw * 8 * 2 + 16 + 48;
    return b;
}

int main(int argc, char **argv) {
    const int N = 300000, G = 2000, n_ref = 10000;
    const int M = argc > 1 ? atoi(argv[1]) : 2048;
    const int mode = argc > 2 ? atoi(argv[2]) : 0;
    const int nbk_lg = argc > 3 ? atoi(argv[3]) : 17;
    const int ref_cap_arg = argc > 4 ? atoi(argv[4]) : 0; // key slots of the packed rank kernel (0: one per reference cell)
    const int eq_arg = argc > 5 ? atoi(argv[5]) : 0;      // 1: distribution-following bucket function
    float *X; CK(hipMalloc(&X, (size_t)N * M * 4));
    k_fill<<<(unsigned)(((long long)N * M + 255) / 256), 256>>>(X, N, M, mode);
    // groups
    std::vector<int> codes(N), counts(G, 0), pos(G + 1, 0), perm(N);
    srand(1);
    for (int i = 0; i < N; ++i) codes[i] = i < n_ref ? 0 : 1 + rand() % (G - 1);
    std::random_shuffle(codes.begin(), codes.end());
    for (int i = 0; i < N; ++i) counts[codes[i]]++;
    for (int g = 0; g < G; ++g) pos[g + 1] = pos[g] + counts[g];
    { std::vector<int> cur(pos.begin(), pos.end() - 1); for (int i = 0; i < N; ++i) perm[cur[codes[i]]++] = i; }
    const int max_nonref = *std::max_element(counts.begin() + 1, counts.end());
    printf("N %d M %d G %d n_ref %d max other group %d mode %d nbk_lg %d\n", N, M, G, counts[0], max_nonref, mode, nbk_lg);
    int *d_perm, *d_pos, *d_counts;
    CK(hipMalloc(&d_perm, N * 4)); CK(hipMalloc(&d_pos, (G + 1) * 4)); CK(hipMalloc(&d_counts, G * 4));
    CK(hipMemcpy(d_perm, perm.data(), N * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(d_pos, pos.data(), (G + 1) * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(d_counts, counts.data(), G * 4, hipMemcpyHostToDevice));
    const long long stride = ((N + 63) & ~63ll) + 64 * 400;
    u32 *Xt; CK(hipMalloc(&Xt, (size_t)M * stride * 4));
    long long *s2u[2]; u64 *stie[2]; double *ssum[2];
    for (int v = 0; v < 2; ++v) {
        CK(hipMalloc(&s2u[v], (size_t)M * G * 8)); CK(hipMalloc(&stie[v], (size_t)M * G * 8)); CK(hipMalloc(&ssum[v], (size_t)M * G * 8));
        CK(hipMemset(s2u[v], 0xEE, (size_t)M * G * 8)); CK(hipMemset(stie[v], 0xEE, (size_t)M * G * 8)); CK(hipMemset(ssum[v], 0xEE, (size_t)M * G * 8));
    }
    u16 *nnz; CK(hipMalloc(&nnz, (size_t)M * G * 2));
    CK(hipDeviceSynchronize());
    hipEvent_t e0, e1, e2; hipEventCreate(&e0); hipEventCreate(&e1); hipEventCreate(&e2);

    // ---- old route ----
    OvoParams P;
    P.Xs = Xt; P.gene_stride = stride; P.pos_ptr = d_pos; P.seg_ptr = nullptr; P.counts = d_counts; P.G = G; P.ref = 0; P.n_genes = M; P.dt = DT_F32; P.is_log1p = 0;
    P.ref_cap = n_ref; P.ref_buckets = 1; P.out_2u = s2u[0]; P.out_tie = stie[0]; P.out_sum = ssum[0];
    const size_t lds_old = old_lds(n_ref, 512);
    auto kold = k_ovo_rank<u32, 4, true, 512, false>;
    CK(hipFuncSetAttribute((const void *)kold, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_old));
    auto run_old = [&]() {
        hipEventRecord(e0);
        dim3 grid((N + 63) / 64, (M + 63) / 64);
        hipLaunchKernelGGL((k_transpose_permute_vec<float, u32, 4>), grid, dim3(256), 0, 0, X, (long long)M, 0ll, M, d_perm, N, Xt, stride, (u32 *)nullptr, 0);
        hipEventRecord(e1);
        hipLaunchKernelGGL(kold, dim3(M), dim3(512), lds_old, 0, P, (const u32 *)nullptr);
        hipEventRecord(e2);
        CK(hipEventSynchronize(e2));
        float a, b; hipEventElapsedTime(&a, e0, e1); hipEventElapsedTime(&b, e1, e2);
        printf("old: transpose %.3f ms  k_ovo_rank %.3f ms   (x %.2f for 8000 genes: %.2f + %.2f)\n", a, b, 8000.0 / M, a * 8000.0 / M, b * 8000.0 / M);
    };
    if (max_nonref <= 256) { run_old(); run_old(); }

    // ---- packed route ----
    const int nseg = gcmp_ref_segments(n_ref);
    u16 *seg_nnz; double *seg_sum; CK(hipMalloc(&seg_nnz, (size_t)M * nseg * 2)); CK(hipMalloc(&seg_sum, (size_t)M * nseg * 8));
    u32 *route; CK(hipMalloc(&route, M * 4)); CK(hipMemset(route, 0, M * 4));
    // blocks of the packed layout (as illico_set_groups builds them)
    std::vector<int> bg0, bg1, bout;
    long long ppos = 0, prow = 0; bool open = false;
    auto close = [&](int end) { bg1.push_back(end); ppos += (prow + 63) & ~63ll; open = false; };
    for (int g = 0; g < G; ++g) {
        if (g == 0) { if (open) close(g); continue; }
        if (!open) { bg0.push_back(g); bout.push_back((int)ppos); prow = 0; open = true; }
        prow += counts[g];
        if (prow >= GCMP_BLOCK_ROWS) close(g + 1);
    }
    if (open) close(G);
    const int nblk = (int)bg0.size(), ref_out = (int)ppos;
    const long long pk_stride = ppos + ((n_ref + 63) & ~63) + 64;
    if (pk_stride > stride) { printf("packed stride %lld > %lld\n", pk_stride, stride); return 1; }
    std::vector<int> pk; pk.insert(pk.end(), bg0.begin(), bg0.end()); pk.insert(pk.end(), bg1.begin(), bg1.end()); pk.insert(pk.end(), bout.begin(), bout.end());
    int *d_pk; CK(hipMalloc(&d_pk, pk.size() * 4)); CK(hipMemcpy(d_pk, pk.data(), pk.size() * 4, hipMemcpyHostToDevice));
    u32 *gofs; CK(hipMalloc(&gofs, (size_t)M * G * 4));
    printf("packed layout: %d blocks, stride %lld (dense %lld)\n", nblk, pk_stride, stride);
    GroupCompactParams Q;
    Q.X = X; Q.ld = M; Q.col0 = 0; Q.ncols = M; Q.perm = d_perm; Q.pos_ptr = d_pos; Q.G = G; Q.ref = 0; Q.nseg = nseg; Q.blk_g0 = d_pk; Q.blk_g1 = d_pk + nblk; Q.blk_out = d_pk + 2 * nblk; Q.nblk = nblk; Q.ref_out = ref_out; Q.gofs = gofs; Q.blk_cnt = nullptr; Q.seg_nnz = seg_nnz; Q.seg_sum = seg_sum; Q.Xt = Xt; Q.xt_stride = stride; Q.nnz = nnz; Q.out_sum = ssum[1];
    OvoCompactParams C;
    C.Xs = Xt; C.gene_stride = stride; C.counts = d_counts; C.nnz = nnz; C.gofs = gofs; C.ref_out = ref_out; C.seg_nnz = seg_nnz; C.seg_sum = seg_sum; C.out_sum = ssum[1]; C.nseg = nseg; C.G = G; C.ref = 0; C.n_genes = M; C.ref_cap = ref_cap_arg > 0 ? ref_cap_arg : n_ref; C.nbk_lg = nbk_lg;
    C.out_2u = s2u[1]; C.out_tie = stie[1]; C.route = route;
    const size_t lds_new = ocr_lds_bytes(C.ref_cap, nbk_lg, 4);
    auto knew = eq_arg ? k_ovo_rank_compact<u32, true> : k_ovo_rank_compact<u32, false>;
    CK(hipFuncSetAttribute((const void *)knew, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_new));
    printf("LDS: old %zu B, packed %zu B\n", lds_old, lds_new);
    auto run_new = [&]() {
        hipEventRecord(e0);
        hipLaunchKernelGGL((k_group_compact<float, u32, true, false>), dim3(((nseg + 7) & ~7) + nblk, (M + 63) / 64), dim3(GCMP_NT), 0, 0, Q);
        hipEventRecord(e1);
        hipMemsetAsync(route, 0, M * 4, 0);
        hipLaunchKernelGGL(knew, dim3(M), dim3(OCR_NT), lds_new, 0, C);
        {
            OvoParams P2 = P;
            P2.out_2u = s2u[1]; P2.out_tie = stie[1]; P2.out_sum = nullptr; P2.nnz = nnz; P2.gofs = gofs; P2.only = route;
            hipLaunchKernelGGL(kold, dim3(M), dim3(512), lds_old, 0, P2, (const u32 *)nullptr);
        }
        hipEventRecord(e2);
        CK(hipEventSynchronize(e2));
        CK(hipGetLastError());
        float a, b; hipEventElapsedTime(&a, e0, e1); hipEventElapsedTime(&b, e1, e2);
        printf("new: compact %.3f ms  k_ovo_rank_compact %.3f ms   (x %.2f for 8000 genes: %.2f + %.2f)\n", a, b, 8000.0 / M, a * 8000.0 / M, b * 8000.0 / M);
    };
    run_new(); run_new(); run_new();

    // ---- compare ----
    if (max_nonref <= 256) {
        const size_t cnt = (size_t)M * G;
        std::vector<long long> a(cnt), b(cnt);
        CK(hipMemcpy(a.data(), s2u[0], cnt * 8, hipMemcpyDeviceToHost)); CK(hipMemcpy(b.data(), s2u[1], cnt * 8, hipMemcpyDeviceToHost));
        size_t bad = 0, first = cnt;
        for (size_t i = 0; i < cnt; ++i) if (a[i] != b[i]) { if (!bad) first = i; ++bad; }
        printf("2U mismatches: %zu of %zu", bad, cnt);
        if (bad) printf("  first at gene %zu group %zu: old %lld new %lld", first / G, first % G, a[first], b[first]);
        printf("\n");
        CK(hipMemcpy(a.data(), stie[0], cnt * 8, hipMemcpyDeviceToHost)); CK(hipMemcpy(b.data(), stie[1], cnt * 8, hipMemcpyDeviceToHost));
        bad = 0;
        size_t nzt = 0;
        for (size_t i = 0; i < cnt; ++i) { if (a[i] != b[i]) { if (!bad) first = i; ++bad; } }
        printf("tie mismatches: %zu of %zu", bad, cnt);
        if (bad) printf("  first at gene %zu group %zu: old %lld new %lld", first / G, first % G, a[first], b[first]);
        printf("\n");
        std::vector<double> sa(cnt), sb(cnt);
        CK(hipMemcpy(sa.data(), ssum[0], cnt * 8, hipMemcpyDeviceToHost)); CK(hipMemcpy(sb.data(), ssum[1], cnt * 8, hipMemcpyDeviceToHost));
        double worst = 0;
        for (size_t i = 0; i < cnt; ++i) { const double d = fabs(sa[i] - sb[i]) / (fabs(sa[i]) + 1e-300); if (d > worst && sa[i] != 0) worst = d; if ((sa[i] == 0) != (sb[i] == 0)) worst = 1; }
        printf("value sums: worst relative difference %.3g\n", worst);
        (void)nzt;
    }
    return 0;
}
