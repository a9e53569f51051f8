// Feasibility probe: rocPRIM segmented radix sort of 1024 segments x 150k (u32 key, u32 value) pairs.
#include <cstring>
#include <hip/hip_runtime.h>
#include <rocprim/rocprim.hpp>
#include <cstdio>
#include <vector>
#include <random>
int main() {
    const int S = 1024; const size_t L = 150000, n = S * L;
    std::vector<unsigned> hk(n), hv(n); std::vector<unsigned> off(S + 1);
    std::mt19937 rng(1);
    for (size_t i = 0; i < n; ++i) { float f = (float)(rng() % 100000) / 7919.0f + 0.01f; unsigned u; memcpy(&u, &f, 4); hk[i] = u | 0x80000000u; hv[i] = rng() % 2000; }
    for (int s = 0; s <= S; ++s) off[s] = (unsigned)(s * L);
    unsigned *ka, *kb, *va, *vb, *d_off;
    (void)hipMalloc(&ka, n * 4); (void)hipMalloc(&kb, n * 4); (void)hipMalloc(&va, n * 4); (void)hipMalloc(&vb, n * 4); (void)hipMalloc(&d_off, (S + 1) * 4);
    (void)hipMemcpy(ka, hk.data(), n * 4, hipMemcpyHostToDevice); (void)hipMemcpy(va, hv.data(), n * 4, hipMemcpyHostToDevice);
    (void)hipMemcpy(d_off, off.data(), (S + 1) * 4, hipMemcpyHostToDevice);
    size_t tmp_bytes = 0; void *tmp = nullptr;
    (void)rocprim::segmented_radix_sort_pairs(nullptr, tmp_bytes, ka, kb, va, vb, (unsigned)n, S, d_off, d_off + 1, 0, 32);
    (void)hipMalloc(&tmp, tmp_bytes);
    hipEvent_t a, b; (void)hipEventCreate(&a); (void)hipEventCreate(&b);
    for (int rep = 0; rep < 3; ++rep) {
        (void)hipEventRecord(a);
        (void)rocprim::segmented_radix_sort_pairs(tmp, tmp_bytes, ka, kb, va, vb, (unsigned)n, S, d_off, d_off + 1, 0, 32);
        (void)hipEventRecord(b); (void)hipEventSynchronize(b);
        float ms; (void)hipEventElapsedTime(&ms, a, b);
        printf("segmented_radix_sort_pairs: %d segments x %zu pairs: %.3f ms (%.2f Gpairs/s), temp %zu MB\n", S, L, ms, n / ms / 1e6, tmp_bytes >> 20);
    }
    std::vector<unsigned> out(L); (void)hipMemcpy(out.data(), kb + 5 * L, L * 4, hipMemcpyDeviceToHost);
    bool ok = true; for (size_t i = 1; i < L; ++i) ok &= out[i - 1] <= out[i];
    printf("segment 5 sorted: %d\n", (int)ok);
    return 0;
}
