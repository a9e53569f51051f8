// Host side of the narrow uploads: a count matrix that lives in host memory travels to the device as BYTES (a quarter of the float32
// bytes over a 50 GB/s link) -- cell = the value itself when it is an integer in [0, 255), else 255, the marker the fused kernels
// already know from k_csr_densify's byte windows ("this gene is not for me").  The conversion runs on the threads that fill the pinned
// staging slots (dense_driver.h: host_windows_pipeline_narrow); float32 rows go through AVX2 when the CPU has it (16 cells per step).
#pragma once
#include <cstdint>
#include <cstdio>
#include <cstring>
#if defined(__linux__)
#include <sched.h>
#include <sys/syscall.h>
#include <unistd.h>
#endif

// ---- where the caller's matrix lives ---------------------------------------------------------------------------------------------
// The threads that fill the pinned slots read the whole matrix once, in row pieces a few KB long.  On a two-socket host those reads
// cost twice as much from the other socket (C2 shape, byte windows, 16 threads: 60 ms with the threads on the matrix's NUMA node,
// 131 - 158 ms on the other one, 84 ms wherever the scheduler puts them; profiles/NOTES_r05.md).  The pages' node is asked of the kernel
// (move_pages in query mode: sixteen pages spread over the buffer); when three quarters of them sit on one node, the producer thread --
// and with it the fill threads it starts -- is confined to that node's CPUs (those of them the process may use at all).
static inline int numa_node_of_buffer(const void *p, size_t bytes) {
#if defined(__linux__) && defined(SYS_move_pages)
    const long page = sysconf(_SC_PAGESIZE);
    if (page <= 0 || bytes < (size_t)page * 64) return -1;
    void *pages[16];
    int status[16];
    for (int i = 0; i < 16; ++i) {
        const uintptr_t a = (uintptr_t)p + (uintptr_t)((double)bytes * ((double)i + 0.5) / 16.0);
        pages[i] = (void *)(a & ~(uintptr_t)(page - 1));
        status[i] = -1;
    }
    if (syscall(SYS_move_pages, 0, 16ul, pages, nullptr, status, 0) != 0) return -1;
    int cnt[64] = {0}, best = 0;
    for (int i = 0; i < 16; ++i)
        if (status[i] >= 0 && status[i] < 64) ++cnt[status[i]];
    for (int n = 1; n < 64; ++n)
        if (cnt[n] > cnt[best]) best = n;
    return cnt[best] * 4 >= 16 * 3 ? best : -1;
#else
    (void)p; (void)bytes;
    return -1;
#endif
}
// confines the CALLING thread (and the threads it starts afterwards) to the CPUs of `node` that its present mask allows; false: nothing changed
static inline bool numa_confine_this_thread(int node) {
#if defined(__linux__)
    if (node < 0) return false;
    char path[96], buf[4096];
    snprintf(path, sizeof path, "/sys/devices/system/node/node%d/cpulist", node);
    FILE *f = fopen(path, "r");
    if (!f) return false;
    const size_t got = fread(buf, 1, sizeof buf - 1, f);
    fclose(f);
    buf[got] = 0;
    cpu_set_t now, want;
    CPU_ZERO(&want);
    if (sched_getaffinity(0, sizeof now, &now) != 0) return false;
    int any = 0;
    for (const char *q = buf; *q;) { // "0-63,128-191"
        char *e;
        const long lo = strtol(q, &e, 10);
        if (e == q) break;
        long hi = lo;
        if (*e == '-') { q = e + 1; hi = strtol(q, &e, 10); }
        for (long cpu = lo; cpu <= hi && cpu < CPU_SETSIZE; ++cpu)
            if (cpu >= 0 && CPU_ISSET((int)cpu, &now)) { CPU_SET((int)cpu, &want); ++any; }
        q = *e == ',' ? e + 1 : e;
        if (*e != ',' ) break;
    }
    if (any < 4) return false; // (a handful of CPUs would starve the fill)
    return sched_setaffinity(0, sizeof want, &want) == 0;
#else
    (void)node;
    return false;
#endif
}
#if defined(__x86_64__)
#include <immintrin.h>
#endif

template <typename InT> static inline uint8_t narrow_cell(InT v);
template <> inline uint8_t narrow_cell<float>(float v) { const int c = (v >= 0.0f && v < 255.0f) ? (int)v : 255; return (float)c == v ? (uint8_t)c : (uint8_t)255; }
template <> inline uint8_t narrow_cell<double>(double v) { const int c = (v >= 0.0 && v < 255.0) ? (int)v : 255; return (double)c == v ? (uint8_t)c : (uint8_t)255; }
template <> inline uint8_t narrow_cell<int32_t>(int32_t v) { return (v >= 0 && v < 255) ? (uint8_t)v : (uint8_t)255; }
template <> inline uint8_t narrow_cell<int64_t>(int64_t v) { return (v >= 0 && v < 255) ? (uint8_t)v : (uint8_t)255; }

template <typename InT> static inline void narrow_cells_plain(const InT *src, uint8_t *dst, int64_t n) {
    for (int64_t i = 0; i < n; ++i) dst[i] = narrow_cell<InT>(src[i]);
}

#if defined(__x86_64__)
__attribute__((target("avx2"))) static inline __m256i narrow_cells8_avx2(const float *p) { // eight cells as 32-bit lanes, each 0 .. 255
    const __m256i c254 = _mm256_set1_epi32(254), c255 = _mm256_set1_epi32(255);
    const __m256 v = _mm256_loadu_ps(p);
    const __m256i t = _mm256_cvttps_epi32(v); // (out of range / NaN: 0x80000000)
    const __m256 exact = _mm256_cmp_ps(_mm256_cvtepi32_ps(t), v, _CMP_EQ_OQ);
    const __m256i ok = _mm256_and_si256(_mm256_castps_si256(exact), _mm256_cmpeq_epi32(_mm256_min_epu32(t, c254), t));
    return _mm256_blendv_epi8(c255, t, ok);
}
__attribute__((target("avx2"))) static inline void narrow_cells_f32_avx2(const float *src, uint8_t *dst, int64_t n) {
    int64_t i = 0;
    for (; i + 16 <= n; i += 16) {
        const __m256i a = narrow_cells8_avx2(src + i), b = narrow_cells8_avx2(src + i + 8);
        const __m256i w = _mm256_permute4x64_epi64(_mm256_packus_epi32(a, b), 0xD8);                       // 16 x u16, in order
        const __m256i q = _mm256_permute4x64_epi64(_mm256_packus_epi16(w, _mm256_setzero_si256()), 0x08); // 16 x u8 in the low half
        _mm_storeu_si128((__m128i *)(dst + i), _mm256_castsi256_si128(q));
    }
    for (; i < n; ++i) dst[i] = narrow_cell<float>(src[i]);
}
static inline bool narrow_have_avx2() { static const bool yes = __builtin_cpu_supports("avx2"); return yes; }
#endif

template <typename InT> static inline void narrow_cells(const InT *src, uint8_t *dst, int64_t n) { narrow_cells_plain<InT>(src, dst, n); }
#if defined(__x86_64__)
template <> inline void narrow_cells<float>(const float *src, uint8_t *dst, int64_t n) {
    if (narrow_have_avx2()) narrow_cells_f32_avx2(src, dst, n);
    else narrow_cells_plain<float>(src, dst, n);
}
#endif
