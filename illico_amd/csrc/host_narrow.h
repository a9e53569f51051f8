// Host side of the narrow uploads: a count matrix that lives in host memory travels to the device as BYTES (a quarter of the float32
// bytes over a 50 GB/s link) -- cell = the value itself when it is an integer in [0, 255), else 255, the marker the fused kernels
// already know from k_csr_densify's byte windows ("this gene is not for me").  The conversion runs on the threads that fill the pinned
// staging slots (dense_driver.h: host_windows_pipeline_narrow); float32 rows go through AVX2 when the CPU has it (16 cells per step).
#pragma once
#include <cstdint>
#include <cstring>
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
