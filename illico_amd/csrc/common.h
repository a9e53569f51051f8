// Device-side helpers shared by every kernel of libillico_hip (gfx950 / CDNA4, wave64).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef unsigned int u32;
typedef unsigned long long u64;
typedef unsigned short u16;

#define ILLICO_WAVE 64

// dtype codes (include/illico_hip.h)
#define DT_F32 0
#define DT_F64 1
#define DT_I32 2
#define DT_I64 3

// ---------------------------------------------------------------------------------------------
// Order-preserving unsigned keys.  The reference compares values with < and == in X's dtype
// (utils/ranking.py:34,93,100); an unsigned key with the same order and the same equality classes
// lets every sort / search / tie test run on integers.  -0.0 is folded onto +0.0 so that it ties
// with it, as == does.  Zero maps to 0x80..0 for every dtype (ZEROK).
// ---------------------------------------------------------------------------------------------
template <typename KeyT> struct KeyInfo;
template <> struct KeyInfo<u32> {
    static constexpr u32 MAXK = 0xFFFFFFFFu;
    static constexpr u32 ZEROK = 0x80000000u;
};
template <> struct KeyInfo<u64> {
    static constexpr u64 MAXK = 0xFFFFFFFFFFFFFFFFull;
    static constexpr u64 ZEROK = 0x8000000000000000ull;
};

__device__ __forceinline__ u32 key_of(float v) {
    u32 b = __float_as_uint(v);
    if (b == 0x80000000u) b = 0u;
    return (b & 0x80000000u) ? ~b : (b | 0x80000000u);
}
__device__ __forceinline__ u64 key_of(double v) {
    u64 b = (u64)__double_as_longlong(v);
    if (b == 0x8000000000000000ull) b = 0ull;
    return (b & 0x8000000000000000ull) ? ~b : (b | 0x8000000000000000ull);
}
__device__ __forceinline__ u32 key_of(int32_t v) { return (u32)v ^ 0x80000000u; }
__device__ __forceinline__ u64 key_of(int64_t v) { return (u64)v ^ 0x8000000000000000ull; }

__device__ __forceinline__ float f32_of_key(u32 k) {
    u32 b = (k & 0x80000000u) ? (k & 0x7FFFFFFFu) : ~k;
    return __uint_as_float(b);
}
__device__ __forceinline__ double f64_of_key(u64 k) {
    u64 b = (k & 0x8000000000000000ull) ? (k & 0x7FFFFFFFFFFFFFFFull) : ~k;
    return __longlong_as_double((long long)b);
}
// value as float64, the type the reference's fold-change accumulator adds in (utils/math.py:27-39)
__device__ __forceinline__ double key_to_double(u32 k, int dt) {
    return dt == DT_F32 ? (double)f32_of_key(k) : (double)(int32_t)(k ^ 0x80000000u);
}
__device__ __forceinline__ double key_to_double(u64 k, int dt) {
    return dt == DT_F64 ? f64_of_key(k) : (double)(int64_t)(k ^ 0x8000000000000000ull);
}
// expm1 taken in X's dtype before the float64 add (utils/math.py:212); numpy promotes ints to f64
__device__ __forceinline__ double key_to_expm1(u32 k, int dt) {
    return dt == DT_F32 ? (double)expm1f(f32_of_key(k)) : expm1((double)(int32_t)(k ^ 0x80000000u));
}
__device__ __forceinline__ double key_to_expm1(u64 k, int dt) {
    return dt == DT_F64 ? expm1(f64_of_key(k)) : expm1((double)(int64_t)(k ^ 0x8000000000000000ull));
}

// ---------------------------------------------------------------------------------------------
// wave64 cross-lane helpers
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ int lane_id() { return (int)(threadIdx.x & 63); }

template <typename T> __device__ __forceinline__ T wave_sum(T x) {
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) x += __shfl_xor(x, d);
    return x;
}

// LDS traffic of one wave is executed in issue order; this only stops the compiler from moving
// LDS accesses across the point and waits for outstanding LDS operations.
__device__ __forceinline__ void wave_lds_fence() {
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
    __builtin_amdgcn_wave_barrier();
}

template <typename T> __device__ __forceinline__ T umin_t(T a, T b) { return a < b ? a : b; }
template <typename T> __device__ __forceinline__ T umax_t(T a, T b) { return a > b ? a : b; }

template <typename KeyT> __device__ __forceinline__ void compex(KeyT &lo, KeyT &hi) {
    KeyT a = umin_t(lo, hi), b = umax_t(lo, hi);
    lo = a;
    hi = b;
}

// Sort 64*K keys held K per lane; sorted position of (lane, r) is lane*K + r.  All-ascending
// ("flip") bitonic network: intra-lane stages are plain register compare-exchanges, cross-lane
// stages exchange with lane^mask.  21*K cross-lane exchanges for 64*K keys.
template <typename KeyT, int K> __device__ __forceinline__ void wave_bitonic_sort(KeyT (&v)[K], int lane) {
#pragma unroll
    for (int size = 2; size <= K; size <<= 1) {
#pragma unroll
        for (int r = 0; r < K; ++r) {
            int p = r ^ (size - 1);
            if (p > r) compex(v[r], v[p]);
        }
#pragma unroll
        for (int stride = size >> 2; stride > 0; stride >>= 1) {
#pragma unroll
            for (int r = 0; r < K; ++r) {
                int p = r ^ stride;
                if (p > r) compex(v[r], v[p]);
            }
        }
    }
#pragma unroll
    for (int size = 2 * K; size <= 64 * K; size <<= 1) {
        const int lm = size / K - 1;
        const bool keep_min = (lane & (size / K / 2)) == 0;
        KeyT pv[K];
#pragma unroll
        for (int r = 0; r < K; ++r) pv[r] = __shfl_xor(v[K - 1 - r], lm);
#pragma unroll
        for (int r = 0; r < K; ++r) v[r] = keep_min ? umin_t(v[r], pv[r]) : umax_t(v[r], pv[r]);
#pragma unroll
        for (int ls = size / K / 4; ls > 0; ls >>= 1) {
            const bool km = (lane & ls) == 0;
#pragma unroll
            for (int r = 0; r < K; ++r) {
                KeyT q = __shfl_xor(v[r], ls);
                v[r] = km ? umin_t(v[r], q) : umax_t(v[r], q);
            }
        }
#pragma unroll
        for (int stride = K >> 1; stride > 0; stride >>= 1) {
#pragma unroll
            for (int r = 0; r < K; ++r) {
                int p = r ^ stride;
                if (p > r) compex(v[r], v[p]);
            }
        }
    }
}

// Sort n keys in LDS with the whole workgroup (NT threads).  Same all-ascending network; slots
// >= n are virtual +inf, so compare-exchanges whose upper index is >= n are no-ops and n need not
// be a power of two.
template <typename KeyT, int NT> __device__ __forceinline__ void block_bitonic_sort(KeyT *A, int n, int tid) {
    int P = 1;
    while (P < n) P <<= 1;
    const int halfP = P >> 1;
    for (int size = 2; size <= P; size <<= 1) {
        const int half = size >> 1;
        for (int t = tid; t < halfP; t += NT) {
            int blk = t / half, off = t & (half - 1);
            int i = blk * size + off, j = blk * size + size - 1 - off;
            if (j < n) {
                KeyT a = A[i], b = A[j];
                if (a > b) { A[i] = b; A[j] = a; }
            }
        }
        __syncthreads();
        for (int stride = half >> 1; stride > 0; stride >>= 1) {
            for (int t = tid; t < halfP; t += NT) {
                int i = (t / stride) * 2 * stride + (t & (stride - 1)), j = i + stride;
                if (j < n) {
                    KeyT a = A[i], b = A[j];
                    if (a > b) { A[i] = b; A[j] = a; }
                }
            }
            __syncthreads();
        }
    }
}

// number of elements of sorted A[0..n) that are < q  (top = largest power of two <= n, 0 if n == 0)
template <typename KeyT> __device__ __forceinline__ u32 lower_bound_pow2(const KeyT *A, u32 n, u32 top, KeyT q) {
    u32 base = 0;
    for (u32 step = top; step > 0; step >>= 1) {
        u32 idx = base + step;
        if (idx <= n && A[idx - 1] < q) base = idx;
    }
    return base;
}
// number of elements of sorted A[0..n) that are <= q
template <typename KeyT> __device__ __forceinline__ u32 upper_bound_pow2(const KeyT *A, u32 n, u32 top, KeyT q) {
    u32 base = 0;
    for (u32 step = top; step > 0; step >>= 1) {
        u32 idx = base + step;
        if (idx <= n && A[idx - 1] <= q) base = idx;
    }
    return base;
}
__device__ __forceinline__ u32 top_pow2(u32 n) { return n ? (1u << (31 - __clz(n))) : 0u; }
