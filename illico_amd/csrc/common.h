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

// ---------------------------------------------------------------------------------------------
// VALU-only lane exchanges (DPP row permutes + v_permlane16/32_swap): none of these touch the LDS
// crossbar, which ds_bpermute / __shfl do.  dpp_ctrl encodings: quad_perm = a|b<<2|c<<4|d<<6,
// row_shl:n = 0x100+n, row_shr:n = 0x110+n, row_ror:n = 0x120+n, wave_shl:1 = 0x130,
// wave_shr:1 = 0x138, row_mirror = 0x140, row_half_mirror = 0x141, row_bcast:15 = 0x142,
// row_bcast:31 = 0x143.
// ---------------------------------------------------------------------------------------------
template <int CTRL, int ROW_MASK = 0xF, int BANK_MASK = 0xF, bool BOUND = false>
__device__ __forceinline__ u32 dpp_u32(u32 old, u32 src) {
    return (u32)__builtin_amdgcn_update_dpp((int)old, (int)src, CTRL, ROW_MASK, BANK_MASK, BOUND);
}
// full permutation inside a row: every lane has a source, so no `old` value is needed (lets the
// compiler fold the move into the consuming VALU instruction's DPP operand)
template <int CTRL> __device__ __forceinline__ u32 dpp_perm(u32 src) {
    return (u32)__builtin_amdgcn_mov_dpp((int)src, CTRL, 0xF, 0xF, true);
}

// value of lane (lane ^ M) for the masks the sort network uses: 1,2,4,8,16,32 and 3,7,15,31,63
template <int M> __device__ __forceinline__ u32 xor_lanes(u32 v, int lane) {
    constexpr int L = M & 15;
    static_assert(L == 0 || L == 1 || L == 2 || L == 3 || L == 4 || L == 7 || L == 8 || L == 15, "unsupported lane mask");
    u32 t = v;
    if constexpr (L == 1) t = dpp_perm<0xB1>(v);        // quad_perm [1,0,3,2]
    else if constexpr (L == 2) t = dpp_perm<0x4E>(v);   // quad_perm [2,3,0,1]
    else if constexpr (L == 3) t = dpp_perm<0x1B>(v);   // quad_perm [3,2,1,0]
    else if constexpr (L == 7) t = dpp_perm<0x141>(v);  // row_half_mirror
    else if constexpr (L == 15) t = dpp_perm<0x140>(v); // row_mirror
    else if constexpr (L == 8) t = dpp_perm<0x128>(v);  // row_ror:8
    else if constexpr (L == 4) {
        t = dpp_u32<0x104, 0xF, 0x5>(v, v);               // banks 0,2 <- lane+4
        t = dpp_u32<0x114, 0xF, 0xA>(t, v);               // banks 1,3 <- lane-4
    }
    if constexpr ((M & 16) != 0) {
        auto r = __builtin_amdgcn_permlane16_swap(t, t, false, false);
        t = (lane & 16) ? (u32)r[0] : (u32)r[1];
    }
    if constexpr ((M & 32) != 0) {
        auto r = __builtin_amdgcn_permlane32_swap(t, t, false, false);
        t = (lane & 32) ? (u32)r[0] : (u32)r[1];
    }
    return t;
}
template <int M> __device__ __forceinline__ u64 xor_lanes(u64 v, int lane) {
    u32 lo = xor_lanes<M>((u32)v, lane), hi = xor_lanes<M>((u32)(v >> 32), lane);
    return ((u64)hi << 32) | lo;
}
template <int M> __device__ __forceinline__ int xor_lanes(int v, int lane) { return (int)xor_lanes<M>((u32)v, lane); }
template <int M> __device__ __forceinline__ double xor_lanes(double v, int lane) {
    return __longlong_as_double((long long)xor_lanes<M>((u64)__double_as_longlong(v), lane));
}

// lane l <- lane l-1 (lane 0 keeps `fill`), lane l <- lane l+1 (lane 63 keeps `fill`)
__device__ __forceinline__ u32 wave_shr1(u32 v, u32 fill) { return dpp_u32<0x138>(fill, v); }
__device__ __forceinline__ u32 wave_shl1(u32 v, u32 fill) { return dpp_u32<0x130>(fill, v); }
__device__ __forceinline__ u64 wave_shr1(u64 v, u64 fill) {
    return ((u64)wave_shr1((u32)(v >> 32), (u32)(fill >> 32)) << 32) | wave_shr1((u32)v, (u32)fill);
}
__device__ __forceinline__ u64 wave_shl1(u64 v, u64 fill) {
    return ((u64)wave_shl1((u32)(v >> 32), (u32)(fill >> 32)) << 32) | wave_shl1((u32)v, (u32)fill);
}

// inclusive scans over the 64 lanes (row_shr 1,2,4,8 then row_bcast 15 / 31)
__device__ __forceinline__ int wave_incl_scan_add(int x) {
    x += (int)dpp_u32<0x111>(0u, (u32)x);
    x += (int)dpp_u32<0x112>(0u, (u32)x);
    x += (int)dpp_u32<0x114>(0u, (u32)x);
    x += (int)dpp_u32<0x118>(0u, (u32)x);
    x += (int)dpp_u32<0x142, 0xA>(0u, (u32)x);
    x += (int)dpp_u32<0x143, 0xC>(0u, (u32)x);
    return x;
}
__device__ __forceinline__ int wave_incl_scan_max(int x) { // identity: INT_MIN
    const u32 id = 0x80000000u;
    x = max(x, (int)dpp_u32<0x111>(id, (u32)x));
    x = max(x, (int)dpp_u32<0x112>(id, (u32)x));
    x = max(x, (int)dpp_u32<0x114>(id, (u32)x));
    x = max(x, (int)dpp_u32<0x118>(id, (u32)x));
    x = max(x, (int)dpp_u32<0x142, 0xA>(id, (u32)x));
    x = max(x, (int)dpp_u32<0x143, 0xC>(id, (u32)x));
    return x;
}

// Sort 64*K keys held K per lane; sorted position of (lane, r) is lane*K + r.  All-ascending
// ("flip") bitonic network: intra-lane stages are plain register compare-exchanges, cross-lane
// stages exchange with lane^mask through DPP / permlane swaps.  21*K cross-lane exchanges.
template <typename KeyT, int K, int SIZE> __device__ __forceinline__ void wave_bitonic_merge(KeyT (&v)[K], int lane) {
    constexpr int lm = SIZE / K - 1;
    const bool keep_min = (lane & (SIZE / K / 2)) == 0;
    KeyT pv[K];
#pragma unroll
    for (int r = 0; r < K; ++r) pv[r] = xor_lanes<lm>(v[K - 1 - r], lane);
#pragma unroll
    for (int r = 0; r < K; ++r) v[r] = keep_min ? umin_t(v[r], pv[r]) : umax_t(v[r], pv[r]);
    if constexpr (SIZE / K / 4 >= 1) {
        constexpr int ls0 = SIZE / K / 4;
#define ILLICO_HALF_CLEAN(LS)                                                   \
    if constexpr (ls0 >= (LS)) {                                                \
        const bool km = (lane & (LS)) == 0;                                     \
        _Pragma("unroll") for (int r = 0; r < K; ++r) {                         \
            KeyT q = xor_lanes<(LS)>(v[r], lane);                               \
            v[r] = km ? umin_t(v[r], q) : umax_t(v[r], q);                      \
        }                                                                       \
    }
        ILLICO_HALF_CLEAN(16)
        ILLICO_HALF_CLEAN(8)
        ILLICO_HALF_CLEAN(4)
        ILLICO_HALF_CLEAN(2)
        ILLICO_HALF_CLEAN(1)
#undef ILLICO_HALF_CLEAN
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
    wave_bitonic_merge<KeyT, K, 2 * K>(v, lane);
    wave_bitonic_merge<KeyT, K, 4 * K>(v, lane);
    wave_bitonic_merge<KeyT, K, 8 * K>(v, lane);
    wave_bitonic_merge<KeyT, K, 16 * K>(v, lane);
    wave_bitonic_merge<KeyT, K, 32 * K>(v, lane);
    wave_bitonic_merge<KeyT, K, 64 * K>(v, lane);
}

// Transpose-reduce: 64 per-lane partial vectors x_0..x_63 (pushed one at a time, x_j's lanes are the
// partials of item j) -> lane j ends up holding sum_l x_j[l].  A binary counter of half-combined
// registers: 63 combines in total instead of 64 six-step butterflies.
template <typename T, int LVL> __device__ __forceinline__ T tr_combine(T a, T b, int lane) {
    // a: earlier items, b: later items; lanes with bit LVL clear keep a's side
    const bool hi = (lane >> LVL) & 1;
    T keep = hi ? b : a, send = hi ? a : b;
    return keep + xor_lanes<(1 << LVL)>(send, lane);
}
template <typename T> struct TrReduce {
    T acc[6];
    T result;
    template <int LVL> __device__ __forceinline__ void push_lvl(T x, int j, int lane) {
        if constexpr (LVL == 6) { result = x; }
        else {
            if (j & (1 << LVL)) push_lvl<LVL + 1>(tr_combine<T, LVL>(acc[LVL], x, lane), j, lane);
            else acc[LVL] = x;
        }
    }
    __device__ __forceinline__ void push(T x, int j, int lane) { push_lvl<0>(x, j, lane); }
};

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

// Half-cleaner cascade of the same network over one 64*K chunk held K per lane (position lane*K + r): element
// strides 32K .. K across lanes (DPP / permlane swaps), then K/2 .. 1 inside the lane.  Turns a chunk whose two halves
// come out of a larger merge stage into a sorted chunk.
template <typename KeyT, int K> __device__ __forceinline__ void wave_half_clean_chunk(KeyT (&v)[K], int lane) {
#define ILLICO_HC(LS)                                                           \
    {                                                                           \
        const bool km = (lane & (LS)) == 0;                                     \
        _Pragma("unroll") for (int r = 0; r < K; ++r) {                         \
            KeyT q = xor_lanes<(LS)>(v[r], lane);                               \
            v[r] = km ? umin_t(v[r], q) : umax_t(v[r], q);                      \
        }                                                                       \
    }
    ILLICO_HC(32)
    ILLICO_HC(16)
    ILLICO_HC(8)
    ILLICO_HC(4)
    ILLICO_HC(2)
    ILLICO_HC(1)
#undef ILLICO_HC
#pragma unroll
    for (int stride = K >> 1; stride > 0; stride >>= 1) {
#pragma unroll
        for (int r = 0; r < K; ++r) {
            int p = r ^ stride;
            if (p > r) compex(v[r], v[p]);
        }
    }
}

// Sort the first `ncap` keys of A (LDS; ncap a multiple of 64*K, slots past the data filled with the largest key) with
// the whole workgroup: 64*K-key chunks are sorted / merged in registers by one wavefront each, only the stages whose
// stride reaches across chunks go through LDS.  For 32768 keys and K = 16: 21 passes over LDS instead of the 120 of
// block_bitonic_sort.  Slots >= ncap are virtual +inf (same all-ascending network, so they never move).
template <typename KeyT, int NT, int K> __device__ __forceinline__ void block_sort_hybrid(KeyT *A, int ncap, int tid) {
    constexpr int CH = 64 * K, NW = NT / 64;
    const int lane = tid & 63, wave = tid >> 6;
    const int nch = ncap / CH;
    for (int c = wave; c < nch; c += NW) {
        KeyT v[K];
        KeyT *p = A + c * CH + lane * K;
#pragma unroll
        for (int r = 0; r < K; ++r) v[r] = p[r];
        wave_bitonic_sort<KeyT, K>(v, lane);
#pragma unroll
        for (int r = 0; r < K; ++r) p[r] = v[r];
    }
    __syncthreads();
    int P = CH;
    while (P < ncap) P <<= 1;
    const int halfP = P >> 1;
    for (int size = 2 * CH; size <= P; size <<= 1) {
        const int half = size >> 1;
        for (int t = tid; t < halfP; t += NT) {
            const int blk = t / half, off = t & (half - 1);
            const int i = blk * size + off, j = blk * size + size - 1 - off;
            if (j < ncap) {
                KeyT a = A[i], b = A[j];
                if (a > b) { A[i] = b; A[j] = a; }
            }
        }
        __syncthreads();
        for (int stride = half >> 1; stride >= CH; stride >>= 1) {
            for (int t = tid; t < halfP; t += NT) {
                const int i = (t / stride) * 2 * stride + (t & (stride - 1)), j = i + stride;
                if (j < ncap) {
                    KeyT a = A[i], b = A[j];
                    if (a > b) { A[i] = b; A[j] = a; }
                }
            }
            __syncthreads();
        }
        for (int c = wave; c < nch; c += NW) {
            KeyT v[K];
            KeyT *p = A + c * CH + lane * K;
#pragma unroll
            for (int r = 0; r < K; ++r) v[r] = p[r];
            wave_half_clean_chunk<KeyT, K>(v, lane);
#pragma unroll
            for (int r = 0; r < K; ++r) p[r] = v[r];
        }
        __syncthreads();
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
