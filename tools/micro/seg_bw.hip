// Microbenchmark: HBM read rate of the fused kernels' access shape.  A wavefront owns a column slice of SEG bytes
// (64 lanes x 4 / 8 / 16 B) and walks rows with UU row segments in flight; the row pitch is 32000 B (8000 floats).
// Build: hipcc --offload-arch=gfx950 -O3 -o seg_bw seg_bw.hip ; run: ./seg_bw
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
template <int VEC, int UU>
__global__ __launch_bounds__(256) void k_read(const float *__restrict__ X, long long ld, int n_rows, int rows_per_wave, float *out) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int tile = blockIdx.x;                       // column slice
    const long long col = ((long long)tile * 64 + lane) * VEC;
    const int chunk = blockIdx.y * 4 + wave;           // row range of this wavefront
    const int r0 = chunk * rows_per_wave, r1 = min(r0 + rows_per_wave, n_rows);
    float acc = 0.f;
    for (int r = r0; r + UU <= r1; r += UU) {
        float v[UU][VEC];
#pragma unroll
        for (int u = 0; u < UU; ++u) {
            const float *p = X + (long long)(r + u) * ld + col;
            if (VEC == 1) v[u][0] = p[0];
            else if (VEC == 2) { float2 t = *(const float2 *)p; v[u][0] = t.x; v[u][1] = t.y; }
            else { float4 t = *(const float4 *)p; v[u][0] = t.x; v[u][1] = t.y; v[u][2] = t.z; v[u][3] = t.w; }
        }
#pragma unroll
        for (int u = 0; u < UU; ++u)
#pragma unroll
            for (int k = 0; k < VEC; ++k) acc += v[u][k];
    }
    if (acc == 12345.678f) out[0] = acc;
}
template <int VEC, int UU> void run(const float *X, long long ld, int N, int M, float *out, int rows_per_wave) {
    dim3 grid(M / (64 * VEC), (N + rows_per_wave * 4 - 1) / (rows_per_wave * 4));
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    k_read<VEC, UU><<<grid, 256>>>(X, ld, N, rows_per_wave, out);
    hipDeviceSynchronize();
    hipEventRecord(a);
    for (int i = 0; i < 5; ++i) k_read<VEC, UU><<<grid, 256>>>(X, ld, N, rows_per_wave, out);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b); ms /= 5;
    printf("segment %4d B, %2d rows in flight, %5d rows per wavefront: %.3f ms  %.2f TB/s\n", VEC * 256, UU, rows_per_wave, ms, (double)N * M * 4 / ms / 1e9);
}
int main() {
    const int N = 300000, M = 8000; const long long ld = M;
    float *X, *out; hipMalloc(&X, (size_t)N * M * 4); hipMalloc(&out, 4);
    hipMemset(X, 0, (size_t)N * M * 4);
    for (int rpw : {150, 1200}) {
        run<1, 32>(X, ld, N, M - M % 64, out, rpw);
        run<2, 32>(X, ld, N, M - M % 128, out, rpw);
        run<2, 16>(X, ld, N, M - M % 128, out, rpw);
        run<4, 16>(X, ld, N, M - M % 256, out, rpw);
        run<4, 8>(X, ld, N, M - M % 256, out, rpw);
    }
    return 0;
}
