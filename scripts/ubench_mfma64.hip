// Does the fp64 matrix pipe (v_mfma_f64_16x16x4_f64) run concurrently with fp64 VALU on MI355X?
// diagnostic tool.   hipcc --offload-arch=gfx950 -O3
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double d4 __attribute__((ext_vector_type(4)));

template <int NM, int NV>
__global__ __launch_bounds__(256) void k_mix(double* out, int iters, double a, double b) {
    d4 acc[4];
    for (int i = 0; i < 4; ++i) acc[i] = (d4){0.0, 0.0, 0.0, 0.0};
    double v[8];
    for (int k = 0; k < 8; ++k) v[k] = threadIdx.x * 1e-3 + k;
    const double ma = 1.0 + threadIdx.x * 1e-9, mb = 1.0 - threadIdx.x * 1e-9;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int m = 0; m < NM; ++m) acc[m & 3] = __builtin_amdgcn_mfma_f64_16x16x4f64(ma, mb, acc[m & 3], 0, 0, 0);
#pragma unroll
        for (int q = 0; q < NV; ++q) v[q & 7] = fma(v[q & 7], a, b);
    }
    double s = 0;
    for (int i = 0; i < 4; ++i) s += acc[i].x + acc[i].y + acc[i].z + acc[i].w;
    for (int k = 0; k < 8; ++k) s += v[k];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <typename F>
static float time_it(F launch) {
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    launch(); hipDeviceSynchronize();
    hipEventRecord(a); for (int i = 0; i < 5; ++i) launch(); hipEventRecord(b); hipEventSynchronize(b);
    float ms = 0; hipEventElapsedTime(&ms, a, b); return ms / 5;
}

int main() {
    double* out; hipMalloc(&out, 256 * 8 * 256 * sizeof(double));
    const int iters = 4000;
    for (int bpc : {1, 2, 4}) {
        const int grid = 256 * bpc;
        const double waves = (double)grid * 4;
        float t_m = time_it([&] { hipLaunchKernelGGL((k_mix<4, 0>), dim3(grid), dim3(256), 0, 0, out, iters, 1.0000001, 1e-9); });
        float t_v = time_it([&] { hipLaunchKernelGGL((k_mix<0, 48>), dim3(grid), dim3(256), 0, 0, out, iters, 1.0000001, 1e-9); });
        float t_b = time_it([&] { hipLaunchKernelGGL((k_mix<4, 48>), dim3(grid), dim3(256), 0, 0, out, iters, 1.0000001, 1e-9); });
        printf("blocks/CU %d: 4 MFMA/iter %.3f ms (%.2f TFLOP/s, %.1f cyc/MFMA/SIMD @2.1GHz)  48 FMA/iter %.3f ms (%.2e lane-op/s)  both %.3f ms  (sum %.3f, max %.3f)\n",
               bpc, t_m, waves * iters * 4 * 2048.0 / (t_m * 1e-3) / 1e12, t_m * 1e-3 * 2.1e9 / (iters * 4.0 * bpc),
               t_v, waves * 64 * iters * 48.0 / (t_v * 1e-3), t_b, t_m + t_v, t_m > t_v ? t_m : t_v);
    }
    return 0;
}
