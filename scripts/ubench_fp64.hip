// Micro-benchmark: what does MI355X sustain for the fp64 instruction mixes of the accumulate
// kernel?  (diagnostic tool, not part of the library)   hipcc --offload-arch=gfx950 -O3
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

template <int R>
__global__ __launch_bounds__(256) void k_fma(double* out, int iters, double a, double b) {
    double v[R];
#pragma unroll
    for (int k = 0; k < R; ++k) v[k] = threadIdx.x * 1e-3 + k;
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int k = 0; k < R; ++k) v[k] = fma(v[k], a, b);
    }
    double s = 0;
#pragma unroll
    for (int k = 0; k < R; ++k) s += v[k];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int R>
__global__ __launch_bounds__(256) void k_mul(double* out, int iters, double a) {
    double v[R];
#pragma unroll
    for (int k = 0; k < R; ++k) v[k] = 1.0 + threadIdx.x * 1e-9 + k * 1e-10;
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int k = 0; k < R; ++k) v[k] = v[k] * a;
    }
    double s = 0;
#pragma unroll
    for (int k = 0; k < R; ++k) s += v[k];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int R>
__global__ __launch_bounds__(256) void k_add(double* out, int iters, double a) {
    double v[R];
#pragma unroll
    for (int k = 0; k < R; ++k) v[k] = threadIdx.x * 1e-3 + k;
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int k = 0; k < R; ++k) v[k] = v[k] + a;
    }
    double s = 0;
#pragma unroll
    for (int k = 0; k < R; ++k) s += v[k];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

// FMA whose three sources are all (non-uniform) VGPR pairs
template <int R>
__global__ __launch_bounds__(256) void k_fma3(double* out, int iters) {
    double v[R], w[R], u[R];
#pragma unroll
    for (int k = 0; k < R; ++k) { v[k] = threadIdx.x * 1e-3 + k; w[k] = 1.0 + 1e-9 * threadIdx.x + 1e-10 * k; u[k] = 1e-9 * (threadIdx.x + k); }
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int k = 0; k < R; ++k) v[k] = fma(v[k], w[k], u[k]);
    }
    double s = 0;
#pragma unroll
    for (int k = 0; k < R; ++k) s += v[k];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

// running fraction with the per-line constants in VGPRs (as after an LDS broadcast read)
template <int R>
__global__ __launch_bounds__(256) void k_rf_vgpr(double* out, int iters, const double* __restrict__ params) {
    double N[R], D[R];
    const double x0 = threadIdx.x * (double)R;
#pragma unroll
    for (int k = 0; k < R; ++k) { N[k] = 0; D[k] = 1; }
    double acc = 0;
    __shared__ double sh[128];
    if (threadIdx.x < 128) sh[threadIdx.x] = params[threadIdx.x];
    __syncthreads();
    for (int i = 0; i < iters; ++i) {
        const double cf = sh[(i & 63) * 2 + 0] + i, a2 = sh[(i & 63) * 2 + 1], KL = sh[((i + 7) & 63) * 2 + 1] * 1e-24;
        const double d0 = x0 - cf;
#pragma unroll
        for (int k = 0; k < R; ++k) {
            const double d = d0 + (double)k;
            const double den = fma(d, d, a2);
            const double t = KL * D[k];
            N[k] = fma(N[k], den, t);
            D[k] *= den;
        }
        if ((i & 31) == 31) {
#pragma unroll
            for (int k = 0; k < R; ++k) { acc += N[k] / D[k]; N[k] = 0; D[k] = 1; }
        }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
}

// running fraction with every multiply and add written as an FMA
template <int R>
__global__ __launch_bounds__(256) void k_rf_fma(double* out, int iters, const double* __restrict__ params, double one, double zero) {
    double N[R], D[R];
    const double x0 = threadIdx.x * (double)R;
#pragma unroll
    for (int k = 0; k < R; ++k) { N[k] = 0; D[k] = 1; }
    double acc = 0;
    for (int i = 0; i < iters; ++i) {
        const double cf = params[(i & 63) * 2 + 0] + i, a2 = params[(i & 63) * 2 + 1];
        const double d0 = x0 - cf;
#pragma unroll
        for (int k = 0; k < R; ++k) {
            const double d = __builtin_fma(one, d0, (double)k);
            const double den = __builtin_fma(d, d, a2);
            const double t = __builtin_fma(1e-20, D[k], zero);
            N[k] = __builtin_fma(N[k], den, t);
            D[k] = __builtin_fma(D[k], den, zero);
        }
        if ((i & 15) == 15) {
#pragma unroll
            for (int k = 0; k < R; ++k) { acc += N[k] / D[k]; N[k] = 0; D[k] = 1; }
        }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
}

// the running-fraction Lorentz body: per point  d = d0+k; den = d*d+a2; t = K*D; N = N*den+t; D *= den
template <int R>
__global__ __launch_bounds__(256) void k_rf(double* out, int iters, const double* __restrict__ params) {
    double N[R], D[R];
    const double x0 = threadIdx.x * (double)R;
#pragma unroll
    for (int k = 0; k < R; ++k) { N[k] = 0; D[k] = 1; }
    double acc = 0;
    for (int i = 0; i < iters; ++i) {
        const double cf = params[(i & 63) * 2 + 0] + i, a2 = params[(i & 63) * 2 + 1];   // uniform -> scalar loads of a tiny table
        const double d0 = x0 - cf;
#pragma unroll
        for (int k = 0; k < R; ++k) {
            const double d = d0 + (double)k;
            const double den = fma(d, d, a2);
            const double t = 1e-20 * D[k];
            N[k] = fma(N[k], den, t);
            D[k] *= den;
        }
        if ((i & 15) == 15) {
#pragma unroll
            for (int k = 0; k < R; ++k) { acc += N[k] / D[k]; N[k] = 0; D[k] = 1; }
        }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
}

// plain divide per pair
template <int R>
__global__ __launch_bounds__(256) void k_div(double* out, int iters, const double* __restrict__ params) {
    double acc[R];
    const double x0 = threadIdx.x * (double)R;
#pragma unroll
    for (int k = 0; k < R; ++k) acc[k] = 0;
    for (int i = 0; i < iters; ++i) {
        const double cf = params[(i & 63) * 2 + 0] + i, a2 = params[(i & 63) * 2 + 1];
        const double d0 = x0 - cf;
#pragma unroll
        for (int k = 0; k < R; ++k) {
            const double d = d0 + (double)k;
            acc[k] += 1e-20 / fma(d, d, a2);
        }
    }
    double s = 0;
#pragma unroll
    for (int k = 0; k < R; ++k) s += acc[k];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

// v_rcp_f64 + one Newton step
template <int R>
__global__ __launch_bounds__(256) void k_rcp(double* out, int iters, const double* __restrict__ params) {
    double acc[R];
    const double x0 = threadIdx.x * (double)R;
#pragma unroll
    for (int k = 0; k < R; ++k) acc[k] = 0;
    for (int i = 0; i < iters; ++i) {
        const double cf = params[(i & 63) * 2 + 0] + i, a2 = params[(i & 63) * 2 + 1];
        const double d0 = x0 - cf;
#pragma unroll
        for (int k = 0; k < R; ++k) {
            const double d = d0 + (double)k;
            const double den = fma(d, d, a2);
            double r = __builtin_amdgcn_rcp(den);
            r = fma(fma(-den, r, 1.0), r, r);
            acc[k] = fma(1e-20, r, acc[k]);
        }
    }
    double s = 0;
#pragma unroll
    for (int k = 0; k < R; ++k) s += acc[k];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int R>
__global__ __launch_bounds__(256) void k_exp(double* out, int iters, double b) {
    double acc[R];
    const double x0 = threadIdx.x * 1e-3;
#pragma unroll
    for (int k = 0; k < R; ++k) acc[k] = 0;
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int k = 0; k < R; ++k) acc[k] += exp(-b * (x0 + k + i * 1e-4));
    }
    double s = 0;
#pragma unroll
    for (int k = 0; k < R; ++k) s += acc[k];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <typename F>
static float time_it(F launch, int reps) {
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    launch();
    hipDeviceSynchronize();
    hipEventRecord(a);
    for (int i = 0; i < reps; ++i) launch();
    hipEventRecord(b);
    hipEventSynchronize(b);
    float ms = 0;
    hipEventElapsedTime(&ms, a, b);
    return ms / reps;
}

int main() {
    const int blocks_per_cu[] = {2, 8};
    double* out; double* params;
    CHECK(hipMalloc(&out, 256 * 8 * 256 * sizeof(double) * 4));
    CHECK(hipMalloc(&params, 128 * sizeof(double)));
    double hp[128];
    for (int i = 0; i < 64; ++i) { hp[2 * i] = 100.0 + i; hp[2 * i + 1] = 4900.0 + i; }
    CHECK(hipMemcpy(params, hp, sizeof hp, hipMemcpyHostToDevice));
    const int iters = 20000;
    printf("%-28s %8s %10s %14s %16s\n", "kernel", "blk/CU", "ms", "lane-op/s", "evals/s");
    for (int bpc : blocks_per_cu) {
        const int grid = 256 * bpc;
        const double lanes = (double)grid * 256;
        float ms;
        ms = time_it([&] { hipLaunchKernelGGL(k_fma<8>, dim3(grid), dim3(256), 0, 0, out, iters, 1.0000001, 1e-9); }, 5);
        printf("%-28s %8d %10.3f %14.3e %16s\n", "fma x8 chains", bpc, ms, lanes * iters * 8 / (ms * 1e-3), "-");
        ms = time_it([&] { hipLaunchKernelGGL(k_mul<8>, dim3(grid), dim3(256), 0, 0, out, iters, 1.0000001); }, 5);
        printf("%-28s %8d %10.3f %14.3e %16s\n", "mul x8 chains", bpc, ms, lanes * iters * 8 / (ms * 1e-3), "-");
        ms = time_it([&] { hipLaunchKernelGGL(k_add<8>, dim3(grid), dim3(256), 0, 0, out, iters, 1e-9); }, 5);
        printf("%-28s %8d %10.3f %14.3e %16s\n", "add x8 chains", bpc, ms, lanes * iters * 8 / (ms * 1e-3), "-");
        ms = time_it([&] { hipLaunchKernelGGL(k_fma3<8>, dim3(grid), dim3(256), 0, 0, out, iters); }, 5);
        printf("%-28s %8d %10.3f %14.3e %16s\n", "fma 3 VGPR sources x8", bpc, ms, lanes * iters * 8 / (ms * 1e-3), "-");
        ms = time_it([&] { hipLaunchKernelGGL(k_rf_vgpr<4>, dim3(grid), dim3(256), 0, 0, out, iters, params); }, 5);
        printf("%-28s %8d %10.3f %14.3e %16.3e\n", "running fraction LDS consts R=4", bpc, ms, lanes * iters * 4 * 5 / (ms * 1e-3), lanes * iters * 4 / (ms * 1e-3));
        ms = time_it([&] { hipLaunchKernelGGL(k_rf_fma<4>, dim3(grid), dim3(256), 0, 0, out, iters, params, 1.0, 0.0); }, 5);
        printf("%-28s %8d %10.3f %14.3e %16.3e\n", "running fraction FMA-only R=4", bpc, ms, lanes * iters * 4 * 5 / (ms * 1e-3), lanes * iters * 4 / (ms * 1e-3));
        ms = time_it([&] { hipLaunchKernelGGL(k_rf<4>, dim3(grid), dim3(256), 0, 0, out, iters, params); }, 5);
        printf("%-28s %8d %10.3f %14.3e %16.3e\n", "running fraction R=4", bpc, ms, lanes * iters * 4 * 5 / (ms * 1e-3), lanes * iters * 4 / (ms * 1e-3));
        ms = time_it([&] { hipLaunchKernelGGL(k_rf<8>, dim3(grid), dim3(256), 0, 0, out, iters, params); }, 5);
        printf("%-28s %8d %10.3f %14.3e %16.3e\n", "running fraction R=8", bpc, ms, lanes * iters * 8 * 5 / (ms * 1e-3), lanes * iters * 8 / (ms * 1e-3));
        ms = time_it([&] { hipLaunchKernelGGL(k_div<4>, dim3(grid), dim3(256), 0, 0, out, iters, params); }, 5);
        printf("%-28s %8d %10.3f %14s %16.3e\n", "IEEE divide R=4", bpc, ms, "-", lanes * iters * 4 / (ms * 1e-3));
        ms = time_it([&] { hipLaunchKernelGGL(k_rcp<4>, dim3(grid), dim3(256), 0, 0, out, iters, params); }, 5);
        printf("%-28s %8d %10.3f %14s %16.3e\n", "rcp+1 Newton R=4", bpc, ms, "-", lanes * iters * 4 / (ms * 1e-3));
        ms = time_it([&] { hipLaunchKernelGGL(k_rcp<8>, dim3(grid), dim3(256), 0, 0, out, iters, params); }, 5);
        printf("%-28s %8d %10.3f %14s %16.3e\n", "rcp+1 Newton R=8", bpc, ms, "-", lanes * iters * 8 / (ms * 1e-3));
        ms = time_it([&] { hipLaunchKernelGGL(k_exp<4>, dim3(grid), dim3(256), 0, 0, out, iters / 10, 0.37); }, 3);
        printf("%-28s %8d %10.3f %14s %16.3e\n", "exp() R=4", bpc, ms, "-", lanes * (iters / 10) * 4 / (ms * 1e-3));
    }
    return 0;
}
