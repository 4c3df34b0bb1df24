// mfma_f64_rate.hip -- NOT part of libmixemt_hip.so.  Round-6 measurement: what v_mfma_f64_16x16x4_f64 and
// v_mfma_f64_4x4x4_4b_f64 issue at on gfx950, against v_fma_f64, before a batched-restart pass is designed around either.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/experiments/mfma_f64_rate.hip -o tools/experiments/_build/mfma_f64_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef double d4 __attribute__((ext_vector_type(4)));

template <int NACC>
__global__ __launch_bounds__(256) void mfma16_kernel(double *out, int iters, double a0, double b0) {
    d4 acc[NACC];
    for (int i = 0; i < NACC; ++i) acc[i] = d4{0.0, 0.0, 0.0, 0.0};
    double a = a0 + threadIdx.x * 1e-9, b = b0;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
    }
    double s = 0.0;
    for (int i = 0; i < NACC; ++i) s += acc[i].x + acc[i].y + acc[i].z + acc[i].w;
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int NACC>
__global__ __launch_bounds__(256) void mfma4_kernel(double *out, int iters, double a0, double b0) {
    double acc[NACC];
    for (int i = 0; i < NACC; ++i) acc[i] = 0.0;
    double a = a0 + threadIdx.x * 1e-9, b = b0;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, acc[i], 0, 0, 0);
    }
    double s = 0.0;
    for (int i = 0; i < NACC; ++i) s += acc[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int NACC>
__global__ __launch_bounds__(256) void fma_kernel(double *out, int iters, double a0, double b0) {
    double acc[NACC];
    for (int i = 0; i < NACC; ++i) acc[i] = i;
    double a = a0 + threadIdx.x * 1e-9, b = b0;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < NACC; ++i) acc[i] = fma(a, acc[i], b);
    }
    double s = 0.0;
    for (int i = 0; i < NACC; ++i) s += acc[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
// MFMAs and independent fp64 FMAs in one stream: do the two pipes overlap?
template <int NACC, int NFMA>
__global__ __launch_bounds__(256) void mixed_kernel(double *out, int iters, double a0, double b0) {
    d4 acc[NACC];
    double f[NFMA];
    for (int i = 0; i < NACC; ++i) acc[i] = d4{0.0, 0.0, 0.0, 0.0};
    for (int i = 0; i < NFMA; ++i) f[i] = i;
    double a = a0 + threadIdx.x * 1e-9, b = b0;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < NACC; ++i) {
            acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
#pragma unroll
            for (int j = 0; j < NFMA / NACC; ++j) f[i * (NFMA / NACC) + j] = fma(a, f[i * (NFMA / NACC) + j], b);
        }
    }
    double s = 0.0;
    for (int i = 0; i < NACC; ++i) s += acc[i].x + acc[i].y + acc[i].z + acc[i].w;
    for (int i = 0; i < NFMA; ++i) s += f[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <typename F>
static float timed(F launch) {
    hipEvent_t a, b;
    hipEventCreate(&a);
    hipEventCreate(&b);
    launch();
    hipDeviceSynchronize();
    hipEventRecord(a, 0);
    launch();
    hipEventRecord(b, 0);
    hipEventSynchronize(b);
    float ms = 0;
    hipEventElapsedTime(&ms, a, b);
    return ms;
}

int main() {
    hipDeviceProp_t prop;
    hipGetDeviceProperties(&prop, 0);
    const int ncu = prop.multiProcessorCount;
    int clk_khz = 0;
    hipDeviceGetAttribute(&clk_khz, hipDeviceAttributeClockRate, 0);
    printf("%s, %d CUs, clock attribute %.0f MHz\n", prop.name, ncu, clk_khz / 1000.0);
    double *out;
    hipMalloc(&out, sizeof(double) * ncu * 8 * 256);
    const int iters = 20000;
    for (int wg_per_cu = 1; wg_per_cu <= 2; ++wg_per_cu) {
        const int grid = ncu * wg_per_cu;                    // 4 waves per workgroup: 1 or 2 waves per SIMD
        const double waves_per_simd = wg_per_cu;
        auto report = [&](const char *name, float ms, double instr_per_wave, double flop_per_instr) {
            const double cyc = ms * 1e-3 * (clk_khz * 1e3) / (instr_per_wave * waves_per_simd);
            const double tf = flop_per_instr * instr_per_wave * grid * 4 / (ms * 1e-3) / 1e12;
            printf("%-46s %d wave(s)/SIMD  %8.3f ms  %6.1f cycles per instruction and SIMD (at the clock attribute)  %7.1f TFLOP/s\n",
                   name, wg_per_cu, ms, cyc, tf);
        };
        float ms;
        ms = timed([&] { hipLaunchKernelGGL(mfma16_kernel<4>, dim3(grid), dim3(256), 0, 0, out, iters, 1.0, 1e-3); });
        report("v_mfma_f64_16x16x4_f64, 4 accumulators", ms, 4.0 * iters, 2048.0);
        ms = timed([&] { hipLaunchKernelGGL(mfma16_kernel<8>, dim3(grid), dim3(256), 0, 0, out, iters, 1.0, 1e-3); });
        report("v_mfma_f64_16x16x4_f64, 8 accumulators", ms, 8.0 * iters, 2048.0);
        ms = timed([&] { hipLaunchKernelGGL(mfma16_kernel<1>, dim3(grid), dim3(256), 0, 0, out, iters, 1.0, 1e-3); });
        report("v_mfma_f64_16x16x4_f64, 1 accumulator (dependent)", ms, 1.0 * iters, 2048.0);
        ms = timed([&] { hipLaunchKernelGGL(mfma4_kernel<8>, dim3(grid), dim3(256), 0, 0, out, iters, 1.0, 1e-3); });
        report("v_mfma_f64_4x4x4_4b_f64, 8 accumulators", ms, 8.0 * iters, 512.0);
        ms = timed([&] { hipLaunchKernelGGL(mfma4_kernel<1>, dim3(grid), dim3(256), 0, 0, out, iters, 1.0, 1e-3); });
        report("v_mfma_f64_4x4x4_4b_f64, 1 accumulator (dependent)", ms, 1.0 * iters, 512.0);
        ms = timed([&] { hipLaunchKernelGGL(fma_kernel<16>, dim3(grid), dim3(256), 0, 0, out, iters, 1.0, 1e-3); });
        report("v_fma_f64, 16 independent chains", ms, 16.0 * iters, 128.0);
        ms = timed([&] { hipLaunchKernelGGL((mixed_kernel<4, 16>), dim3(grid), dim3(256), 0, 0, out, iters, 1.0, 1e-3); });
        report("4 MFMA 16x16x4 + 16 v_fma_f64 interleaved (per MFMA)", ms, 4.0 * iters, 2048.0 + 4 * 128.0);
        ms = timed([&] { hipLaunchKernelGGL((mixed_kernel<4, 32>), dim3(grid), dim3(256), 0, 0, out, iters, 1.0, 1e-3); });
        report("4 MFMA 16x16x4 + 32 v_fma_f64 interleaved (per MFMA)", ms, 4.0 * iters, 2048.0 + 8 * 128.0);
    }
    hipFree(out);
    return 0;
}
