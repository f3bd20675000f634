// What the fp64 pipes of an MI355X sustain, registers only: v_mfma_f64_16x16x4_f64, v_mfma_f64_4x4x4_4b_f64 and
// v_fma_f64, with the shader clock measured by the kernel itself (s_memtime cycles over s_memrealtime's 100 MHz).
//   hipcc --offload-arch=gfx950 -O3 profiles/r03/fp64_rate_probe.hip -o fp64_rate_probe
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double v4f64 __attribute__((ext_vector_type(4)));

template <int MODE, int NACC>
__global__ __launch_bounds__(256) void k(double* out, long long* clk, int iters) {
    const long long c0 = clock64(), w0 = wall_clock64();
    double s = 0;
    if (MODE == 0) {
        v4f64 acc[NACC];
        for (int i = 0; i < NACC; ++i) acc[i] = (v4f64){0, 0, 0, 0};
        double a = threadIdx.x * 1e-3, b = 1.0 + blockIdx.x * 1e-6;
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
        }
        for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    } else if (MODE == 1) {
        double acc[NACC];
        for (int i = 0; i < NACC; ++i) acc[i] = 0;
        double a = threadIdx.x * 1e-3, b = 1.0 + blockIdx.x * 1e-6;
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, acc[i], 0, 0, 0);
        }
        for (int i = 0; i < NACC; ++i) s += acc[i];
    } else {
        double acc[NACC];
        for (int i = 0; i < NACC; ++i) acc[i] = i;
        double a = 1.0 + threadIdx.x * 1e-9, b = blockIdx.x * 1e-6;
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int i = 0; i < NACC; ++i) acc[i] = __builtin_fma(acc[i], a, b);
        }
        for (int i = 0; i < NACC; ++i) s += acc[i];
    }
    const long long c1 = clock64(), w1 = wall_clock64();
    out[blockIdx.x * 256 + threadIdx.x] = s;
    if (threadIdx.x == 0) {
        clk[2 * blockIdx.x] = c1 - c0;
        clk[2 * blockIdx.x + 1] = w1 - w0;
    }
}

template <int MODE, int NACC>
void run(const char* name, int blocks, int iters, double flops_per_instr) {
    double* d;
    long long *c, hc[2];
    hipMalloc(&d, (size_t)blocks * 256 * 8);
    hipMalloc(&c, (size_t)blocks * 16);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    k<MODE, NACC><<<blocks, 256>>>(d, c, 100);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    k<MODE, NACC><<<blocks, 256>>>(d, c, iters);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    hipMemcpy(hc, c, 16, hipMemcpyDeviceToHost);
    const double flops = (double)blocks * 4 * iters * NACC * flops_per_instr;
    const double waves_per_simd = blocks / 256.0;  // 4 waves per block, 4 SIMDs per CU, 256 CUs
    const double mhz = (double)hc[0] / ((double)hc[1] / 100.0);
    printf("%-34s NACC=%2d: %7.2f TFLOP/s  %.3f ms | shader clock %.0f MHz, one instruction per %.1f cycles per SIMD\n", name, NACC,
           flops / ms / 1e9, ms, mhz, ms * 1e-3 * mhz * 1e6 / ((double)iters * NACC * waves_per_simd));
    hipFree(d);
    hipFree(c);
}

int main() {
    for (int rep = 0; rep < 2; ++rep) {
        run<0, 8>("v_mfma_f64_16x16x4 1 wave/SIMD", 256 * 1, 10000, 2048.0);
        run<0, 8>("v_mfma_f64_16x16x4 2 waves/SIMD", 256 * 2, 10000, 2048.0);
        run<0, 8>("v_mfma_f64_16x16x4 4 waves/SIMD", 256 * 4, 10000, 2048.0);
        run<1, 16>("v_mfma_f64_4x4x4_4b 1 wave/SIMD", 256 * 1, 40000, 512.0);
        run<1, 16>("v_mfma_f64_4x4x4_4b 2 waves/SIMD", 256 * 2, 40000, 512.0);
        run<1, 16>("v_mfma_f64_4x4x4_4b 4 waves/SIMD", 256 * 4, 20000, 512.0);
        run<1, 4>("v_mfma_f64_4x4x4_4b 4 acc, 1 w/S", 256 * 1, 40000, 512.0);
        run<2, 16>("v_fma_f64 4 waves/SIMD", 256 * 4, 20000, 128.0);
    }
    return 0;
}
