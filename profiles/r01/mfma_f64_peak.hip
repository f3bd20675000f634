// fp64 MFMA issue-rate microbenchmark: 4 waves/CU x NACC independent accumulators, registers only.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double v4f64 __attribute__((ext_vector_type(4)));
template <int NACC>
__global__ __launch_bounds__(256) void k(double* out, int iters) {
    v4f64 acc[NACC];
    for (int i = 0; i < NACC; ++i) acc[i] = (v4f64){0, 0, 0, 0};
    double a = threadIdx.x * 1e-3, b = 1.0 + blockIdx.x * 1e-6;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
    }
    double s = 0;
    for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}
template <int NACC>
void run(int blocks_per_cu) {
    const int blocks = 256 * blocks_per_cu, iters = 20000;
    double* d; hipMalloc(&d, blocks * 256 * 8);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    k<NACC><<<blocks, 256>>>(d, 100); hipDeviceSynchronize();
    hipEventRecord(e0); k<NACC><<<blocks, 256>>>(d, iters); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    double flops = (double)blocks * 4 * iters * NACC * 2048.0;
    printf("NACC=%d blocks/CU=%d: %.2f TFLOP/s (%.3f ms)\n", NACC, blocks_per_cu, flops / ms / 1e9, ms);
    hipFree(d);
}
int main() { run<1>(1); run<4>(1); run<16>(1); run<4>(2); run<8>(2); run<4>(4); run<2>(8); return 0; }
