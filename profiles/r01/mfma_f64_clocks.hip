#include <hip/hip_runtime.h>
#include <cstdio>
typedef double v4f64 __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(256) void k(double* out, int iters) {
    v4f64 acc[4];
    for (int i = 0; i < 4; ++i) acc[i] = (v4f64){0, 0, 0, 0};
    double a = threadIdx.x * 1e-3, b = 1.0 + blockIdx.x * 1e-6;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 4; ++i) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
    }
    double s = 0;
    for (int i = 0; i < 4; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}
int main() {
    const int blocks = 256 * 4, iters = 20000;
    double* d; hipMalloc(&d, blocks * 256 * 8);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int rep = 0; rep < 12; ++rep) {
        hipEventRecord(e0);
        for (int j = 0; j < 40; ++j) k<<<blocks, 256>>>(d, iters);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        printf("rep %d: %.2f TFLOP/s\n", rep, (double)blocks * 4 * iters * 4 * 2048.0 * 40 / ms / 1e9);
        fflush(stdout);
    }
    return 0;
}
