#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
// global -> LDS without registers: lane l of the wave writes 16 bytes at lds_base + 16 l
__global__ __launch_bounds__(256) void probe(const double* __restrict__ src, double* __restrict__ dst, int n) {
    extern __shared__ __attribute__((aligned(16))) double smem[];
    const int tid = threadIdx.x, w = tid >> 6, lane = tid & 63;
    // each wave loads 2 slots of 64 x 16 bytes
    for (int k = 0; k < 2; ++k) {
        const double* g = src + (size_t)blockIdx.x * 1024 + (w * 2 + k) * 128 + 2 * lane;
        __builtin_amdgcn_global_load_lds(g, (__attribute__((address_space(3))) void*)(smem + (w * 2 + k) * 128), 16, 0, 0);
    }
    __builtin_amdgcn_s_waitcnt(0);
    __syncthreads();
    for (int i = tid; i < 1024; i += 256) dst[(size_t)blockIdx.x * 1024 + i] = smem[i] * 2.0;
}
int main() {
    const int nb = 4, n = nb * 1024;
    std::vector<double> h(n), o(n);
    for (int i = 0; i < n; ++i) h[i] = i + 0.5;
    double *d, *e;
    hipMalloc(&d, n * 8); hipMalloc(&e, n * 8);
    hipMemcpy(d, h.data(), n * 8, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(probe, dim3(nb), dim3(256), 1024 * 8, 0, d, e, n);
    hipMemcpy(o.data(), e, n * 8, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int i = 0; i < n; ++i) if (o[i] != 2.0 * h[i]) { if (bad < 5) printf("mismatch %d: %g vs %g\n", i, o[i], 2 * h[i]); ++bad; }
    printf("bad %d of %d\n", bad, n);
    return bad != 0;
}
