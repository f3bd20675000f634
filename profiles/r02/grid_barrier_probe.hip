// probe: cost of a grid-wide barrier among G co-resident workgroups (atomic arrive + spin on a generation word)
#include <hip/hip_runtime.h>
#include <cstdio>
__device__ __forceinline__ void grid_barrier(unsigned* count, unsigned* gen, unsigned nwg, unsigned& my_gen) {
    __syncthreads();
    if (threadIdx.x == 0) {
        __threadfence();
        const unsigned g = my_gen;
        if (atomicAdd(count, 1u) == nwg - 1) {
            atomicExch(count, 0u);
            __threadfence();
            atomicAdd(gen, 1u);
        } else {
            while (__hip_atomic_load(gen, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) == g) __builtin_amdgcn_s_sleep(1);
        }
        __threadfence();
    }
    my_gen += 1;
    __syncthreads();
}
__global__ void k(unsigned* count, unsigned* gen, int iters, double* data, int n) {
    unsigned my_gen = 0;
    for (int i = 0; i < iters; ++i) {
        // a little work between barriers: each WG touches its slice
        for (int j = blockIdx.x * blockDim.x + threadIdx.x; j < n; j += gridDim.x * blockDim.x) data[j] += 1.0;
        grid_barrier(count, gen, gridDim.x, my_gen);
    }
}
// only the workgroups that land on one XCD (blockIdx % 8 == 0 under round-robin dispatch) take part
__global__ void k1(unsigned* count, unsigned* gen, int iters, double* data, int n) {
    if (blockIdx.x & 7) return;
    const int wg = blockIdx.x >> 3, nwg = gridDim.x >> 3;
    unsigned my_gen = 0;
    for (int i = 0; i < iters; ++i) {
        for (int j = wg * blockDim.x + threadIdx.x; j < n; j += nwg * blockDim.x) data[j] += 1.0;
        grid_barrier(count, gen, nwg, my_gen);
    }
}
int main() {
    unsigned* c; double* d;
    (void)hipMalloc(&c, 256); (void)hipMemset(c, 0, 256);
    const int n = 2 * 148 * 148;
    (void)hipMalloc(&d, n * 8); (void)hipMemset(d, 0, n * 8);
    for (int G : {16, 32, 64, 128, 256}) {
        for (int T : {64, 256}) {
            (void)hipMemset(c, 0, 256);
            const int iters = 2000;
            k<<<G, T>>>(c, c + 32, 10, d, n);
            (void)hipDeviceSynchronize();
            (void)hipMemset(c, 0, 256);
            hipEvent_t a, b; (void)hipEventCreate(&a); (void)hipEventCreate(&b);
            (void)hipEventRecord(a);
            k<<<G, T>>>(c, c + 32, iters, d, n);
            (void)hipEventRecord(b); (void)hipEventSynchronize(b);
            float ms; (void)hipEventElapsedTime(&ms, a, b);
            printf("G=%3d threads=%3d: %.2f us per (work + barrier)\n", G, T, ms * 1e3 / iters);
        }
    }
    for (int G : {8, 16, 32, 64}) {
        (void)hipMemset(c, 0, 256);
        const int iters = 2000;
        k1<<<G * 8, 256>>>(c, c + 32, 10, d, n);
        (void)hipDeviceSynchronize();
        (void)hipMemset(c, 0, 256);
        hipEvent_t a, b; (void)hipEventCreate(&a); (void)hipEventCreate(&b);
        (void)hipEventRecord(a);
        k1<<<G * 8, 256>>>(c, c + 32, iters, d, n);
        (void)hipEventRecord(b); (void)hipEventSynchronize(b);
        float ms; (void)hipEventElapsedTime(&ms, a, b);
        printf("one XCD, G=%3d: %.2f us per (work + barrier)\n", G, ms * 1e3 / iters);
    }
    return 0;
}
