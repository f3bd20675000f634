// Probe v_mfma_f64_4x4x4_4b_f64 on gfx950: operand lane layout and issue rate against 16x16x4.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef double v4d __attribute__((ext_vector_type(4)));

__global__ void layout_kernel(int* out) {  // out[la*64+lb] = D lane that became non-zero (or -1), value encoded
    const int lane = threadIdx.x;
    for (int la = 0; la < 64; ++la)
        for (int lb = 0; lb < 64; ++lb) {
            const double a = lane == la ? 1.0 : 0.0, b = lane == lb ? 1.0 : 0.0;
            double d = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, 0.0, 0, 0, 0);
            unsigned long long m = __ballot(d != 0.0);
            if (lane == 0) out[la * 64 + lb] = m ? (int)__builtin_ctzll(m) + 64 * (int)__popcll(m) : -1;
        }
}

template <int KIND>
__global__ void rate_kernel(double* out, int iters) {
    const int lane = threadIdx.x & 63;
    double a = lane * 0.001, b = 1.0 + lane * 0.002;
    if (KIND == 0) {
        double d0 = 0, d1 = 0, d2 = 0, d3 = 0;
        for (int i = 0; i < iters; ++i) {
            d0 = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, d0, 0, 0, 0);
            d1 = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, d1, 0, 0, 0);
            d2 = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, d2, 0, 0, 0);
            d3 = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, d3, 0, 0, 0);
        }
        out[blockIdx.x * blockDim.x + threadIdx.x] = d0 + d1 + d2 + d3;
    } else {
        v4d d0 = {0, 0, 0, 0}, d1 = d0, d2 = d0, d3 = d0;
        for (int i = 0; i < iters; ++i) {
            d0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, d0, 0, 0, 0);
            d1 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, d1, 0, 0, 0);
            d2 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, d2, 0, 0, 0);
            d3 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, d3, 0, 0, 0);
        }
        out[blockIdx.x * blockDim.x + threadIdx.x] = d0[0] + d1[1] + d2[2] + d3[3];
    }
}

int main() {
    int* d_out;
    hipMalloc(&d_out, 4096 * sizeof(int));
    layout_kernel<<<1, 64>>>(d_out);
    std::vector<int> h(4096);
    hipMemcpy(h.data(), d_out, 4096 * sizeof(int), hipMemcpyDeviceToHost);
    printf("layout: rows la (A lane), cols lb (B lane): D lane if exactly one else code\n");
    for (int la = 0; la < 64; ++la) {
        printf("A%02d:", la);
        for (int lb = 0; lb < 64; ++lb) {
            int v = h[la * 64 + lb];
            if (v < 0) printf(" ..");
            else printf(" %02d", v % 64);
        }
        printf("\n");
    }
    double* d_o;
    hipMalloc(&d_o, 1024 * 256 * sizeof(double));
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    for (int kind = 0; kind < 2; ++kind)
        for (int wpb : {64, 128, 256}) {  // 1, 2, 4 waves per CU... blocks = 1024 -> 4 per CU
            const int iters = 20000;
            for (int rep = 0; rep < 2; ++rep) {
                hipEventRecord(e0);
                if (kind == 0) rate_kernel<0><<<1024, wpb>>>(d_o, iters);
                else rate_kernel<1><<<1024, wpb>>>(d_o, iters);
                hipEventRecord(e1);
                hipEventSynchronize(e1);
            }
            float ms;
            hipEventElapsedTime(&ms, e0, e1);
            const double n = 1024.0 * (wpb / 64) * iters * 4;  // MFMA instructions
            const double flop = n * (kind == 0 ? 512.0 : 2048.0);
            printf("%s waves/block %d: %.3f ms, %.1f TFLOP/s, %.2f G instr/s\n", kind == 0 ? "4x4x4_4b " : "16x16x4  ",
                   wpb / 64, ms, flop / ms / 1e9, n / ms / 1e6);
        }
    return 0;
}
