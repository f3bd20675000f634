// probe: v_fmac_f64 with DPP row_newbcast on gfx950 -- semantics and rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ void sem(const double* x, const double* t, double* out) {
    double d = x[threadIdx.x], tv = t[threadIdx.x], acc = 0.0;
    asm volatile("v_fmac_f64_dpp %0, %1, %2 row_newbcast:5 row_mask:0xf bank_mask:0xf" : "+v"(acc) : "v"(d), "v"(tv));
    out[threadIdx.x] = acc;
}
template <bool DPP>
__global__ void rate(double* out, int iters) {
    double a0 = threadIdx.x, a1 = 1, a2 = 2, a3 = 3, a4 = 4, a5 = 5, a6 = 6, a7 = 7, d = 1e-9 * threadIdx.x, tv = 1.0000001;
    for (int i = 0; i < iters; ++i) {
        if (DPP) {
            asm volatile("v_fmac_f64_dpp %0, %8, %9 row_newbcast:1 row_mask:0xf bank_mask:0xf\n"
                         "v_fmac_f64_dpp %1, %8, %9 row_newbcast:2 row_mask:0xf bank_mask:0xf\n"
                         "v_fmac_f64_dpp %2, %8, %9 row_newbcast:3 row_mask:0xf bank_mask:0xf\n"
                         "v_fmac_f64_dpp %3, %8, %9 row_newbcast:4 row_mask:0xf bank_mask:0xf\n"
                         "v_fmac_f64_dpp %4, %8, %9 row_newbcast:5 row_mask:0xf bank_mask:0xf\n"
                         "v_fmac_f64_dpp %5, %8, %9 row_newbcast:6 row_mask:0xf bank_mask:0xf\n"
                         "v_fmac_f64_dpp %6, %8, %9 row_newbcast:7 row_mask:0xf bank_mask:0xf\n"
                         "v_fmac_f64_dpp %7, %8, %9 row_newbcast:8 row_mask:0xf bank_mask:0xf\n"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(d), "v"(tv));
        } else {
            asm volatile("v_fmac_f64_e32 %0, %8, %9\nv_fmac_f64_e32 %1, %8, %9\nv_fmac_f64_e32 %2, %8, %9\nv_fmac_f64_e32 %3, %8, %9\n"
                         "v_fmac_f64_e32 %4, %8, %9\nv_fmac_f64_e32 %5, %8, %9\nv_fmac_f64_e32 %6, %8, %9\nv_fmac_f64_e32 %7, %8, %9\n"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(d), "v"(tv));
        }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
}
int main() {
    double *x, *t, *o;
    hipMalloc(&x, 64 * 8); hipMalloc(&t, 64 * 8); hipMalloc(&o, 1 << 24);
    std::vector<double> hx(64), ht(64), ho(64);
    for (int i = 0; i < 64; ++i) { hx[i] = 100 + i; ht[i] = 1.0; }
    hipMemcpy(x, hx.data(), 512, hipMemcpyHostToDevice); hipMemcpy(t, ht.data(), 512, hipMemcpyHostToDevice);
    sem<<<1, 64>>>(x, t, o);
    hipMemcpy(ho.data(), o, 512, hipMemcpyDeviceToHost);
    printf("row_newbcast:5 -> lane0 %.0f lane7 %.0f lane16 %.0f lane17 %.0f lane40 %.0f lane63 %.0f\n", ho[0], ho[7], ho[16], ho[17], ho[40], ho[63]);
    const int iters = 20000, wgs = 256 * 8;
    for (int rep = 0; rep < 2; ++rep) {
        hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
        float ms;
        hipEventRecord(a); rate<false><<<wgs, 256>>>(o, iters); hipEventRecord(b); hipEventSynchronize(b); hipEventElapsedTime(&ms, a, b);
        printf("plain fmac: %.3f ms -> %.1f G wave-instr/s\n", ms, (double)wgs * 4 * iters * 8 / ms / 1e6);
        hipEventRecord(a); rate<true><<<wgs, 256>>>(o, iters); hipEventRecord(b); hipEventSynchronize(b); hipEventElapsedTime(&ms, a, b);
        printf("dpp   fmac: %.3f ms -> %.1f G wave-instr/s\n", ms, (double)wgs * 4 * iters * 8 / ms / 1e6);
    }
    return 0;
}
