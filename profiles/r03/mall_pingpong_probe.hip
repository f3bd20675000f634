// Does the 256 MB Infinity Cache help a kernel that streams the same 0.97 GB every launch, if alternate launches walk
// their ranges backwards (what was touched last is touched first)?  Persistent workgroups over contiguous ranges, 16-byte
// loads, plain or non-temporal.
//   hipcc --offload-arch=gfx950 -O3 profiles/r03/mall_pingpong_probe.hip -o mall_probe
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double d2 __attribute__((ext_vector_type(2)));
template <bool NT>
__global__ __launch_bounds__(256) void k(const double* __restrict__ src, double* __restrict__ out, long per_wg, int reverse) {
    const double* base = src + (long)blockIdx.x * per_wg;
    const long nchunk = per_wg / 512;  // 256 threads x 2 doubles
    d2 acc = {0.0, 0.0};
    for (long c = 0; c < nchunk; c += 4) {
        d2 v[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            long cc = c + u;
            cc = cc < nchunk ? cc : nchunk - 1;
            const long pos = reverse ? nchunk - 1 - cc : cc;
            const d2* p = reinterpret_cast<const d2*>(base + pos * 512 + threadIdx.x * 2);
            v[u] = NT ? __builtin_nontemporal_load(p) : *p;
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) acc += v[u];
    }
    if (acc.x + acc.y == 12345.678) out[blockIdx.x] = acc.x;
}
template <bool NT>
void run(const char* name, const double* d, double* o, long n, int wgs, bool pingpong) {
    const long per_wg = n / wgs / 512 * 512;
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    for (int i = 0; i < 4; ++i) k<NT><<<wgs, 256>>>(d, o, per_wg, pingpong ? (i & 1) : 0);
    hipEventRecord(e0);
    const int reps = 20;
    for (int i = 0; i < reps; ++i) k<NT><<<wgs, 256>>>(d, o, per_wg, pingpong ? (i & 1) : 0);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    printf("%-44s %4d workgroups: %.1f us per pass, %.2f TB/s\n", name, wgs, ms / reps * 1e3, (double)per_wg * wgs * 8 / (ms / reps * 1e-3) / 1e12);
}
int main() {
    const long n = 972581408 / 8;
    double *d, *o;
    hipMalloc(&d, n * 8);
    hipMalloc(&o, 1 << 20);
    hipMemset(d, 0, n * 8);
    for (int wgs : {512, 2048}) {
        run<false>("plain loads, always forward", d, o, n, wgs, false);
        run<false>("plain loads, forward / backward", d, o, n, wgs, true);
        run<true>("non-temporal loads, always forward", d, o, n, wgs, false);
        run<true>("non-temporal loads, forward / backward", d, o, n, wgs, true);
    }
    return 0;
}
