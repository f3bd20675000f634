// How fast can one workgroup per CU bring a contiguous range of HBM into LDS, chunk by chunk, and does it matter how?
// 0.992 GB (the N_AO = 148 packed J/K tensor in jk_m4.hip's layout) split into 256 contiguous ranges, one persistent
// workgroup each, chunks of 24 KB into a ring of R LDS buffers, W waves issuing, one barrier per chunk (or none):
//   dma  : global_load_lds_dwordx4 (HBM -> LDS, no registers), explicit s_waitcnt vmcnt
//   reg  : non-temporal global_load_dwordx4 into registers, ds_write_b128 a chunk later
// Nothing reads the data (an LDS checksum of the last chunk keeps the writes alive).
//   hipcc --offload-arch=gfx950 -O3 profiles/r03/lds_dma_stream_probe.hip -o dma_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#pragma clang diagnostic ignored "-Winline-asm"
typedef __attribute__((address_space(3))) void* lds_vp;
typedef double d2 __attribute__((ext_vector_type(2)));
constexpr int CHUNK = 24576;  // bytes

template <int N>
__device__ __forceinline__ void wait_vm() {
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

// W waves, ring of R chunks, D chunks in flight (D <= R - 1)
template <int W, int R, int D, bool BARRIER, int POL, int MODE = 0, int BURST = 0>
__global__ __launch_bounds__(MODE >= 3 ? 512 : W * 64, 1) void dma_kernel(const char* __restrict__ src, double* __restrict__ out, long per_wg) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int LPT = CHUNK / (W * 64 * 16);  // instructions per wave and chunk
    static_assert(LPT * W * 64 * 16 == CHUNK, "chunk");
    int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const char* base = src + (long)blockIdx.x * per_wg;
    const int nchunk = (int)(per_wg / CHUNK);
    if (MODE >= 3) {  // eight waves, the upper four load
        if (wave < 4) {
            for (int g = 0; g < nchunk; ++g) __syncthreads();
            __syncthreads();
            return;
        }
        tid -= 256;
        wave -= 4;
    }
    // MODE >= 1: the chunks of jk_m4.hip's tiles -- 190, 188, 150, 175 blocks of 128 bytes in a 192-block buffer; the
    // lanes past a chunk's end re-read its last 16 bytes (every chunk the same number of instructions)
    auto issue = [&](int g) {
        const int k = g & 3;
        const int len = MODE == 0 ? CHUNK : (k == 0 ? 190 : k == 1 ? 188 : k == 2 ? 150 : 175) * 128;
        const int start = MODE == 0 ? 0 : (k == 0 ? 0 : k == 1 ? 190 : k == 2 ? 378 : 528) * 128;
        const char* p = MODE == 0 ? base + (long)(g < nchunk ? g : 0) * CHUNK
                                  : base + (long)(g < nchunk ? (g >> 2) : 0) * (703 * 128) + start;
        char* buf = smem + (g % R) * CHUNK;
#pragma unroll
        for (int s = 0; s < LPT; ++s) {
            unsigned off = (unsigned)((s * W * 64 + tid) * 16);
            if (MODE >= 1) off = off < (unsigned)len - 16 ? off : (unsigned)len - 16;
            const unsigned lds_a = (unsigned)(size_t)(lds_vp)(buf + (s * W + wave) * 1024);
            if (POL == 0) asm volatile("s_mov_b32 m0, %2\n\tglobal_load_lds_dwordx4 %0, %1 nt" : : "v"(off), "s"(p), "s"(lds_a) : "memory", "m0");
            else asm volatile("s_mov_b32 m0, %2\n\tglobal_load_lds_dwordx4 %0, %1 sc0 sc1 nt" : : "v"(off), "s"(p), "s"(lds_a) : "memory", "m0");
        }
    };
#pragma unroll
    for (int g = 0; g < D; ++g) issue(g);
    for (int g = 0; g < nchunk; ++g) {
        issue(g + D);
        if ((MODE == 2 || MODE == 3) && (g & 3) == 0) out[4096 + (long)blockIdx.x * 1024 + tid] = (double)g;  // (a tile's partial row leaves: same place every tile)
        if ((MODE == 4 || MODE == 5) && (g & 3) == 0) {  // the same 2.4 KB to FRESH lines, tile after tile (what jk_m4's row-q partials do)
            double* dst = out + 4096 + ((long)blockIdx.x * (nchunk / 4 + 1) + (g >> 2)) * 296;
            if (MODE == 4 || tid <= ((g >> 2) * 7) % 148) {  // MODE 5: a third of the row on average (columns <= q)
                dst[tid] = (double)g;
                if (tid < 40) dst[256 + tid] = (double)g;
            }
        }
        if (BURST > 0 && (g & 3) == 3 && ((g >> 2) + 1) % BURST == 0) {  // the rows of the last BURST tiles, together, mid-stream
            for (int t = (g >> 2) + 1 - BURST; t <= (g >> 2); ++t) {
                double* dst = out + 4096 + ((long)blockIdx.x * (nchunk / 4 + 1) + t) * 296;
                if (tid <= (t * 7) % 148) {
                    dst[tid] = (double)t;
                    if (tid < 40) dst[256 + tid] = (double)t;
                }
            }
        }
        wait_vm<D * LPT>();  // chunk g has landed
        if (BARRIER) __syncthreads();
    }
    wait_vm<0>();
    if (MODE >= 6) {  // every tile's row at the END of the range, one burst (the rows would have waited in registers)
        for (int t = 0; t < nchunk / 4; ++t) {
            double* dst = out + 4096 + ((long)blockIdx.x * (nchunk / 4 + 1) + t) * 296;
            if (MODE == 6 || tid <= (t * 7) % 148) {
                dst[tid] = (double)t;
                if (tid < 40) dst[256 + tid] = (double)t;
            }
        }
    }
    __syncthreads();
    const double v = reinterpret_cast<const double*>(smem)[tid];
    if (v == 12345.678) out[blockIdx.x] = v;
}

// registers: D chunks in flight in registers (D * LPT * 4 VGPRs), written to LDS when they arrive
template <int W, int D, bool BARRIER>
__global__ __launch_bounds__(W * 64, 1) void reg_kernel(const char* __restrict__ src, double* __restrict__ out, long per_wg) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int LPT = CHUNK / (W * 64 * 16);
    const int tid = threadIdx.x;
    const char* base = src + (long)blockIdx.x * per_wg;
    const int nchunk = (int)(per_wg / CHUNK);
    d2 r[D][LPT];
    auto load = [&](int g, d2 (&v)[LPT]) {
        const char* p = base + (long)(g < nchunk ? g : 0) * CHUNK;
#pragma unroll
        for (int s = 0; s < LPT; ++s) v[s] = __builtin_nontemporal_load(reinterpret_cast<const d2*>(p + (s * W * 64 + tid) * 16));
    };
#pragma unroll
    for (int g = 0; g < D; ++g) load(g, r[g]);
    for (int g0 = 0; g0 < nchunk; g0 += D) {
#pragma unroll
        for (int u = 0; u < D; ++u) {
            char* buf = smem + ((g0 + u) & 1) * CHUNK;
#pragma unroll
            for (int s = 0; s < LPT; ++s) *reinterpret_cast<d2*>(buf + (s * W * 64 + tid) * 16) = r[u][s];
            load(g0 + u + D, r[u]);
            if (BARRIER) __syncthreads();
        }
    }
    __syncthreads();
    const double v = reinterpret_cast<const double*>(smem)[tid];
    if (v == 12345.678) out[blockIdx.x] = v;
}

template <typename K>
void run(const char* name, K kern, int threads, size_t lds, const char* d, double* o, long n, int wgs = 256) {
    const long per_wg = n / wgs / CHUNK * CHUNK;
    hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    for (int i = 0; i < 4; ++i) kern<<<wgs, threads, lds>>>(d, o, per_wg);
    hipEventRecord(e0);
    const int reps = 20;
    for (int i = 0; i < reps; ++i) kern<<<wgs, threads, lds>>>(d, o, per_wg);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    printf("%-64s %.1f us per pass, %.2f TB/s\n", name, ms / reps * 1e3, (double)per_wg * wgs / (ms / reps * 1e-3) / 1e12);
}

int main() {
    const long n = 992163840;
    char* d;
    double* o;
    hipMalloc(&d, n + (1 << 20));
    hipMalloc(&o, 64 << 20);
    hipMemset(d, 0, n + (1 << 20));
    run("dma  4 waves ring 5, 3 chunks in flight, barrier", dma_kernel<4, 5, 3, true, 0>, 256, 5 * CHUNK, d, o, n);
    run("dma  4 waves ring 5, 3 in flight, no barrier", dma_kernel<4, 5, 3, false, 0>, 256, 5 * CHUNK, d, o, n);
    run("dma  4 waves ring 5, 4 in flight, no barrier", dma_kernel<4, 5, 4, false, 0>, 256, 5 * CHUNK, d, o, n);
    run("dma  4 waves ring 3, 2 in flight, barrier", dma_kernel<4, 3, 2, true, 0>, 256, 3 * CHUNK, d, o, n);
    run("dma  4 waves ring 2, 1 in flight, barrier", dma_kernel<4, 2, 1, true, 0>, 256, 2 * CHUNK, d, o, n);
    run("dma  8 waves ring 5, 3 in flight, barrier", dma_kernel<8, 5, 3, true, 0>, 512, 5 * CHUNK, d, o, n);
    run("dma  8 waves ring 5, 3 in flight, no barrier", dma_kernel<8, 5, 3, false, 0>, 512, 5 * CHUNK, d, o, n);
    run("dma  2 waves ring 5, 3 in flight, barrier", dma_kernel<2, 5, 3, true, 0>, 128, 5 * CHUNK, d, o, n);
    run("dma  4 waves ring 5, 3 in flight, barrier, sc0 sc1 nt", dma_kernel<4, 5, 3, true, 1>, 256, 5 * CHUNK, d, o, n);
    run("dma  8 waves ring 5, 3 in flight, barrier, sc0 sc1 nt", dma_kernel<8, 5, 3, true, 1>, 512, 5 * CHUNK, d, o, n);
    run("dma  4 waves ring 5, 3 in flight, barrier, ragged chunks", dma_kernel<4, 5, 3, true, 0, 1>, 256, 5 * CHUNK, d, o, n);
    run("dma  4 waves ring 5, 3 in flight, barrier, ragged + store", dma_kernel<4, 5, 3, true, 0, 2>, 256, 5 * CHUNK, d, o, n);
    run("dma  ragged + store, 512-thread workgroup, waves 4-7 load", dma_kernel<4, 5, 3, true, 0, 3>, 512, 5 * CHUNK, d, o, n);
    run("dma  512 threads, ragged, 2.4 KB per tile to FRESH lines", dma_kernel<4, 5, 3, true, 0, 4>, 512, 5 * CHUNK, d, o, n);
    run("dma  512 threads, ragged, 0.8 KB per tile to fresh lines", dma_kernel<4, 5, 3, true, 0, 5>, 512, 5 * CHUNK, d, o, n);
    run("dma  512 threads, ragged, 2.4 KB per tile, all at the END", dma_kernel<4, 5, 3, true, 0, 6>, 512, 5 * CHUNK, d, o, n);
    run("dma  512 threads, ragged, 0.8 KB per tile, all at the END", dma_kernel<4, 5, 3, true, 0, 7>, 512, 5 * CHUNK, d, o, n);
    run("dma  512 threads, ragged, 0.8 KB per tile in bursts of 20 tiles", dma_kernel<4, 5, 3, true, 0, 3, 20>, 512, 5 * CHUNK, d, o, n);
    run("dma  512 threads, ragged, 0.8 KB per tile in bursts of 10 tiles", dma_kernel<4, 5, 3, true, 0, 3, 10>, 512, 5 * CHUNK, d, o, n);
    run("dma  512 threads, ragged, 0.8 KB per tile in bursts of 4 tiles", dma_kernel<4, 5, 3, true, 0, 3, 4>, 512, 5 * CHUNK, d, o, n);
    run("dma  the same with 153 KB of LDS", dma_kernel<4, 5, 3, true, 0, 3>, 512, 153 * 1024, d, o, n);
    run("dma  the same, 153 KB, 251 workgroups", dma_kernel<4, 5, 3, true, 0, 3>, 512, 153 * 1024, d, o, n, 251);
    run("reg  4 waves, 2 chunks in registers, barrier", reg_kernel<4, 2, true>, 256, 2 * CHUNK, d, o, n);
    run("reg  8 waves, 2 chunks in registers, barrier", reg_kernel<8, 2, true>, 512, 2 * CHUNK, d, o, n);
    run("reg  8 waves, 4 chunks in registers, barrier", reg_kernel<8, 4, true>, 512, 2 * CHUNK, d, o, n);
    run("reg  8 waves, 2 chunks in registers, no barrier", reg_kernel<8, 2, false>, 512, 2 * CHUNK, d, o, n);
    return 0;
}
