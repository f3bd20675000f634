// libnbx: deterministic synthetic (pq|rs) generator.
//
// Same counter hash as oracle/synth.py, so CPU and GPU see identical doubles:
//   u = splitmix64(((stream<<48)|canon(p,q,r,s)) ^ seed), val = (u>>11)*2^-53*2-1,
//   eri = val / N.  The tensor is 8-fold symmetric by construction.
#include "nbx_common.h"
#include "synth_device.h"

// One workgroup per (p,q) row pair; threads sweep the contiguous (r,s) tile.
__global__ __launch_bounds__(256) void synth_eri_kernel(double* __restrict__ eri, int64_t nao,
                                                        int64_t p0, uint64_t seed) {
    const int64_t pq_local = blockIdx.x;
    const int64_t p = p0 + pq_local / nao;
    const int64_t q = pq_local % nao;
    const uint64_t pq = nbx_tri((uint64_t)p, (uint64_t)q);
    const double scale = 1.0 / (double)nao;
    double* tile = eri + pq_local * nao * nao;
    const int64_t n2 = nao * nao;
    for (int64_t f = threadIdx.x; f < n2; f += blockDim.x) {
        const uint64_t r = (uint64_t)(f / nao);
        const uint64_t s = (uint64_t)(f - (int64_t)r * nao);
        tile[f] = nbx_synth_val(0, nbx_tri(pq, nbx_tri(r, s)), seed) * scale;
    }
}

extern "C" int nbx_synth_eri(nbx_ctx* ctx, int64_t nao, int64_t p0, int64_t p1, uint64_t seed,
                             double* d_eri) {
    NBX_CHECK_ARG(ctx != nullptr && d_eri != nullptr);
    NBX_CHECK_ARG(nao > 0 && p0 >= 0 && p1 >= p0 && p1 <= nao);
    const int64_t blocks = (p1 - p0) * nao;
    if (blocks == 0) return NBX_OK;
    NBX_CHECK_ARG(blocks < (int64_t)1 << 31);
    hipLaunchKernelGGL(synth_eri_kernel, dim3((unsigned)blocks), dim3(256), 0, ctx->stream, d_eri, nao,
                       p0, seed);
    NBX_LAUNCH_CHECK();
    return NBX_OK;
}
