// libnbx: J/K from a three-index factor of the two-electron integrals (include/nbx.h "density-fitted J/K").
//
// the dense contraction is a matrix-vector product (0.75 flop per byte, HBM
// bound when the tensor is stored, ALU bound when it is generated), so at N_AO = 2000 -- where (pq|rs) is 128 TB -- the
// only J/K that is GEMM shaped is the factorised one,
//
//     (pq|rs) ~ sum_L B_L[p][q] B_L[r][s]          B_L symmetric, L < N_aux   (density fitting / Cholesky vectors)
//     J       = sum_L B_L <B_L, D_a + D_b>
//     K^x     = sum_L (B_L C^x) (B_L C^x)^T         D^x = C^x C^x^T, C^x the occupied orbitals of spin x
//
// This is what PySCF does behind the same get_veff call (nbed/scf/huzinaga_scf.py:156, driver.py:344, 847) when the
// mean-field object is built with `.density_fit()`; the reference never asks for it, so nothing here is on the parity
// path of the exact integrals: an EXTRA, with its own tests.
//
// A slab of LS auxiliary functions at a time:
//   Yt_L (nocc x N) = C_occ^T B_L          one batched 'T','N' GEMM per spin   (2 nocc N^2 flop per L)
//   rho_L           = sum_x <Yt^x_L, C_occ^x^T>                                  (= <B_L, D_a + D_b>)
//   K^x            += Yt^T Yt               one 'T','N' GEMM with k = LS nocc, both spins as a batch of two
//   J              += sum_L rho_L B_L       one streaming pass over the slab
// Both GEMMs are 'T','N' products with long k: gemm_m4_tn_kernel (v_mfma_f64_4x4x4_4b_f64, k-tiles written into LDS by the
// load unit).  B is read three times per build (once per spin's GEMM, once for J): 3 x 8 N^2 N_aux bytes against
// 8 nocc N^2 N_aux flop -- 170 flop per byte at nocc = 512.
//
// The L axis is additive: a rank holding a slab of the auxiliary functions gets partial J and K, summed by the same
// all-reduce as the dense row slabs.
#include "nbx_common.h"
#include "synth_device.h"

namespace {

constexpr int DF_LS = 64;  // auxiliary functions per slab

// rho[l] = sum_x sum_{i < nocc_x, q} yt[x][l][i][q] * ct[x][i][q]      one workgroup per l
__global__ __launch_bounds__(256) void df_rho_kernel(const double* __restrict__ yt, int64_t yt_spin_stride, int64_t yt_l_stride,
                                                     const double* __restrict__ ct, int64_t ct_spin_stride, int ndm, int nocc_a,
                                                     int nocc_b, int N, double* __restrict__ rho) {
    __shared__ double red[17];
    const int l = blockIdx.x;
    double t = 0.0;
    for (int x = 0; x < ndm; ++x) {
        const int64_t n = (int64_t)(x == 0 ? nocc_a : nocc_b) * N;
        const double* y = yt + x * yt_spin_stride + (int64_t)l * yt_l_stride;
        const double* c = ct + x * ct_spin_stride;
        for (int64_t e = threadIdx.x; e < n; e += 256) t = fma(y[e], c[e], t);
    }
    t = nbx_block_sum(t, red);
    if (threadIdx.x == 0) rho[l] = ndm == 1 ? 2.0 * t : t;  // (one density given = the closed-shell D_a = D_b)
}

// j[e] (+)= sum_{l < ls} rho[l] * b[l][e]       two doubles per thread, the slab streamed once
__global__ __launch_bounds__(256) void df_j_kernel(const double* __restrict__ b, const double* __restrict__ rho, int ls,
                                                   int64_t n2, double* __restrict__ j, int accumulate) {
    __shared__ double r[DF_LS];
    if (threadIdx.x < ls) r[threadIdx.x] = rho[threadIdx.x];
    __syncthreads();
    const int64_t e = ((int64_t)blockIdx.x * 256 + threadIdx.x) * 2;
    if (e >= n2) return;
    const bool two = e + 1 < n2;
    double2 acc = make_double2(0.0, 0.0);
    for (int l = 0; l < ls; ++l) {
        const double* p = b + (int64_t)l * n2 + e;
        if (two) {
            const double2 v = *reinterpret_cast<const double2*>(p);
            acc.x = fma(r[l], v.x, acc.x);
            acc.y = fma(r[l], v.y, acc.y);
        } else {
            acc.x = fma(r[l], p[0], acc.x);
        }
    }
    if (accumulate) {
        acc.x += j[e];
        if (two) acc.y += j[e + 1];
    }
    j[e] = acc.x;
    if (two) j[e + 1] = acc.y;
}

// b[l - l0][p][q] = scale * val(stream 9, l * npair + tri(p, q)), l in [l0, l1): symmetric in (p, q) by construction
__global__ __launch_bounds__(256) void df_synth_kernel(double* __restrict__ b, int N, int64_t l0, uint64_t seed, double scale) {
    const int64_t n2 = (int64_t)N * N;
    const uint64_t npair = (uint64_t)N * (N + 1) / 2;
    const int64_t l = l0 + blockIdx.y;
    double* dst = b + (int64_t)blockIdx.y * n2;
    for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < n2; e += (int64_t)gridDim.x * 256) {
        const uint32_t p = (uint32_t)(e / N), q = (uint32_t)(e - (int64_t)p * N);
        dst[e] = scale * nbx_synth_val(9, (uint64_t)l * npair + nbx_tri_u32(p, q), seed);
    }
}

// ct[i][q] = c[q][i], i < nocc, q < N (c: N x N row-major, its first nocc columns)
__global__ void df_transpose_occ_kernel(const double* __restrict__ c, int N, int nocc, double* __restrict__ ct) {
    __shared__ double t[32][33];
    const int q0 = blockIdx.x * 32, i0 = blockIdx.y * 32;
    for (int r = threadIdx.y; r < 32; r += 8) {
        const int q = q0 + r, i = i0 + threadIdx.x;
        t[r][threadIdx.x] = (q < N && i < nocc) ? c[(int64_t)q * N + i] : 0.0;
    }
    __syncthreads();
    for (int r = threadIdx.y; r < 32; r += 8) {
        const int i = i0 + r, q = q0 + threadIdx.x;
        if (i < nocc && q < N) ct[(int64_t)i * N + q] = t[threadIdx.x][r];
    }
}

size_t df_align(size_t x) { return (x + 255) & ~(size_t)255; }

struct DfPlan {
    size_t yt_off, ct_off, rho_off, total;
    int64_t yt_spin, ct_spin;
};
DfPlan df_plan(int64_t N, int64_t nocc_max, int64_t ndm) {
    DfPlan pl;
    pl.yt_spin = DF_LS * nocc_max * N;
    pl.ct_spin = nocc_max * N;
    size_t off = 0;
    pl.yt_off = off; off += df_align((size_t)(ndm * pl.yt_spin) * sizeof(double));
    pl.ct_off = off; off += df_align((size_t)(ndm * pl.ct_spin) * sizeof(double));
    pl.rho_off = off; off += df_align((size_t)DF_LS * sizeof(double));
    pl.total = off;
    return pl;
}

}  // namespace

size_t nbx_jk_df_worksize(int64_t nao, int64_t ndm, int64_t nocc_max) {
    if (nao <= 0 || ndm < 1 || ndm > 2 || nocc_max < 0 || nocc_max > nao) return 0;
    return df_plan(nao, nocc_max > 0 ? nocc_max : 1, ndm).total;
}

int nbx_df_synth(nbx_ctx* ctx, int64_t nao, int64_t l0, int64_t l1, uint64_t seed, double scale, double* d_b) {
    NBX_CHECK_ARG(ctx && d_b && nao > 0 && nao < 92681 && l0 >= 0 && l1 >= l0 && l1 - l0 <= 65535);
    NBX_CHECK_ARG((uint64_t)l1 * ((uint64_t)nao * (nao + 1) / 2) < (1ull << 48));
    if (l1 == l0) return NBX_OK;
    const int64_t n2 = nao * nao;
    const unsigned gx = (unsigned)(nbx_cdiv(n2, 256) < 4096 ? nbx_cdiv(n2, 256) : 4096);
    hipLaunchKernelGGL(df_synth_kernel, dim3(gx, (unsigned)(l1 - l0)), dim3(256), 0, ctx->stream, d_b, (int)nao, l0, seed, scale);
    NBX_LAUNCH_CHECK();
    return NBX_OK;
}

// d_b: (naux, N, N) symmetric matrices (this rank's auxiliary functions); d_c: (ndm, N, N) orbital coefficients, the first
// nocc[x] COLUMNS of d_c[x] are the occupied orbitals of spin x (ndm = 1: D_a = D_b = C_occ C_occ^T, one K);
// d_jk: (1 + ndm, N, N) = J, K_a[, K_b] -- partial results if d_b is a slab of the auxiliary basis.
int nbx_jk_df(nbx_ctx* ctx, int64_t nao, int64_t naux, const double* d_b, int64_t ndm, const double* d_c, const int64_t* nocc,
              double* d_jk, void* d_work, size_t work_bytes) {
    NBX_CHECK_ARG(ctx && d_b && d_c && nocc && d_jk && d_work && nao > 0 && naux >= 0 && (ndm == 1 || ndm == 2));
    const int64_t N = nao, n2 = N * N;
    int64_t nocc_max = 0;
    for (int x = 0; x < ndm; ++x) {
        NBX_CHECK_ARG(nocc[x] >= 0 && nocc[x] <= N);
        nocc_max = nocc[x] > nocc_max ? nocc[x] : nocc_max;
    }
    const DfPlan pl = df_plan(N, nocc_max > 0 ? nocc_max : 1, ndm);
    if (work_bytes < pl.total) {
        nbx_set_error("nbx_jk_df: workspace %zu < %zu bytes", work_bytes, pl.total);
        return NBX_E_NOMEM;
    }
    int rc = nbx_memset(ctx, d_jk, 0, (size_t)((1 + ndm) * n2) * sizeof(double));
    if (rc != NBX_OK) return rc;
    if (naux == 0 || nocc_max == 0) return NBX_OK;
    char* base = static_cast<char*>(d_work);
    double* yt = reinterpret_cast<double*>(base + pl.yt_off);
    double* ct = reinterpret_cast<double*>(base + pl.ct_off);
    double* rho = reinterpret_cast<double*>(base + pl.rho_off);
    // rows of Yt past a spin's own occupied count must read as zero in the two-spin K product (k = LS nocc_max)
    rc = nbx_memset(ctx, yt, 0, (size_t)(ndm * pl.yt_spin) * sizeof(double));
    if (rc != NBX_OK) return rc;
    // Ct[x] = C_occ^x^T (nocc x N): the first nocc columns of C, transposed
    for (int x = 0; x < ndm; ++x) {
        if (nocc[x] == 0) continue;
        hipLaunchKernelGGL(df_transpose_occ_kernel, dim3((unsigned)nbx_cdiv(N, 32), (unsigned)nbx_cdiv(nocc[x], 32)), dim3(32, 8), 0,
                           ctx->stream, d_c + x * n2, (int)N, (int)nocc[x], ct + x * pl.ct_spin);
        NBX_LAUNCH_CHECK();
    }
    for (int64_t l0 = 0; l0 < naux; l0 += DF_LS) {
        const int64_t ls = naux - l0 < DF_LS ? naux - l0 : DF_LS;
        const double* bs = d_b + l0 * n2;
        for (int x = 0; x < ndm; ++x) {
            if (nocc[x] == 0) continue;
            // Yt_l = C_occ^T B_l, l in the slab: A = C (k = N rows, the first nocc columns), shared by the batch
            rc = nbx_gemm(ctx, 'T', 'N', nocc[x], N, N, 1.0, d_c + x * n2, N, 0, bs, N, n2, 0.0, yt + x * pl.yt_spin, N,
                          nocc_max * N, ls);
            if (rc != NBX_OK) return rc;
        }
        hipLaunchKernelGGL(df_rho_kernel, dim3((unsigned)ls), dim3(256), 0, ctx->stream, yt, pl.yt_spin, nocc_max * N, ct, pl.ct_spin,
                           (int)ndm, (int)nocc[0], (int)(ndm > 1 ? nocc[1] : 0), (int)N, rho);
        NBX_LAUNCH_CHECK();
        // K^x += Yt^T Yt over the slab's ls * nocc_max rows (those past a spin's occupied count are zero), both spins
        rc = nbx_gemm(ctx, 'T', 'N', N, N, ls * nocc_max, 1.0, yt, N, pl.yt_spin, yt, N, pl.yt_spin, l0 > 0 ? 1.0 : 0.0, d_jk + n2, N, n2,
                      ndm);
        if (rc != NBX_OK) return rc;
        hipLaunchKernelGGL(df_j_kernel, dim3((unsigned)nbx_cdiv(n2, 512)), dim3(256), 0, ctx->stream, bs, rho, (int)ls, n2, d_jk,
                           l0 > 0 ? 1 : 0);
        NBX_LAUNCH_CHECK();
    }
    return NBX_OK;
}
