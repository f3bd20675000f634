// libnbx: the two point-wise producers of the exchange-correlation quadrature (SURVEY 8 row f3 -- what the
// reference gets from PySCF's `dft.gen_grid` and `numint.eval_ao` behind `scf.UKS(...)`, nbed/driver.py:86-104,
// 315-431): Becke's cell weights of the molecular grid and the values / gradients of the contracted Cartesian
// Gaussians on it.  Both are one thread per grid point, no communication: bound by fp64 exp / sqrt throughput and
// by the (G x nao) stores.
#include "nbx_common.h"

namespace {

// share[g] = w_owner(r_g) / sum_i w_i(r_g),  w_i = prod_{j != i} s(mu_ij),  mu_ij = (r_i - r_j) / R_ij,
// nu = mu + a_ij (1 - mu^2), s = (1 - p(p(p(nu)))) / 2, p(x) = (3 x - x^3) / 2          (Becke, JCP 88, 2547)
// TABLES_IN_LDS: the two (natm x natm) tables fit the LDS (natm <= 96); otherwise they are read from L2
template <bool TABLES_IN_LDS>
__global__ __launch_bounds__(256) void becke_share_kernel(int64_t npts, const double* __restrict__ pts, int natm,
                                                          const double* __restrict__ centres,
                                                          const double* __restrict__ aij,
                                                          const double* __restrict__ inv_dist, int owner,
                                                          double* __restrict__ share) {
    extern __shared__ __attribute__((aligned(16))) double smem[];
    double* cs = smem;                 // [natm][3]
    const double* as = aij;            // [natm][natm]
    const double* rs = inv_dist;       // [natm][natm]
    for (int i = threadIdx.x; i < 3 * natm; i += blockDim.x) cs[i] = centres[i];
    if (TABLES_IN_LDS) {
        double* as_l = cs + 3 * natm;
        double* rs_l = as_l + natm * natm;
        for (int i = threadIdx.x; i < natm * natm; i += blockDim.x) {
            as_l[i] = aij[i];
            rs_l[i] = inv_dist[i];
        }
        as = as_l;
        rs = rs_l;
    }
    __syncthreads();
    const int64_t g = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= npts) return;
    const double x = pts[3 * g], y = pts[3 * g + 1], z = pts[3 * g + 2];
    double tot = 0.0, mine = 0.0;
    for (int i = 0; i < natm; ++i) {
        const double dxi = x - cs[3 * i], dyi = y - cs[3 * i + 1], dzi = z - cs[3 * i + 2];
        const double ri = sqrt(fma(dxi, dxi, fma(dyi, dyi, dzi * dzi)));
        double cell = 1.0;
        for (int j = 0; j < natm; ++j) {
            if (j == i) continue;
            const double dxj = x - cs[3 * j], dyj = y - cs[3 * j + 1], dzj = z - cs[3 * j + 2];
            const double rj = sqrt(fma(dxj, dxj, fma(dyj, dyj, dzj * dzj)));
            const double mu = (ri - rj) * rs[i * natm + j];
            double f = fma(as[i * natm + j], 1.0 - mu * mu, mu);
#pragma unroll
            for (int it = 0; it < 3; ++it) f = 1.5 * f - 0.5 * f * f * f;
            cell *= 0.5 * (1.0 - f);
        }
        tot += cell;
        if (i == owner) mine = cell;
    }
    share[g] = mine / tot;
}

constexpr int AO_MAX_PRIM = 24;

// One thread per grid point, all shells: Cartesian components x^l y^m z^n sum_k c_k exp(-a_k r^2) and their
// gradients.  shell_i[s] = {first component, number of components, first primitive, number of primitives};
// comp_lmn[c] = {l, m, n, offset of the component's coefficients}; out (G, ncart) row-major, dout (3, G, ncart).
__global__ __launch_bounds__(256) void eval_ao_kernel(int64_t npts, const double* __restrict__ pts, int nshell,
                                                      const int* __restrict__ shell_i,
                                                      const double* __restrict__ shell_centre,
                                                      const int* __restrict__ comp_lmn, const double* __restrict__ exps,
                                                      const double* __restrict__ coefs, int ncart,
                                                      double* __restrict__ out, double* __restrict__ dout) {
    const int64_t g = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= npts) return;
    const double px = pts[3 * g], py = pts[3 * g + 1], pz = pts[3 * g + 2];
    double* row = out + g * ncart;
    const int64_t plane = npts * (int64_t)ncart;
    double* drow = dout ? dout + g * ncart : nullptr;
    for (int s = 0; s < nshell; ++s) {
        const int c0 = shell_i[4 * s], nc = shell_i[4 * s + 1], k0 = shell_i[4 * s + 2], nk = shell_i[4 * s + 3];
        const double dx = px - shell_centre[3 * s], dy = py - shell_centre[3 * s + 1], dz = pz - shell_centre[3 * s + 2];
        const double r2 = fma(dx, dx, fma(dy, dy, dz * dz));
        double ex[AO_MAX_PRIM];
#pragma unroll
        for (int k = 0; k < AO_MAX_PRIM; ++k) ex[k] = k < nk ? exp(-exps[k0 + k] * r2) : 0.0;
        // powers 0..3 of the displacements (l <= 3) and the power one lower times its exponent
        const double xp[4] = {1.0, dx, dx * dx, dx * dx * dx};
        const double yp[4] = {1.0, dy, dy * dy, dy * dy * dy};
        const double zp[4] = {1.0, dz, dz * dz, dz * dz * dz};
        for (int c = 0; c < nc; ++c) {
            const int l = comp_lmn[4 * (c0 + c)], m = comp_lmn[4 * (c0 + c) + 1], n = comp_lmn[4 * (c0 + c) + 2];
            const double* cf = coefs + comp_lmn[4 * (c0 + c) + 3];
            double rad = 0.0, drad = 0.0;
#pragma unroll
            for (int k = 0; k < AO_MAX_PRIM; ++k) {
                if (k < nk) {
                    const double t = cf[k] * ex[k];
                    rad += t;
                    drad = fma(-2.0 * exps[k0 + k], t, drad);
                }
            }
            const double fx = xp[l], fy = yp[m], fz = zp[n];
            const double poly = fx * fy * fz;
            row[c0 + c] = poly * rad;
            if (drow) {
                const double pd = poly * drad;
                const double gx = l ? (double)l * xp[l - 1] * fy * fz * rad : 0.0;
                const double gy = m ? (double)m * fx * yp[m - 1] * fz * rad : 0.0;
                const double gz = n ? (double)n * fx * fy * zp[n - 1] * rad : 0.0;
                drow[c0 + c] = fma(pd, dx, gx);
                drow[plane + c0 + c] = fma(pd, dy, gy);
                drow[2 * plane + c0 + c] = fma(pd, dz, gz);
            }
        }
    }
}

}  // namespace

extern "C" int nbx_becke_share(nbx_ctx* ctx, int64_t npts, const double* d_pts, int64_t natm, const double* d_centres,
                               const double* d_aij, const double* d_inv_dist, int64_t owner, double* d_share) {
    NBX_CHECK_ARG(ctx && d_pts && d_centres && d_aij && d_inv_dist && d_share && npts >= 0);
    NBX_CHECK_ARG(natm >= 1 && natm <= 4096 && owner >= 0 && owner < natm);
    if (npts == 0) return NBX_OK;
    const dim3 grid((unsigned)nbx_cdiv(npts, 256));
    if (natm <= 96) {  // 3 natm + 2 natm^2 doubles of LDS (<= 148 KB)
        const size_t lds = (size_t)(3 * natm + 2 * natm * natm) * sizeof(double);
        static bool attr_set = false;
        if (!attr_set) {
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&becke_share_kernel<true>),
                                      hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
            attr_set = true;
        }
        hipLaunchKernelGGL(becke_share_kernel<true>, grid, dim3(256), lds, ctx->stream, npts, d_pts, (int)natm,
                           d_centres, d_aij, d_inv_dist, (int)owner, d_share);
    } else {
        hipLaunchKernelGGL(becke_share_kernel<false>, grid, dim3(256), (size_t)(3 * natm) * sizeof(double), ctx->stream,
                           npts, d_pts, (int)natm, d_centres, d_aij, d_inv_dist, (int)owner, d_share);
    }
    NBX_LAUNCH_CHECK();
    return NBX_OK;
}

extern "C" int nbx_eval_ao(nbx_ctx* ctx, int64_t npts, const double* d_pts, int64_t nshell, const int* d_shell_i,
                           const double* d_shell_centre, const int* d_comp_lmn, const double* d_exps,
                           const double* d_coefs, int64_t ncart, int64_t max_prim, double* d_out, double* d_dout) {
    NBX_CHECK_ARG(ctx && d_pts && d_shell_i && d_shell_centre && d_comp_lmn && d_exps && d_coefs && d_out);
    NBX_CHECK_ARG(npts >= 0 && nshell >= 1 && ncart >= 1 && ncart <= (1 << 20));
    if (max_prim < 1 || max_prim > AO_MAX_PRIM) {
        nbx_set_error("nbx_eval_ao: %lld primitives in a shell (limit %d)", (long long)max_prim, AO_MAX_PRIM);
        return NBX_E_UNSUPPORTED;
    }
    if (npts == 0) return NBX_OK;
    hipLaunchKernelGGL(eval_ao_kernel, dim3((unsigned)nbx_cdiv(npts, 256)), dim3(256), 0, ctx->stream, npts, d_pts,
                       (int)nshell, d_shell_i, d_shell_centre, d_comp_lmn, d_exps, d_coefs, (int)ncart, d_out, d_dout);
    NBX_LAUNCH_CHECK();
    return NBX_OK;
}
