// libnbx: the exchange-correlation quadrature of the embedding-potential producer -- what the
// reference gets from PySCF's numint + libxc behind `dft.UKS.get_veff` (nbed/driver.py:155-191 global Kohn-Sham,
// :315-431 `_subsystem_dft`, :845-852 the embedding potential, :1138-1231 DFT-in-DFT).  One evaluation of
// (E_xc, v_xc) for a two-spin density matrix is three kernels over the stored AO values ao (G, nao) and gradients
// dao (3, G, nao) of the molecular grid (grid.hip makes them):
//
//   xc_rho_kernel      c = ao D on the matrix cores (v_mfma_f64_16x16x4_f64), tile by tile, BOTH spins from one pass
//                      over ao; the tile never leaves the registers: rho = sum_m c ao and grad rho = 2 sum_m c dao
//                      are reduced in the epilogue (dao is read exactly once)                      -- MFMA bound
//   xc_functional_kernel  one thread per grid point: energy density and its first derivatives with respect to
//                      (rho_a, rho_b, sigma_aa, sigma_ab, sigma_bb) written out analytically (Slater, Becke 88,
//                      VWN-RPA / VWN5, LYP in Miehlich's form: libxc's B3LYP = 0.08 S + 0.72 B88 + 0.19 VWN_RPA +
//                      0.81 LYP + 0.2 HF), quadrature weights folded in, E_xc and the electron count reduced in a
//                      fixed order                                                                  -- ALU (exp, log, cbrt)
//   xc_vmat_kernel     v[m][n] = sum_g ao[g][m] half[g][n],  half = v_rho / 2 ao + (2 v_ss grad rho_s + v_ab grad
//                      rho_o) . dao, with `half` built ON THE FLY as the B operand (it never exists in memory), split
//                      over chunks of grid points; xc_vmat_reduce_kernel adds the chunks in a fixed order and
//                      symmetrises (v + v^T)                                                        -- MFMA bound
#include "nbx_common.h"

namespace {

typedef double xc_v4 __attribute__((ext_vector_type(4)));

// ------------------------------------------------------------------------------------------------ density pass
// A workgroup = 4 waves = 64 grid points; wave w owns points g0 + 16 w .. + 16.  For every 16-column tile of
// c = ao D: the (nao x 16) slab of D (both spins) goes to LDS once per workgroup; A fragments ao[g][k] are kept
// in registers across the column tiles when nao <= 4 AREG.  C layout of the 16x16x4 product: acc[r] of lane
// (fk = lane >> 4, fr = lane & 15) is element (row fk + 4 r, column fr).
template <int AREG>
__global__ __launch_bounds__(256) void xc_rho_kernel(int64_t npts, int nao, const double* __restrict__ ao,
                                                     const double* __restrict__ dao, const double* __restrict__ dm,
                                                     double* __restrict__ rho, double* __restrict__ grad) {
    extern __shared__ __attribute__((aligned(16))) double smem[];
    const int KP = (nao + 3) & ~3, NT = (nao + 15) >> 4, NK = KP >> 2;
    double* ds0 = smem;             // [KP][16]  D_alpha[:, tile]
    double* ds1 = smem + KP * 16;   // [KP][16]  D_beta
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, fr = lane & 15, fk = lane >> 4;
    const int64_t gw = (int64_t)blockIdx.x * 64 + wave * 16;
    const int64_t plane = npts * (int64_t)nao, n2 = (int64_t)nao * nao;
    const int64_t ga = gw + fr;  // the A-fragment row of this lane
    const bool a_ok = ga < npts;
    const double* arow = ao + (a_ok ? ga : 0) * nao;
    double afr[AREG > 0 ? AREG : 1];
    if (AREG > 0) {
#pragma unroll
        for (int j = 0; j < AREG; ++j) {
            const int k = 4 * j + fk;
            afr[j] = (a_ok && k < nao) ? arow[k] : 0.0;
        }
    }
    double pr[2][4], px[2][4], py[2][4], pz[2][4];
#pragma unroll
    for (int x = 0; x < 2; ++x)
#pragma unroll
        for (int r = 0; r < 4; ++r) pr[x][r] = px[x][r] = py[x][r] = pz[x][r] = 0.0;

    for (int ct = 0; ct < NT; ++ct) {
        __syncthreads();  // the previous tile's fragments have been read
        for (int e = threadIdx.x; e < KP * 16; e += 256) {
            const int k = e >> 4, c = 16 * ct + (e & 15);
            const bool in = k < nao && c < nao;
            ds0[e] = in ? dm[(int64_t)k * nao + c] : 0.0;
            ds1[e] = in ? dm[n2 + (int64_t)k * nao + c] : 0.0;
        }
        __syncthreads();
        xc_v4 acc0 = (xc_v4){0.0, 0.0, 0.0, 0.0}, acc1 = (xc_v4){0.0, 0.0, 0.0, 0.0};
        if (AREG > 0) {
#pragma unroll
            for (int j = 0; j < AREG; ++j) {
                if (j < NK) {
                    const int k = 4 * j + fk;
                    acc0 = __builtin_amdgcn_mfma_f64_16x16x4f64(afr[j], ds0[k * 16 + fr], acc0, 0, 0, 0);
                    acc1 = __builtin_amdgcn_mfma_f64_16x16x4f64(afr[j], ds1[k * 16 + fr], acc1, 0, 0, 0);
                }
            }
        } else {
            for (int j = 0; j < NK; ++j) {
                const int k = 4 * j + fk;
                const double a = (a_ok && k < nao) ? arow[k] : 0.0;
                acc0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, ds0[k * 16 + fr], acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, ds1[k * 16 + fr], acc1, 0, 0, 0);
            }
        }
        // epilogue: this tile of c against the same tile of ao / dao
        const int m = 16 * ct + fr;
        if (m < nao) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int64_t g = gw + fk + 4 * r;
                if (g < npts) {
                    const int64_t off = g * nao + m;
                    const double v0 = ao[off], vx = dao[off], vy = dao[plane + off], vz = dao[2 * plane + off];
                    pr[0][r] = fma(acc0[r], v0, pr[0][r]);
                    px[0][r] = fma(acc0[r], vx, px[0][r]);
                    py[0][r] = fma(acc0[r], vy, py[0][r]);
                    pz[0][r] = fma(acc0[r], vz, pz[0][r]);
                    pr[1][r] = fma(acc1[r], v0, pr[1][r]);
                    px[1][r] = fma(acc1[r], vx, px[1][r]);
                    py[1][r] = fma(acc1[r], vy, py[1][r]);
                    pz[1][r] = fma(acc1[r], vz, pz[1][r]);
                }
            }
        }
    }
    // sum over the 16 lanes (columns) that share the rows fk + 4 r: xor shuffles stay inside the 16-lane group
#pragma unroll
    for (int x = 0; x < 2; ++x)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            double a = pr[x][r], b = px[x][r], c = py[x][r], d = pz[x][r];
#pragma unroll
            for (int o = 8; o > 0; o >>= 1) {
                a += __shfl_xor(a, o, 64);
                b += __shfl_xor(b, o, 64);
                c += __shfl_xor(c, o, 64);
                d += __shfl_xor(d, o, 64);
            }
            const int64_t g = gw + fk + 4 * r;
            if (fr == 0 && g < npts) {
                rho[x * npts + g] = a;
                grad[(3 * x + 0) * npts + g] = 2.0 * b;
                grad[(3 * x + 1) * npts + g] = 2.0 * c;
                grad[(3 * x + 2) * npts + g] = 2.0 * d;
            }
        }
}

// ------------------------------------------------------------------------------------------------ functionals
// Each piece returns the energy per volume and adds its first derivatives, scaled by `s`, to the five outputs.
struct XcDer {
    double e, va, vb, vaa, vab, vbb;
};

__device__ __forceinline__ void xc_slater(double s, double ra, double rb, XcDer& o) {
    const double cx = 0.9305257363491002;  // (3/2) (3 / (4 pi))^(1/3)
    const double ca = cbrt(ra), cb = cbrt(rb);
    o.e += s * (-cx * (ca * ra + cb * rb));
    o.va += s * (-(4.0 / 3.0) * cx * ca);
    o.vb += s * (-(4.0 / 3.0) * cx * cb);
}

// Becke's 1988 gradient correction for one spin channel: e = -beta rho^(4/3) g(x), g = x^2 / (1 + 6 beta x asinh x),
// x = sqrt(sigma) / rho^(4/3);  de/drho = -(4/3) beta rho^(1/3) (g - x g'),  de/dsigma = -beta (g'/x) / (2 rho^(4/3))
__device__ __forceinline__ void xc_b88(double s, double r, double sg, double& e, double& vr, double& vs) {
    const double beta = 0.0042;
    const double r13 = cbrt(r), r43 = r13 * r;
    const double x = sqrt(sg) / r43;
    const double ash = asinh(x);
    const double den = fma(6.0 * beta * x, ash, 1.0);
    const double g = x * x / den;
    const double gpx = (2.0 * den - 6.0 * beta * x * (ash + x / sqrt(fma(x, x, 1.0)))) / (den * den);
    e += s * (-beta * r43 * g);
    vr += s * (-(4.0 / 3.0) * beta * r13 * (g - x * x * gpx));
    vs += s * (-beta * gpx / (2.0 * r43));
}

// VWN's interpolation formula eps(x), x = sqrt(rs), and d eps / dx
__device__ __forceinline__ void xc_vwn_fit(double x, double a, double x0, double b, double c, double& e, double& de) {
    const double q = sqrt(4.0 * c - b * b);
    const double X = fma(x, x + b, c), X0 = fma(x0, x0 + b, c);
    const double tb = 2.0 * x + b;
    const double at = atan(q / tb);
    const double den = fma(tb, tb, q * q);
    const double xm = x - x0;
    e = a * (log(x * x / X) + 2.0 * b / q * at - b * x0 / X0 * (log(xm * xm / X) + 2.0 * (b + 2.0 * x0) / q * at));
    de = a * (2.0 / x - tb / X - 4.0 * b / den - b * x0 / X0 * (2.0 / xm - tb / X - 4.0 * (b + 2.0 * x0) / den));
}

// VWN correlation: RPA = false the fit to the Ceperley-Alder data with the paper's spin interpolation (libxc
// LDA_C_VWN: PySCF's "vwn"), RPA = true the fit to the RPA data (LDA_C_VWN_RPA: the one inside B3LYP)
template <bool RPA>
__device__ __forceinline__ void xc_vwn(double s, double ra, double rb, XcDer& o) {
    const double rho = ra + rb;
    const double zeta = (ra - rb) / rho;
    const double x = sqrt(cbrt(0.238732414637843 / rho));  // sqrt(rs), rs = (3 / (4 pi rho))^(1/3)
    const double c43 = 0.5198420997897464;                   // 2^(4/3) - 2
    const double cp = cbrt(1.0 + zeta), cm = cbrt(1.0 - zeta);
    const double fz = (cp * (1.0 + zeta) + cm * (1.0 - zeta) - 2.0) / c43;
    const double dfz = (4.0 / 3.0) * (cp - cm) / c43;
    double ep, dep, ef, def, eps, dx, dz;
    if (RPA) {
        xc_vwn_fit(x, 0.0310907, -0.409286, 13.0720, 42.7198, ep, dep);
        xc_vwn_fit(x, 0.01554535, -0.743294, 20.1231, 101.578, ef, def);
        eps = fma(fz, ef - ep, ep);
        dx = fma(fz, def - dep, dep);
        dz = dfz * (ef - ep);
    } else {
        double al, dal;
        xc_vwn_fit(x, 0.0310907, -0.10498, 3.72744, 12.9352, ep, dep);
        xc_vwn_fit(x, 0.01554535, -0.32500, 7.06042, 18.0578, ef, def);
        xc_vwn_fit(x, -0.01688686394038963, -0.0047584, 1.13107, 13.0045, al, dal);  // -1 / (6 pi^2): spin stiffness
        const double fpp0 = 1.7099209341613653;  // 4 / (9 (2^(1/3) - 1))
        const double z3 = zeta * zeta * zeta, z4 = z3 * zeta;
        eps = ep + al * fz / fpp0 * (1.0 - z4) + (ef - ep) * fz * z4;
        dx = dep + dal * fz / fpp0 * (1.0 - z4) + (def - dep) * fz * z4;
        dz = al / fpp0 * (dfz * (1.0 - z4) - 4.0 * fz * z3) + (ef - ep) * (dfz * z4 + 4.0 * fz * z3);
    }
    // e = rho eps(rs, zeta):  d/d rho_s = eps - (rs / 3) d eps / d rs  +-  (1 -+ zeta) d eps / d zeta,  d/d rs = (d/dx) / (2 x)
    const double common = eps - (x / 6.0) * dx;
    o.e += s * rho * eps;
    o.va += s * (common + (1.0 - zeta) * dz);
    o.vb += s * (common - (1.0 + zeta) * dz);
}

// Lee-Yang-Parr in the gradient-only form of Miehlich, Savin, Stoll and Preuss (CPL 157, 200 (1989)):
//   e = -4 a rho_a rho_b / (rho (1 + d r)) - a b omega [rho_a rho_b P + Q],   r = rho^(-1/3),
//   omega = exp(-c r) / (1 + d r) rho^(-11/3),  delta = c r + d r / (1 + d r),
//   d omega / d rho = omega (delta - 11) / (3 rho),  d delta / d rho = -(delta - (d r / (1 + d r))^2) / (3 rho)
__device__ __forceinline__ void xc_lyp(double s, double ra, double rb, double saa, double sab, double sbb, XcDer& o) {
    const double a = 0.04918, b = 0.132, c = 0.2533, d = 0.349;
    const double rho = ra + rb, rho2 = rho * rho;
    const double r = 1.0 / cbrt(rho);
    const double den = fma(d, r, 1.0);
    const double r113 = r * r / (rho2 * rho);  // rho^(-11/3) = rho^(-2/3) / rho^3
    const double omega = exp(-c * r) / den * r113;
    const double dr = d * r / den;
    const double delta = c * r + dr;
    const double k1 = 36.46239897876477;  // 2^(11/3) (3/10) (3 pi^2)^(2/3)
    const double st = saa + 2.0 * sab + sbb;
    const double ca = cbrt(ra), cb = cbrt(rb);
    const double ra53 = ca * ca * ra, rb53 = cb * cb * rb;  // rho^(5/3)
    const double mix = (ra * saa + rb * sbb) / rho;
    const double P = k1 * (ra53 * ra + rb53 * rb) + (47.0 / 18.0 - 7.0 * delta / 18.0) * st - (2.5 - delta / 18.0) * (saa + sbb) -
                     (delta - 11.0) / 9.0 * mix;
    const double Q = -2.0 / 3.0 * rho2 * st + (2.0 / 3.0 * rho2 - ra * ra) * sbb + (2.0 / 3.0 * rho2 - rb * rb) * saa;
    const double B = ra * rb * P + Q;
    const double ab = a * b;
    o.e += s * (-4.0 * a * ra * rb / (rho * den) - ab * omega * B);
    const double dom = omega * (delta - 11.0) / (3.0 * rho);
    const double ddel = -(delta - dr * dr) / (3.0 * rho);
    const double dPdd = -7.0 * st / 18.0 + (saa + sbb) / 18.0 - mix / 9.0;
    const double t1c = d * r / (3.0 * rho * den * den);
    {   // d / d rho_a
        const double dt1 = -4.0 * a * (rb / (rho * den) - ra * rb / (rho2 * den) + ra * rb / rho * t1c);
        const double dP = k1 * (8.0 / 3.0) * ra53 + ddel * dPdd - (delta - 11.0) / 9.0 * (saa - mix) / rho;
        const double dQ = -4.0 / 3.0 * rho * st + (4.0 / 3.0 * rho - 2.0 * ra) * sbb + 4.0 / 3.0 * rho * saa;
        o.va += s * (dt1 - ab * (dom * B + omega * (rb * P + ra * rb * dP + dQ)));
    }
    {   // d / d rho_b
        const double dt1 = -4.0 * a * (ra / (rho * den) - ra * rb / (rho2 * den) + ra * rb / rho * t1c);
        const double dP = k1 * (8.0 / 3.0) * rb53 + ddel * dPdd - (delta - 11.0) / 9.0 * (sbb - mix) / rho;
        const double dQ = -4.0 / 3.0 * rho * st + (4.0 / 3.0 * rho - 2.0 * rb) * saa + 4.0 / 3.0 * rho * sbb;
        o.vb += s * (dt1 - ab * (dom * B + omega * (ra * P + ra * rb * dP + dQ)));
    }
    const double abw = -ab * omega;
    o.vaa += s * abw * (ra * rb * (1.0 / 9.0 - delta / 3.0 - (delta - 11.0) / 9.0 * ra / rho) - rb * rb);
    o.vbb += s * abw * (ra * rb * (1.0 / 9.0 - delta / 3.0 - (delta - 11.0) / 9.0 * rb / rho) - ra * ra);
    o.vab += s * abw * (ra * rb * 2.0 * (47.0 / 18.0 - 7.0 * delta / 18.0) - 4.0 / 3.0 * rho2);
}

constexpr int XC_FN_THREADS = 256;

// code: NBX_XC_SLATER / _LDA_VWN_RPA / _LDA_VWN5 / _B3LYP (the semi-local part; the exact-exchange fraction is the
// caller's).  Conventions of nbed_amd.xc.XCProvider.__call__ (the torch expression this kernel replaces): densities
// clamped from below at rho_floor / 2, sigma_aa and sigma_bb lifted by 1e-40, points with rho_a + rho_b <= rho_floor
// dropped; outputs carry the quadrature weights:
//   vr[x][g]     = w keep dE/drho_x
//   vec[x][a][g] = w keep (2 dE/dsigma_xx grad rho_x + dE/dsigma_ab grad rho_other)[a]
//   part[blk]    = (sum w keep e, sum w (rho_a + rho_b)) of the block
__global__ __launch_bounds__(XC_FN_THREADS) void xc_functional_kernel(int code, int64_t npts, const double* __restrict__ rho,
                                                                      const double* __restrict__ grad,
                                                                      const double* __restrict__ w, double rho_floor,
                                                                      double* __restrict__ vr, double* __restrict__ vec,
                                                                      double* __restrict__ part) {
    __shared__ double red[17];
    const int64_t g = (int64_t)blockIdx.x * XC_FN_THREADS + threadIdx.x;
    double e_w = 0.0, n_w = 0.0;
    if (g < npts) {
        const double r0 = rho[g], r1 = rho[npts + g], wg = w[g];
        const double gax = grad[g], gay = grad[npts + g], gaz = grad[2 * npts + g];
        const double gbx = grad[3 * npts + g], gby = grad[4 * npts + g], gbz = grad[5 * npts + g];
        n_w = wg * (r0 + r1);
        const double keep = (r0 + r1) > rho_floor ? 1.0 : 0.0;
        const double ra = fmax(r0, 0.5 * rho_floor), rb = fmax(r1, 0.5 * rho_floor);
        const double saa = fma(gax, gax, fma(gay, gay, gaz * gaz)) + 1.0e-40;
        const double sbb = fma(gbx, gbx, fma(gby, gby, gbz * gbz)) + 1.0e-40;
        const double sab = fma(gax, gbx, fma(gay, gby, gaz * gbz));
        XcDer o = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
        if (code == NBX_XC_SLATER) {
            xc_slater(1.0, ra, rb, o);
        } else if (code == NBX_XC_LDA_VWN_RPA) {
            xc_slater(1.0, ra, rb, o);
            xc_vwn<true>(1.0, ra, rb, o);
        } else if (code == NBX_XC_LDA_VWN5) {
            xc_slater(1.0, ra, rb, o);
            xc_vwn<false>(1.0, ra, rb, o);
        } else {  // B3LYP
            xc_slater(0.8, ra, rb, o);
            xc_b88(0.72, ra, saa, o.e, o.va, o.vaa);
            xc_b88(0.72, rb, sbb, o.e, o.vb, o.vbb);
            xc_vwn<true>(0.19, ra, rb, o);
            xc_lyp(0.81, ra, rb, saa, sab, sbb, o);
        }
        const double wk = wg * keep;
        e_w = wk * o.e;
        vr[g] = wk * o.va;
        vr[npts + g] = wk * o.vb;
        const double a2 = 2.0 * wk * o.vaa, b2 = 2.0 * wk * o.vbb, ab = wk * o.vab;
        vec[g] = fma(a2, gax, ab * gbx);
        vec[npts + g] = fma(a2, gay, ab * gby);
        vec[2 * npts + g] = fma(a2, gaz, ab * gbz);
        vec[3 * npts + g] = fma(b2, gbx, ab * gax);
        vec[4 * npts + g] = fma(b2, gby, ab * gay);
        vec[5 * npts + g] = fma(b2, gbz, ab * gaz);
    }
    const double es = nbx_block_sum(e_w, red);
    const double ns = nbx_block_sum(n_w, red);
    if (threadIdx.x == 0) {
        part[2 * blockIdx.x] = es;
        part[2 * blockIdx.x + 1] = ns;
    }
}

// sums the block partials in a fixed order (one workgroup): out[0] = E_xc, out[1] = integrated electron count
__global__ __launch_bounds__(256) void xc_sums_kernel(int nblk, const double* __restrict__ part, double* __restrict__ out) {
    __shared__ double red[17];
    double e = 0.0, n = 0.0;
    for (int i = threadIdx.x; i < nblk; i += 256) {
        e += part[2 * i];
        n += part[2 * i + 1];
    }
    e = nbx_block_sum(e, red);
    n = nbx_block_sum(n, red);
    if (threadIdx.x == 0) {
        out[0] = e;
        out[1] = n;
    }
}

// ------------------------------------------------------------------------------------------------ potential pass
// Workgroup = BT waves; block (bm, bn) of the (nao x nao) result, 16 BT rows and columns; wave w owns tile row w of
// the block and all BT column tiles.  Grid points are consumed 16 at a time: the threads build the 16 x (16 BT)
// tile of `half` into LDS (double buffered: one barrier per step), every wave multiplies its ao^T fragments (read
// straight from global memory: 16 consecutive AO columns per grid point) with it.
template <int BT>
__global__ __launch_bounds__(64 * BT) void xc_vmat_kernel(int64_t npts, int nao, int64_t chunk, int nblk,
                                                          const double* __restrict__ ao, const double* __restrict__ dao,
                                                          const double* __restrict__ vr, const double* __restrict__ vec,
                                                          double* __restrict__ part) {
    constexpr int W = 16 * BT, LD = (W % 32 == 16) ? W : W + 16, NT = 64 * BT;
    __shared__ __attribute__((aligned(16))) double hs[2][16 * LD];
    const int x = blockIdx.z;
    const int bm = blockIdx.y / nblk, bn = blockIdx.y - bm * nblk;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, fr = lane & 15, fk = lane >> 4;
    const int64_t g_begin = (int64_t)blockIdx.x * chunk, g_end = min(npts, g_begin + chunk);
    const int64_t plane = npts * (int64_t)nao;
    const double* vrx = vr + (int64_t)x * npts;
    const double* vcx = vec + (int64_t)(3 * x) * npts;
    const int m = W * bm + 16 * wave + fr;  // the A-fragment column of ao (row of the result) of this lane
    const bool m_ok = m < nao;
    xc_v4 acc[BT];
#pragma unroll
    for (int c = 0; c < BT; ++c) acc[c] = (xc_v4){0.0, 0.0, 0.0, 0.0};

    auto produce = [&](int buf, int64_t g0) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {  // 16 W elements over NT = 4 W threads
            const int e = threadIdx.x + i * NT;
            const int gl = e / W, nl = e - gl * W;
            const int64_t g = g0 + gl;
            const int n = W * bn + nl;
            double h = 0.0;
            if (g < g_end && n < nao) {
                const int64_t off = g * nao + n;
                h = fma(0.5 * vrx[g], ao[off],
                        fma(vcx[g], dao[off], fma(vcx[npts + g], dao[plane + off], vcx[2 * npts + g] * dao[2 * plane + off])));
            }
            hs[buf][gl * LD + nl] = h;
        }
    };

    int buf = 0;
    if (g_begin < g_end) produce(0, g_begin);
    for (int64_t g0 = g_begin; g0 < g_end; g0 += 16) {
        __syncthreads();  // hs[buf] is complete; hs[buf ^ 1] has been consumed
        if (g0 + 16 < g_end) produce(buf ^ 1, g0 + 16);
        double a[4];
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
            const int64_t g = g0 + 4 * kk + fk;
            a[kk] = (m_ok && g < g_end) ? ao[g * nao + m] : 0.0;
        }
#pragma unroll
        for (int kk = 0; kk < 4; ++kk)
#pragma unroll
            for (int c = 0; c < BT; ++c)
                acc[c] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[kk], hs[buf][(4 * kk + fk) * LD + 16 * c + fr], acc[c], 0, 0, 0);
        buf ^= 1;
    }
    // this chunk's contribution to the block: part[chunk][x][row][col]
    double* out = part + ((int64_t)blockIdx.x * 2 + x) * (int64_t)nao * nao;
#pragma unroll
    for (int c = 0; c < BT; ++c) {
        const int col = W * bn + 16 * c + fr;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int row = W * bm + 16 * wave + fk + 4 * r;
            if (row < nao && col < nao) out[(int64_t)row * nao + col] = acc[c][r];
        }
    }
}

// v[x][m][n] = sum_chunks (part[c][x][m][n] + part[c][x][n][m]), chunks in ascending order
__global__ __launch_bounds__(256) void xc_vmat_reduce_kernel(int nao, int nchunk, const double* __restrict__ part,
                                                             double* __restrict__ v) {
    const int64_t n2 = (int64_t)nao * nao;
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= 2 * n2) return;
    const int x = (int)(i / n2);
    const int64_t e = i - x * n2;
    const int m = (int)(e / nao), n = (int)(e - (int64_t)m * nao);
    const int64_t et = (int64_t)n * nao + m;
    double s = 0.0;
    for (int c = 0; c < nchunk; ++c) {
        const double* p = part + ((int64_t)c * 2 + x) * n2;
        s += p[e] + p[et];
    }
    v[i] = s;
}

int vmat_bt(int64_t nao) {  // tiles per block side: the least padded 16 BT grid over nao, the larger BT on a tie
    int best = 1;
    int64_t best_pad = -1;
    for (int bt = 1; bt <= 5; ++bt) {
        const int64_t w = 16 * bt, pad = nbx_cdiv(nao, w) * w;
        if (best_pad < 0 || pad <= best_pad) {
            best = bt;
            best_pad = pad;
        }
    }
    return best;
}

int64_t vmat_chunk(int64_t npts, int64_t nao) {  // grid points per workgroup: enough workgroups to fill the chip
    const int bt = vmat_bt(nao);
    const int64_t nblk = nbx_cdiv(nao, 16 * bt);
    int64_t chunk = 2048;
    while (chunk > 256 && nbx_cdiv(npts, chunk) * nblk * nblk * 2 < 1024) chunk >>= 1;
    return chunk;
}

size_t xc_align256(size_t x) { return (x + 255) & ~(size_t)255; }

}  // namespace

extern "C" int nbx_xc_rho(nbx_ctx* ctx, int64_t npts, int64_t nao, const double* d_ao, const double* d_dao,
                          const double* d_dm, double* d_rho, double* d_grad) {
    NBX_CHECK_ARG(ctx && d_ao && d_dao && d_dm && d_rho && d_grad && npts >= 0 && nao >= 1);
    const int64_t kp = (nao + 3) & ~3ll;
    const size_t lds = (size_t)(2 * kp * 16) * sizeof(double);
    if (lds > 160 * 1024) {
        nbx_set_error("nbx_xc_rho: nao = %lld beyond the LDS-resident density slab (limit 640)", (long long)nao);
        return NBX_E_UNSUPPORTED;
    }
    if (npts == 0) return NBX_OK;
    const dim3 grid((unsigned)nbx_cdiv(npts, 64));
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&xc_rho_kernel<40>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                  160 * 1024);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&xc_rho_kernel<0>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                  160 * 1024);
        attr_set = true;
    }
    if (kp <= 160)
        hipLaunchKernelGGL(xc_rho_kernel<40>, grid, dim3(256), lds, ctx->stream, npts, (int)nao, d_ao, d_dao, d_dm, d_rho, d_grad);
    else
        hipLaunchKernelGGL(xc_rho_kernel<0>, grid, dim3(256), lds, ctx->stream, npts, (int)nao, d_ao, d_dao, d_dm, d_rho, d_grad);
    NBX_LAUNCH_CHECK();
    return NBX_OK;
}

extern "C" size_t nbx_xc_functional_worksize(int64_t npts) {
    return npts <= 0 ? 256 : xc_align256((size_t)(2 * nbx_cdiv(npts, XC_FN_THREADS)) * sizeof(double));
}

extern "C" int nbx_xc_functional(nbx_ctx* ctx, int code, int64_t npts, const double* d_rho, const double* d_grad,
                                 const double* d_w, double rho_floor, double* d_vr, double* d_vec, double* d_sums, void* d_work,
                                 size_t work_bytes) {
    NBX_CHECK_ARG(ctx && d_rho && d_grad && d_w && d_vr && d_vec && d_sums && d_work && npts >= 0);
    NBX_CHECK_ARG(code == NBX_XC_SLATER || code == NBX_XC_LDA_VWN_RPA || code == NBX_XC_LDA_VWN5 || code == NBX_XC_B3LYP);
    NBX_CHECK_ARG(rho_floor > 0.0 && work_bytes >= nbx_xc_functional_worksize(npts));
    if (npts == 0) return nbx_memset(ctx, d_sums, 0, 2 * sizeof(double));
    const int64_t nblk = nbx_cdiv(npts, XC_FN_THREADS);
    NBX_CHECK_ARG(nblk < (1ll << 30));
    double* part = static_cast<double*>(d_work);
    hipLaunchKernelGGL(xc_functional_kernel, dim3((unsigned)nblk), dim3(XC_FN_THREADS), 0, ctx->stream, code, npts, d_rho,
                       d_grad, d_w, rho_floor, d_vr, d_vec, part);
    NBX_LAUNCH_CHECK();
    hipLaunchKernelGGL(xc_sums_kernel, dim3(1), dim3(256), 0, ctx->stream, (int)nblk, part, d_sums);
    NBX_LAUNCH_CHECK();
    return NBX_OK;
}

extern "C" size_t nbx_xc_vmat_worksize(int64_t npts, int64_t nao) {
    if (npts <= 0 || nao <= 0) return 256;
    const int64_t nchunk = nbx_cdiv(npts, vmat_chunk(npts, nao));
    return xc_align256((size_t)(nchunk * 2 * nao * nao) * sizeof(double));
}

extern "C" int nbx_xc_vmat(nbx_ctx* ctx, int64_t npts, int64_t nao, const double* d_ao, const double* d_dao,
                           const double* d_vr, const double* d_vec, double* d_vxc, void* d_work, size_t work_bytes) {
    NBX_CHECK_ARG(ctx && d_ao && d_dao && d_vr && d_vec && d_vxc && npts >= 0 && nao >= 1 && nao <= (1 << 15));
    const int64_t n2 = nao * nao;
    if (npts == 0) return nbx_memset(ctx, d_vxc, 0, (size_t)(2 * n2) * sizeof(double));
    NBX_CHECK_ARG(d_work && work_bytes >= nbx_xc_vmat_worksize(npts, nao));
    const int bt = vmat_bt(nao);
    const int64_t chunk = vmat_chunk(npts, nao), nchunk = nbx_cdiv(npts, chunk), nblk = nbx_cdiv(nao, 16 * bt);
    NBX_CHECK_ARG(nblk * nblk <= 65535 && nchunk < (1ll << 31));
    double* part = static_cast<double*>(d_work);
    const dim3 grid((unsigned)nchunk, (unsigned)(nblk * nblk), 2);
#define NBX_VMAT(BT_)                                                                                                       \
    hipLaunchKernelGGL(xc_vmat_kernel<BT_>, grid, dim3(64 * BT_), 0, ctx->stream, npts, (int)nao, chunk, (int)nblk, d_ao, d_dao, \
                       d_vr, d_vec, part)
    switch (bt) {
        case 1: NBX_VMAT(1); break;
        case 2: NBX_VMAT(2); break;
        case 3: NBX_VMAT(3); break;
        case 4: NBX_VMAT(4); break;
        default: NBX_VMAT(5); break;
    }
#undef NBX_VMAT
    NBX_LAUNCH_CHECK();
    hipLaunchKernelGGL(xc_vmat_reduce_kernel, dim3((unsigned)nbx_cdiv(2 * n2, 256)), dim3(256), 0, ctx->stream, (int)nao,
                       (int)nchunk, part, d_vxc);
    NBX_LAUNCH_CHECK();
    return NBX_OK;
}
