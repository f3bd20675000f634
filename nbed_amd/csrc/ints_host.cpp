// Two-electron AO integrals of a real molecule, on the host cores.
//
// The reference gets (pq|rs) from PySCF/libcint (gto.Mole.intor("int2e"), reached through
// scf.UKS(mol).kernel() and ao2mo at nbed/driver.py:86-191, nbed/ham_builder.py:139-170).  They are the
// INPUT of the embedded-SCF hot path, produced once per molecule, so -- like libcint -- this is host code:
// McMurchie-Davidson over contracted shells of angular momentum <= 3, the shell quartets of the
// eight-fold unique set spread over a pool of threads.  The product's Python engine
// (nbed_amd/integrals.py) evaluates the same scheme shell pair by shell pair in numpy; this one exists
// because a 148-function molecule (octane / 6-31G*, the configuration BASELINE.json's metric is quoted
// on) has 6.4 million shell quartets.
//
//   per shell pair   Hermite "densities": for every primitive pair and Cartesian component pair the
//                    coefficients E_t^x E_u^y E_v^z c_a c_b over t + u + v <= la + lb
//   per quartet      R_tuv(alpha, PQ) from the Boys function (tabulated Taylor expansion, downward
//                    recursion), contracted first with the ket's densities, then with the bra's
//   per quartet      Cartesian -> spherical with the matrices the caller passes (they carry the
//                    normalisation), scattered to the eight images in the dense (nao)^4 tensor
#include <sys/mman.h>

#include <atomic>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <thread>
#include <vector>

#include "../../include/nbx.h"

namespace {

constexpr int LMAX = 3;               // per shell (s, p, d, f)
constexpr int LTOT = 4 * LMAX;        // highest Hermite order of a quartet
constexpr int NCUBE = LTOT + 1;

inline int ncart(int l) { return (l + 1) * (l + 2) / 2; }
inline int nherm(int l) { return (l + 1) * (l + 2) * (l + 3) / 6; }

// Cartesian components in the order of integrals.py's _CART: xx xy xz yy yz zz for d, xxx xxy xxz xyy .. zzz for f
void cart_list(int l, int (*out)[3]) {
    int k = 0;
    for (int lx = l; lx >= 0; --lx)
        for (int ly = l - lx; ly >= 0; --ly) {
            out[k][0] = lx; out[k][1] = ly; out[k][2] = l - lx - ly;
            ++k;
        }
}

// ------------------------------------------------------------------------------------------ Boys function
struct BoysTable {
    static constexpr double STEP = 0.1, TMAX = 42.0;
    static constexpr int NT = 421, NORD = LTOT + 10;
    std::vector<double> f;  // [NT][NORD + 1]
    BoysTable() : f(size_t(NT) * (NORD + 1)) {
        for (int i = 0; i < NT; ++i) {
            const double t = i * STEP;
            // F_top by its convergent series e^-T sum_k (2T)^k / ((2n+1)(2n+3)...(2n+2k+1)), then downwards
            const int n = NORD;
            double term = 1.0 / (2 * n + 1), sum = term;
            for (int k = 1; k < 400; ++k) {
                term *= 2.0 * t / (2 * n + 2 * k + 1);
                sum += term;
                if (term < 1e-18 * sum) break;
            }
            const double et = std::exp(-t);
            double* row = &f[size_t(i) * (NORD + 1)];
            row[n] = et * sum;
            for (int m = n; m > 0; --m) row[m - 1] = (2.0 * t * row[m] + et) / (2 * m - 1);
        }
    }
    // F_0 .. F_l at T
    void eval(int l, double t, double* out) const {
        if (t >= TMAX - 1.0) {  // asymptotic F_0, upward recursion (stable for large T)
            const double et = std::exp(-t);
            out[0] = 0.5 * std::sqrt(M_PI / t);
            for (int m = 0; m < l; ++m) out[m + 1] = ((2 * m + 1) * out[m] - et) / (2.0 * t);
            return;
        }
        const int i = int(t / STEP + 0.5);
        const double d = i * STEP - t;  // |d| <= STEP / 2
        const double* row = &f[size_t(i) * (NORD + 1)];
        double acc = 0.0, pw = 1.0;
        for (int k = 0; k < 9; ++k) {  // F_l(T) = sum_k F_{l+k}(T0) (T0 - T)^k / k!
            acc += row[l + k] * pw;
            pw *= d / (k + 1);
        }
        out[l] = acc;
        const double et = std::exp(-t);
        for (int m = l; m > 0; --m) out[m - 1] = (2.0 * t * out[m] + et) / (2 * m - 1);
    }
};

// ------------------------------------------------------------------------------------------ shell pairs
struct Shell {
    int l, nprim, ncart_, nsph, ao0;
    const double *exps, *coefs, *sph;
    double c[3];
};

struct Pair {
    int ia, ib, lab, nab, nh, nprim;
    std::vector<double> p, px, py, pz;  // per surviving primitive pair
    std::vector<double> h;              // [prim][nab][nh]
    double schwarz = 0.0;
};

struct HermIndex {  // compact list of (t, u, v), t + u + v <= l
    int n;
    int tuv[165][3];
    explicit HermIndex(int l = 0) {
        n = 0;
        for (int t = 0; t <= l; ++t)
            for (int u = 0; u <= l - t; ++u)
                for (int v = 0; v <= l - t - u; ++v) {
                    tuv[n][0] = t; tuv[n][1] = u; tuv[n][2] = v;
                    ++n;
                }
    }
};

void hermite_e(int la, int lb, double a, double b, double xab, double e[LMAX + 1][LMAX + 1][2 * LMAX + 1]) {
    const double p = a + b, mu = a * b / p;
    const double xpa = -b / p * xab, xpb = a / p * xab, half = 0.5 / p;
    for (int i = 0; i <= LMAX; ++i)
        for (int j = 0; j <= LMAX; ++j)
            for (int t = 0; t <= 2 * LMAX; ++t) e[i][j][t] = 0.0;
    e[0][0][0] = std::exp(-mu * xab * xab);
    for (int i = 0; i < la; ++i)
        for (int t = 0; t <= i + 1; ++t) {
            double v = xpa * e[i][0][t];
            if (t > 0) v += half * e[i][0][t - 1];
            if (t + 1 <= i) v += (t + 1) * e[i][0][t + 1];
            e[i + 1][0][t] = v;
        }
    for (int i = 0; i <= la; ++i)
        for (int j = 0; j < lb; ++j)
            for (int t = 0; t <= i + j + 1; ++t) {
                double v = xpb * e[i][j][t];
                if (t > 0) v += half * e[i][j][t - 1];
                if (t + 1 <= i + j) v += (t + 1) * e[i][j][t + 1];
                e[i][j + 1][t] = v;
            }
}

void build_pair(const Shell& sa, const Shell& sb, int ia, int ib, double prim_cutoff, const HermIndex* hidx, Pair& pr) {
    pr.ia = ia; pr.ib = ib;
    pr.lab = sa.l + sb.l;
    pr.nab = sa.ncart_ * sb.ncart_;
    pr.nh = nherm(pr.lab);
    int ca[10][3], cb[10][3];
    cart_list(sa.l, ca);
    cart_list(sb.l, cb);
    const HermIndex& hi = hidx[pr.lab];
    double r2 = 0.0, ab[3];
    for (int d = 0; d < 3; ++d) { ab[d] = sa.c[d] - sb.c[d]; r2 += ab[d] * ab[d]; }
    for (int i = 0; i < sa.nprim; ++i)
        for (int j = 0; j < sb.nprim; ++j) {
            const double a = sa.exps[i], b = sb.exps[j], p = a + b;
            const double w = sa.coefs[i] * sb.coefs[j];
            if (std::fabs(w) * std::exp(-a * b / p * r2) < prim_cutoff) continue;
            double ex[LMAX + 1][LMAX + 1][2 * LMAX + 1], ey[LMAX + 1][LMAX + 1][2 * LMAX + 1], ez[LMAX + 1][LMAX + 1][2 * LMAX + 1];
            hermite_e(sa.l, sb.l, a, b, ab[0], ex);
            hermite_e(sa.l, sb.l, a, b, ab[1], ey);
            hermite_e(sa.l, sb.l, a, b, ab[2], ez);
            pr.p.push_back(p);
            pr.px.push_back((a * sa.c[0] + b * sb.c[0]) / p);
            pr.py.push_back((a * sa.c[1] + b * sb.c[1]) / p);
            pr.pz.push_back((a * sa.c[2] + b * sb.c[2]) / p);
            const size_t base = pr.h.size();
            pr.h.resize(base + size_t(pr.nab) * pr.nh);
            double* h = &pr.h[base];
            for (int x = 0; x < sa.ncart_; ++x)
                for (int y = 0; y < sb.ncart_; ++y) {
                    double* row = h + size_t(x * sb.ncart_ + y) * pr.nh;
                    for (int k = 0; k < hi.n; ++k) {
                        const int t = hi.tuv[k][0], u = hi.tuv[k][1], v = hi.tuv[k][2];
                        double val = 0.0;
                        if (t <= ca[x][0] + cb[y][0] && u <= ca[x][1] + cb[y][1] && v <= ca[x][2] + cb[y][2])
                            val = w * ex[ca[x][0]][cb[y][0]][t] * ey[ca[x][1]][cb[y][1]][u] * ez[ca[x][2]][cb[y][2]][v];
                        row[k] = val;
                    }
                }
        }
    pr.nprim = int(pr.p.size());
}

// R_tuv^0 for t + u + v <= l into a cube of edge NCUBE
void hermite_r(int l, double alpha, double x, double y, double z, const BoysTable& boys, double* r /* NCUBE^3 */, double* tmp) {
    double f[LTOT + 1];
    boys.eval(l, alpha * (x * x + y * y + z * z), f);
    double* cur = r;
    double* old = tmp;
    // the final result must land in r: choose the starting buffer by the parity of the number of passes
    if (l % 2 == 1) { cur = tmp; old = r; }
    double pw[LTOT + 1];
    pw[0] = 1.0;
    for (int n = 1; n <= l; ++n) pw[n] = pw[n - 1] * (-2.0 * alpha);
    for (int n = l; n >= 0; --n) {
        const int top = l - n;
        cur[0] = pw[n] * f[n];
        for (int t = 0; t <= top; ++t)
            for (int u = 0; u <= top - t; ++u)
                for (int v = 0; v <= top - t - u; ++v) {
                    if (t + u + v == 0) continue;
                    double val;
                    if (t > 0) {
                        val = x * old[((t - 1) * NCUBE + u) * NCUBE + v];
                        if (t > 1) val += (t - 1) * old[((t - 2) * NCUBE + u) * NCUBE + v];
                    } else if (u > 0) {
                        val = y * old[(t * NCUBE + u - 1) * NCUBE + v];
                        if (u > 1) val += (u - 1) * old[(t * NCUBE + u - 2) * NCUBE + v];
                    } else {
                        val = z * old[(t * NCUBE + u) * NCUBE + v - 1];
                        if (v > 1) val += (v - 1) * old[(t * NCUBE + u) * NCUBE + v - 2];
                    }
                    cur[(t * NCUBE + u) * NCUBE + v] = val;
                }
        double* s = cur; cur = old; old = s;
    }
}

struct Engine {
    std::vector<Shell> shells;
    std::vector<Pair> pairs;  // ia >= ib, index ia (ia + 1) / 2 + ib
    HermIndex hidx[2 * LMAX + 1];
    // ridx[lab][lcd][k_ab * n_cd + k_cd] = cube offset of (t + tau, u + nu, v + phi); sign of the ket index
    std::vector<int> ridx[2 * LMAX + 1][2 * LMAX + 1];
    double sgn[2 * LMAX + 1][165];
    BoysTable boys;
    int nao = 0;
    double cutoff = 0.0;

    // Cartesian block (nab x ncd) of the quartet (pairs ab, cd)
    void quartet_cart(const Pair& ab, const Pair& cd, double* blk, double* rbuf, double* rtmp, double* w) const {
        const int nab = ab.nab, ncd = cd.nab, nhab = ab.nh, nhcd = cd.nh, l = ab.lab + cd.lab;
        std::memset(blk, 0, sizeof(double) * nab * ncd);
        const int* ri = ridx[ab.lab][cd.lab].data();
        const double* sg = sgn[cd.lab];
        for (int i = 0; i < ab.nprim; ++i) {
            const double p = ab.p[i];
            const double* hab = &ab.h[size_t(i) * nab * nhab];
            for (int j = 0; j < cd.nprim; ++j) {
                const double q = cd.p[j];
                const double alpha = p * q / (p + q);
                hermite_r(l, alpha, ab.px[i] - cd.px[j], ab.py[i] - cd.py[j], ab.pz[i] - cd.pz[j], boys, rbuf, rtmp);
                const double pref = 2.0 * std::pow(M_PI, 2.5) / (p * q * std::sqrt(p + q));
                const double* hcd = &cd.h[size_t(j) * ncd * nhcd];
                // w[c][k_ab] = sum_k_cd (-1)^(tau+nu+phi) H_cd[c][k_cd] R[k_ab + k_cd]
                for (int c = 0; c < ncd; ++c) {
                    const double* hc = hcd + size_t(c) * nhcd;
                    double* wc = w + size_t(c) * nhab;
                    for (int ka = 0; ka < nhab; ++ka) {
                        const int* rk = ri + size_t(ka) * nhcd;
                        double acc = 0.0;
                        for (int kc = 0; kc < nhcd; ++kc) acc += sg[kc] * hc[kc] * rbuf[rk[kc]];
                        wc[ka] = acc * pref;
                    }
                }
                for (int a = 0; a < nab; ++a) {
                    const double* ha = hab + size_t(a) * nhab;
                    double* out = blk + size_t(a) * ncd;
                    for (int c = 0; c < ncd; ++c) {
                        const double* wc = w + size_t(c) * nhab;
                        double acc = 0.0;
                        for (int ka = 0; ka < nhab; ++ka) acc += ha[ka] * wc[ka];
                        out[c] += acc;
                    }
                }
            }
        }
    }

    // Cartesian (na nb nc nd) -> spherical, in place through a scratch buffer; returns the result pointer
    const double* to_spherical(const Pair& ab, const Pair& cd, double* blk, double* scr) const {
        const Shell* sh[4] = {&shells[ab.ia], &shells[ab.ib], &shells[cd.ia], &shells[cd.ib]};
        int dims[4] = {sh[0]->ncart_, sh[1]->ncart_, sh[2]->ncart_, sh[3]->ncart_};
        double* src = blk;
        double* dst = scr;
        for (int ax = 0; ax < 4; ++ax) {
            if (sh[ax]->l < 2) continue;  // identity for s and p (their normalisation sits in the coefficients)
            const int nc = dims[ax], ns = sh[ax]->nsph;
            int outer = 1, inner = 1;
            for (int k = 0; k < ax; ++k) outer *= dims[k];
            for (int k = ax + 1; k < 4; ++k) inner *= dims[k];
            for (int o = 0; o < outer; ++o)
                for (int m = 0; m < ns; ++m)
                    for (int in = 0; in < inner; ++in) {
                        double acc = 0.0;
                        for (int c = 0; c < nc; ++c) acc += sh[ax]->sph[m * nc + c] * src[(size_t(o) * nc + c) * inner + in];
                        dst[(size_t(o) * ns + m) * inner + in] = acc;
                    }
            dims[ax] = ns;
            double* s = src; src = dst; dst = s;
        }
        return src;
    }
};

}  // namespace

extern "C" int nbx_host_eri(int nshell, const int* ang, const int* nprim, const int* nfunc, const double* centres,
                            const double* exps, const double* coefs, const double* sph, double cutoff, int nthreads,
                            double* out) {
    if (nshell <= 0 || !ang || !nprim || !nfunc || !centres || !exps || !coefs || !sph || !out) return NBX_E_INVALID;
    Engine eng;
    eng.cutoff = cutoff;
    eng.shells.resize(nshell);
    int poff = 0, soff = 0, ao = 0;
    for (int s = 0; s < nshell; ++s) {
        if (ang[s] < 0 || ang[s] > LMAX || nprim[s] <= 0) return NBX_E_INVALID;
        Shell& sh = eng.shells[s];
        sh.l = ang[s];
        sh.nprim = nprim[s];
        sh.ncart_ = ncart(sh.l);
        sh.nsph = nfunc[s];  // 2 l + 1 spherical functions, or all Cartesian components (PySCF's mol.cart)
        if (sh.nsph != 2 * sh.l + 1 && sh.nsph != sh.ncart_) return NBX_E_INVALID;
        sh.exps = exps + poff;
        sh.coefs = coefs + poff;
        sh.sph = sph + soff;
        sh.ao0 = ao;
        for (int d = 0; d < 3; ++d) sh.c[d] = centres[3 * s + d];
        poff += sh.nprim;
        soff += sh.nsph * sh.ncart_;
        ao += sh.nsph;
    }
    eng.nao = ao;
    for (int l = 0; l <= 2 * LMAX; ++l) eng.hidx[l] = HermIndex(l);
    for (int lab = 0; lab <= 2 * LMAX; ++lab)
        for (int lcd = 0; lcd <= 2 * LMAX; ++lcd) {
            const HermIndex &ha = eng.hidx[lab], &hc = eng.hidx[lcd];
            auto& tab = eng.ridx[lab][lcd];
            tab.resize(size_t(ha.n) * hc.n);
            for (int ka = 0; ka < ha.n; ++ka)
                for (int kc = 0; kc < hc.n; ++kc)
                    tab[size_t(ka) * hc.n + kc] = ((ha.tuv[ka][0] + hc.tuv[kc][0]) * NCUBE + ha.tuv[ka][1] + hc.tuv[kc][1]) * NCUBE +
                                                  ha.tuv[ka][2] + hc.tuv[kc][2];
        }
    for (int l = 0; l <= 2 * LMAX; ++l)
        for (int k = 0; k < eng.hidx[l].n; ++k)
            eng.sgn[l][k] = ((eng.hidx[l].tuv[k][0] + eng.hidx[l].tuv[k][1] + eng.hidx[l].tuv[k][2]) & 1) ? -1.0 : 1.0;

    const int64_t npair = int64_t(nshell) * (nshell + 1) / 2;
    eng.pairs.resize(npair);
    if (nthreads <= 0) nthreads = int(std::thread::hardware_concurrency());
    if (nthreads <= 0) nthreads = 1;
    const double prim_cutoff = cutoff * 1e-4;

    auto run_pool = [&](auto&& body) {
        std::vector<std::thread> pool;
        for (int t = 1; t < nthreads; ++t) pool.emplace_back(body);
        body();
        for (auto& th : pool) th.join();
    };

    {   // pair data and Schwarz bounds sqrt(max (ab|ab))
        std::atomic<int64_t> next{0};
        run_pool([&] {
            std::vector<double> blk(100 * 100), w(100 * 84), rbuf(NCUBE * NCUBE * NCUBE), rtmp(NCUBE * NCUBE * NCUBE);
            for (;;) {
                const int64_t ij = next.fetch_add(1);
                if (ij >= npair) break;
                int ia = int((std::sqrt(8.0 * double(ij) + 1.0) - 1.0) / 2.0);
                while (int64_t(ia) * (ia + 1) / 2 > ij) --ia;
                while (int64_t(ia + 1) * (ia + 2) / 2 <= ij) ++ia;
                const int ib = int(ij - int64_t(ia) * (ia + 1) / 2);
                Pair& pr = eng.pairs[ij];
                build_pair(eng.shells[ia], eng.shells[ib], ia, ib, prim_cutoff, eng.hidx, pr);
                if (pr.nprim == 0) continue;
                eng.quartet_cart(pr, pr, blk.data(), rbuf.data(), rtmp.data(), w.data());
                double mx = 0.0;
                for (int a = 0; a < pr.nab; ++a) mx = std::fmax(mx, std::fabs(blk[size_t(a) * pr.nab + a]));
                pr.schwarz = std::sqrt(mx);
            }
        });
    }

    const int n = eng.nao;
    const size_t n2 = size_t(n) * n, n3 = n2 * n;
    {   // first touch of the output (3.8 GB at 148 functions) is page faults, not arithmetic: huge pages where the
        // kernel grants them, and every thread clears a contiguous slice
        const size_t bytes = sizeof(double) * n3 * n;
#ifdef MADV_HUGEPAGE
        const uintptr_t lo = (reinterpret_cast<uintptr_t>(out) + 4095) & ~uintptr_t(4095);
        const uintptr_t hi = (reinterpret_cast<uintptr_t>(out) + bytes) & ~uintptr_t(4095);
        if (hi > lo) (void)madvise(reinterpret_cast<void*>(lo), hi - lo, MADV_HUGEPAGE);
#endif
        std::atomic<int> slice{0};
        const int nsl = nthreads * 8;
        run_pool([&] {
            for (;;) {
                const int k = slice.fetch_add(1);
                if (k >= nsl) break;
                const size_t a = bytes / nsl * k, b = k + 1 == nsl ? bytes : bytes / nsl * (k + 1);
                std::memset(reinterpret_cast<char*>(out) + a, 0, b - a);
            }
        });
    }
    std::atomic<int64_t> next{0};
    run_pool([&] {
        std::vector<double> blk(10000), scr(10000), w(100 * 84), rbuf(NCUBE * NCUBE * NCUBE), rtmp(NCUBE * NCUBE * NCUBE);
        for (;;) {
            // heaviest bra pairs first would need a sort; descending index is a fair proxy (more kets)
            const int64_t ij = npair - 1 - next.fetch_add(1);
            if (ij < 0) break;
            const Pair& ab = eng.pairs[ij];
            if (ab.nprim == 0) continue;
            const Shell &sa = eng.shells[ab.ia], &sb = eng.shells[ab.ib];
            for (int64_t kl = 0; kl <= ij; ++kl) {
                const Pair& cd = eng.pairs[kl];
                if (cd.nprim == 0 || ab.schwarz * cd.schwarz < cutoff) continue;
                eng.quartet_cart(ab, cd, blk.data(), rbuf.data(), rtmp.data(), w.data());
                const double* v = eng.to_spherical(ab, cd, blk.data(), scr.data());
                const Shell &sc = eng.shells[cd.ia], &sd = eng.shells[cd.ib];
                const int na = sa.nsph, nb = sb.nsph, nc = sc.nsph, nd = sd.nsph;
                for (int a = 0; a < na; ++a)
                    for (int b = 0; b < nb; ++b)
                        for (int c = 0; c < nc; ++c)
                            for (int d = 0; d < nd; ++d) {
                                const double val = v[((size_t(a) * nb + b) * nc + c) * nd + d];
                                const size_t p = sa.ao0 + a, q = sb.ao0 + b, r = sc.ao0 + c, s = sd.ao0 + d;
                                out[p * n3 + q * n2 + r * n + s] = val;
                                out[q * n3 + p * n2 + r * n + s] = val;
                                out[p * n3 + q * n2 + s * n + r] = val;
                                out[q * n3 + p * n2 + s * n + r] = val;
                                out[r * n3 + s * n2 + p * n + q] = val;
                                out[s * n3 + r * n2 + p * n + q] = val;
                                out[r * n3 + s * n2 + q * n + p] = val;
                                out[s * n3 + r * n2 + q * n + p] = val;
                            }
            }
        }
    });
    return NBX_OK;
}

// ------------------------------------------------------------------------------------------------------
// One-electron matrices: overlap, kinetic energy, nuclear attraction (gto.Mole.intor("int1e_ovlp" / "int1e_kin"
// / "int1e_nuc"), reached through scf.UKS(mol).get_ovlp() / get_hcore(), nbed/driver.py:155-191).  Shell pairs
// over the thread pool; the same Hermite coefficients and R_tuv as above.
namespace {

constexpr int LB2 = LMAX + 2;  // the kinetic energy raises the ket by two

// E[i][j][t], i <= la, j <= lb, for one direction
void hermite_e_wide(int la, int lb, double a, double b, double xab, double e[LMAX + 1][LB2 + 1][LMAX + LB2 + 1]) {
    const double p = a + b, mu = a * b / p;
    const double xpa = -b / p * xab, xpb = a / p * xab, half = 0.5 / p;
    for (int i = 0; i <= LMAX; ++i)
        for (int j = 0; j <= LB2; ++j)
            for (int t = 0; t <= LMAX + LB2; ++t) e[i][j][t] = 0.0;
    e[0][0][0] = std::exp(-mu * xab * xab);
    for (int i = 0; i < la; ++i)
        for (int t = 0; t <= i + 1; ++t) {
            double v = xpa * e[i][0][t];
            if (t > 0) v += half * e[i][0][t - 1];
            if (t + 1 <= i) v += (t + 1) * e[i][0][t + 1];
            e[i + 1][0][t] = v;
        }
    for (int i = 0; i <= la; ++i)
        for (int j = 0; j < lb; ++j)
            for (int t = 0; t <= i + j + 1; ++t) {
                double v = xpb * e[i][j][t];
                if (t > 0) v += half * e[i][j][t - 1];
                if (t + 1 <= i + j) v += (t + 1) * e[i][j][t + 1];
                e[i][j + 1][t] = v;
            }
}

// (na x nb) Cartesian block -> (fa x fb) AO block
void block_to_ao(const Shell& sa, const Shell& sb, const double* blk, double* out) {
    double tmp[10 * 10];
    const int na = sa.ncart_, nb = sb.ncart_, fa = sa.nsph, fb = sb.nsph;
    for (int m = 0; m < fa; ++m)
        for (int y = 0; y < nb; ++y) {
            double acc = 0.0;
            if (sa.l < 2) acc = blk[m * nb + y];
            else
                for (int x = 0; x < na; ++x) acc += sa.sph[m * na + x] * blk[x * nb + y];
            tmp[m * nb + y] = acc;
        }
    for (int m = 0; m < fa; ++m)
        for (int n = 0; n < fb; ++n) {
            double acc = 0.0;
            if (sb.l < 2) acc = tmp[m * nb + n];
            else
                for (int y = 0; y < nb; ++y) acc += sb.sph[n * nb + y] * tmp[m * nb + y];
            out[m * fb + n] = acc;
        }
}

}  // namespace

extern "C" int nbx_host_1e(int nshell, const int* ang, const int* nprim, const int* nfunc, const double* centres,
                           const double* exps, const double* coefs, const double* sph, int natm, const double* charges,
                           const double* atom_xyz, int nthreads, double* s_out, double* t_out, double* v_out) {
    if (nshell <= 0 || !ang || !nprim || !nfunc || !centres || !exps || !coefs || !sph || natm < 0 || !s_out || !t_out ||
        !v_out || (natm > 0 && (!charges || !atom_xyz)))
        return NBX_E_INVALID;
    std::vector<Shell> shells(nshell);
    int poff = 0, soff = 0, ao = 0;
    for (int s = 0; s < nshell; ++s) {
        if (ang[s] < 0 || ang[s] > LMAX || nprim[s] <= 0) return NBX_E_INVALID;
        Shell& sh = shells[s];
        sh.l = ang[s];
        sh.nprim = nprim[s];
        sh.ncart_ = ncart(sh.l);
        sh.nsph = nfunc[s];
        if (sh.nsph != 2 * sh.l + 1 && sh.nsph != sh.ncart_) return NBX_E_INVALID;
        sh.exps = exps + poff;
        sh.coefs = coefs + poff;
        sh.sph = sph + soff;
        sh.ao0 = ao;
        for (int d = 0; d < 3; ++d) sh.c[d] = centres[3 * s + d];
        poff += sh.nprim;
        soff += sh.nsph * sh.ncart_;
        ao += sh.nsph;
    }
    const int n = ao;
    const BoysTable boys;
    const int64_t npair = int64_t(nshell) * (nshell + 1) / 2;
    if (nthreads <= 0) nthreads = int(std::thread::hardware_concurrency());
    if (nthreads <= 0) nthreads = 1;
    std::atomic<int64_t> next{0};
    auto body = [&] {
        std::vector<double> rbuf(NCUBE * NCUBE * NCUBE), rtmp(NCUBE * NCUBE * NCUBE);
        double bs[100], bt[100], bv[100], os[100], ot[100], ov[100];
        for (;;) {
            const int64_t ij = next.fetch_add(1);
            if (ij >= npair) break;
            int ia = int((std::sqrt(8.0 * double(ij) + 1.0) - 1.0) / 2.0);
            while (int64_t(ia) * (ia + 1) / 2 > ij) --ia;
            while (int64_t(ia + 1) * (ia + 2) / 2 <= ij) ++ia;
            const int ib = int(ij - int64_t(ia) * (ia + 1) / 2);
            const Shell &sa = shells[ia], &sb = shells[ib];
            int ca[10][3], cb[10][3];
            cart_list(sa.l, ca);
            cart_list(sb.l, cb);
            const int na = sa.ncart_, nb = sb.ncart_, lab = sa.l + sb.l;
            for (int k = 0; k < na * nb; ++k) bs[k] = bt[k] = bv[k] = 0.0;
            double ab[3];
            for (int d = 0; d < 3; ++d) ab[d] = sa.c[d] - sb.c[d];
            for (int i = 0; i < sa.nprim; ++i)
                for (int j = 0; j < sb.nprim; ++j) {
                    const double a = sa.exps[i], b = sb.exps[j], p = a + b, w = sa.coefs[i] * sb.coefs[j];
                    double e[3][LMAX + 1][LB2 + 1][LMAX + LB2 + 1];
                    for (int d = 0; d < 3; ++d) hermite_e_wide(sa.l, sb.l + 2, a, b, ab[d], e[d]);
                    const double pref = std::pow(M_PI / p, 1.5) * w;
                    const double pc[3] = {(a * sa.c[0] + b * sb.c[0]) / p, (a * sa.c[1] + b * sb.c[1]) / p,
                                          (a * sa.c[2] + b * sb.c[2]) / p};
                    // one-dimensional kinetic factors  t(i, j) = -2 b^2 s(i, j+2) + b (2j+1) s(i, j) - j(j-1)/2 s(i, j-2)
                    auto s1 = [&](int d, int ii, int jj) { return jj < 0 ? 0.0 : e[d][ii][jj][0]; };
                    auto t1 = [&](int d, int ii, int jj) {
                        return -2.0 * b * b * s1(d, ii, jj + 2) + b * (2 * jj + 1) * s1(d, ii, jj) -
                               0.5 * jj * (jj - 1) * s1(d, ii, jj - 2);
                    };
                    for (int x = 0; x < na; ++x)
                        for (int y = 0; y < nb; ++y) {
                            const double sx = s1(0, ca[x][0], cb[y][0]), sy = s1(1, ca[x][1], cb[y][1]),
                                         sz = s1(2, ca[x][2], cb[y][2]);
                            bs[x * nb + y] += pref * sx * sy * sz;
                            bt[x * nb + y] += pref * (t1(0, ca[x][0], cb[y][0]) * sy * sz + sx * t1(1, ca[x][1], cb[y][1]) * sz +
                                                      sx * sy * t1(2, ca[x][2], cb[y][2]));
                        }
                    const double vpref = 2.0 * M_PI / p * w;
                    for (int c = 0; c < natm; ++c) {
                        hermite_r(lab, p, pc[0] - atom_xyz[3 * c], pc[1] - atom_xyz[3 * c + 1], pc[2] - atom_xyz[3 * c + 2], boys,
                                  rbuf.data(), rtmp.data());
                        const double z = -charges[c] * vpref;
                        for (int x = 0; x < na; ++x)
                            for (int y = 0; y < nb; ++y) {
                                double acc = 0.0;
                                for (int t = 0; t <= ca[x][0] + cb[y][0]; ++t)
                                    for (int u = 0; u <= ca[x][1] + cb[y][1]; ++u)
                                        for (int v = 0; v <= ca[x][2] + cb[y][2]; ++v)
                                            acc += e[0][ca[x][0]][cb[y][0]][t] * e[1][ca[x][1]][cb[y][1]][u] *
                                                   e[2][ca[x][2]][cb[y][2]][v] * rbuf[(t * NCUBE + u) * NCUBE + v];
                                bv[x * nb + y] += z * acc;
                            }
                    }
                }
            block_to_ao(sa, sb, bs, os);
            block_to_ao(sa, sb, bt, ot);
            block_to_ao(sa, sb, bv, ov);
            for (int m = 0; m < sa.nsph; ++m)
                for (int k = 0; k < sb.nsph; ++k) {
                    const size_t r = size_t(sa.ao0 + m), c = size_t(sb.ao0 + k);
                    s_out[r * n + c] = s_out[c * n + r] = os[m * sb.nsph + k];
                    t_out[r * n + c] = t_out[c * n + r] = ot[m * sb.nsph + k];
                    v_out[r * n + c] = v_out[c * n + r] = ov[m * sb.nsph + k];
                }
        }
    };
    std::vector<std::thread> pool;
    for (int t = 1; t < nthreads; ++t) pool.emplace_back(body);
    body();
    for (auto& th : pool) th.join();
    return NBX_OK;
}
