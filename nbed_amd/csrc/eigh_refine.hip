// libnbx: warm-started symmetric eigensolve by iterative refinement on the matrix cores.
//
// Inside an SCF the matrix to diagonalise differs little from the previous cycle's, whose
// eigenvectors X are at hand (nbx_eigh_warm).  Instead of rotating the perturbation away one
// Jacobi sweep at a time (one workgroup per matrix, ~0.35 ms at N = 148), the pair (X, lambda)
// is refined with the Ogita-Aishima iteration (Japan J. Indust. Appl. Math. 35 (2018) 1007),
// which is nothing but GEMMs:
//
//     R = I - X^T X          S = X^T (A X)          lambda_i = S_ii / (1 - R_ii)
//     E_ij = (S_ij + lambda_j R_ij) / (lambda_j - lambda_i)      |lambda_i - lambda_j| > omega
//          = R_ij / 2                                             otherwise (and i == j)
//     X <- X (I + E)         omega = 2 (||S - diag||_F + ||A||_F ||R||_F)
//
// Two departures from the paper's formulas, both for matrices whose spectrum spans many orders of magnitude (the
// mu-shifted Fock matrices of nbed/driver.py:518: n_env eigenvalues at mu = 1e6 above a valence spectrum of O(10)):
//  * S is symmetrised, S_ij <- (S_ij + S_ji) / 2, before E is formed.  The two are computed by different dot products
//    and differ by rounding noise ~ eps ||A||; E_ij + E_ji = R_ij then holds only to noise / gap, and the updated
//    vectors lose orthonormality at that level (1e-9 where ||A|| = 1e6) instead of O(E^2) -- which a solver that
//    takes them as an orthonormal start (the Jacobi fallback of the next cycle) turns into residuals of 1e-3.
//  * omega is taken per pair: omega_ij = 2 (||N||_F + max(|lambda_i|, |lambda_j|) ||R||_F), N_ij = the smaller in
//    magnitude of the two one-sided residuals S_ij + lambda_j R_ij and S_ij + lambda_i R_ij (= S_ij when R = 0).
//    With the global norms a loss of orthonormality of 1e-8 among the levels at 1e6 (harmless: their own gaps are
//    O(1)) declared every valence pair closer than 0.05 a cluster, never rotated and never accepted.  Whatever the
//    choice of omega, a matrix is ACCEPTED only on what the last iteration measured (max|E|, cluster couplings).
//
// It converges quadratically once max|E| is small.  Everything is decided on the device: the
// E kernel of iteration k sets status[b] = k+1 when max|E| < tol (that update is still applied,
// then the matrix is finished), -1 when the iteration is not contracting; the GEMMs of later
// iterations and the Jacobi fallback that follows are queued unconditionally and gated on that
// word, so the host never waits.  A matrix is accepted only if no pair of eigenvalues closer than
// omega is coupled by more than rounding noise in S -- otherwise vectors inside a near-degenerate
// cluster would be left unrotated -- such cases go to Jacobi, which resolves them.
#include "nbx_common.h"

namespace {

constexpr int RF_THREADS = 256;
constexpr int RF_WGS = 16;
constexpr int RF_MAX_ITER = 6;
constexpr int RF_UNROLL = 4;
constexpr double RF_TOL = 3.0e-8;       // accepted update has max|E| below this: error ~ tol^2
constexpr double RF_GIVE_UP = 0.25;     // not in the contracting regime
constexpr double RF_CLUSTER_NOISE = 1.0e-14;
// A matrix is also finished when every off-diagonal residual S_ij + lambda_j R_ij the update divides is at the level
// of the rounding errors made in forming S = X^T (A X): |.| <= RF_NOISE sqrt(N) ||A||_F.  Nothing further can be
// gained then -- the next S would carry the same noise -- and max|E| = noise / gap need not be below RF_TOL: the
// mu-shifted Fock matrices of nbed/driver.py:518 (mu = 1e6: ||A||_F ~ 5e6, noise ~ 1e-9, level gaps ~ 1e-2) stall at
// max|E| ~ 1e-7..1e-6, the accuracy any backward-stable solver (LAPACK's dsygvd in the reference) reaches on them.
// For matrices of ordinary norm the test is stricter than RF_TOL and changes nothing.
constexpr double RF_NOISE = 8.0 * 1.1102230246251565e-16;
constexpr int RF_PART = 6;  // doubles each workgroup publishes per matrix

// G = X^T X, S = X^T A X  ->  Ep = I + E, lambda, status.  RF_WGS workgroups per matrix (grid
// (RF_WGS, batch)); the two global quantities the element-wise work needs -- the norms behind
// omega before it, max|E| after it -- go through agent-scope atomics: partial results are
// published, a per-matrix counter is bumped, and (1) every workgroup waits until all RF_WGS have
// arrived, then adds the partials up in fixed order (identical omega everywhere); (2) the last
// workgroup to finish phase 2 reduces the maxima, writes the status word and clears the counter.
// All RF_WGS * batch (<= 16 * 512 small) workgroups are co-resident, so the wait cannot deadlock.
__device__ __forceinline__ void publish(double* p, double v) {
    __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ double peek(const double* p) {
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

__global__ __launch_bounds__(RF_THREADS) void refine_e_kernel(int N, const double* __restrict__ A,
                                                              const double* __restrict__ S,
                                                              const double* __restrict__ G, double* __restrict__ Ep,
                                                              double* __restrict__ lam_out, int* __restrict__ status,
                                                              double* __restrict__ norm_a, int iter, int max_iter,
                                                              double* __restrict__ partial, int* __restrict__ counters,
                                                              int* __restrict__ final_status = nullptr,
                                                              const double* __restrict__ norm_part = nullptr,
                                                              int norm_tiles = 0) {
    extern __shared__ double sm[];
    double* lam = sm;
    double* red = sm + N;  // 3 * RF_THREADS/64 + 8 doubles
    const int b = blockIdx.y, wg = blockIdx.x, nwg = gridDim.x;
    if (iter > 0 && status[b] != 0) return;
    const int64_t n2 = (int64_t)N * N;
    if (A != nullptr) A += b * n2;  // nullptr: ||S||_F stands in for ||A||_F (S is A in a nearly orthonormal basis)
    S += b * n2;
    G += b * n2;
    Ep += b * n2;
    double* part = partial + (int64_t)b * nwg * RF_PART;
    int* counter = counters + b;
    for (int i = threadIdx.x; i < N; i += RF_THREADS) lam[i] = S[(int64_t)i * N + i] / G[(int64_t)i * N + i];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    constexpr int NW = RF_THREADS / 64;
    const int total = N * N;
    const int first = wg * RF_THREADS + threadIdx.x, stride = nwg * RF_THREADS;
    const float inv_n = 1.0f / (float)N;
    auto row_of = [&](int e) {  // e / N for e < 2^24 without an integer division
        int i = (int)((float)e * inv_n);
        i -= (i * N > e) ? 1 : 0;
        i += ((i + 1) * N <= e) ? 1 : 0;
        return i;
    };
    // ---- phase 1: ||S - diag||_F^2, ||R||_F^2 (and ||A||_F^2 once)
    double off2 = 0.0, r2 = 0.0, a2 = 0.0;
    // `norm_part`: the GEMM that produced S and G left their squared norms tile by tile
    // (gemm_small_kernel): every workgroup adds them up in the same fixed order -- no pass over the
    // matrices and no waiting for the sibling workgroups
    if (norm_part != nullptr) {
        const double* ps = norm_part + (int64_t)b * 2 * norm_tiles * 2;
        const double* pg = ps + (int64_t)norm_tiles * 2;
        for (int w = threadIdx.x; w < norm_tiles; w += RF_THREADS) {
            off2 += ps[2 * w];
            a2 += ps[2 * w + 1];
            r2 += pg[2 * w];
        }
    }
    __syncthreads();  // lam[] is complete
    for (int e0 = first; e0 < total && norm_part == nullptr; e0 += stride * RF_UNROLL) {
        double sv[RF_UNROLL], gv[RF_UNROLL], av[RF_UNROLL];
#pragma unroll
        for (int u = 0; u < RF_UNROLL; ++u) {
            const int e = e0 + u * stride;
            const bool in = e < total;
            sv[u] = in ? S[e] : 0.0;
            gv[u] = in ? G[e] : 0.0;
            av[u] = (in && iter == 0) ? (A != nullptr ? A[e] : sv[u]) : 0.0;
        }
#pragma unroll
        for (int u = 0; u < RF_UNROLL; ++u) {
            const int e = e0 + u * stride;
            if (e < total) {
                const int i = row_of(e), j = e - i * N;
                const double r = (i == j) ? 1.0 - gv[u] : -gv[u];
                if (i != j) {
                    // the smaller of the two one-sided residuals of the pair (R_ij = -G_ij)
                    const double n1 = fabs(sv[u] - lam[j] * gv[u]), n2 = fabs(sv[u] - lam[i] * gv[u]);
                    const double nm = fmin(n1, n2);
                    off2 = fma(nm, nm, off2);
                }
                r2 = fma(r, r, r2);
                a2 = fma(av[u], av[u], a2);
            }
        }
    }
    off2 = nbx_wave_sum_dpp(off2);  // (every lane is active here; lane moves instead of six LDS round trips each)
    r2 = nbx_wave_sum_dpp(r2);
    a2 = nbx_wave_sum_dpp(a2);
    __syncthreads();
    if (lane == 0) {
        red[wave] = off2;
        red[NW + wave] = r2;
        red[2 * NW + wave] = a2;
    }
    __syncthreads();
    if (threadIdx.x == 0 && norm_part != nullptr) {
        double t0 = 0.0, t1 = 0.0, t2 = 0.0;
        for (int w = 0; w < NW; ++w) {
            t0 += red[w];
            t1 += red[NW + w];
            t2 += red[2 * NW + w];
        }
        red[3 * NW + 3] = 0.0;
        red[3 * NW + 0] = t0;
        red[3 * NW + 1] = t1;
        red[3 * NW + 2] = t2;
    } else if (threadIdx.x == 0) {
        double t0 = 0.0, t1 = 0.0, t2 = 0.0;
        for (int w = 0; w < NW; ++w) {
            t0 += red[w];
            t1 += red[NW + w];
            t2 += red[2 * NW + w];
        }
        publish(part + wg * RF_PART + 0, t0);
        publish(part + wg * RF_PART + 1, t1);
        publish(part + wg * RF_PART + 2, t2);
        __hip_atomic_fetch_add(counter, 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
        // bounded wait (~50 ms): if a sibling never arrives the matrix is handed to Jacobi
        int spins = 0;
        while (__hip_atomic_load(counter, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < nwg && ++spins < (1 << 20))
            __builtin_amdgcn_s_sleep(2);
        red[3 * NW + 3] = (spins >= (1 << 20)) ? 1.0 : 0.0;
        t0 = t1 = t2 = 0.0;
        for (int w = 0; w < nwg; ++w) {
            t0 += peek(part + w * RF_PART + 0);
            t1 += peek(part + w * RF_PART + 1);
            t2 += peek(part + w * RF_PART + 2);
        }
        red[3 * NW + 0] = t0;
        red[3 * NW + 1] = t1;
        red[3 * NW + 2] = t2;
    }
    // `final_status`: this launch is the whole solve (one iteration queued, the caller takes the
    // updated vectors in place and lam as the eigenvalues): accepted only if lam is already
    // ascending -- refinement keeps column i with eigenvalue i, so the order changes only when
    // two levels cross -- and the verdict goes to the caller's status word (1000 + 1 / <= 0)
    int unsorted = 0;
    if (final_status != nullptr)
        for (int i = threadIdx.x + 1; i < N; i += RF_THREADS) unsorted |= lam[i - 1] > lam[i] ? 1 : 0;
    unsorted = __syncthreads_or(unsorted);
    off2 = red[3 * NW + 0];
    r2 = red[3 * NW + 1];
    a2 = red[3 * NW + 2];
    double na;
    if (iter == 0) {
        na = sqrt(a2);
        if (wg == 0 && threadIdx.x == 0) norm_a[b] = na;
    } else {
        na = norm_a[b];
    }
    const double om_s = 2.0 * sqrt(off2), om_r = 2.0 * sqrt(r2);  // omega_ij = om_s + max(|lam_i|, |lam_j|) om_r
    // ---- phase 2: E
    double emax = 0.0, cmax = 0.0, nmax = 0.0;
    for (int e0 = first; e0 < total; e0 += stride * RF_UNROLL) {
        double sv[RF_UNROLL], gv[RF_UNROLL], st[RF_UNROLL];
        int ri[RF_UNROLL];
#pragma unroll
        for (int u = 0; u < RF_UNROLL; ++u) {
            const int e = e0 + u * stride;
            const bool in = e < total;
            ri[u] = in ? row_of(e) : 0;
            sv[u] = in ? S[e] : 0.0;
            gv[u] = in ? G[e] : 0.0;
            st[u] = in ? S[(int64_t)(e - ri[u] * N) * N + ri[u]] : 0.0;  // S_ji (the matrix lives in L2)
        }
#pragma unroll
        for (int u = 0; u < RF_UNROLL; ++u) {
            const int e = e0 + u * stride;
            if (e < total) {
                const int i = ri[u], j = e - i * N;
                const double sx = 0.5 * (sv[u] + st[u]), g = gv[u];
                double ev;
                if (i == j) {
                    ev = 0.5 * (1.0 - g);
                    emax = fmax(emax, fabs(ev));
                    Ep[e] = 1.0 + ev;
                } else {
                    const double li = lam[i], lj = lam[j];
                    const double d = lj - li;
                    if (fabs(d) > om_s + fmax(fabs(li), fabs(lj)) * om_r) {
                        const double num = sx - lj * g;  // R_ij = -G_ij
                        ev = num / d;
                        nmax = fmax(nmax, fabs(num));
                    } else {
                        ev = -0.5 * g;
                        cmax = fmax(cmax, fabs(sx));
                    }
                    emax = fmax(emax, fabs(ev));
                    Ep[e] = ev;
                }
            }
        }
    }
    emax = -nbx_wave_min_dpp(-emax);
    cmax = -nbx_wave_min_dpp(-cmax);
    nmax = -nbx_wave_min_dpp(-nmax);
    __syncthreads();
    if (lane == 0) {
        red[wave] = emax;
        red[NW + wave] = cmax;
        red[2 * NW + wave] = nmax;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int w = 0; w < NW; ++w) {
            emax = fmax(emax, red[w]);
            cmax = fmax(cmax, red[NW + w]);
            nmax = fmax(nmax, red[2 * NW + w]);
        }
        if (red[3 * NW + 3] != 0.0) emax = 1.0e300;  // timed out above: force the fallback
        publish(part + wg * RF_PART + 3, emax);
        publish(part + wg * RF_PART + 4, cmax);
        publish(part + wg * RF_PART + 5, nmax);
        const int arrived = __hip_atomic_fetch_add(counter, 1, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
        if (arrived == (norm_part != nullptr ? 1 : 2) * nwg - 1) {  // last workgroup of this matrix
            for (int w = 0; w < nwg; ++w) {
                emax = fmax(emax, peek(part + w * RF_PART + 3));
                cmax = fmax(cmax, peek(part + w * RF_PART + 4));
                nmax = fmax(nmax, peek(part + w * RF_PART + 5));
            }
            const bool at_noise = nmax <= RF_NOISE * sqrt((double)N) * na && emax < 1.0e-3;
            int st = 0;
            if (!(emax < RF_GIVE_UP)) st = -1;  // also NaN
            else if ((emax < RF_TOL || at_noise) && cmax <= RF_CLUSTER_NOISE * na) st = iter + 1;
            else if (iter == max_iter - 1) st = -1;
            if (final_status != nullptr) {
                if (st > 0 && unsorted) st = -3;
                final_status[b] = st > 0 ? 1000 + st : (st < 0 ? st : -2);
            }
            status[b] = st;
            __hip_atomic_store(counter, 0, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
    if (wg == 0)
        for (int i = threadIdx.x; i < N; i += RF_THREADS) lam_out[(int64_t)b * N + i] = lam[i];
}

// Accepted matrices: eigenvalues ascending, eigenvector columns permuted to match.
// grid (row blocks, batch); every workgroup recomputes the (cheap) ranks.
constexpr int RF_ROWS = 8;
__global__ __launch_bounds__(256) void refine_finish_kernel(int N, const double* __restrict__ xb0,
                                                            const double* __restrict__ xb1,
                                                            const double* __restrict__ lam_in,
                                                            const int* __restrict__ status, double* __restrict__ w,
                                                            double* __restrict__ v, int* __restrict__ jacobi_status) {
    extern __shared__ double sm[];
    double* lam = sm;
    int* rank = reinterpret_cast<int*>(sm + N);
    const int b = blockIdx.y;
    const int st = status[b];
    if (st <= 0) {  // not accepted: say so (a fallback solver that follows overwrites the word)
        if (blockIdx.x == 0 && threadIdx.x == 0 && jacobi_status != nullptr) jacobi_status[b] = st < 0 ? st : -2;
        return;
    }
    const int64_t n2 = (int64_t)N * N;
    const double* X = ((st & 1) ? xb1 : xb0) + b * n2;
    for (int i = threadIdx.x; i < N; i += blockDim.x) lam[i] = lam_in[(int64_t)b * N + i];
    __syncthreads();
    // Refinement keeps column i with eigenvalue i, so between SCF cycles the order hardly ever
    // changes: if the values are already ascending the ranks are the identity (N comparisons
    // instead of N^2).
    int unsorted = 0;
    for (int i = threadIdx.x + 1; i < N; i += blockDim.x) unsorted |= lam[i - 1] > lam[i] ? 1 : 0;
    unsorted = __syncthreads_or(unsorted);
    for (int i = threadIdx.x; i < N; i += blockDim.x) {
        const double li = lam[i];
        int r = i;
        if (unsorted) {
            r = 0;
            for (int j = 0; j < N; ++j) {
                const double lj = lam[j];
                r += (lj < li || (lj == li && j < i)) ? 1 : 0;
            }
        }
        rank[i] = r;
        if (blockIdx.x == 0) w[(int64_t)b * N + r] = li;
    }
    if (blockIdx.x == 0 && threadIdx.x == 0 && jacobi_status != nullptr) jacobi_status[b] = 1000 + st;
    __syncthreads();
    const int r0 = blockIdx.x * RF_ROWS;
    for (int rr = 0; rr < RF_ROWS; ++rr) {
        const int r = r0 + rr;
        if (r >= N) break;
        for (int i = threadIdx.x; i < N; i += blockDim.x)
            v[b * n2 + (int64_t)r * N + rank[i]] = X[(int64_t)r * N + i];
    }
}

size_t align256(size_t x) { return (x + 255) & ~(size_t)255; }

struct RefineLayout {
    size_t xb0, xb1, y, s, g, ep, lam, norm, partial, status, npart, total;
};

RefineLayout rlayout(int64_t n, int64_t batch) {
    RefineLayout L;
    const size_t mat = align256((size_t)(n * n * batch) * sizeof(double));
    size_t off = 0;
    L.xb0 = off; off += mat;
    L.xb1 = off; off += mat;
    L.y = off; off += mat;
    L.s = off; off += mat;
    L.g = off; off += mat;
    L.ep = off; off += mat;
    L.lam = off; off += align256((size_t)(n * batch) * sizeof(double));
    L.norm = off; off += align256((size_t)batch * sizeof(double));
    L.partial = off; off += align256((size_t)(batch * RF_WGS * RF_PART) * sizeof(double));
    L.status = off; off += align256((size_t)batch * sizeof(int));
    L.npart = off; off += align256((size_t)nbx_gemm_small_norm_doubles(n, n, batch) * sizeof(double));
    L.total = off;
    return L;
}

}  // namespace

bool nbx_eigh_refine_supported(int64_t n, int64_t batch) {
    return n >= 2 && n <= 4096 && batch <= NBX_COUNTERS;
}

size_t nbx_eigh_refine_worksize(int64_t n, int64_t batch) { return rlayout(n, batch).total; }

int nbx_eigh_refine(nbx_ctx* ctx, int64_t n, int64_t batch, const double* d_a, const double* d_v0, double* d_w,
                    double* d_v, void* d_work, int* d_jacobi_status, const int** d_status_out, int max_iter) {
    NBX_CHECK_ARG(max_iter >= 1 && max_iter <= RF_MAX_ITER);
    const RefineLayout L = rlayout(n, batch);
    char* base = static_cast<char*>(d_work);
    double* xb[2] = {reinterpret_cast<double*>(base + L.xb0), reinterpret_cast<double*>(base + L.xb1)};
    double* y = reinterpret_cast<double*>(base + L.y);
    double* s = reinterpret_cast<double*>(base + L.s);
    double* g = reinterpret_cast<double*>(base + L.g);
    double* ep = reinterpret_cast<double*>(base + L.ep);
    double* lam = reinterpret_cast<double*>(base + L.lam);
    double* norm = reinterpret_cast<double*>(base + L.norm);
    double* partial = reinterpret_cast<double*>(base + L.partial);
    int* status = reinterpret_cast<int*>(base + L.status);
    double* npart = reinterpret_cast<double*>(base + L.npart);
    const int norm_tiles = (int)(nbx_cdiv(n, 16) * nbx_cdiv(n, 16));
    const int64_t n2 = n * n;
    const bool small = nbx_gemm_small_supported(n, n, n, batch);
    for (int it = 0; it < max_iter; ++it) {
        // iteration `it` reads src and writes dst; the first one reads V0 in place and its
        // GEMMs are not gated (its E kernel initialises the status words)
        const double* src = (it == 0) ? d_v0 : xb[it & 1];
        double* dst = xb[(it + 1) & 1];
        const int* gate = (it == 0) ? nullptr : status;
        int rc = nbx_gemm_gated(ctx, 'N', 'N', n, n, n, 1.0, d_a, n, n2, src, n, n2, 0.0, y, n, n2, batch, gate, 0, 0);
        if (rc != NBX_OK) return rc;
        if (small) {
            // S = X^T Y and G = X^T X in one launch (they share op(A) = X^T), which also leaves the
            // squared norms the E kernel needs (||S||_F then stands in for ||A||_F: X is orthonormal
            // to rounding, it is the previous solve's result)
            rc = nbx_gemm_small_gated(ctx, 'T', 'N', n, n, n, 1.0, src, n, n2, y, n, n2, 0.0, s, n, n2, batch, gate, 0, 0,
                                      src, g, npart);
            if (rc != NBX_OK) return rc;
        } else {
            rc = nbx_gemm_gated(ctx, 'T', 'N', n, n, n, 1.0, src, n, n2, y, n, n2, 0.0, s, n, n2, batch, gate, 0, 0);
            if (rc != NBX_OK) return rc;
            rc = nbx_gemm_gated(ctx, 'T', 'N', n, n, n, 1.0, src, n, n2, src, n, n2, 0.0, g, n, n2, batch, gate, 0, 0);
            if (rc != NBX_OK) return rc;
        }
        hipLaunchKernelGGL(refine_e_kernel, dim3(RF_WGS, (unsigned)batch), dim3(RF_THREADS),
                           (size_t)(n + 3 * (RF_THREADS / 64) + 8) * sizeof(double), ctx->stream, (int)n, d_a, s, g, ep,
                           lam, status, norm, it, max_iter, partial, ctx->d_counters, static_cast<int*>(nullptr),
                           small ? npart : static_cast<const double*>(nullptr), norm_tiles);
        NBX_LAUNCH_CHECK();
        rc = nbx_gemm_gated(ctx, 'N', 'N', n, n, n, 1.0, src, n, n2, ep, n, n2, 0.0, dst, n, n2, batch, status, 0, it + 1);
        if (rc != NBX_OK) return rc;
    }
    hipLaunchKernelGGL(refine_finish_kernel, dim3((unsigned)nbx_cdiv(n, RF_ROWS), (unsigned)batch), dim3(256),
                       (size_t)n * sizeof(double) + (size_t)n * sizeof(int), ctx->stream, (int)n, xb[0], xb[1], lam,
                       status, d_w, d_v, d_jacobi_status);
    NBX_LAUNCH_CHECK();
    *d_status_out = status;
    return NBX_OK;
}

// ---------------------------------------------------------------------------------------------
// The same iteration on the generalised problem F C = S C eps, started from the previous SCF
// cycle's MO coefficients C0 (S-orthonormal eigenvectors of a nearby F):
//     G = C^T S C  (R = I - G)      S~ = C^T F C      lambda_i = S~_ii / G_ii      C <- C (I + E)
// No Loewdin transform of F, no back-transform of the vectors and NO fallback solver behind it:
// four launches per iteration plus the sort -- {Yt = C^T F, Zt = C^T S} and {S~ = C^T Yt^T,
// G = C^T Zt^T} are pairs sharing op(A) = C^T.  d_status[b] = 1000 + iterations when matrix b was
// accepted (d_w, d_c written), <= 0 when not (outputs untouched): the caller must check it.
namespace {
struct GeigLayout {
    size_t cb0, cb1, yt, zt, s, g, ep, lam, norm, partial, status, npart, total;
};
GeigLayout glayout(int64_t n, int64_t batch) {
    GeigLayout L;
    const size_t mat = align256((size_t)(n * n * batch) * sizeof(double));
    size_t off = 0;
    L.cb0 = off; off += mat;
    L.cb1 = off; off += mat;
    L.yt = off; off += mat;
    L.zt = off; off += mat;
    L.s = off; off += mat;
    L.g = off; off += mat;
    L.ep = off; off += mat;
    L.lam = off; off += align256((size_t)(n * batch) * sizeof(double));
    L.norm = off; off += align256((size_t)batch * sizeof(double));
    L.partial = off; off += align256((size_t)(batch * RF_WGS * RF_PART) * sizeof(double));
    L.status = off; off += align256((size_t)batch * sizeof(int));
    L.npart = off; off += align256((size_t)nbx_gemm_small_norm_doubles(n, n, batch) * sizeof(double));
    L.total = off;
    return L;
}
}  // namespace

extern "C" size_t nbx_geig_refine_worksize(int64_t n, int64_t batch) {
    if (n <= 0 || batch <= 0) return 0;
    return glayout(n, batch).total;
}

extern "C" int nbx_geig_refine(nbx_ctx* ctx, int64_t n, int64_t batch, const double* d_f, const double* d_s,
                               const double* d_c0, double* d_w, double* d_c, int* d_status, void* d_work,
                               size_t work_bytes, int max_iter) {
    NBX_CHECK_ARG(ctx && d_f && d_s && d_c0 && d_w && d_c && d_status);
    NBX_CHECK_ARG(n >= 2 && n <= 4096 && batch >= 1 && batch <= NBX_COUNTERS);
    NBX_CHECK_ARG(max_iter >= 1 && max_iter <= RF_MAX_ITER);
    const GeigLayout L = glayout(n, batch);
    if (d_work == nullptr || work_bytes < L.total) {
        nbx_set_error("nbx_geig_refine: workspace %zu < %zu bytes", work_bytes, L.total);
        return NBX_E_NOMEM;
    }
    char* base = static_cast<char*>(d_work);
    double* cb[2] = {reinterpret_cast<double*>(base + L.cb0), reinterpret_cast<double*>(base + L.cb1)};
    double* yt = reinterpret_cast<double*>(base + L.yt);
    double* zt = reinterpret_cast<double*>(base + L.zt);
    double* s = reinterpret_cast<double*>(base + L.s);
    double* g = reinterpret_cast<double*>(base + L.g);
    double* ep = reinterpret_cast<double*>(base + L.ep);
    double* lam = reinterpret_cast<double*>(base + L.lam);
    double* norm = reinterpret_cast<double*>(base + L.norm);
    double* partial = reinterpret_cast<double*>(base + L.partial);
    int* status = reinterpret_cast<int*>(base + L.status);
    double* npart = reinterpret_cast<double*>(base + L.npart);
    const int norm_tiles = (int)(nbx_cdiv(n, 16) * nbx_cdiv(n, 16));
    const int64_t n2 = n * n;
    const bool pair = nbx_gemm_small_supported(n, n, n, 2 * batch);
    // One iteration queued (the steady state of an SCF): the update GEMM writes the caller's d_c,
    // the E kernel writes d_w and the verdict, and no sorting pass follows (4 launches).
    const bool direct = max_iter == 1 && d_c != d_c0;
    for (int it = 0; it < max_iter; ++it) {
        const double* src = (it == 0) ? d_c0 : cb[it & 1];
        double* dst = direct ? d_c : cb[(it + 1) & 1];
        const int* gate = (it == 0) ? nullptr : status;
        int rc;
        if (pair) {
            rc = nbx_gemm_small_gated(ctx, 'T', 'N', n, n, n, 1.0, src, n, n2, d_f, n, n2, 0.0, yt, n, n2, batch, gate, 0,
                                      0, d_s, zt);
            if (rc != NBX_OK) return rc;
            // ... and leaves the squared norms of S~ and I - G for the E kernel, tile by tile
            rc = nbx_gemm_small_gated(ctx, 'T', 'T', n, n, n, 1.0, src, n, n2, yt, n, n2, 0.0, s, n, n2, batch, gate, 0, 0,
                                      zt, g, npart);
            if (rc != NBX_OK) return rc;
        } else {
            rc = nbx_gemm_gated(ctx, 'T', 'N', n, n, n, 1.0, src, n, n2, d_f, n, n2, 0.0, yt, n, n2, batch, gate, 0, 0);
            if (rc != NBX_OK) return rc;
            rc = nbx_gemm_gated(ctx, 'T', 'N', n, n, n, 1.0, src, n, n2, d_s, n, n2, 0.0, zt, n, n2, batch, gate, 0, 0);
            if (rc != NBX_OK) return rc;
            rc = nbx_gemm_gated(ctx, 'T', 'T', n, n, n, 1.0, src, n, n2, yt, n, n2, 0.0, s, n, n2, batch, gate, 0, 0);
            if (rc != NBX_OK) return rc;
            rc = nbx_gemm_gated(ctx, 'T', 'T', n, n, n, 1.0, src, n, n2, zt, n, n2, 0.0, g, n, n2, batch, gate, 0, 0);
            if (rc != NBX_OK) return rc;
        }
        hipLaunchKernelGGL(refine_e_kernel, dim3(RF_WGS, (unsigned)batch), dim3(RF_THREADS),
                           (size_t)(n + 3 * (RF_THREADS / 64) + 8) * sizeof(double), ctx->stream, (int)n,
                           static_cast<const double*>(nullptr), s, g, ep, direct ? d_w : lam, status, norm, it, max_iter,
                           partial, ctx->d_counters, direct ? d_status : static_cast<int*>(nullptr),
                           pair ? npart : static_cast<const double*>(nullptr), norm_tiles);
        NBX_LAUNCH_CHECK();
        rc = nbx_gemm_gated(ctx, 'N', 'N', n, n, n, 1.0, src, n, n2, ep, n, n2, 0.0, dst, n, n2, batch, status, 0, it + 1);
        if (rc != NBX_OK) return rc;
    }
    if (direct) return NBX_OK;
    hipLaunchKernelGGL(refine_finish_kernel, dim3((unsigned)nbx_cdiv(n, RF_ROWS), (unsigned)batch), dim3(256),
                       (size_t)n * sizeof(double) + (size_t)n * sizeof(int), ctx->stream, (int)n, cb[0], cb[1], lam,
                       status, d_w, d_c, d_status);
    NBX_LAUNCH_CHECK();
    return NBX_OK;
}
