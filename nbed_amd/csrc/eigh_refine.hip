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
// It converges quadratically once max|E| is small.  Everything is decided on the device: the
// E kernel of iteration k sets status[b] = k+1 when max|E| < tol (that update is still applied,
// then the matrix is finished), -1 when the iteration is not contracting; the GEMMs of later
// iterations and the Jacobi fallback that follows are queued unconditionally and gated on that
// word, so the host never waits.  A matrix is accepted only if no pair of eigenvalues closer than
// omega is coupled by more than rounding noise in S -- otherwise vectors inside a near-degenerate
// cluster would be left unrotated -- such cases go to Jacobi, which resolves them.
#include "nbx_common.h"

namespace {

constexpr int RF_THREADS = 1024;
constexpr int RF_MAX_ITER = 3;
constexpr int RF_UNROLL = 8;
constexpr double RF_TOL = 3.0e-8;       // accepted update has max|E| below this: error ~ tol^2
constexpr double RF_GIVE_UP = 0.25;     // not in the contracting regime
constexpr double RF_CLUSTER_NOISE = 1.0e-14;

// One workgroup per matrix.  G = X^T X, S = X^T A X  ->  Ep = I + E, lambda, status.
__global__ __launch_bounds__(RF_THREADS) void refine_e_kernel(int N, const double* __restrict__ A,
                                                              const double* __restrict__ S,
                                                              const double* __restrict__ G, double* __restrict__ Ep,
                                                              double* __restrict__ lam_out, int* __restrict__ status,
                                                              double* __restrict__ norm_a, int iter,
                                                              int max_iter) {
    extern __shared__ double sm[];
    double* lam = sm;
    double* red = sm + N;
    const int b = blockIdx.x;
    if (iter > 0 && status[b] != 0) return;
    const int64_t n2 = (int64_t)N * N;
    A += b * n2;
    S += b * n2;
    G += b * n2;
    Ep += b * n2;
    for (int i = threadIdx.x; i < N; i += RF_THREADS) lam[i] = S[(int64_t)i * N + i] / G[(int64_t)i * N + i];
    double off2 = 0.0, r2 = 0.0, a2 = 0.0;
    // flat element index e = tid + k * RF_THREADS, RF_UNROLL of them per trip so that their loads
    // are all in flight together (the matrices sit in L2; one workgroup is latency bound otherwise)
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int total = N * N;
    const float inv_n = 1.0f / (float)N;
    auto row_of = [&](int e) {  // e / N for e < 2^24 without an integer division
        int i = (int)((float)e * inv_n);
        i -= (i * N > e) ? 1 : 0;
        i += ((i + 1) * N <= e) ? 1 : 0;
        return i;
    };
    for (int e0 = threadIdx.x; e0 < total; e0 += RF_THREADS * RF_UNROLL) {
        double sv[RF_UNROLL], gv[RF_UNROLL], av[RF_UNROLL];
#pragma unroll
        for (int u = 0; u < RF_UNROLL; ++u) {
            const int e = e0 + u * RF_THREADS;
            const bool in = e < total;
            sv[u] = in ? S[e] : 0.0;
            gv[u] = in ? G[e] : 0.0;
            av[u] = (in && iter == 0) ? A[e] : 0.0;
        }
#pragma unroll
        for (int u = 0; u < RF_UNROLL; ++u) {
            const int e = e0 + u * RF_THREADS;
            if (e < total) {
                const int i = row_of(e), j = e - i * N;
                const double r = (i == j) ? 1.0 - gv[u] : -gv[u];
                if (i != j) off2 = fma(sv[u], sv[u], off2);
                r2 = fma(r, r, r2);
                a2 = fma(av[u], av[u], a2);
            }
        }
    }
    // three sums with one pair of barriers: wave partials side by side in LDS
    off2 = nbx_wave_sum(off2);
    r2 = nbx_wave_sum(r2);
    a2 = nbx_wave_sum(a2);
    __syncthreads();
    if (lane == 0) {
        red[wave] = off2;
        red[16 + wave] = r2;
        red[32 + wave] = a2;
    }
    __syncthreads();
    off2 = r2 = a2 = 0.0;
    for (int w = 0; w < RF_THREADS / 64; ++w) {
        off2 += red[w];
        r2 += red[16 + w];
        a2 += red[32 + w];
    }
    __syncthreads();
    double na;
    if (iter == 0) {
        na = sqrt(a2);
        if (threadIdx.x == 0) norm_a[b] = na;
    } else {
        na = norm_a[b];
    }
    const double omega = 2.0 * (sqrt(off2) + na * sqrt(r2));
    double emax = 0.0, cmax = 0.0;
    for (int e0 = threadIdx.x; e0 < total; e0 += RF_THREADS * RF_UNROLL) {
        double sv[RF_UNROLL], gv[RF_UNROLL];
#pragma unroll
        for (int u = 0; u < RF_UNROLL; ++u) {
            const int e = e0 + u * RF_THREADS;
            const bool in = e < total;
            sv[u] = in ? S[e] : 0.0;
            gv[u] = in ? G[e] : 0.0;
        }
#pragma unroll
        for (int u = 0; u < RF_UNROLL; ++u) {
            const int e = e0 + u * RF_THREADS;
            if (e < total) {
                const int i = row_of(e), j = e - i * N;
                const double sx = sv[u], g = gv[u];
                double ev;
                if (i == j) {
                    ev = 0.5 * (1.0 - g);
                    emax = fmax(emax, fabs(ev));
                    Ep[e] = 1.0 + ev;
                } else {
                    const double lj = lam[j];
                    const double d = lj - lam[i];
                    if (fabs(d) > omega) {
                        ev = (sx - lj * g) / d;  // R_ij = -G_ij
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
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        emax = fmax(emax, __shfl_xor(emax, o));
        cmax = fmax(cmax, __shfl_xor(cmax, o));
    }
    if (lane == 0) {
        red[wave] = emax;
        red[16 + wave] = cmax;
    }
    __syncthreads();
    for (int w = 0; w < RF_THREADS / 64; ++w) {
        emax = fmax(emax, red[w]);
        cmax = fmax(cmax, red[16 + w]);
    }
    if (threadIdx.x == 0) {
        int st = 0;
        if (!(emax < RF_GIVE_UP)) st = -1;  // also NaN
        else if (emax < RF_TOL && cmax <= RF_CLUSTER_NOISE * na) st = iter + 1;
        else if (iter == max_iter - 1) st = -1;
        status[b] = st;
#ifdef NBX_REFINE_DEBUG
        printf("refine b=%d iter=%d emax=%.3e cmax=%.3e omega=%.3e na=%.3e off=%.3e r=%.3e st=%d\n", b, iter, emax, cmax, omega, na, sqrt(off2), sqrt(r2), st);
#endif
    }
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
    if (st <= 0) return;
    const int64_t n2 = (int64_t)N * N;
    const double* X = ((st & 1) ? xb1 : xb0) + b * n2;
    for (int i = threadIdx.x; i < N; i += blockDim.x) lam[i] = lam_in[(int64_t)b * N + i];
    __syncthreads();
    for (int i = threadIdx.x; i < N; i += blockDim.x) {
        const double li = lam[i];
        int r = 0;
        for (int j = 0; j < N; ++j) {
            const double lj = lam[j];
            r += (lj < li || (lj == li && j < i)) ? 1 : 0;
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
    size_t xb0, xb1, y, s, g, ep, lam, norm, status, total;
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
    L.status = off; off += align256((size_t)batch * sizeof(int));
    L.total = off;
    return L;
}

}  // namespace

bool nbx_eigh_refine_supported(int64_t n, int64_t batch) {
    return n >= 2 && n <= 4096 && nbx_gemm_small_supported(n, n, n, batch) && (size_t)(n + 48) * sizeof(double) <= 64 * 1024;
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
    int* status = reinterpret_cast<int*>(base + L.status);
    const int64_t n2 = n * n;
    for (int it = 0; it < max_iter; ++it) {
        // iteration `it` reads src and writes dst; the first one reads V0 in place and its
        // GEMMs are not gated (its E kernel initialises the status words)
        const double* src = (it == 0) ? d_v0 : xb[it & 1];
        double* dst = xb[(it + 1) & 1];
        const int* gate = (it == 0) ? nullptr : status;
        int rc = nbx_gemm_small_gated(ctx, 'N', 'N', n, n, n, 1.0, d_a, n, n2, src, n, n2, 0.0, y, n, n2, batch, gate, 0, 0);
        if (rc != NBX_OK) return rc;
        // S = X^T Y and G = X^T X in one launch (they share op(A) = X^T)
        rc = nbx_gemm_small_gated(ctx, 'T', 'N', n, n, n, 1.0, src, n, n2, y, n, n2, 0.0, s, n, n2, batch, gate, 0, 0,
                                  src, g);
        if (rc != NBX_OK) return rc;
        hipLaunchKernelGGL(refine_e_kernel, dim3((unsigned)batch), dim3(RF_THREADS), (size_t)(n + 48) * sizeof(double),
                           ctx->stream, (int)n, d_a, s, g, ep, lam, status, norm, it, max_iter);
        NBX_LAUNCH_CHECK();
        rc = nbx_gemm_small_gated(ctx, 'N', 'N', n, n, n, 1.0, src, n, n2, ep, n, n2, 0.0, dst, n, n2, batch, status, 0,
                                  it + 1);
        if (rc != NBX_OK) return rc;
    }
    hipLaunchKernelGGL(refine_finish_kernel, dim3((unsigned)nbx_cdiv(n, RF_ROWS), (unsigned)batch), dim3(256),
                       (size_t)n * sizeof(double) + (size_t)n * sizeof(int), ctx->stream, (int)n, xb[0], xb[1], lam,
                       status, d_w, d_v, d_jacobi_status);
    NBX_LAUNCH_CHECK();
    *d_status_out = status;
    return NBX_OK;
}
