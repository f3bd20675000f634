// libnbx: singular values + right singular vectors (include/nbx.h "SVD").
//
// One-sided (Hestenes) Jacobi: the columns of A (m x n) are rotated in pairs until they
// are mutually orthogonal, G = A V; then sigma_j = ||g_j|| and the accumulated V holds the
// right singular vectors, including an orthonormal basis of the null space when m < n
// (np.linalg.svd full_matrices=True semantics, which spade.py:132-134 relies on).
// One-sided Jacobi keeps tiny singular values to high RELATIVE accuracy, which the
// `sigma >= 1e-15` shell test of concentric.py:164,211 depends on.
//
// Layout: the working arrays hold columns as contiguous ROWS (Gt: NP x m, Vw: NP x NP), in
// the same physically-permuted round-robin order as eigh.hip, so a pivot pair is two
// adjacent rows owned by one wavefront: three wavefront-reduced dot products, one rotation,
// rows written to their next-step positions (ping-pong), one barrier per step.
#include "nbx_common.h"

namespace {

constexpr int SVD_THREADS = 1024;
constexpr int SVD_WAVES = SVD_THREADS / 64;
constexpr int SVD_MAX_SWEEPS = 60;

__device__ __forceinline__ int rr_next(int i, int m) {
    if (m == 1) return i;
    const int k = i >> 1;
    if ((i & 1) == 0) {
        if (k == 0) return 0;
        if (k == m - 1) return 2 * m - 1;
        return 2 * (k + 1);
    }
    if (k == 0) return 2;
    return 2 * k - 1;
}

__global__ __launch_bounds__(SVD_THREADS) void svd_jacobi_kernel(const double* __restrict__ a, int M, int N,
                                                                 double* __restrict__ s_out,
                                                                 double* __restrict__ vt_out,
                                                                 double* __restrict__ work,
                                                                 int* __restrict__ status) {
    extern __shared__ __attribute__((aligned(16))) double smem[];
    const int NP = (N + 1) & ~1;
    const int np = NP / 2;  // pairs
    double* sig = smem;                              // [NP]
    int* rank = reinterpret_cast<int*>(sig + NP);    // [NP]
    int& nrot = rank[NP];
    double* red = reinterpret_cast<double*>(rank + NP + 2);  // [17] (8-byte aligned: NP even)

    double* G0 = work;
    double* G1 = G0 + (int64_t)NP * M;
    double* V0 = G1 + (int64_t)NP * M;
    double* V1 = V0 + (int64_t)NP * NP;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;

    // Gt[j][i] = A[i][j]; Vw = I; ||A||_F^2
    double fro = 0.0;
    for (int64_t idx = tid; idx < (int64_t)NP * M; idx += SVD_THREADS) {
        const int j = (int)(idx / M), i = (int)(idx - (int64_t)j * M);
        const double v = (j < N) ? a[(int64_t)i * N + j] : 0.0;
        G0[idx] = v;
        fro = fma(v, v, fro);
    }
    for (int64_t idx = tid; idx < (int64_t)NP * NP; idx += SVD_THREADS) {
        const int r = (int)(idx / NP), c = (int)(idx - (int64_t)r * NP);
        V0[idx] = (r == c) ? 1.0 : 0.0;
    }
    fro = nbx_block_sum(fro, red);
    const double eps = 2.220446049250313e-16;
    const double tol = eps * sqrt((double)(M > 1 ? M : 1));
    // columns with ||g|| <= max(m,n) eps ||A||_F are numerically zero (the usual rank
    // tolerance): pairs involving one are left alone -- their V columns already span the
    // numerical null space and V stays orthogonal whatever we skip
    const double dim = (double)(M > N ? M : N);
    const double floor2 = dim * dim * eps * eps * fro;
    __syncthreads();

    const int steps = (np == 1) ? 1 : NP - 1;
    double* Gc = G0;
    double* Gn = G1;
    double* Vc = V0;
    double* Vn = V1;
    int sweep = 0;
    bool converged = false;
    for (; sweep < SVD_MAX_SWEEPS && !converged; ++sweep) {
        if (tid == 0) nrot = 0;
        __syncthreads();
        for (int step = 0; step < steps; ++step) {
            for (int k = wave; k < np; k += SVD_WAVES) {
                const double* gp = Gc + (int64_t)(2 * k) * M;
                const double* gq = gp + M;
                double al = 0.0, be = 0.0, ga = 0.0;
                for (int i = lane; i < M; i += 64) {
                    const double x = gp[i], y = gq[i];
                    al = fma(x, x, al);
                    be = fma(y, y, be);
                    ga = fma(x, y, ga);
                }
                al = nbx_wave_sum(al);
                be = nbx_wave_sum(be);
                ga = nbx_wave_sum(ga);
                double c = 1.0, s = 0.0;
                const bool rotate = fabs(ga) > tol * sqrt(al * be) && al > floor2 && be > floor2;
                if (rotate) {
                    const double zeta = (be - al) / (2.0 * ga);
                    const double az = fabs(zeta);
                    double t = (az > 1.0e150) ? 0.5 / az : 1.0 / (az + sqrt(az * az + 1.0));
                    if (zeta < 0.0) t = -t;
                    c = 1.0 / sqrt(t * t + 1.0);
                    s = t * c;
                    if (lane == 0) atomicAdd(&nrot, 1);
                }
                const int rp = rr_next(2 * k, np), rq = rr_next(2 * k + 1, np);
                double* gpn = Gn + (int64_t)rp * M;
                double* gqn = Gn + (int64_t)rq * M;
                for (int i = lane; i < M; i += 64) {
                    const double x = gp[i], y = gq[i];
                    gpn[i] = c * x - s * y;
                    gqn[i] = s * x + c * y;
                }
                const double* vp = Vc + (int64_t)(2 * k) * NP;
                const double* vq = vp + NP;
                double* vpn = Vn + (int64_t)rp * NP;
                double* vqn = Vn + (int64_t)rq * NP;
                for (int i = lane; i < NP; i += 64) {
                    const double x = vp[i], y = vq[i];
                    vpn[i] = c * x - s * y;
                    vqn[i] = s * x + c * y;
                }
            }
            __syncthreads();
            double* tG = Gc; Gc = Gn; Gn = tG;
            double* tV = Vc; Vc = Vn; Vn = tV;
        }
        converged = (nrot == 0);
        __syncthreads();
    }

    // singular values = row norms of Gt; the padded row (if any) is forced last
    for (int j = wave; j < NP; j += SVD_WAVES) {
        double acc = 0.0;
        for (int i = lane; i < M; i += 64) {
            const double x = Gc[(int64_t)j * M + i];
            acc = fma(x, x, acc);
        }
        acc = nbx_wave_sum(acc);
        if (lane == 0) sig[j] = sqrt(acc);
    }
    __syncthreads();
    // which working row is the padding?  the one whose Vw row has its weight on column N
    // (it never rotates, so it is exactly e_N); mark it with sigma = -1
    if (NP > N) {
        for (int j = tid; j < NP; j += SVD_THREADS)
            if (Vc[(int64_t)j * NP + N] != 0.0) sig[j] = -1.0;
        __syncthreads();
    }
    for (int i = tid; i < NP; i += SVD_THREADS) {
        const double si = sig[i];
        int rk = 0;
        for (int j = 0; j < NP; ++j) {
            const double sj = sig[j];
            rk += (sj > si || (sj == si && j < i)) ? 1 : 0;
        }
        rank[i] = rk;
        const int nsv = M < N ? M : N;
        if (rk < nsv) s_out[rk] = si;
    }
    __syncthreads();
    for (int64_t idx = tid; idx < (int64_t)NP * N; idx += SVD_THREADS) {
        const int j = (int)(idx / N), c = (int)(idx - (int64_t)j * N);
        const int rk = rank[j];
        if (rk < N) vt_out[(int64_t)rk * N + c] = Vc[(int64_t)j * NP + c];
    }
    if (tid == 0) status[0] = converged ? sweep : -sweep;
}

size_t align256(size_t x) { return (x + 255) & ~(size_t)255; }

}  // namespace

extern "C" size_t nbx_svd_worksize(int64_t m, int64_t n) {
    if (m <= 0 || n <= 0) return 0;
    const int64_t np = (n + 1) & ~1ll;
    return align256((size_t)(2 * np * m + 2 * np * np) * sizeof(double)) + 256;
}

extern "C" int nbx_svd_right(nbx_ctx* ctx, int64_t m, int64_t n, const double* d_a, double* d_s, double* d_vt,
                             void* d_work, size_t work_bytes) {
    NBX_CHECK_ARG(ctx && d_a && d_s && d_vt && m > 0 && n > 0);
    NBX_CHECK_ARG(n <= 4096 && m <= (1 << 20));
    const size_t need = nbx_svd_worksize(m, n);
    if (d_work == nullptr || work_bytes < need) {
        nbx_set_error("nbx_svd_right: workspace %zu < %zu bytes", work_bytes, need);
        return NBX_E_NOMEM;
    }
    const int64_t np = (n + 1) & ~1ll;
    int* status = reinterpret_cast<int*>(static_cast<char*>(d_work) +
                                         align256((size_t)(2 * np * m + 2 * np * np) * sizeof(double)));
    const size_t lds = (size_t)np * sizeof(double) + (size_t)(np + 2) * sizeof(int) + 17 * sizeof(double);
    {
        nbx_prof_scope prof(ctx, NBX_PROF_SVD);
        hipLaunchKernelGGL(svd_jacobi_kernel, dim3(1), dim3(SVD_THREADS), lds, ctx->stream, d_a, (int)m, (int)n, d_s,
                           d_vt, static_cast<double*>(d_work), status);
    }
    NBX_LAUNCH_CHECK();
    return NBX_OK;
}

extern "C" int nbx_svd_status(nbx_ctx* ctx, int64_t m, int64_t n, const void* d_work, int* h_sweeps) {
    NBX_CHECK_ARG(ctx && d_work && h_sweeps && m > 0 && n > 0);
    const int64_t np = (n + 1) & ~1ll;
    const char* status = static_cast<const char*>(d_work) + align256((size_t)(2 * np * m + 2 * np * np) * sizeof(double));
    int rc = nbx_memcpy_d2h(ctx, h_sweeps, status, sizeof(int));
    if (rc != NBX_OK) return rc;
    if (h_sweeps[0] <= 0) {
        nbx_set_error("nbx_svd_right: no convergence in %d sweeps", -h_sweeps[0]);
        return NBX_E_NOCONV;
    }
    return NBX_OK;
}
