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
//
// Two kernels.  svd_lds_kernel (n <= 196 and the transposed working matrix <= 150 KB, i.e. every
// SVD of the embedding path at the benchmark sizes) keeps the columns in LDS and rotates them in
// place: the pairs of a step come from the tournament ring by index arithmetic (jacobi_ring.h), a
// 16-lane group owns a pair (64 pairs per round), one LDS-only barrier per step, and the right
// vectors are not touched at all -- only (c, s) per pair and step is recorded and the replay
// kernel of the eigensolver (one wavefront per row of V, all CUs) builds V^T afterwards.
// svd_jacobi_kernel (any size) is the global-memory fallback: ~25x slower at 115 x 115 (one CU's
// L2 port).
#include "jacobi_ring.h"
#include "nbx_common.h"

namespace {

constexpr int SVD_THREADS = 1024;
constexpr int SL_MAX_NP = 196;
constexpr int SL_MAX_ELEMS = 19200;  // NP * M doubles of LDS (150 KB)
constexpr int SL_MAX_SWEEPS = 40;

// Sum over the 16 lanes of a DPP row (all 16 lanes get the total): quad_perm butterflies, then the
// half-row and row mirrors -- VALU cross-lane moves, no ds_bpermute round trips.
template <int CTRL>
__device__ __forceinline__ double dpp_f64(double v) {
    const long long bits = __builtin_bit_cast(long long, v);
    const int lo = (int)(bits & 0xffffffffll), hi = (int)(bits >> 32);
    const int lo2 = __builtin_amdgcn_update_dpp(0, lo, CTRL, 0xf, 0xf, false);
    const int hi2 = __builtin_amdgcn_update_dpp(0, hi, CTRL, 0xf, 0xf, false);
    return __builtin_bit_cast(double, ((long long)(unsigned)hi2 << 32) | (long long)(unsigned)lo2);
}
__device__ __forceinline__ double group16_sum(double v) {
    v += dpp_f64<0xB1>(v);   // quad_perm [1,0,3,2]
    v += dpp_f64<0x4E>(v);   // quad_perm [2,3,0,1]
    v += dpp_f64<0x141>(v);  // row_half_mirror
    v += dpp_f64<0x140>(v);  // row_mirror
    return v;
}

__global__ __launch_bounds__(SVD_THREADS) void svd_lds_kernel(const double* __restrict__ a, int M, int N, int NP,
                                                              int steps, double2* __restrict__ rot,
                                                              int* __restrict__ any_flags, int* __restrict__ nsteps_out,
                                                              double* __restrict__ s_out, int* __restrict__ rank_out,
                                                              int* __restrict__ status) {
    extern __shared__ __attribute__((aligned(16))) double smem[];
    double* Gt = smem;                                   // [NP][M]: column j of A is row j
    double* sig = Gt + (size_t)NP * M;                   // [NP]
    double* red = sig + NP;                              // [17]
    int* nrot = reinterpret_cast<int*>(red + 17);        // [1]
    const int tid = threadIdx.x, lane = tid & 63;
    const int grp = tid >> 4, gl = tid & 15;             // 64 groups of 16 lanes: one pair each
    const int mp = NP / 2, R = NP - 1;

    double fro = 0.0;
    for (int idx = tid; idx < NP * M; idx += SVD_THREADS) {
        const int i = idx / NP, j = idx - i * NP;        // consecutive threads: consecutive columns of a row of A
        const double v = (j < N) ? a[(int64_t)i * N + j] : 0.0;
        Gt[(size_t)j * M + i] = v;
        fro = fma(v, v, fro);
    }
    fro = nbx_block_sum(fro, red);
    const double eps = 2.220446049250313e-16;
    const double tol = eps * sqrt((double)(M > 1 ? M : 1));
    // columns with ||g|| <= max(m,n) eps ||A||_F are numerically zero: pairs involving one are left alone
    const double dim = (double)(M > N ? M : N);
    const double floor2 = dim * dim * eps * eps * fro;
    __syncthreads();

    auto index_at = [&](int pos, int t) {  // original column at ring position `pos` after t turns
        if (pos < 0) return 0;
        int q = pos - t;
        if (q < 0) q += R;
        return ring_index0(q, mp);
    };

    int sweep = 0;
    bool converged = false;
    for (; sweep < SL_MAX_SWEEPS && !converged; ++sweep) {
        if (tid == 0) nrot[0] = 0;
        __syncthreads();
        for (int step = 0; step < steps; ++step) {
            const int gs = sweep * steps + step;
            bool rotated = false;
            for (int k = grp; k < mp; k += SVD_THREADS / 16) {
                const int p = index_at(k == 0 ? -1 : ring_pos_top(k), step);
                const int q = index_at(ring_pos_bot(k, mp), step);
                double* gp = Gt + (size_t)p * M;
                double* gq = Gt + (size_t)q * M;
                double al = 0.0, be = 0.0, ga = 0.0;
                for (int i = gl; i < M; i += 16) {
                    const double x = gp[i], y = gq[i];
                    al = fma(x, x, al);
                    be = fma(y, y, be);
                    ga = fma(x, y, ga);
                }
                al = group16_sum(al);
                be = group16_sum(be);
                ga = group16_sum(ga);
                double c = 1.0, sn = 0.0;
                const bool rotate = fabs(ga) > tol * sqrt(al * be) && al > floor2 && be > floor2;
                if (rotate) {
                    const double zeta = (be - al) / (2.0 * ga);
                    const double az = fabs(zeta);
                    double t = (az > 1.0e150) ? 0.5 / az : 1.0 / (az + sqrt(az * az + 1.0));
                    if (zeta < 0.0) t = -t;
                    c = 1.0 / sqrt(t * t + 1.0);
                    sn = t * c;
                    for (int i = gl; i < M; i += 16) {
                        const double x = gp[i], y = gq[i];
                        gp[i] = c * x - sn * y;
                        gq[i] = sn * x + c * y;
                    }
                    rotated = true;
                }
                if (gl == 0) rot[(int64_t)gs * mp + k] = make_double2(c, sn);
            }
            // one LDS atomic and one flag store per wavefront that rotated something
            const unsigned long long any = __ballot(rotated);
            if (any != 0ull && lane == 0) {
                atomicAdd(nrot, 1);
                any_flags[gs] = 1;
            }
            asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");  // LDS only: the log stores stay in flight
        }
        __syncthreads();
        converged = (nrot[0] == 0);
        __syncthreads();
    }

    // singular values = column norms; the padding column (index N when n is odd) is forced last
    for (int j = grp; j < NP; j += SVD_THREADS / 16) {
        double acc = 0.0;
        for (int i = gl; i < M; i += 16) {
            const double x = Gt[(size_t)j * M + i];
            acc = fma(x, x, acc);
        }
        acc = group16_sum(acc);
        if (gl == 0) sig[j] = (j < N) ? sqrt(acc) : -1.0;
    }
    __syncthreads();
    for (int i = tid; i < NP; i += SVD_THREADS) {
        const double si = sig[i];
        int rk = 0;
        for (int j = 0; j < NP; ++j) {
            const double sj = sig[j];
            rk += (sj > si || (sj == si && j < i)) ? 1 : 0;
        }
        rank_out[i] = rk;
        const int nsv = M < N ? M : N;
        if (rk < nsv) s_out[rk] = si;
    }
    if (tid == 0) {
        nsteps_out[0] = sweep * steps;
        status[0] = converged ? sweep : -sweep;
    }
}

bool svd_lds_fits(int64_t m, int64_t n) {
    const int64_t np = (n + 1) & ~1ll;
    return np <= SL_MAX_NP && np * m <= SL_MAX_ELEMS;
}

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

// workspace: [ping-pong G and V of the fallback | status word | rotation log, step flags, step count, ranks]
static size_t svd_status_offset(int64_t m, int64_t n) {
    const int64_t np = (n + 1) & ~1ll;
    return align256((size_t)(2 * np * m + 2 * np * np) * sizeof(double));
}

extern "C" size_t nbx_svd_worksize(int64_t m, int64_t n) {
    if (m <= 0 || n <= 0) return 0;
    const int64_t np = (n + 1) & ~1ll;
    size_t total = svd_status_offset(m, n) + 256;
    if (svd_lds_fits(m, n)) {
        const int64_t steps = np == 2 ? 1 : np - 1;
        total += align256((size_t)(SL_MAX_SWEEPS * steps * (np / 2)) * sizeof(double2)) +
                 align256((size_t)(SL_MAX_SWEEPS * steps) * sizeof(int)) + 256 + align256((size_t)np * sizeof(int));
    }
    return total;
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
    int* status = reinterpret_cast<int*>(static_cast<char*>(d_work) + svd_status_offset(m, n));
    if (svd_lds_fits(m, n)) {
        const int64_t steps = np == 2 ? 1 : np - 1;
        char* base = static_cast<char*>(d_work) + svd_status_offset(m, n) + 256;
        double2* rot = reinterpret_cast<double2*>(base);
        base += align256((size_t)(SL_MAX_SWEEPS * steps * (np / 2)) * sizeof(double2));
        int* flags = reinterpret_cast<int*>(base);
        const size_t flag_bytes = align256((size_t)(SL_MAX_SWEEPS * steps) * sizeof(int));
        base += flag_bytes;
        int* nsteps = reinterpret_cast<int*>(base);
        base += 256;
        int* rank = reinterpret_cast<int*>(base);
        int rc = nbx_memset(ctx, flags, 0, flag_bytes);
        if (rc != NBX_OK) return rc;
        const size_t lds = (size_t)(np * m + np + 17) * sizeof(double) + 16;
        static bool attr_set = false;
        if (!attr_set) {
            NBX_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(svd_lds_kernel),
                                        hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
            attr_set = true;
        }
        {
            nbx_prof_scope prof(ctx, NBX_PROF_SVD);
            hipLaunchKernelGGL(svd_lds_kernel, dim3(1), dim3(SVD_THREADS), lds, ctx->stream, d_a, (int)m, (int)n, (int)np,
                               (int)steps, rot, flags, nsteps, d_s, rank, status);
            NBX_LAUNCH_CHECK();
            rc = nbx_apply_rotation_log_t(ctx, (int)n, (int)np, (int)steps, rot, flags, nsteps, rank, d_vt);
        }
        return rc;
    }
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
    const char* status = static_cast<const char*>(d_work) + svd_status_offset(m, n);
    int rc = nbx_memcpy_d2h(ctx, h_sweeps, status, sizeof(int));
    if (rc != NBX_OK) return rc;
    if (h_sweeps[0] <= 0) {
        nbx_set_error("nbx_svd_right: no convergence in %d sweeps", -h_sweeps[0]);
        return NBX_E_NOCONV;
    }
    return NBX_OK;
}
