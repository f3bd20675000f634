// libnbx: symmetric eigensolver for N > 196 (beyond the LDS-resident Jacobi of eigh_lds.hip).
//
// LAPACK-style pipeline (dsytrd / dstebz / dstein / dormtr restated for one GPU), followed by
// the Jacobi solver of eigh.hip as a polisher when the result is not yet at rounding level:
//   K1 tridiag_kernel      Householder reduction A = Q T Q^T, one workgroup per matrix; the
//                          full symmetric matrix stays in global memory and every pass over the
//                          trailing block is column-per-thread (coalesced rows), like jk.hip
//   K2 bisect_kernel       eigenvalues of T: one wavefront per eigenvalue, 64-way multisection
//                          on Sturm counts (9-10 rounds instead of 53 bisections)
//   K3 invit_kernel        eigenvectors of T by inverse iteration, one thread per eigenvalue
//                          (partial-pivoting LU of T - lambda I, three solves, hash start vector)
//   K4 backtransform_kernel  V = Q Z: Householder reflectors applied column-per-thread
//   K5 cgs2_kernel         re-orthonormalisation (classical Gram-Schmidt, twice), vectors as rows
// then R = V^T A V is formed with the MFMA GEMM; if its largest off-diagonal element is above
// 1e-13 ||A|| (clustered eigenvalues: inverse iteration does not separate them) the Jacobi
// solver is warm-started from V, which then needs one or two sweeps.
#include "nbx_common.h"
#include "synth_device.h"

// eigh_grid.hip: tridiagonalisation on the whole chip and blocked back-transformation (N > 198)
bool nbx_tdg_covers(int64_t n);
size_t nbx_tdg_work_doubles(int64_t n, int64_t batch);
int nbx_tdg_tridiag(nbx_ctx* ctx, int64_t n, int64_t batch, const double* d_a, double* d, double* e, double* tau, double* Vg,
                    double* work, int* status);
int nbx_tdg_backtransform(nbx_ctx* ctx, int64_t n, int64_t batch, const double* Vg, const double* tau, double* Z, double* work);

namespace {

constexpr int TD_THREADS = 1024;
constexpr int TD_MAXSEG = 2;  // column segments per thread: N <= 2048

size_t align256(size_t x) { return (x + 255) & ~(size_t)255; }

__device__ __forceinline__ double td_rcp(double x) {
    double y = __builtin_amdgcn_rcp(x);
    y = fma(y, fma(-x, y, 1.0), y);
    y = fma(y, fma(-x, y, 1.0), y);
    return y;
}

// ---------------------------------------------------------------- K1: Householder tridiagonalisation
// W: (N,N) full symmetric copy of A (lower triangle of the input mirrored), overwritten.
// d[N], e[N] (e[N-1] unused), tau[N], Vh[k][0..L): reflector k (v[0] = 1) acting on rows k+1..N-1.
__global__ __launch_bounds__(TD_THREADS) void tridiag_kernel(const double* __restrict__ a_in, int N,
                                                             double* __restrict__ Wb, double* __restrict__ db,
                                                             double* __restrict__ eb, double* __restrict__ taub,
                                                             double* __restrict__ Vhb) {
    extern __shared__ __attribute__((aligned(16))) double smem[];
    double* v = smem;            // [N]
    double* p = v + N;           // [N]
    double* w = p + N;           // [N]
    double* part = w + N;        // [TD_THREADS] partial sums of the symv
    double* red = part + TD_THREADS;  // [20]
    const int b = blockIdx.x;
    const int64_t n2 = (int64_t)N * N;
    a_in += b * n2;
    double* W = Wb + b * n2;
    double* d = db + (int64_t)b * N;
    double* e = eb + (int64_t)b * N;
    double* tau = taub + (int64_t)b * N;
    double* Vh = Vhb + b * n2;
    const int tid = threadIdx.x;

    for (int64_t idx = tid; idx < n2; idx += TD_THREADS) {
        const int i = (int)(idx / N), j = (int)(idx - (int64_t)i * N);
        W[idx] = (i >= j) ? a_in[idx] : a_in[(int64_t)j * N + i];  // UPLO = 'L'
    }
    __syncthreads();

    for (int k = 0; k < N - 1; ++k) {
        const int L = N - k - 1;
        const double* rowk = W + (int64_t)k * N + k + 1;  // = column k below the diagonal (symmetric)
        // ---- (a) reflector
        double ss = 0.0;
        for (int t = tid; t < L; t += TD_THREADS) {
            const double x = rowk[t];
            v[t] = x;
            if (t > 0) ss = fma(x, x, ss);
        }
        ss = nbx_block_sum(ss, red);  // contains the barriers that publish v[]
        double alpha = v[0], beta, tk, scale;
        if (ss == 0.0) {
            beta = alpha;
            tk = 0.0;
            scale = 0.0;
        } else {
            beta = -copysign(sqrt(fma(alpha, alpha, ss)), alpha);
            tk = (beta - alpha) / beta;
            scale = 1.0 / (alpha - beta);
        }
        __syncthreads();
        for (int t = tid; t < L; t += TD_THREADS) {
            const double vt = (t == 0) ? 1.0 : v[t] * scale;
            v[t] = vt;
            Vh[(int64_t)k * N + t] = vt;
        }
        if (tid == 0) {
            d[k] = W[(int64_t)k * N + k];
            e[k] = beta;
            tau[k] = tk;
        }
        __syncthreads();
        if (tk == 0.0) continue;  // uniform

        // column-per-thread layout over the trailing L x L block
        const int nseg = (L + TD_THREADS - 1) / TD_THREADS;  // 1 or 2
        const int G = nseg > 1 ? 1 : (TD_THREADS / L > 0 ? TD_THREADS / L : 1);
        const int g = nseg > 1 ? 0 : tid / L;
        const int jc = nseg > 1 ? tid : tid - g * L;
        const bool active = nseg > 1 ? true : (g < G);
        const double* W22 = W + (int64_t)(k + 1) * N + (k + 1);
        // ---- (b) p = tau * W22 v   (p_j = sum_i W22[i][j] v_i, symmetric)
        double acc[TD_MAXSEG] = {0.0, 0.0};
        if (active) {
#pragma unroll 8
            for (int i = g; i < L; i += G) {
                const double vi = v[i];
                const double* row = W22 + (int64_t)i * N;
#pragma unroll
                for (int sgm = 0; sgm < TD_MAXSEG; ++sgm) {
                    const int j = jc + sgm * TD_THREADS;
                    if (sgm < nseg && j < L) acc[sgm] = fma(row[j], vi, acc[sgm]);
                }
            }
        }
        if (nseg == 1) {
            part[tid] = active ? acc[0] : 0.0;
            __syncthreads();
            if (tid < L) {
                double t = 0.0;
                for (int gg = 0; gg < G; ++gg) t += part[gg * L + tid];
                p[tid] = tk * t;
            }
        } else {
#pragma unroll
            for (int sgm = 0; sgm < TD_MAXSEG; ++sgm) {
                const int j = jc + sgm * TD_THREADS;
                if (j < L) p[j] = tk * acc[sgm];
            }
        }
        __syncthreads();
        // ---- (c) w = p - (tau/2) (p^T v) v
        double dot = 0.0;
        for (int t = tid; t < L; t += TD_THREADS) dot = fma(p[t], v[t], dot);
        dot = nbx_block_sum(dot, red);
        const double alpha2 = -0.5 * tk * dot;
        for (int t = tid; t < L; t += TD_THREADS) w[t] = fma(alpha2, v[t], p[t]);
        __syncthreads();
        // ---- (d) W22 -= v w^T + w v^T
        if (active) {
            double vj[TD_MAXSEG], wj[TD_MAXSEG];
#pragma unroll
            for (int sgm = 0; sgm < TD_MAXSEG; ++sgm) {
                const int j = jc + sgm * TD_THREADS;
                vj[sgm] = (sgm < nseg && j < L) ? v[j] : 0.0;
                wj[sgm] = (sgm < nseg && j < L) ? w[j] : 0.0;
            }
#pragma unroll 8
            for (int i = g; i < L; i += G) {
                const double vi = v[i], wi = w[i];
                double* row = W + (int64_t)(k + 1 + i) * N + (k + 1);
#pragma unroll
                for (int sgm = 0; sgm < TD_MAXSEG; ++sgm) {
                    const int j = jc + sgm * TD_THREADS;
                    if (sgm < nseg && j < L) row[j] -= fma(vi, wj[sgm], wi * vj[sgm]);
                }
            }
        }
        __syncthreads();
    }
    if (tid == 0) {
        d[N - 1] = W[(int64_t)(N - 1) * N + (N - 1)];
        e[N - 1] = 0.0;
        tau[N - 1] = 0.0;
    }
}

// ---------------------------------------------------------------- K1 for N <= 198: the matrix in registers
// The same reduction with the whole (full, symmetric) matrix held in the registers of one 512-thread workgroup
// (eight waves: 256 registers each and a cheap barrier).  Thread (gr, gc) of a GR x GC grid keeps the TR x TC
// elements A[gr + r GR][gc + s GC] -- interleaved in both directions, so the shrinking trailing block stays
// spread over all threads, and 2-D, so that a step reads TR + TC LDS values per TR TC elements, not 2 per element.
// A step touches no memory but LDS vectors and is arranged for THREE barriers (tridiag_kernel: twelve, and two
// passes over the trailing block in L2 -- 7.4 us per step at N = 148):
//   1  row k (= column k: the raw reflector x) and row k+1 go to LDS from the registers that hold them
//   2  partial sums of q = A22 x over each thread's rows, and of |x|^2, to LDS.  With v = (x - beta e1)/(alpha -
//      beta) the product A22 v = (q - beta A22[:,1]) / (alpha - beta) needs no second pass once beta is known
//   3  everyone adds up |x|^2 -> beta, tau; the first N threads finish p = tau A22 v, publish v, p and the partial
//      sums of p^T v;  then every thread forms w = p - (tau/2)(p^T v) v for its rows and columns and updates its
//      registers:  A22 -= v w^T + w v^T.
// GR, GC, TR, TC are compile-time: the LDS address of "my row r" is one register plus an immediate (with run-time
// strides the compiler keeps an address register per row and per vector alive across the k loop and spills).
// Sums run in a fixed order: bit-reproducible.  Outputs as tridiag_kernel (d, e, tau, Vh); W is not written.
constexpr int TDR_THREADS = 512;
constexpr int TDR_LD = 224;  // length of the LDS vectors (>= GR TR, GC TC of every instance)

template <int TR, int TC, int GR, int GC>
__global__ __launch_bounds__(TDR_THREADS) void tridiag_reg_kernel(const double* __restrict__ a_in, int N,
                                                                 double* __restrict__ db, double* __restrict__ eb,
                                                                 double* __restrict__ taub, double* __restrict__ Vhb) {
    static_assert(GR * GC <= TDR_THREADS && GR * TR <= TDR_LD && GC * TC <= TDR_LD, "instance geometry");
    __shared__ __attribute__((aligned(16))) double xs[TDR_LD];         // raw reflector: row k (index = column)
    __shared__ __attribute__((aligned(16))) double a1[TDR_LD];         // row k+1 (= column k+1 of the trailing block)
    __shared__ __attribute__((aligned(16))) double vs[TDR_LD];         // v
    __shared__ __attribute__((aligned(16))) double ps[TDR_LD];         // p
    __shared__ __attribute__((aligned(16))) double part[GR * TDR_LD];  // partial q per row group
    __shared__ double ssp[TDR_THREADS / 64], dotp[TDR_THREADS / 64];   // per-wave |x|^2, p^T v
    const int b = blockIdx.x;
    const int64_t n2 = (int64_t)N * N;
    a_in += b * n2;
    double* d = db + (int64_t)b * N;
    double* e = eb + (int64_t)b * N;
    double* tau = taub + (int64_t)b * N;
    double* Vh = Vhb + b * n2;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int gr = tid / GC, gc = tid - gr * GC;
    const bool active = gr < GR;

    double a[TR][TC];
#pragma unroll
    for (int r = 0; r < TR; ++r)
#pragma unroll
        for (int sc = 0; sc < TC; ++sc) {
            const int i = gr + r * GR, c = gc + sc * GC;
            a[r][sc] = (active && i < N && c < N) ? ((i >= c) ? a_in[(int64_t)i * N + c] : a_in[(int64_t)c * N + i])
                                                   : 0.0;  // UPLO = 'L'
        }

    for (int k = 0; k < N - 1; ++k) {
        // ---- 1: rows k and k+1 out of the registers
        {
            const int g0 = k % GR, r0 = k / GR, g1 = (k + 1) % GR, r1 = (k + 1) / GR;
            if (active && (gr == g0 || gr == g1)) {
#pragma unroll
                for (int sc = 0; sc < TC; ++sc) {
                    const int c = gc + sc * GC;
                    double x0 = 0.0, x1 = 0.0;
#pragma unroll
                    for (int r = 0; r < TR; ++r) {
                        x0 = (r == r0) ? a[r][sc] : x0;
                        x1 = (r == r1) ? a[r][sc] : x1;
                    }
                    if (gr == g0) {
                        if (c > k && c < N) xs[c] = x0;
                        if (c == k) d[k] = x0;
                    }
                    if (gr == g1 && c > k && c < N) a1[c] = x1;
                }
            }
        }
        __syncthreads();
        // ---- 2: partial q_c = sum_{i > k, i in my rows} A[i][c] x_i ; |x|^2 without its first element
        {
            if (active) {
                double q[TC];
#pragma unroll
                for (int sc = 0; sc < TC; ++sc) q[sc] = 0.0;
#pragma unroll
                for (int r = 0; r < TR; ++r) {
                    const int i = gr + r * GR;
                    const double xi = (i > k && i < N) ? xs[i] : 0.0;
#pragma unroll
                    for (int sc = 0; sc < TC; ++sc) q[sc] = fma(a[r][sc], xi, q[sc]);
                }
#pragma unroll
                for (int sc = 0; sc < TC; ++sc) part[gr * TDR_LD + gc + sc * GC] = q[sc];
            }
            double ss = 0.0;
            if (tid < N && tid > k + 1) ss = xs[tid] * xs[tid];
            ss = nbx_wave_sum(ss);
            if (lane == 0) ssp[wave] = ss;
        }
        __syncthreads();
        // ---- 3: beta, tau; p, v, p^T v
        double ss = 0.0;
#pragma unroll
        for (int w = 0; w < TDR_THREADS / 64; ++w) ss += ssp[w];
        const double alpha = xs[k + 1];
        double beta, tk, scale;
        if (ss == 0.0) {
            beta = alpha;
            tk = 0.0;
            scale = 0.0;
        } else {
            beta = -copysign(sqrt(fma(alpha, alpha, ss)), alpha);
            tk = (beta - alpha) / beta;
            scale = 1.0 / (alpha - beta);
        }
        {
            double pv = 0.0;
            if (tid < N && tid > k) {
                double q = 0.0;
#pragma unroll
                for (int gg = 0; gg < GR; ++gg) q += part[gg * TDR_LD + tid];
                const double vc = (tid == k + 1) ? 1.0 : xs[tid] * scale;
                const double pc = tk * scale * fma(-beta, a1[tid], q);
                vs[tid] = vc;
                ps[tid] = pc;
                Vh[(int64_t)k * N + (tid - k - 1)] = vc;
                pv = pc * vc;
            }
            pv = nbx_wave_sum(pv);
            if (lane == 0) dotp[wave] = pv;
            if (tid == 0) {
                e[k] = beta;
                tau[k] = tk;
            }
        }
        __syncthreads();
        if (tk == 0.0) continue;  // uniform: H = I
        double dot = 0.0;
#pragma unroll
        for (int w = 0; w < TDR_THREADS / 64; ++w) dot += dotp[w];
        const double alpha2 = -0.5 * tk * dot;
        if (active) {
            // rows and columns at or before k get v = w = 0: their elements (already final) stay as they are
            double vcol[TC], wcol[TC];
#pragma unroll
            for (int sc = 0; sc < TC; ++sc) {
                const int c = gc + sc * GC;
                const bool in = c > k && c < N;
                vcol[sc] = in ? vs[c] : 0.0;
                wcol[sc] = in ? fma(alpha2, vcol[sc], ps[c]) : 0.0;
            }
#pragma unroll
            for (int r = 0; r < TR; ++r) {
                const int i = gr + r * GR;
                const bool in = i > k && i < N;
                const double vi = in ? vs[i] : 0.0;
                const double wi = in ? fma(alpha2, vi, ps[i]) : 0.0;
#pragma unroll
                for (int sc = 0; sc < TC; ++sc) a[r][sc] -= fma(vi, wcol[sc], wi * vcol[sc]);
            }
        }
        // (the next step's phase 1 writes xs / a1 only, which nobody reads after barrier 3)
    }
    // the last diagonal element
    {
        const int g0 = (N - 1) % GR, r0 = (N - 1) / GR;
        if (active && gr == g0) {
#pragma unroll
            for (int sc = 0; sc < TC; ++sc) {
                double x0 = 0.0;
#pragma unroll
                for (int r = 0; r < TR; ++r) x0 = (r == r0) ? a[r][sc] : x0;
                if (gc + sc * GC == N - 1) {
                    d[N - 1] = x0;
                    e[N - 1] = 0.0;
                    tau[N - 1] = 0.0;
                }
            }
        }
    }
}

static bool tridiag_reg_launch(nbx_ctx* ctx, int N, int64_t batch, const double* d_a, double* d, double* e, double* tau,
                               double* Vh) {
#define NBX_TDR_GO(TR_, TC_, GR_, GC_)                                                                              \
    hipLaunchKernelGGL((tridiag_reg_kernel<TR_, TC_, GR_, GC_>), dim3((unsigned)batch), dim3(TDR_THREADS), 0,       \
                       ctx->stream, d_a, N, d, e, tau, Vh)
    if (N < 2) return false;
    if (N <= 64) NBX_TDR_GO(4, 2, 16, 32);
    else if (N <= 128) NBX_TDR_GO(8, 4, 16, 32);
    else if (N <= 150) NBX_TDR_GO(8, 6, 19, 25);
    else if (N <= 176) NBX_TDR_GO(8, 8, 22, 23);
    else if (N <= 198) NBX_TDR_GO(9, 9, 22, 23);
    else return false;  // (a 10 x 10 tile spills: slower than the in-memory kernel)
#undef NBX_TDR_GO
    return true;
}

// ---------------------------------------------------------------- K2: eigenvalues of T (Sturm multisection)
__device__ __forceinline__ int sturm_count(const double* __restrict__ d, const double* __restrict__ e2, int N,
                                           double x, double pivmin) {
    double q = d[0] - x;
    if (fabs(q) < pivmin) q = -pivmin;
    int cnt = q < 0.0 ? 1 : 0;
    for (int i = 1; i < N; ++i) {
        q = d[i] - x - e2[i - 1] * td_rcp(q);
        if (fabs(q) < pivmin) q = -pivmin;
        cnt += q < 0.0 ? 1 : 0;
    }
    return cnt;
}

// grid (ceil(N / W), batch), 64 W threads: wavefront w of a workgroup finds eigenvalue W blockIdx.x + w (ascending).  The
// W wavefronts share ONE copy of d and e^2 in LDS (every one of them writes it, with the same values, and forms the
// Gershgorin bounds over all of it as the one-wavefront form did: same bits): at N = 2000 the 32 KB copy per 64 threads
// left five wavefronts on a CU and the kernel took 4.4 ms; W = 8 fills it.
__global__ __launch_bounds__(512) void bisect_kernel(const double* __restrict__ db, const double* __restrict__ eb, int N,
                                                     double* __restrict__ wb) {
    extern __shared__ __attribute__((aligned(16))) double smem[];
    double* d = smem;        // [N]
    double* e2 = d + N;      // [N]
    const int b = blockIdx.y, lane = threadIdx.x & 63;
    const int j = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    const double* dg = db + (int64_t)b * N;
    const double* eg = eb + (int64_t)b * N;
    double gl = 1.0e300, gu = -1.0e300, emax = 0.0, tnorm = 0.0;
    for (int i = lane; i < N; i += 64) {
        const double di = dg[i];
        const double el = i > 0 ? fabs(eg[i - 1]) : 0.0, er = i < N - 1 ? fabs(eg[i]) : 0.0;
        d[i] = di;
        e2[i] = i < N - 1 ? eg[i] * eg[i] : 0.0;
        gl = fmin(gl, di - el - er);
        gu = fmax(gu, di + el + er);
        emax = fmax(emax, er * er);
        tnorm = fmax(tnorm, fabs(di) + el + er);
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        gl = fmin(gl, __shfl_xor(gl, off, 64));
        gu = fmax(gu, __shfl_xor(gu, off, 64));
        emax = fmax(emax, __shfl_xor(emax, off, 64));
        tnorm = fmax(tnorm, __shfl_xor(tnorm, off, 64));
    }
    __syncthreads();
    const double eps = 2.220446049250313e-16, safmin = 2.2250738585072014e-308;
    const double pivmin = safmin * fmax(1.0, emax);
    double lo = gl - 2.0 * tnorm * eps * N - 2.0 * pivmin;
    double hi = gu + 2.0 * tnorm * eps * N + 2.0 * pivmin;
    for (int round = 0; round < 14; ++round) {
        const double width = hi - lo;
        if (width <= 2.0 * eps * fmax(fabs(lo), fabs(hi)) + 2.0 * pivmin) break;
        const double x = lo + width * ((double)(lane + 1) * (1.0 / 65.0));
        const int c = sturm_count(d, e2, N, x, pivmin);
        const unsigned long long mask = __ballot(c >= j + 1);
        const int first = mask ? __ffsll((long long)mask) - 1 : 64;
        const double xlo = first == 0 ? lo : __shfl(x, first - 1 > 63 ? 63 : first - 1, 64);
        const double xhi = first == 64 ? hi : __shfl(x, first > 63 ? 63 : first, 64);
        lo = xlo;
        hi = xhi;
    }
    if (lane == 0 && j < N) wb[(int64_t)b * N + j] = 0.5 * (lo + hi);
}

// ---------------------------------------------------------------- K3: eigenvectors of T (inverse iteration)
// One thread per eigenvalue: LU of T - lambda I with partial pivoting, three solves from a hash start vector.  The
// chain of N dependent steps per sweep is latency bound, so what matters is where the thread's six work arrays
// live: `row stride NT` elements apart in global memory (invit_kernel: 256 threads per workgroup, any N -- each
// step then waits for L2), or in LDS (invit_lds_kernel: 16 threads per workgroup, 6 N 16 doubles <= 160 KB, i.e.
// N <= 208 -- N/16 workgroups on as many CUs, three times faster at N = 148).
template <typename WorkPtr>
__device__ __forceinline__ void invit_solve(const double* __restrict__ d, const double* __restrict__ e, double lam, int N,
                                            int64_t NT, WorkPtr ua, WorkPtr ub, WorkPtr uc, WorkPtr lm, WorkPtr sw,
                                            WorkPtr x, int j, int b, double* __restrict__ zcol) {
    double tnorm = 0.0;
    for (int i = 0; i < N; ++i)
        tnorm = fmax(tnorm, fabs(d[i]) + (i > 0 ? fabs(e[i - 1]) : 0.0) + (i < N - 1 ? fabs(e[i]) : 0.0));
    const double eps = 2.220446049250313e-16;
    const double pivtol = fmax(eps * tnorm, 1.0e-300);

    // LU with partial pivoting of T - lam I  (LAPACK dlagtf restated); the pivots are kept as reciprocals
    double ai = d[0] - lam;
    double bi = N > 1 ? e[0] : 0.0;  // super-diagonal entry of the current row
#pragma unroll 16
    for (int i = 0; i < N - 1; ++i) {
        const double ci = e[i];                        // sub-diagonal entry (row i+1, col i)
        double anext = d[i + 1] - lam;                 // diagonal of row i+1
        const double bnext = i < N - 2 ? e[i + 1] : 0.0;  // super-diagonal of row i+1
        if (fabs(ai) >= fabs(ci) || fabs(ci) < pivtol) {
            double piv = ai;
            if (fabs(piv) < pivtol) piv = copysign(pivtol, piv == 0.0 ? 1.0 : piv);
            const double rp = 1.0 / piv;
            const double mult = ci * rp;
            ua[(int64_t)i * NT] = rp;
            ub[(int64_t)i * NT] = bi;
            uc[(int64_t)i * NT] = 0.0;
            lm[(int64_t)i * NT] = mult;
            sw[(int64_t)i * NT] = 0.0;
            ai = anext - mult * bi;
            bi = bnext;
        } else {
            const double rp = 1.0 / ci;
            const double mult = ai * rp;
            ua[(int64_t)i * NT] = rp;
            ub[(int64_t)i * NT] = anext;
            uc[(int64_t)i * NT] = bnext;
            lm[(int64_t)i * NT] = mult;
            sw[(int64_t)i * NT] = 1.0;
            ai = bi - mult * anext;
            bi = -mult * bnext;
        }
    }
    if (fabs(ai) < pivtol) ai = copysign(pivtol, ai == 0.0 ? 1.0 : ai);
    ua[(int64_t)(N - 1) * NT] = 1.0 / ai;
    ub[(int64_t)(N - 1) * NT] = 0.0;
    uc[(int64_t)(N - 1) * NT] = 0.0;

    // start vector: deterministic pseudo-random in (-1, 1)
    for (int i = 0; i < N; ++i) x[(int64_t)i * NT] = nbx_synth_val(7, (uint64_t)j * 4096u + (uint64_t)i, 12345u + b);

    for (int it = 0; it < 3; ++it) {
        // forward: apply the row operations of the factorisation to the right-hand side
        double yi = x[0];
#pragma unroll 16
        for (int i = 0; i < N - 1; ++i) {
            double yn = x[(int64_t)(i + 1) * NT];
            if (sw[(int64_t)i * NT] != 0.0) {
                const double t = yi;
                yi = yn;
                yn = t;
            }
            yn -= lm[(int64_t)i * NT] * yi;
            x[(int64_t)i * NT] = yi;
            yi = yn;
        }
        x[(int64_t)(N - 1) * NT] = yi;
        // backward substitution with the three diagonals of U
        double x1 = 0.0, x2 = 0.0, amax = 0.0;
#pragma unroll 16
        for (int i = N - 1; i >= 0; --i) {
            const double t = x[(int64_t)i * NT] - ub[(int64_t)i * NT] * x1 - uc[(int64_t)i * NT] * x2;
            const double xi = t * ua[(int64_t)i * NT];
            x[(int64_t)i * NT] = xi;
            x2 = x1;
            x1 = xi;
            amax = fmax(amax, fabs(xi));
        }
        const double s = amax > 0.0 ? 1.0 / amax : 1.0;
        for (int i = 0; i < N; ++i) x[(int64_t)i * NT] *= s;
    }
    double nrm = 0.0;
    for (int i = 0; i < N; ++i) nrm = fma(x[(int64_t)i * NT], x[(int64_t)i * NT], nrm);
    const double s = nrm > 0.0 ? 1.0 / sqrt(nrm) : 1.0;
    for (int i = 0; i < N; ++i) zcol[(int64_t)i * N] = x[(int64_t)i * NT] * s;
}

__global__ __launch_bounds__(256) void invit_kernel(const double* __restrict__ db, const double* __restrict__ eb,
                                                    const double* __restrict__ wb, int N, double* __restrict__ scratch,
                                                    double* __restrict__ Zb) {
    const int b = blockIdx.y;
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= N) return;
    const int64_t NT = N;  // stride between consecutive rows of one thread's arrays
    double* base = scratch + (int64_t)b * 6 * N * NT;
    // (restrict: the arrays do not overlap, so the loads of the next rows can be issued ahead of the recurrence)
    double* __restrict__ ua = base + j;              // 1 / (U diagonal)
    double* __restrict__ ub = ua + (int64_t)N * NT;  // U first superdiagonal
    double* __restrict__ uc = ub + (int64_t)N * NT;  // U second superdiagonal
    double* __restrict__ lm = uc + (int64_t)N * NT;  // multipliers
    double* __restrict__ sw = lm + (int64_t)N * NT;  // 1.0 if rows i, i+1 were swapped
    double* __restrict__ x = sw + (int64_t)N * NT;   // solution / rhs
    invit_solve(db + (int64_t)b * N, eb + (int64_t)b * N, wb[(int64_t)b * N + j], N, NT, ua, ub, uc, lm, sw, x, j, b,
                Zb + (int64_t)b * N * N + j);
}

constexpr int IVL_TPB = 16;
constexpr int IVL_MAX_N = 208;  // 6 N 16 doubles + d, e within 160 KB

__global__ __launch_bounds__(IVL_TPB) void invit_lds_kernel(const double* __restrict__ db, const double* __restrict__ eb,
                                                            const double* __restrict__ wb, int N, double* __restrict__ Zb) {
    extern __shared__ __attribute__((aligned(16))) double smem[];
    const int b = blockIdx.y;
    const int t = threadIdx.x;
    const int j = blockIdx.x * IVL_TPB + t;
    double* ds = smem;          // [N] diagonal
    double* es = ds + N;        // [N] off-diagonal
    double* work = es + N;      // six arrays [N][IVL_TPB]
    for (int i = t; i < N; i += IVL_TPB) {
        ds[i] = db[(int64_t)b * N + i];
        es[i] = eb[(int64_t)b * N + i];
    }
    __syncthreads();
    if (j >= N) return;
    const int64_t NT = IVL_TPB;
    double* ua = work + t;
    double* ub = ua + (int64_t)N * NT;
    double* uc = ub + (int64_t)N * NT;
    double* lm = uc + (int64_t)N * NT;
    double* sw = lm + (int64_t)N * NT;
    double* x = sw + (int64_t)N * NT;
    invit_solve(ds, es, wb[(int64_t)b * N + j], N, NT, ua, ub, uc, lm, sw, x, j, b, Zb + (int64_t)b * N * N + j);
}

static bool invit_lds_launch(nbx_ctx* ctx, int N, int64_t batch, const double* d, const double* e, const double* w,
                             double* Z) {
    if (N > IVL_MAX_N) return false;
    const size_t lds = (size_t)(2 * N + 6 * N * IVL_TPB) * sizeof(double);
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&invit_lds_kernel),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        attr_set = true;
    }
    hipLaunchKernelGGL(invit_lds_kernel, dim3((unsigned)nbx_cdiv(N, IVL_TPB), (unsigned)batch), dim3(IVL_TPB), lds,
                       ctx->stream, d, e, w, N, Z);
    return true;
}

// ---------------------------------------------------------------- K4: V = Q Z (apply the reflectors)
// One workgroup per tile of BT_COLS columns of Z; thread (tg, j) = (row group, column).  For each
// reflector: dot_j = sum_t v[t] Z[k+1+t][j] (row groups reduced through LDS), then the rank-1
// update.  Rows of the tile are contiguous 128-byte segments; N/16 workgroups run concurrently.
constexpr int BT_COLS = 16, BT_GROUPS = 16, BT_THREADS = BT_COLS * BT_GROUPS;

__global__ __launch_bounds__(BT_THREADS) void backtransform_kernel(const double* __restrict__ Vhb,
                                                                   const double* __restrict__ taub, int N,
                                                                   double* __restrict__ Zb) {
    extern __shared__ __attribute__((aligned(16))) double smem[];
    double* vk = smem;                  // [N]
    double* partial = vk + N;           // [BT_GROUPS][BT_COLS]
    const int b = blockIdx.y;
    const int j = threadIdx.x % BT_COLS, tg = threadIdx.x / BT_COLS;
    const int col = blockIdx.x * BT_COLS + j;
    const double* Vh = Vhb + (int64_t)b * N * N;
    const double* tau = taub + (int64_t)b * N;
    double* Z = Zb + (int64_t)b * N * N;
    const bool live = col < N;
    for (int k = N - 2; k >= 0; --k) {
        const double tk = tau[k];
        if (tk == 0.0) continue;  // uniform
        const int L = N - k - 1;
        for (int t = threadIdx.x; t < L; t += BT_THREADS) vk[t] = Vh[(int64_t)k * N + t];
        __syncthreads();
        double dot = 0.0;
        if (live)
            for (int t = tg; t < L; t += BT_GROUPS) dot = fma(vk[t], Z[(int64_t)(k + 1 + t) * N + col], dot);
        partial[tg * BT_COLS + j] = dot;
        __syncthreads();
        double tot = 0.0;
#pragma unroll
        for (int gg = 0; gg < BT_GROUPS; ++gg) tot += partial[gg * BT_COLS + j];
        tot *= tk;
        if (live)
            for (int t = tg; t < L; t += BT_GROUPS) Z[(int64_t)(k + 1 + t) * N + col] -= tot * vk[t];
        __syncthreads();
    }
}

// K4 for N <= 208: the same tile of Z in registers -- thread (tg, j) keeps the rows tg + 16 r of its column -- so that
// a reflector costs two barriers and LDS traffic only (backtransform_kernel: three barriers and a read-modify-write of
// the tile in L2, 2 us per reflector); the next reflector is fetched while the current one is applied.
template <int RB>
__global__ __launch_bounds__(BT_THREADS) void backtransform_reg_kernel(const double* __restrict__ Vhb,
                                                                       const double* __restrict__ taub, int N,
                                                                       double* __restrict__ Zb) {
    constexpr int LD = RB * BT_GROUPS;
    __shared__ __attribute__((aligned(16))) double vk[2][LD];              // reflector, indexed by the row it acts on
    __shared__ __attribute__((aligned(16))) double partial[BT_GROUPS * BT_COLS];
    const int b = blockIdx.y;
    const int j = threadIdx.x % BT_COLS, tg = threadIdx.x / BT_COLS;
    const int col = blockIdx.x * BT_COLS + j;
    const double* Vh = Vhb + (int64_t)b * N * N;
    const double* tau = taub + (int64_t)b * N;
    double* Z = Zb + (int64_t)b * N * N;
    const bool live = col < N;
    double z[RB];
#pragma unroll
    for (int r = 0; r < RB; ++r) {
        const int i = tg + r * BT_GROUPS;
        z[r] = (live && i < N) ? Z[(int64_t)i * N + col] : 0.0;
    }
    // this thread's element of a reflector: row threadIdx.x (LD <= 256 = BT_THREADS)
    const int row = threadIdx.x;
    auto fetch = [&](int k) -> double {  // v_k at row `row` (0 outside rows k+1 .. N-1)
        return (k >= 0 && row > k && row < N) ? Vh[(int64_t)k * N + (row - k - 1)] : 0.0;
    };
    int cur = 0;
    if (row < LD) vk[0][row] = fetch(N - 2);
    double tk_next = N > 1 ? tau[N - 2] : 0.0;
    __syncthreads();
    for (int k = N - 2; k >= 0; --k) {
        const double tk = tk_next;
        const double nxt = fetch(k - 1);  // (both loads are consumed only after this reflector's arithmetic)
        tk_next = k > 0 ? tau[k - 1] : 0.0;
        if (tk != 0.0) {  // uniform
            double dot = 0.0;
#pragma unroll
            for (int r = 0; r < RB; ++r) dot = fma(vk[cur][tg + r * BT_GROUPS], z[r], dot);  // (zero outside the rows)
            partial[tg * BT_COLS + j] = dot;
        }
        if (row < LD) vk[cur ^ 1][row] = nxt;
        __syncthreads();
        if (tk != 0.0) {
            double tot = 0.0;
#pragma unroll
            for (int gg = 0; gg < BT_GROUPS; ++gg) tot += partial[gg * BT_COLS + j];
            tot *= tk;
#pragma unroll
            for (int r = 0; r < RB; ++r) z[r] = fma(-tot, vk[cur][tg + r * BT_GROUPS], z[r]);
        }
        __syncthreads();  // partial[] and vk[cur] are rewritten by the next two steps
        cur ^= 1;
    }
#pragma unroll
    for (int r = 0; r < RB; ++r) {
        const int i = tg + r * BT_GROUPS;
        if (live && i < N) Z[(int64_t)i * N + col] = z[r];
    }
}

static bool backtransform_reg_launch(nbx_ctx* ctx, int N, int64_t batch, const double* Vh, const double* tau,
                                     double* Z) {
    const dim3 grid((unsigned)nbx_cdiv(N, BT_COLS), (unsigned)batch);
#define NBX_BTR_GO(RB_) \
    hipLaunchKernelGGL(backtransform_reg_kernel<RB_>, grid, dim3(BT_THREADS), 0, ctx->stream, Vh, tau, N, Z)
    if (N <= 64) NBX_BTR_GO(4);
    else if (N <= 128) NBX_BTR_GO(8);
    else if (N <= 160) NBX_BTR_GO(10);
    else if (N <= 208) NBX_BTR_GO(13);
    else return false;
#undef NBX_BTR_GO
    return true;
}

// ---------------------------------------------------------------- K5: CGS2 orthonormalisation, vectors = rows of Vt
__global__ __launch_bounds__(TD_THREADS) void cgs2_kernel(double* __restrict__ Vtb, int N) {
    extern __shared__ __attribute__((aligned(16))) double smem[];
    double* vj = smem;        // [N] current vector
    double* cf = vj + N;      // [N] projection coefficients
    double* red = cf + N;     // [20]
    double* Vt = Vtb + (int64_t)blockIdx.x * N * N;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    constexpr int NW = TD_THREADS / 64;
    for (int j = 0; j < N; ++j) {
        for (int attempt = 0; attempt < 4; ++attempt) {
            for (int t = tid; t < N; t += TD_THREADS)
                vj[t] = attempt == 0 ? Vt[(int64_t)j * N + t] : (t == (j + attempt - 1) % N ? 1.0 : 0.0);
            __syncthreads();
            for (int pass = 0; pass < 2; ++pass) {
                // c_i = <v_i, v_j> for i < j : one wavefront per i
                for (int i = wave; i < j; i += NW) {
                    double acc = 0.0;
                    for (int t = lane; t < N; t += 64) acc = fma(Vt[(int64_t)i * N + t], vj[t], acc);
                    acc = nbx_wave_sum(acc);
                    if (lane == 0) cf[i] = acc;
                }
                __syncthreads();
                // v_j -= sum_i c_i v_i : one thread per component (coalesced over i rows)
                for (int t = tid; t < N; t += TD_THREADS) {
                    double acc = vj[t];
                    for (int i = 0; i < j; ++i) acc = fma(-cf[i], Vt[(int64_t)i * N + t], acc);
                    vj[t] = acc;
                }
                __syncthreads();
            }
            double nn = 0.0;
            for (int t = tid; t < N; t += TD_THREADS) nn = fma(vj[t], vj[t], nn);
            nn = nbx_block_sum(nn, red);
            if (nn > 1.0e-16 || attempt == 3) {  // input rows have unit norm: below 1e-8 means dependent
                const double s = nn > 0.0 ? 1.0 / sqrt(nn) : 0.0;
                for (int t = tid; t < N; t += TD_THREADS) Vt[(int64_t)j * N + t] = vj[t] * s;
                __syncthreads();
                break;
            }
            __syncthreads();
        }
    }
}

// G <- 1.5 I - 0.5 G   (Newton-Schulz factor: V <- V (3 I - V^T V) / 2)
__global__ void ns_factor_kernel(double* __restrict__ G, int N, int64_t total) {
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= total) return;
    const int64_t r = idx % ((int64_t)N * N);
    const int i = (int)(r / N), j = (int)(r - (int64_t)i * N);
    G[idx] = (i == j ? 1.5 : 0.0) - 0.5 * G[idx];
}

// NaN-propagating max (fmax drops a NaN operand)
__device__ __forceinline__ double td_nanmax(double a, double b) { return (a != a || b != b) ? a + b : fmax(a, b); }

// Per (N,N) matrix: max_{i != j} |M_ij| and max_i |M_ii - DIAG_SHIFT| -> out[b*2 + {0,1}]; a NaN anywhere comes out
// as NaN.  One 1024-thread workgroup per matrix, a wave per row (no index division, 30 independent loads per
// thread at N = 148: 6 us where the element-per-thread loop of 256 threads took 33).
//   DIAG_SHIFT = 1: the defect of a Gram matrix;  0: the off-diagonal weight of R = V^T A V against its diagonal
template <int DIAG_SHIFT>
__global__ __launch_bounds__(1024) void offdiag_diag_max_kernel(const double* __restrict__ M, int N,
                                                                double* __restrict__ out) {
    __shared__ double red_o[16], red_d[16];
    const int b = blockIdx.x;
    const double* m = M + (int64_t)b * N * N;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    double mo = 0.0, md = 0.0;
    // (gridDim.y > 1: the rows are dealt to gridDim.y workgroups, whose partial maxima offdiag_max_combine_kernel joins
    //  -- a matrix of 2000 rows took 1.45 ms on one workgroup)
    for (int i = wave + 16 * blockIdx.y; i < N; i += 16 * gridDim.y) {
        const double* row = m + (int64_t)i * N;
        for (int j = lane; j < N; j += 64) {
            const double x = row[j];
            if (i == j) md = td_nanmax(md, fabs(x - (double)DIAG_SHIFT));
            else mo = td_nanmax(mo, fabs(x));
        }
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        mo = td_nanmax(mo, __shfl_xor(mo, off, 64));
        md = td_nanmax(md, __shfl_xor(md, off, 64));
    }
    if (lane == 0) {
        red_o[wave] = mo;
        red_d[wave] = md;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        double o = red_o[0], d = red_d[0];
#pragma unroll
        for (int w = 1; w < 16; ++w) {
            o = td_nanmax(o, red_o[w]);
            d = td_nanmax(d, red_d[w]);
        }
        out[2 * (b * gridDim.y + blockIdx.y)] = o;
        out[2 * (b * gridDim.y + blockIdx.y) + 1] = d;
    }
}

__global__ void offdiag_max_combine_kernel(const double* __restrict__ part, int nblk, double* __restrict__ out) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= (int)gridDim.x * (int)blockDim.x) return;
    double o = part[2 * b * nblk], d = part[2 * b * nblk + 1];
    for (int k = 1; k < nblk; ++k) {
        o = td_nanmax(o, part[2 * (b * nblk + k)]);
        d = td_nanmax(d, part[2 * (b * nblk + k) + 1]);
    }
    out[2 * b] = o;
    out[2 * b + 1] = d;
}

// max |off-diagonal| and max |diagonal - DIAG_SHIFT| of `batch` matrices into out[2 b + {0, 1}]
template <int DIAG_SHIFT>
static int offdiag_diag_max(nbx_ctx* ctx, const double* M, int N, int64_t batch, double* out) {
    const int nblk = N > 512 ? 32 : 1;
    double* part = ctx->d_scratch + NBX_SCRATCH_DOUBLES / 2;  // (the callers' results live in the first half)
    if (nblk == 1 || 2 * nblk * batch > NBX_SCRATCH_DOUBLES / 2) {
        hipLaunchKernelGGL(offdiag_diag_max_kernel<DIAG_SHIFT>, dim3((unsigned)batch), dim3(1024), 0, ctx->stream, M, N, out);
        NBX_LAUNCH_CHECK();
        return NBX_OK;
    }
    hipLaunchKernelGGL(offdiag_diag_max_kernel<DIAG_SHIFT>, dim3((unsigned)batch, (unsigned)nblk), dim3(1024), 0, ctx->stream, M, N, part);
    NBX_LAUNCH_CHECK();
    hipLaunchKernelGGL(offdiag_max_combine_kernel, dim3(1), dim3((unsigned)batch), 0, ctx->stream, part, nblk, out);
    NBX_LAUNCH_CHECK();
    return NBX_OK;
}

struct TdLayout {
    size_t w_off, d_off, e_off, tau_off, vh_off, z_off, zt_off, scr_off, flag_off, total;
};
TdLayout layout(int64_t n, int64_t batch) {
    TdLayout L;
    size_t off = 0;
    const size_t mat = align256((size_t)(batch * n * n) * sizeof(double));
    const size_t vec = align256((size_t)(batch * n) * sizeof(double));
    L.w_off = off; off += mat;
    L.d_off = off; off += vec;
    L.e_off = off; off += vec;
    L.tau_off = off; off += vec;
    L.vh_off = off; off += mat;
    L.z_off = off; off += mat;
    L.zt_off = off; off += mat;
    L.scr_off = off; off += align256((size_t)(batch * 6 * n * n) * sizeof(double));
    L.flag_off = off; off += 256;  // the whole-chip reduction's status word (scr is reused by the stages after it)
    L.total = off;
    return L;
}

}  // namespace

size_t nbx_eigh_tridiag_worksize(int64_t n, int64_t batch) { return layout(n, batch).total; }

// Householder reduction, eigenvalues, inverse iteration, back-transformation, and the Gram matrix W = Z^T Z of
// the vectors with its defect (max |off-diagonal|, max |diagonal - 1| per matrix) in ctx->d_scratch: everything
// of the tridiagonal route that needs no decision.
static int td_pipeline(nbx_ctx* ctx, int64_t n, int64_t batch, const double* d_a, double* d_w, const TdLayout& L,
                       char* base) {
    double* W = reinterpret_cast<double*>(base + L.w_off);
    double* d = reinterpret_cast<double*>(base + L.d_off);
    double* e = reinterpret_cast<double*>(base + L.e_off);
    double* tau = reinterpret_cast<double*>(base + L.tau_off);
    double* Vh = reinterpret_cast<double*>(base + L.vh_off);
    double* Z = reinterpret_cast<double*>(base + L.z_off);
    double* scr = reinterpret_cast<double*>(base + L.scr_off);
    const int N = (int)n;
    static const bool in_memory = getenv("NBX_TRIDIAG_IN_MEMORY") != nullptr;  // A/B switch
    // N > 198 (beyond one workgroup's registers): the reduction on the whole chip, the matrix in the LDS of up to 256
    // workgroups, and the back-transformation in compact-WY blocks on the GEMM (eigh_grid.hip; NBX_TRIDIAG_GRID=0: off)
    const bool grid = !in_memory && N > 198 && nbx_tdg_covers(n) &&
                      nbx_tdg_work_doubles(n, batch) <= (size_t)(batch * 6 * n * n);
    if (grid) {
        const int rc = nbx_tdg_tridiag(ctx, n, batch, d_a, d, e, tau, Vh, scr, reinterpret_cast<int*>(base + L.flag_off));
        if (rc != NBX_OK) return rc;
    } else if (in_memory || !tridiag_reg_launch(ctx, N, batch, d_a, d, e, tau, Vh))
        hipLaunchKernelGGL(tridiag_kernel, dim3((unsigned)batch), dim3(TD_THREADS),
                           (size_t)(3 * N + TD_THREADS + 20) * sizeof(double), ctx->stream, d_a, N, W, d, e, tau, Vh);
    NBX_LAUNCH_CHECK();
    const int bw = N > 512 ? 8 : 1;  // wavefronts (eigenvalues) per workgroup
    hipLaunchKernelGGL(bisect_kernel, dim3((unsigned)nbx_cdiv(N, bw), (unsigned)batch), dim3(64 * bw),
                       (size_t)(2 * N) * sizeof(double), ctx->stream, d, e, N, d_w);
    NBX_LAUNCH_CHECK();
    if (in_memory || !invit_lds_launch(ctx, N, batch, d, e, d_w, Z))
        hipLaunchKernelGGL(invit_kernel, dim3((unsigned)nbx_cdiv(N, 256), (unsigned)batch), dim3(256), 0, ctx->stream, d,
                           e, d_w, N, scr, Z);
    NBX_LAUNCH_CHECK();
    if (grid) {
        const int rc = nbx_tdg_backtransform(ctx, n, batch, Vh, tau, Z, scr);
        if (rc != NBX_OK) return rc;
    } else if (in_memory || !backtransform_reg_launch(ctx, N, batch, Vh, tau, Z))
        hipLaunchKernelGGL(backtransform_kernel, dim3((unsigned)nbx_cdiv(N, BT_COLS), (unsigned)batch), dim3(BT_THREADS),
                           (size_t)(N + BT_GROUPS * BT_COLS) * sizeof(double), ctx->stream, Vh, tau, N, Z);
    NBX_LAUNCH_CHECK();
    int rc = nbx_gemm(ctx, 'T', 'N', n, n, n, 1.0, Z, n, n * n, Z, n, n * n, 0.0, W, n, n * n, batch);
    if (rc != NBX_OK) return rc;
    NBX_CHECK_ARG(2 * batch <= NBX_SCRATCH_DOUBLES);
    return offdiag_diag_max<1>(ctx, W, N, batch, ctx->d_scratch);
}

static int td_check(int64_t n, int64_t batch, const void* d_work, size_t work_bytes, const TdLayout& L) {
    if (d_work == nullptr || work_bytes < L.total) {
        nbx_set_error("nbx_eigh(tridiagonal): workspace %zu < %zu bytes", work_bytes, L.total);
        return NBX_E_NOMEM;
    }
    if (n > TD_MAXSEG * TD_THREADS) {
        nbx_set_error("nbx_eigh(tridiagonal): N=%lld > %d unsupported", (long long)n, TD_MAXSEG * TD_THREADS);
        return NBX_E_UNSUPPORTED;
    }
    (void)batch;
    return NBX_OK;
}

// (tdg: NULL, or the word the whole-chip reduction sets when a hand-over between its workgroups never arrived -- the
//  tridiagonal matrix is then garbage, and orthonormal eigenvectors of garbage would pass the Gram test)
__global__ void approx_status_kernel(const double* __restrict__ defect, int batch, int* __restrict__ status,
                                     const int* __restrict__ tdg) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    const bool reduced = tdg == nullptr || *tdg == 0;
    if (b < batch) status[b] = (reduced && defect[2 * b] <= 1.0e-6 && defect[2 * b + 1] <= 1.0e-6) ? 1 : -1;  // NaN: -1
}

// include/nbx.h: eigenpairs to inverse-iteration accuracy with NOTHING read back -- a start for a warm solver
extern "C" size_t nbx_eigh_approx_worksize(int64_t n, int64_t batch) {
    return (n <= 0 || batch <= 0) ? 0 : layout(n, batch).total;
}

extern "C" int nbx_eigh_approx(nbx_ctx* ctx, int64_t n, int64_t batch, const double* d_a, double* d_w, double* d_v,
                               void* d_work, size_t work_bytes, int* d_status) {
    NBX_CHECK_ARG(ctx && d_a && d_w && d_v && d_status && n > 0 && batch > 0 && batch <= 1024);
    const TdLayout L = layout(n, batch);
    int rc = td_check(n, batch, d_work, work_bytes, L);
    if (rc != NBX_OK) return rc;
    NBX_CHECK_ARG((reinterpret_cast<uintptr_t>(d_work) & 15) == 0);
    char* base = static_cast<char*>(d_work);
    double* W = reinterpret_cast<double*>(base + L.w_off);
    double* Z = reinterpret_cast<double*>(base + L.z_off);
    nbx_prof_scope prof(ctx, NBX_PROF_EIGH);
    rc = td_pipeline(ctx, n, batch, d_a, d_w, L, base);
    if (rc != NBX_OK) return rc;
    const bool grid = getenv("NBX_TRIDIAG_IN_MEMORY") == nullptr && n > 198 && nbx_tdg_covers(n) &&
                      nbx_tdg_work_doubles(n, batch) <= (size_t)(batch * 6 * n * n);  // (td_pipeline's own test)
    const int* tdg = grid ? reinterpret_cast<const int*>(base + L.flag_off) : nullptr;
    hipLaunchKernelGGL(approx_status_kernel, dim3((unsigned)nbx_cdiv(batch, 64)), dim3(64), 0, ctx->stream,
                       ctx->d_scratch, (int)batch, d_status, tdg);
    NBX_LAUNCH_CHECK();
    // one Newton-Schulz step V = Z (3I - Z^T Z)/2: squares a defect below 1e-6 (status 1); vectors that
    // clusters left nearly dependent stay so, and status says it (-1)
    const int64_t total = batch * n * n;
    hipLaunchKernelGGL(ns_factor_kernel, dim3((unsigned)nbx_cdiv(total, 256)), dim3(256), 0, ctx->stream, W, (int)n, total);
    NBX_LAUNCH_CHECK();
    return nbx_gemm(ctx, 'N', 'N', n, n, n, 1.0, Z, n, n * n, W, n, n * n, 0.0, d_v, n, n * n, batch);
}

// status[b] = 1 when the Gram defect of the inverse-iteration vectors was small enough for one Newton-Schulz
// step (<= 1e-6) AND the off-diagonal weight of V^T A V relative to its diagonal is <= 1e-13 (the host route's
// acceptance test, nbx_eigh_tridiag + nbx_eigh_warm_ex); 0 otherwise (also NaN)
__global__ void quality_status_kernel(const double* __restrict__ gram, const double* __restrict__ rq, int batch,
                                      int* __restrict__ status, int* __restrict__ skip) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= batch) return;
    const bool ortho = gram[2 * b] <= 1.0e-6 && gram[2 * b + 1] <= 1.0e-6;
    const double o = rq[2 * b], d = rq[2 * b + 1];
    const double q = d > 0.0 ? o / d : o;
    status[b] = skip[b] = (ortho && q <= 1.0e-13) ? 1 : 0;
}

// The tridiagonal route with its verdict left ON THE DEVICE (nbx_eigh_warm_ex's cold start inside a queued SCF
// cycle must not wait for the host): eigenpairs as nbx_eigh_tridiag's accepted branch writes them (one
// Newton-Schulz step on the inverse-iteration vectors), d_status[b] = d_skip[b] = 1 accepted / 0 not -- the caller
// queues its fallback solver gated on d_skip (d_status is what that solver overwrites with its sweep count).  The Gram-Schmidt branch of the host route is not taken here: vectors that
// clusters left nearly dependent simply count as not accepted.
int nbx_eigh_tridiag_dev(nbx_ctx* ctx, int64_t n, int64_t batch, const double* d_a, double* d_w, double* d_v,
                         void* d_work, size_t work_bytes, int* d_status, int* d_skip) {
    const TdLayout L = layout(n, batch);
    int rc = td_check(n, batch, d_work, work_bytes, L);
    if (rc != NBX_OK) return rc;
    NBX_CHECK_ARG(4 * batch <= NBX_SCRATCH_DOUBLES && d_status && d_skip);
    char* base = static_cast<char*>(d_work);
    double* W = reinterpret_cast<double*>(base + L.w_off);
    double* Z = reinterpret_cast<double*>(base + L.z_off);
    const int N = (int)n;
    nbx_prof_scope prof(ctx, NBX_PROF_EIGH);
    rc = td_pipeline(ctx, n, batch, d_a, d_w, L, base);  // Gram defect in d_scratch[0, 2 batch)
    if (rc != NBX_OK) return rc;
    const int64_t total = batch * n * n;
    hipLaunchKernelGGL(ns_factor_kernel, dim3((unsigned)nbx_cdiv(total, 256)), dim3(256), 0, ctx->stream, W, N, total);
    NBX_LAUNCH_CHECK();
    rc = nbx_gemm(ctx, 'N', 'N', n, n, n, 1.0, Z, n, n * n, W, n, n * n, 0.0, d_v, n, n * n, batch);
    if (rc != NBX_OK) return rc;
    // quality: R = V^T A V  (W and Z are free again)
    rc = nbx_gemm(ctx, 'T', 'N', n, n, n, 1.0, d_v, n, n * n, d_a, n, n * n, 0.0, W, n, n * n, batch);
    if (rc != NBX_OK) return rc;
    rc = nbx_gemm(ctx, 'N', 'N', n, n, n, 1.0, W, n, n * n, d_v, n, n * n, 0.0, Z, n, n * n, batch);
    if (rc != NBX_OK) return rc;
    double* rq = ctx->d_scratch + 2 * batch;
    rc = offdiag_diag_max<0>(ctx, Z, N, batch, rq);
    if (rc != NBX_OK) return rc;
    hipLaunchKernelGGL(quality_status_kernel, dim3((unsigned)nbx_cdiv(batch, 64)), dim3(64), 0, ctx->stream,
                       ctx->d_scratch, rq, (int)batch, d_status, d_skip);
    NBX_LAUNCH_CHECK();
    return NBX_OK;
}

// Approximate eigenpairs by the tridiagonal route: d_w (batch,N) ascending, d_v (batch,N,N) with
// orthonormal columns.  h_quality[b] = max off-diagonal of V^T A V divided by max |diagonal|.
int nbx_eigh_tridiag(nbx_ctx* ctx, int64_t n, int64_t batch, const double* d_a, double* d_w, double* d_v,
                     void* d_work, size_t work_bytes, double* h_quality) {
    const TdLayout L = layout(n, batch);
    int rc = td_check(n, batch, d_work, work_bytes, L);
    if (rc != NBX_OK) return rc;
    char* base = static_cast<char*>(d_work);
    double* W = reinterpret_cast<double*>(base + L.w_off);
    double* Z = reinterpret_cast<double*>(base + L.z_off);
    double* Zt = reinterpret_cast<double*>(base + L.zt_off);
    const int N = (int)n;
    nbx_prof_scope prof(ctx, NBX_PROF_EIGH);
    // Orthonormalise the columns of Z.  Inverse-iteration vectors of well separated eigenvalues
    // are orthogonal to ~1e-10 already: one Newton-Schulz step V <- V (3I - V^T V)/2 on the MFMA
    // GEMM squares that defect.  Only when clusters left nearly dependent vectors (defect > 1e-6)
    // does the sequential Gram-Schmidt kernel run.
    rc = td_pipeline(ctx, n, batch, d_a, d_w, L, base);
    if (rc != NBX_OK) return rc;
    NBX_HIP(hipMemcpyAsync(ctx->h_pinned, ctx->d_scratch, (size_t)(2 * batch) * sizeof(double), hipMemcpyDeviceToHost,
                           ctx->stream));
    NBX_HIP(hipStreamSynchronize(ctx->stream));
    bool nearly_orthonormal = true;
    for (int64_t bb = 0; bb < 2 * batch; ++bb) nearly_orthonormal = nearly_orthonormal && (ctx->h_pinned[bb] <= 1.0e-6);
    if (nearly_orthonormal) {
        const int64_t total = batch * n * n;
        hipLaunchKernelGGL(ns_factor_kernel, dim3((unsigned)nbx_cdiv(total, 256)), dim3(256), 0, ctx->stream, W, N, total);
        NBX_LAUNCH_CHECK();
        rc = nbx_gemm(ctx, 'N', 'N', n, n, n, 1.0, Z, n, n * n, W, n, n * n, 0.0, d_v, n, n * n, batch);
        if (rc != NBX_OK) return rc;
    } else {
        rc = nbx_transpose(ctx, n, n, batch, Z, Zt);
        if (rc != NBX_OK) return rc;
        hipLaunchKernelGGL(cgs2_kernel, dim3((unsigned)batch), dim3(TD_THREADS), (size_t)(2 * N + 20) * sizeof(double),
                           ctx->stream, Zt, N);
        NBX_LAUNCH_CHECK();
        rc = nbx_transpose(ctx, n, n, batch, Zt, d_v);
        if (rc != NBX_OK) return rc;
    }
    // quality: R = V^T A V  (W and Z are free again)
    rc = nbx_gemm(ctx, 'T', 'N', n, n, n, 1.0, d_v, n, n * n, d_a, n, n * n, 0.0, W, n, n * n, batch);
    if (rc != NBX_OK) return rc;
    rc = nbx_gemm(ctx, 'N', 'N', n, n, n, 1.0, W, n, n * n, d_v, n, n * n, 0.0, Z, n, n * n, batch);
    if (rc != NBX_OK) return rc;
    NBX_CHECK_ARG(2 * batch <= NBX_SCRATCH_DOUBLES);
    rc = offdiag_diag_max<0>(ctx, Z, N, batch, ctx->d_scratch);
    if (rc != NBX_OK) return rc;
    NBX_HIP(hipMemcpyAsync(ctx->h_pinned, ctx->d_scratch, (size_t)(2 * batch) * sizeof(double), hipMemcpyDeviceToHost,
                           ctx->stream));
    NBX_HIP(hipStreamSynchronize(ctx->stream));
    for (int64_t b = 0; b < batch; ++b) {
        const double dmax = ctx->h_pinned[2 * b + 1];
        h_quality[b] = dmax > 0.0 ? ctx->h_pinned[2 * b] / dmax : ctx->h_pinned[2 * b];
    }
    return NBX_OK;
}
