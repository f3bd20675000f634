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

// grid (N, batch), 64 threads: eigenvalue index blockIdx.x (ascending)
__global__ __launch_bounds__(64) void bisect_kernel(const double* __restrict__ db, const double* __restrict__ eb, int N,
                                                    double* __restrict__ wb) {
    extern __shared__ __attribute__((aligned(16))) double smem[];
    double* d = smem;        // [N]
    double* e2 = d + N;      // [N]
    const int b = blockIdx.y, j = blockIdx.x, lane = threadIdx.x;
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
    if (lane == 0) wb[(int64_t)b * N + j] = 0.5 * (lo + hi);
}

// ---------------------------------------------------------------- K3: eigenvectors of T (inverse iteration)
// one thread per eigenvalue; per-thread work arrays with stride NT (threads) for coalescing
__global__ __launch_bounds__(256) void invit_kernel(const double* __restrict__ db, const double* __restrict__ eb,
                                                    const double* __restrict__ wb, int N, double* __restrict__ scratch,
                                                    double* __restrict__ Zb) {
    const int b = blockIdx.y;
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= N) return;
    const double* d = db + (int64_t)b * N;
    const double* e = eb + (int64_t)b * N;
    const double lam = wb[(int64_t)b * N + j];
    const int64_t NT = N;  // stride between consecutive rows of one thread's arrays
    double* base = scratch + (int64_t)b * 6 * N * NT;
    double* ua = base + j;              // U diagonal
    double* ub = ua + (int64_t)N * NT;  // U first superdiagonal
    double* uc = ub + (int64_t)N * NT;  // U second superdiagonal
    double* lm = uc + (int64_t)N * NT;  // multipliers
    double* sw = lm + (int64_t)N * NT;  // 1.0 if rows i, i+1 were swapped
    double* x = sw + (int64_t)N * NT;   // solution / rhs

    double tnorm = 0.0;
    for (int i = 0; i < N; ++i)
        tnorm = fmax(tnorm, fabs(d[i]) + (i > 0 ? fabs(e[i - 1]) : 0.0) + (i < N - 1 ? fabs(e[i]) : 0.0));
    const double eps = 2.220446049250313e-16;
    const double pivtol = fmax(eps * tnorm, 1.0e-300);

    // LU with partial pivoting of T - lam I  (LAPACK dlagtf restated)
    double ai = d[0] - lam;
    double bi = N > 1 ? e[0] : 0.0;  // super-diagonal entry of the current row
    for (int i = 0; i < N - 1; ++i) {
        const double ci = e[i];                        // sub-diagonal entry (row i+1, col i)
        double anext = d[i + 1] - lam;                 // diagonal of row i+1
        const double bnext = i < N - 2 ? e[i + 1] : 0.0;  // super-diagonal of row i+1
        if (fabs(ai) >= fabs(ci) || fabs(ci) < pivtol) {
            double piv = ai;
            if (fabs(piv) < pivtol) piv = copysign(pivtol, piv == 0.0 ? 1.0 : piv);
            const double mult = ci / piv;
            ua[(int64_t)i * NT] = piv;
            ub[(int64_t)i * NT] = bi;
            uc[(int64_t)i * NT] = 0.0;
            lm[(int64_t)i * NT] = mult;
            sw[(int64_t)i * NT] = 0.0;
            ai = anext - mult * bi;
            bi = bnext;
        } else {
            const double mult = ai / ci;
            ua[(int64_t)i * NT] = ci;
            ub[(int64_t)i * NT] = anext;
            uc[(int64_t)i * NT] = bnext;
            lm[(int64_t)i * NT] = mult;
            sw[(int64_t)i * NT] = 1.0;
            ai = bi - mult * anext;
            bi = -mult * bnext;
        }
    }
    if (fabs(ai) < pivtol) ai = copysign(pivtol, ai == 0.0 ? 1.0 : ai);
    ua[(int64_t)(N - 1) * NT] = ai;
    ub[(int64_t)(N - 1) * NT] = 0.0;
    uc[(int64_t)(N - 1) * NT] = 0.0;

    // start vector: deterministic pseudo-random in (-1, 1)
    for (int i = 0; i < N; ++i) x[(int64_t)i * NT] = nbx_synth_val(7, (uint64_t)j * 4096u + (uint64_t)i, 12345u + b);

    for (int it = 0; it < 3; ++it) {
        // forward: apply the row operations of the factorisation to the right-hand side
        for (int i = 0; i < N - 1; ++i) {
            double yi = x[(int64_t)i * NT], yn = x[(int64_t)(i + 1) * NT];
            if (sw[(int64_t)i * NT] != 0.0) {
                const double t = yi;
                yi = yn;
                yn = t;
            }
            yn -= lm[(int64_t)i * NT] * yi;
            x[(int64_t)i * NT] = yi;
            x[(int64_t)(i + 1) * NT] = yn;
        }
        // backward substitution with the three diagonals of U
        double x1 = 0.0, x2 = 0.0, amax = 0.0;
        for (int i = N - 1; i >= 0; --i) {
            const double t = x[(int64_t)i * NT] - ub[(int64_t)i * NT] * x1 - uc[(int64_t)i * NT] * x2;
            const double xi = t / ua[(int64_t)i * NT];
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
    double* Z = Zb + (int64_t)b * N * N;
    for (int i = 0; i < N; ++i) Z[(int64_t)i * N + j] = x[(int64_t)i * NT] * s;
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

// max_{i != j} |G_ij| and max_i |G_ii - 1| -> out[b*2 + {0,1}]; a NaN anywhere in G comes out as NaN
__device__ __forceinline__ double td_nanmax(double a, double b) { return (a != a || b != b) ? a + b : fmax(a, b); }
__global__ __launch_bounds__(256) void gram_defect_kernel(const double* __restrict__ G, int N, double* __restrict__ out) {
    __shared__ double red_o[4], red_d[4];
    const int b = blockIdx.x;
    const double* r = G + (int64_t)b * N * N;
    double mo = 0.0, md = 0.0;
    for (int64_t idx = threadIdx.x; idx < (int64_t)N * N; idx += 256) {
        const int i = (int)(idx / N), j = (int)(idx - (int64_t)i * N);
        if (i == j) md = td_nanmax(md, fabs(r[idx] - 1.0));
        else mo = td_nanmax(mo, fabs(r[idx]));
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        mo = td_nanmax(mo, __shfl_xor(mo, off, 64));
        md = td_nanmax(md, __shfl_xor(md, off, 64));
    }
    if ((threadIdx.x & 63) == 0) {
        red_o[threadIdx.x >> 6] = mo;
        red_d[threadIdx.x >> 6] = md;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        out[2 * b] = td_nanmax(td_nanmax(red_o[0], red_o[1]), td_nanmax(red_o[2], red_o[3]));
        out[2 * b + 1] = td_nanmax(td_nanmax(red_d[0], red_d[1]), td_nanmax(red_d[2], red_d[3]));
    }
}

// max_{i != j} |R_ij| and max_i |R_ii| of each (N,N) matrix -> out[b*2 + {0,1}]
__global__ __launch_bounds__(256) void offdiag_max_kernel(const double* __restrict__ R, int N, double* __restrict__ out) {
    __shared__ double red_o[4], red_d[4];
    const int b = blockIdx.x;
    const double* r = R + (int64_t)b * N * N;
    double mo = 0.0, md = 0.0;
    for (int64_t idx = threadIdx.x; idx < (int64_t)N * N; idx += 256) {
        const int i = (int)(idx / N), j = (int)(idx - (int64_t)i * N);
        const double a = fabs(r[idx]);
        if (i == j) md = fmax(md, a);
        else mo = fmax(mo, a);
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        mo = fmax(mo, __shfl_xor(mo, off, 64));
        md = fmax(md, __shfl_xor(md, off, 64));
    }
    if ((threadIdx.x & 63) == 0) {
        red_o[threadIdx.x >> 6] = mo;
        red_d[threadIdx.x >> 6] = md;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        out[2 * b] = fmax(fmax(red_o[0], red_o[1]), fmax(red_o[2], red_o[3]));
        out[2 * b + 1] = fmax(fmax(red_d[0], red_d[1]), fmax(red_d[2], red_d[3]));
    }
}

struct TdLayout {
    size_t w_off, d_off, e_off, tau_off, vh_off, z_off, zt_off, scr_off, total;
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
    hipLaunchKernelGGL(tridiag_kernel, dim3((unsigned)batch), dim3(TD_THREADS),
                       (size_t)(3 * N + TD_THREADS + 20) * sizeof(double), ctx->stream, d_a, N, W, d, e, tau, Vh);
    NBX_LAUNCH_CHECK();
    hipLaunchKernelGGL(bisect_kernel, dim3((unsigned)N, (unsigned)batch), dim3(64), (size_t)(2 * N) * sizeof(double),
                       ctx->stream, d, e, N, d_w);
    NBX_LAUNCH_CHECK();
    hipLaunchKernelGGL(invit_kernel, dim3((unsigned)nbx_cdiv(N, 256), (unsigned)batch), dim3(256), 0, ctx->stream, d, e,
                       d_w, N, scr, Z);
    NBX_LAUNCH_CHECK();
    hipLaunchKernelGGL(backtransform_kernel, dim3((unsigned)nbx_cdiv(N, BT_COLS), (unsigned)batch), dim3(BT_THREADS),
                       (size_t)(N + BT_GROUPS * BT_COLS) * sizeof(double), ctx->stream, Vh, tau, N, Z);
    NBX_LAUNCH_CHECK();
    int rc = nbx_gemm(ctx, 'T', 'N', n, n, n, 1.0, Z, n, n * n, Z, n, n * n, 0.0, W, n, n * n, batch);
    if (rc != NBX_OK) return rc;
    NBX_CHECK_ARG(2 * batch <= NBX_SCRATCH_DOUBLES);
    hipLaunchKernelGGL(gram_defect_kernel, dim3((unsigned)batch), dim3(256), 0, ctx->stream, W, N, ctx->d_scratch);
    NBX_LAUNCH_CHECK();
    return NBX_OK;
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

__global__ void approx_status_kernel(const double* __restrict__ defect, int batch, int* __restrict__ status) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b < batch) status[b] = (defect[2 * b] <= 1.0e-6 && defect[2 * b + 1] <= 1.0e-6) ? 1 : -1;  // NaN: -1
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
    hipLaunchKernelGGL(approx_status_kernel, dim3((unsigned)nbx_cdiv(batch, 64)), dim3(64), 0, ctx->stream,
                       ctx->d_scratch, (int)batch, d_status);
    NBX_LAUNCH_CHECK();
    // one Newton-Schulz step V = Z (3I - Z^T Z)/2: squares a defect below 1e-6 (status 1); vectors that
    // clusters left nearly dependent stay so, and status says it (-1)
    const int64_t total = batch * n * n;
    hipLaunchKernelGGL(ns_factor_kernel, dim3((unsigned)nbx_cdiv(total, 256)), dim3(256), 0, ctx->stream, W, (int)n, total);
    NBX_LAUNCH_CHECK();
    return nbx_gemm(ctx, 'N', 'N', n, n, n, 1.0, Z, n, n * n, W, n, n * n, 0.0, d_v, n, n * n, batch);
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
    hipLaunchKernelGGL(offdiag_max_kernel, dim3((unsigned)batch), dim3(256), 0, ctx->stream, Z, N, ctx->d_scratch);
    NBX_LAUNCH_CHECK();
    NBX_HIP(hipMemcpyAsync(ctx->h_pinned, ctx->d_scratch, (size_t)(2 * batch) * sizeof(double), hipMemcpyDeviceToHost,
                           ctx->stream));
    NBX_HIP(hipStreamSynchronize(ctx->stream));
    for (int64_t b = 0; b < batch; ++b) {
        const double dmax = ctx->h_pinned[2 * b + 1];
        h_quality[b] = dmax > 0.0 ? ctx->h_pinned[2 * b] / dmax : ctx->h_pinned[2 * b];
    }
    return NBX_OK;
}
