// libnbx: the Householder tridiagonalisation and the back-transformation of the symmetric eigensolver on the WHOLE
// chip, for the sizes beyond one workgroup's registers (N > 198; scipy.linalg.eigh / fractional_matrix_power of
// nbed/scf/huzinaga_scf.py:128,145,168 at the scaling sizes and at N_AO = 2000).
//
// eigh_tridiag.hip's tridiag_kernel runs a matrix on ONE workgroup with the matrix in global memory: 2.05 s at
// N = 2000, a chain of 2000 steps of twelve barriers and two passes over the trailing block in L2.  Here
//   * tdg_kernel: the matrix lives in the LDS of up to 256 workgroups for the whole reduction -- workgroup g holds the
//     full rows i = g (mod P) (N = 2000: eight rows, 128 KB) -- and a Householder step costs ONE grid-wide barrier:
//     a workgroup needs of the others only the raw product y = A22 v of the step before (its own rows' entries it
//     makes itself: the rows are complete) and the NEXT pivot row as it was before that step's update, both handed over
//     through two small vectors in global memory; from them every wave of every workgroup forms w, the updated pivot
//     row, the new reflector and its tau redundantly and in the same order of operations (so all agree bit for bit),
//     then updates its rows with the previous reflector and takes their products with the new one in one pass over LDS;
//   * the back-transformation V = Q Z as compact-WY blocks of 64 reflectors on the MFMA GEMM (Q_b = I - Y T Y^T:
//     three products per block) instead of one reflector at a time on column tiles (62 ms -> a few ms at N = 2000).
// Hand-off discipline (MI355X: private L2 per XCD, L1 never refreshed by other CUs' stores): every handed-over double
// is stored and loaded with agent-scope relaxed atomics (global_store/load ... sc1), every storing wave drains its
// stores, the workgroup meets at its barrier, ONE lane adds to the arrival counter and polls it; the other waves load
// behind the workgroup barrier that lane then joins.  The launch is cooperative (all workgroups resident, one per CU)
// and the poll gives up after ~seconds (status word set, every workgroup still reaches the end of the kernel).
// Conventions of d, e, tau and the reflectors are LAPACK dsytd2's (UPLO = 'L'), as in eigh_tridiag.hip.
#include "nbx_common.h"

namespace {

constexpr int TDG_THREADS = 256;
constexpr unsigned TDG_SPIN_LIMIT = 1u << 22;

__device__ __forceinline__ double tdg_ld(const double* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void tdg_st(double* p, double v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

// all workgroups of the grid have passed here `target / gridDim.x` times
__device__ __forceinline__ void tdg_grid_sync(unsigned* counter, unsigned target, int* status) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) {
        __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        unsigned spins = 0;
        while (__hip_atomic_load(counter, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
            __builtin_amdgcn_s_sleep(1);
            if (++spins > TDG_SPIN_LIMIT) {  // (a workgroup that never came: give up, say so; every workgroup ends)
                __hip_atomic_store(status, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                break;
            }
        }
    }
    __syncthreads();
}

// NM: registers per lane and vector (N <= 64 NM); NBT: matrices reduced side by side by one launch.
// a_in (NBT, N, N) lower triangle read; d, e, tau (NBT, N); Vg (NBT, N, N): row k = reflector k in the coordinates of
// the matrix (zeros up to column k, 1 at column k + 1); xch: 4 NBT 64 NM doubles (y and pivot-row vectors, two of each);
// counter: zero at launch.
template <int NM, int NBT>
__global__ __launch_bounds__(TDG_THREADS, 1) void tdg_kernel(const double* __restrict__ a_in, int N, int R, int clog,
                                                             double* __restrict__ dg, double* __restrict__ eg,
                                                             double* __restrict__ taug, double* __restrict__ Vg,
                                                             double* __restrict__ xch, unsigned* __restrict__ counter,
                                                             int* __restrict__ status) {
    constexpr int NP = 64 * NM;  // padded row length
    extern __shared__ __attribute__((aligned(16))) double smem[];
    double* rows = smem;                               // [NBT][R][NP]
    double* mini = smem + (size_t)NBT * R * NP;        // [4 waves][NBT][3][8]: vp, wp, vn at this workgroup's rows
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int g = blockIdx.x, P = gridDim.x;
    const int gl = g & 63, g64 = g >> 6, cmask = (1 << clog) - 1;
    const int64_t n2 = (int64_t)N * N;

    // ---- this workgroup's rows into LDS (full rows from the lower triangle), zero padded
    for (int b = 0; b < NBT; ++b)
        for (int q = 0; q < R; ++q) {
            const int i = g + P * q;
            for (int k = tid; k < NP; k += TDG_THREADS) {
                double v = 0.0;
                if (i < N && k < N) v = (i >= k) ? a_in[b * n2 + (int64_t)i * N + k] : a_in[b * n2 + (int64_t)k * N + i];
                rows[((size_t)b * R + q) * NP + k] = v;
            }
        }
    __syncthreads();

    double vp[NBT][NM], wp[NBT][NM], vn[NBT][NM];
    double taup[NBT];
#pragma unroll
    for (int b = 0; b < NBT; ++b) {
        taup[b] = 0.0;
#pragma unroll
        for (int m = 0; m < NM; ++m) vp[b][m] = wp[b][m] = vn[b][m] = 0.0;
    }

    for (int j = 0; j < N - 1; ++j) {
        double* ybuf_w = xch + (size_t)(j & 1) * 2 * NBT * NP;          // this phase's products and next pivot row
        const double* ybuf_r = xch + (size_t)((j + 1) & 1) * 2 * NBT * NP;  // the phase before's
#pragma unroll
        for (int b = 0; b < NBT; ++b) {
            // ---- the vectors, by every wave: w of the step before, the pivot row j, the reflector of this step
            double row[NM];
            if (j > 0) {
                double yv[NM];
#pragma unroll
                for (int m = 0; m < NM; ++m) {
                    const int k = lane + 64 * m;
                    const bool in = k >= j && k < N;
                    yv[m] = in ? tdg_ld(ybuf_r + (size_t)b * NP + k) : 0.0;
                    row[m] = in ? tdg_ld(ybuf_r + (size_t)(NBT + b) * NP + k) : 0.0;
                }
                double dot = 0.0, yj = 0.0;
#pragma unroll
                for (int m = 0; m < NM; ++m) {
                    const int k = lane + 64 * m;
                    yv[m] *= taup[b];  // p = tau A22 v
                    dot = fma(yv[m], vp[b][m], dot);
                    yj += (k == j) ? yv[m] : 0.0;
                }
                dot = nbx_wave_sum(dot);
                yj = nbx_wave_sum(yj);
                const double alpha2 = -0.5 * taup[b] * dot;
                const double wpj = yj + alpha2;  // (v of the step before is 1 at index j)
#pragma unroll
                for (int m = 0; m < NM; ++m) {
                    wp[b][m] = fma(alpha2, vp[b][m], yv[m]);
                    row[m] = row[m] - wp[b][m] - wpj * vp[b][m];
                }
            } else {
#pragma unroll
                for (int m = 0; m < NM; ++m) {
                    const int k = lane + 64 * m;
                    row[m] = k < N ? a_in[b * n2 + (int64_t)k * N] : 0.0;  // row 0 = column 0 of the lower triangle
                    wp[b][m] = 0.0;
                }
            }
            double dj = 0.0, alpha = 0.0, ss = 0.0;
#pragma unroll
            for (int m = 0; m < NM; ++m) {
                const int k = lane + 64 * m;
                dj += (k == j) ? row[m] : 0.0;
                alpha += (k == j + 1) ? row[m] : 0.0;
                ss = (k > j + 1) ? fma(row[m], row[m], ss) : ss;
            }
            dj = nbx_wave_sum(dj);
            alpha = nbx_wave_sum(alpha);
            ss = nbx_wave_sum(ss);
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
#pragma unroll
            for (int m = 0; m < NM; ++m) {
                const int k = lane + 64 * m;
                vn[b][m] = (k == j + 1) ? 1.0 : ((k > j + 1 && k < N) ? row[m] * scale : 0.0);
            }
            if (g == 0 && tid == 0) {
                dg[(size_t)b * N + j] = dj;
                eg[(size_t)b * N + j] = beta;
                taug[(size_t)b * N + j] = tk;
            }
            if (g == j % P && wave == 0) {
#pragma unroll
                for (int m = 0; m < NM; ++m) {
                    const int k = lane + 64 * m;
                    if (k < N) Vg[b * n2 + (int64_t)j * N + k] = vn[b][m];
                }
            }
            // the entries at this workgroup's rows (i = g + P q: lane g % 64, register g / 64 + (P / 64) q), for the row pass
            double* mn = mini + ((size_t)(wave * NBT + b) * 3) * 8;
#pragma unroll
            for (int m = 0; m < NM; ++m) {
                if (lane == gl && m >= g64 && ((m - g64) & cmask) == 0) {
                    const int q = (m - g64) >> clog;
                    if (q < 8) {
                        mn[q] = vp[b][m];
                        mn[8 + q] = wp[b][m];
                        mn[16 + q] = vn[b][m];
                    }
                }
            }
            // ---- this wave's rows: update with the reflector of the step before, product with the new one
            for (int q = wave; q < R; q += 4) {
                const int i = g + P * q;
                if (i <= j || i >= N) continue;  // (uniform) the row has left the trailing block
                const double vpi = mn[q], wpi = mn[8 + q];
                double* rw = rows + ((size_t)b * R + q) * NP;
                double* pub = ybuf_w + (size_t)(NBT + b) * NP;
                const bool publish = i == j + 1;  // the next pivot row, as it is before this step's update
                double ysum = 0.0;
#pragma unroll
                for (int m = 0; m < NM; ++m) {
                    const int k = lane + 64 * m;
                    double a = rw[k];
                    a -= fma(vpi, wp[b][m], wpi * vp[b][m]);
                    rw[k] = a;
                    ysum = fma(a, vn[b][m], ysum);
                    if (publish && k < N) tdg_st(pub + k, a);
                }
                ysum = nbx_wave_sum(ysum);
                if (lane == 0) tdg_st(ybuf_w + (size_t)b * NP + i, ysum);
            }
#pragma unroll
            for (int m = 0; m < NM; ++m) vp[b][m] = vn[b][m];
            taup[b] = tk;
        }
        tdg_grid_sync(counter, (unsigned)(j + 1) * (unsigned)P, status);
    }
    // the last diagonal element: its row has every update (the reflector of the last step has tau = 0)
    {
        const int i = N - 1;
        if (g == i % P && tid == 0) {
            for (int b = 0; b < NBT; ++b) {
                dg[(size_t)b * N + i] = rows[((size_t)b * R + i / P) * NP + i];
                eg[(size_t)b * N + i] = 0.0;
                taug[(size_t)b * N + i] = 0.0;
            }
        }
        if (g == 0 && wave == 0)
            for (int b = 0; b < NBT; ++b)
                for (int k = lane; k < N; k += 64) Vg[b * n2 + (int64_t)i * N + k] = 0.0;
    }
}

// T of the compact-WY form of a block of nb reflectors (LAPACK dlarft, forward, columnwise): T[i][i] = tau_i,
// T[0:i, i] = -tau_i T[0:i, 0:i] (Y^T Y)[0:i, i].  G = Y^T Y (nb x nb, ld nb); one workgroup per (block, matrix).
__global__ __launch_bounds__(64) void wy_tfactor_kernel(const double* __restrict__ G, const double* __restrict__ tau, int nb,
                                                        int nbk, double* __restrict__ T) {
    __shared__ double t[64][65];
    __shared__ double z[64];
    const int i0 = threadIdx.x;
    for (int c = 0; c < nb; ++c) t[i0][c] = 0.0;
    __syncthreads();
    for (int i = 0; i < nbk; ++i) {
        const double ti = tau[i];
        // z = T[0:i, 0:i] G[0:i, i]  (upper triangular T: row r uses columns r .. i - 1)
        double s = 0.0;
        if (i0 < i)
            for (int c = i0; c < i; ++c) s = fma(t[i0][c], G[c * nb + i], s);
        z[i0] = s;
        __syncthreads();
        if (i0 < i) t[i0][i] = -ti * z[i0];
        if (i0 == i) t[i][i] = ti;
        __syncthreads();
    }
    for (int c = 0; c < nb; ++c) T[i0 * nb + c] = (i0 < nb) ? t[i0][c] : 0.0;
}

}  // namespace

// doubles of workspace the two entry points below need beside the caller's matrices
size_t nbx_tdg_work_doubles(int64_t n, int64_t batch) {
    const int64_t np = (n + 63) / 64 * 64;
    return (size_t)(4 * batch * np + 64 /* counter, status */ + batch * (2 * 64 * 64 + 2 * 64 * n));
}

bool nbx_tdg_covers(int64_t n) {
    static const bool on = getenv("NBX_TRIDIAG_GRID") == nullptr || atoi(getenv("NBX_TRIDIAG_GRID")) != 0;
    return on && n > 64 && n <= 2048;
}

// Householder reduction of `batch` symmetric matrices (lower triangles of d_a): d, e, tau (batch, n), Vg (batch, n, n)
// with row k = reflector k in matrix coordinates.  work: nbx_tdg_work_doubles(n, batch) doubles.
int nbx_tdg_tridiag(nbx_ctx* ctx, int64_t n, int64_t batch, const double* d_a, double* d, double* e, double* tau, double* Vg,
                    double* work) {
    const int N = (int)n;
    const int nm = (N + 63) / 64;
    const int NMt = nm <= 8 ? 8 : (nm <= 16 ? 16 : 32);
    const int NP = 64 * NMt;
    int c = nm < 4 ? nm : 4;          // workgroups = 64 c: every one holds at least one row
    if (c == 3) c = 2;                // (a power of two: the row of a register is found by a shift)
    const int clog = c == 4 ? 2 : (c == 2 ? 1 : 0);
    const int P = 64 * c;
    const int R = (N + P - 1) / P;
    NBX_CHECK_ARG(R <= 8 && N <= 2048);
    unsigned* counter = reinterpret_cast<unsigned*>(work + 4 * batch * (size_t)(64 * 32));
    int* status = reinterpret_cast<int*>(counter + 8);
    // matrices side by side in one launch while their vectors fit the registers (N <= 1024), else one after the other
    const int per = (NMt <= 16 && batch >= 2) ? 2 : 1;
    for (int64_t b0 = 0; b0 < batch; b0 += per) {
        const int nbt = (int)((batch - b0) < per ? (batch - b0) : per);
        int rc = nbx_memset(ctx, counter, 0, 64);
        if (rc != NBX_OK) return rc;
        const size_t lds = ((size_t)nbt * R * NP + 4 * nbt * 3 * 8) * sizeof(double);
        const double* a_ = d_a + b0 * n * n;
        double *d_ = d + b0 * n, *e_ = e + b0 * n, *t_ = tau + b0 * n, *v_ = Vg + b0 * n * n, *x_ = work;
        int N_ = N, R_ = R, cl_ = clog;
        void* args[] = {(void*)&a_, (void*)&N_, (void*)&R_, (void*)&cl_, (void*)&d_, (void*)&e_, (void*)&t_, (void*)&v_,
                        (void*)&x_, (void*)&counter, (void*)&status};
        const void* fn = nullptr;
        if (NMt == 8) fn = nbt == 2 ? (const void*)&tdg_kernel<8, 2> : (const void*)&tdg_kernel<8, 1>;
        else if (NMt == 16) fn = nbt == 2 ? (const void*)&tdg_kernel<16, 2> : (const void*)&tdg_kernel<16, 1>;
        else fn = (const void*)&tdg_kernel<32, 1>;
        (void)hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        NBX_CHECK_ARG(lds <= 160 * 1024);
        const hipError_t err = hipLaunchCooperativeKernel(fn, dim3((unsigned)P), dim3(TDG_THREADS), args, lds, ctx->stream);
        if (err != hipSuccess) {
            nbx_set_error("nbx_tdg_tridiag: cooperative launch of %d workgroups failed: %s", P, hipGetErrorString(err));
            return NBX_E_HIP;
        }
    }
    return NBX_OK;
}

// Z (batch, n, n) <- Q Z with Q = H_0 H_1 ... H_{n-2} from Vg / tau of nbx_tdg_tridiag, in compact-WY blocks of 64
// reflectors, last block first.  work: the same workspace (its tail).
int nbx_tdg_backtransform(nbx_ctx* ctx, int64_t n, int64_t batch, const double* Vg, const double* tau, double* Z, double* work) {
    const int64_t nb = 64, n2 = n * n;
    const int64_t np = (n + 63) / 64 * 64;
    double* base = work + 4 * batch * np + 64;
    double* G = base;                       // (batch, nb, nb)
    double* T = G + batch * nb * nb;        // (batch, nb, nb)
    double* W1 = T + batch * nb * nb;       // (batch, nb, n)
    double* W2 = W1 + batch * nb * n;       // (batch, nb, n)
    const int64_t nref = n - 1;
    const int64_t nblk = (nref + nb - 1) / nb;
    for (int64_t blk = nblk - 1; blk >= 0; --blk) {
        const int64_t k0 = blk * nb, nbk = (nref - k0) < nb ? (nref - k0) : nb;
        const double* Y = Vg + k0 * n;  // (nbk, n) rows of reflectors, ld n
        int rc = nbx_gemm(ctx, 'N', 'T', nbk, nbk, n, 1.0, Y, n, n2, Y, n, n2, 0.0, G, nb, nb * nb, batch);
        if (rc != NBX_OK) return rc;
        for (int64_t b = 0; b < batch; ++b) {
            hipLaunchKernelGGL(wy_tfactor_kernel, dim3(1), dim3(64), 0, ctx->stream, G + b * nb * nb, tau + b * n + k0, (int)nb,
                               (int)nbk, T + b * nb * nb);
            NBX_LAUNCH_CHECK();
        }
        rc = nbx_gemm(ctx, 'N', 'N', nbk, n, n, 1.0, Y, n, n2, Z, n, n2, 0.0, W1, n, nb * n, batch);
        if (rc != NBX_OK) return rc;
        rc = nbx_gemm(ctx, 'N', 'N', nbk, n, nbk, 1.0, T, nb, nb * nb, W1, n, nb * n, 0.0, W2, n, nb * n, batch);
        if (rc != NBX_OK) return rc;
        rc = nbx_gemm(ctx, 'T', 'N', n, n, nbk, -1.0, Y, n, n2, W2, n, nb * n, 1.0, Z, n, n2, batch);
        if (rc != NBX_OK) return rc;
    }
    return NBX_OK;
}
