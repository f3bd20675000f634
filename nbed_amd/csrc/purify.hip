// libnbx: the occupied-space projector of a symmetric matrix WITHOUT diagonalising it
// (include/nbx.h "density by purification").
//
// An SCF cycle needs of its eigenproblem only the density D = C_occ C_occ^T (nbed/scf/huzinaga_scf.py:166-174:
// eigh, aufbau occupation, make_rdm1); the eigenpairs themselves matter once, at the end.  While the Fock
// matrix still moves a lot from cycle to cycle the warm-started solvers of eigh*.hip are no better than a cold
// one -- 7-9 Jacobi sweeps, 2.5-3 ms at N = 148 on a real molecule -- whereas the
// projector follows from GEMMs alone: Niklasson's trace-correcting purification (SP2, Phys. Rev. B 66, 155115),
//
//     P_0 = (l_max I - F) / (l_max - l_min)            (Gershgorin bounds: spectrum mapped into [0, 1], reversed)
//     P_{k+1} = P_k^2  or  2 P_k - P_k^2               whichever brings tr P nearer the number of occupied levels
//
// converges quadratically to the projector on the n_occ LOWEST eigenvectors -- the aufbau density -- whenever
// a gap separates level n_occ from level n_occ + 1; 30-50 steps of one (N x N) product each for a spectrum 25 Ha
// wide with a 0.02-0.9 Ha gap, agreement with the eigenvector projector 1e-14 (measured on the Fock matrices of
// octane / 6-31G*).  Everything is decided on the device: every workgroup of the update kernel derives the
// step's traces from the diagonals (the same sums in the same order everywhere), the launches are queued
// unconditionally and gated on a per-matrix word, the verdict is a status word the host reads one cycle late.
#include "nbx_common.h"

namespace {

constexpr int PUR_THREADS = 256;
constexpr int PUR_ALL = 49;  // MFMA steps (4 k each) of a purification step whose operands are requested at once: N <= 196
constexpr int PUR_MAX_ITER = 72;  // (also the stride of the trace log)
constexpr double PUR_IDEM = 2.0e-12;  // tr(P - P^2) of the step before below this (a step squares the error on one
                                      // side of the gap and doubles it on the other: the result is good to ~2x this)

// One workgroup per matrix: Gershgorin bounds, then P0; status[b] = 0 (-2: no usable bounds).
__global__ __launch_bounds__(PUR_THREADS) void pur_init_kernel(int N, const double* __restrict__ F, double* __restrict__ P,
                                                               int* __restrict__ status) {
    __shared__ double lo[PUR_THREADS], hi[PUR_THREADS];
    const int b = blockIdx.x;
    const int64_t n2 = (int64_t)N * N;
    F += b * n2;
    P += b * n2;
    double mn = 1.0e300, mx = -1.0e300;
    for (int i = threadIdx.x; i < N; i += PUR_THREADS) {
        // the COLUMN discs (Gershgorin holds for them as for the rows, and F is symmetric to rounding anyway):
        // consecutive threads read consecutive addresses, where the row sums were a stride-N walk per thread
        double r = 0.0;
        const double d = F[(int64_t)i * N + i];
        // (sixteen loads in flight: unrolled by four this loop was the 59 us of the kernel -- 148 trips to L2 in 37 rounds)
#pragma unroll 16
        for (int j = 0; j < N; ++j) r += (j == i) ? 0.0 : fabs(F[(int64_t)j * N + i]);
        mn = fmin(mn, d - r);
        mx = fmax(mx, d + r);
    }
    lo[threadIdx.x] = mn;
    hi[threadIdx.x] = mx;
    __syncthreads();
    for (int o = PUR_THREADS / 2; o > 0; o >>= 1) {
        if (threadIdx.x < o) {
            lo[threadIdx.x] = fmin(lo[threadIdx.x], lo[threadIdx.x + o]);
            hi[threadIdx.x] = fmax(hi[threadIdx.x], hi[threadIdx.x + o]);
        }
        __syncthreads();
    }
    const double lmin = lo[0], lmax = hi[0];
    const double inv = 1.0 / (lmax - lmin);
    // (gridDim.y workgroups per matrix: each finds the same bounds -- that is eight rounds of loads -- and writes every
    //  gridDim.y-th group of four rows of P0; with one workgroup the 37 rounds of this loop were 50 us of the kernel)
    for (int i = (threadIdx.x >> 6) + (PUR_THREADS >> 6) * blockIdx.y; i < N; i += (PUR_THREADS >> 6) * gridDim.y)
        for (int j = threadIdx.x & 63; j < N; j += 64) {  // a wave per row: no index division
            const int64_t e = (int64_t)i * N + j;
            P[e] = ((i == j ? lmax : 0.0) - F[e]) * inv;
        }
    if (threadIdx.x == 0 && blockIdx.y == 0) status[b] = (lmax > lmin && isfinite(inv)) ? 0 : -2;
}

typedef double pur_v4 __attribute__((ext_vector_type(4)));

// One SP2 step, ONE launch: grid (tiles, tiles, batch), one wave per 16 x 16 tile of the result.
//   * every wave derives tr P_k from the diagonal of `cur` (the same sum in the same order everywhere) and reads
//     tr P_{k-1} from `traces` (left there by the launch before): |tr P_k - tr P_{k-1}| = tr(P - P^2) of step k-1;
//   * converged (below PUR_IDEM with the right trace): P_k already is the step that squares the error away -- it is
//     copied to the other buffer (the caller reads the first whatever the parity) and the status word takes k + 1;
//     launches after that see a status that is neither 0 nor their own k + 1 and return at once;
//   * otherwise out = P_k^2 if tr P_k > n_occ (the trace must come down) else 2 P_k - P_k^2, on the matrix cores
//     (v_mfma_f64_16x16x4_f64, operands straight from the L2-resident matrix, as gemm_small_kernel does).
__global__ __launch_bounds__(64) void pur_step_kernel(int N, const double* __restrict__ cur, double* __restrict__ out,
                                                      int nocc_a, int nocc_b, int iter, int max_iter,
                                                      double* __restrict__ traces, int* __restrict__ status) {
    const int b = blockIdx.z;
    const int st = status[b];
    if (st != 0 && st != iter + 1) return;
    const int64_t n2 = (int64_t)N * N;
    cur += b * n2;
    out += b * n2;
    const int lane = threadIdx.x, fr = lane & 15, fk = lane >> 4;
    double t = 0.0;
    for (int i = lane; i < N; i += 64) t += cur[(int64_t)i * N + i];
    // N <= 196: the operands of the whole product are requested NOW, behind the diagonal (loads return in order: the
    // trace below does not wait for them) -- a step is then two memory latencies (status word, everything else)
    // where the 16-wide trips of the general loop below were ten
    const int i0 = blockIdx.y * 16, j0 = blockIdx.x * 16;
    const int col = j0 + fr;
    const int row_a = i0 + fr;
    const bool a_ok = row_a < N, b_ok = col < N;
    const double* ap = cur + (int64_t)(a_ok ? row_a : 0) * N;  // A[row][k]: consecutive k
    const double* bp = cur + (b_ok ? col : 0);                 // B[k][col]: stride N
    const bool all_at_once = N <= 4 * PUR_ALL;
    double av[PUR_ALL], bv[PUR_ALL];
    if (all_at_once) {
#pragma unroll
        for (int j = 0; j < PUR_ALL; ++j) {
            const int k = 4 * j + fk;
            const bool in = k < N;
            av[j] = (in && a_ok) ? ap[k] : 0.0;
            bv[j] = (in && b_ok) ? bp[(int64_t)k * N] : 0.0;
        }
    }
    t = nbx_wave_sum(t);
    const double nocc = (double)(b == 0 ? nocc_a : nocc_b);
    const bool first_tile = blockIdx.x == 0 && blockIdx.y == 0;
    const bool finite = isfinite(t);
    bool done = false;
    if (iter > 0) {
        const double tp = traces[b * PUR_MAX_ITER + iter - 1];
        done = finite && fabs(t - tp) < PUR_IDEM && fabs(t - nocc) < 1.0e-6;
    }
    if (first_tile && lane == 0) {
        traces[b * PUR_MAX_ITER + iter] = t;
        if (done) status[b] = iter + 1;
        else if (!finite || iter == max_iter - 1) status[b] = -1;
    }
    if (done) {
        if (col < N)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = i0 + fk + 4 * r;
                if (row < N) out[(int64_t)row * N + col] = cur[(int64_t)row * N + col];
            }
        return;
    }
    const bool square = t > nocc;
    pur_v4 acc = (pur_v4){0.0, 0.0, 0.0, 0.0};
    if (all_at_once) {  // (steps past N multiply zeros)
#pragma unroll
        for (int j = 0; j < PUR_ALL; ++j)
            if (4 * j < N) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(av[j], bv[j], acc, 0, 0, 0);
    }
    const int kfull = all_at_once ? 0 : (N & ~15);
    for (int k0 = 0; k0 < kfull; k0 += 16) {
        double av[4], bv[4];  // (shadow the single-trip arrays)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            av[j] = ap[k0 + 4 * j + fk];
            bv[j] = bp[(int64_t)(k0 + 4 * j + fk) * N];
        }
#pragma unroll
        for (int j = 0; j < 4; ++j)
            acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a_ok ? av[j] : 0.0, b_ok ? bv[j] : 0.0, acc, 0, 0, 0);
    }
    if (!all_at_once && kfull < N) {
        double av[4], bv[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int k = kfull + 4 * j + fk;
            const bool in = k < N;
            av[j] = (in && a_ok) ? ap[k] : 0.0;
            bv[j] = (in && b_ok) ? bp[(int64_t)k * N] : 0.0;
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(av[j], bv[j], acc, 0, 0, 0);
    }
    if (b_ok)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int row = i0 + fk + 4 * r;
            if (row < N) {
                const int64_t e = (int64_t)row * N + col;
                out[e] = square ? acc[r] : 2.0 * cur[e] - acc[r];
            }
        }
}

}  // namespace

extern "C" size_t nbx_purify_worksize(int64_t n, int64_t batch) {
    if (n <= 0 || batch <= 0 || batch > 64) return 0;
    return (size_t)(n * n * batch) * sizeof(double) + (size_t)(batch * PUR_MAX_ITER) * sizeof(double) + 256;
}

// d_f: (batch, n, n) symmetric matrices (an orthonormal basis: plain eigenproblem); d_p: out, the projector on
// the nocc lowest eigenvectors of each; d_status[b]: steps taken (> 0) or < 0 (no gap found / not finite).
// matrix 0 takes nocc_a, every other matrix nocc_b: batches of more than two need nocc_a == nocc_b (refused
// otherwise).  max_iter <= 0: the limit (72).
extern "C" int nbx_purify(nbx_ctx* ctx, int64_t n, int64_t batch, const double* d_f, int64_t nocc_a, int64_t nocc_b, double* d_p,
               void* d_work, size_t work_bytes, int max_iter, int* d_status) {
    NBX_CHECK_ARG(ctx && d_f && d_p && d_work && d_status && n > 0 && batch > 0 && batch <= 64);
    NBX_CHECK_ARG(nocc_a >= 0 && nocc_a <= n && nocc_b >= 0 && nocc_b <= n);
    NBX_CHECK_ARG(batch <= 2 || nocc_a == nocc_b);
    if (work_bytes < nbx_purify_worksize(n, batch)) {
        nbx_set_error("nbx_purify: workspace %zu < %zu bytes", work_bytes, nbx_purify_worksize(n, batch));
        return NBX_E_NOMEM;
    }
    if (max_iter <= 0 || max_iter > PUR_MAX_ITER) max_iter = PUR_MAX_ITER;
    const int64_t n2 = n * n;
    double* pb = static_cast<double*>(d_work);
    double* traces = pb + n2 * batch;
    hipLaunchKernelGGL(pur_init_kernel, dim3((unsigned)batch, (unsigned)(n >= 64 ? 8 : 1)), dim3(PUR_THREADS), 0, ctx->stream,
                       (int)n, d_f, d_p, d_status);
    NBX_LAUNCH_CHECK();
    const unsigned tiles = (unsigned)nbx_cdiv(n, 16);
    for (int it = 0; it < max_iter; ++it) {
        const double* cur = (it & 1) ? pb : d_p;
        double* out = (it & 1) ? d_p : pb;
        hipLaunchKernelGGL(pur_step_kernel, dim3(tiles, tiles, (unsigned)batch), dim3(64), 0, ctx->stream, (int)n, cur, out,
                           (int)nocc_a, (int)(batch > 1 ? nocc_b : nocc_a), it, max_iter, traces, d_status);
        NBX_LAUNCH_CHECK();
    }
    return NBX_OK;
}
