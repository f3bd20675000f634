// libnbx: Fock assembly, Huzinaga symmetrisation, trace/energy reductions, DIIS vector
// algebra, transposes and the spin-orbital scatter (include/nbx.h).
// All of these are O(N^2) (or O(n^4) for the scatter) and HBM/L2-bound: coalesced
// accesses, transposed operands staged through a padded LDS tile, wavefront
// reductions for the scalar outputs.
#include "nbx_common.h"
#include "jk_m4_layout.h"
#include "jk_s4_layout.h"

namespace {

constexpr int TILE = 32;

// ---------------------------------------------------------------- reductions
// Stage 2 of every scalar reduction: out[o] = sum_b partial[b * nout + o] (fixed order).
// Outputs o >= sqrt_from are stored as sqrt(sum) (Frobenius norms); `tail_n` ints from `tail` are
// appended as doubles (status words that travel to the host in the same copy).
__global__ void final_reduce_kernel(const double* __restrict__ partial, int nblocks, int nout,
                                    double* __restrict__ out, int sqrt_from, const int* __restrict__ tail = nullptr,
                                    int tail_n = 0) {
    __shared__ double red[17];
    if ((int)threadIdx.x < tail_n) out[nout + threadIdx.x] = (double)tail[threadIdx.x];
    for (int o = 0; o < nout; ++o) {
        double t = 0.0;
        for (int b = threadIdx.x; b < nblocks; b += blockDim.x) t += partial[(int64_t)b * nout + o];
        t = nbx_block_sum(t, red);
        if (threadIdx.x == 0) out[o] = (o >= sqrt_from) ? sqrt(t) : t;
        __syncthreads();
    }
}

// Load the TILE x TILE tile of M (N x N, row-major) whose top-left corner is (r0, c0)
// into lds[r][c] (zero padded).  blockDim = (TILE, 8).
__device__ __forceinline__ void load_tile(const double* __restrict__ M, int N, int r0, int c0,
                                          double (*lds)[TILE + 1]) {
    for (int r = threadIdx.y; r < TILE; r += blockDim.y) {
        const int gr = r0 + r, gc = c0 + threadIdx.x;
        lds[r][threadIdx.x] = (gr < N && gc < N) ? M[(int64_t)gr * N + gc] : 0.0;
    }
}

// out[b] partial: sum_ij A[i,j] B[j,i] over one tile pair.
__global__ void trace_prod_kernel(const double* __restrict__ A, const double* __restrict__ B, int N,
                                  double* __restrict__ partial) {
    __shared__ double bt[TILE][TILE + 1];
    __shared__ double red[17];
    const int b = blockIdx.z;
    const int64_t n2 = (int64_t)N * N;
    A += b * n2;
    B += b * n2;
    const int i0 = blockIdx.y * TILE, j0 = blockIdx.x * TILE;
    load_tile(B, N, j0, i0, bt);  // bt[j][i] = B[j0+j][i0+i]
    __syncthreads();
    double acc = 0.0;
    for (int r = threadIdx.y; r < TILE; r += blockDim.y) {
        const int gi = i0 + r, gj = j0 + threadIdx.x;
        if (gi < N && gj < N) acc = fma(A[(int64_t)gi * N + gj], bt[threadIdx.x][r], acc);
    }
    // flatten the 2-D block for the block reduction
    const int tid = threadIdx.y * blockDim.x + threadIdx.x;
    double v = nbx_wave_sum(acc);
    if ((tid & 63) == 0) red[tid >> 6] = v;
    __syncthreads();
    if (tid == 0) {
        double t = 0.0;
        for (int w = 0; w < (int)(blockDim.x * blockDim.y) / 64; ++w) t += red[w];
        const int nblk = gridDim.x * gridDim.y;
        partial[(int64_t)b * nblk + blockIdx.y * gridDim.x + blockIdx.x] = t;
    }
}

// F[x] = h + v[x] + J - K[x] ; vhf[x] = J - K[x]
__global__ void fock_uhf_kernel(const double* __restrict__ h, int h3d, const double* __restrict__ vemb,
                                const double* __restrict__ jk, double* __restrict__ fock,
                                double* __restrict__ vhf, int64_t n2) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n2) return;
    const double j = jk[i];
#pragma unroll
    for (int x = 0; x < 2; ++x) {
        const double v = j - jk[(1 + x) * n2 + i];
        double f = h[h3d ? x * n2 + i : i] + v;
        if (vemb) f += vemb[x * n2 + i];
        fock[x * n2 + i] = f;
        if (vhf) vhf[x * n2 + i] = v;
    }
}

// Hz = -kappa (F DS + (F DS)^T) and F' = F + Hz in ONE launch (the product and its symmetrisation
// were a GEMM and an element-wise kernel: one launch less on the SCF's critical path).  One
// wavefront per pair of 16 x 16 tiles (I, J <= I): T1 = F[I,:] DS[:,J] and T2 = F[J,:] DS[:,I] on the
// matrix cores (k-slot order of gemm_small_kernel), Hz[I,J] = -kappa (T1 + T2^T) and its mirror image
// through a 16 x 17 LDS tile.  Out of place: every workgroup reads all of F.
typedef double ew_v4f64 __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(64) void huzinaga_fused_kernel(const double* __restrict__ F,
                                                            const double* __restrict__ DS, int N, double kappa,
                                                            double* __restrict__ hz, double* __restrict__ fock_out) {
    __shared__ double t2[16][17], hv[16][17];
    const int ti = blockIdx.y, tj = blockIdx.x;
    if (tj > ti) return;
    const int64_t n2 = (int64_t)N * N;
    F += blockIdx.z * n2;
    DS += blockIdx.z * n2;
    hz += blockIdx.z * n2;
    if (fock_out != nullptr) fock_out += blockIdx.z * n2;
    const int lane = threadIdx.x, fr = lane & 15, fk = lane >> 4;
    const int ri = ti * 16 + fr, rj = tj * 16 + fr;  // A-fragment rows of the two products = B-fragment columns
    const bool oki = ri < N, okj = rj < N;
    const double* fi = F + (int64_t)(oki ? ri : 0) * N;  // row ri of F (k contiguous)
    const double* fj = F + (int64_t)(okj ? rj : 0) * N;
    const double* dsj = DS + (okj ? rj : 0);  // column rj of DS (k strided by N)
    const double* dsi = DS + (oki ? ri : 0);
    ew_v4f64 acc1 = {0.0, 0.0, 0.0, 0.0}, acc2 = {0.0, 0.0, 0.0, 0.0};
    const bool diag = ti == tj;
    // The operands of HZ_STEPS MFMA steps (4 k each) are requested in ONE trip to L2 -- N = 148 is two trips -- where a
    // 16-wide k loop waited out a memory latency per trip: ten of them, the whole 11.6 us this kernel took on the SCF's
    // critical path (the anti-pattern gemm_small_kernel was written to avoid).  One wavefront per workgroup: the
    // 4 x HZ_STEPS operand registers are there.  Same k order, same MFMAs: bitwise the same result.
    constexpr int HZ_STEPS = 20;
    for (int k0 = 0; k0 < N; k0 += 4 * HZ_STEPS) {
        double a1[HZ_STEPS], b1[HZ_STEPS], a2[HZ_STEPS], b2[HZ_STEPS];
#pragma unroll
        for (int j = 0; j < HZ_STEPS; ++j) {
            const int k = k0 + 4 * j + fk;
            const bool in = k < N;
            a1[j] = (in && oki) ? fi[k] : 0.0;
            b1[j] = (in && okj) ? dsj[(int64_t)k * N] : 0.0;
            a2[j] = (in && okj && !diag) ? fj[k] : 0.0;
            b2[j] = (in && oki && !diag) ? dsi[(int64_t)k * N] : 0.0;
        }
#pragma unroll
        for (int j = 0; j < HZ_STEPS; ++j) {
            if (k0 + 4 * j < N) {  // uniform (steps past N would multiply zeros; the old loop ran to the next multiple of 16)
                acc1 = __builtin_amdgcn_mfma_f64_16x16x4f64(a1[j], b1[j], acc1, 0, 0, 0);
                if (!diag) acc2 = __builtin_amdgcn_mfma_f64_16x16x4f64(a2[j], b2[j], acc2, 0, 0, 0);
            }
        }
    }
    if (diag) acc2 = acc1;
    // accumulator element r of a lane: tile row fk + 4 r, tile column fr
#pragma unroll
    for (int r = 0; r < 4; ++r) t2[fk + 4 * r][fr] = acc2[r];
    __syncthreads();
    double val[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        val[r] = -kappa * (acc1[r] + t2[fr][fk + 4 * r]);  // T1[i][j] + T2[j][i]
        hv[fk + 4 * r][fr] = val[r];
    }
    __syncthreads();
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int row = ti * 16 + fk + 4 * r, col = tj * 16 + fr;
        if (row < N && col < N) {
            const int64_t o = (int64_t)row * N + col;
            hz[o] = val[r];
            if (fock_out != nullptr) fock_out[o] = F[o] + val[r];
        }
        if (!diag) {  // the mirror tile: Hz[J rows][I columns], element [j][i] = value of [i][j]
            const int mrow = tj * 16 + fk + 4 * r, mcol = ti * 16 + fr;
            if (mrow < N && mcol < N) {
                const int64_t o = (int64_t)mrow * N + mcol;
                const double v = hv[fr][fk + 4 * r];
                hz[o] = v;
                if (fock_out != nullptr) fock_out[o] = F[o] + v;
            }
        }
    }
}

// Hz = -kappa (FDS + FDS^T); optionally F += Hz
__global__ void huzinaga_sym_kernel(const double* __restrict__ fds, int N, double kappa,
                                    double* __restrict__ hz, double* __restrict__ fock) {
    __shared__ double t[TILE][TILE + 1];
    const int b = blockIdx.z;
    const int64_t n2 = (int64_t)N * N;
    fds += b * n2;
    hz += b * n2;
    if (fock) fock += b * n2;
    const int i0 = blockIdx.y * TILE, j0 = blockIdx.x * TILE;
    load_tile(fds, N, j0, i0, t);  // t[j][i] = FDS[j0+j][i0+i]
    __syncthreads();
    for (int r = threadIdx.y; r < TILE; r += blockDim.y) {
        const int gi = i0 + r, gj = j0 + threadIdx.x;
        if (gi < N && gj < N) {
            const int64_t o = (int64_t)gi * N + gj;
            const double v = -kappa * (fds[o] + t[threadIdx.x][r]);
            hz[o] = v;
            if (fock) fock[o] += v;
        }
    }
}

// out = A^T - A per batch entry: the CDIIS error vector F D S - S D F from A = S D F (pyscf.scf.diis.CDIIS
// behind scf.hf.kernel, nbed/driver.py:533) in one launch
__global__ void antisym_kernel(const double* __restrict__ a, int N, double* __restrict__ out) {
    __shared__ double t[TILE][TILE + 1];
    const int b = blockIdx.z;
    const int64_t n2 = (int64_t)N * N;
    a += b * n2;
    out += b * n2;
    const int i0 = blockIdx.y * TILE, j0 = blockIdx.x * TILE;
    load_tile(a, N, j0, i0, t);  // t[j][i] = A[j0+j][i0+i]
    __syncthreads();
    for (int r = threadIdx.y; r < TILE; r += blockDim.y) {
        const int gi = i0 + r, gj = j0 + threadIdx.x;
        if (gi < N && gj < N) {
            const int64_t o = (int64_t)gi * N + gj;
            out[o] = t[threadIdx.x][r] - a[o];
        }
    }
}

// partial[blk*4 + {0,1}] = sum (h + v + 0.5 vhf + hz)[x][i,j] * D[x][j,i];  [2,3] = sum (D-Dold)^2
// With `out`: the workgroup that arrives last (device-scope counter, left at zero) also does the
// second stage in the fixed order of final_reduce_kernel -- out[0,1] the sums, out[2,3] the square
// roots, then `tail_n` status words as doubles, then 1.0 (stored last, system scope) -- so the cycle's
// scalars cost one launch.  `out` may be pinned host memory: the values then need no copy.
// DENS (T = 16, 128 threads): the density is not read but MADE here -- D[x] = C[x][:, :nocc_x] C[x][:, :nocc_x]^T, the
// tile of spin x on the matrix cores by wavefront x with the k-slot order of gemm_small_kernel (gemm.hip: bitwise the
// density nbx_gemm would have written), stored to `dm_w` and used from LDS; D is symmetric to the bit (the products of
// (i, j) and (j, i) are the same numbers added in the same order), so the transposed tile the energy needs is this one.
// One launch instead of the density product + this kernel on the SCF's critical path.
template <int T, bool DENS = false>  // tile edge: 16 x 16 tiles give N = 148 a hundred workgroups instead of 25
__global__ void huz_scalars_kernel(const double* __restrict__ h, int h3d, const double* __restrict__ vemb,
                                   const double* __restrict__ vhf, const double* __restrict__ hz,
                                   const double* __restrict__ dm, const double* __restrict__ dm_old, int N,
                                   double* __restrict__ partial, double* __restrict__ out_final = nullptr,
                                   const int* __restrict__ tail = nullptr, int tail_n = 0,
                                   int* __restrict__ counter = nullptr, double* __restrict__ dts = nullptr,
                                   int s4_nb_ = 0, int s4_lpt_ = 0, int dts_m4 = 0,
                                   const double* __restrict__ dtail = nullptr, int dtail_n = 0,
                                   const double* __restrict__ cmo = nullptr, int nocc_a = 0, int nocc_b = 0,
                                   double* __restrict__ dm_w = nullptr) {
    // A kernel on the SCF's critical path: everything it reads is requested in one trip to memory (both spins'
    // tiles at once), the sums travel by lane moves, and the workgroup that arrives last adds the four columns
    // of partials side by side -- three dependent round trips to L2 where there were nine.
    __shared__ double dt[2][T][T + 1];
    __shared__ double red[4][16];
    __shared__ int last;
    const int64_t n2 = (int64_t)N * N;
    const int i0 = blockIdx.y * T, j0 = blockIdx.x * T;
    const int tid = threadIdx.y * blockDim.x + threadIdx.x;
    const int nthr = blockDim.x * blockDim.y, nwave = nthr / 64;
    constexpr int R = T / 8;  // rows of the tile per thread (blockDim.y == 8)
    double out[4] = {0.0, 0.0, 0.0, 0.0};
    // With `dts`: this kernel also leaves Dtot' of D (the density the NEXT J/K build contracts) in
    // the staging order of nbx_jk_packed -- sum_x (D_ab + D_ba) below the diagonal, sum_x D_aa on it
    // -- which saves that build its preparation launch.  Tiles on or below the diagonal write.
    double dsum[R], tsum[R];
    double ham[2][R], dv[2][R], dold[2][R];
    constexpr int DSTEPS = 16;  // MFMA steps of four k: nocc <= 64 (the caller checks)
    double ca[DENS ? DSTEPS : 1], cb[DENS ? DSTEPS : 1];
    if (DENS) {
        // wavefront x: operands of the whole k range of spin x requested in one trip (A: row i0 + fr, B: row j0 + fr of C)
        const int x = tid >> 6, lane = tid & 63, fr = lane & 15, fk = lane >> 4;
        const int K = x == 0 ? nocc_a : nocc_b;
        const int ra = i0 + fr, rb = j0 + fr;
        const double* pa = cmo + x * n2 + (int64_t)(ra < N ? ra : 0) * N;
        const double* pb = cmo + x * n2 + (int64_t)(rb < N ? rb : 0) * N;
#pragma unroll
        for (int j = 0; j < DSTEPS; ++j) {
            const int k = 4 * j + fk;
            ca[j] = (k < K && ra < N) ? pa[k] : 0.0;
            cb[j] = (k < K && rb < N) ? pb[k] : 0.0;
        }
    }
#pragma unroll
    for (int x = 0; x < 2; ++x)
#pragma unroll
        for (int k = 0; k < R; ++k) {
            const int r = threadIdx.y + 8 * k;
            const int gr = j0 + r, gc = i0 + threadIdx.x;  // dt[x][j][i] = D[x][j0+j][i0+i]
            if (!DENS) dt[x][r][threadIdx.x] = (gr < N && gc < N) ? dm[x * n2 + (int64_t)gr * N + gc] : 0.0;
            const int gi = i0 + r, gj = j0 + threadIdx.x;
            const bool in = gi < N && gj < N;
            const int64_t o = in ? (int64_t)gi * N + gj : 0;
            double hm = h[h3d ? x * n2 + o : o] + 0.5 * vhf[x * n2 + o];
            if (hz) hm += hz[x * n2 + o];  // (no Huzinaga operator: the mu-shift cycle of scf_cycle.hip)
            if (vemb) hm += vemb[x * n2 + o];
            ham[x][k] = in ? hm : 0.0;
            if (!DENS) dv[x][k] = in ? dm[x * n2 + o] : 0.0;
            dold[x][k] = in ? dm_old[x * n2 + o] : 0.0;
        }
    if (DENS) {
        const int x = tid >> 6, lane = tid & 63, fr = lane & 15, fk = lane >> 4;
        const int K = x == 0 ? nocc_a : nocc_b;
        ew_v4f64 acc = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int j = 0; j < DSTEPS; ++j)
            if (4 * j < K) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(ca[j], cb[j], acc, 0, 0, 0);  // (uniform)
        // accumulator element r of a lane: tile row fk + 4 r, tile column fr; dt[x][i][j] = D[x][i0+i][j0+j] here
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int row = fk + 4 * r;
            dt[x][row][fr] = acc[r];
            if (i0 + row < N && j0 + fr < N) dm_w[x * n2 + (int64_t)(i0 + row) * N + j0 + fr] = acc[r];
        }
    }
    __syncthreads();
    if (DENS) {
#pragma unroll
        for (int x = 0; x < 2; ++x)
#pragma unroll
            for (int k = 0; k < R; ++k) dv[x][k] = dt[x][threadIdx.y + 8 * k][threadIdx.x];  // (zero outside the matrix)
    }
#pragma unroll
    for (int k = 0; k < R; ++k) dsum[k] = tsum[k] = 0.0;
#pragma unroll
    for (int x = 0; x < 2; ++x)
#pragma unroll
        for (int k = 0; k < R; ++k) {
            const int r = threadIdx.y + 8 * k;
            const double t = DENS ? dv[x][k] : dt[x][threadIdx.x][r];  // D[j][i] (DENS: = D[i][j], bit for bit)
            out[x] = fma(ham[x][k], t, out[x]);
            const double dd = dv[x][k] - dold[x][k];
            out[2 + x] = fma(dd, dd, out[2 + x]);
            dsum[k] += dv[x][k];
            tsum[k] += t;
        }
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const double v = nbx_wave_sum_dpp(out[k]);
        if ((tid & 63) == 0) red[k][tid >> 6] = v;
    }
    __syncthreads();
    if (tid < 4) {
        double t = 0.0;
        for (int w = 0; w < nwave; ++w) t += red[tid][w];
        partial[(int64_t)(blockIdx.y * gridDim.x + blockIdx.x) * 4 + tid] = t;
    }
    // (the Dtot' table: stored AFTER the workgroup has announced its partial sums -- nobody but the next kernel reads
    //  it, and ahead of the fence the stores' completion sat on the path to the last workgroup's final sums)
    auto store_dts = [&]() {
        if (dts != nullptr && i0 >= j0 && dts_m4) {  // (jk_m4.hip's order; s4_nb_, s4_lpt_, dts_m4 = r1, r2, r3 << 8 | LPT of nbx_jk_m4_weight_layout)
    #pragma unroll
            for (int k = 0; k < R; ++k) {
                const int gi = i0 + threadIdx.y + 8 * k, gj = j0 + threadIdx.x;
                if (gi < N && gj <= gi) dts[m4_weight_index_rt(s4_nb_, s4_lpt_, dts_m4 >> 8, dts_m4 & 255, gi, gj)] = gi == gj ? dsum[k] : dsum[k] + tsum[k];
            }
        } else if (dts != nullptr && i0 >= j0) {
            const S4Geom g = s4_geom(N, s4_nb_);
    #pragma unroll
            for (int k = 0; k < R; ++k) {
                const int gi = i0 + threadIdx.y + 8 * k, gj = j0 + threadIdx.x;
                if (gi < N && gj <= gi) dts[s4_dts_index(g, s4_lpt_, gi, gj)] = gi == gj ? dsum[k] : dsum[k] + tsum[k];
            }
        }
    };
    if (out_final == nullptr) {
        store_dts();
        return;
    }
    const int nblocks = gridDim.x * gridDim.y;
    __threadfence();  // the partials above are visible device-wide before the arrival is
    __syncthreads();
    if (tid == 0) {
        const int prev = __hip_atomic_fetch_add(counter, 1, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
        last = prev == nblocks - 1;
        if (last) __hip_atomic_store(counter, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    __syncthreads();
    if (!last) {
        store_dts();
        return;
    }
    __threadfence();
    if (tid < tail_n) out_final[4 + tid] = (double)tail[tid];
    // `dtail`: doubles an earlier kernel of the chain left on the device (the orbital-gradient sums of the mu-shift
    // cycle), carried to the host behind the status words
    if (tid < dtail_n) out_final[4 + tail_n + tid] = dtail[tid];
    double t4[4] = {0.0, 0.0, 0.0, 0.0};
    for (int b = tid; b < nblocks; b += nthr)
#pragma unroll
        for (int o = 0; o < 4; ++o)
            t4[o] += __hip_atomic_load(partial + (int64_t)b * 4 + o, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __syncthreads();  // (red is reused)
#pragma unroll
    for (int o = 0; o < 4; ++o) {
        const double v = nbx_wave_sum_dpp(t4[o]);
        if ((tid & 63) == 0) red[o][tid >> 6] = v;
    }
    __syncthreads();
    if (tid < 4) {
        double tot = 0.0;
        for (int w = 0; w < nwave; ++w) tot += red[tid][w];
        out_final[tid] = (tid >= 2) ? sqrt(tot) : tot;
    }
    __syncthreads();
    // the word behind the results says they are all there: a host that polls it (pinned memory) needs no event
    // on the stream -- an event record between two cycles holds the next cycle's first kernel back by ~10 us
    if (tid == 0) {
        __threadfence_system();
        __hip_atomic_store(out_final + 4 + tail_n + dtail_n, 1.0, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
    store_dts();  // (this workgroup's part of the table, last of all)
}

__global__ void axpby_kernel(int64_t n, double a, const double* __restrict__ x, double b, double* __restrict__ y) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
        y[i] = (b == 0.0) ? a * x[i] : fma(a, x[i], b * y[i]);
}

constexpr int MAX_LINCOMB = 16;
struct Coefs {
    double c[MAX_LINCOMB];
};

__global__ void lincomb_kernel(int64_t n, int nvec, Coefs cf, const double* __restrict__ vecs, int64_t stride,
                               double* __restrict__ out) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        double t = 0.0;
        for (int k = 0; k < nvec; ++k) t = fma(cf.c[k], vecs[k * stride + i], t);
        out[i] = t;
    }
}

// partial[blk * nvec + k] = sum_i x[i] * vecs[k][i]
__global__ void dots_kernel(int64_t n, int nvec, const double* __restrict__ x, const double* __restrict__ vecs,
                            int64_t stride, double* __restrict__ partial) {
    __shared__ double red[17];
    for (int k = 0; k < nvec; ++k) {
        double t = 0.0;
        for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
            t = fma(x[i], vecs[k * stride + i], t);
        t = nbx_block_sum(t, red);
        if (threadIdx.x == 0) partial[(int64_t)blockIdx.x * nvec + k] = t;
        __syncthreads();
    }
}

// ------------------------------------------------------------------ device-resident DIIS
constexpr int DIIS_MAX_SPACE = 16;
constexpr int DIIS_M = DIIS_MAX_SPACE + 1;
constexpr int DIIS_SLOTS = ((DIIS_M + 1) * (DIIS_M + 1) + 63) / 64;  // matrix elements per lane
// Behind the coefficients in d_coef: the 8 x 8 eigenvector basis the Pulay solve of the previous update ended
// with and a word that says it is one (zero-initialised by the caller = none yet), twice: the copy being read
// and the copy being written.
constexpr int DIIS_STATE_HALF = 68;
constexpr int DIIS_STATE_DOUBLES = 2 * DIIS_STATE_HALF;
constexpr long long DIIS_STATE_MAGIC = 0x4e42585f44494953ll;

// xs[slot] = x, es[slot] = e = (err given ? err : x - xprev), partial[blk*nd + k] = sum_i e[i] * es[k][i]
// (anti_n > 0: err points at matrices A of order anti_n and the error vector is A^T - A, formed on the fly -- the CDIIS
//  error F D S - S D F from A = S D F without a launch of its own)
__global__ __launch_bounds__(256) void diis_push_kernel(int64_t n, int nd, int slot, const double* __restrict__ x,
                                                        const double* __restrict__ xprev,
                                                        const double* __restrict__ err, double* __restrict__ xs,
                                                        double* __restrict__ es, double* __restrict__ partial,
                                                        double* __restrict__ state, int anti_n = 0) {
    __shared__ double red[4][DIIS_MAX_SPACE];
    // the solver's basis of the previous update becomes the one this update starts from (see
    // diis_solve8_lincomb_kernel: its workgroups read one copy while workgroup 0 writes the other)
    if (blockIdx.x == 0 && threadIdx.x < DIIS_STATE_HALF) state[threadIdx.x] = state[DIIS_STATE_HALF + threadIdx.x];
    double acc[DIIS_MAX_SPACE];
#pragma unroll
    for (int k = 0; k < DIIS_MAX_SPACE; ++k) acc[k] = 0.0;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const double xv = x[i];
        double e;
        if (anti_n > 0) {
            const int64_t a2 = (int64_t)anti_n * anti_n, bb = i / a2, rc = i - bb * a2;
            const int64_t r = rc / anti_n, c = rc - r * anti_n;
            e = err[bb * a2 + c * anti_n + r] - err[i];
        } else {
            e = (err != nullptr) ? err[i] : xv - xprev[i];
        }
        xs[(int64_t)slot * n + i] = xv;
        es[(int64_t)slot * n + i] = e;
#pragma unroll
        for (int k = 0; k < DIIS_MAX_SPACE; ++k)
            if (k < nd) acc[k] = fma(e, (k == slot) ? e : es[(int64_t)k * n + i], acc[k]);
    }
    // every sum of the block in one pass: the wavefronts' sums by lane moves (no LDS round trips), one barrier,
    // then thread k adds the four of sum k in wave order
#pragma unroll
    for (int k = 0; k < DIIS_MAX_SPACE; ++k) {
        if (k < nd) {  // uniform
            const double t = nbx_wave_sum_dpp(acc[k]);
            if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6][k] = t;
        }
    }
    __syncthreads();
    if ((int)threadIdx.x < nd)
        partial[(int64_t)blockIdx.x * nd + threadIdx.x] =
            ((red[0][threadIdx.x] + red[1][threadIdx.x]) + red[2][threadIdx.x]) + red[3][threadIdx.x];
}

// Parallel-order Jacobi on a matrix of order <= 8 held ONE ELEMENT PER LANE (lane = 8 r + c: a =
// A[r][c], v = V[r][c]; rows / columns >= m are a decoupled zero pad): the rotation parameters and
// the partner elements travel by lane shuffles, no LDS arrays, no barriers -- about half the
// instructions of the LDS version per step.  Same round-robin pairing (with M = 8), same
// rotations, same stopping rules.
__device__ __forceinline__ void diis_jacobi_reg8(double& a, double& v, int m, double fro, int lane) {
    const int r = lane >> 3, c = lane & 7;
    for (int sweep = 0; sweep < 40; ++sweep) {
        double off = (r < c) ? a * a : 0.0;
        off = nbx_wave_sum(off);
        if (off <= 1e-31 * fro) break;
        double mind = (r == c && r < m) ? fabs(a) : 1.0e300;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) mind = fmin(mind, __shfl_xor(mind, o));
        if (mind - sqrt(2.0 * off) > 1.0e-14) break;  // Weyl: see the LDS version below
        for (int step = 0; step < 7; ++step) {
            // player 7 is fixed and meets `step`; k = step +- l (mod 7) meet each other
            auto mate = [&](int k) {
                if (k == 7) return step;
                if (k == step) return 7;
                int x = 2 * step - k;
                x += x < 0 ? 7 : 0;
                x -= x >= 7 ? 7 : 0;
                return x;
            };
            const int mc = mate(c), mr = mate(r);
            const int p = min(c, mc), q = max(c, mc);
            const double app = __shfl(a, 9 * p), aqq = __shfl(a, 9 * q), apq = __shfl(a, 8 * p + q);
            const double a_rmc = __shfl(a, 8 * r + mc), a_mrc = __shfl(a, 8 * mr + c), a_mrmc = __shfl(a, 8 * mr + mc);
            const double v_rmc = __shfl(v, 8 * r + mc);
            double cs = 1.0, sn = 0.0;
            if (fabs(apq) > 1e-290) {
                const double d = aqq - app, b = 2.0 * apq;
                const double inv = __builtin_amdgcn_rcp(fmax(fabs(d), fabs(b)));
                const double ds = d * inv, bs = b * inv;
                const double rt = sqrt(fma(ds, ds, bs * bs));
                const double t = bs / (ds + (ds >= 0.0 ? rt : -rt));
                cs = rsqrt(fma(t, t, 1.0));
                sn = t * cs;
            }
            // column c of A J is kc * col_c + ks * col_mate(c); the row pair's parameters are those
            // of column r, which lane (0, r) has just computed
            const double kc_c = cs, ks_c = (c < mc) ? -sn : sn;
            const double kc_r = __shfl(kc_c, r), ks_r = __shfl(ks_c, r);
            const double x = kc_c * a + ks_c * a_rmc;       // (A J)[r][c]
            const double y = kc_c * a_mrc + ks_c * a_mrmc;  // (A J)[mate(r)][c]
            a = (ks_r != 0.0 && mr == c) ? 0.0 : kc_r * x + ks_r * y;  // the rotated pair is exactly 0
            v = kc_c * v + ks_c * v_rmc;
        }
    }
}

// One wavefront: finish the dot products, update H, solve the Pulay system as PySCF does.
//   eigenvalues by parallel-order Jacobi in LDS;
//   any |w| < 1e-14  ->  c = V_keep diag(1/w_keep) V_keep^T g      (g = e_0)
//   otherwise        ->  c = LU solve with partial pivoting (numpy.linalg.solve)
__global__ __launch_bounds__(64) void diis_solve_kernel(const double* __restrict__ partial, int nblocks, int nd,
                                                        int slot, double* __restrict__ H, int ldh,
                                                        double* __restrict__ coef) {
    __shared__ double A[DIIS_M + 1][DIIS_M + 2], V[DIIS_M + 1][DIIS_M + 2], A0[DIIS_M][DIIS_M + 1];
    __shared__ double row[DIIS_MAX_SPACE], w[DIIS_M], c_out[DIIS_M];
    __shared__ double kc[DIIS_M + 1], ks[DIIS_M + 1];
    __shared__ int kmate[DIIS_M + 1];
    __shared__ int lu_failed;
    const int lane = threadIdx.x;
    const int m = nd + 1;
    // One trip to L2 for everything the kernel reads: the old H entries (the row / column being
    // replaced is patched from `row` below, and stored once A is built) and all partial sums.
    double hold[DIIS_SLOTS];
#pragma unroll
    for (int e = 0; e < DIIS_SLOTS; ++e) {
        const int idx = lane + 64 * e;
        hold[e] = 0.0;
        if (idx < m * m) {
            const int r = idx / m, c = idx - r * m;
            if (!((r == slot + 1 && c >= 1) || (c == slot + 1 && r >= 1))) hold[e] = H[(int64_t)r * ldh + c];
        }
    }
    {
        double acc[DIIS_MAX_SPACE];
#pragma unroll
        for (int k = 0; k < DIIS_MAX_SPACE; ++k) acc[k] = 0.0;
        for (int b = lane; b < nblocks; b += 64) {
#pragma unroll
            for (int k = 0; k < DIIS_MAX_SPACE; ++k)
                if (k < nd) acc[k] += partial[(int64_t)b * nd + k];
        }
#pragma unroll
        for (int k = 0; k < DIIS_MAX_SPACE; ++k) {
            const double t = nbx_wave_sum(acc[k]);
            if (lane == 0 && k < nd) row[k] = t;
        }
    }
    __syncthreads();
#pragma unroll
    for (int e = 0; e < DIIS_SLOTS; ++e) {
        const int idx = lane + 64 * e;
        if (idx < m * m) {
            const int r = idx / m, c = idx - r * m;
            double v = hold[e];
            if (r == slot + 1 && c >= 1) v = row[c - 1];
            else if (c == slot + 1 && r >= 1) v = row[r - 1];
            A[r][c] = v;
            A0[r][c] = v;
            V[r][c] = (r == c) ? 1.0 : 0.0;
        }
    }
    if (lane < nd) {
        H[(int64_t)(slot + 1) * ldh + lane + 1] = row[lane];
        H[(int64_t)(lane + 1) * ldh + slot + 1] = row[lane];
    }
    __syncthreads();
    // Parallel-order (round-robin) Jacobi: M/2 disjoint rotations per step, M-1 steps per sweep;
    // an odd m is padded with a decoupled zero row/column that never rotates.
    const int M = (m + 1) & ~1, npair = M / 2;
    if (M > m) {
        for (int k = lane; k < M; k += 64) {
            A[M - 1][k] = 0.0;
            A[k][M - 1] = 0.0;
            V[M - 1][k] = V[k][M - 1] = (k == M - 1) ? 1.0 : 0.0;
        }
    }
    __syncthreads();
    double fro = 0.0;
    for (int idx = lane; idx < m * m; idx += 64) {
        const double v = A[idx / m][idx % m];
        fro = fma(v, v, fro);
    }
    fro = nbx_wave_sum(fro);
    // This kernel is ONE wavefront on the SCF's critical path: it issues an instruction every four
    // cycles, so what counts is the instruction count.  The element a lane updates in the rotation
    // pass is fixed: its indices are computed once (no integer division per step), the rotation is
    // computed without forming tau (one square root, one reciprocal square root), and the column
    // and row rotations are one pass over the matrix (three LDS round trips per step, not five).
    int er[DIIS_SLOTS], ec[DIIS_SLOTS];  // the elements (row, column) this lane updates; row M: none
#pragma unroll
    for (int e = 0; e < DIIS_SLOTS; ++e) {
        const int idx = lane + 64 * e;
        er[e] = idx < M * M ? idx / M : M;
        ec[e] = idx < M * M ? idx - (idx / M) * M : 0;
    }
    if (m <= 8) {  // the usual case (pyscf.lib.diis space 6): one element per lane, no LDS traffic
        const int r8 = lane >> 3, c8 = lane & 7;
        double a = (r8 < m && c8 < m) ? A[r8][c8] : 0.0;
        double v = (r8 == c8) ? 1.0 : 0.0;
        diis_jacobi_reg8(a, v, m, fro, lane);
        __syncthreads();
        if (r8 < m && c8 < m) {
            A[r8][c8] = a;
            V[r8][c8] = v;
        }
        __syncthreads();
    }
    for (int sweep = 0; sweep < (m <= 8 ? 0 : 40); ++sweep) {
        double off = 0.0;
#pragma unroll
        for (int e = 0; e < DIIS_SLOTS; ++e)
            if (er[e] < ec[e] && ec[e] < m) off = fma(A[er[e]][ec[e]], A[er[e]][ec[e]], off);
        off = nbx_wave_sum(off);
        if (off <= 1e-31 * fro) break;  // eigenvalue error ~ off^2 / gap: far below 1e-16 |H|
        {
            // Early exit: every eigenvalue lies within ||offdiag||_F of a diagonal entry (Weyl), so
            // once min|a_kk| - ||offdiag||_F > 1e-14 no eigenvalue can be below PySCF's threshold:
            // the LU branch is certain and the remaining sweeps (and the vectors) are not needed.
            double mind = 1.0e300;
            for (int k = lane; k < m; k += 64) mind = fmin(mind, fabs(A[k][k]));
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) mind = fmin(mind, __shfl_xor(mind, o));
            if (mind - sqrt(2.0 * off) > 1.0e-14) break;
        }
        // ring positions of this lane's pair: (step + lane) and (step - lane) mod (M - 1), both
        // advanced by one per step (lane 0 pairs the fixed player M - 1 with `step`)
        int ring_p = lane % (M - 1), ring_q = (M - 1 - lane % (M - 1)) % (M - 1);
        for (int step = 0; step < M - 1; ++step) {
            if (lane < npair) {  // pair `lane` of this step
                int p = lane == 0 ? M - 1 : ring_p, q = lane == 0 ? step : ring_q;
                if (p > q) {
                    const int tmp = p;
                    p = q;
                    q = tmp;
                }
                const double apq = A[p][q], app = A[p][p], aqq = A[q][q];
                double c = 1.0, sn = 0.0;
                if (fabs(apq) > 1e-290) {
                    // t = sgn(tau) / (|tau| + sqrt(1 + tau^2)), tau = d / (2 apq), written without tau:
                    // t = 2 apq / (d + sgn(d) sqrt(d^2 + 4 apq^2)); scaled so the squares cannot underflow
                    // (t is scale invariant: an approximate reciprocal is as good as a division here)
                    const double d = aqq - app, b = 2.0 * apq;
                    const double inv = __builtin_amdgcn_rcp(fmax(fabs(d), fabs(b)));
                    const double ds = d * inv, bs = b * inv;
                    const double r = sqrt(fma(ds, ds, bs * bs));
                    const double t = bs / (ds + (ds >= 0.0 ? r : -r));
                    c = rsqrt(fma(t, t, 1.0));
                    sn = t * c;
                }
                // per index: column k of A J is kc[k] * col_k + ks[k] * col_mate(k)
                kc[p] = c;
                ks[p] = -sn;
                kmate[p] = q;
                kc[q] = c;
                ks[q] = sn;
                kmate[q] = p;
            }
            ring_p = ring_p + 1 == M - 1 ? 0 : ring_p + 1;
            ring_q = ring_q + 1 == M - 1 ? 0 : ring_q + 1;
            __syncthreads();
            // A <- J^T A J and V <- V J in one pass: every element from the old matrices ...
            double na[DIIS_SLOTS], nv[DIIS_SLOTS];
#pragma unroll
            for (int e = 0; e < DIIS_SLOTS; ++e) {
                const int r = er[e], c = ec[e];
                if (r < M) {
                    const int mr = kmate[r], mc = kmate[c];
                    const double cc = kc[c], sc = ks[c], cr = kc[r], sr = ks[r];
                    const double x = cc * A[r][c] + sc * A[r][mc];    // (A J)[r][c]
                    const double y = cc * A[mr][c] + sc * A[mr][mc];  // (A J)[mate(r)][c]
                    na[e] = (sr != 0.0 && mr == c) ? 0.0 : cr * x + sr * y;  // the rotated pair is exactly 0
                    nv[e] = cc * V[r][c] + sc * V[r][mc];
                }
            }
            __syncthreads();
            // ... then stored
#pragma unroll
            for (int e = 0; e < DIIS_SLOTS; ++e) {
                const int r = er[e], c = ec[e];
                if (r < M) {
                    A[r][c] = na[e];
                    V[r][c] = nv[e];
                }
            }
            __syncthreads();
        }
    }
    if (lane < m) w[lane] = A[lane][lane];
    if (lane == 0) lu_failed = 0;
    __syncthreads();
    bool singular = false;
    for (int k = 0; k < m; ++k) singular = singular || (fabs(w[k]) < 1e-14);
    if (!singular) {
        // Gaussian elimination with partial pivoting on A0, rhs e_0 (numpy.linalg.solve / dgesv),
        // by the whole wavefront: pivot search by a lane reduction, the row swap and the rank-one
        // update spread over the lanes (each element sees the same operations as in the serial
        // right-looking elimination), back substitution with lane-parallel dot products.
        __shared__ double rhs_s[DIIS_M];
        if (lane < m) rhs_s[lane] = (lane == 0) ? 1.0 : 0.0;
        __syncthreads();
        for (int k = 0; k < m; ++k) {
            // first row index with the largest |A0[i][k]|, i >= k
            double best = (lane >= k && lane < m) ? fabs(A0[lane][k]) : -1.0;
            int piv = lane;
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) {
                const double ob = __shfl_xor(best, o);
                const int op = __shfl_xor(piv, o);
                if (ob > best || (ob == best && op < piv)) {
                    best = ob;
                    piv = op;
                }
            }
            if (best == 0.0) {  // uniform: every lane holds the reduced values
                if (lane == 0) lu_failed = 1;
                break;
            }
            if (piv != k) {
                if (lane < m) {
                    const double tmp = A0[k][lane];
                    A0[k][lane] = A0[piv][lane];
                    A0[piv][lane] = tmp;
                }
                if (lane == 0) {
                    const double tmp = rhs_s[k];
                    rhs_s[k] = rhs_s[piv];
                    rhs_s[piv] = tmp;
                }
            }
            __syncthreads();
            const double inv = 1.0 / A0[k][k];
            const int rem = m - 1 - k;  // rows / columns below / right of the pivot
            double upd[DIIS_SLOTS];
            double rupd = 0.0;
#pragma unroll
            for (int e = 0; e < DIIS_SLOTS; ++e) {
                const int idx = lane + 64 * e;
                upd[e] = 0.0;
                if (idx < rem * rem) {
                    const int i = k + 1 + idx / rem, jj = k + 1 + idx % rem;
                    const double f = A0[i][k] * inv;
                    upd[e] = A0[i][jj] - f * A0[k][jj];
                }
            }
            if (lane < rem) {
                const double f = A0[k + 1 + lane][k] * inv;
                rupd = rhs_s[k + 1 + lane] - f * rhs_s[k];
            }
            __syncthreads();
#pragma unroll
            for (int e = 0; e < DIIS_SLOTS; ++e) {
                const int idx = lane + 64 * e;
                if (idx < rem * rem) A0[k + 1 + idx / rem][k + 1 + idx % rem] = upd[e];
            }
            if (lane < rem) rhs_s[k + 1 + lane] = rupd;
            __syncthreads();
        }
        __syncthreads();
        if (!lu_failed) {
            for (int i = m - 1; i >= 0; --i) {
                double t = (lane > i && lane < m) ? A0[i][lane] * c_out[lane] : 0.0;
                t = nbx_wave_sum(t);
                if (lane == 0) c_out[i] = (rhs_s[i] - t) / A0[i][i];
                __syncthreads();
            }
        }
        __syncthreads();
    }
    if (singular || lu_failed) {
        if (lane < m) {
            double t = 0.0;
            for (int k = 0; k < m; ++k)
                if (!singular || fabs(w[k]) > 1e-14) t += V[lane][k] * (1.0 / w[k]) * V[0][k];
            c_out[lane] = t;
        }
        __syncthreads();
    }
    if (lane >= 1 && lane < m) coef[lane - 1] = c_out[lane];
}

// ---- the Pulay solve for a space of at most 7 vectors (pyscf.lib.diis keeps 6), as one wavefront's work ----
// Same rule as diis_solve_kernel above; what differs is how long the wavefront takes (30 us there, the longest
// stage of a settled SCF cycle outside J/K):
//   * the partial sums of the new Pulay row arrive in ONE trip to L2 (lane 8 g + c adds the blocks g, g + 8, ...
//     of column c - 1; the eight lanes of a column meet in LDS) where the general kernel waited for sixteen
//     loads one after the other;
//   * the Jacobi steps exchange matrix elements through one LDS write and one batch of reads, and every lane
//     works out BOTH rotations its element takes part in (its column's pair and its row's pair) instead of
//     asking another lane for the second: one exchange per step, not two;
//   * the rotation itself: tan of the angle from single-precision sqrt / rcp (one instruction each), two Newton
//     steps on b t^2 + 2 d t - b = 0 in double, cos from a single-precision rsq with two Newton steps; the double
//     precision sqrt, divide and rsqrt sequences it replaces are ~55 dependent instructions, this is ~30;
//   * the sweep's convergence sums by lane moves (nbx_wave_sum_dpp).
// Results agree with the general kernel to rounding (the same rotations to ~1e-16, sums in another order).
__device__ __forceinline__ void diis_rotation(double app, double aqq, double apq, double& cs, double& sn) {
    const double d = aqq - app, b = 2.0 * apq;
    const double inv = __builtin_amdgcn_rcp(fmax(fabs(d), fabs(b)));  // t is scale invariant: approximate is fine
    const double ds = d * inv, bs = b * inv;
    // the root of smaller magnitude, t = b / (d + sgn(d) sqrt(d^2 + b^2)), first in single precision ...
    const float dsf = (float)ds, bsf = (float)bs;
    const float rt = __builtin_amdgcn_sqrtf(fmaf(dsf, dsf, bsf * bsf));
    const float den = dsf + (ds >= 0.0 ? rt : -rt);  // |den| >= 1
    double t = (double)(bsf * __builtin_amdgcn_rcpf(den));
    // ... then Newton on g(t) = b t^2 + 2 d t - b, g'(t) / 2 = b t + d (|g'| / 2 = sqrt(d^2 + b^2) >= 1 at the
    // root): relative error 3e-7 -> 1e-13 -> rounding.  The second step reuses the first step's 1 / g'.
    const float rgp = __builtin_amdgcn_rcpf((float)fma(bs, t, ds));
    double g = fma(fma(bs, t, 2.0 * ds), t, -bs);
    t = fma(-0.5 * g, (double)rgp, t);
    g = fma(fma(bs, t, 2.0 * ds), t, -bs);
    t = fma(-0.5 * g, (double)rgp, t);
    // c = (1 + t^2)^(-1/2), 1 <= 1 + t^2 <= 2
    const double x = fma(t, t, 1.0), h = 0.5 * x;
    double y = (double)__builtin_amdgcn_rsqf((float)x);
    y = y * fma(-(h * y), y, 1.5);
    y = y * fma(-(h * y), y, 1.5);
    const bool rot = fabs(apq) > 1e-290;  // (also discards the NaNs of d = b = 0)
    cs = rot ? y : 1.0;
    sn = rot ? t * y : 0.0;
}

// lane = 8 r + c holds a = A[r][c], v = V[r][c]; on return the eigenvalues are the diagonal, V the vectors
__device__ __forceinline__ void diis_jacobi8(double& a, double& v, int m, double fro, int lane, double* As,
                                             double* Vs) {
    const int r = lane >> 3, c = lane & 7;
    for (int sweep = 0; sweep < 40; ++sweep) {
        const double off = nbx_wave_sum_dpp((r < c) ? a * a : 0.0);
        if (off <= 1e-31 * fro) break;
        const double mind = nbx_wave_min_dpp((r == c && r < m) ? fabs(a) : 1.0e300);
        if (mind - sqrt(2.0 * off) > 1.0e-14) break;  // Weyl: see diis_solve_kernel
#pragma unroll
        for (int step = 0; step < 7; ++step) {
            // player 7 is fixed and meets `step`; k = step +- l (mod 7) meet each other
            auto mate = [&](int k) {
                int x = 2 * step - k;
                x += x < 0 ? 7 : 0;
                x -= x >= 7 ? 7 : 0;
                return k == 7 ? step : (k == step ? 7 : x);
            };
            const int mc = mate(c), mr = mate(r);
            As[lane] = a;
            Vs[lane] = v;
            __syncthreads();  // (one wavefront per workgroup: a wait on the LDS counter, no s_barrier)
            const double a_cc = As[9 * c], a_mm = As[9 * mc], a_cm = As[8 * min(c, mc) + max(c, mc)];
            const double a_rr = As[9 * r], a_nn = As[9 * mr], a_rn = As[8 * min(r, mr) + max(r, mr)];
            const double a_rmc = As[8 * r + mc], a_mrc = As[8 * mr + c], a_mrmc = As[8 * mr + mc];
            const double v_rmc = Vs[8 * r + mc];
            __syncthreads();
            double cc, sc, cr, sr;
            diis_rotation(c < mc ? a_cc : a_mm, c < mc ? a_mm : a_cc, a_cm, cc, sc);
            diis_rotation(r < mr ? a_rr : a_nn, r < mr ? a_nn : a_rr, a_rn, cr, sr);
            // column k of A J is kc * col_k + ks * col_mate(k), ks = -sin for the smaller index of the pair
            const double ks_c = (c < mc) ? -sc : sc, ks_r = (r < mr) ? -sr : sr;
            const double x = cc * a + ks_c * a_rmc;       // (A J)[r][c]
            const double y = cc * a_mrc + ks_c * a_mrmc;  // (A J)[mate(r)][c]
            a = (ks_r != 0.0 && mr == c) ? 0.0 : cr * x + ks_r * y;  // the rotated pair is exactly 0
            v = cc * v + ks_c * v_rmc;
        }
    }
}

// One wavefront per workgroup; EVERY workgroup solves the same system with the same instructions (the answer is
// the same bits everywhere) and then extrapolates its share of the vector: the solve costs no launch of its own.
// Workgroup 0 also stores the new row / column of H and the coefficients.  nd <= 7.
constexpr int DIIS8_PER_LANE = 4;  // vector elements per lane
__global__ __launch_bounds__(64) void diis_solve8_lincomb_kernel(const double* __restrict__ partial, int nblocks,
                                                                 int nd, int slot, double* __restrict__ H, int ldh,
                                                                 double* __restrict__ coef,
                                                                 double* __restrict__ state, int64_t n,
                                                                 const double* __restrict__ vecs, int64_t stride,
                                                                 double* __restrict__ out) {
    __shared__ double As[64], Vs[64], Ts[64];
    __shared__ double A0[8][9], rhs_s[8], c_out[8];
    __shared__ int lu_failed;
    const int lane = threadIdx.x, r = lane >> 3, c = lane & 7;
    const int m = nd + 1;
    // the vector elements this lane extrapolates: requested now, used after the solve
    const int64_t i0 = (int64_t)blockIdx.x * (64 * DIIS8_PER_LANE) + lane;
    const bool inside = r < m && c < m;
    const bool new_row = inside && r == slot + 1 && c >= 1, new_col = inside && c == slot + 1 && r >= 1;
    const double hold = (inside && !new_row && !new_col) ? H[(int64_t)r * ldh + c] : 0.0;
    double xv[DIIS8_PER_LANE][7];  // (in flight during the solve; 1e300 would show a use of an unloaded one)
#pragma unroll
    for (int j = 0; j < DIIS8_PER_LANE; ++j)
#pragma unroll
        for (int k = 0; k < 7; ++k) xv[j][k] = (k < nd && i0 + 64 * j < n) ? vecs[k * stride + i0 + 64 * j] : 0.0;
    const double v_prev = state[lane];
    // (a basis of a LARGER system would mix the zero padding into the live indices: cold start then)
    const bool warm = __double_as_longlong(state[64]) == DIIS_STATE_MAGIC && state[65] <= (double)(nd + 1);  // uniform
    double acc = 0.0;
    if (c >= 1 && c <= nd) {
#pragma unroll 4
        for (int b = r; b < nblocks; b += 8) acc += partial[(int64_t)b * nd + c - 1];
    }
    As[lane] = acc;
    __syncthreads();
    double row_c = 0.0, row_r = 0.0;  // <e_new, e_(c-1)> and <e_new, e_(r-1)>
#pragma unroll
    for (int g = 0; g < 8; ++g) {
        row_c += As[8 * g + c];
        row_r += As[8 * g + r];
    }
    __syncthreads();
    double a = new_row ? row_c : (new_col ? row_r : hold);
    if (blockIdx.x == 0 && (new_row || new_col)) H[(int64_t)r * ldh + c] = a;
    A0[r][c] = a;
    if (lane == 0) lu_failed = 0;
    double v = (r == c) ? 1.0 : 0.0;
    if (warm) {
        // Start from the basis the previous update's solve ended in: one row / column of H has changed since,
        // so V^T H V is close to diagonal -- 0-3 sweeps where a cold start takes 4-7 (none at all once the error
        // vectors have shrunk to rounding level, the settled cycles of a long run).  Any orthogonal V is a valid
        // start; the rounding drift of many updates is removed first by one Newton-Schulz step,
        // V <- V (3 I - V^T V) / 2 (departure from orthogonality e -> e^2).
        Vs[lane] = v_prev;
        __syncthreads();
        double gram = 0.0;
#pragma unroll
        for (int k = 0; k < 8; ++k) gram = fma(Vs[8 * k + r], Vs[8 * k + c], gram);
        Ts[lane] = gram;
        __syncthreads();
        double vg = 0.0;
#pragma unroll
        for (int k = 0; k < 8; ++k) vg = fma(Vs[8 * r + k], Ts[8 * k + c], vg);
        v = 1.5 * v_prev - 0.5 * vg;
        __syncthreads();
        Vs[lane] = v;
        As[lane] = a;
        __syncthreads();
        double hv = 0.0;  // (H V)[r][c]
#pragma unroll
        for (int k = 0; k < 8; ++k) hv = fma(As[8 * r + k], Vs[8 * k + c], hv);
        Ts[lane] = hv;
        __syncthreads();
        double vhv = 0.0;  // (V^T H V)[r][c]
#pragma unroll
        for (int k = 0; k < 8; ++k) vhv = fma(Vs[8 * k + r], Ts[8 * k + c], vhv);
        __syncthreads();
        Ts[lane] = vhv;
        __syncthreads();
        a = 0.5 * (vhv + Ts[8 * c + r]);
        __syncthreads();
    }
    const double fro = nbx_wave_sum_dpp(a * a);
    diis_jacobi8(a, v, m, fro, lane, As, Vs);
    As[lane] = a;
    Vs[lane] = v;
    if (blockIdx.x == 0) {
        state[DIIS_STATE_HALF + lane] = v;
        if (lane == 0) state[DIIS_STATE_HALF + 64] = __longlong_as_double(DIIS_STATE_MAGIC);
        if (lane == 1) state[DIIS_STATE_HALF + 65] = (double)m;
    }
    __syncthreads();
    const bool singular = __any(r == c && r < m && fabs(a) < 1e-14);
    if (!singular) {
        // Gaussian elimination with partial pivoting on A0, rhs e_0 (numpy.linalg.solve / dgesv): pivot search by
        // a lane reduction, the row swap and the rank-one update spread over the lanes (each element sees the
        // operations of the serial right-looking elimination), back substitution with lane-parallel dot products
        if (lane < m) rhs_s[lane] = (lane == 0) ? 1.0 : 0.0;
        __syncthreads();
        for (int k = 0; k < m; ++k) {
            double best = (lane >= k && lane < m) ? fabs(A0[lane][k]) : -1.0;  // first row with the largest |A0[i][k]|
            int piv = lane;
#pragma unroll
            for (int o = 4; o > 0; o >>= 1) {  // rows live in lanes 0..7
                const double ob = __shfl_xor(best, o);
                const int op = __shfl_xor(piv, o);
                if (ob > best || (ob == best && op < piv)) {
                    best = ob;
                    piv = op;
                }
            }
            best = nbx_readlane_f64(best, 0);
            piv = __builtin_amdgcn_readlane(piv, 0);
            if (best == 0.0) {
                if (lane == 0) lu_failed = 1;
                break;
            }
            if (piv != k) {
                if (lane < m) {
                    const double tmp = A0[k][lane];
                    A0[k][lane] = A0[piv][lane];
                    A0[piv][lane] = tmp;
                }
                if (lane == 0) {
                    const double tmp = rhs_s[k];
                    rhs_s[k] = rhs_s[piv];
                    rhs_s[piv] = tmp;
                }
            }
            __syncthreads();
            const double inv = 1.0 / A0[k][k];
            const bool below = r > k && r < m, right = c > k && c < m;
            double upd = 0.0, rupd = 0.0;
            if (below && right) upd = A0[r][c] - (A0[r][k] * inv) * A0[k][c];
            if (below && c == 0) rupd = rhs_s[r] - (A0[r][k] * inv) * rhs_s[k];
            __syncthreads();
            if (below && right) A0[r][c] = upd;
            if (below && c == 0) rhs_s[r] = rupd;
            __syncthreads();
        }
        __syncthreads();
        if (!lu_failed) {
            for (int i = m - 1; i >= 0; --i) {
                const double t = nbx_wave_sum_dpp((lane > i && lane < m) ? A0[i][lane] * c_out[lane] : 0.0);
                if (lane == 0) c_out[i] = (rhs_s[i] - t) / A0[i][i];
                __syncthreads();
            }
        }
        __syncthreads();
    }
    if (singular || lu_failed) {
        // c = V_keep diag(1 / w_keep) V_keep^T e_0: lane (r, k) holds term k of c[r], the eight meet by lane moves
        const double w = As[9 * c], v0 = Vs[c];
        const bool keep = c < m && (!singular || fabs(w) > 1e-14);
        const double t = nbx_sum8_dpp(keep ? v * (1.0 / w) * v0 : 0.0);
        if (c == 0 && r < m) c_out[r] = t;
        __syncthreads();
    }
    if (blockIdx.x == 0 && lane >= 1 && lane < m) coef[lane - 1] = c_out[lane];
    double cf[7];
#pragma unroll
    for (int k = 0; k < 7; ++k) cf[k] = (k < nd) ? c_out[k + 1] : 0.0;
#pragma unroll
    for (int j = 0; j < DIIS8_PER_LANE; ++j) {
        const int64_t i = i0 + 64 * j;
        if (i < n) {
            double t = 0.0;
#pragma unroll
            for (int k = 0; k < 7; ++k)
                if (k < nd) t = fma(cf[k], xv[j][k], t);
            out[i] = t;
        }
    }
}

// out[i] = sum_k coef[k] * vecs[k][i] with the coefficients read from device memory
__global__ void lincomb_dev_kernel(int64_t n, int nvec, const double* __restrict__ coef,
                                   const double* __restrict__ vecs, int64_t stride, double* __restrict__ out) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        double t = 0.0;
        for (int k = 0; k < nvec; ++k) t = fma(coef[k], vecs[k * stride + i], t);
        out[i] = t;
    }
}

// B[b] (cols x rows) = A[b]^T (A: rows x cols)
__global__ void transpose_kernel(const double* __restrict__ A, double* __restrict__ B, int64_t rows, int64_t cols) {
    __shared__ double t[TILE][TILE + 1];
    const int64_t b = blockIdx.z;
    A += b * rows * cols;
    B += b * rows * cols;
    const int64_t r0 = (int64_t)blockIdx.y * TILE, c0 = (int64_t)blockIdx.x * TILE;
    for (int r = threadIdx.y; r < TILE; r += blockDim.y) {
        const int64_t gr = r0 + r, gc = c0 + threadIdx.x;
        if (gr < rows && gc < cols) t[r][threadIdx.x] = A[gr * cols + gc];
    }
    __syncthreads();
    for (int c = threadIdx.y; c < TILE; c += blockDim.y) {
        const int64_t gc = c0 + c, gr = r0 + threadIdx.x;
        if (gr < rows && gc < cols) B[gc * rows + gr] = t[threadIdx.x][c];
    }
}

// out[x] = sum over a >= nocc[x], i < nocc[x] of fmo[x][a][i]^2: the squared norm of the
// virtual-occupied block of the MO-basis Fock matrix (the orbital gradient of PySCF's kernel)
__global__ void vo_sumsq_kernel(const double* __restrict__ fmo, int N, int nocc_a, int nocc_b,
                                double* __restrict__ partial) {
    __shared__ double red[17];
    const int x = blockIdx.y;
    const int nocc = x == 0 ? nocc_a : nocc_b;
    const int nvir = N - nocc;
    const int64_t total = (int64_t)nvir * nocc;
    double t = 0.0;
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (int64_t)gridDim.x * blockDim.x) {
        const int a = nocc + (int)(e / nocc), i = (int)(e % nocc);
        const double v = fmo[((int64_t)x * N + a) * N + i];
        t = fma(v, v, t);
    }
    t = nbx_block_sum(t, red);
    if (threadIdx.x == 0) partial[(int64_t)blockIdx.x * 2 + x] = t;
}

__global__ void scale_cols_kernel(int64_t rows, int64_t cols, const double* __restrict__ s, double* __restrict__ a) {
    const int64_t b = blockIdx.y;
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= rows * cols) return;
    a[b * rows * cols + i] *= s[b * cols + (i % cols)];
}

// h2[P,Q,R,S] over the (2n)^4 output, coalesced stores (nbed/ham_builder.py:180-214)
// elements [idx0, idx0 + count) of the flattened tensor go to h2[0 .. count)
__global__ void spinorb_h2_kernel(int n, const double* __restrict__ tb, double tol, double scale,
                                  double* __restrict__ h2, int64_t idx0, int64_t count) {
    const int64_t nq = 2 * (int64_t)n;
    const int64_t n4 = (int64_t)n * n * n * n;
    for (int64_t off = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; off < count;
         off += (int64_t)gridDim.x * blockDim.x) {
        const int64_t idx = idx0 + off;
        const int64_t S = idx % nq;
        int64_t t = idx / nq;
        const int64_t R = t % nq;
        t /= nq;
        const int64_t Q = t % nq;
        const int64_t P = t / nq;
        const int sp = (int)(P & 1), sq = (int)(Q & 1), sr = (int)(R & 1), ss = (int)(S & 1);
        int blk = -1;
        if (sp == sq && sq == sr && sr == ss) blk = sp;          // aaaa -> 0, bbbb -> 1
        else if (sp == 0 && sq == 1 && sr == 1 && ss == 0) blk = 2;  // [2p,2q+1,2r+1,2s] <- aabb block
        else if (sp == 1 && sq == 0 && sr == 0 && ss == 1) blk = 3;  // [2p+1,2q,2r,2s+1] <- bbaa block
        double v = 0.0;
        if (blk >= 0) {
            const int64_t p = P >> 1, q = Q >> 1, r = R >> 1, s = S >> 1;
            v = tb[blk * n4 + ((p * n + q) * n + r) * n + s];
            if (fabs(v) < tol) v = 0.0;
            v *= scale;
        }
        h2[off] = v;
    }
}

// x <- (|x| < tol ? 0 : x) * scale: the truncation and the 1/2 of nbed/ham_builder.py:213-214,254 applied to
// a spatial block (the "packed" Hamiltonian output keeps the blocks instead of the scattered tensor)
__global__ void threshold_scale_kernel(int64_t n, double tol, double scale, double* __restrict__ x) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const double v = x[i];
        x[i] = (fabs(v) < tol ? 0.0 : v) * scale;
    }
}

__global__ void spinorb_h1_kernel(int n, const double* __restrict__ ob, double tol, double* __restrict__ h1) {
    const int nq = 2 * n;
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= nq * nq) return;
    const int P = idx / nq, Q = idx % nq;
    double v = 0.0;
    if ((P & 1) == (Q & 1)) {
        v = ob[(int64_t)(P & 1) * n * n + (int64_t)(P >> 1) * n + (Q >> 1)];
        if (fabs(v) < tol) v = 0.0;
    }
    h1[idx] = v;
}

inline unsigned grid1d(int64_t n, int block, int64_t cap = 65536) {
    int64_t g = nbx_cdiv(n, block);
    if (g > cap) g = cap;
    if (g < 1) g = 1;
    return (unsigned)g;
}

// second-stage reduce + copy to host (synchronises)
int finish_reduction(nbx_ctx* ctx, int nblocks, int nout, double* h_out) {
    hipLaunchKernelGGL(final_reduce_kernel, dim3(1), dim3(256), 0, ctx->stream, ctx->d_scratch, nblocks, nout,
                       ctx->d_scratch + NBX_SCRATCH_DOUBLES - 64, nout);
    NBX_LAUNCH_CHECK();
    NBX_HIP(hipMemcpyAsync(ctx->h_pinned, ctx->d_scratch + NBX_SCRATCH_DOUBLES - 64, nout * sizeof(double),
                           hipMemcpyDeviceToHost, ctx->stream));
    NBX_HIP(hipStreamSynchronize(ctx->stream));
    for (int i = 0; i < nout; ++i) h_out[i] = ctx->h_pinned[i];
    return NBX_OK;
}

}  // namespace

extern "C" {

int nbx_fock_uhf(nbx_ctx* ctx, int64_t nao, const double* d_hcore, int hcore_ndim, const double* d_vemb,
                 const double* d_jk, double* d_fock, double* d_vhf) {
    NBX_CHECK_ARG(ctx && d_hcore && d_jk && d_fock && nao > 0);
    NBX_CHECK_ARG(hcore_ndim == 2 || hcore_ndim == 3);
    const int64_t n2 = nao * nao;
    hipLaunchKernelGGL(fock_uhf_kernel, dim3((unsigned)nbx_cdiv(n2, 256)), dim3(256), 0, ctx->stream, d_hcore,
                       hcore_ndim == 3 ? 1 : 0, d_vemb, d_jk, d_fock, d_vhf, n2);
    NBX_LAUNCH_CHECK();
    return NBX_OK;
}

int nbx_huzinaga_sym(nbx_ctx* ctx, int64_t nao, int64_t batch, const double* d_fds, double kappa, double* d_hz,
                     double* d_fock_io) {
    NBX_CHECK_ARG(ctx && d_fds && d_hz && nao > 0 && batch > 0 && batch < 65536);
    const unsigned g = (unsigned)nbx_cdiv(nao, TILE);
    hipLaunchKernelGGL(huzinaga_sym_kernel, dim3(g, g, (unsigned)batch), dim3(TILE, 8), 0, ctx->stream, d_fds,
                       (int)nao, kappa, d_hz, d_fock_io);
    NBX_LAUNCH_CHECK();
    return NBX_OK;
}

int nbx_huzinaga_fused(nbx_ctx* ctx, int64_t nao, int64_t batch, const double* d_f, const double* d_ds, double kappa,
                       double* d_hz, double* d_fock_out) {
    NBX_CHECK_ARG(ctx && d_f && d_ds && d_hz && nao > 0 && batch > 0 && batch < 65536);
    NBX_CHECK_ARG(d_fock_out != d_f);  // every workgroup reads all of F: the update is out of place
    const unsigned g = (unsigned)nbx_cdiv(nao, 16);
    hipLaunchKernelGGL(huzinaga_fused_kernel, dim3(g, g, (unsigned)batch), dim3(64), 0, ctx->stream, d_f, d_ds,
                       (int)nao, kappa, d_hz, d_fock_out);
    NBX_LAUNCH_CHECK();
    return NBX_OK;
}

int nbx_trace_prod(nbx_ctx* ctx, int64_t nao, int64_t batch, const double* d_a, const double* d_b,
                   double* h_out) {
    NBX_CHECK_ARG(ctx && d_a && d_b && h_out && nao > 0 && batch > 0);
    const int64_t g = nbx_cdiv(nao, TILE);
    NBX_CHECK_ARG(g * g <= NBX_SCRATCH_DOUBLES - 64 && batch <= 64);
    // one batch entry at a time keeps the scratch small (batch is the spin axis: 1 or 2)
    for (int64_t b = 0; b < batch; ++b) {
        hipLaunchKernelGGL(trace_prod_kernel, dim3((unsigned)g, (unsigned)g, 1), dim3(TILE, 8), 0, ctx->stream,
                           d_a + b * nao * nao, d_b + b * nao * nao, (int)nao, ctx->d_scratch);
        NBX_LAUNCH_CHECK();
        const int rc = finish_reduction(ctx, (int)(g * g), 1, h_out + b);
        if (rc != NBX_OK) return rc;
    }
    return NBX_OK;
}

int nbx_huz_cycle_scalars(nbx_ctx* ctx, int64_t nao, const double* d_hcore, int hcore_ndim,
                          const double* d_vemb, const double* d_vhf, const double* d_hz, const double* d_dm,
                          const double* d_dm_old, double* h_out) {
    NBX_CHECK_ARG(ctx && d_hcore && d_vhf && d_hz && d_dm && d_dm_old && h_out && nao > 0);
    NBX_CHECK_ARG(hcore_ndim == 2 || hcore_ndim == 3);
    const int64_t g = nbx_cdiv(nao, TILE);
    NBX_CHECK_ARG(g * g * 4 <= NBX_SCRATCH_DOUBLES - 64);
    hipLaunchKernelGGL(huz_scalars_kernel<TILE>, dim3((unsigned)g, (unsigned)g), dim3(TILE, 8), 0, ctx->stream, d_hcore,
                       hcore_ndim == 3 ? 1 : 0, d_vemb, d_vhf, d_hz, d_dm, d_dm_old, (int)nao, ctx->d_scratch);
    NBX_LAUNCH_CHECK();
    const int rc = finish_reduction(ctx, (int)(g * g), 4, h_out);
    if (rc != NBX_OK) return rc;
    h_out[2] = sqrt(h_out[2]);
    h_out[3] = sqrt(h_out[3]);
    return NBX_OK;
}

int nbx_huz_cycle_scalars_dev(nbx_ctx* ctx, int64_t nao, const double* d_hcore, int hcore_ndim,
                              const double* d_vemb, const double* d_vhf, const double* d_hz, const double* d_dm,
                              const double* d_dm_old, double* d_out, const int* d_tail, int64_t tail_n) {
    return nbx_huz_cycle_scalars_dts(ctx, nao, d_hcore, hcore_ndim, d_vemb, d_vhf, d_hz, d_dm, d_dm_old, d_out, d_tail,
                                     tail_n, nullptr);
}

int nbx_huz_cycle_scalars_dts(nbx_ctx* ctx, int64_t nao, const double* d_hcore, int hcore_ndim,
                              const double* d_vemb, const double* d_vhf, const double* d_hz, const double* d_dm,
                              const double* d_dm_old, double* d_out, const int* d_tail, int64_t tail_n,
                              double* d_dts) {
    NBX_CHECK_ARG(d_hz != nullptr);
    return nbx_cycle_scalars_launch(ctx, nao, d_hcore, hcore_ndim, d_vemb, d_vhf, d_hz, d_dm, d_dm_old, d_out, d_tail,
                                    tail_n, d_dts, nullptr, 0);
}

}  // extern "C"

// The launch behind the nbx_huz_cycle_scalars_* entry points; also (d_hz == NULL, d_dtail: dtail_n device doubles
// forwarded behind the status words) the scalars of the mu-shift cycle (scf_cycle.hip).
int nbx_cycle_scalars_launch(nbx_ctx* ctx, int64_t nao, const double* d_hcore, int hcore_ndim, const double* d_vemb,
                             const double* d_vhf, const double* d_hz, const double* d_dm, const double* d_dm_old,
                             double* d_out, const int* d_tail, int64_t tail_n, double* d_dts, const double* d_dtail,
                             int64_t dtail_n) {
    NBX_CHECK_ARG(ctx && d_hcore && d_vhf && d_dm && d_dm_old && d_out && nao > 0);
    int m4 = 0, nb4 = d_dts ? s4_nb(nao) : 0, lpt4 = d_dts ? s4_lpt(nao) : 0;
    if (d_dts && (nbx_jk_m8_covers(nao) || nbx_jk_m4_covers(nao))) {  // (the table in jk_m4.hip's / jk_m8.hip's order: its layout parameters travel in these three)
        int wl[4];
        if (nbx_jk_m8_covers(nao)) nbx_jk_m8_weight_layout(nao, wl);
        else nbx_jk_m4_weight_layout(nao, wl);
        nb4 = wl[0];
        lpt4 = wl[1];
        m4 = (wl[2] << 8) | wl[3];
    }
    NBX_CHECK_ARG(d_dts == nullptr || lpt4 > 0);  // the table exists for sizes nbx_jk_packed covers
    NBX_CHECK_ARG(tail_n >= 0 && tail_n <= 64 && (tail_n == 0 || d_tail != nullptr));
    NBX_CHECK_ARG(dtail_n >= 0 && dtail_n <= 64 && (dtail_n == 0 || d_dtail != nullptr));
    NBX_CHECK_ARG(hcore_ndim == 2 || hcore_ndim == 3);
    // 16 x 16 tiles while the partials fit the scratch (N <= 500): more workgroups for small N
    const bool fine = nbx_cdiv(nao, 16) * nbx_cdiv(nao, 16) * 4 <= NBX_SCRATCH_DOUBLES - 64;
    const int64_t g = nbx_cdiv(nao, fine ? 16 : TILE);
    NBX_CHECK_ARG(g * g * 4 <= NBX_SCRATCH_DOUBLES - 64);
    if (fine)
        hipLaunchKernelGGL(huz_scalars_kernel<16>, dim3((unsigned)g, (unsigned)g), dim3(16, 8), 0, ctx->stream, d_hcore,
                           hcore_ndim == 3 ? 1 : 0, d_vemb, d_vhf, d_hz, d_dm, d_dm_old, (int)nao, ctx->d_scratch,
                           d_out, d_tail, (int)tail_n, ctx->d_counters + NBX_COUNTERS - 1, d_dts, nb4, lpt4, m4, d_dtail,
                           (int)dtail_n);
    else
        hipLaunchKernelGGL(huz_scalars_kernel<TILE>, dim3((unsigned)g, (unsigned)g), dim3(TILE, 8), 0, ctx->stream,
                           d_hcore, hcore_ndim == 3 ? 1 : 0, d_vemb, d_vhf, d_hz, d_dm, d_dm_old, (int)nao,
                           ctx->d_scratch, d_out, d_tail, (int)tail_n, ctx->d_counters + NBX_COUNTERS - 1, d_dts, nb4,
                           lpt4, m4, d_dtail, (int)dtail_n);
    NBX_LAUNCH_CHECK();
    return NBX_OK;
}

// The aufbau density D[x] = C[x][:, :nocc_x] C[x][:, :nocc_x]^T AND the scalars of the cycle that judge it, in one
// launch (huz_scalars_kernel<16, true>): what nbx_gemm('N','T') + nbx_huz_cycle_scalars_dts give, bit for bit.
// Returns NBX_E_UNSUPPORTED (nothing launched) where the fused form does not apply -- more than 64 occupied orbitals,
// an empty spin, N > 500 -- and the caller takes the two launches.
int nbx_density_scalars_launch(nbx_ctx* ctx, int64_t nao, const double* d_hcore, int hcore_ndim, const double* d_vemb,
                               const double* d_vhf, const double* d_hz, const double* d_c, int64_t nocc_a, int64_t nocc_b,
                               double* d_dm_out, const double* d_dm_old, double* d_out, const int* d_tail, int64_t tail_n,
                               double* d_dts) {
    NBX_CHECK_ARG(ctx && d_hcore && d_vhf && d_c && d_dm_out && d_dm_old && d_out && nao > 0);
    const bool fine = nbx_cdiv(nao, 16) * nbx_cdiv(nao, 16) * 4 <= NBX_SCRATCH_DOUBLES - 64;
    if (!fine || nocc_a <= 0 || nocc_b <= 0 || nocc_a > 64 || nocc_b > 64) return NBX_E_UNSUPPORTED;
    int m4 = 0, nb4 = d_dts ? s4_nb(nao) : 0, lpt4 = d_dts ? s4_lpt(nao) : 0;
    if (d_dts && (nbx_jk_m8_covers(nao) || nbx_jk_m4_covers(nao))) {
        int wl[4];
        if (nbx_jk_m8_covers(nao)) nbx_jk_m8_weight_layout(nao, wl);
        else nbx_jk_m4_weight_layout(nao, wl);
        nb4 = wl[0];
        lpt4 = wl[1];
        m4 = (wl[2] << 8) | wl[3];
    }
    NBX_CHECK_ARG(d_dts == nullptr || lpt4 > 0);
    NBX_CHECK_ARG(tail_n >= 0 && tail_n <= 64 && (tail_n == 0 || d_tail != nullptr));
    NBX_CHECK_ARG(hcore_ndim == 2 || hcore_ndim == 3);
    const int64_t g = nbx_cdiv(nao, 16);
    hipLaunchKernelGGL((huz_scalars_kernel<16, true>), dim3((unsigned)g, (unsigned)g), dim3(16, 8), 0, ctx->stream, d_hcore,
                       hcore_ndim == 3 ? 1 : 0, d_vemb, d_vhf, d_hz, d_dm_out, d_dm_old, (int)nao, ctx->d_scratch, d_out,
                       d_tail, (int)tail_n, ctx->d_counters + NBX_COUNTERS - 1, d_dts, nb4, lpt4, m4, nullptr, 0, d_c,
                       (int)nocc_a, (int)nocc_b, d_dm_out);
    NBX_LAUNCH_CHECK();
    return NBX_OK;
}

// out[b] = A[b]^T - A[b] (antisym_kernel)
int nbx_antisym(nbx_ctx* ctx, int64_t nao, int64_t batch, const double* d_a, double* d_out) {
    NBX_CHECK_ARG(ctx && d_a && d_out && d_a != d_out && nao > 0 && batch > 0 && batch < 65536);
    const unsigned g = (unsigned)nbx_cdiv(nao, TILE);
    hipLaunchKernelGGL(antisym_kernel, dim3(g, g, (unsigned)batch), dim3(TILE, 8), 0, ctx->stream, d_a, (int)nao, d_out);
    NBX_LAUNCH_CHECK();
    return NBX_OK;
}

extern "C" {

size_t nbx_diis_coef_doubles(int64_t space) { return space > 0 ? (size_t)(space + DIIS_STATE_DOUBLES) : 0; }

int nbx_diis_update(nbx_ctx* ctx, int64_t n, int64_t space, int64_t slot, int64_t nd, const double* d_x,
                    double* d_xprev, double* d_xs, double* d_es, double* d_h, double* d_coef) {
    return nbx_diis_update_err(ctx, n, space, slot, nd, d_x, nullptr, d_xprev, d_xs, d_es, d_h, d_coef);
}

int nbx_diis_update_err(nbx_ctx* ctx, int64_t n, int64_t space, int64_t slot, int64_t nd, const double* d_x,
                        const double* d_err, double* d_xprev, double* d_xs, double* d_es, double* d_h,
                        double* d_coef) {
    return nbx_diis_update_anti(ctx, n, space, slot, nd, d_x, d_err, 0, d_xprev, d_xs, d_es, d_h, d_coef);
}

}  // extern "C"

// nbx_diis_update_err with the error vector given as the matrices A (order anti_n > 0; n = batch anti_n^2) whose
// antisymmetric part A^T - A it is (scf_cycle.hip: the CDIIS step of the mu-shift cycle); anti_n = 0: d_err as it is
int nbx_diis_update_anti(nbx_ctx* ctx, int64_t n, int64_t space, int64_t slot, int64_t nd, const double* d_x,
                         const double* d_err, int64_t anti_n, double* d_xprev, double* d_xs, double* d_es, double* d_h,
                         double* d_coef) {
    NBX_CHECK_ARG(ctx && d_x && d_xprev && d_xs && d_es && d_h && d_coef && n > 0);
    NBX_CHECK_ARG(anti_n == 0 || (d_err != nullptr && anti_n > 0 && n % (anti_n * anti_n) == 0));
    NBX_CHECK_ARG(space >= 1 && space <= DIIS_MAX_SPACE && slot >= 0 && slot < space && nd >= 1 && nd <= space &&
                  slot < nd);
    const unsigned blocks = grid1d(n, 256, 128);
    NBX_CHECK_ARG((int64_t)blocks * nd <= NBX_SCRATCH_DOUBLES - 64);
    hipLaunchKernelGGL(diis_push_kernel, dim3(blocks), dim3(256), 0, ctx->stream, n, (int)nd, (int)slot, d_x,
                       d_xprev, d_err, d_xs, d_es, ctx->d_scratch, d_coef + space, (int)anti_n);
    NBX_LAUNCH_CHECK();
    if (nd <= 7 && nbx_cdiv(n, 64 * DIIS8_PER_LANE) <= 65535) {  // the usual case: solve and extrapolation in one launch
        hipLaunchKernelGGL(diis_solve8_lincomb_kernel, dim3((unsigned)nbx_cdiv(n, 64 * DIIS8_PER_LANE)), dim3(64), 0,
                           ctx->stream, ctx->d_scratch, (int)blocks, (int)nd, (int)slot, d_h, (int)(space + 1), d_coef,
                           d_coef + space, n, d_xs, n, d_xprev);
        NBX_LAUNCH_CHECK();
        return NBX_OK;
    }
    hipLaunchKernelGGL(diis_solve_kernel, dim3(1), dim3(64), 0, ctx->stream, ctx->d_scratch, (int)blocks, (int)nd,
                       (int)slot, d_h, (int)(space + 1), d_coef);
    NBX_LAUNCH_CHECK();
    hipLaunchKernelGGL(lincomb_dev_kernel, dim3(grid1d(n, 256)), dim3(256), 0, ctx->stream, n, (int)nd, d_coef, d_xs,
                       n, d_xprev);
    NBX_LAUNCH_CHECK();
    return NBX_OK;
}

extern "C" {

int nbx_vo_sumsq(nbx_ctx* ctx, int64_t nao, const double* d_fmo, int64_t nocc_a, int64_t nocc_b, double* d_out) {
    NBX_CHECK_ARG(ctx && d_fmo && d_out && nao > 0 && nocc_a >= 0 && nocc_a <= nao && nocc_b >= 0 && nocc_b <= nao);
    // (the orbital gradient of a cycle at N ~ 150 is 2 x 3800 numbers: one workgroup per spin writes its sum itself)
    const int64_t na = (nao - nocc_a) * nocc_a, nb = (nao - nocc_b) * nocc_b;
    if ((na > nb ? na : nb) <= 16384) {
        hipLaunchKernelGGL(vo_sumsq_kernel, dim3(1, 2), dim3(256), 0, ctx->stream, d_fmo, (int)nao, (int)nocc_a, (int)nocc_b,
                           d_out);
        NBX_LAUNCH_CHECK();
        return NBX_OK;
    }
    const int blocks = 32;
    hipLaunchKernelGGL(vo_sumsq_kernel, dim3(blocks, 2), dim3(256), 0, ctx->stream, d_fmo, (int)nao, (int)nocc_a,
                       (int)nocc_b, ctx->d_scratch);
    NBX_LAUNCH_CHECK();
    hipLaunchKernelGGL(final_reduce_kernel, dim3(1), dim3(256), 0, ctx->stream, ctx->d_scratch, blocks, 2, d_out, 2);
    NBX_LAUNCH_CHECK();
    return NBX_OK;
}

int nbx_axpby(nbx_ctx* ctx, int64_t n, double a, const double* d_x, double b, double* d_y) {
    NBX_CHECK_ARG(ctx && n >= 0);
    if (n == 0) return NBX_OK;
    NBX_CHECK_ARG(d_x && d_y);
    hipLaunchKernelGGL(axpby_kernel, dim3(grid1d(n, 256)), dim3(256), 0, ctx->stream, n, a, d_x, b, d_y);
    NBX_LAUNCH_CHECK();
    return NBX_OK;
}

int nbx_lincomb(nbx_ctx* ctx, int64_t n, int64_t nvec, const double* h_coef, const double* d_vecs,
                int64_t stride, double* d_out) {
    NBX_CHECK_ARG(ctx && n >= 0 && nvec > 0 && nvec <= MAX_LINCOMB && h_coef && d_vecs && d_out);
    if (n == 0) return NBX_OK;
    Coefs cf;
    for (int k = 0; k < MAX_LINCOMB; ++k) cf.c[k] = k < nvec ? h_coef[k] : 0.0;
    hipLaunchKernelGGL(lincomb_kernel, dim3(grid1d(n, 256)), dim3(256), 0, ctx->stream, n, (int)nvec, cf, d_vecs,
                       stride, d_out);
    NBX_LAUNCH_CHECK();
    return NBX_OK;
}

int nbx_dots(nbx_ctx* ctx, int64_t n, int64_t nvec, const double* d_x, const double* d_vecs, int64_t stride,
             double* h_out) {
    NBX_CHECK_ARG(ctx && n > 0 && nvec > 0 && nvec <= MAX_LINCOMB && d_x && d_vecs && h_out);
    const unsigned blocks = grid1d(n, 256, 128);
    hipLaunchKernelGGL(dots_kernel, dim3(blocks), dim3(256), 0, ctx->stream, n, (int)nvec, d_x, d_vecs, stride,
                       ctx->d_scratch);
    NBX_LAUNCH_CHECK();
    return finish_reduction(ctx, (int)blocks, (int)nvec, h_out);
}

int nbx_transpose(nbx_ctx* ctx, int64_t rows, int64_t cols, int64_t batch, const double* d_a, double* d_b) {
    NBX_CHECK_ARG(ctx && rows >= 0 && cols >= 0 && batch >= 0);
    if (rows == 0 || cols == 0 || batch == 0) return NBX_OK;
    NBX_CHECK_ARG(d_a && d_b && d_a != d_b);
    const int64_t gy = nbx_cdiv(rows, TILE), gx = nbx_cdiv(cols, TILE);
    if (gy > 65535) {
        nbx_set_error("nbx_transpose: more than 65535*32 rows per matrix is unsupported");
        return NBX_E_UNSUPPORTED;
    }
    for (int64_t b0 = 0; b0 < batch; b0 += 65535) {
        const int64_t nb = (batch - b0) < 65535 ? (batch - b0) : 65535;
        hipLaunchKernelGGL(transpose_kernel, dim3((unsigned)gx, (unsigned)gy, (unsigned)nb), dim3(TILE, 8), 0,
                           ctx->stream, d_a + b0 * rows * cols, d_b + b0 * rows * cols, rows, cols);
        NBX_LAUNCH_CHECK();
    }
    return NBX_OK;
}

int nbx_scale_cols(nbx_ctx* ctx, int64_t rows, int64_t cols, int64_t batch, const double* d_s, double* d_a) {
    NBX_CHECK_ARG(ctx && rows > 0 && cols > 0 && batch > 0 && batch < 65536 && d_s && d_a);
    hipLaunchKernelGGL(scale_cols_kernel, dim3((unsigned)nbx_cdiv(rows * cols, 256), (unsigned)batch), dim3(256), 0,
                       ctx->stream, rows, cols, d_s, d_a);
    NBX_LAUNCH_CHECK();
    return NBX_OK;
}

int nbx_chem_to_phys(nbx_ctx* ctx, int64_t n1, int64_t n2, int64_t n3, int64_t n4, const double* d_in,
                     double* d_out) {
    // out[a,c,d,b] = in[a,b,c,d]: per a, transpose the (n2) x (n3*n4) matrix
    return nbx_transpose(ctx, n2, n3 * n4, n1, d_in, d_out);
}

int nbx_spinorb_scatter(nbx_ctx* ctx, int64_t n, const double* d_one_body, const double* d_two_body, double tol,
                        double h2_scale, double* d_h1, double* d_h2) {
    NBX_CHECK_ARG(ctx && n > 0 && n < 1024 && d_one_body && d_two_body && d_h1 && d_h2);
    const int64_t nq = 2 * n;
    hipLaunchKernelGGL(spinorb_h1_kernel, dim3((unsigned)nbx_cdiv(nq * nq, 256)), dim3(256), 0, ctx->stream, (int)n,
                       d_one_body, tol, d_h1);
    NBX_LAUNCH_CHECK();
    const int64_t total = nq * nq * nq * nq;
    hipLaunchKernelGGL(spinorb_h2_kernel, dim3(grid1d(total, 256, 8192)), dim3(256), 0, ctx->stream, (int)n,
                       d_two_body, tol, h2_scale, d_h2, (int64_t)0, total);
    NBX_LAUNCH_CHECK();
    return NBX_OK;
}

int nbx_spinorb_scatter_h1(nbx_ctx* ctx, int64_t n, const double* d_one_body, double tol, double* d_h1) {
    NBX_CHECK_ARG(ctx && n > 0 && n < 1024 && d_one_body && d_h1);
    const int64_t nq = 2 * n;
    hipLaunchKernelGGL(spinorb_h1_kernel, dim3((unsigned)nbx_cdiv(nq * nq, 256)), dim3(256), 0, ctx->stream, (int)n,
                       d_one_body, tol, d_h1);
    NBX_LAUNCH_CHECK();
    return NBX_OK;
}

int nbx_spinorb_scatter_range(nbx_ctx* ctx, int64_t n, const double* d_two_body, double tol, double h2_scale,
                              int64_t idx0, int64_t count, double* d_h2_part) {
    NBX_CHECK_ARG(ctx && n > 0 && n < 1024 && d_two_body && idx0 >= 0 && count >= 0);
    const int64_t nq = 2 * n;
    NBX_CHECK_ARG(idx0 + count <= nq * nq * nq * nq);
    if (count == 0) return NBX_OK;
    NBX_CHECK_ARG(d_h2_part != nullptr);
    hipLaunchKernelGGL(spinorb_h2_kernel, dim3(grid1d(count, 256, 8192)), dim3(256), 0, ctx->stream, (int)n,
                       d_two_body, tol, h2_scale, d_h2_part, idx0, count);
    NBX_LAUNCH_CHECK();
    return NBX_OK;
}


int nbx_threshold_scale(nbx_ctx* ctx, int64_t n, double tol, double scale, double* d_x) {
    NBX_CHECK_ARG(ctx && n >= 0);
    if (n == 0) return NBX_OK;
    NBX_CHECK_ARG(d_x != nullptr);
    hipLaunchKernelGGL(threshold_scale_kernel, dim3(grid1d(n, 256, 8192)), dim3(256), 0, ctx->stream, n, tol, scale, d_x);
    NBX_LAUNCH_CHECK();
    return NBX_OK;
}

}  // extern "C"
