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
// Hand-off discipline (MI355X: private L2 per XCD, L1 never refreshed by other CUs' stores): a handed-over double
// travels as a 16-byte word (value, tag) -- one sc1 store, one sc1 load -- whose tag mixes (launch nonce, step) with the
// bits of the value; a reader polls the words it needs until each verifies (tdg_put / tdg_verified below).  There is
// no grid-wide barrier: what orders two workgroups is only the data one needs of the other, a workgroup whose rows
// have all left the trailing block leaves the kernel, and the two parities of the hand-over buffers cannot be
// overwritten early (a word of step j + 1 can only be written after every word of step j was read, which was written
// after its writer had read step j - 1).  The launch is cooperative (every workgroup resident, one per CU: a poll must
// not wait for a workgroup that has no CU) and a poll gives up after ~seconds (status word set, every workgroup then
// ends the same way).  First form of this kernel: sc1 vectors + drain + arrival counter + poll per step, 9-12 us a
// step; with the tagged words 5.5-8 us (N = 2000: 23.9 -> 16.1 ms per matrix).
// Conventions of d, e, tau and the reflectors are LAPACK dsytd2's (UPLO = 'L'), as in eigh_tridiag.hip.
#include <atomic>
#include <cstdlib>

#include "nbx_common.h"

namespace {

constexpr int TDG_THREADS = 256;
constexpr unsigned TDG_SPIN_LIMIT = 1u << 22;
constexpr int TDG_XCH = 8 * 2048;  // doubles of hand-over words per matrix (y and pivot row, two of each, 16 bytes a word, N <= 2048)

// A hand-over word: 16 bytes = (value, tag), written by ONE 16-byte sc1 store and read by one 16-byte sc1 load, with
// tag = (launch nonce, step) mixed with the bits of the value -- a reader polls the word itself until the tag fits the
// value it came with: no drain of the writer's stores, no counter, no grid-wide barrier between a product and its
// readers (two trips to memory instead of five), and a torn or a stale word (an older step of this launch, another
// launch on the same workspace) simply does not verify.  (16-byte sc1 accesses are observed untorn on gfx950 --
// MI355X_MICROARCH.md, visibility -- but nothing here relies on that.)
typedef double tdg_d2 __attribute__((ext_vector_type(2)));
typedef unsigned long long tdg_u64;
constexpr tdg_u64 TDG_MIX = 0x9E3779B97F4A7C15ull;

__device__ __forceinline__ void tdg_put(tdg_d2* p, double v, tdg_u64 key) {
    tdg_d2 w;
    w.x = v;
    w.y = __longlong_as_double((long long)(key ^ (tdg_u64)__double_as_longlong(v)));
    asm volatile("global_store_dwordx4 %0, %1, off sc1" ::"v"(p), "v"(w) : "memory");
}
__device__ __forceinline__ tdg_d2 tdg_get_issue(const tdg_d2* p) {
    tdg_d2 w;
    asm volatile("global_load_dwordx4 %0, %1, off sc1" : "=v"(w) : "v"(p) : "memory");
    return w;
}
__device__ __forceinline__ bool tdg_verified(const tdg_d2& w, tdg_u64 key) {
    return (tdg_u64)__double_as_longlong(w.y) == (key ^ (tdg_u64)__double_as_longlong(w.x));
}

// NM: registers per lane and vector: a wave owns the columns [wq, wq + 1) NP / WPM of its matrix (NP = 64 NM WPM
// >= N), lane l of it the columns c0 + l + 64 m; NBT: matrices reduced side by side by one launch -- with two, waves
// 0-1 work on the first and waves 2-3 on the second.  Sums over a matrix's columns go through LDS (partials of its
// waves added in wave order by every wave: the same bits everywhere).
// a_in (NBT, N, N) lower triangle read; d, e, tau (NBT, N); Vg (NBT, N, N): row k = reflector k in the coordinates of
// the matrix (zeros up to column k, 1 at column k + 1); xch: 4 NBT NP hand-over words (y and pivot-row vectors, two
// of each); nonce: a number no earlier launch on this workspace has used.
template <int NM, int NBT>
__global__ __launch_bounds__(TDG_THREADS, 1) void tdg_kernel(const double* __restrict__ a_in, int N, int R, int P64,
                                                             double* __restrict__ dg, double* __restrict__ eg,
                                                             double* __restrict__ taug, double* __restrict__ Vg,
                                                             double* __restrict__ xch_, unsigned nonce,
                                                             int* __restrict__ status) {
    constexpr int WPM = 4 / NBT;       // waves per matrix
    constexpr int NP = 64 * NM * WPM;  // padded row length
    extern __shared__ __attribute__((aligned(16))) double smem[];
    double* rows = smem;                               // [NBT][R][NP]
    double* red = smem + (size_t)NBT * R * NP;         // [2 parities][4 waves][16] partial sums
    double* mini = red + 2 * 4 * 16;                   // [NBT][2][8]: vp, wp at this workgroup's rows
    int* gflag = reinterpret_cast<int*>(mini + 2 * 16); // "a wave of this workgroup gave up waiting"
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int b = wave / WPM, wq = wave % WPM;         // this wave's matrix, and its place among that matrix's waves
    const int c0 = wq * 64 * NM;                       // first column of this wave
    const int g = blockIdx.x, P = gridDim.x;
    const int64_t n2 = (int64_t)N * N;
    a_in += b * n2;
    dg += (size_t)b * N;
    eg += (size_t)b * N;
    taug += (size_t)b * N;
    Vg += b * n2;
    tdg_d2* xch = reinterpret_cast<tdg_d2*>(xch_);
    // the last row this workgroup holds: once the reduction has passed it the workgroup has nothing left to give
    int i_max = g + P * (R - 1);
    while (i_max >= N) i_max -= P;
    if (g == 0 && wq == 0)  // (reflector N - 1 does not exist: its row of Vg is zero)
        for (int k = lane; k < N; k += 64) Vg[(int64_t)(N - 1) * N + k] = 0.0;

    // ---- this workgroup's rows into LDS (full rows from the lower triangle), zero padded
    for (int bb = 0; bb < NBT; ++bb)
        for (int q = 0; q < R; ++q) {
            const int i = g + P * q;
            const double* src = a_in + (bb - b) * n2;
            for (int k = tid; k < NP; k += TDG_THREADS) {
                double v = 0.0;
                if (i < N && k < N) v = (i >= k) ? src[(int64_t)i * N + k] : src[(int64_t)k * N + i];
                rows[((size_t)bb * R + q) * NP + k] = v;
            }
        }
    if (tid == 0) *gflag = 0;
    __syncthreads();

    // sum of `nv` values over the waves of this wave's matrix (all four waves call this together)
    int rpar = 0;
    auto wsum = [&](double (&v)[8], int nv) {
        double* rd = red + rpar * 64;
        rpar ^= 1;
#pragma unroll
        for (int x = 0; x < 8; ++x)
            if (x < nv) v[x] = nbx_wave_sum_dpp(v[x]);
        if (lane == 0)
            for (int x = 0; x < nv; ++x) rd[wave * 16 + x] = v[x];
        __syncthreads();
#pragma unroll
        for (int x = 0; x < 8; ++x)
            if (x < nv) {
                double t = 0.0;
#pragma unroll
                for (int w = 0; w < WPM; ++w) t += rd[(b * WPM + w) * 16 + x];
                v[x] = t;
            }
    };

    double vp[NM], wp[NM], vn[NM];
    double taup = 0.0;
#pragma unroll
    for (int m = 0; m < NM; ++m) vp[m] = wp[m] = vn[m] = 0.0;
    double* mn = mini + (size_t)b * 16;

    bool gave_up = false;
    for (int j = 0; j < N - 1; ++j) {
        if (i_max <= j || gave_up) break;  // (uniform over the workgroup)
        tdg_d2* ybuf_w = xch + (size_t)(j & 1) * 2 * NBT * NP;              // this step's products and next pivot row
        const tdg_d2* ybuf_r = xch + (size_t)((j + 1) & 1) * 2 * NBT * NP;  // the step before's
        const tdg_u64 key_r = (((tdg_u64)nonce << 32) | (tdg_u64)j) * TDG_MIX;        // words written in step j - 1
        const tdg_u64 key_w = (((tdg_u64)nonce << 32) | (tdg_u64)(j + 1)) * TDG_MIX;  // words written in this step
        // ---- the vectors (each wave its columns): w of the step before, the pivot row j, the reflector of this step
        double row[NM];
        if (j > 0) {
            double yv[NM];
            // poll the 2 NM words of this lane until every one verifies
            unsigned need = 0;
#pragma unroll
            for (int m = 0; m < NM; ++m) {
                const int k = c0 + lane + 64 * m;
                yv[m] = row[m] = 0.0;
                if (k >= j && k < N) need |= 3u << (2 * m);
            }
            for (unsigned spins = 0;; ++spins) {
                tdg_d2 wy[NM], wr[NM];
#pragma unroll
                for (int m = 0; m < NM; ++m) {
                    const int k = c0 + lane + 64 * m;
                    if (need & (1u << (2 * m))) wy[m] = tdg_get_issue(ybuf_r + (size_t)b * NP + k);
                    if (need & (2u << (2 * m))) wr[m] = tdg_get_issue(ybuf_r + (size_t)(NBT + b) * NP + k);
                }
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
                for (int m = 0; m < NM; ++m) {
                    asm volatile("" : "+v"(wy[m]), "+v"(wr[m]));
                    if ((need & (1u << (2 * m))) && tdg_verified(wy[m], key_r)) {
                        yv[m] = wy[m].x;
                        need &= ~(1u << (2 * m));
                    }
                    if ((need & (2u << (2 * m))) && tdg_verified(wr[m], key_r)) {
                        row[m] = wr[m].x;
                        need &= ~(2u << (2 * m));
                    }
                }
                if (!__any(need != 0)) break;
                if (spins > TDG_SPIN_LIMIT) {  // (a word that never came: say so; every workgroup ends the same way)
                    __hip_atomic_store(status, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    gave_up = true;
                    break;
                }
                __builtin_amdgcn_s_sleep(2);
            }
            if (gave_up) *gflag = 1;
            __syncthreads();
            gave_up = *gflag != 0;
            if (gave_up) break;
            double s1[8] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
#pragma unroll
            for (int m = 0; m < NM; ++m) {
                const int k = c0 + lane + 64 * m;
                yv[m] *= taup;  // p = tau A22 v
                s1[0] = fma(yv[m], vp[m], s1[0]);
                s1[1] += (k == j) ? yv[m] : 0.0;
            }
            wsum(s1, 2);
            const double alpha2 = -0.5 * taup * s1[0];
            const double wpj = s1[1] + alpha2;  // (v of the step before is 1 at index j)
#pragma unroll
            for (int m = 0; m < NM; ++m) {
                wp[m] = fma(alpha2, vp[m], yv[m]);
                row[m] = row[m] - wp[m] - wpj * vp[m];
            }
        } else {
#pragma unroll
            for (int m = 0; m < NM; ++m) {
                const int k = c0 + lane + 64 * m;
                row[m] = k < N ? a_in[(int64_t)k * N] : 0.0;  // row 0 = column 0 of the lower triangle
                wp[m] = 0.0;
            }
        }
        double s2[8] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int m = 0; m < NM; ++m) {
            const int k = c0 + lane + 64 * m;
            s2[0] += (k == j) ? row[m] : 0.0;
            s2[1] += (k == j + 1) ? row[m] : 0.0;
            s2[2] = (k > j + 1) ? fma(row[m], row[m], s2[2]) : s2[2];
        }
        wsum(s2, 3);
        const double dj = s2[0], alpha = s2[1], ss = s2[2];
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
            const int k = c0 + lane + 64 * m;
            vn[m] = (k == j + 1) ? 1.0 : ((k > j + 1 && k < N) ? row[m] * scale : 0.0);
        }
        if (g == (j + 1) % P && wq == 0 && lane == 0) {  // (the owner of row j + 1: a workgroup that is still in the loop)
            dg[j] = dj;
            eg[j] = beta;
            taug[j] = tk;
        }
        if (g == (j + 1) % P) {
#pragma unroll
            for (int m = 0; m < NM; ++m) {
                const int k = c0 + lane + 64 * m;
                if (k < N) Vg[(int64_t)j * N + k] = vn[m];
            }
        }
        // the entries of vp, wp at this workgroup's rows i = g + P q, for the row pass: written by whoever holds them
#pragma unroll
        for (int m = 0; m < NM; ++m) {
            const int k = c0 + lane + 64 * m;
            const int kq = k - g;
            if (kq >= 0 && (kq & (P - 1)) == 0 && (kq / P) < 8) {  // (P is a power of two)
                mn[kq / P] = vp[m];
                mn[8 + kq / P] = wp[m];
            }
        }
        __syncthreads();
        // ---- the workgroup's rows (each wave its columns): update with the reflector of the step before, product with the new one
        double ys[8] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            const int i = g + P * q;
            if (q >= R || i <= j || i >= N) continue;  // (uniform) the row has left the trailing block
            const double vpi = mn[q], wpi = mn[8 + q];
            double* rw = rows + ((size_t)b * R + q) * NP;
            tdg_d2* pub = ybuf_w + (size_t)(NBT + b) * NP;
            const bool publish = i == j + 1;  // the next pivot row, as it is before this step's update
#pragma unroll
            for (int m = 0; m < NM; ++m) {
                const int k = c0 + lane + 64 * m;
                double a = rw[k];
                a -= fma(vpi, wp[m], wpi * vp[m]);
                rw[k] = a;
                ys[q] = fma(a, vn[m], ys[q]);
                if (publish && k < N) tdg_put(pub + k, a, key_w);
            }
        }
        wsum(ys, 8);
        if (wq == 0 && lane < 8) {
            const int i = g + P * lane;
            if (lane < R && i > j && i < N) {
                double yv_ = ys[0];
#pragma unroll
                for (int x = 1; x < 8; ++x) yv_ = lane == x ? ys[x] : yv_;
                tdg_put(ybuf_w + (size_t)b * NP + i, yv_, key_w);
            }
        }
#pragma unroll
        for (int m = 0; m < NM; ++m) vp[m] = vn[m];
        taup = tk;
    }
    // the last diagonal element: its row has every update (the reflector of the last step has tau = 0)
    {
        const int i = N - 1;
        if (g == i % P && wq == 0 && lane == 0) {
            dg[i] = rows[((size_t)b * R + i / P) * NP + i];
            eg[i] = 0.0;
            taug[i] = 0.0;
        }
    }
    (void)P64;
}

// T of the compact-WY form of a block of 64 reflectors (LAPACK dlarft, forward, columnwise): T[i][i] = tau_i,
// T[0:i, i] = -tau_i T[0:i, 0:i] (Y^T Y)[0:i, i].  G = Y^T Y (64 x 64); one wavefront per (block, matrix): lane r owns
// row r of T (upper triangular: its entries left of the diagonal are zero, so the sum over c < i needs no mask), G
// in LDS (column i read as broadcasts).  Rows >= nbk (a short last block) come out zero.
__global__ __launch_bounds__(64) void wy_tfactor_kernel(const double* __restrict__ Gb, const double* __restrict__ taub,
                                                        int nblk, int nref, int64_t n, double* __restrict__ Tb) {
    __shared__ double gs[64][65];
    __shared__ double tl[64][65];  // tl[c][r] = T[r][c]: column r is lane r's own (no other lane reads it)
    const int r = threadIdx.x;
    const int blk = blockIdx.x % nblk, mat = blockIdx.x / nblk;
    const double* G = Gb + ((int64_t)mat * nblk + blk) * 4096;
    const double* tau = taub + mat * n + blk * 64;
    double* T = Tb + ((int64_t)mat * nblk + blk) * 4096;
    const int nbk = nref - blk * 64 < 64 ? nref - blk * 64 : 64;
    for (int c = 0; c < 64; ++c) {
        gs[c][r] = G[c * 64 + r];
        tl[c][r] = 0.0;
    }
    __syncthreads();
    for (int i = 0; i < 64; ++i) {
        const double ti = i < nbk ? tau[i] : 0.0;
        double s0 = 0.0, s1 = 0.0;
        int c = 0;
        for (; c + 1 < i; c += 2) {
            s0 = fma(tl[c][r], gs[c][i], s0);
            s1 = fma(tl[c + 1][r], gs[c + 1][i], s1);
        }
        if (c < i) s0 = fma(tl[c][r], gs[c][i], s0);
        tl[i][r] = r < i ? -ti * (s0 + s1) : (r == i ? ti : 0.0);
    }
    for (int c = 0; c < 64; ++c) T[r * 64 + c] = tl[c][r];
}

}  // namespace

// doubles of workspace the two entry points below need beside the caller's matrices
size_t nbx_tdg_work_doubles(int64_t n, int64_t batch) {
    const int64_t nblk = (n + 63) / 64;
    return (size_t)(TDG_XCH * batch + 64 /* status */ + batch * (2 * nblk * 4096 + 2 * 64 * n));
}

bool nbx_tdg_covers(int64_t n) {
    static const bool on = getenv("NBX_TRIDIAG_GRID") == nullptr || atoi(getenv("NBX_TRIDIAG_GRID")) != 0;
    return on && n > 64 && n <= 2048;
}

// Householder reduction of `batch` symmetric matrices (lower triangles of d_a): d, e, tau (batch, n), Vg (batch, n, n)
// with row k = reflector k in matrix coordinates.  work: nbx_tdg_work_doubles(n, batch) doubles.  status: a device
// word of the caller's (64 bytes), zeroed here and set to 1 by a launch in which a hand-over never arrived.
int nbx_tdg_tridiag(nbx_ctx* ctx, int64_t n, int64_t batch, const double* d_a, double* d, double* e, double* tau, double* Vg,
                    double* work, int* status) {
    const int N = (int)n;
    NBX_CHECK_ARG(N > 64 && N <= 2048);
    // workgroups: a power of two, at most 256, every one with a row; rows per workgroup at most eight
    int P = 64;
    while (P < 256 && 2 * P <= N) P *= 2;
    const int R = (N + P - 1) / P;
    NBX_CHECK_ARG(R <= 8);
    // matrices side by side in one launch (two waves each) while their rows fit the LDS, else one after the other
    {
        const int rc = nbx_memset(ctx, status, 0, 64);
        if (rc != NBX_OK) return rc;
    }
    const int per = (batch >= 2 && N <= 1024) ? 2 : 1;
    for (int64_t b0 = 0; b0 < batch; b0 += per) {
        const int nbt = (int)((batch - b0) < per ? (batch - b0) : per);
        const int wpm = 4 / nbt;
        const int nm = (N + 64 * wpm - 1) / (64 * wpm);  // registers per lane and vector
        const int NMt = nm <= 2 ? 2 : (nm <= 4 ? 4 : 8);
        const int NP = 64 * NMt * wpm;
        static std::atomic<unsigned> launch_nonce{0};  // (one sequence per process: contexts of several threads draw from it)
        unsigned nonce = ++launch_nonce;
        if (nonce == 1) {  // (first use of a workspace in this process: no word may verify by accident of old bytes)
            const int rc = nbx_memset(ctx, work, 0, (size_t)TDG_XCH * batch * sizeof(double) + 256);
            if (rc != NBX_OK) return rc;
        }
        const size_t lds = ((size_t)nbt * R * NP + 2 * 4 * 16 + 2 * 16 + 2) * sizeof(double);
        NBX_CHECK_ARG(lds <= 160 * 1024 && NP >= N && 8 * nbt * NP <= TDG_XCH * nbt);
        const double* a_ = d_a + b0 * n * n;
        double *d_ = d + b0 * n, *e_ = e + b0 * n, *t_ = tau + b0 * n, *v_ = Vg + b0 * n * n, *x_ = work;
        int N_ = N, R_ = R, p64 = P / 64;
        void* args[] = {(void*)&a_, (void*)&N_, (void*)&R_, (void*)&p64, (void*)&d_, (void*)&e_, (void*)&t_, (void*)&v_,
                        (void*)&x_, (void*)&nonce, (void*)&status};
        const void* fn = nullptr;
        if (nbt == 2) fn = NMt == 2 ? (const void*)&tdg_kernel<2, 2> : (NMt == 4 ? (const void*)&tdg_kernel<4, 2> : (const void*)&tdg_kernel<8, 2>);
        else fn = NMt == 2 ? (const void*)&tdg_kernel<2, 1> : (NMt == 4 ? (const void*)&tdg_kernel<4, 1> : (const void*)&tdg_kernel<8, 1>);
        if (hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess) {
            (void)hipGetLastError();
            nbx_set_error("nbx_tdg_tridiag: the kernel cannot have 160 KB of dynamic LDS (static LDS in it?)");
            return NBX_E_HIP;
        }
        const hipError_t err = hipLaunchCooperativeKernel(fn, dim3((unsigned)P), dim3(TDG_THREADS), args, lds, ctx->stream);
        if (err != hipSuccess) {
            nbx_set_error("nbx_tdg_tridiag: cooperative launch of %d workgroups failed: %s", P, hipGetErrorString(err));
            return NBX_E_HIP;
        }
    }
    return NBX_OK;
}

// Z (batch, n, n) <- Q Z with Q = H_0 H_1 ... H_{n-2} from Vg / tau of nbx_tdg_tridiag, in compact-WY blocks of 64
// reflectors (Q_b = I - Y T Y^T), last block first.  work: the same workspace (its tail).
int nbx_tdg_backtransform(nbx_ctx* ctx, int64_t n, int64_t batch, const double* Vg, const double* tau, double* Z, double* work) {
    const int64_t nb = 64, n2 = n * n;
    const int64_t nref = n - 1;
    const int64_t nblk = (nref + nb - 1) / nb;
    double* base = work + (size_t)TDG_XCH * batch + 64;
    double* G = base;                            // (batch, nblk, nb, nb)
    double* T = G + batch * nblk * nb * nb;      // (batch, nblk, nb, nb)
    double* W1 = T + batch * nblk * nb * nb;     // (batch, nb, n)
    double* W2 = W1 + batch * nb * n;            // (batch, nb, n)
    // G_b = Y_b Y_b^T of every block (rows beyond the last reflector are zero: Vg's row n - 1), then every T_b
    int rc = nbx_memset(ctx, G, 0, (size_t)(batch * nblk * nb * nb) * sizeof(double));
    if (rc != NBX_OK) return rc;
    for (int64_t b = 0; b < batch; ++b) {
        const int64_t full = nref / nb;  // blocks of nb whole reflectors
        if (full > 0) {
            rc = nbx_gemm(ctx, 'N', 'T', nb, nb, n, 1.0, Vg + b * n2, n, nb * n, Vg + b * n2, n, nb * n, 0.0,
                          G + b * nblk * nb * nb, nb, nb * nb, full);
            if (rc != NBX_OK) return rc;
        }
        if (full < nblk) {
            const int64_t k0 = full * nb, nbk = nref - k0;
            rc = nbx_gemm(ctx, 'N', 'T', nbk, nbk, n, 1.0, Vg + b * n2 + k0 * n, n, 0, Vg + b * n2 + k0 * n, n, 0, 0.0,
                          G + (b * nblk + full) * nb * nb, nb, 0, 1);
            if (rc != NBX_OK) return rc;
        }
    }
    hipLaunchKernelGGL(wy_tfactor_kernel, dim3((unsigned)(batch * nblk)), dim3(64), 0, ctx->stream, G, tau, (int)nblk, (int)nref, n, T);
    NBX_LAUNCH_CHECK();
    for (int64_t blk = nblk - 1; blk >= 0; --blk) {
        const int64_t k0 = blk * nb, nbk = (nref - k0) < nb ? (nref - k0) : nb;
        const double* Y = Vg + k0 * n;  // (nbk, n) rows of reflectors, ld n
        rc = nbx_gemm(ctx, 'N', 'N', nbk, n, n, 1.0, Y, n, n2, Z, n, n2, 0.0, W1, n, nb * n, batch);
        if (rc != NBX_OK) return rc;
        rc = nbx_gemm(ctx, 'N', 'N', nbk, n, nbk, 1.0, T + blk * nb * nb, nb, nblk * nb * nb, W1, n, nb * n, 0.0, W2, n, nb * n, batch);
        if (rc != NBX_OK) return rc;
        rc = nbx_gemm(ctx, 'T', 'N', n, n, nbk, -1.0, Y, n, n2, W2, n, nb * n, 1.0, Z, n, n2, batch);
        if (rc != NBX_OK) return rc;
    }
    return NBX_OK;
}
