// libnbx: LDS-resident symmetric Jacobi eigensolver for N <= 196 (the dense-ERI regime's
// latency-critical size; one SCF cycle runs it once for both spins).
//
// Why this shape.  A single workgroup streaming a global-memory matrix is limited by one
// CU's ~64 B/clk L2 port (measured: 11 us per Jacobi step at N = 148, 15 ms per solve).  Here
//   * the symmetric matrix lives in the CU's 160 KB LDS (N(N+1)/2 doubles = 88 KB at N = 148)
//     and is rotated in place by the round-robin parallel Jacobi ordering: every 2x2 block
//     (pair I x pair K, I < K) of a step is owned by one thread, diagonal blocks by the thread
//     that computed the rotation -- no element has two writers, two barriers per step;
//   * storage is by CIRCULANT DIAGONALS of the tournament ring.  Index 0 is fixed, the other
//     R = N-1 indices sit on a ring that turns by one position per step, so the element that a
//     thread's block needs at step t is the one whose two ring positions were (P-t, Q-t) at
//     t = 0: the ring distance d = (Q-P) mod R never changes.  Storing element (a, b) at
//     [min(d, R-d)][a or b] makes the address of every operand of every step
//         base(d) + ((off - t) mod R)
//     -- one decrement per element per step, no index tables, no global loads in the loop --
//     and consecutive lanes (consecutive K) touch consecutive diagonals, i.e. an odd stride of
//     R doubles: conflict-free LDS banks on one branch, 2-way on the mirrored one (a packed
//     row-major triangle gave ~3.5-way conflicts on random addresses);
//   * rotation count / max |tan| are reduced with ballot and shuffles (same-address LDS
//     atomics serialise per lane and cost ~2500 cycles per step);
//   * eigenvector accumulation is DEFERRED: the kernel only records (c, s) per pair and step;
//     a second kernel then applies the whole rotation sequence to the rows of V, one
//     wavefront per row (rows are independent), spread over all CUs.  A warm start simply
//     initialises those rows from the previous eigenvectors.
// Termination: a sweep without rotations, or a sweep whose largest |tan| was < 1e-8
// (quadratic convergence: the next sweep's rotations would be < 1e-16).
#include <algorithm>

#include "jacobi_ring.h"
#include "nbx_common.h"

namespace {

constexpr int JL_THREADS = 1024;
constexpr int JL_MAXR = 5;          // block rounds per thread: m(m-1)/2 <= 5 * 1024  (m <= 98)
constexpr int JL_MAX_SWEEPS = 24;
constexpr int JL_MAX_NP = 196;
constexpr double JL_PAD_VALUE = 1.0e300;

// Address of the element at ring positions (P, Q) as base + offset, offset in [0, R).
// P or Q == -1 denotes the fixed index 0.  Layout: diagonal d (0..m-1) at d*R + a, then the
// row of index 0 at m*R + b, then element (0,0) at m*R + R.
struct ElemRef {
    int base, off;
};
__host__ __device__ inline ElemRef elem_ref(int P, int Q, int m) {
    const int R = 2 * m - 1;
    ElemRef e;
    if (P < 0 && Q < 0) {
        e.base = m * R + R;
        e.off = -1;  // not on the ring: never updated
    } else if (P < 0 || Q < 0) {
        e.base = m * R;
        e.off = P < 0 ? Q : P;
    } else {
        int d = Q - P;
        if (d < 0) d += R;
        if (d <= m - 1) {
            e.base = d * R;
            e.off = P;
        } else {
            e.base = (R - d) * R;
            e.off = Q;
        }
    }
    return e;
}
// static address of element (i, j) of the original matrix (t = 0 positions)
__host__ __device__ inline int elem_addr(int i, int j, int m) {
    const ElemRef e = elem_ref(i == 0 ? -1 : ring_pos_of(i, m), j == 0 ? -1 : ring_pos_of(j, m), m);
    return e.off < 0 ? e.base : e.base + e.off;
}

__device__ __forceinline__ double fast_rcp(double x) {
    double y = __builtin_amdgcn_rcp(x);
    y = fma(y, fma(-x, y, 1.0), y);
    y = fma(y, fma(-x, y, 1.0), y);
    return y;
}
__device__ __forceinline__ double fast_rsqrt(double x) {
    double y = __builtin_amdgcn_rsq(x);
    y = y * fma(-0.5 * x * y, y, 1.5);
    y = y * fma(-0.5 * x * y, y, 1.5);
    return y;
}

// Workgroup barrier that waits for this wave's LDS traffic only: the per-step global stores
// (rotation log) stay in flight across it.  __syncthreads() would add s_waitcnt vmcnt(0) and
// put a global-memory round trip on every Jacobi step.
__device__ __forceinline__ void lds_barrier() {
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

// one ring turn: offset <- (offset - 1) mod R   (offset == -1 marks the fixed element (0,0))
__device__ __forceinline__ int turn(int off, int R) {
    const int o = off - 1;
    return off <= 0 ? (off < 0 ? -1 : R - 1) : o;
}

__global__ __launch_bounds__(JL_THREADS) void eigh_lds_kernel(
    const double* __restrict__ a_in, int N, int NP, int steps, double2* __restrict__ rot,
    int* __restrict__ any_flags, int* __restrict__ nsteps_out, double* __restrict__ w_out,
    int* __restrict__ rank_out, int* __restrict__ status, int64_t rot_stride, int64_t flag_stride,
    const int* __restrict__ gate) {
    extern __shared__ __attribute__((aligned(16))) double smem[];
    if (gate != nullptr && gate[blockIdx.x] > 0) return;  // eigh_refine.hip already delivered this matrix
    const int m = NP / 2;
    const int R = NP - 1;
    const int npk = NP * (NP + 1) / 2;
    const int nblk = m * (m - 1) / 2;
    double* A = smem;
    double2* cs = reinterpret_cast<double2*>(A + ((npk + 1) & ~1));
    double* dg = reinterpret_cast<double*>(cs + m);
    int* rank = reinterpret_cast<int*>(dg + NP);
    int* misc = rank + NP;  // [0] rotations in sweep, [2..3] per-wave max |t| bits, [4..7] per-step counts [parity][wave]

    const int b = blockIdx.x;
    a_in += (int64_t)b * N * N;
    w_out += (int64_t)b * N;
    rank_out += (int64_t)b * NP;
    rot += (int64_t)b * rot_stride;
    any_flags += (int64_t)b * flag_stride;
    const int tid = threadIdx.x;

    // numpy.linalg.eigh reads the LOWER triangle (UPLO='L'): A(i,j), i <= j, := a_in[j][i]
    for (int idx = tid; idx < NP * NP; idx += JL_THREADS) {
        const int i = idx / NP, j = idx - i * NP;
        if (i > j) continue;
        double v = 0.0;
        if (j < N) v = a_in[(int64_t)j * N + i];
        else if (i == j) v = JL_PAD_VALUE;
        A[elem_addr(i, j, m)] = v;
    }
    if (tid < 8) misc[tid] = 0;

    // this thread's blocks (fixed for the whole solve): b_r = tid + r * 1024 -> pair slots (I, K)
    int bI[JL_MAXR], bK[JL_MAXR];
    int base[JL_MAXR][4], off[JL_MAXR][4];
#pragma unroll
    for (int r = 0; r < JL_MAXR; ++r) {
        const int bb = tid + r * JL_THREADS;
        bI[r] = -1;
        bK[r] = 0;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            base[r][e] = 0;
            off[r][e] = 0;
        }
        if (bb < nblk) {
            // invert bb = I*m - I(I+1)/2 + (K - I - 1)
            int I = (int)((2.0 * m - 1.0 - sqrt((2.0 * m - 1.0) * (2.0 * m - 1.0) - 8.0 * bb)) * 0.5);
            while (I > 0 && I * m - I * (I + 1) / 2 > bb) --I;
            while ((I + 1) * m - (I + 1) * (I + 2) / 2 <= bb) ++I;
            const int K = bb - (I * m - I * (I + 1) / 2) + I + 1;
            bI[r] = I;
            bK[r] = K;
            const int pI[2] = {I == 0 ? -1 : ring_pos_top(I), ring_pos_bot(I, m)};
            const int pK[2] = {ring_pos_top(K), ring_pos_bot(K, m)};  // K >= 1
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const ElemRef er = elem_ref(pI[e >> 1], pK[e & 1], m);
                base[r][e] = er.base;
                off[r][e] = er.off;
            }
        }
    }
    // pair owned in phase A (tid < m): a_pp, a_qq, a_pq
    int pbase[3] = {0, 0, 0}, poff[3] = {0, 0, 0};
    if (tid < m) {
        const int P = tid == 0 ? -1 : ring_pos_top(tid), Q = ring_pos_bot(tid, m);
        const ElemRef e0 = elem_ref(P, P, m), e1 = elem_ref(Q, Q, m), e2 = elem_ref(P, Q, m);
        pbase[0] = e0.base; poff[0] = e0.off;
        pbase[1] = e1.base; poff[1] = e1.off;
        pbase[2] = e2.base; poff[2] = e2.off;
    }
    __syncthreads();

    const double eps = 2.220446049250313e-16;
    int sweep = 0;
    bool converged = false;
    float tmax_wave = 0.f;  // largest |tan| this wave rotated by in the current sweep
    for (; sweep < JL_MAX_SWEEPS && !converged; ++sweep) {
        for (int step = 0; step < steps; ++step) {
            const int gstep = sweep * steps + step;
            const int par = gstep & 1;  // the per-step count slots alternate (also across sweeps)

            // ---- phase A: rotation parameters + diagonal blocks (waves 0 and 1: m <= 98 pairs)
            if (tid < 128) {
                double c = 1.0, s = 0.0;
                float tabs = 0.f;
                bool rotated = false;
                if (tid < m) {
                    const int ipp = pbase[0] + (poff[0] < 0 ? 0 : poff[0]);
                    const int iqq = pbase[1] + poff[1];
                    const int ipq = pbase[2] + poff[2];
                    const double app = A[ipp], aqq = A[iqq], apq = A[ipq];
                    const double aa = fabs(apq);
                    // rotate iff |a_pq| > eps sqrt(|a_pp a_qq|), compared in squares (no sqrt)
                    if (aa * aa > eps * eps * fabs(app) * fabs(aqq) && aa > 1.0e-150) {
                        const double theta = 0.5 * (aqq - app) * fast_rcp(apq);
                        const double at = fabs(theta);
                        double t;
                        if (at > 1.0e150) {
                            t = 0.5 * fast_rcp(at);
                        } else {
                            const double z = fma(at, at, 1.0);
                            t = fast_rcp(at + z * fast_rsqrt(z));  // 1 / (|theta| + sqrt(theta^2 + 1))
                        }
                        if (theta < 0.0) t = -t;
                        c = fast_rsqrt(fma(t, t, 1.0));
                        s = t * c;
                        A[ipp] = app - t * apq;
                        A[iqq] = aqq + t * apq;
                        A[ipq] = 0.0;
                        rotated = true;
                        tabs = (float)fabs(t);
                    }
                    cs[tid] = make_double2(c, s);
                    rot[(int64_t)gstep * m + tid] = make_double2(c, s);
#pragma unroll
                    for (int e = 0; e < 3; ++e) poff[e] = turn(poff[e], R);
                }
                const int cnt = __popcll(__ballot(rotated));
#pragma unroll
                for (int o = 32; o > 0; o >>= 1) tabs = fmaxf(tabs, __shfl_xor(tabs, o, 64));
                tmax_wave = fmaxf(tmax_wave, tabs);
                if ((tid & 63) == 0) misc[4 + 2 * par + (tid >> 6)] = cnt;
            }
            lds_barrier();
            const int nr = misc[4 + 2 * par] + misc[4 + 2 * par + 1];
            // ---- phase B: off-diagonal 2x2 blocks  E <- J_I^T E J_K  in place
#pragma unroll
            for (int r = 0; r < JL_MAXR; ++r) {
                if (bI[r] < 0) continue;
                const int i11 = base[r][0] + off[r][0], i12 = base[r][1] + off[r][1];
                const int i21 = base[r][2] + off[r][2], i22 = base[r][3] + off[r][3];
#pragma unroll
                for (int e = 0; e < 4; ++e) off[r][e] = turn(off[r][e], R);
                if (nr == 0) continue;
                const double2 ri = cs[bI[r]], rk = cs[bK[r]];
                if (ri.y == 0.0 && rk.y == 0.0) continue;
                const double e11 = A[i11], e12 = A[i12], e21 = A[i21], e22 = A[i22];
                // rows: J_I^T = [[c,-s],[s,c]]
                const double u11 = ri.x * e11 - ri.y * e21, u12 = ri.x * e12 - ri.y * e22;
                const double u21 = ri.y * e11 + ri.x * e21, u22 = ri.y * e12 + ri.x * e22;
                // cols: J_K = [[c,s],[-s,c]]
                A[i11] = u11 * rk.x - u12 * rk.y;
                A[i12] = u11 * rk.y + u12 * rk.x;
                A[i21] = u21 * rk.x - u22 * rk.y;
                A[i22] = u21 * rk.y + u22 * rk.x;
            }
            if (tid == 0) {
                any_flags[gstep] = nr;
                misc[0] += nr;
            }
            lds_barrier();
        }
        if (tid == 0 || tid == 64) misc[2 + (tid >> 6)] = (int)__float_as_uint(tmax_wave);
        lds_barrier();
        const float tmax = fmaxf(__uint_as_float((unsigned)misc[2]), __uint_as_float((unsigned)misc[3]));
        converged = (misc[0] == 0) || (tmax < 1.0e-8f);
        tmax_wave = 0.f;
        lds_barrier();
        if (tid == 0) misc[0] = 0;
        lds_barrier();
    }
    __syncthreads();

    // eigenvalues = diagonal; rank them ascending (the padded index sorts last)
    for (int i = tid; i < NP; i += JL_THREADS) dg[i] = A[elem_addr(i, i, m)];
    __syncthreads();
    for (int i = tid; i < NP; i += JL_THREADS) {
        const double di = dg[i];
        int rk = 0;
        for (int j = 0; j < NP; ++j) {
            const double dj = dg[j];
            rk += (dj < di || (dj == di && j < i)) ? 1 : 0;
        }
        rank_out[i] = rk;
        if (rk < N) w_out[rk] = di;
    }
    if (tid == 0) {
        nsteps_out[b] = sweep * steps;
        status[b] = converged ? sweep : -sweep;
    }
}

// One wavefront per row of V: apply the recorded rotation sequence  row <- row J_1 J_2 ...
constexpr int AV_GROUP = 4;  // steps fetched per software-pipeline stage

__global__ __launch_bounds__(64) void eigh_apply_rot_kernel(const double* __restrict__ v0, int N, int NP, int steps,
                                                            const double2* __restrict__ rot,
                                                            const int* __restrict__ any_flags,
                                                            const int* __restrict__ nsteps_in,
                                                            const int* __restrict__ rank_in,
                                                            double* __restrict__ v_out, int64_t rot_stride,
                                                            int64_t flag_stride, const int* __restrict__ gate,
                                                            int transpose_out) {
    if (gate != nullptr && gate[blockIdx.y] > 0) return;
    extern __shared__ __attribute__((aligned(16))) double row[];
    const int b = blockIdx.y, r = blockIdx.x, lane = threadIdx.x;
    const int m = NP / 2, R = NP - 1;
    rot += (int64_t)b * rot_stride;
    any_flags += (int64_t)b * flag_stride;
    rank_in += (int64_t)b * NP;
    v_out += (int64_t)b * N * N;
    for (int i = lane; i < NP; i += 64) {
        double v = (i == r) ? 1.0 : 0.0;
        if (v0 != nullptr) v = (i < N) ? v0[(int64_t)b * N * N + (int64_t)r * N + i] : 0.0;
        row[i] = v;
    }
    __syncthreads();
    const int nsteps = nsteps_in[b];
    const int k0 = lane, k1 = lane + 64;  // m <= 98: two rounds cover every pair
    // ring positions of this lane's two pair slots (top, bot); -1 = the fixed index 0
    const int pt0 = k0 == 0 ? -1 : ring_pos_top(k0), pb0 = ring_pos_bot(k0, m);
    const int pt1 = ring_pos_top(k1), pb1 = ring_pos_bot(k1, m);
    auto index_at = [&](int pos, int t) {  // original index at ring position `pos` after t turns
        if (pos < 0) return 0;
        int q = pos - t;
        if (q < 0) q += R;
        return ring_index0(q, m);
    };

    double2 cs[AV_GROUP][2];
    int flag[AV_GROUP];
    auto fetch = [&](int g0) {
#pragma unroll
        for (int u = 0; u < AV_GROUP; ++u) {
            const int gs = g0 + u;
            flag[u] = 0;
            if (gs < nsteps) flag[u] = any_flags[gs];
            if (flag[u] > 0) {
                if (k0 < m) cs[u][0] = rot[(int64_t)gs * m + k0];
                if (k1 < m) cs[u][1] = rot[(int64_t)gs * m + k1];
            }
        }
    };
    fetch(0);
    for (int g0 = 0; g0 < nsteps; g0 += AV_GROUP) {
        double2 ccs[AV_GROUP][2];
        int cflag[AV_GROUP];
#pragma unroll
        for (int u = 0; u < AV_GROUP; ++u) {
            cflag[u] = flag[u];
            ccs[u][0] = cs[u][0];
            ccs[u][1] = cs[u][1];
        }
        fetch(g0 + AV_GROUP);  // next group in flight while this one is applied
#pragma unroll
        for (int u = 0; u < AV_GROUP; ++u) {
            if (cflag[u] <= 0) continue;  // wave-uniform
            const int t = (g0 + u) % steps;
            if (k0 < m && ccs[u][0].y != 0.0) {
                const int p = index_at(pt0, t), q = index_at(pb0, t);
                const double x = row[p], y = row[q];
                row[p] = ccs[u][0].x * x - ccs[u][0].y * y;
                row[q] = ccs[u][0].y * x + ccs[u][0].x * y;
            }
            if (k1 < m && ccs[u][1].y != 0.0) {
                const int p = index_at(pt1, t), q = index_at(pb1, t);
                const double x = row[p], y = row[q];
                row[p] = ccs[u][1].x * x - ccs[u][1].y * y;
                row[q] = ccs[u][1].y * x + ccs[u][1].x * y;
            }
            // single-wave workgroup: LDS instructions of one wave execute in order, so only
            // the compiler must be kept from reordering across steps (no s_waitcnt vmcnt(0),
            // which would stall on the prefetched next group)
            asm volatile("" ::: "memory");
        }
    }
    __syncthreads();
    for (int i = lane; i < NP; i += 64) {
        const int rk = rank_in[i];
        if (rk < N) v_out[transpose_out ? (int64_t)rk * N + r : (int64_t)r * N + rk] = row[i];
    }
}

size_t align256(size_t x) { return (x + 255) & ~(size_t)255; }

struct LdsLayout {
    size_t rot_off, flag_off, nsteps_off, rank_off, status_off, skip_off, a0_off, tmp_off, refine_off, total;
    int64_t rot_stride, flag_stride;
};

LdsLayout layout(int64_t n, int64_t batch) {
    const int64_t np = (n + 1) & ~1ll, m = np / 2;
    const int64_t steps = (m == 1) ? 1 : np - 1;
    LdsLayout L;
    L.rot_stride = (int64_t)JL_MAX_SWEEPS * steps * m;
    L.flag_stride = (int64_t)JL_MAX_SWEEPS * steps;
    size_t off = 0;
    L.rot_off = off; off += align256((size_t)(L.rot_stride * batch) * sizeof(double2));
    L.flag_off = off; off += align256((size_t)(L.flag_stride * batch) * sizeof(int));
    L.nsteps_off = off; off += align256((size_t)batch * sizeof(int));
    L.rank_off = off; off += align256((size_t)(batch * np) * sizeof(int));
    L.status_off = off; off += align256((size_t)batch * sizeof(int));
    L.skip_off = off; off += align256((size_t)batch * sizeof(int));
    L.a0_off = off; off += align256((size_t)(batch * n * n) * sizeof(double));
    L.tmp_off = off; off += align256((size_t)(batch * n * n) * sizeof(double));
    L.refine_off = off;
    if (nbx_eigh_refine_supported(n, batch)) off += nbx_eigh_refine_worksize(n, batch);
    L.total = off;
    return L;
}

}  // namespace

bool nbx_eigh_lds_supported(int64_t n) { return ((n + 1) & ~1ll) <= JL_MAX_NP; }

size_t nbx_eigh_lds_worksize(int64_t n, int64_t batch) { return layout(n, batch).total; }

// d_v0 == nullptr: cold start.  Otherwise Jacobi runs on V0^T A V0 and accumulates onto V0.
// d_skip (cold starts only): device int[batch]; matrices with d_skip[b] > 0 were already delivered by a solver
// queued ahead (nbx_eigh_tridiag_dev) and are left alone -- d_skip may be this solver's own status words.
int nbx_eigh_lds(nbx_ctx* ctx, int64_t n, int64_t batch, const double* d_a, const double* d_v0, double* d_w,
                 double* d_v, void* d_work, size_t work_bytes, int refine_iters, const int* d_skip) {
    const LdsLayout L = layout(n, batch);
    if (d_work == nullptr || work_bytes < L.total) {
        nbx_set_error("nbx_eigh: workspace %zu < %zu bytes", work_bytes, L.total);
        return NBX_E_NOMEM;
    }
    const int N = (int)n, NP = (int)((n + 1) & ~1ll), m = NP / 2;
    const int steps = (m == 1) ? 1 : NP - 1;
    char* base = static_cast<char*>(d_work);
    double2* rot = reinterpret_cast<double2*>(base + L.rot_off);
    int* flags = reinterpret_cast<int*>(base + L.flag_off);
    int* nsteps = reinterpret_cast<int*>(base + L.nsteps_off);
    int* rank = reinterpret_cast<int*>(base + L.rank_off);
    int* status = reinterpret_cast<int*>(base + L.status_off);
    const double* a_use = d_a;
    const int* gate = (d_v0 == nullptr) ? d_skip : nullptr;
    if (d_v0 != nullptr) {
        double* a0 = reinterpret_cast<double*>(base + L.a0_off);
        double* tmp = reinterpret_cast<double*>(base + L.tmp_off);
        int rc;
        if (nbx_eigh_refine_supported(n, batch) && refine_iters > 0) {
            // GEMM-only refinement of (V0, w) first; Jacobi below runs only for the matrices whose
            // status word says the refinement did not get there (decided on the device)
            nbx_prof_scope prof(ctx, NBX_PROF_EIGH);
            rc = nbx_eigh_refine(ctx, n, batch, d_a, d_v0, d_w, d_v, base + L.refine_off, status, &gate,
                                 refine_iters < NBX_EIGH_REFINE_MAX ? refine_iters : NBX_EIGH_REFINE_MAX);
            if (rc != NBX_OK) return rc;
            rc = nbx_gemm_gated(ctx, 'T', 'N', n, n, n, 1.0, d_v0, n, n * n, d_a, n, n * n, 0.0, tmp, n, n * n, batch, gate,
                                0, -1);
            if (rc != NBX_OK) return rc;
            rc = nbx_gemm_gated(ctx, 'N', 'N', n, n, n, 1.0, tmp, n, n * n, d_v0, n, n * n, 0.0, a0, n, n * n, batch, gate,
                                0, -1);
            if (rc != NBX_OK) return rc;
        } else {
            rc = nbx_gemm(ctx, 'T', 'N', n, n, n, 1.0, d_v0, n, n * n, d_a, n, n * n, 0.0, tmp, n, n * n, batch);
            if (rc != NBX_OK) return rc;
            rc = nbx_gemm(ctx, 'N', 'N', n, n, n, 1.0, tmp, n, n * n, d_v0, n, n * n, 0.0, a0, n, n * n, batch);
            if (rc != NBX_OK) return rc;
        }
        a_use = a0;
    }
    const int npk = NP * (NP + 1) / 2;
    const size_t lds = (size_t)((npk + 1) & ~1) * sizeof(double) + (size_t)m * sizeof(double2) +
                       (size_t)NP * sizeof(double) + (size_t)(NP + 8) * sizeof(int);
    if (lds > 160 * 1024) {
        nbx_set_error("nbx_eigh: N=%d does not fit the LDS solver", N);
        return NBX_E_UNSUPPORTED;
    }
    static bool attr_set = false;
    if (!attr_set) {
        NBX_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(eigh_lds_kernel),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        attr_set = true;
    }
    {
        nbx_prof_scope prof(ctx, NBX_PROF_EIGH);
        hipLaunchKernelGGL(eigh_lds_kernel, dim3((unsigned)batch), dim3(JL_THREADS), lds, ctx->stream, a_use, N, NP,
                           steps, rot, flags, nsteps, d_w, rank, status, L.rot_stride, L.flag_stride, gate);
        NBX_LAUNCH_CHECK();
        hipLaunchKernelGGL(eigh_apply_rot_kernel, dim3((unsigned)N, (unsigned)batch), dim3(64),
                           (size_t)NP * sizeof(double), ctx->stream, d_v0, N, NP, steps, rot, flags, nsteps, rank, d_v,
                           L.rot_stride, L.flag_stride, gate, 0);
        NBX_LAUNCH_CHECK();
    }
    return NBX_OK;
}

size_t nbx_eigh_lds_status_offset(int64_t n, int64_t batch) { return layout(n, batch).status_off; }

// batch ints beside the status words for a caller's "already delivered" flags (nbx_eigh_lds d_skip)
int* nbx_eigh_lds_skip_ptr(int64_t n, int64_t batch, void* d_work) {
    return reinterpret_cast<int*>(static_cast<char*>(d_work) + layout(n, batch).skip_off);
}

const int* nbx_eigh_lds_status_ptr(int64_t n, int64_t batch, const void* d_work) {
    return reinterpret_cast<const int*>(static_cast<const char*>(d_work) + layout(n, batch).status_off);
}

// V^T (rows = accumulated columns, reordered by rank) of the identity under a recorded rotation
// sequence: the deferred right-vector accumulation of the LDS one-sided Jacobi SVD (svd.hip).
int nbx_apply_rotation_log_t(nbx_ctx* ctx, int n, int np_even, int steps, const void* d_rot, const int* d_flags,
                             const int* d_nsteps, const int* d_rank, double* d_vt) {
    hipLaunchKernelGGL(eigh_apply_rot_kernel, dim3((unsigned)n, 1u), dim3(64), (size_t)np_even * sizeof(double),
                       ctx->stream, (const double*)nullptr, n, np_even, steps, static_cast<const double2*>(d_rot), d_flags,
                       d_nsteps, d_rank, d_vt, (int64_t)0, (int64_t)0, (const int*)nullptr, 1);
    NBX_LAUNCH_CHECK();
    return NBX_OK;
}
