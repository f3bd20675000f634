// libnbx: LDS-resident symmetric Jacobi eigensolver for N <= 196 (the dense-ERI regime's
// latency-critical size; one SCF cycle runs it once for both spins).
//
// Why this shape.  A single workgroup streaming a global-memory matrix is limited by one
// CU's ~64 B/clk L2 port (measured: 11 us per Jacobi step at N = 148, 15 ms per solve).  Here
//   * the symmetric matrix lives PACKED (upper triangle, N(N+1)/2 doubles = 88 KB at N=148)
//     in the CU's 160 KB LDS and is rotated in place: every 2x2 block (pair I x pair K,
//     I < K) of the round-robin step is owned by one thread, diagonal blocks by the thread
//     that computed the rotation -- no element has two writers, two barriers per step;
//   * the index pairs of the round-robin ordering depend only on N, so all LDS addresses of
//     all steps come from a table built once per N (cached in the context, read through L2
//     and prefetched one step ahead) instead of being recomputed with integer arithmetic --
//     the kernel is fp64-FMA-bound on one CU (16 flop per block);
//   * eigenvector accumulation is DEFERRED: the kernel only records (c, s) per pair and step;
//     a second kernel then applies the whole rotation sequence to the rows of V, one
//     wavefront per row (rows are independent), spread over all CUs.  A warm start simply
//     initialises those rows from the previous eigenvectors.
// Termination: a sweep without rotations, or a sweep whose largest |tan| was < 1e-8
// (quadratic convergence: the next sweep's rotations would be < 1e-16).
#include <algorithm>

#include "nbx_common.h"

namespace {

constexpr int JL_THREADS = 1024;
constexpr int JL_MAXR = 5;          // block rounds per thread: m(m-1)/2 <= 5 * 1024  (m <= 98)
constexpr int JL_MAX_SWEEPS = 24;
constexpr int JL_MAX_NP = 196;
constexpr double JL_PAD_VALUE = 1.0e300;

__host__ __device__ inline int pk(int i, int j, int NP) {  // packed upper index, i <= j
    return i * NP - i * (i - 1) / 2 + (j - i);
}
__host__ __device__ inline int pks(int i, int j, int NP) { return i <= j ? pk(i, j, NP) : pk(j, i, NP); }

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
// (rotation log) and the schedule prefetch stay in flight across it.  __syncthreads() would
// add s_waitcnt vmcnt(0) and put a global-memory round trip on every Jacobi step.
__device__ __forceinline__ void lds_barrier() {
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

__global__ __launch_bounds__(JL_THREADS) void eigh_lds_kernel(
    const double* __restrict__ a_in, int N, int NP, int steps, const ushort4* __restrict__ sched_blocks,
    const ushort4* __restrict__ sched_pairs, double2* __restrict__ rot, int* __restrict__ any_flags,
    int* __restrict__ nsteps_out, double* __restrict__ w_out, int* __restrict__ rank_out,
    int* __restrict__ status, int64_t rot_stride, int64_t flag_stride) {
    extern __shared__ __attribute__((aligned(16))) double smem[];
    const int m = NP / 2;
    const int npk = NP * (NP + 1) / 2;
    const int nblk = m * (m - 1) / 2;
    double* A = smem;
    double2* cs = reinterpret_cast<double2*>(A + ((npk + 1) & ~1));
    double* dg = reinterpret_cast<double*>(cs + m);
    int* rank = reinterpret_cast<int*>(dg + NP);
    int* misc = rank + NP;  // [0] rotations in sweep, [1],[2] per-step counters (parity), [3] max |t| bits

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
        A[pk(i, j, NP)] = v;
    }
    if (tid < 4) misc[tid] = 0;

    // this thread's blocks (fixed for the whole solve): b_r = tid + r * 1024 -> positions (I, K)
    int bI[JL_MAXR], bK[JL_MAXR];
#pragma unroll
    for (int r = 0; r < JL_MAXR; ++r) {
        const int bb = tid + r * JL_THREADS;
        bI[r] = -1;
        bK[r] = 0;
        if (bb < nblk) {
            // invert bb = I*m - I(I+1)/2 + (K - I - 1)
            int I = (int)((2.0 * m - 1.0 - sqrt((2.0 * m - 1.0) * (2.0 * m - 1.0) - 8.0 * bb)) * 0.5);
            while (I > 0 && I * m - I * (I + 1) / 2 > bb) --I;
            while ((I + 1) * m - (I + 1) * (I + 2) / 2 <= bb) ++I;
            bI[r] = I;
            bK[r] = bb - (I * m - I * (I + 1) / 2) + I + 1;
        }
    }
    __syncthreads();

    const double eps = 2.220446049250313e-16;
    int sweep = 0;
    bool converged = false;
    // schedule entries are fetched one step ahead (registers), off the critical path
    ushort4 addr_next[JL_MAXR];
    ushort4 pair_next = make_ushort4(0, 0, 0, 0);
#pragma unroll
    for (int r = 0; r < JL_MAXR; ++r) {
        addr_next[r] = make_ushort4(0, 0, 0, 0);
        if (bI[r] >= 0) addr_next[r] = sched_blocks[tid + r * JL_THREADS];
    }
    if (tid < m) pair_next = sched_pairs[tid];
    for (; sweep < JL_MAX_SWEEPS && !converged; ++sweep) {
        for (int step = 0; step < steps; ++step) {
            const int gstep = sweep * steps + step;
            const int par = gstep & 1;  // global parity: the two per-step counters alternate across sweeps too
            ushort4 addr[JL_MAXR];
#pragma unroll
            for (int r = 0; r < JL_MAXR; ++r) addr[r] = addr_next[r];
            const ushort4 pi = pair_next;
            {
                const int nstep = (step + 1 == steps) ? 0 : step + 1;
#pragma unroll
                for (int r = 0; r < JL_MAXR; ++r)
                    if (bI[r] >= 0) addr_next[r] = sched_blocks[(int64_t)nstep * nblk + tid + r * JL_THREADS];
                if (tid < m) pair_next = sched_pairs[nstep * m + tid];
            }

            // ---- phase A: rotation parameters + diagonal blocks
            if (tid == 0) misc[1 + (par ^ 1)] = 0;
            if (tid < m) {
                const double app = A[pi.x], aqq = A[pi.y], apq = A[pi.z];
                double c = 1.0, s = 0.0;
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
                    A[pi.x] = app - t * apq;
                    A[pi.y] = aqq + t * apq;
                    A[pi.z] = 0.0;
                    atomicAdd(&misc[1 + par], 1);
                    atomicMax(reinterpret_cast<unsigned*>(&misc[3]), __float_as_uint((float)fabs(t)));
                }
                cs[tid] = make_double2(c, s);
                rot[(int64_t)gstep * m + tid] = make_double2(c, s);
            }
            lds_barrier();
            const int nr = misc[1 + par];
            // ---- phase B: off-diagonal 2x2 blocks  E <- J_I^T E J_K  in place
            if (nr > 0) {
#pragma unroll
                for (int r = 0; r < JL_MAXR; ++r) {
                    if (bI[r] < 0) continue;
                    const double2 ri = cs[bI[r]], rk = cs[bK[r]];
                    if (ri.y == 0.0 && rk.y == 0.0) continue;
                    const double e11 = A[addr[r].x], e12 = A[addr[r].y], e21 = A[addr[r].z], e22 = A[addr[r].w];
                    // rows: J_I^T = [[c,-s],[s,c]]
                    const double u11 = ri.x * e11 - ri.y * e21, u12 = ri.x * e12 - ri.y * e22;
                    const double u21 = ri.y * e11 + ri.x * e21, u22 = ri.y * e12 + ri.x * e22;
                    // cols: J_K = [[c,s],[-s,c]]
                    A[addr[r].x] = u11 * rk.x - u12 * rk.y;
                    A[addr[r].y] = u11 * rk.y + u12 * rk.x;
                    A[addr[r].z] = u21 * rk.x - u22 * rk.y;
                    A[addr[r].w] = u21 * rk.y + u22 * rk.x;
                }
            }
            if (tid == 0) {
                any_flags[gstep] = nr;
                misc[0] += nr;
            }
            lds_barrier();
        }
        const float tmax = __uint_as_float((unsigned)misc[3]);
        converged = (misc[0] == 0) || (tmax < 1.0e-8f);
        lds_barrier();
        if (tid == 0) {
            misc[0] = 0;
            misc[3] = 0;
        }
        lds_barrier();
    }
    __syncthreads();

    // eigenvalues = diagonal; rank them ascending (the padded index sorts last)
    for (int i = tid; i < NP; i += JL_THREADS) dg[i] = A[pk(i, i, NP)];
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
                                                            const ushort2* __restrict__ sched_pq,
                                                            const double2* __restrict__ rot,
                                                            const int* __restrict__ any_flags,
                                                            const int* __restrict__ nsteps_in,
                                                            const int* __restrict__ rank_in,
                                                            double* __restrict__ v_out, int64_t rot_stride,
                                                            int64_t flag_stride) {
    extern __shared__ __attribute__((aligned(16))) double row[];
    const int b = blockIdx.y, r = blockIdx.x, lane = threadIdx.x;
    const int m = NP / 2;
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

    ushort2 pq[AV_GROUP][2];
    double2 cs[AV_GROUP][2];
    int flag[AV_GROUP];
    auto fetch = [&](int g0) {
#pragma unroll
        for (int u = 0; u < AV_GROUP; ++u) {
            const int gs = g0 + u;
            flag[u] = 0;
            if (gs < nsteps) flag[u] = any_flags[gs];
            if (flag[u] > 0) {
                const int st = gs % steps;
                if (k0 < m) {
                    pq[u][0] = sched_pq[st * m + k0];
                    cs[u][0] = rot[(int64_t)gs * m + k0];
                }
                if (k1 < m) {
                    pq[u][1] = sched_pq[st * m + k1];
                    cs[u][1] = rot[(int64_t)gs * m + k1];
                }
            }
        }
    };
    fetch(0);
    for (int g0 = 0; g0 < nsteps; g0 += AV_GROUP) {
        ushort2 cpq[AV_GROUP][2];
        double2 ccs[AV_GROUP][2];
        int cflag[AV_GROUP];
#pragma unroll
        for (int u = 0; u < AV_GROUP; ++u) {
            cflag[u] = flag[u];
            cpq[u][0] = pq[u][0];
            cpq[u][1] = pq[u][1];
            ccs[u][0] = cs[u][0];
            ccs[u][1] = cs[u][1];
        }
        fetch(g0 + AV_GROUP);  // next group in flight while this one is applied
#pragma unroll
        for (int u = 0; u < AV_GROUP; ++u) {
            if (cflag[u] <= 0) continue;  // wave-uniform
            if (k0 < m && ccs[u][0].y != 0.0) {
                const double x = row[cpq[u][0].x], y = row[cpq[u][0].y];
                row[cpq[u][0].x] = ccs[u][0].x * x - ccs[u][0].y * y;
                row[cpq[u][0].y] = ccs[u][0].y * x + ccs[u][0].x * y;
            }
            if (k1 < m && ccs[u][1].y != 0.0) {
                const double x = row[cpq[u][1].x], y = row[cpq[u][1].y];
                row[cpq[u][1].x] = ccs[u][1].x * x - ccs[u][1].y * y;
                row[cpq[u][1].y] = ccs[u][1].y * x + ccs[u][1].x * y;
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
        if (rk < N) v_out[(int64_t)r * N + rk] = row[i];
    }
}

size_t align256(size_t x) { return (x + 255) & ~(size_t)255; }

struct LdsLayout {
    size_t rot_off, flag_off, nsteps_off, rank_off, status_off, a0_off, tmp_off, total;
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
    L.a0_off = off; off += align256((size_t)(batch * n * n) * sizeof(double));
    L.tmp_off = off; off += align256((size_t)(batch * n * n) * sizeof(double));
    L.total = off;
    return L;
}

// Round-robin schedule tables for a given NP, built on the host once and cached in the context.
nbx_sched* get_sched(nbx_ctx* ctx, int NP) {
    for (auto& s : ctx->sched)
        if (s.np == NP) return &s;
    const int m = NP / 2, nblk = m * (m - 1) / 2, steps = (m == 1) ? 1 : NP - 1;
    std::vector<ushort4> blocks((size_t)steps * std::max(nblk, 1));
    std::vector<ushort4> pairs((size_t)steps * m);
    std::vector<ushort2> pqs((size_t)steps * m);
    std::vector<int> top(m), bot(m), nt(m), nb(m);
    for (int k = 0; k < m; ++k) {
        top[k] = 2 * k;
        bot[k] = 2 * k + 1;
    }
    for (int s = 0; s < steps; ++s) {
        for (int k = 0; k < m; ++k) {
            const int p = top[k], q = bot[k];
            pairs[(size_t)s * m + k] = make_ushort4((unsigned short)pk(p, p, NP), (unsigned short)pk(q, q, NP),
                                                    (unsigned short)pks(p, q, NP), 0);
            pqs[(size_t)s * m + k] = make_ushort2((unsigned short)p, (unsigned short)q);
        }
        int bidx = 0;
        for (int I = 0; I < m; ++I)
            for (int K = I + 1; K < m; ++K, ++bidx)
                blocks[(size_t)s * nblk + bidx] =
                    make_ushort4((unsigned short)pks(top[I], top[K], NP), (unsigned short)pks(top[I], bot[K], NP),
                                 (unsigned short)pks(bot[I], top[K], NP), (unsigned short)pks(bot[I], bot[K], NP));
        if (m >= 2) {
            nt[0] = top[0];
            nt[1] = bot[0];
            for (int k = 2; k < m; ++k) nt[k] = top[k - 1];
            for (int k = 0; k < m - 1; ++k) nb[k] = bot[k + 1];
            nb[m - 1] = top[m - 1];
            top.swap(nt);
            bot.swap(nb);
        }
    }
    nbx_sched sc;
    sc.np = NP;
    sc.d_blocks = sc.d_pairs = sc.d_pq = nullptr;
    if (hipMalloc(&sc.d_blocks, std::max(blocks.size(), (size_t)1) * sizeof(ushort4)) != hipSuccess) return nullptr;
    if (hipMalloc(&sc.d_pairs, pairs.size() * sizeof(ushort4)) != hipSuccess) return nullptr;
    if (hipMalloc(&sc.d_pq, pqs.size() * sizeof(ushort2)) != hipSuccess) return nullptr;
    (void)hipMemcpy(sc.d_blocks, blocks.data(), blocks.size() * sizeof(ushort4), hipMemcpyHostToDevice);
    (void)hipMemcpy(sc.d_pairs, pairs.data(), pairs.size() * sizeof(ushort4), hipMemcpyHostToDevice);
    (void)hipMemcpy(sc.d_pq, pqs.data(), pqs.size() * sizeof(ushort2), hipMemcpyHostToDevice);
    ctx->sched.push_back(sc);
    return &ctx->sched.back();
}

}  // namespace

bool nbx_eigh_lds_supported(int64_t n) { return ((n + 1) & ~1ll) <= JL_MAX_NP; }

size_t nbx_eigh_lds_worksize(int64_t n, int64_t batch) { return layout(n, batch).total; }

// d_v0 == nullptr: cold start.  Otherwise Jacobi runs on V0^T A V0 and accumulates onto V0.
int nbx_eigh_lds(nbx_ctx* ctx, int64_t n, int64_t batch, const double* d_a, const double* d_v0, double* d_w,
                 double* d_v, void* d_work, size_t work_bytes) {
    const LdsLayout L = layout(n, batch);
    if (d_work == nullptr || work_bytes < L.total) {
        nbx_set_error("nbx_eigh: workspace %zu < %zu bytes", work_bytes, L.total);
        return NBX_E_NOMEM;
    }
    const int N = (int)n, NP = (int)((n + 1) & ~1ll), m = NP / 2;
    const int steps = (m == 1) ? 1 : NP - 1;
    nbx_sched* sc = get_sched(ctx, NP);
    if (sc == nullptr) {
        nbx_set_error("nbx_eigh: could not build the rotation schedule for N=%d", N);
        return NBX_E_NOMEM;
    }
    char* base = static_cast<char*>(d_work);
    double2* rot = reinterpret_cast<double2*>(base + L.rot_off);
    int* flags = reinterpret_cast<int*>(base + L.flag_off);
    int* nsteps = reinterpret_cast<int*>(base + L.nsteps_off);
    int* rank = reinterpret_cast<int*>(base + L.rank_off);
    int* status = reinterpret_cast<int*>(base + L.status_off);
    const double* a_use = d_a;
    if (d_v0 != nullptr) {
        double* a0 = reinterpret_cast<double*>(base + L.a0_off);
        double* tmp = reinterpret_cast<double*>(base + L.tmp_off);
        int rc = nbx_gemm(ctx, 'T', 'N', n, n, n, 1.0, d_v0, n, n * n, d_a, n, n * n, 0.0, tmp, n, n * n, batch);
        if (rc != NBX_OK) return rc;
        rc = nbx_gemm(ctx, 'N', 'N', n, n, n, 1.0, tmp, n, n * n, d_v0, n, n * n, 0.0, a0, n, n * n, batch);
        if (rc != NBX_OK) return rc;
        a_use = a0;
    }
    const int npk = NP * (NP + 1) / 2;
    const size_t lds = (size_t)((npk + 1) & ~1) * sizeof(double) + (size_t)m * sizeof(double2) +
                       (size_t)NP * sizeof(double) + (size_t)(NP + 4) * sizeof(int);
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
                           steps, static_cast<const ushort4*>(sc->d_blocks), static_cast<const ushort4*>(sc->d_pairs),
                           rot, flags, nsteps, d_w, rank, status, L.rot_stride, L.flag_stride);
        NBX_LAUNCH_CHECK();
        hipLaunchKernelGGL(eigh_apply_rot_kernel, dim3((unsigned)N, (unsigned)batch), dim3(64),
                           (size_t)NP * sizeof(double), ctx->stream, d_v0, N, NP, steps,
                           static_cast<const ushort2*>(sc->d_pq), rot, flags, nsteps, rank, d_v, L.rot_stride,
                           L.flag_stride);
        NBX_LAUNCH_CHECK();
    }
    return NBX_OK;
}

const int* nbx_eigh_lds_status_ptr(int64_t n, int64_t batch, const void* d_work) {
    return reinterpret_cast<const int*>(static_cast<const char*>(d_work) + layout(n, batch).status_off);
}
