// libnbx: dense J/K contraction (HBM-streaming), include/nbx.h "J/K contraction".
//
//   J_pq  = sum_rs (pq|rs) Dtot_rs          Dtot = sum_x D_x
//   K^x_pb = sum_q sum_a (pq|ab) D^x_qa     (uses (pq|ab) = (pq|ba))
//
// Data layout in HBM: the ERI slab is the plain C-order (np,N,N,N) tensor, so the
// (a,b) tile of a fixed (p,q) is N*N contiguous doubles.  Persistent workgroups each own
// an equal contiguous range of tiles; a tile is streamed exactly once with 16-byte
// non-temporal loads that are
// contiguous across the workgroup (thread t always sees the same column pair b, so
// the K accumulators live in registers and need no cross-lane reduction), three
// FMAs per loaded double.  Algorithmic traffic: 8*N^4 bytes per J/K build
// (SURVEY.md section 8d); 0.75 flop/byte => HBM-bound.
//
// Dtot (N*N doubles, read by every workgroup, served from L2) is loaded once per
// group of QB tiles; the D rows the K update needs (QB rows per spin) sit in LDS.
#include "nbx_common.h"
#include "synth_device.h"

namespace {

constexpr int JK_THREADS = 256;
constexpr int JK_QB = 4;

template <bool VEC2>
struct ColVec;
template <>
struct ColVec<true> {
    using type = double2;
    static constexpr int W = 2;
};
template <>
struct ColVec<false> {
    using type = double;
    static constexpr int W = 1;
};

__device__ __forceinline__ double2 ld(const double2* p) { return *p; }
__device__ __forceinline__ double ld(const double* p) { return *p; }
__device__ __forceinline__ void fma_acc(double2& acc, double s, double2 v) {
    acc.x = fma(s, v.x, acc.x);
    acc.y = fma(s, v.y, acc.y);
}
__device__ __forceinline__ void fma_acc(double& acc, double s, double v) { acc = fma(s, v, acc); }
__device__ __forceinline__ double dot_acc(double acc, double2 a, double2 b) {
    return fma(a.y, b.y, fma(a.x, b.x, acc));
}
__device__ __forceinline__ double dot_acc(double acc, double a, double b) { return fma(a, b, acc); }
__device__ __forceinline__ void zero(double2& v) { v.x = 0.0; v.y = 0.0; }
__device__ __forceinline__ void zero(double& v) { v = 0.0; }

// Where the (a, b) tile values of (pq|ab) come from: the dense slab in HBM, or the counter hash
// of SURVEY.md section 8d evaluated in registers (N_AO = 2000: a dense tensor would be 128 TB).
struct TileLoad {
    const double* tile0;  // &eri[p_local][q][0][0]
    int64_t n2;
    __device__ __forceinline__ double2 get2(int j, int64_t o, int, int) const {
        typedef double nbx_d2 __attribute__((ext_vector_type(2)));
        const nbx_d2 t = __builtin_nontemporal_load(reinterpret_cast<const nbx_d2*>(tile0 + j * n2 + o));
        return make_double2(t.x, t.y);
    }
    __device__ __forceinline__ double get1(int j, int64_t o, int, int) const { return tile0[j * n2 + o]; }
};
struct TileGen {
    uint64_t pq[JK_QB];  // tri(p, q + j)
    uint64_t seed;
    double scale;
    __device__ __forceinline__ double val(int j, int a, int b) const {
        return nbx_synth_val(0, nbx_tri_u32((uint32_t)pq[j], nbx_tri_pair_u32((uint32_t)a, (uint32_t)b)), seed) * scale;
    }
    __device__ __forceinline__ double2 get2(int j, int64_t, int a, int b) const {
        return make_double2(val(j, a, b), val(j, a, b + 1));
    }
    __device__ __forceinline__ double get1(int j, int64_t, int a, int b) const { return val(j, a, b); }
};
template <class SRC>
__device__ __forceinline__ double2 src_get(const SRC& s, int j, int64_t o, int a, int b, double2*) { return s.get2(j, o, a, b); }
template <class SRC>
__device__ __forceinline__ double src_get(const SRC& s, int j, int64_t o, int a, int b, double*) { return s.get1(j, o, a, b); }

// Stream QB tiles (q .. q+QB-1 of row p) and accumulate.
//   tile0 : &eri[p_local][q][0][0]
//   dsh   : LDS, dsh[(x*JK_QB + j)*N + a] = D^x[q+j][a]
template <int QB, int NDM, int CS, bool VEC2, class SRC>
__device__ __forceinline__ void jk_group(const SRC& src, const double* __restrict__ dtot,
                                         const double* dsh, int N, int row0, int rstep, int cx,
                                         int cstep, int CX, double (&jacc)[JK_QB],
                                         typename ColVec<VEC2>::type (&kacc)[NDM][CS]) {
    using V = typename ColVec<VEC2>::type;
    constexpr int W = ColVec<VEC2>::W;
#pragma unroll
    for (int seg = 0; seg < CS; ++seg) {
        const int c = cx + seg * cstep;
        if (c >= CX) break;
        const int col = c * W;
        int a = row0;
        // two rows per trip: (QB+1)*2 independent 16-byte loads in flight per thread
        for (; a + rstep < N; a += 2 * rstep) {
            const int64_t o0 = (int64_t)a * N + col;
            const int64_t o1 = o0 + (int64_t)rstep * N;
            V t0[QB], t1[QB];
#pragma unroll
            for (int j = 0; j < QB; ++j) {
                t0[j] = src_get(src, j, o0, a, col, (V*)nullptr);
                t1[j] = src_get(src, j, o1, a + rstep, col, (V*)nullptr);
            }
            const V d0 = ld(reinterpret_cast<const V*>(dtot + o0));
            const V d1 = ld(reinterpret_cast<const V*>(dtot + o1));
#pragma unroll
            for (int j = 0; j < QB; ++j) {
                jacc[j] = dot_acc(jacc[j], t0[j], d0);
                jacc[j] = dot_acc(jacc[j], t1[j], d1);
#pragma unroll
                for (int x = 0; x < NDM; ++x) {
                    fma_acc(kacc[x][seg], dsh[(x * JK_QB + j) * N + a], t0[j]);
                    fma_acc(kacc[x][seg], dsh[(x * JK_QB + j) * N + a + rstep], t1[j]);
                }
            }
        }
        if (a < N) {
            const int64_t o0 = (int64_t)a * N + col;
            V t0[QB];
#pragma unroll
            for (int j = 0; j < QB; ++j) t0[j] = src_get(src, j, o0, a, col, (V*)nullptr);
            const V d0 = ld(reinterpret_cast<const V*>(dtot + o0));
#pragma unroll
            for (int j = 0; j < QB; ++j) {
                jacc[j] = dot_acc(jacc[j], t0[j], d0);
#pragma unroll
                for (int x = 0; x < NDM; ++x) fma_acc(kacc[x][seg], dsh[(x * JK_QB + j) * N + a], t0[j]);
            }
        }
    }
}

// One persistent workgroup per occupancy slot of the chip.  The np*N tiles of the slab are numbered
// t = p_local*N + q and split into equal contiguous ranges of L tiles, one range per workgroup, so
// every slot streams the same number of bytes and they all finish together (a grid of (p, q-chunk)
// workgroups loses up to 30 % to the partially filled last round: 1924 workgroups on 1024 slots).
// A range walks q inside one p in groups of up to QB tiles; when p changes (and at the end of the
// range) the K accumulators are written to the workgroup's next partial slot:
//   kpart[(w*S + (p - p_first(w)))*NDM + x][b],  p_first(w) = (w*L) / N.
template <int NDM, int CS, bool VEC2, bool GEN>
__global__ __launch_bounds__(JK_THREADS) __attribute__((amdgpu_waves_per_eu(CS == 4 ? 2 : 4)))
void jk_dense_kernel(
    const double* __restrict__ eri, const double* __restrict__ dm, const double* __restrict__ dtot,
    double* __restrict__ jout, double* __restrict__ kpart, int N, int64_t ntiles, int L, int S, int p0,
    uint64_t seed) {
    using V = typename ColVec<VEC2>::type;
    constexpr int W = ColVec<VEC2>::W;
    extern __shared__ __attribute__((aligned(16))) double smem[];
    // smem: dsh[NDM*QB*N] | kred[R*NDM*N (CS==1)] | red[17]
    double* dsh = smem;
    double* kred = dsh + NDM * JK_QB * N;

    int64_t t = (int64_t)blockIdx.x * L;
    const int64_t t_end = min(ntiles, t + L);
    if (t >= t_end) return;  // uniform for the whole workgroup
    const int p_first = (int)(t / N);

    const int CX = (N + W - 1) / W;  // column groups per row
    int R, rowg, cx, cstep;
    bool active;
    if (CS == 1) {
        R = max(1, JK_THREADS / CX);
        rowg = threadIdx.x / CX;
        cx = threadIdx.x - rowg * CX;
        active = rowg < R;
        cstep = 0;
    } else {
        R = 1;
        rowg = 0;
        cx = threadIdx.x;
        active = true;
        cstep = JK_THREADS;
    }
    // `red` (block-reduction scratch) sits after kred; CS > 1 needs no kred at all
    double* red = (CS == 1) ? kred + (size_t)R * NDM * N : kred;

    V kacc[NDM][CS];
#pragma unroll
    for (int x = 0; x < NDM; ++x)
#pragma unroll
        for (int s = 0; s < CS; ++s) zero(kacc[x][s]);

    const int64_t n2 = (int64_t)N * N;

    // K partials of the p just finished: reduce the R row groups through LDS, store, clear.
    auto flush = [&](int p_local) {
        double* kout = kpart + ((int64_t)blockIdx.x * S + (p_local - p_first)) * NDM * N;
        if (CS == 1) {
            __syncthreads();
            if (active) {
#pragma unroll
                for (int x = 0; x < NDM; ++x) {
                    double* dst = kred + ((size_t)rowg * NDM + x) * N + cx * W;
                    *reinterpret_cast<V*>(dst) = kacc[x][0];
                }
            }
            __syncthreads();
            for (int i = threadIdx.x; i < NDM * N; i += JK_THREADS) {
                const int x = i / N;
                const int b = i - x * N;
                double tot = 0.0;
                for (int g = 0; g < R; ++g) tot += kred[((size_t)g * NDM + x) * N + b];
                kout[i] = tot;
            }
        } else {
#pragma unroll
            for (int x = 0; x < NDM; ++x)
#pragma unroll
                for (int s = 0; s < CS; ++s) {
                    const int c = cx + s * cstep;
                    if (c < CX) *reinterpret_cast<V*>(kout + (int64_t)x * N + c * W) = kacc[x][s];
                }
        }
#pragma unroll
        for (int x = 0; x < NDM; ++x)
#pragma unroll
            for (int s = 0; s < CS; ++s) zero(kacc[x][s]);
    };

    int p_cur = p_first;
    while (t < t_end) {
        const int p_local = (int)(t / N);
        const int q = (int)(t - (int64_t)p_local * N);
        if (p_local != p_cur) {
            flush(p_cur);
            p_cur = p_local;
        }
        const int nq = (int)min((int64_t)min(JK_QB, N - q), t_end - t);
        __syncthreads();  // previous group's dsh (and kred) reads are done
        for (int i = threadIdx.x; i < NDM * JK_QB * N; i += JK_THREADS) {
            const int x = i / (JK_QB * N);
            const int rem = i - x * JK_QB * N;
            const int j = rem / N;
            const int a = rem - j * N;
            dsh[i] = (j < nq) ? dm[(int64_t)x * n2 + (int64_t)(q + j) * N + a] : 0.0;
        }
        __syncthreads();
        double jacc[JK_QB] = {0.0, 0.0, 0.0, 0.0};
        if (active) {
            if constexpr (GEN) {
                TileGen src;
                src.seed = seed;
                src.scale = 1.0 / (double)N;
                const uint64_t pg = (uint64_t)(p0 + p_local);
#pragma unroll
                for (int j = 0; j < JK_QB; ++j) src.pq[j] = nbx_tri(pg, (uint64_t)min(q + j, N - 1));
                switch (nq) {
                    case 4: jk_group<4, NDM, CS, VEC2>(src, dtot, dsh, N, rowg, R, cx, cstep, CX, jacc, kacc); break;
                    case 3: jk_group<3, NDM, CS, VEC2>(src, dtot, dsh, N, rowg, R, cx, cstep, CX, jacc, kacc); break;
                    case 2: jk_group<2, NDM, CS, VEC2>(src, dtot, dsh, N, rowg, R, cx, cstep, CX, jacc, kacc); break;
                    default: jk_group<1, NDM, CS, VEC2>(src, dtot, dsh, N, rowg, R, cx, cstep, CX, jacc, kacc); break;
                }
            } else {
                TileLoad src;
                src.n2 = n2;
                src.tile0 = eri + t * n2;
                switch (nq) {
                    case 4: jk_group<4, NDM, CS, VEC2>(src, dtot, dsh, N, rowg, R, cx, cstep, CX, jacc, kacc); break;
                    case 3: jk_group<3, NDM, CS, VEC2>(src, dtot, dsh, N, rowg, R, cx, cstep, CX, jacc, kacc); break;
                    case 2: jk_group<2, NDM, CS, VEC2>(src, dtot, dsh, N, rowg, R, cx, cstep, CX, jacc, kacc); break;
                    default: jk_group<1, NDM, CS, VEC2>(src, dtot, dsh, N, rowg, R, cx, cstep, CX, jacc, kacc); break;
                }
            }
        }
#pragma unroll
        for (int j = 0; j < JK_QB; ++j) {
            const double tot = nbx_block_sum(jacc[j], red);
            if (threadIdx.x == 0 && j < nq) jout[t + j] = tot;
        }
        t += nq;
    }
    flush(p_cur);
}

// dtot = sum_x dm[x]
__global__ void jk_dtot_kernel(const double* __restrict__ dm, double* __restrict__ dtot, int64_t n2, int ndm) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n2) return;
    double t = dm[i];
    for (int x = 1; x < ndm; ++x) t += dm[x * n2 + i];
    dtot[i] = t;
}

// jk[1+x][p][b] = sum over the workgroups w whose tile range meets row p of their partial for p
// (fixed order => bitwise reproducible)
__global__ void jk_reduce_kernel(const double* __restrict__ kpart, double* __restrict__ kout, int N, int np,
                                 int ndm, int L, int S) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (int64_t)ndm * np * N) return;
    const int b = (int)(i % N);
    const int p = (int)((i / N) % np);
    const int x = (int)(i / ((int64_t)N * np));
    const int64_t w_lo = ((int64_t)p * N) / L, w_hi = ((int64_t)(p + 1) * N - 1) / L;
    double t = 0.0;
    for (int64_t w = w_lo; w <= w_hi; ++w) {
        const int slot = p - (int)((w * L) / N);
        t += kpart[((w * S + slot) * ndm + x) * N + b];
    }
    kout[i] = t;
}

// Launch geometry.  Fixed by (N, np, ndm) alone -- not by the device it runs on -- so results are
// bitwise identical everywhere: the slot count is that of the MI355X (256 CUs; 4 workgroups of
// 4 waves per CU at the kernel's ~100 VGPRs, fewer when the D rows in LDS are large).
struct JkPlan {
    int cs, wgs, L, S;
    bool vec2;
    size_t lds_bytes;
};

constexpr int JK_CUS = 256;
constexpr size_t JK_LDS_PER_CU = 160 * 1024;

JkPlan jk_plan(int64_t N, int64_t np, int64_t ndm) {
    JkPlan pl;
    pl.vec2 = (N % 2 == 0);
    const int64_t CX = pl.vec2 ? N / 2 : N;
    pl.cs = CX <= JK_THREADS ? 1 : (CX <= 2 * JK_THREADS ? 2 : 4);
    const int64_t R = pl.cs == 1 ? (JK_THREADS / CX > 0 ? JK_THREADS / CX : 1) : 0;
    pl.lds_bytes = (size_t)(ndm * JK_QB * N + R * ndm * N + 17) * sizeof(double);
    int64_t per_cu = pl.cs == 4 ? 2 : 4;  // matches amdgpu_waves_per_eu of the kernel
    const int64_t by_lds = (int64_t)(JK_LDS_PER_CU / (pl.lds_bytes + 256));
    if (per_cu > by_lds) per_cu = by_lds > 0 ? by_lds : 1;
    const int64_t slots = JK_CUS * per_cu;
    const int64_t ntiles = np * N;
    const int64_t L = nbx_cdiv(ntiles, slots) > 0 ? nbx_cdiv(ntiles, slots) : 1;
    pl.L = (int)L;
    pl.wgs = (int)nbx_cdiv(ntiles, L);
    pl.S = (int)((L + N - 2) / N + 1);
    return pl;
}

}  // namespace

extern "C" size_t nbx_jk_dense_worksize(int64_t nao, int64_t np, int64_t ndm) {
    if (nao <= 0 || np < 0 || ndm <= 0) return 0;
    const JkPlan pl = jk_plan(nao, np, ndm);
    return (size_t)(nao * nao + (int64_t)pl.wgs * pl.S * ndm * nao) * sizeof(double);
}

template <int NDM, int CS, bool VEC2>
static void jk_launch(nbx_ctx* ctx, const JkPlan& pl, const double* eri, const double* dm, const double* dtot,
                      double* jout, double* kpart, int N, int np, int p0, bool gen, uint64_t seed) {
    if (gen)
        hipLaunchKernelGGL((jk_dense_kernel<NDM, CS, VEC2, true>), dim3((unsigned)pl.wgs), dim3(JK_THREADS),
                           pl.lds_bytes, ctx->stream, eri, dm, dtot, jout, kpart, N, (int64_t)np * N, pl.L, pl.S, p0,
                           seed);
    else
        hipLaunchKernelGGL((jk_dense_kernel<NDM, CS, VEC2, false>), dim3((unsigned)pl.wgs), dim3(JK_THREADS),
                           pl.lds_bytes, ctx->stream, eri, dm, dtot, jout, kpart, N, (int64_t)np * N, pl.L, pl.S, p0,
                           seed);
}

static int jk_impl(nbx_ctx* ctx, int64_t nao, int64_t p0, int64_t p1, const double* d_eri, bool gen, uint64_t seed,
                   const double* d_dm, int64_t ndm, double* d_jk, void* d_work, size_t work_bytes) {
    NBX_CHECK_ARG(ctx != nullptr && d_dm != nullptr);
    NBX_CHECK_ARG(nao > 0 && p0 >= 0 && p1 >= p0 && p1 <= nao);
    NBX_CHECK_ARG(p0 == p1 || ((gen || d_eri != nullptr) && d_jk != nullptr));  // an empty slab has no storage
    NBX_CHECK_ARG(ndm == 1 || ndm == 2);
    const int64_t np = p1 - p0;
    if (np == 0) return NBX_OK;
    if (nao > 2048 || (nao % 2 == 1 && nao > 1024)) {
        nbx_set_error("nbx_jk: N=%lld exceeds the kernel limit (2048 even / 1024 odd)", (long long)nao);
        return NBX_E_UNSUPPORTED;
    }
    const JkPlan pl = jk_plan(nao, np, ndm);
    const size_t need = nbx_jk_dense_worksize(nao, np, ndm);
    if (d_work == nullptr || work_bytes < need) {
        nbx_set_error("nbx_jk: workspace %zu < %zu bytes", work_bytes, need);
        return NBX_E_NOMEM;
    }
    NBX_CHECK_ARG((gen || (reinterpret_cast<uintptr_t>(d_eri) & 15) == 0) && (reinterpret_cast<uintptr_t>(d_work) & 15) == 0);
    double* dtot = static_cast<double*>(d_work);
    double* kpart = dtot + nao * nao;
    const int64_t n2 = nao * nao;
    hipLaunchKernelGGL(jk_dtot_kernel, dim3((unsigned)nbx_cdiv(n2, 256)), dim3(256), 0, ctx->stream, d_dm, dtot,
                       n2, (int)ndm);
    const int N = (int)nao;
    {
    nbx_prof_scope prof(ctx, NBX_PROF_JK_DENSE);
#define NBX_JK_CASE(NDM, CS, V2) \
    jk_launch<NDM, CS, V2>(ctx, pl, d_eri, d_dm, dtot, d_jk, kpart, N, (int)np, (int)p0, gen, seed)
    if (ndm == 2) {
        if (pl.vec2) {
            if (pl.cs == 1) NBX_JK_CASE(2, 1, true);
            else if (pl.cs == 2) NBX_JK_CASE(2, 2, true);
            else NBX_JK_CASE(2, 4, true);
        } else {
            if (pl.cs == 1) NBX_JK_CASE(2, 1, false);
            else if (pl.cs == 2) NBX_JK_CASE(2, 2, false);
            else NBX_JK_CASE(2, 4, false);
        }
    } else {
        if (pl.vec2) {
            if (pl.cs == 1) NBX_JK_CASE(1, 1, true);
            else if (pl.cs == 2) NBX_JK_CASE(1, 2, true);
            else NBX_JK_CASE(1, 4, true);
        } else {
            if (pl.cs == 1) NBX_JK_CASE(1, 1, false);
            else if (pl.cs == 2) NBX_JK_CASE(1, 2, false);
            else NBX_JK_CASE(1, 4, false);
        }
    }
#undef NBX_JK_CASE
    }
    NBX_LAUNCH_CHECK();
    hipLaunchKernelGGL(jk_reduce_kernel, dim3((unsigned)nbx_cdiv(ndm * np * nao, 256)), dim3(256), 0, ctx->stream,
                       kpart, d_jk + np * nao, N, (int)np, (int)ndm, pl.L, pl.S);
    NBX_LAUNCH_CHECK();
    return NBX_OK;
}

extern "C" int nbx_jk_dense(nbx_ctx* ctx, int64_t nao, int64_t p0, int64_t p1, const double* d_eri,
                            const double* d_dm, int64_t ndm, double* d_jk, void* d_work,
                            size_t work_bytes) {
    return jk_impl(ctx, nao, p0, p1, d_eri, false, 0, d_dm, ndm, d_jk, d_work, work_bytes);
}

extern "C" int nbx_jk_synth(nbx_ctx* ctx, int64_t nao, int64_t p0, int64_t p1, uint64_t seed, const double* d_dm,
                            int64_t ndm, double* d_jk, void* d_work, size_t work_bytes) {
    return jk_impl(ctx, nao, p0, p1, nullptr, true, seed, d_dm, ndm, d_jk, d_work, work_bytes);
}
