// libnbx: dense J/K contraction that reads every (pq| tile ONCE FOR BOTH (p,q) AND (q,p)
// (include/nbx.h "J/K contraction, symmetric form").
//
//   (pq|rs) = (qp|rs): the N x N tile of (p,q) is the tile of (q,p).  Only the tiles q <= p are
//   streamed -- half the bytes of jk.hip's kernel, which is HBM bound -- and each one feeds
//       J_pq = J_qp = sum_ab T_ab Dtot_ab
//       K^x_pb += sum_a T_ab D^x_qa                (row p: stays in registers while p is fixed)
//       K^x_qb += sum_a T_ab D^x_pa   (q < p)      (row q: one partial row per tile)
//   PySCF's own J/K (libcvhf, which the reference calls) works on 8-fold packed integrals; this is
//   the part of that symmetry a streaming kernel gets for free.
//
// Work distribution as in jk.hip: persistent workgroups, one per occupancy slot, each owning an
// equal contiguous range of the triangular tile sequence T(p,q) = p(p+1)/2 + q; 16-byte
// non-temporal loads contiguous over the workgroup; thread t always sees the same column pair.
// Everything is summed in a fixed order (bitwise reproducible):
//   kpart1[w][slot][x][b]   row-p partials of workgroup w (flushed when its range crosses a row)
//   kpart2[q][p_local][x][b] row-q partial of tile (p,q), reduced over p by jk_sym_reduce_kernel
// The slab interface is additive: rows p in [p0,p1) of the tensor give PARTIAL full-size J and K
// matrices; the sum over slabs (an all-reduce across GPUs) is the result.
#include "nbx_common.h"
#include "synth_device.h"

namespace {

constexpr int JS_THREADS = 256;
// Tiles in flight per workgroup (QB) and the waves per SIMD the register budget is held to.  The
// row-q accumulators cost 4 VGPRs per tile and spin on top of the plain kernel's: QB = 4 fits
// 3 waves/SIMD (157 VGPRs), QB = 2 fits 4 (101).  Measured: QB = 2 wins for short rows
// (N = 148: 0.389 vs 0.403 ms), QB = 4 for long ones (N = 256: 2.89 vs 3.07 ms).
constexpr int js_waves(int qb) { return qb <= 2 ? 4 : 3; }
constexpr int JS_CUS = 256;
constexpr size_t JS_LDS_PER_CU = 160 * 1024;

__device__ __forceinline__ void fma2(double2& acc, double s, double2 v) {
    acc.x = fma(s, v.x, acc.x);
    acc.y = fma(s, v.y, acc.y);
}
__device__ __forceinline__ double dot2(double acc, double2 a, double2 b) { return fma(a.y, b.y, fma(a.x, b.x, acc)); }
__device__ __forceinline__ double2 ldnt(const double* p) {
    typedef double nbx_d2 __attribute__((ext_vector_type(2)));
    const nbx_d2 t = __builtin_nontemporal_load(reinterpret_cast<const nbx_d2*>(p));
    return make_double2(t.x, t.y);
}

__host__ __device__ __forceinline__ int64_t tri_index(int64_t p, int64_t q) { return p * (p + 1) / 2 + q; }
// row of the triangular index T: largest p with p(p+1)/2 <= T
__host__ __device__ __forceinline__ int tri_row(int64_t T) {
    int64_t p = (int64_t)((sqrt(8.0 * (double)T + 1.0) - 1.0) * 0.5);
    while (p * (p + 1) / 2 > T) --p;
    while ((p + 1) * (p + 2) / 2 <= T) ++p;
    return (int)p;
}

// One group of NQ tiles (p, q .. q+NQ-1): rows a = row0, row0 + rstep, ... of every tile.
//   dsh[a*NDM*(QB+1) + x*(QB+1) + j] = D^x[q+j][a] (j < QB), D^x[p][a] (j == QB): what a row needs
//   sits at constant offsets from one address (no per-read address arithmetic; -12 % at N = 192)
template <int NQ, int NDM, int QB>
__device__ __forceinline__ void js_group(const double* __restrict__ tile0, int64_t n2, const double* __restrict__ dtot,
                                         const double* dsh, int N, int row0, int rstep, int col,
                                         double (&jacc)[QB], double2 (&kacc1)[NDM], double2 (&kacc2)[NDM][QB]) {
    int a = row0;
    for (; a + rstep < N; a += 2 * rstep) {
        const int64_t o0 = (int64_t)a * N + col;
        const int64_t o1 = o0 + (int64_t)rstep * N;
        double2 t0[NQ], t1[NQ];
#pragma unroll
        for (int j = 0; j < NQ; ++j) {
            t0[j] = ldnt(tile0 + j * n2 + o0);
            t1[j] = ldnt(tile0 + j * n2 + o1);
        }
        const double2 d0 = *reinterpret_cast<const double2*>(dtot + o0);
        const double2 d1 = *reinterpret_cast<const double2*>(dtot + o1);
        double dp0[NDM], dp1[NDM];
#pragma unroll
        for (int x = 0; x < NDM; ++x) {
            dp0[x] = dsh[a * (NDM * (QB + 1)) + x * (QB + 1) + QB];
            dp1[x] = dsh[(a + rstep) * (NDM * (QB + 1)) + x * (QB + 1) + QB];
        }
#pragma unroll
        for (int j = 0; j < NQ; ++j) {
            jacc[j] = dot2(jacc[j], t0[j], d0);
            jacc[j] = dot2(jacc[j], t1[j], d1);
#pragma unroll
            for (int x = 0; x < NDM; ++x) {
                fma2(kacc1[x], dsh[a * (NDM * (QB + 1)) + x * (QB + 1) + j], t0[j]);
                fma2(kacc1[x], dsh[(a + rstep) * (NDM * (QB + 1)) + x * (QB + 1) + j], t1[j]);
                fma2(kacc2[x][j], dp0[x], t0[j]);
                fma2(kacc2[x][j], dp1[x], t1[j]);
            }
        }
    }
    if (a < N) {
        const int64_t o0 = (int64_t)a * N + col;
        double2 t0[NQ];
#pragma unroll
        for (int j = 0; j < NQ; ++j) t0[j] = ldnt(tile0 + j * n2 + o0);
        const double2 d0 = *reinterpret_cast<const double2*>(dtot + o0);
#pragma unroll
        for (int j = 0; j < NQ; ++j) {
            jacc[j] = dot2(jacc[j], t0[j], d0);
#pragma unroll
            for (int x = 0; x < NDM; ++x) {
                fma2(kacc1[x], dsh[a * (NDM * (QB + 1)) + x * (QB + 1) + j], t0[j]);
                fma2(kacc2[x][j], dsh[a * (NDM * (QB + 1)) + x * (QB + 1) + QB], t0[j]);
            }
        }
    }
}

template <int NDM, int QB>
__global__ __launch_bounds__(JS_THREADS) __attribute__((amdgpu_waves_per_eu(js_waves(QB), js_waves(QB))))
void jk_sym_kernel(const double* __restrict__ eri, const double* __restrict__ dm, const double* __restrict__ dtot,
                   double* __restrict__ jfull, double* __restrict__ kpart1, double* __restrict__ kpart2, int N,
                   int p0, int np, int64_t t_begin, int64_t t_end, int L, int S) {
    extern __shared__ __attribute__((aligned(16))) double smem[];
    // smem: dsh[N][NDM][QB+1] | kred[R*NDM*QB*N] | red[17]
    double* dsh = smem;
    double* kred = dsh + NDM * (QB + 1) * N;

    int64_t T = t_begin + (int64_t)blockIdx.x * L;
    const int64_t T_end = min(t_end, T + L);
    if (T >= T_end) return;  // uniform for the whole workgroup
    int p = tri_row(T);
    int q = (int)(T - tri_index(p, 0));
    const int p_first = p;

    const int CX = N / 2;  // column pairs per row (N even)
    const int R = max(1, JS_THREADS / CX);
    const int rowg = threadIdx.x / CX;
    const int cx = threadIdx.x - rowg * CX;
    const bool active = rowg < R;
    const int col = cx * 2;
    double* red = kred + (size_t)R * NDM * QB * N;

    double2 kacc1[NDM];
#pragma unroll
    for (int x = 0; x < NDM; ++x) kacc1[x] = make_double2(0.0, 0.0);
    const int64_t n2 = (int64_t)N * N;

    // row-p partials: reduce the R row groups through LDS, store, clear
    auto flush1 = [&](int prow) {
        double* kout = kpart1 + ((int64_t)blockIdx.x * S + (prow - p_first)) * NDM * N;
        __syncthreads();
        if (active) {
#pragma unroll
            for (int x = 0; x < NDM; ++x)
                *reinterpret_cast<double2*>(kred + ((size_t)rowg * NDM + x) * N + col) = kacc1[x];
        }
        __syncthreads();
        for (int i = threadIdx.x; i < NDM * N; i += JS_THREADS) {
            const int x = i / N, b = i - x * N;
            double tot = 0.0;
            for (int g = 0; g < R; ++g) tot += kred[((size_t)g * NDM + x) * N + b];
            kout[i] = tot;
        }
#pragma unroll
        for (int x = 0; x < NDM; ++x) kacc1[x] = make_double2(0.0, 0.0);
    };

    int p_cur = p_first;
    while (T < T_end) {
        if (p != p_cur) {
            flush1(p_cur);
            p_cur = p;
        }
        const int nq = (int)min((int64_t)min(QB, p - q + 1), T_end - T);
        __syncthreads();  // previous group's dsh / kred reads are done
        for (int i = threadIdx.x; i < NDM * (QB + 1) * N; i += JS_THREADS) {
            const int x = i / ((QB + 1) * N);
            const int rem = i - x * (QB + 1) * N;
            const int j = rem / N;
            const int a = rem - j * N;
            const int row = (j == QB) ? p : q + j;
            dsh[a * (NDM * (QB + 1)) + x * (QB + 1) + j] =
                (j == QB || j < nq) ? dm[(int64_t)x * n2 + (int64_t)row * N + a] : 0.0;
        }
        __syncthreads();
        double jacc[QB];
        double2 kacc2[NDM][QB];
#pragma unroll
        for (int j = 0; j < QB; ++j) {
            jacc[j] = 0.0;
#pragma unroll
            for (int x = 0; x < NDM; ++x) kacc2[x][j] = make_double2(0.0, 0.0);
        }
        if (active) {
            const double* tile0 = eri + ((int64_t)(p - p0) * N + q) * n2;
            if (nq == QB) {
                js_group<QB, NDM, QB>(tile0, n2, dtot, dsh, N, rowg, R, col, jacc, kacc1, kacc2);
            } else if (QB >= 4 && nq == 3) {
                js_group<(QB >= 4 ? 3 : 1), NDM, QB>(tile0, n2, dtot, dsh, N, rowg, R, col, jacc, kacc1, kacc2);
            } else if (QB >= 3 && nq == 2) {
                js_group<(QB >= 3 ? 2 : 1), NDM, QB>(tile0, n2, dtot, dsh, N, rowg, R, col, jacc, kacc1, kacc2);
            } else {
                js_group<1, NDM, QB>(tile0, n2, dtot, dsh, N, rowg, R, col, jacc, kacc1, kacc2);
            }
        }
        // J_{p,q+j} = J_{q+j,p}
#pragma unroll
        for (int j = 0; j < QB; ++j) {
            const double tot = nbx_block_sum(jacc[j], red);
            if (threadIdx.x == 0 && j < nq) {
                jfull[(int64_t)p * N + q + j] = tot;
                jfull[(int64_t)(q + j) * N + p] = tot;
            }
        }
        // row-(q+j) partials of the tiles below the diagonal
        if (active) {
#pragma unroll
            for (int x = 0; x < NDM; ++x)
#pragma unroll
                for (int j = 0; j < QB; ++j)
                    *reinterpret_cast<double2*>(kred + (((size_t)rowg * NDM + x) * QB + j) * N + col) = kacc2[x][j];
        }
        __syncthreads();
        for (int i = threadIdx.x; i < NDM * QB * N; i += JS_THREADS) {
            const int x = i / (QB * N);
            const int rem = i - x * QB * N;
            const int j = rem / N;
            const int b = rem - j * N;
            if (j < nq && q + j < p && b <= q + j) {  // (columns <= the row only: K is symmetric, the reduction mirrors the sums)
                double tot = 0.0;
                for (int g = 0; g < R; ++g) tot += kred[(((size_t)g * NDM + x) * QB + j) * N + b];
                kpart2[(((int64_t)(q + j) * np + (p - p0)) * NDM + x) * N + b] = tot;
            }
        }
        T += nq;
        q += nq;
        if (q > p) {
            ++p;
            q = 0;
        }
    }
    flush1(p_cur);
}

// out[(1+x)][row][b] for all N rows = (row in slab ? sum of its kpart1 partials : 0)
//                                     + sum over slab rows p > row of kpart2[row][p - p0][x][b]
// grid (N, NDM, ceil(N/64)); 256 threads = 4 row-chunks x 64 column lanes, 8 independent loads in
// flight per thread; fixed summation order.
template <int NCH>  // row-chunks per workgroup (64 NCH threads): the sum over p is NCH interleaved partial sums, added in order
__global__ __launch_bounds__(64 * NCH) void jk_sym_reduce_kernel(const double* __restrict__ kpart1,
                                                            const double* __restrict__ kpart2,
                                                            double* __restrict__ kout, int N, int p0, int np, int ndm,
                                                            int64_t t_begin, int L, int S, int accumulate,
                                                            const double* __restrict__ jfull = nullptr,
                                                            const double* __restrict__ hv = nullptr,
                                                            double* __restrict__ fock = nullptr,
                                                            double* __restrict__ vhf = nullptr,
                                                            int k2_tile_order = 0, int k_lower = 0) {
    // k_lower: the producer kernel stored only the columns b <= row of its partial rows (K is symmetric for the
    // symmetric densities the interface asks for: jk_m4.hip writes a third of the bytes); this kernel then sums the
    // elements b <= row and writes each to both places.  Across slabs the partial outputs are symmetrised partials,
    // whose sum is K all the same.
    __shared__ double part[NCH][64];
    const int row = blockIdx.x, x = blockIdx.y;
    const int lane = threadIdx.x & 63, chunk = threadIdx.x >> 6;
    if (k_lower && (int)blockIdx.z * 64 > row) return;  // (uniform: nothing of this block lies on or below the diagonal)
    const int b0 = blockIdx.z * 64 + lane;
    const int b = (k_lower && b0 > row) ? N : b0;  // (lanes right of the diagonal: idle)
    const int pl_lo = max(0, row + 1 - p0);  // first local p with global p > row
    double t = 0.0;
    if (b < N && k2_tile_order) {
        // kpart2[T(p, row) - t_begin][x][b]: the row-q partials in the order the tiles are visited (a
        // workgroup's stores are then sequential: with [q][p] every tile's 2.4 KB landed on a page of
        // its own, 350 KB from the last, and the address translation misses stalled the stream behind)
        const double* src = kpart2 + (int64_t)x * N + b;
        const int64_t stride = (int64_t)ndm * N;
        auto at = [&](int pl) {
            const int64_t pg = p0 + pl;
            return src[(pg * (pg + 1) / 2 + row - t_begin) * stride];
        };
        int pl = pl_lo + chunk;
        for (; pl + 7 * NCH < np; pl += 8 * NCH) {
            double v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = at(pl + NCH * u);
#pragma unroll
            for (int u = 0; u < 8; ++u) t += v[u];
        }
        for (; pl < np; pl += NCH) t += at(pl);
    } else if (b < N) {
        const double* src = kpart2 + (((int64_t)row * np) * ndm + x) * N + b;
        const int64_t stride = (int64_t)ndm * N;
        int pl = pl_lo + chunk;
        for (; pl + 7 * NCH < np; pl += 8 * NCH) {
            double v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = src[(pl + NCH * u) * stride];
#pragma unroll
            for (int u = 0; u < 8; ++u) t += v[u];
        }
        for (; pl < np; pl += NCH) t += src[pl * stride];
    }
    part[chunk][lane] = t;
    __syncthreads();
    if (chunk == 0 && b < N) {
        double tot = part[0][lane];
#pragma unroll
        for (int c = 1; c < NCH; ++c) tot += part[c][lane];
        if (row >= p0 && row < p0 + np) {
            const int64_t w_lo = (tri_index(row, 0) - t_begin) / L, w_hi = (tri_index(row, row) - t_begin) / L;
            for (int64_t w = w_lo; w <= w_hi; ++w) {
                const int slot = row - tri_row(t_begin + w * L);
                tot += kpart1[((w * S + slot) * ndm + x) * N + b];
            }
        }
        double* dst = kout + ((int64_t)x * N + row) * N + b;
        *dst = accumulate ? *dst + tot : tot;
        if (fock != nullptr) {  // Fock epilogue (whole tensor on this device): F = hv + J - K, vhf = J - K
            const int64_t o = ((int64_t)x * N + row) * N + b;
            const double v = jfull[(int64_t)row * N + b] - tot;
            fock[o] = hv[o] + v;
            if (vhf != nullptr) vhf[o] = v;
        }
        if (k_lower && b < row) {  // the mirrored element (J is symmetric too: jfull holds both [p][q] and [q][p])
            double* dst2 = kout + ((int64_t)x * N + b) * N + row;
            *dst2 = accumulate ? *dst2 + tot : tot;
            if (fock != nullptr) {
                const int64_t o = ((int64_t)x * N + b) * N + row;
                const double v = jfull[(int64_t)b * N + row] - tot;
                fock[o] = hv[o] + v;
                if (vhf != nullptr) vhf[o] = v;
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------
// The same contraction with the synthetic (pq|rs) of nbx_synth_eri GENERATED in registers (the
// N_AO = 2000 configuration): with the tiles q <= p only, half the hash evaluations of
// nbx_jk_synth, which is VALU bound.  One thread owns column pairs c, c + 256, ... (CS of them);
// the column-segment loop is the outer one, so only the row-p accumulators of all CS segments
// persist (and nothing needs a cross-thread reduction: a thread sees every row of its columns).
// The density rows a group needs are staged through LDS in chunks of GS_AC rows.
constexpr int GS_QB = 4;
constexpr int GS_AC = 256;

template <int NDM, int CS>
__global__ __launch_bounds__(JS_THREADS) void jk_synth_sym_kernel(
    const double* __restrict__ dm, const double* __restrict__ dtot, double* __restrict__ jfull,
    double* __restrict__ kpart1, double* __restrict__ kpart2, int N, int p0, int np, int64_t t_begin, int64_t t_end,
    int L, int S, uint64_t seed) {
    __shared__ double dsh[NDM * (GS_QB + 1) * GS_AC];
    __shared__ double red[17];
    int64_t T = t_begin + (int64_t)blockIdx.x * L;
    const int64_t T_end = min(t_end, T + L);
    if (T >= T_end) return;  // uniform for the whole workgroup
    int p = tri_row(T);
    int q = (int)(T - tri_index(p, 0));
    const int p_first = p;
    const int CXp = N / 2;
    const double scale = 1.0 / (double)N;
    const int64_t n2 = (int64_t)N * N;

    double2 kacc1[NDM][CS];
#pragma unroll
    for (int x = 0; x < NDM; ++x)
#pragma unroll
        for (int sg = 0; sg < CS; ++sg) kacc1[x][sg] = make_double2(0.0, 0.0);

    auto flush1 = [&](int prow) {
        double* kout = kpart1 + ((int64_t)blockIdx.x * S + (prow - p_first)) * NDM * N;
#pragma unroll
        for (int sg = 0; sg < CS; ++sg) {
            const int c = threadIdx.x + sg * JS_THREADS;
            if (c < CXp) {
#pragma unroll
                for (int x = 0; x < NDM; ++x) *reinterpret_cast<double2*>(kout + (int64_t)x * N + 2 * c) = kacc1[x][sg];
            }
#pragma unroll
            for (int x = 0; x < NDM; ++x) kacc1[x][sg] = make_double2(0.0, 0.0);
        }
    };

    int p_cur = p_first;
    while (T < T_end) {
        if (p != p_cur) {
            flush1(p_cur);
            p_cur = p;
        }
        const int nq = (int)min((int64_t)min(GS_QB, p - q + 1), T_end - T);
        uint32_t pq[GS_QB];
#pragma unroll
        for (int j = 0; j < GS_QB; ++j) pq[j] = nbx_tri_pair_u32((uint32_t)p, (uint32_t)min(q + j, p));
        double jacc[GS_QB];
#pragma unroll
        for (int j = 0; j < GS_QB; ++j) jacc[j] = 0.0;
#pragma unroll
        for (int sg = 0; sg < CS; ++sg) {
            const int c = threadIdx.x + sg * JS_THREADS;
            const bool mine = c < CXp;
            const int col = 2 * c;
            double2 kacc2[NDM][GS_QB];
#pragma unroll
            for (int x = 0; x < NDM; ++x)
#pragma unroll
                for (int j = 0; j < GS_QB; ++j) kacc2[x][j] = make_double2(0.0, 0.0);
            for (int a0 = 0; a0 < N; a0 += GS_AC) {
                const int na = min(GS_AC, N - a0);
                __syncthreads();  // the previous chunk has been consumed
                for (int i = threadIdx.x; i < NDM * (GS_QB + 1) * GS_AC; i += JS_THREADS) {
                    const int x = i / ((GS_QB + 1) * GS_AC);
                    const int rem = i - x * (GS_QB + 1) * GS_AC;
                    const int j = rem / GS_AC;
                    const int a = rem - j * GS_AC;
                    const int row = (j == GS_QB) ? p : q + j;
                    dsh[i] = ((j == GS_QB || j < nq) && a < na) ? dm[(int64_t)x * n2 + (int64_t)row * N + a0 + a] : 0.0;
                }
                __syncthreads();
                if (mine) {
                    for (int a = 0; a < na; ++a) {
                        const int ag = a0 + a;
                        const double2 d = *reinterpret_cast<const double2*>(dtot + (int64_t)ag * N + col);
                        const uint32_t ab0 = nbx_tri_pair_u32((uint32_t)ag, (uint32_t)col);
                        const uint32_t ab1 = nbx_tri_pair_u32((uint32_t)ag, (uint32_t)(col + 1));
                        double dp[NDM];
#pragma unroll
                        for (int x = 0; x < NDM; ++x) dp[x] = dsh[(x * (GS_QB + 1) + GS_QB) * GS_AC + a];
#pragma unroll
                        for (int j = 0; j < GS_QB; ++j) {
                            if (j < nq) {  // uniform
                                double2 t;
                                t.x = nbx_synth_val(0, nbx_tri_u32(pq[j], ab0), seed) * scale;
                                t.y = nbx_synth_val(0, nbx_tri_u32(pq[j], ab1), seed) * scale;
                                jacc[j] = dot2(jacc[j], t, d);
#pragma unroll
                                for (int x = 0; x < NDM; ++x) {
                                    fma2(kacc1[x][sg], dsh[(x * (GS_QB + 1) + j) * GS_AC + a], t);
                                    fma2(kacc2[x][j], dp[x], t);
                                }
                            }
                        }
                    }
                }
            }
            // this segment's columns of the row-(q+j) partials: complete, straight to the partial buffer
            if (mine) {
#pragma unroll
                for (int j = 0; j < GS_QB; ++j) {
                    if (j < nq && q + j < p) {
#pragma unroll
                        for (int x = 0; x < NDM; ++x)
                            *reinterpret_cast<double2*>(kpart2 + (((int64_t)(q + j) * np + (p - p0)) * NDM + x) * N + col) =
                                kacc2[x][j];
                    }
                }
            }
        }
#pragma unroll
        for (int j = 0; j < GS_QB; ++j) {
            const double tot = nbx_block_sum(jacc[j], red);
            if (threadIdx.x == 0 && j < nq) {
                jfull[(int64_t)p * N + q + j] = tot;
                jfull[(int64_t)(q + j) * N + p] = tot;
            }
        }
        T += nq;
        q += nq;
        if (q > p) {
            ++p;
            q = 0;
        }
    }
    flush1(p_cur);
}

__global__ void js_dtot_kernel(const double* __restrict__ dm, double* __restrict__ dtot, int64_t n2, int ndm) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n2) return;
    double t = dm[i];
    for (int x = 1; x < ndm; ++x) t += dm[x * n2 + i];
    dtot[i] = t;
}

struct JsPlan {
    int wgs, L, S, R;
    size_t lds_bytes, dtot_off, k1_off, k2_off, total;
};

size_t align256(size_t x) { return (x + 255) & ~(size_t)255; }

int js_qb(int64_t N) { return N <= 192 ? 2 : 4; }

JsPlan js_plan(int64_t N, int64_t p0, int64_t np, int64_t ndm) {
    JsPlan pl;
    const int QB = js_qb(N);
    const int64_t CX = N / 2;
    pl.R = (int)(JS_THREADS / CX > 0 ? JS_THREADS / CX : 1);
    pl.lds_bytes = (size_t)(ndm * (QB + 1) * N + (int64_t)pl.R * ndm * QB * N + 17) * sizeof(double);
    int64_t per_cu = js_waves(QB);  // workgroups of 4 waves per CU = waves per SIMD
    const int64_t by_lds = (int64_t)(JS_LDS_PER_CU / (pl.lds_bytes + 256));
    if (per_cu > by_lds) per_cu = by_lds > 0 ? by_lds : 1;
    const int64_t slots = JS_CUS * per_cu;
    const int64_t ntiles = tri_index(p0 + np, 0) - tri_index(p0, 0);
    int64_t L = nbx_cdiv(ntiles, slots);
    if (L < 1) L = 1;
    pl.L = (int)L;
    pl.wgs = (int)nbx_cdiv(ntiles, L);
    pl.S = (int)sqrt(2.0 * (double)L) + 3;
    size_t off = 0;
    pl.dtot_off = off; off += align256((size_t)(N * N) * sizeof(double));
    pl.k1_off = off; off += align256((size_t)((int64_t)pl.wgs * pl.S * ndm * N) * sizeof(double));
    pl.k2_off = off; off += align256((size_t)(N * np * ndm * N) * sizeof(double));
    pl.total = off;
    return pl;
}

}  // namespace

bool nbx_jk_sym_supported(int64_t nao) { return nao >= 2 && nao % 2 == 0 && nao / 2 <= JS_THREADS; }

// K[x][row][b] from the row-p / row-q partial buffers (shared with jk_s4.hip, same layouts)
int nbx_jk_sym_reduce(nbx_ctx* ctx, const double* k1, const double* k2, double* d_k, int64_t N, int64_t p0, int64_t np,
                      int64_t ndm, int64_t t_begin, int L, int S, const double* d_j, const double* d_hv, double* d_fock,
                      double* d_vhf, int k2_tile_order, int k_lower) {
    if (k_lower)  // (jk_m4.hip: sixteen row-chunks -- the sum over p in two trips to memory instead of five)
        hipLaunchKernelGGL(jk_sym_reduce_kernel<16>, dim3((unsigned)N, (unsigned)ndm, (unsigned)nbx_cdiv(N, 64)), dim3(1024), 0,
                           ctx->stream, k1, k2, d_k, (int)N, (int)p0, (int)np, (int)ndm, t_begin, L, S, 0, d_j, d_hv, d_fock,
                           d_vhf, k2_tile_order, k_lower);
    else
        hipLaunchKernelGGL(jk_sym_reduce_kernel<4>, dim3((unsigned)N, (unsigned)ndm, (unsigned)nbx_cdiv(N, 64)), dim3(256), 0,
                           ctx->stream, k1, k2, d_k, (int)N, (int)p0, (int)np, (int)ndm, t_begin, L, S, 0, d_j, d_hv, d_fock,
                           d_vhf, k2_tile_order, k_lower);
    NBX_LAUNCH_CHECK();
    return NBX_OK;
}

extern "C" size_t nbx_jk_dense_sym_worksize(int64_t nao, int64_t p0, int64_t p1, int64_t ndm) {
    if (nao <= 0 || p0 < 0 || p1 < p0 || p1 > nao || ndm <= 0) return 0;
    if (!nbx_jk_sym_supported(nao))  // falls back to the plain kernel + a scatter into full-size matrices
        return align256(nbx_jk_dense_worksize(nao, p1 - p0, ndm)) +
               align256((size_t)((1 + ndm) * (p1 - p0) * nao) * sizeof(double));
    return js_plan(nao, p0, p1 - p0, ndm).total;
}

namespace {
// fallback path: rows [p0,p1) of a slab result into the full-size (1+ndm, N, N) partial matrices
__global__ void js_scatter_rows_kernel(const double* __restrict__ slab, double* __restrict__ full, int N, int p0, int np,
                                       int nmat) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (int64_t)nmat * N * N) return;
    const int b = (int)(i % N);
    const int row = (int)((i / N) % N);
    const int m = (int)(i / ((int64_t)N * N));
    full[i] = (row >= p0 && row < p0 + np) ? slab[((int64_t)m * np + (row - p0)) * N + b] : 0.0;
}
}  // namespace

static int jk_dense_sym_impl(nbx_ctx* ctx, int64_t nao, int64_t p0, int64_t p1, const double* d_eri, const double* d_dm,
                             int64_t ndm, double* d_jk, void* d_work, size_t work_bytes, const double* d_hv, double* d_fock,
                             double* d_vhf);

extern "C" int nbx_jk_dense_sym(nbx_ctx* ctx, int64_t nao, int64_t p0, int64_t p1, const double* d_eri,
                                const double* d_dm, int64_t ndm, double* d_jk, void* d_work, size_t work_bytes) {
    return jk_dense_sym_impl(ctx, nao, p0, p1, d_eri, d_dm, ndm, d_jk, d_work, work_bytes, nullptr, nullptr, nullptr);
}

// The whole tensor with two densities and the Fock assembly F = hv + J - K[x], vhf = J - K[x] in the reduction kernel's
// epilogue (the SCF cycle on the dense tensor: N < 97 and the sizes the packed kernels have no instance for) -- what
// nbx_jk_dense_sym + nbx_fock_uhf give, bit for bit, in one launch less.  NBX_E_UNSUPPORTED (nothing launched): no symmetric
// kernel for this size (odd N): the caller takes the two calls.
int nbx_jk_dense_sym_fock(nbx_ctx* ctx, int64_t nao, const double* d_eri, const double* d_dm, const double* d_hv, double* d_jk,
                          double* d_fock, double* d_vhf, void* d_work, size_t work_bytes) {
    NBX_CHECK_ARG(d_hv && d_fock);
    if (!nbx_jk_sym_supported(nao)) return NBX_E_UNSUPPORTED;
    return jk_dense_sym_impl(ctx, nao, 0, nao, d_eri, d_dm, 2, d_jk, d_work, work_bytes, d_hv, d_fock, d_vhf);
}

static int jk_dense_sym_impl(nbx_ctx* ctx, int64_t nao, int64_t p0, int64_t p1, const double* d_eri, const double* d_dm,
                             int64_t ndm, double* d_jk, void* d_work, size_t work_bytes, const double* d_hv, double* d_fock,
                             double* d_vhf) {
    NBX_CHECK_ARG(ctx && d_dm && d_jk);
    NBX_CHECK_ARG(nao > 0 && p0 >= 0 && p1 >= p0 && p1 <= nao);
    NBX_CHECK_ARG(d_eri != nullptr || p0 == p1);  // an empty slab has no storage
    NBX_CHECK_ARG(ndm == 1 || ndm == 2);
    const int64_t np = p1 - p0, N = nao, n2 = N * N;
    const size_t need = nbx_jk_dense_sym_worksize(nao, p0, p1, ndm);
    if (d_work == nullptr || work_bytes < need) {
        nbx_set_error("nbx_jk_dense_sym: workspace %zu < %zu bytes", work_bytes, need);
        return NBX_E_NOMEM;
    }
    NBX_CHECK_ARG((reinterpret_cast<uintptr_t>(d_eri) & 15) == 0 && (reinterpret_cast<uintptr_t>(d_work) & 15) == 0);
    if (np == 0) return nbx_memset(ctx, d_jk, 0, (size_t)((1 + ndm) * n2) * sizeof(double));
    if (!nbx_jk_sym_supported(nao)) {
        char* base = static_cast<char*>(d_work);
        const size_t w0 = align256(nbx_jk_dense_worksize(nao, np, ndm));
        double* slab = reinterpret_cast<double*>(base + w0);
        int rc = nbx_jk_dense(ctx, nao, p0, p1, d_eri, d_dm, ndm, slab, base, w0);
        if (rc != NBX_OK) return rc;
        const int64_t tot = (1 + ndm) * n2;
        hipLaunchKernelGGL(js_scatter_rows_kernel, dim3((unsigned)nbx_cdiv(tot, 256)), dim3(256), 0, ctx->stream, slab,
                           d_jk, (int)N, (int)p0, (int)np, (int)(1 + ndm));
        NBX_LAUNCH_CHECK();
        return NBX_OK;
    }
    const JsPlan pl = js_plan(N, p0, np, ndm);
    char* base = static_cast<char*>(d_work);
    double* dtot = reinterpret_cast<double*>(base + pl.dtot_off);
    double* k1 = reinterpret_cast<double*>(base + pl.k1_off);
    double* k2 = reinterpret_cast<double*>(base + pl.k2_off);
    if (np < N) {  // J entries this slab does not own must read as zero
        const int rc = nbx_memset(ctx, d_jk, 0, (size_t)n2 * sizeof(double));
        if (rc != NBX_OK) return rc;
    }
    hipLaunchKernelGGL(js_dtot_kernel, dim3((unsigned)nbx_cdiv(n2, 256)), dim3(256), 0, ctx->stream, d_dm, dtot, n2,
                       (int)ndm);
    NBX_LAUNCH_CHECK();
    const int64_t t_begin = tri_index(p0, 0), t_end = tri_index(p1, 0);
    {
        nbx_prof_scope prof(ctx, NBX_PROF_JK_DENSE);
#define NBX_JS_GO(NDM_, QB_)                                                                                          \
    do {                                                                                                              \
        static bool attr_set = false;                                                                                 \
        if (!attr_set) {                                                                                              \
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&jk_sym_kernel<NDM_, QB_>),                       \
                                      hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);                        \
            attr_set = true;                                                                                          \
        }                                                                                                             \
        hipLaunchKernelGGL((jk_sym_kernel<NDM_, QB_>), dim3((unsigned)pl.wgs), dim3(JS_THREADS), pl.lds_bytes,        \
                           ctx->stream, d_eri, d_dm, dtot, d_jk, k1, k2, (int)N, (int)p0, (int)np, t_begin, t_end,    \
                           pl.L, pl.S);                                                                               \
    } while (0)
        if (js_qb(N) == 2) {
            if (ndm == 2) NBX_JS_GO(2, 2);
            else NBX_JS_GO(1, 2);
        } else {
            if (ndm == 2) NBX_JS_GO(2, 4);
            else NBX_JS_GO(1, 4);
        }
#undef NBX_JS_GO
    }
    NBX_LAUNCH_CHECK();
    hipLaunchKernelGGL(jk_sym_reduce_kernel<4>, dim3((unsigned)N, (unsigned)ndm, (unsigned)nbx_cdiv(N, 64)), dim3(256), 0,
                       ctx->stream, k1, k2,
                       d_jk + n2, (int)N, (int)p0, (int)np, (int)ndm, t_begin, pl.L, pl.S, 0, d_fock ? d_jk : nullptr, d_hv, d_fock,
                       d_vhf, 0, 1);
    NBX_LAUNCH_CHECK();
    return NBX_OK;
}

// ------------------------------------------------------------------ streamed symmetric form
namespace {
struct GsPlan {
    int wgs, L, S, cs;
    int64_t pc;  // slab rows per pass (bounds the row-q partial buffer)
    size_t dtot_off, k1_off, k2_off, total;
};

GsPlan gs_plan(int64_t N, int64_t np, int64_t ndm) {
    GsPlan pl;
    const int64_t CXp = N / 2;
    pl.cs = CXp <= JS_THREADS ? 1 : (CXp <= 2 * JS_THREADS ? 2 : 4);
    int64_t pc = (int64_t)(4.0e9 / ((double)N * (double)N * (double)ndm * 8.0));
    if (pc < 1) pc = 1;
    if (pc > np) pc = np > 0 ? np : 1;
    pl.pc = pc;
    // geometry of the largest pass (the last rows are the longest): used for every pass
    const int64_t slots = JS_CUS * (pl.cs == 4 ? 3 : 4);  // 142 VGPRs at 4 segments: 3 workgroups per CU
    const int64_t ntiles_max = tri_index(N, 0) - tri_index(N - pc, 0);
    int64_t L = nbx_cdiv(ntiles_max, slots);
    if (L < 1) L = 1;
    pl.L = (int)L;
    pl.wgs = (int)nbx_cdiv(ntiles_max, L);
    pl.S = (int)sqrt(2.0 * (double)L) + 3;
    size_t off = 0;
    pl.dtot_off = off; off += align256((size_t)(N * N) * sizeof(double));
    pl.k1_off = off; off += align256((size_t)((int64_t)pl.wgs * pl.S * ndm * N) * sizeof(double));
    pl.k2_off = off; off += align256((size_t)(N * pc * ndm * N) * sizeof(double));
    pl.total = off;
    return pl;
}
}  // namespace

bool nbx_jk_synth_sym_supported(int64_t nao) { return nao >= 2 && nao % 2 == 0 && nao / 2 <= 4 * JS_THREADS; }

extern "C" size_t nbx_jk_synth_sym_worksize(int64_t nao, int64_t p0, int64_t p1, int64_t ndm) {
    if (nao <= 0 || p0 < 0 || p1 < p0 || p1 > nao || ndm <= 0 || !nbx_jk_synth_sym_supported(nao)) return 0;
    return gs_plan(nao, p1 - p0, ndm).total;
}

extern "C" int nbx_jk_synth_sym(nbx_ctx* ctx, int64_t nao, int64_t p0, int64_t p1, uint64_t seed, const double* d_dm,
                                int64_t ndm, double* d_jk, void* d_work, size_t work_bytes) {
    NBX_CHECK_ARG(ctx && d_dm && d_jk);
    NBX_CHECK_ARG(nao > 0 && p0 >= 0 && p1 >= p0 && p1 <= nao);
    NBX_CHECK_ARG(ndm == 1 || ndm == 2);
    if (!nbx_jk_synth_sym_supported(nao)) {
        nbx_set_error("nbx_jk_synth_sym: N=%lld is not covered (even N <= 2048)", (long long)nao);
        return NBX_E_UNSUPPORTED;
    }
    const int64_t N = nao, n2 = N * N, np_all = p1 - p0;
    const size_t need = nbx_jk_synth_sym_worksize(nao, p0, p1, ndm);
    if (d_work == nullptr || work_bytes < need) {
        nbx_set_error("nbx_jk_synth_sym: workspace %zu < %zu bytes", work_bytes, need);
        return NBX_E_NOMEM;
    }
    NBX_CHECK_ARG((reinterpret_cast<uintptr_t>(d_work) & 15) == 0);
    int rc = nbx_memset(ctx, d_jk, 0, (size_t)((1 + ndm) * n2) * sizeof(double));
    if (rc != NBX_OK || np_all == 0) return rc;
    const GsPlan pl = gs_plan(N, np_all, ndm);
    char* base = static_cast<char*>(d_work);
    double* dtot = reinterpret_cast<double*>(base + pl.dtot_off);
    double* k1 = reinterpret_cast<double*>(base + pl.k1_off);
    double* k2 = reinterpret_cast<double*>(base + pl.k2_off);
    hipLaunchKernelGGL(js_dtot_kernel, dim3((unsigned)nbx_cdiv(n2, 256)), dim3(256), 0, ctx->stream, d_dm, dtot, n2,
                       (int)ndm);
    NBX_LAUNCH_CHECK();
    for (int64_t c0 = p0; c0 < p1; c0 += pl.pc) {
        const int64_t c1 = (c0 + pl.pc < p1) ? c0 + pl.pc : p1;
        const int64_t np = c1 - c0;
        const int64_t t_begin = tri_index(c0, 0), t_end = tri_index(c1, 0);
        const unsigned wgs = (unsigned)nbx_cdiv(t_end - t_begin, pl.L);
        {
            nbx_prof_scope prof(ctx, NBX_PROF_JK_DENSE);
#define NBX_GS_GO(NDM_, CS_)                                                                                        \
    hipLaunchKernelGGL((jk_synth_sym_kernel<NDM_, CS_>), dim3(wgs), dim3(JS_THREADS), 0, ctx->stream, d_dm, dtot, d_jk, \
                       k1, k2, (int)N, (int)c0, (int)np, t_begin, t_end, pl.L, pl.S, seed)
            if (ndm == 2) {
                if (pl.cs == 1) NBX_GS_GO(2, 1);
                else if (pl.cs == 2) NBX_GS_GO(2, 2);
                else NBX_GS_GO(2, 4);
            } else {
                if (pl.cs == 1) NBX_GS_GO(1, 1);
                else if (pl.cs == 2) NBX_GS_GO(1, 2);
                else NBX_GS_GO(1, 4);
            }
#undef NBX_GS_GO
        }
        NBX_LAUNCH_CHECK();
        hipLaunchKernelGGL(jk_sym_reduce_kernel<4>, dim3((unsigned)N, (unsigned)ndm, (unsigned)nbx_cdiv(N, 64)), dim3(256), 0,
                           ctx->stream, k1, k2, d_jk + n2, (int)N, (int)c0, (int)np, (int)ndm, t_begin, pl.L, pl.S, 1);
        NBX_LAUNCH_CHECK();
    }
    return NBX_OK;
}
