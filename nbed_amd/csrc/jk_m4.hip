// libnbx: the packed J/K contraction with its "walk" on the matrix cores (include/nbx.h "J/K contraction, packed form").
//
// jk_s4.hip streams the 4-fold packed tiles at 0.55 of the HBM roofline because the two symmetric matrix x vector
// products per spin that every tile feeds (K[p,:] += M d_q, K[q,:] += M d_p, M = the (r,s) matrix of the tile) are
// walked on the vector ALU: 36-50 instructions per 16 FMAs with 37 of 64 lanes live at N = 148.
// Here the same products run as  out[t][0:4] = sum_c M[t][c] X[c][0:4],  X[c] = (D^a_q, D^a_p, D^b_q, D^b_p)[c],  on
// v_mfma_f64_4x4x4_4b_f64: four independent 4 x 4 x 4 products per instruction, 256 multiply-adds, all of them useful
// (the 16x16x4 form would leave 12 of its 16 columns idle).  Measured in isolation (profiles/r03/
// jk_mfma_walk_probe.hip): 1.65 us per tile per CU with four waves against the 3.5 us at which tiles stream through a
// CU -- the walk fits under the stream.
//
// Layout (nbx_eri_pack makes it for the sizes this kernel serves): the tiles T(p,q) = p(p+1)/2 + q, q <= p, in
// sequence; a tile is the lower triangle of M in 4 x 4 BLOCKS, block (T, C <= T) at T(T+1)/2 + C, 16 doubles each,
// element (row i, column k) at 4 k + i, the upper part of the diagonal blocks stored as zeros.  With lane = 16 a + 4 b + c
// the operands of one MFMA are
//   row part    item (G, C):  A = block (4 G + b, C) element (c, a)    B = X[4 C + a][c]   D -> out rows 16 G + 4 b + a
//   column part item (T, H):  A = block (T, 4 H + b) element (a, c)    B = X[4 T + a][c]   D -> out rows 16 H + 4 b + a
// -- the column part reads four consecutive blocks (512 contiguous bytes, conflict free), the row part four runs of
// 128 bytes.  (Both MFMA operands carry the contraction index in lane >> 4: one register can be contracted over one of
// its two indices only, so the column part is a second LDS read of the block in the transposed lane map; the triangle
// is read twice per tile.)  Wave w of the four takes block columns / block rows = w (mod 4), which makes every loop bound a
// compile-time constant: straight-line code, LDS reads batched ahead of the MFMAs by the compiler.
//
// Streaming, work distribution and reductions as in jk_s4.hip: persistent workgroups over equal contiguous ranges of the
// tile sequence, two per CU; a tile arrives in four chunks (whole block rows, equal block counts) through registers
// (non-temporal 16-byte loads, a whole tile in flight) into two LDS buffers, one barrier per chunk; J is a flat dot
// product of the staged registers with a Dtot' table held in registers; row-q partials go out per tile, row-p
// partials when the range crosses a row, both summed in a fixed order by jk_sym_reduce_kernel (with the Fock epilogue).
#include <cstdlib>

#include "nbx_common.h"

namespace {

constexpr int M4_THREADS = 512, M4_WAVES = M4_THREADS / 64, M4_CUS = 256, M4_NCH = 4;
// ONE eight-wave workgroup per CU (two waves per SIMD: LDS and matrix-pipe latencies of one wave hide behind the other):
// the per-thread share of the Dtot' table and of the staged tile is half that of a four-wave workgroup (48 + 48
// registers), which is what lets two waves per SIMD fit the register file without spilling
constexpr int M4_PER_CU = 1;

__host__ __device__ constexpr int m4_tri(int k) { return k * (k + 1) / 2; }

template <int NB>
struct M4Geom {
    static constexpr int N = 4 * NB, NG = (NB + 3) / 4, NBLK = m4_tri(NB), TILE = 16 * NBLK;
    // chunk k holds the block rows [row0(k), row0(k + 1)): the first block row at which a quarter of the blocks is reached
    static constexpr int row0(int k) {
        if (k <= 0) return 0;
        if (k >= M4_NCH) return NB;
        int t = 0;
        while (m4_tri(t) * M4_NCH < k * NBLK) ++t;
        return t;
    }
    static constexpr int blocks(int k) { return m4_tri(row0(k + 1)) - m4_tri(row0(k)); }
    static constexpr int max_blocks() {
        int m = 0;
        for (int k = 0; k < M4_NCH; ++k) m = blocks(k) > m ? blocks(k) : m;
        return m;
    }
    static constexpr int LPT = (max_blocks() * 128 + M4_THREADS * 16 - 1) / (M4_THREADS * 16);  // 16-byte loads per thread
    static constexpr int BUF = LPT * M4_THREADS * 2;                                            // doubles per LDS buffer
};

typedef double m4_d2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ double2 m4_ldnt(const double* p) {
    const m4_d2 t = __builtin_nontemporal_load(reinterpret_cast<const m4_d2*>(p));
    return make_double2(t.x, t.y);
}

__host__ __device__ __forceinline__ int m4_tri_row(int64_t T) {
    int64_t p = (int64_t)((sqrt(8.0 * (double)T + 1.0) - 1.0) * 0.5);
    while (p * (p + 1) / 2 > T) --p;
    while ((p + 1) * (p + 2) / 2 <= T) ++p;
    return (int)p;
}

// ---------------------------------------------------------------------------------------------- pack and weights
// slab rows [p0, p0 + np) of the dense tensor -> block-major tiles; one workgroup per tile
template <int NB>
__global__ __launch_bounds__(256) void m4_pack_kernel(const double* __restrict__ eri, double* __restrict__ out, int p0,
                                                      int64_t t_begin) {
    using G = M4Geom<NB>;
    const int64_t T = t_begin + blockIdx.x;
    const int p = m4_tri_row(T), q = (int)(T - (int64_t)p * (p + 1) / 2);
    const double* src = eri + ((int64_t)(p - p0) * G::N + q) * G::N * G::N;
    double* dst = out + (int64_t)blockIdx.x * G::TILE;
    for (int e = threadIdx.x; e < G::TILE; e += 256) {
        const int blk = e >> 4, k = (e >> 2) & 3, i = e & 3;
        const int bt = m4_tri_row(blk), bc = blk - m4_tri(bt);
        const int row = 4 * bt + i, col = 4 * bc + k;
        dst[e] = col <= row ? src[(int64_t)row * G::N + col] : 0.0;
    }
}

// Dtot' in the staging order of the main kernel: wt[(k LPT + s) M4_THREADS + tid] = the weights of the two doubles that
// thread tid holds in slot s of chunk k: Dtot[r][c] + Dtot[c][r] below the diagonal, Dtot[r][r] on it, 0 elsewhere
// (the zeros of the diagonal blocks, the clamped tail loads of a chunk)
template <int NB>
__global__ __launch_bounds__(256) void m4_weights_kernel(const double* __restrict__ dm, int ndm, double* __restrict__ wt) {
    using G = M4Geom<NB>;
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= M4_NCH * G::LPT * M4_THREADS) return;
    const int tid = i % M4_THREADS, s = (i / M4_THREADS) % G::LPT, k = (i / M4_THREADS) / G::LPT;
    const int d0 = 16 * m4_tri(G::row0(k)) + (s * M4_THREADS + tid) * 2, dend = 16 * m4_tri(G::row0(k + 1));
    const int64_t n2 = (int64_t)G::N * G::N;
    double out[2] = {0.0, 0.0};
    for (int e = 0; e < 2; ++e) {
        const int d = d0 + e;
        if (d >= dend) continue;
        const int blk = d >> 4, kk = (d >> 2) & 3, ii = d & 3;
        const int bt = m4_tri_row(blk), bc = blk - m4_tri(bt);
        const int row = 4 * bt + ii, col = 4 * bc + kk;
        if (col > row) continue;
        double v = 0.0, vt = 0.0;
        for (int x = 0; x < ndm; ++x) {
            v += dm[x * n2 + (int64_t)row * G::N + col];
            vt += dm[x * n2 + (int64_t)col * G::N + row];
        }
        out[e] = row == col ? v : v + vt;
    }
    *reinterpret_cast<double2*>(wt + 2 * (int64_t)i) = make_double2(out[0], out[1]);
}

// ---------------------------------------------------------------------------------------------- the walk of one chunk
// buf: the chunk in LDS (block (T, C) at 16 (tri(T) - tri(RA) + C)); xs: X[N][4] of this tile; acc[G]: this wave's partial
// out rows 16 G + 4 b + a (D layout of the 4x4x4 product: lane 16 i + 4 b + j holds D_b[i][j]).
// Eight waves: wave = 4 h + w4 takes the block columns C = 4 j + w4 with j = h (mod 2) of the row part and the block rows
// T = RA + 4 j + w4 with j = 1 - h (mod 2) of the column part (HP = h: a template parameter, so that every bound stays static).
template <int NB, int K, int HP>
__device__ __forceinline__ void m4_walk_chunk(const double* __restrict__ buf, const double* __restrict__ xs, int w4, int a,
                                              int b, int c, double (&acc)[M4Geom<NB>::NG]) {
    using G_ = M4Geom<NB>;
    constexpr int RA = G_::row0(K), RB = G_::row0(K + 1), NG = G_::NG;
    constexpr int BASE = m4_tri(RA);
    // ---- row part: items (G, C = 4 j + w4), C <= T = 4 G + b, T in [RA, RB)
    const int lo_row = 4 * a + c;
#pragma unroll
    for (int j = HP; j < (NB + 3) / 4; j += 2) {
        if (4 * j >= RB) continue;  // static: no block of this column group lies in the chunk
        const int C = 4 * j + w4;
        if (4 * j + 3 >= NB && C >= NB) continue;  // scalar (last j only)
        const double bx = xs[4 * (4 * C + a) + c];
#pragma unroll
        for (int G = 0; G < NG; ++G) {
            if (G < j || 4 * G + 3 < RA || 4 * G >= RB) continue;  // static
            const int T = 4 * G + b;
            double av;
            const int addr = 16 * (m4_tri(4 * G) - BASE + 4 * j) + 16 * (G * 4 * b + m4_tri(b) + w4) + lo_row;
            constexpr bool interior = true;
            if (interior && 4 * G >= RA && 4 * G + 3 < RB && 4 * G + 3 < NB && j < G) {
                av = buf[addr];
            } else {
                av = (T >= RA && T < RB && T < NB && C <= T) ? buf[addr] : 0.0;
            }
            acc[G] = __builtin_amdgcn_mfma_f64_4x4x4f64(av, bx, acc[G], 0, 0, 0);
        }
        // (one column group at a time: the scheduler would otherwise hoist every LDS read of the chunk -- 100 of them --
        // ahead of the first MFMA, and the kernel has no registers for that next to the Dtot' table)
        __builtin_amdgcn_sched_barrier(0);
    }
    // ---- column part: items (T = RA + 4 j + w4, H), block columns 4 H + b < T, or == T with the strict lower part
    const int lo_col = 16 * b + 4 * c + a;
#pragma unroll
    for (int j = 1 - HP; j < (RB - RA + 3) / 4; j += 2) {
        const int T = RA + 4 * j + w4;
        if (RA + 4 * j + 3 >= RB && T >= RB) continue;  // scalar (last j only)
        const double bt = xs[4 * (4 * T + a) + c];
        const double* lt = buf + 16 * (m4_tri(T) - BASE) + lo_col;
#pragma unroll
        for (int H = 0; H < NG; ++H) {
            if (4 * H > RA + 4 * j + 3) continue;  // static: the whole group lies right of every T of this j
            double av;
            if (4 * H + 3 < RA + 4 * j) {
                av = lt[64 * H];  // static: every block column of the group is left of T
            } else {
                const int cb = 4 * H + b;
                av = (cb < T || (cb == T && c < a)) ? lt[64 * H] : 0.0;
            }
            acc[H] = __builtin_amdgcn_mfma_f64_4x4x4f64(av, bt, acc[H], 0, 0, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
    }
}

// ---------------------------------------------------------------------------------------------- the kernel
// kpart1[(w S + slot) NDM + x][N]: row-p partial of workgroup w for the slot-th row of its range;
// kpart2[(T - t_begin) NDM + x][N]: row-q partial of tile T (q < p); jfull (N, N): J[p][q] = J[q][p] of the tiles visited
template <int NB, int NDM>
__global__ __launch_bounds__(M4_THREADS, M4_PER_CU) void jk_m4_kernel(const double* __restrict__ packed, const double* __restrict__ dm,
                                                              const double* __restrict__ wtab, double* __restrict__ jfull,
                                                              double* __restrict__ kpart1, double* __restrict__ kpart2,
                                                              int64_t t_begin, int64_t t_end, int L, int S) {
    using G_ = M4Geom<NB>;
    constexpr int N = G_::N, NG = G_::NG, LPT = G_::LPT, BUF = G_::BUF, TILE = G_::TILE;
    extern __shared__ __attribute__((aligned(16))) double smem[];
    double* buf0 = smem;                 // [2][BUF] chunk buffers
    double* xs0 = smem + 2 * BUF;        // [2][N][4] X of the current / next tile
    double* jred = xs0 + 2 * 4 * N;      // [8]
    double* wlds = jred + 16;            // [NCH][LPT][M4_THREADS] double2: the Dtot' table in the staging order
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int w4 = wave & 3, hp = wave >> 2;
    const int64_t n2 = (int64_t)N * N;

    int64_t T = t_begin + (int64_t)blockIdx.x * L;
    const int64_t T_end = min(t_end, T + L);
    if (T >= T_end) return;  // uniform for the whole workgroup
    const int p_first = m4_tri_row(T);
    int p = p_first, q = (int)(T - (int64_t)p * (p + 1) / 2);

    // the Dtot' table (the same for every tile, 96 KB) lives in LDS: one workgroup per CU leaves the room, and the
    // registers it would take (48 per thread) are what the walk needs to run two waves per SIMD without spilling
#pragma unroll
    for (int k = 0; k < M4_NCH; ++k)
#pragma unroll
        for (int s = 0; s < LPT; ++s) {
            const int o = 2 * ((k * LPT + s) * M4_THREADS + tid);
            *reinterpret_cast<double2*>(wlds + o) = *reinterpret_cast<const double2*>(wtab + o);
        }

    // X of a tile: xs[n][c] = D^{c / 2}[c & 1 ? p : q][n]  (NDM = 1: columns 2, 3 are zero)
    constexpr int XU = (4 * N + M4_THREADS - 1) / M4_THREADS;
    auto fetch_x = [&](int pp, int qq, double (&v)[XU]) {
#pragma unroll
        for (int u = 0; u < XU; ++u) {
            const int e = tid + M4_THREADS * u;  // element e = 4 n + c
            const int n = e >> 2, cc = e & 3, x = cc >> 1;
            const int off = x * N * N + ((cc & 1) ? pp : qq) * N + n;  // (32-bit: scalar base + one offset register)
            v[u] = (e < 4 * N && x < NDM) ? dm[off] : 0.0;
        }
    };
    auto store_x = [&](double* xs, const double (&v)[XU]) {
#pragma unroll
        for (int u = 0; u < XU; ++u) {
            const int e = tid + M4_THREADS * u;
            if (e < 4 * N) xs[e] = v[u];
        }
    };
    {
        double v[XU];
        fetch_x(p, q, v);
        store_x(xs0, v);
    }

    // staging: chunk k of the tile at `tile` -> the k-th register set (clamped tail: the weights of those slots are
    // zero).  A whole tile is in flight: set k is reloaded with chunk k of the NEXT tile as soon as it has been written
    // to LDS (one workgroup per CU has to keep ~90 KB in flight to cover the HBM latency at 25 GB/s per CU).
    double2 r[M4_NCH][LPT];
    // `real` = false: the same number of loads from one cache line of the tile (the last tile of the range has nothing
    // to prefetch; a CONDITIONAL reload would make the compiler count the loads in flight for the path without it, and
    // every wait in the steady state would drain a chunk too many)
    auto load_chunk = [&](const double* tile, int k, bool real) {
        const int begin = 16 * m4_tri(G_::row0(k)), end = 16 * m4_tri(G_::row0(k + 1));
#pragma unroll
        for (int s = 0; s < LPT; ++s) {
            int d = begin + (s * M4_THREADS + tid) * 2;
            d = min(d, end - 2);
            r[k][s] = m4_ldnt(tile + (real ? d : 2 * (lane & 3)));
        }
    };
    const double* tile = packed + (T - t_begin) * (int64_t)TILE;
#pragma unroll
    for (int k = 0; k < M4_NCH; ++k) {
        load_chunk(tile, k, true);
        // issued in chunk order, as the loop re-issues them: the wait before chunk k's first use is then the same count
        // on the entry path and on the back edge (vmcnt is in order; a first fill in another order makes it 0 for good)
        asm volatile("" : : : "memory");
    }

    double acc[NG];
#pragma unroll
    for (int g = 0; g < NG; ++g) acc[g] = 0.0;
    double jacc = 0.0;
    int par = 0;  // parity of the tile: which X buffer is current

    for (; T < T_end; ++T) {
        const double* xs = xs0 + par * 4 * N;
        // next tile's indices and X values (in flight during this tile)
        int pn = p, qn = q + 1;
        if (qn > pn) {
            ++pn;
            qn = 0;
        }
        const bool more = T + 1 < T_end;
        double xv[XU];
        fetch_x(more ? pn : p, more ? qn : q, xv);  // (unconditional, as the chunk reloads)
#pragma unroll
        for (int k = 0; k < M4_NCH; ++k) {
            double* buf = buf0 + (k & 1) * BUF;
            // chunk k: registers -> LDS, its J contribution, then the registers take the next chunk
#pragma unroll
            for (int s = 0; s < LPT; ++s) *reinterpret_cast<double2*>(buf + (s * M4_THREADS + tid) * 2) = r[k][s];
#pragma unroll
            for (int s = 0; s < LPT; ++s) {
#ifndef NBX_M4_NO_J
                const double2 w = *reinterpret_cast<const double2*>(wlds + 2 * ((k * LPT + s) * M4_THREADS + tid));
                jacc = fma(r[k][s].y, w.y, fma(r[k][s].x, w.x, jacc));
#endif
            }
            // the register set is dead from here: pinned, so that its reload lands in the SAME registers (a J product
            // sunk below the loads keeps the old values alive, the loads then get other registers, and the copy back
            // at the end of the tile costs a vmcnt(0) -- the whole prefetch drained once per tile)
            asm volatile("" : "+v"(jacc) : : "memory");
            load_chunk(more ? tile + TILE : tile, k, more);
            __syncthreads();  // chunk k is in LDS; everyone is done with the buffer the next chunk will take
#ifndef NBX_M4_NO_WALK
            // (the lane is made opaque per chunk: the ~100 per-lane LDS addresses of a walk are loop invariant, and
            // hoisted out of the tile loop they would occupy -- spill -- a register each)
            int lane_o = lane;
            asm volatile("" : "+v"(lane_o));
            const int a = lane_o >> 4, b = (lane_o >> 2) & 3, c = lane_o & 3;
            if (hp == 0) {
                if (k == 0) m4_walk_chunk<NB, 0, 0>(buf, xs, w4, a, b, c, acc);
                else if (k == 1) m4_walk_chunk<NB, 1, 0>(buf, xs, w4, a, b, c, acc);
                else if (k == 2) m4_walk_chunk<NB, 2, 0>(buf, xs, w4, a, b, c, acc);
                else m4_walk_chunk<NB, 3, 0>(buf, xs, w4, a, b, c, acc);
            } else {
                if (k == 0) m4_walk_chunk<NB, 0, 1>(buf, xs, w4, a, b, c, acc);
                else if (k == 1) m4_walk_chunk<NB, 1, 1>(buf, xs, w4, a, b, c, acc);
                else if (k == 2) m4_walk_chunk<NB, 2, 1>(buf, xs, w4, a, b, c, acc);
                else m4_walk_chunk<NB, 3, 1>(buf, xs, w4, a, b, c, acc);
            }
#endif
        }
        // ---- end of tile: J[p][q], and the row-q halves of the partial rows (odd columns c: they used D[p][:]) leave the
        // registers -- the eight waves' partials are summed through LDS into the tile's row-q partial; the row-p halves
        // (even c) stay in the registers until the row changes.  Scratch: the buffer of chunk 2 (walked by everyone,
        // chunk 0 of the next tile is written there behind the second barrier).
#ifndef NBX_M4_NO_EPI
        double* red = buf0;  // [8][NG][32]
        static_assert(M4_WAVES * NG * 32 <= BUF, "reduction scratch fits one chunk buffer");
        const bool odd = lane & 1;
        jacc = nbx_wave_sum(jacc);
        if (lane == 0) jred[wave] = jacc;
        jacc = 0.0;
#pragma unroll
        for (int g = 0; g < NG; ++g) {
            if (odd) red[(wave * NG + g) * 32 + (lane >> 1)] = acc[g];
            acc[g] = odd ? 0.0 : acc[g];
        }
        if (more) store_x(xs0 + (par ^ 1) * 4 * N, xv);
        __syncthreads();
        if (tid == 0) {
            double j = 0.0;
#pragma unroll
            for (int w = 0; w < M4_WAVES; ++w) j += jred[w];
            jfull[(int64_t)p * N + q] = j;
            jfull[(int64_t)q * N + p] = j;
        }
        if (tid < NG * 32 && q < p) {
            const int g = tid >> 5, l = 2 * (tid & 31) + 1;
            const int row = 16 * g + 4 * ((l >> 2) & 3) + (l >> 4), x = (l & 3) >> 1;
            if (row < N && x < NDM) {
                double v = 0.0;
#pragma unroll
                for (int w = 0; w < M4_WAVES; ++w) v += red[w * NG * 32 + tid];
                kpart2[((T - t_begin) * NDM + x) * (int64_t)N + row] = v;
            }
        }
        __syncthreads();
        if (pn != p || !more) {  // the row is complete for this workgroup: its row-p partial (even c) -> kpart1 slot
#pragma unroll
            for (int g = 0; g < NG; ++g) {
                if (!odd) red[(wave * NG + g) * 32 + (lane >> 1)] = acc[g];
                acc[g] = 0.0;
            }
            __syncthreads();
            if (tid < NG * 32) {
                const int g = tid >> 5, l = 2 * (tid & 31);
                const int row = 16 * g + 4 * ((l >> 2) & 3) + (l >> 4), x = (l & 3) >> 1;
                if (row < N && x < NDM) {
                    double v = 0.0;
#pragma unroll
                    for (int w = 0; w < M4_WAVES; ++w) v += red[w * NG * 32 + tid];
                    kpart1[(((int64_t)blockIdx.x * S + (p - p_first)) * NDM + x) * N + row] = v;
                }
            }
            __syncthreads();
        }
#else
        if (more) store_x(xs0 + (par ^ 1) * 4 * N, xv);
        __syncthreads();
#endif
        p = pn;
        q = qn;
        par ^= 1;
        tile += TILE;
    }
}

size_t m4_align256(size_t x) { return (x + 255) & ~(size_t)255; }

struct M4Plan {
    int wgs, L, S;
    size_t wt_off, k1_off, k2_off, total, lds_bytes;
};

template <int NB>
M4Plan m4_plan_nb(int64_t p0, int64_t np, int64_t ndm) {
    using G = M4Geom<NB>;
    M4Plan pl;
    const int64_t ntiles = m4_tri((int)(p0 + np)) - m4_tri((int)p0);
    const int64_t slots = M4_CUS * M4_PER_CU;
    int64_t L = nbx_cdiv(ntiles, slots);
    if (L < 1) L = 1;
    pl.L = (int)L;
    pl.wgs = (int)nbx_cdiv(ntiles, L);
    pl.S = (int)sqrt(2.0 * (double)L) + 3;
    pl.lds_bytes = (size_t)(2 * G::BUF + 2 * 4 * G::N + 16 + M4_NCH * G::LPT * M4_THREADS * 2) * sizeof(double);
    size_t off = 0;
    pl.wt_off = off; off += m4_align256((size_t)(M4_NCH * G::LPT * M4_THREADS * 2) * sizeof(double));
    pl.k1_off = off; off += m4_align256((size_t)((int64_t)pl.wgs * pl.S * ndm * G::N) * sizeof(double));
    pl.k2_off = off; off += m4_align256((size_t)(ntiles * ndm * G::N) * sizeof(double));
    pl.total = off;
    return pl;
}

template <int NB>
int m4_run(nbx_ctx* ctx, int64_t p0, int64_t p1, const double* d_packed, const double* d_dm, int64_t ndm, double* d_jk,
           void* d_work, const double* d_hv, double* d_fock, double* d_vhf) {
    using G = M4Geom<NB>;
    const int64_t np = p1 - p0, N = G::N, n2 = N * N;
    const M4Plan pl = m4_plan_nb<NB>(p0, np, ndm);
    char* base = static_cast<char*>(d_work);
    double* wt = reinterpret_cast<double*>(base + pl.wt_off);
    double* k1 = reinterpret_cast<double*>(base + pl.k1_off);
    double* k2 = reinterpret_cast<double*>(base + pl.k2_off);
    if (np < N) {  // J entries this slab does not own must read as zero
        const int rc = nbx_memset(ctx, d_jk, 0, (size_t)n2 * sizeof(double));
        if (rc != NBX_OK) return rc;
    }
    hipLaunchKernelGGL(m4_weights_kernel<NB>, dim3((unsigned)nbx_cdiv(M4_NCH * G::LPT * M4_THREADS, 256)), dim3(256), 0, ctx->stream,
                       d_dm, (int)ndm, wt);
    NBX_LAUNCH_CHECK();
    const int64_t t_begin = m4_tri((int)p0), t_end = m4_tri((int)p1);
    {
        nbx_prof_scope prof(ctx, NBX_PROF_JK_DENSE);
        static bool attr_set = false;
        if (!attr_set) {
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&jk_m4_kernel<NB, 1>),
                                      hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&jk_m4_kernel<NB, 2>),
                                      hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
            attr_set = true;
        }
        if (ndm == 2)
            hipLaunchKernelGGL((jk_m4_kernel<NB, 2>), dim3((unsigned)pl.wgs), dim3(M4_THREADS), pl.lds_bytes, ctx->stream,
                               d_packed, d_dm, wt, d_jk, k1, k2, t_begin, t_end, pl.L, pl.S);
        else
            hipLaunchKernelGGL((jk_m4_kernel<NB, 1>), dim3((unsigned)pl.wgs), dim3(M4_THREADS), pl.lds_bytes, ctx->stream,
                               d_packed, d_dm, wt, d_jk, k1, k2, t_begin, t_end, pl.L, pl.S);
    }
    NBX_LAUNCH_CHECK();
    return nbx_jk_sym_reduce(ctx, k1, k2, d_jk + n2, N, p0, np, ndm, t_begin, pl.L, pl.S, d_jk, d_hv, d_fock, d_vhf, 1);
}

}  // namespace

// The sizes this kernel has an instance for (N = 4 NB).  Opt-in while it is being tuned: NBX_JK_M4=1 in the environment
// (read once per process); otherwise jk_s4.hip serves every size.
bool nbx_jk_m4_covers(int64_t N) {
    static const bool on = getenv("NBX_JK_M4") != nullptr && atoi(getenv("NBX_JK_M4")) != 0;
    return on && N == 148;
}

size_t nbx_jk_m4_packed_bytes(int64_t N, int64_t p0, int64_t p1) {
    const int64_t ntiles = m4_tri((int)p1) - m4_tri((int)p0);
    return (size_t)(ntiles * M4Geom<37>::TILE) * sizeof(double) + 256;  // (+ slack: the staging of the last tile prefetches nothing beyond)
}

size_t nbx_jk_m4_worksize(int64_t N, int64_t p0, int64_t p1, int64_t ndm) { return m4_plan_nb<37>(p0, p1 - p0, ndm).total; }

int nbx_jk_m4_pack(nbx_ctx* ctx, int64_t N, int64_t p0, int64_t p1, const double* d_eri, double* d_packed) {
    NBX_CHECK_ARG(nbx_jk_m4_covers(N) && d_eri && d_packed);
    const int64_t ntiles = m4_tri((int)p1) - m4_tri((int)p0);
    hipLaunchKernelGGL(m4_pack_kernel<37>, dim3((unsigned)ntiles), dim3(256), 0, ctx->stream, d_eri, d_packed, (int)p0,
                       (int64_t)m4_tri((int)p0));
    NBX_LAUNCH_CHECK();
    return NBX_OK;
}

int nbx_jk_m4(nbx_ctx* ctx, int64_t N, int64_t p0, int64_t p1, const double* d_packed, const double* d_dm, int64_t ndm,
              double* d_jk, void* d_work, const double* d_hv, double* d_fock, double* d_vhf) {
    NBX_CHECK_ARG(nbx_jk_m4_covers(N));
    return m4_run<37>(ctx, p0, p1, d_packed, d_dm, ndm, d_jk, d_work, d_hv, d_fock, d_vhf);
}
