// libnbx: batched fp64 GEMM on the CDNA4 matrix cores (include/nbx.h "dense products").
//
//   C[b] = alpha * op(A[b]) * op(B[b]) + beta * C[b]      row-major, op in {N, T}
//
// v_mfma_f64_16x16x4_f64 (one wave: 16x16 tile, K = 4):
//   A operand: lane l holds A[i = l & 15][k = l >> 4]
//   B operand: lane l holds B[k = l >> 4][j = l & 15]
//   C/D      : 4 doubles per lane, reg r -> row (l >> 4) + 4 r, col l & 15
// Workgroup = WR x WC waves (2 x 4 for the 128 x 128 tile: 64 accumulator + ~70 other registers
// per lane, so two workgroups = 4 waves/SIMD fit a CU -- one wave/SIMD cannot saturate the fp64
// matrix pipe, see profiles/r01/mfma_f64_peak.txt); block tile BM x BN x 16 staged through LDS as
// As[k][m] / Bs[k][n] with row stride (BM|BN) + 16 doubles so that the two
// 16-lane groups a ds_read_b64 services together fall in different halves of the
// 256-byte bank row (conflict-free fragment reads).  The next k-tile is fetched into
// registers while the current one feeds the MFMAs.
#include "nbx_common.h"
#include "synth_device.h"

// (m0 is named as a clobber of the LDS-DMA asm in gemm_m4_tn_kernel; clang calls that a reserved register)
#pragma clang diagnostic ignored "-Winline-asm"

namespace {

typedef double v4f64 __attribute__((ext_vector_type(4)));

// B operand produced by the counter hash instead of loaded (streamed quarter-1 of the synthetic
// (pq|rs), ao2mo_synth.hip): batch entry z is the (p,q) matrix of the pair (r, s0 + z).
struct GemmGen {
    uint64_t seed;
    double scale;
    int r, s0;
    // "triangular batch" (tri_unit > 0): batch entry b has (tri_m0 + b + 1) * tri_unit rows and its
    // C block follows those of the entries before it -- the (i, j <= i) pairs of a symmetric
    // transform, stored compactly (ao2mo.hip)
    int tri_unit = 0, tri_m0 = 0;
    // "pair scatter" (pair_n > 0): batch entry b is the pair (i, j <= i), b = i (i+1)/2 + j; its C
    // block (stride_c doubles) is stored at block index i * pair_n + j AND at j * pair_n + i
    int pair_n = 0;
    // "symmetric-packed A" (a_symn = N > 0): row x of op(A) is row r = x % N of the symmetric N x N
    // matrix number x / N, stored as its packed lower triangle (N(N+1)/2 doubles per matrix);
    // element (r, k) is read from offset T(max(r,k), min(r,k)).  A 'N' operand only.
    int a_symn = 0;
};

constexpr int BK = 16;
constexpr int PAD = 16;

// Stage one operand tile T[k][x] (k < BK, x < BX) where element (x, k) of op(.) lives at
// base[x * sx + k * sk].  KCONTIG: memory is contiguous along k (sk == 1), else along x (sx == 1).
template <int BX, bool KCONTIG, int GEMM_THREADS>
struct Stager {
    static constexpr int PAIRS = BK * BX / 2;            // double2 items per tile
    static constexpr int PER_THREAD = PAIRS / GEMM_THREADS;
    static_assert(PAIRS % GEMM_THREADS == 0, "tile must divide evenly");
    double2 reg[PER_THREAD];

    __device__ __forceinline__ void load(const double* __restrict__ base, int64_t ld, int x0, int k0,
                                         int xmax, int kmax, bool vec_ok) {
#pragma unroll
        for (int i = 0; i < PER_THREAD; ++i) {
            const int item = threadIdx.x + i * GEMM_THREADS;
            int x, k;
            if (KCONTIG) {  // pairs along k: BK/2 pairs per x
                x = item / (BK / 2);
                k = (item - x * (BK / 2)) * 2;
            } else {  // pairs along x: BX/2 pairs per k
                k = item / (BX / 2);
                x = (item - k * (BX / 2)) * 2;
            }
            const int gx = x0 + x, gk = k0 + k;
            double2 v = make_double2(0.0, 0.0);
            if (KCONTIG) {
                if (gx < xmax) {
                    const double* p = base + (int64_t)gx * ld + gk;
                    if (vec_ok && gk + 1 < kmax) {
                        v = *reinterpret_cast<const double2*>(p);
                    } else {
                        if (gk < kmax) v.x = p[0];
                        if (gk + 1 < kmax) v.y = p[1];
                    }
                }
            } else {
                if (gk < kmax) {
                    const double* p = base + (int64_t)gk * ld + gx;
                    if (vec_ok && gx + 1 < xmax) {
                        v = *reinterpret_cast<const double2*>(p);
                    } else {
                        if (gx < xmax) v.x = p[0];
                        if (gx + 1 < xmax) v.y = p[1];
                    }
                }
            }
            reg[i] = v;
        }
    }

    // KCONTIG operand whose rows are rows of packed symmetric matrices (GemmGen::a_symn)
    __device__ __forceinline__ void load_sym(const double* __restrict__ base, int n, int x0, int k0, int xmax,
                                             int kmax) {
        static_assert(KCONTIG || BX > 0, "");
        const int64_t nt = (int64_t)n * (n + 1) / 2;
#pragma unroll
        for (int i = 0; i < PER_THREAD; ++i) {
            const int item = threadIdx.x + i * GEMM_THREADS;
            const int x = item / (BK / 2);
            const int k = (item - x * (BK / 2)) * 2;
            const int gx = x0 + x, gk = k0 + k;
            double2 v = make_double2(0.0, 0.0);
            if (gx < xmax) {
                const int mat = gx / n, r = gx - mat * n;
                const double* p = base + mat * nt;
                auto at = [&](int kk) {
                    const int hi = r > kk ? r : kk, lo = r > kk ? kk : r;
                    return p[(int64_t)hi * (hi + 1) / 2 + lo];
                };
                if (gk < kmax) v.x = at(gk);
                if (gk + 1 < kmax) v.y = at(gk + 1);
            }
            reg[i] = v;
        }
    }

    // same register tile, values generated: element (x, k) = val(canon(k, x, r, s)) * scale
    __device__ __forceinline__ void generate(uint64_t rs, uint64_t seed, double scale, int x0, int k0, int xmax,
                                             int kmax) {
        static_assert(!KCONTIG, "generated operand is laid out along x");
#pragma unroll
        for (int i = 0; i < PER_THREAD; ++i) {
            const int item = threadIdx.x + i * GEMM_THREADS;
            const int k = item / (BX / 2);
            const int x = (item - k * (BX / 2)) * 2;
            const int gx = x0 + x, gk = k0 + k;
            double2 v = make_double2(0.0, 0.0);
            if (gk < kmax) {
                // pair indices fit 32 bits (N < 92681): 32 x 32 -> 64-bit products instead of 64 x 64
                if (gx < xmax)
                    v.x = nbx_synth_val(0, nbx_tri_u32(nbx_tri_pair_u32((uint32_t)gk, (uint32_t)gx), (uint32_t)rs), seed) * scale;
                if (gx + 1 < xmax)
                    v.y = nbx_synth_val(0, nbx_tri_u32(nbx_tri_pair_u32((uint32_t)gk, (uint32_t)(gx + 1)), (uint32_t)rs), seed) *
                          scale;
            }
            reg[i] = v;
        }
    }

    __device__ __forceinline__ void store(double* __restrict__ tile /* [BK][BX+PAD] */) const {
#pragma unroll
        for (int i = 0; i < PER_THREAD; ++i) {
            const int item = threadIdx.x + i * GEMM_THREADS;
            if (KCONTIG) {
                const int x = item / (BK / 2);
                const int k = (item - x * (BK / 2)) * 2;
                tile[k * (BX + PAD) + x] = reg[i].x;
                tile[(k + 1) * (BX + PAD) + x] = reg[i].y;
            } else {
                const int k = item / (BX / 2);
                const int x = (item - k * (BX / 2)) * 2;
                *reinterpret_cast<double2*>(&tile[k * (BX + PAD) + x]) = reg[i];
            }
        }
    }
};

template <int BM, int BN, int WR, int WC, bool A_KC, bool B_KC, bool B_GEN = false>
__global__ __launch_bounds__(64 * WR * WC) void gemm_f64_kernel(
    int M, int N, int K, double alpha, const double* __restrict__ A, int64_t lda, int64_t stride_a,
    const double* __restrict__ B, int64_t ldb, int64_t stride_b, double beta, double* __restrict__ C,
    int64_t ldc, int64_t stride_c, int vec_a, int vec_b, GemmGen gen = GemmGen{},
    const int* __restrict__ gate = nullptr, int gate_a = 0, int gate_b = 0) {
    constexpr int GEMM_THREADS = 64 * WR * WC;
    constexpr int WM = BM / WR, WN = BN / WC;  // WR x WC waves
    constexpr int MT = WM / 16, NT = WN / 16;
    __shared__ __attribute__((aligned(16))) double As[BK * (BM + PAD)];
    __shared__ __attribute__((aligned(16))) double Bs[BK * (BN + PAD)];

    const int batch = blockIdx.z;
    if (gate != nullptr) {  // device-side "run only if" (nbx_gemm_gated)
        const int g = gate[batch];
        if (g != gate_a && g != gate_b) return;
    }
    A += (int64_t)batch * stride_a;
    if (!B_GEN) B += (int64_t)batch * stride_b;
    C += (int64_t)batch * stride_c;
    double* C2 = nullptr;
    if (gen.pair_n > 0) {
        int pi = (int)((sqrt(8.0 * (double)batch + 1.0) - 1.0) * 0.5);
        while (pi * (pi + 1) / 2 > batch) --pi;
        while ((pi + 1) * (pi + 2) / 2 <= batch) ++pi;
        const int pj = batch - pi * (pi + 1) / 2;
        C += ((int64_t)pi * gen.pair_n + pj - batch) * stride_c;  // (C already points at block `batch`)
        if (pi != pj) C2 = C + ((int64_t)pj * gen.pair_n + pi - ((int64_t)pi * gen.pair_n + pj)) * stride_c;
    }
    if (gen.tri_unit > 0) {
        const int64_t gi = gen.tri_m0 + batch, g0 = gen.tri_m0;
        M = min(M, (int)((gi + 1) * gen.tri_unit));
        C += (gi * (gi + 1) / 2 - g0 * (g0 + 1) / 2) * gen.tri_unit * ldc;
        if ((int)blockIdx.y * BM >= M) return;  // uniform for the workgroup
    }
    const uint64_t gen_rs = B_GEN ? nbx_tri((uint64_t)gen.r, (uint64_t)(gen.s0 + batch)) : 0;
    const int m0 = blockIdx.y * BM;
    const int n0 = blockIdx.x * BN;

    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const int wr = wave / WC, wc = wave % WC;
    const int fr = lane & 15;   // index inside the 16-wide fragment
    const int fk = lane >> 4;   // k (A/B) or row group (C)

    v4f64 acc[MT][NT];
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j) acc[i][j] = (v4f64){0.0, 0.0, 0.0, 0.0};

    Stager<BM, A_KC, GEMM_THREADS> sa;
    Stager<BN, B_KC, GEMM_THREADS> sb;
    const int nkt = (K + BK - 1) / BK;
    if (A_KC && gen.a_symn > 0) sa.load_sym(A, gen.a_symn, m0, 0, M, K);
    else sa.load(A, lda, m0, 0, M, K, vec_a);
    if constexpr (B_GEN) sb.generate(gen_rs, gen.seed, gen.scale, n0, 0, N, K);
    else sb.load(B, ldb, n0, 0, N, K, vec_b);
    sa.store(As);
    sb.store(Bs);
    __syncthreads();

    for (int kt = 0; kt < nkt; ++kt) {
        const bool more = kt + 1 < nkt;
        if (more) {
            if (A_KC && gen.a_symn > 0) sa.load_sym(A, gen.a_symn, m0, (kt + 1) * BK, M, K);
            else sa.load(A, lda, m0, (kt + 1) * BK, M, K, vec_a);
            if constexpr (B_GEN) sb.generate(gen_rs, gen.seed, gen.scale, n0, (kt + 1) * BK, N, K);
            else sb.load(B, ldb, n0, (kt + 1) * BK, N, K, vec_b);
        }
        // (the last k-tile of K = 148 holds 4 of its 16 columns: MFMA steps that would multiply the zero padding
        // are skipped -- 7.5 % of the matrix-pipe work of every N = 148 product)
        const int ksteps = more ? BK / 4 : (K - kt * BK + 3) / 4;
#pragma unroll
        for (int kk = 0; kk < BK / 4; ++kk) {
            if (kk >= ksteps) break;  // uniform
            double af[MT], bf[NT];
            const int krow = kk * 4 + fk;
#pragma unroll
            for (int i = 0; i < MT; ++i) af[i] = As[krow * (BM + PAD) + wr * WM + i * 16 + fr];
#pragma unroll
            for (int j = 0; j < NT; ++j) bf[j] = Bs[krow * (BN + PAD) + wc * WN + j * 16 + fr];
#pragma unroll
            for (int i = 0; i < MT; ++i)
#pragma unroll
                for (int j = 0; j < NT; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(af[i], bf[j], acc[i][j], 0, 0, 0);
        }
        __syncthreads();
        if (more) {
            sa.store(As);
            sb.store(Bs);
            __syncthreads();
        }
    }

    // epilogue: lane holds rows fk + 4 r (r < 4), column fr of each 16 x 16 tile
#pragma unroll
    for (int i = 0; i < MT; ++i) {
#pragma unroll
        for (int j = 0; j < NT; ++j) {
            const int col = n0 + wc * WN + j * 16 + fr;
            if (col >= N) continue;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = m0 + wr * WM + i * 16 + fk + 4 * r;
                if (row < M) {
                    double* c = C + (int64_t)row * ldc + col;
                    const double v = alpha * acc[i][j][r];
                    *c = (beta == 0.0) ? v : fma(beta, *c, v);
                    if (C2 != nullptr) C2[(int64_t)row * ldc + col] = v;  // pair scatter (beta = 0)
                }
            }
        }
    }
}

// ---- C = alpha A^T B + beta C for operands that are both contiguous along the non-contracted index (A: K x M,
// B: K x N row-major: quarters 1, 2 and 4 of the transform), on the instruction that reaches the fp64 matrix peak of
// gfx950 and with the k-tiles brought into LDS by the load unit itself.
//   * v_mfma_f64_4x4x4_4b_f64 (four independent 4 x 4 x 4 blocks per instruction) issues once per 16.5 cycles per SIMD:
//     75 TFLOP/s on the chip with registers only, where v_mfma_f64_16x16x4_f64 issues once per 104-143 cycles (38-48
//     TFLOP/s; profiles/r03/fp64_rate_probe.txt).  The four blocks are four groups of four ROWS against the same four
//     columns: the A operand is the 16-row fragment the big instruction takes (lane l: row l & 15, k = l >> 4), the B
//     operand is B[k][col] replicated over the row groups (the broadcast controls CBSZ / ABID are ignored by this
//     instruction: profiles/r03/mfma_f64_4x4x4_cbsz_probe.hip), and a lane reads the four adjacent columns
//     4 (l & 3) .. + 3 of a 16-column fragment with one 32-byte LDS read: MFMA t of a fragment multiplies column
//     4 (l & 3) + t, so the lane ends up with C[row 4 b + (l >> 4)][4 (l & 3) .. + 3] -- 32 contiguous bytes to store.
//     Four waves of 128 x 32 outputs each (128 accumulator registers; per 4-row step 8 A fragments and 2 x 32 bytes of
//     B per lane: 8 LDS instructions for 64 MFMAs, where 64 x 64 per wave took 12), two workgroups per CU.
//   * global_load_lds_dwordx4: lane l of a wave lands its 16 bytes at the instruction's LDS base + 16 l -- one
//     instruction is one 128-double row of a tile, no staging registers, no ds_write, and the loads of three tiles
//     are in flight while a fourth feeds the matrix pipe (a ring of four 8-row tiles per operand, 72 KB).  The
//     register-staged kernel above holds ONE tile in flight and waits for it every 16 rows.
//   * every load instruction is always issued (rows past K re-read row K - 1, lanes past the edge re-read column 0
//     of their row: what they bring is never multiplied into a stored element), so a wave's tile is exactly four
//     instructions and `s_waitcnt vmcnt(8)` means "my part of the tile two behind the newest has landed".
//   * the fragments of a 4-row step are read from LDS while the step before it runs (two register sets), across
//     tiles too: the barrier that publishes tile kt + 1 sits between the two steps of tile kt.
// Needs K % 4 == 0 (MFMA steps past K are skipped, not zero-filled), M and N even, 16-byte aligned rows.
// B_GEN: the B operand is not loaded but generated (GemmGen: the streamed quarter-1 of the synthetic tensor): every
// thread makes its four values of tile kt + 3 and stores them into the ring while tile kt is on the matrix pipe.
// Where the time goes (tools/build_gemm_variant.sh builds with NBX_TN_DBG, tools/time_gemm_q1.py; M = 128):
//   K = 2000:  50.7 TFLOP/s as is (the 16 x 16 x 4 kernel: 50.4 on the same box, 46.9 with the small instruction and
//              register staging); 53.5 without the loads in the loop; 55.2 without the barrier too; 62.8 without the
//              LDS reads (12 instructions per 64 MFMAs: the B fragments are read four times over).
//   K = 148:   43.7 as is (16 x 16 x 4 kernel: 40.4); 50.0 when nothing is stored; 55.8 when, in addition, the rows
//              come from cache -- the product moves (K + M) / (2 K M) = 1 byte per 17 flops, 2.5 TB/s of reads AND
//              writes at that rate: both rooflines at once.
#ifndef NBX_TN_DBG
#define NBX_TN_DBG 0  // measurement builds, bits: 1 no loads in the loop, 2 no barrier, 4 no LDS reads in the loop,
#endif                // 8 nothing stored, 16 the same rows loaded again and again
constexpr int DK = 8, DNB = 4, DLD = 128 + 16;
typedef __attribute__((address_space(3))) void* gemm_lds_vp;

template <bool B_GEN>
__global__ __launch_bounds__(256, 2) void gemm_m4_tn_kernel(int M, int N, int K, double alpha, const double* __restrict__ A,
                                                            int64_t lda, int64_t stride_a, const double* __restrict__ B,
                                                            int64_t ldb, int64_t stride_b, double beta,
                                                            double* __restrict__ C, int64_t ldc, int64_t stride_c,
                                                            int pair_n, const int* __restrict__ gate, int gate_a,
                                                            int gate_b, GemmGen gen = GemmGen{}) {
    __shared__ __attribute__((aligned(32))) double smem[DNB * 2 * DK * DLD];
    const int batch = blockIdx.z;
    if (gate != nullptr) {
        const int g = gate[batch];
        if (g != gate_a && g != gate_b) return;
    }
    A += (int64_t)batch * stride_a;
    if (!B_GEN) B += (int64_t)batch * stride_b;
    C += (int64_t)batch * stride_c;
    const uint64_t gen_rs = B_GEN ? nbx_tri((uint64_t)gen.r, (uint64_t)(gen.s0 + batch)) : 0;
    double* C2 = nullptr;
    if (pair_n > 0) {  // pair scatter: see GemmGen::pair_n
        int pi = (int)((sqrt(8.0 * (double)batch + 1.0) - 1.0) * 0.5);
        while (pi * (pi + 1) / 2 > batch) --pi;
        while ((pi + 1) * (pi + 2) / 2 <= batch) ++pi;
        const int pj = batch - pi * (pi + 1) / 2;
        C += ((int64_t)pi * pair_n + pj - batch) * stride_c;
        if (pi != pj) C2 = C + ((int64_t)pj * pair_n + pi - ((int64_t)pi * pair_n + pj)) * stride_c;
    }
    const int m0 = blockIdx.y * 128, n0 = blockIdx.x * 128;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int wc = wave;  // four waves side by side: 128 rows x 32 columns each (TM x TN fragments of 16)
    constexpr int TM = 8, TN = 2;
    const int fr = lane & 15, fk = lane >> 4;

    // this lane's 16 bytes of a row: doubles 2 lane, 2 lane + 1 of the tile (column 0 of the row past the edge)
    const unsigned a_off = (m0 + 2 * lane < M) ? 16u * (unsigned)lane : 0u;
    const unsigned b_off = (n0 + 2 * lane < N) ? 16u * (unsigned)lane : 0u;
    const int nkt = (K + DK - 1) / DK;
    auto issue = [&](int kt) {  // rows 2 wave, 2 wave + 1 of both operands' tile kt into ring slot kt % DNB
        const int slot = kt & (DNB - 1);
#pragma unroll
        for (int r = 0; r < 2; ++r) {
            const int kl = 2 * wave + r;
            int k = ((NBX_TN_DBG & 16) ? 0 : kt * DK) + kl;
            k = k < K ? k : K - 1;
            const double* asrc = A + (int64_t)k * lda + m0;
            const unsigned a_lds = (unsigned)(size_t)(gemm_lds_vp)(smem + (slot * 2 + 0) * (DK * DLD) + kl * DLD);
            asm volatile("s_mov_b32 m0, %2\n\tglobal_load_lds_dwordx4 %0, %1" : : "v"(a_off), "s"(asrc), "s"(a_lds) : "memory", "m0");
            if constexpr (!B_GEN) {
                const double* bsrc = B + (int64_t)k * ldb + n0;
                const unsigned b_lds = (unsigned)(size_t)(gemm_lds_vp)(smem + (slot * 2 + 1) * (DK * DLD) + kl * DLD);
                asm volatile("s_mov_b32 m0, %2\n\tglobal_load_lds_dwordx4 %0, %1" : : "v"(b_off), "s"(bsrc), "s"(b_lds) : "memory", "m0");
            }
        }
        if constexpr (B_GEN) {
            // element (k, x) = val(canon(k, x, r, s)) * scale, two adjacent x per item, two items per thread
            double* bt = smem + (slot * 2 + 1) * (DK * DLD);
#pragma unroll
            for (int it = 0; it < (DK * 128 / 2) / 256; ++it) {
                const int item = threadIdx.x + it * 256;
                const int kl = item >> 6, x = (item & 63) * 2;
                const int gk = kt * DK + kl, gx = n0 + x;
                double2 v = make_double2(0.0, 0.0);
                if (gk < K) {
                    if (gx < N)
                        v.x = nbx_synth_val(0, nbx_tri_u32(nbx_tri_pair_u32((uint32_t)gk, (uint32_t)gx), (uint32_t)gen_rs), gen.seed) * gen.scale;
                    if (gx + 1 < N)
                        v.y = nbx_synth_val(0, nbx_tri_u32(nbx_tri_pair_u32((uint32_t)gk, (uint32_t)(gx + 1)), (uint32_t)gen_rs), gen.seed) * gen.scale;
                }
                *reinterpret_cast<double2*>(&bt[kl * DLD + x]) = v;
            }
        }
    };
    // (a wave's tile is four load instructions, two when B is generated: the vmcnt immediates below)
    auto wait_two_behind = [&]() {
        if constexpr (B_GEN) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    };

    v4f64 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) acc[i][j] = (v4f64){0.0, 0.0, 0.0, 0.0};

    auto read_frags = [&](int kt, int kk, double (&af)[TM], v4f64 (&bq)[TN]) {
        const double* as = smem + ((kt & (DNB - 1)) * 2 + 0) * (DK * DLD);
        const double* bs = smem + ((kt & (DNB - 1)) * 2 + 1) * (DK * DLD);
        const int krow = kk * 4 + fk;
#pragma unroll
        for (int i = 0; i < TM; ++i) af[i] = as[krow * DLD + i * 16 + fr];
#pragma unroll
        for (int j = 0; j < TN; ++j) bq[j] = *reinterpret_cast<const v4f64*>(&bs[krow * DLD + wc * 32 + j * 16 + 4 * (lane & 3)]);
    };
    auto products = [&](const double (&af)[TM], const v4f64 (&bq)[TN]) {
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int t = 0; t < 4; ++t)
                    acc[i][j][t] = __builtin_amdgcn_mfma_f64_4x4x4f64(af[i], bq[j][t], acc[i][j][t], 0, 0, 0);
    };
    static_assert(DK == 8, "two steps per tile");
    double af0[TM], af1[TM];
    v4f64 bq0[TN], bq1[TN];
    issue(0);
    issue(1);
    issue(2);
    wait_two_behind();  // my loads of tile 0
    __syncthreads();
    issue(3);
    read_frags(0, 0, af0, bq0);
    if (NBX_TN_DBG & 4) read_frags(0, 1, af1, bq1);
    for (int kt = 0; kt < nkt; ++kt) {
        const bool second = K - kt * DK > 4;  // (the last tile of K = 148 holds one step)
        if (second && !(NBX_TN_DBG & 4)) read_frags(kt, 1, af1, bq1);
        products(af0, bq0);
        // tile kt + 1: my loads of it have landed (tiles kt + 2, kt + 3 may be in flight) ...
        if (!(NBX_TN_DBG & 1)) wait_two_behind();
        if (!(NBX_TN_DBG & 2)) __syncthreads();  // ... and everybody's; nobody reads tile kt - 1 any more: its slot takes tile kt + 4
        if (!(NBX_TN_DBG & 1)) issue(kt + 4);
        if (kt + 1 < nkt && !(NBX_TN_DBG & 4)) read_frags(kt + 1, 0, af0, bq0);
        if (second) products(af1, bq1);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // (the padding loads of the last iterations)

    if ((NBX_TN_DBG & 8) && acc[0][0][0] != 12345.678) return;
    // epilogue: the lane holds row 4 b + fk, columns 4 (lane & 3) .. + 3 of each 16 x 16 tile
    const bool vec_c = ((reinterpret_cast<uintptr_t>(C) | reinterpret_cast<uintptr_t>(C2)) & 15) == 0 && ldc % 2 == 0;
#pragma unroll
    for (int i = 0; i < TM; ++i) {
        const int row = m0 + i * 16 + 4 * ((lane >> 2) & 3) + fk;
        if (row >= M) continue;
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int col = n0 + wc * 32 + j * 16 + 4 * (lane & 3);
            if (col >= N) continue;
            double* c = C + (int64_t)row * ldc + col;
            double v[4];
#pragma unroll
            for (int t = 0; t < 4; ++t) v[t] = alpha * acc[i][j][t];
            if (vec_c && col + 3 < N) {
                if (beta != 0.0) {
#pragma unroll
                    for (int t = 0; t < 4; ++t) v[t] = fma(beta, c[t], v[t]);
                }
                *reinterpret_cast<double2*>(c) = make_double2(v[0], v[1]);
                *reinterpret_cast<double2*>(c + 2) = make_double2(v[2], v[3]);
                if (C2 != nullptr) {  // pair scatter (beta = 0)
                    double* c2 = C2 + (int64_t)row * ldc + col;
                    *reinterpret_cast<double2*>(c2) = make_double2(v[0], v[1]);
                    *reinterpret_cast<double2*>(c2 + 2) = make_double2(v[2], v[3]);
                }
            } else {
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    if (col + t < N) {
                        c[t] = (beta == 0.0) ? v[t] : fma(beta, c[t], v[t]);
                        if (C2 != nullptr) C2[(int64_t)row * ldc + col + t] = v[t];
                    }
                }
            }
        }
    }
}

// Latency-bound small products (the N x N x N GEMMs of an SCF cycle at N ~ 150: 18 workgroups
// of the 64 x 64 kernel, each crawling through K in 16-wide steps at one HBM/L2 latency per
// step = 16 us).  Here one wavefront owns one 16 x 16 tile of C and feeds the MFMA operands
// straight from global memory (the operands are a few hundred KB: L2 resident), no LDS and no
// barriers, so every load of the whole K range can be in flight at once: ~10 x 10 x batch
// single-wave workgroups spread over the chip, one memory latency end to end.
//
// The k-slot mapping and the order of accumulation are those of gemm_f64_kernel (MFMA j takes
// k = 4j .. 4j+3, lane group fk supplies k = 4j + fk), so the two kernels give bitwise identical
// results: which one a product is routed to (it depends on the slab height when the outer index
// is sharded over GPUs) never changes a bit of the answer.
template <bool A_KC, bool B_KC>
__global__ __launch_bounds__(64) void gemm_small_kernel(int M, int N, int K, double alpha,
                                                        const double* __restrict__ A, int64_t lda, int64_t stride_a,
                                                        const double* __restrict__ B, int64_t ldb, int64_t stride_b,
                                                        double beta, double* __restrict__ C, int64_t ldc,
                                                        int64_t stride_c, const int* __restrict__ gate, int gate_a,
                                                        int gate_b, const double* __restrict__ B2,
                                                        double* __restrict__ C2, int split,
                                                        double* __restrict__ norm_part) {
    // z >= split: second product of a pair sharing op(A) (C2 = op(A) op(B2)), same batch entries
    int batch = blockIdx.z;
    const bool second = batch >= split;
    if (second) {
        batch -= split;
        B = B2;
        C = C2;
    }
    if (gate != nullptr) {  // device-side "run only if": see nbx_gemm_small_gated
        const int g = gate[batch];
        if (g != gate_a && g != gate_b) return;
    }
    A += (int64_t)batch * stride_a;
    B += (int64_t)batch * stride_b;
    C += (int64_t)batch * stride_c;
    const int lane = threadIdx.x;
    const int fr = lane & 15, fk = lane >> 4;
    const int row_a = blockIdx.y * 16 + fr;  // A fragment: row fr, k-slot fk
    const int col_b = blockIdx.x * 16 + fr;  // B fragment: k-slot fk, column fr
    const bool a_ok = row_a < M, b_ok = col_b < N;
    // element (x, k) of op(.) sits at base[x * sx + k * sk]
    const int64_t a_sx = A_KC ? lda : 1, a_sk = A_KC ? 1 : lda;
    const int64_t b_sx = B_KC ? ldb : 1, b_sk = B_KC ? 1 : ldb;
    const double* ap = A + (int64_t)(a_ok ? row_a : 0) * a_sx;
    const double* bp = B + (int64_t)(b_ok ? col_b : 0) * b_sx;

    v4f64 acc = (v4f64){0.0, 0.0, 0.0, 0.0};
    const int kfull = K & ~15;
    int k0 = 0;
    // 64 k per trip: 32 independent loads are issued before the first MFMA needs one
    for (; k0 + 64 <= kfull; k0 += 64) {
        double a[16], b[16];
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            a[j] = ap[(int64_t)(k0 + 4 * j + fk) * a_sk];
            b[j] = bp[(int64_t)(k0 + 4 * j + fk) * b_sk];
        }
#pragma unroll
        for (int j = 0; j < 16; ++j)
            acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a_ok ? a[j] : 0.0, b_ok ? b[j] : 0.0, acc, 0, 0, 0);
    }
    for (; k0 < kfull; k0 += 16) {
        double a[4], b[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            a[j] = ap[(int64_t)(k0 + 4 * j + fk) * a_sk];
            b[j] = bp[(int64_t)(k0 + 4 * j + fk) * b_sk];
        }
#pragma unroll
        for (int j = 0; j < 4; ++j)
            acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a_ok ? a[j] : 0.0, b_ok ? b[j] : 0.0, acc, 0, 0, 0);
    }
    if (kfull < K) {
        double a[4], b[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int k = kfull + 4 * j + fk;
            const bool in = k < K;
            a[j] = (in && a_ok) ? ap[(int64_t)k * a_sk] : 0.0;
            b[j] = (in && b_ok) ? bp[(int64_t)k * b_sk] : 0.0;
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a[j], b[j], acc, 0, 0, 0);
    }
    double q0 = 0.0, q1 = 0.0;
    if (b_ok) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int row = blockIdx.y * 16 + fk + 4 * r;
            if (row < M) {
                double* c = C + (int64_t)row * ldc + col_b;
                const double v = alpha * acc[r];
                *c = (beta == 0.0) ? v : fma(beta, *c, v);
                if (second) {  // ||I - C2||_F^2
                    const double d = (row == col_b) ? 1.0 - v : v;
                    q0 = fma(d, d, q0);
                } else {  // ||C - diag||_F^2 and ||C||_F^2
                    if (row != col_b) q0 = fma(v, v, q0);
                    q1 = fma(v, v, q1);
                }
            }
        }
    }
    // norm_part[((batch * 2 + second) * tiles + tile) * 2 + {0, 1}]: this tile's share of the three
    // squared norms the refinement's E kernel needs of a pair (C = S~, C2 = G) -- see nbx_geig_refine
    if (norm_part != nullptr) {
        q0 = nbx_wave_sum(q0);
        q1 = nbx_wave_sum(q1);
        if (lane == 0) {
            const int64_t tiles = (int64_t)gridDim.x * gridDim.y, tile = (int64_t)blockIdx.y * gridDim.x + blockIdx.x;
            double* np = norm_part + (((int64_t)batch * 2 + (second ? 1 : 0)) * tiles + tile) * 2;
            np[0] = q0;
            np[1] = q1;
        }
    }
}

inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

template <int BM, int BN, int WR, int WC>
void launch(nbx_ctx* ctx, bool a_kc, bool b_kc, int M, int N, int K, double alpha, const double* A, int64_t lda,
            int64_t sa, const double* B, int64_t ldb, int64_t sb, double beta, double* C, int64_t ldc, int64_t sc,
            int batch, int vec_a, int vec_b, const int* gate, int gate_a, int gate_b, GemmGen gen = GemmGen{}) {
    dim3 grid((unsigned)nbx_cdiv(N, BN), (unsigned)nbx_cdiv(M, BM), (unsigned)batch);
    dim3 block(64 * WR * WC);
#define NBX_GEMM_GO(AK, BKC)                                                                                       \
    hipLaunchKernelGGL((gemm_f64_kernel<BM, BN, WR, WC, AK, BKC>), grid, block, 0, ctx->stream, M, N, K, alpha, A, \
                       lda, sa, B, ldb, sb, beta, C, ldc, sc, vec_a, vec_b, gen, gate, gate_a, gate_b)
    if (a_kc) {
        if (b_kc) NBX_GEMM_GO(true, true);
        else NBX_GEMM_GO(true, false);
    } else {
        if (b_kc) NBX_GEMM_GO(false, true);
        else NBX_GEMM_GO(false, false);
    }
#undef NBX_GEMM_GO
}

// gemm_m4_tn_kernel applies: 'T','N' operands with even extents, K a multiple of 4, 16-byte aligned rows
// (NBX_GEMM_DMA=0: the register-staged 16 x 16 x 4 kernel instead, for comparison)
inline bool tn_dma_ok(int64_t m, int64_t n, int64_t k, int vec_a, int vec_b) {
    static const bool on = [] {
        const char* e = getenv("NBX_GEMM_DMA");
        return e == nullptr || e[0] != '0';
    }();
    return on && vec_a && vec_b && m > 64 && n > 64 && k >= 4 && k % 4 == 0 && m % 2 == 0 && n % 2 == 0;
}

void launch_tn_dma(nbx_ctx* ctx, int M, int N, int K, double alpha, const double* A, int64_t lda, int64_t sa, const double* B,
                   int64_t ldb, int64_t sb, double beta, double* C, int64_t ldc, int64_t sc, int batch, int pair_n,
                   const int* gate, int gate_a, int gate_b) {
    dim3 grid((unsigned)nbx_cdiv(N, 128), (unsigned)nbx_cdiv(M, 128), (unsigned)batch);
    hipLaunchKernelGGL(gemm_m4_tn_kernel<false>, grid, dim3(256), 0, ctx->stream, M, N, K, alpha, A, lda, sa, B, ldb, sb, beta,
                       C, ldc, sc, pair_n, gate, gate_a, gate_b, GemmGen{});
}

}  // namespace

// Y[z] (m x n) = op(A) (m x k) . G_z (k x n),  G_z[p][q] = val(canon(p, q, r, s0 + z)) * scale,
// z < batch: quarter-1 of the streamed transform with the integrals born in the B-operand
// registers of the GEMM (they never exist in memory).  A is (k x m) row-major ('T').
int nbx_gemm_q1_synth(nbx_ctx* ctx, int64_t m, int64_t n, int64_t k, const double* d_a, int64_t lda, uint64_t seed,
                      double scale, int64_t r, int64_t s0, double* d_y, int64_t ldy, int64_t stride_y,
                      int64_t batch) {
    NBX_CHECK_ARG(ctx && d_a && d_y && m > 0 && n > 0 && k > 0 && batch > 0 && batch <= 65535 && lda >= m &&
                  ldy >= n);
    NBX_CHECK_ARG(m < (1ll << 31) && n < 92681 && k < 92681 && r + 1 < 92681 && s0 + batch < 92681);
    GemmGen gen{seed, scale, (int)r, (int)s0};
    const int vec_a = (aligned16(d_a) && lda % 2 == 0) ? 1 : 0;
    if (tn_dma_ok(m, n, k, vec_a, 1)) {  // A by LDS-DMA, the integrals made straight into the LDS ring
        dim3 grid((unsigned)nbx_cdiv(n, 128), (unsigned)nbx_cdiv(m, 128), (unsigned)batch);
        hipLaunchKernelGGL(gemm_m4_tn_kernel<true>, grid, dim3(256), 0, ctx->stream, (int)m, (int)n, (int)k, 1.0, d_a, lda,
                           (int64_t)0, nullptr, (int64_t)0, (int64_t)0, 0.0, d_y, ldy, stride_y, 0, nullptr, 0, 0, gen);
    } else if (m > 64 && n > 64) {
        dim3 grid((unsigned)nbx_cdiv(n, 128), (unsigned)nbx_cdiv(m, 128), (unsigned)batch);
        hipLaunchKernelGGL((gemm_f64_kernel<128, 128, 2, 4, false, false, true>), grid, dim3(512), 0, ctx->stream, (int)m,
                           (int)n, (int)k, 1.0, d_a, lda, (int64_t)0, nullptr, (int64_t)0, (int64_t)0, 0.0, d_y, ldy,
                           stride_y, vec_a, 0, gen);
    } else {
        dim3 grid((unsigned)nbx_cdiv(n, 64), (unsigned)nbx_cdiv(m, 64), (unsigned)batch);
        hipLaunchKernelGGL((gemm_f64_kernel<64, 64, 2, 2, false, false, true>), grid, dim3(256), 0, ctx->stream, (int)m,
                           (int)n, (int)k, 1.0, d_a, lda, (int64_t)0, nullptr, (int64_t)0, (int64_t)0, 0.0, d_y, ldy,
                           stride_y, vec_a, 0, gen);
    }
    NBX_LAUNCH_CHECK();
    return NBX_OK;
}

bool nbx_gemm_small_supported(int64_t m, int64_t n, int64_t k, int64_t batch) {
    return batch <= 65535 && nbx_cdiv(m, 16) * nbx_cdiv(n, 16) * batch <= 512 && k <= 4096;
}

// Small-product kernel whose workgroups of batch entry b return at once unless gate[b] is gate_a
// or gate_b when the kernel starts: lets a sequence of launches be queued whose tail depends on
// a result computed on the device (eigh_refine.hip), with no host round trip.
int nbx_gemm_small_gated(nbx_ctx* ctx, char trans_a, char trans_b, int64_t m, int64_t n, int64_t k, double alpha,
                         const double* d_a, int64_t lda, int64_t stride_a, const double* d_b, int64_t ldb,
                         int64_t stride_b, double beta, double* d_c, int64_t ldc, int64_t stride_c, int64_t batch,
                         const int* d_gate, int gate_a, int gate_b, const double* d_b2, double* d_c2,
                         double* d_norm_part) {
    NBX_CHECK_ARG(ctx && d_a && d_b && d_c && m > 0 && n > 0 && k > 0 && batch > 0);
    NBX_CHECK_ARG(d_norm_part == nullptr || (d_b2 != nullptr && beta == 0.0));
    NBX_CHECK_ARG(nbx_gemm_small_supported(m, n, k, batch));
    NBX_CHECK_ARG((d_b2 == nullptr) == (d_c2 == nullptr));
    const bool a_kc = !(trans_a == 'T' || trans_a == 't');
    const bool b_kc = (trans_b == 'T' || trans_b == 't');
    const bool pair = d_b2 != nullptr;
    dim3 grid((unsigned)nbx_cdiv(n, 16), (unsigned)nbx_cdiv(m, 16), (unsigned)(pair ? 2 * batch : batch));
    const int split = pair ? (int)batch : (1 << 30);
#define NBX_GEMM_SMALL(AK, BKC)                                                                               \
    hipLaunchKernelGGL((gemm_small_kernel<AK, BKC>), grid, dim3(64), 0, ctx->stream, (int)m, (int)n, (int)k, \
                       alpha, d_a, lda, stride_a, d_b, ldb, stride_b, beta, d_c, ldc, stride_c, d_gate, gate_a, gate_b, \
                       d_b2, d_c2, split, d_norm_part)
    if (a_kc) {
        if (b_kc) NBX_GEMM_SMALL(true, true);
        else NBX_GEMM_SMALL(true, false);
    } else {
        if (b_kc) NBX_GEMM_SMALL(false, true);
        else NBX_GEMM_SMALL(false, false);
    }
#undef NBX_GEMM_SMALL
    NBX_LAUNCH_CHECK();
    return NBX_OK;
}

// C_b = op(A_b) op(B_b) for a "triangular batch": entry b (global index tri_m0 + b) has
// (tri_m0 + b + 1) * unit rows; A_b = A + b * stride_a (lda), the C blocks are stored one after the
// other (row length ldc).  Operands 'N','N' row-major; 128 x 128 tiles.
int nbx_gemm_tri(nbx_ctx* ctx, int64_t unit, int64_t tri_m0, int64_t nbatch, int64_t n, int64_t k, const double* d_a,
                 int64_t lda, int64_t stride_a, const double* d_b, int64_t ldb, double* d_c, int64_t ldc,
                 int64_t a_symn) {
    NBX_CHECK_ARG(ctx && d_a && d_b && d_c && unit > 0 && tri_m0 >= 0 && nbatch > 0 && n > 0 && k > 0);
    const int64_t m_max = (tri_m0 + nbatch) * unit;
    NBX_CHECK_ARG(m_max < (1ll << 31) && nbatch <= 65535);
    GemmGen gen{};
    gen.tri_unit = (int)unit;
    gen.tri_m0 = (int)tri_m0;
    gen.a_symn = (int)a_symn;  // > 0: the rows of A are rows of packed symmetric matrices of this order
    const int vec_a = (aligned16(d_a) && lda % 2 == 0 && stride_a % 2 == 0) ? 1 : 0;
    const int vec_b = (aligned16(d_b) && ldb % 2 == 0) ? 1 : 0;
    launch<128, 128, 2, 4>(ctx, true, false, (int)m_max, (int)n, (int)k, 1.0, d_a, lda, stride_a, d_b, ldb, 0, 0.0, d_c,
                           ldc, 0, (int)nbatch, vec_a, vec_b, nullptr, 0, 0, gen);
    NBX_LAUNCH_CHECK();
    return NBX_OK;
}

// C[(i,j)] = C[(j,i)] = A^T B_b for the pairs b = i(i+1)/2 + j, j <= i < pair_n: A is (k x m) shared,
// B_b = B + b * stride_b (k x n, row length ldb), the (m x n) results are blocks of an
// (pair_n, pair_n, m, n) tensor.
int nbx_gemm_pair_scatter(nbx_ctx* ctx, int64_t pair_n, int64_t m, int64_t n, int64_t k, const double* d_a, int64_t lda,
                          const double* d_b, int64_t ldb, int64_t stride_b, double* d_c) {
    NBX_CHECK_ARG(ctx && d_a && d_b && d_c && pair_n > 0 && m > 0 && n > 0 && k > 0);
    const int64_t npairs = pair_n * (pair_n + 1) / 2;
    NBX_CHECK_ARG(npairs <= 65535 && m < (1ll << 31) && n < (1ll << 31));
    GemmGen gen{};
    gen.pair_n = (int)pair_n;
    const int vec_a = (aligned16(d_a) && lda % 2 == 0) ? 1 : 0;
    const int vec_b = (aligned16(d_b) && ldb % 2 == 0 && stride_b % 2 == 0) ? 1 : 0;
    if (tn_dma_ok(m, n, k, vec_a, vec_b))
        launch_tn_dma(ctx, (int)m, (int)n, (int)k, 1.0, d_a, lda, 0, d_b, ldb, stride_b, 0.0, d_c, n, m * n, (int)npairs,
                      (int)pair_n, nullptr, 0, 0);
    else if (m > 64 && n > 64)
        launch<128, 128, 2, 4>(ctx, false, false, (int)m, (int)n, (int)k, 1.0, d_a, lda, 0, d_b, ldb, stride_b, 0.0, d_c, n,
                               m * n, (int)npairs, vec_a, vec_b, nullptr, 0, 0, gen);
    else
        launch<64, 64, 2, 2>(ctx, false, false, (int)m, (int)n, (int)k, 1.0, d_a, lda, 0, d_b, ldb, stride_b, 0.0, d_c, n,
                             m * n, (int)npairs, vec_a, vec_b, nullptr, 0, 0, gen);
    NBX_LAUNCH_CHECK();
    return NBX_OK;
}

extern "C" int nbx_gemm(nbx_ctx* ctx, char trans_a, char trans_b, int64_t m, int64_t n, int64_t k,
                        double alpha, const double* d_a, int64_t lda, int64_t stride_a, const double* d_b,
                        int64_t ldb, int64_t stride_b, double beta, double* d_c, int64_t ldc,
                        int64_t stride_c, int64_t batch) {
    return nbx_gemm_gated(ctx, trans_a, trans_b, m, n, k, alpha, d_a, lda, stride_a, d_b, ldb, stride_b, beta, d_c, ldc,
                          stride_c, batch, nullptr, 0, 0);
}

// nbx_gemm whose workgroups of batch entry b return at once unless d_gate[b] is gate_a or gate_b
// when the kernel starts (d_gate == nullptr: plain nbx_gemm).  Any shape, any kernel.
int nbx_gemm_gated(nbx_ctx* ctx, char trans_a, char trans_b, int64_t m, int64_t n, int64_t k, double alpha,
                   const double* d_a, int64_t lda, int64_t stride_a, const double* d_b, int64_t ldb, int64_t stride_b,
                   double beta, double* d_c, int64_t ldc, int64_t stride_c, int64_t batch, const int* d_gate,
                   int gate_a, int gate_b) {
    NBX_CHECK_ARG(ctx != nullptr);
    NBX_CHECK_ARG(d_gate == nullptr || batch <= 65535);
    NBX_CHECK_ARG(trans_a == 'N' || trans_a == 'T' || trans_a == 'n' || trans_a == 't');
    NBX_CHECK_ARG(trans_b == 'N' || trans_b == 'T' || trans_b == 'n' || trans_b == 't');
    NBX_CHECK_ARG(m >= 0 && n >= 0 && k >= 0 && batch >= 0);
    if (m == 0 || n == 0 || batch == 0) return NBX_OK;
    NBX_CHECK_ARG(d_c != nullptr && ldc >= n);
    NBX_CHECK_ARG(k == 0 || (d_a != nullptr && d_b != nullptr));
    const bool ta = (trans_a == 'T' || trans_a == 't');
    const bool tb = (trans_b == 'T' || trans_b == 't');
    NBX_CHECK_ARG(k == 0 || lda >= (ta ? m : k));
    NBX_CHECK_ARG(k == 0 || ldb >= (tb ? k : n));
    NBX_CHECK_ARG(m < (1ll << 31) && n < (1ll << 31) && k < (1ll << 31));
    // op(A)[m][k]: 'N' -> contiguous along k; 'T' -> contiguous along m.
    const bool a_kc = !ta;
    // op(B)[k][n]: 'N' -> contiguous along n; 'T' -> contiguous along k.
    const bool b_kc = tb;
    const int vec_a = (k > 0 && aligned16(d_a) && lda % 2 == 0 && stride_a % 2 == 0) ? 1 : 0;
    const int vec_b = (k > 0 && aligned16(d_b) && ldb % 2 == 0 && stride_b % 2 == 0) ? 1 : 0;

    // z-dimension of a grid is limited to 65535: loop over batch chunks
    const int64_t zmax = 65535;
    for (int64_t b0 = 0; b0 < batch; b0 += zmax) {
        const int nb = (int)((batch - b0) < zmax ? (batch - b0) : zmax);
        const double* A = d_a ? d_a + b0 * stride_a : nullptr;
        const double* B = d_b ? d_b + b0 * stride_b : nullptr;
        double* C = d_c + b0 * stride_c;
        const int64_t tiles128 = nbx_cdiv(m, 128) * nbx_cdiv(n, 128) * nb;
        const int64_t tiles16 = nbx_cdiv(m, 16) * nbx_cdiv(n, 16) * nb;
        if (tiles16 <= 512 && k <= 4096) {
            dim3 grid((unsigned)nbx_cdiv(n, 16), (unsigned)nbx_cdiv(m, 16), (unsigned)nb);
#define NBX_GEMM_SMALL(AK, BKC)                                                                               \
    hipLaunchKernelGGL((gemm_small_kernel<AK, BKC>), grid, dim3(64), 0, ctx->stream, (int)m, (int)n, (int)k, \
                       alpha, A, lda, stride_a, B, ldb, stride_b, beta, C, ldc, stride_c, d_gate, gate_a, gate_b, nullptr, \
                       nullptr, 1 << 30, nullptr)
            if (a_kc) {
                if (b_kc) NBX_GEMM_SMALL(true, true);
                else NBX_GEMM_SMALL(true, false);
            } else {
                if (b_kc) NBX_GEMM_SMALL(false, true);
                else NBX_GEMM_SMALL(false, false);
            }
#undef NBX_GEMM_SMALL
        } else if ((tiles128 >= 512 || (tiles128 >= 128 && k >= 2048)) && !a_kc && !b_kc && tn_dma_ok(m, n, k, vec_a, vec_b)) {
            launch_tn_dma(ctx, (int)m, (int)n, (int)k, alpha, A, lda, stride_a, B, ldb, stride_b, beta, C, ldc, stride_c, nb,
                          0, d_gate, gate_a, gate_b);
        } else if (tiles128 >= 512 && m > 64 && n > 64) {
            launch<128, 128, 2, 4>(ctx, a_kc, b_kc, (int)m, (int)n, (int)k, alpha, A, lda, stride_a, B, ldb, stride_b,
                             beta, C, ldc, stride_c, nb, vec_a, vec_b, d_gate, gate_a, gate_b);
        } else if (m <= 32 || n <= 32 || nbx_cdiv(m, 64) * nbx_cdiv(n, 64) * nb < 128) {
            // (also where 64 x 64 tiles would leave half the chip idle -- the 64-row panels of the blocked
            //  back-transformation, csrc/eigh_grid.hip, batched matrices of a few hundred rows: four times the workgroups)
            launch<32, 32, 2, 2>(ctx, a_kc, b_kc, (int)m, (int)n, (int)k, alpha, A, lda, stride_a, B, ldb, stride_b,
                           beta, C, ldc, stride_c, nb, vec_a, vec_b, d_gate, gate_a, gate_b);
        } else {
            launch<64, 64, 2, 2>(ctx, a_kc, b_kc, (int)m, (int)n, (int)k, alpha, A, lda, stride_a, B, ldb, stride_b,
                           beta, C, ldc, stride_c, nb, vec_a, vec_b, d_gate, gate_a, gate_b);
        }
        NBX_LAUNCH_CHECK();
    }
    return NBX_OK;
}
