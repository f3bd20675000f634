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
#pragma unroll
        for (int kk = 0; kk < BK / 4; ++kk) {
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
    if (m > 64 && n > 64) {
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
    if (m > 64 && n > 64)
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
        } else if (tiles128 >= 512 && m > 64 && n > 64) {
            launch<128, 128, 2, 4>(ctx, a_kc, b_kc, (int)m, (int)n, (int)k, alpha, A, lda, stride_a, B, ldb, stride_b,
                             beta, C, ldc, stride_c, nb, vec_a, vec_b, d_gate, gate_a, gate_b);
        } else if (m <= 32 || n <= 32) {
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
