// The J/K "walk" of one (pq| tile on the matrix cores, IN ISOLATION: the tile -- the packed
// lower triangle L of the symmetric (r,s) matrix M = (pq|rs), N (N + 1) / 2 doubles -- and the four density columns
// X[c][0:4] = (D^a_q, D^a_p, D^b_q, D^b_p)[c] are LDS resident; the kernel computes
//       out[t][0:4] = sum_c M[t][c] X[c][0:4] = (L X + strict(L)^T X)[t]
// `iters` times over, nothing streams.  v_mfma_f64_4x4x4_4b_f64: four independent 4 x 4 x 4 products per
// instruction, D_b[i][j] += sum_k A_b[i][k] B_b[k][j], A_b[i][k] at lane 16 k + 4 b + i, B_b[k][j] at 16 k + 4 b + j,
// D_b[i][j] at 16 i + 4 b + j (profiles/r02/mfma_f64_4x4x4_probe.txt): with lane = 16 a + 4 b + c,
//   row part, item (G, C):  A = L[16 G + 4 b + c][4 C + a]   B = X[4 C + a][c]   D -> acc[G] (rows 16 G + 4 b + a)
//   col part, item (T, H):  A = L[4 T + a][16 H + 4 b + c]   B = X[4 T + a][c]   D -> acc[H] (rows 16 H + 4 b + a)
// Both operands carry the contraction index in lane >> 4, so ONE register holding a block of L can only ever be
// contracted over one of its two indices: the column part needs the block in the transposed lane map, i.e. a second
// ds_read of the same data (the triangle is read twice, not 5.6 times as jk_s8.hip's 16-row strips did).
//
// Measured here: us per tile walk per CU with 1 or 2 four-wave workgroups per CU, against the 0.87 us / chunk
// (3.5 us / tile) period at which the production kernel streams tiles through a CU.
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/walk_probe profiles/r03/jk_mfma_walk_probe.hip && /tmp/walk_probe
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <vector>

constexpr int N = 148, NB = N / 4, NG = (NB + 3) / 4;  // 37 block rows of 4, 10 groups of 4 block rows
constexpr int TRI = N * (N + 1) / 2;
constexpr int NBLK = NB * (NB + 1) / 2, BLK = 16 * NBLK;  // block-major tile: 4 x 4 blocks (T, C <= T), 16 doubles each

__device__ __forceinline__ int tri(int r) { return (r * (r + 1)) >> 1; }

// mode 0: the MFMA walk; 1: operands read but MFMAs replaced by a cheap add (LDS side alone); 2: MFMAs on constant
// operands (matrix pipe alone)
template <int MODE, int NW>
__global__ __launch_bounds__(64 * NW) void walk_kernel(const double* __restrict__ tiles, const double* __restrict__ xtab,
                                                   double* __restrict__ out, int iters) {
    extern __shared__ __attribute__((aligned(16))) double smem[];
    double* ls = smem;            // [NBLK][16]: block (T, C) at T (T + 1) / 2 + C, element (row i, column k) at 4 k + i;
                                  // the upper part of the diagonal blocks is stored as zeros
    double* xs = smem + BLK;      // [N][4]
    double* red = xs + 4 * N;     // [NW waves][NG][64] partial accumulators
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);  // (a scalar: the loops below are wave uniform)
    const int a = lane >> 4, b = (lane >> 2) & 3, c = lane & 3;
    for (int e = threadIdx.x; e < BLK; e += 64 * NW) {
        const int blk = e >> 4, k = (e >> 2) & 3, i = e & 3;
        int T = (int)((sqrtf(8.0f * blk + 1.0f) - 1.0f) * 0.5f);
        while (tri(T + 1) <= blk) ++T;
        while (tri(T) > blk) --T;
        const int C = blk - tri(T), row = 4 * T + i, col = 4 * C + k;
        ls[e] = col <= row ? tiles[(size_t)blockIdx.x * TRI + tri(row) + col] : 0.0;
    }
    for (int i = threadIdx.x; i < 4 * N; i += 64 * NW) xs[i] = xtab[(size_t)blockIdx.x * 4 * N + i];
    __syncthreads();
    double acc[NG];
#pragma unroll
    for (int g = 0; g < NG; ++g) acc[g] = 0.0;
    for (int it = 0; it < iters; ++it) {
        // Work split: groups of four waves; within a group wave w4 = wave & 3 takes block columns (row part) / block
        // rows (column part) 4 j + w4, so that every bound below is STATIC (C <= 4 G + 3 <=> G >= j; H <= T / 4 <=> H <= j):
        // straight-line code, the compiler batches the LDS reads ahead of the MFMAs.  NW = 8: waves 0-3 do the row
        // part, waves 4-7 the column part; NW = 4: every wave does both.
        constexpr int NJ = (NB + 3) / 4;
        const int w4 = wave & 3;
        const bool do_row = NW == 4 || wave < 4, do_col = NW == 4 || wave >= 4;
        // ---- row part: item (G, C), C <= 4 G + 3.  Lane (a, b, c) reads block (T = 4 G + b, C), element (row c,
        // column a): 4 a + c
        const int lo_row = 4 * a + c;
        if (do_row) {
#pragma unroll
            for (int j = 0; j < NJ; ++j) {
                const int C = 4 * j + w4;
                if (C >= NB) continue;  // scalar (last j only)
                const double bx = xs[4 * (4 * C + a) + c];
#pragma unroll
                for (int G = j; G < NG; ++G) {
                    const int T = 4 * G + b;
                    double av = 0.0;
                    if (MODE != 2 && (4 * G + 3 < NB || T < NB) && C <= T) av = ls[16 * (tri(T) + C) + lo_row];
                    if (MODE == 0) acc[G] = __builtin_amdgcn_mfma_f64_4x4x4f64(av, bx, acc[G], 0, 0, 0);
                    else if (MODE == 1) acc[G] += av * bx;
                    else acc[G] = __builtin_amdgcn_mfma_f64_4x4x4f64(bx, bx, acc[G], 0, 0, 0);
                }
            }
        }
        // ---- column part: item (T, H), H <= T / 4, strictly lower.  Lane (a, b, c) reads block (T, 4 H + b), element
        // (row a, column c): 4 c + a -- four consecutive blocks, 512 contiguous bytes
        const int lo_col = 16 * b + 4 * c + a;
        if (do_col) {
#pragma unroll
            for (int j = 0; j < NJ; ++j) {
                const int T = 4 * j + w4;
                if (T >= NB) continue;  // scalar (last j only)
                const double bt = xs[4 * (4 * T + a) + c];
                const double* lt = ls + 16 * tri(T) + lo_col;
#pragma unroll
                for (int H = 0; H <= j; ++H) {
                    double av = 0.0;
                    // blocks right of the diagonal one do not exist; the diagonal block contributes its strict lower part
                    if (MODE != 2 && (H < j || 4 * H + b < T || (4 * H + b == T && c < a))) av = lt[64 * H];
                    if (MODE == 0) acc[H] = __builtin_amdgcn_mfma_f64_4x4x4f64(av, bt, acc[H], 0, 0, 0);
                    else if (MODE == 1) acc[H] += av * bt;
                    else acc[H] = __builtin_amdgcn_mfma_f64_4x4x4f64(bt, bt, acc[H], 0, 0, 0);
                }
            }
        }
    }
    // ---- the four waves' partial sums: D_b[i][j] at lane 16 i + 4 b + j -> out row 16 G + 4 b + i, column j
#pragma unroll
    for (int g = 0; g < NG; ++g) red[(wave * NG + g) * 64 + lane] = acc[g];
    __syncthreads();
    for (int e = threadIdx.x; e < NG * 64; e += 64 * NW) {
        const int g = e >> 6, l = e & 63;
        const int row = 16 * g + 4 * ((l >> 2) & 3) + (l >> 4), colx = l & 3;
        double s = 0.0;
        for (int w = 0; w < NW; ++w) s += red[(w * NG + g) * 64 + l];
        if (row < N) out[((size_t)blockIdx.x * N + row) * 4 + colx] = s;
    }
}

// The same walk with RUN-TIME loop bounds (one kernel for every N): wave w owns the output groups G = w, w + 4, ...
// for the row part and H = 3 - w, 7 - w, ... for the column part (balances the triangular item counts), each with two
// accumulators (even / odd steps: consecutive MFMAs are independent); no cross-wave reduction is needed.
template <int NW>
__global__ __launch_bounds__(64 * NW) void walk_rt_kernel(const double* __restrict__ tiles, const double* __restrict__ xtab,
                                                          double* __restrict__ out, int iters, int nb) {
    extern __shared__ __attribute__((aligned(16))) double smem[];
    double* ls = smem;
    double* xs = smem + BLK;
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int a = lane >> 4, b = (lane >> 2) & 3, c = lane & 3;
    for (int e = threadIdx.x; e < BLK; e += 64 * NW) {
        const int blk = e >> 4, k = (e >> 2) & 3, i = e & 3;
        int T = (int)((sqrtf(8.0f * blk + 1.0f) - 1.0f) * 0.5f);
        while (tri(T + 1) <= blk) ++T;
        while (tri(T) > blk) --T;
        const int C = blk - tri(T), row = 4 * T + i, col = 4 * C + k;
        ls[e] = col <= row ? tiles[(size_t)blockIdx.x * TRI + tri(row) + col] : 0.0;
    }
    for (int i = threadIdx.x; i < 4 * N; i += 64 * NW) xs[i] = xtab[(size_t)blockIdx.x * 4 * N + i];
    __syncthreads();
    const int ng = (nb + 3) >> 2;
    constexpr int MAXG = 4;  // groups per wave: N <= 256
    double accr[MAXG], accc[MAXG];
#pragma unroll
    for (int g = 0; g < MAXG; ++g) accr[g] = accc[g] = 0.0;
    const int lo_row = 4 * a + c, lo_col = 16 * b + 4 * c + a;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int gi = 0; gi < MAXG; ++gi) {
            // ---- row part of group G: out rows 16 G + 4 b + a, block columns C = 0 .. min(4 G + 3, nb - 1)
            const int G = wave + NW * gi;
            if (G < ng) {
                const int T = 4 * G + b;
                const bool t_ok = T < nb;
                const double* lp = ls + 16 * tri(t_ok ? T : 0) + lo_row;
                const double* xp = xs + 4 * a + c;
                const int cmax = min(4 * G + 3, nb - 1);
                double d0 = accr[gi], d1 = 0.0;
                int C = 0;
                for (; C + 1 <= cmax; C += 2) {
                    const double a0 = (t_ok && C <= T) ? lp[16 * C] : 0.0;
                    const double a1 = (t_ok && C + 1 <= T) ? lp[16 * C + 16] : 0.0;
                    const double b0 = xp[16 * C], b1 = xp[16 * C + 16];
                    d0 = __builtin_amdgcn_mfma_f64_4x4x4f64(a0, b0, d0, 0, 0, 0);
                    d1 = __builtin_amdgcn_mfma_f64_4x4x4f64(a1, b1, d1, 0, 0, 0);
                }
                if (C <= cmax) {
                    const double a0 = (t_ok && C <= T) ? lp[16 * C] : 0.0;
                    d0 = __builtin_amdgcn_mfma_f64_4x4x4f64(a0, xp[16 * C], d0, 0, 0, 0);
                }
                accr[gi] = d0 + d1;
            }
            // ---- column part of group H: out rows 16 H + 4 b + a, block rows T = 4 H .. nb - 1, strictly lower
            const int H = (NW - 1 - wave) + NW * gi;
            if (H < ng) {
                const int Cb = 4 * H + b;  // this lane's block column
                double d0 = accc[gi], d1 = 0.0;
                const double* xp = xs + 4 * a + c;
                int T = 4 * H;
                for (; T + 1 < nb; T += 2) {
                    const double* l0 = ls + 16 * (tri(T) + 4 * H) + lo_col;
                    const double* l1 = ls + 16 * (tri(T + 1) + 4 * H) + lo_col;
                    const double a0 = (Cb < T || (Cb == T && c < a)) ? l0[0] : 0.0;
                    const double a1 = (Cb < T + 1 || (Cb == T + 1 && c < a)) ? l1[0] : 0.0;
                    d0 = __builtin_amdgcn_mfma_f64_4x4x4f64(a0, xp[16 * T], d0, 0, 0, 0);
                    d1 = __builtin_amdgcn_mfma_f64_4x4x4f64(a1, xp[16 * T + 16], d1, 0, 0, 0);
                }
                if (T < nb) {
                    const double* l0 = ls + 16 * (tri(T) + 4 * H) + lo_col;
                    const double a0 = (Cb < T || (Cb == T && c < a)) ? l0[0] : 0.0;
                    d0 = __builtin_amdgcn_mfma_f64_4x4x4f64(a0, xp[16 * T], d0, 0, 0, 0);
                }
                accc[gi] = d0 + d1;
            }
        }
    }
    // every output row has exactly one row-part owner and one column-part owner: two plain stores through LDS
    __syncthreads();
    double* red = ls;  // (the tile is no longer needed)
#pragma unroll
    for (int gi = 0; gi < MAXG; ++gi) {
        const int G = wave + NW * gi, H = (NW - 1 - wave) + NW * gi;
        if (G < ng) red[G * 64 + lane] = accr[gi];
        if (H < ng) red[(ng + H) * 64 + lane] = accc[gi];
    }
    __syncthreads();
    for (int e = threadIdx.x; e < ng * 64; e += 64 * NW) {
        const int g = e >> 6, l = e & 63;
        const int row = 16 * g + 4 * ((l >> 2) & 3) + (l >> 4), colx = l & 3;
        if (row < N) out[((size_t)blockIdx.x * N + row) * 4 + colx] = red[e] + red[ng * 64 + e];
    }
}

int main() {
    const int wgs_max = 512;
    std::vector<double> tiles((size_t)wgs_max * TRI), xt((size_t)wgs_max * 4 * N);
    unsigned long long s = 88172645463325252ull;
    auto rnd = [&]() {
        s ^= s << 13;
        s ^= s >> 7;
        s ^= s << 17;
        return (double)(s >> 11) / 9007199254740992.0 - 0.5;
    };
    for (auto& v : tiles) v = rnd();
    for (auto& v : xt) v = rnd();
    double *d_t, *d_x, *d_o;
    hipMalloc(&d_t, tiles.size() * 8);
    hipMalloc(&d_x, xt.size() * 8);
    hipMalloc(&d_o, (size_t)wgs_max * N * 4 * 8);
    hipMemcpy(d_t, tiles.data(), tiles.size() * 8, hipMemcpyHostToDevice);
    hipMemcpy(d_x, xt.data(), xt.size() * 8, hipMemcpyHostToDevice);
    const size_t lds = (size_t)(BLK + 4 * N + 8 * NG * 64) * 8;
    printf("N = %d, tile %d doubles (%.1f KB), LDS per workgroup %.1f KB\n", N, TRI, TRI * 8 / 1024.0, lds / 1024.0);
#define ATTR(M, W) hipFuncSetAttribute((const void*)walk_kernel<M, W>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024)
    ATTR(0, 4); ATTR(1, 4); ATTR(2, 4); ATTR(0, 8); ATTR(1, 8); ATTR(2, 8);
    // correctness: one walk of workgroup 0 against the host
    walk_kernel<0, 8><<<1, 512, lds>>>(d_t, d_x, d_o, 1);
    std::vector<double> got((size_t)N * 4);
    hipMemcpy(got.data(), d_o, got.size() * 8, hipMemcpyDeviceToHost);
    double err = 0.0, big = 0.0;
    for (int t = 0; t < N; ++t)
        for (int j = 0; j < 4; ++j) {
            double ref = 0.0;
            for (int cc = 0; cc < N; ++cc) {
                const int hi = t > cc ? t : cc, lo = t > cc ? cc : t;
                ref += tiles[(size_t)hi * (hi + 1) / 2 + lo] * xt[4 * cc + j];
            }
            err = fmax(err, fabs(ref - got[4 * t + j]));
            big = fmax(big, fabs(ref));
        }
    printf("check: max |out - ref| = %.3e (max |ref| %.3f)\n", err, big);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    const int iters = 400;
    for (int mode = 0; mode < 3; ++mode)
        for (int nw : {4, 8}) {
            const int wgs = 256;
            float ms = 0.f;
            for (int rep = 0; rep < 3; ++rep) {
                hipEventRecord(e0);
#define RUN(M, W) walk_kernel<M, W><<<wgs, 64 * W, lds>>>(d_t, d_x, d_o, iters)
                if (mode == 0 && nw == 4) RUN(0, 4);
                else if (mode == 0) RUN(0, 8);
                else if (mode == 1 && nw == 4) RUN(1, 4);
                else if (mode == 1) RUN(1, 8);
                else if (nw == 4) RUN(2, 4);
                else RUN(2, 8);
                hipEventRecord(e1);
                hipEventSynchronize(e1);
                hipEventElapsedTime(&ms, e0, e1);
            }
            const double us_tile = ms * 1e3 / iters;  // one workgroup (one tile) per CU
            printf("%s, %d waves per CU on one tile: %.3f ms for %d walks -> %.3f us per tile walk per CU "
                   "(stream period 3.5 us / tile; 43 tiles per CU per build -> %.1f us per J/K build)\n",
                   mode == 0 ? "MFMA walk          " : mode == 1 ? "LDS reads, VALU fma" : "MFMA only (no LDS) ", nw, ms, iters,
                   us_tile, us_tile * 43.1);
        }
    hipFuncSetAttribute((const void*)walk_rt_kernel<4>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    walk_rt_kernel<4><<<1, 256, lds>>>(d_t, d_x, d_o, 1, NB);
    hipMemcpy(got.data(), d_o, got.size() * 8, hipMemcpyDeviceToHost);
    err = 0.0;
    for (int t = 0; t < N; ++t)
        for (int j = 0; j < 4; ++j) {
            double ref = 0.0;
            for (int cc = 0; cc < N; ++cc) {
                const int hi = t > cc ? t : cc, lo = t > cc ? cc : t;
                ref += tiles[(size_t)hi * (hi + 1) / 2 + lo] * xt[4 * cc + j];
            }
            err = fmax(err, fabs(ref - got[4 * t + j]));
        }
    printf("run-time-bounds walk: check max |out - ref| = %.3e\n", err);
    {
        float ms = 0.f;
        for (int rep = 0; rep < 3; ++rep) {
            hipEventRecord(e0);
            walk_rt_kernel<4><<<256, 256, lds>>>(d_t, d_x, d_o, iters, NB);
            hipEventRecord(e1);
            hipEventSynchronize(e1);
            hipEventElapsedTime(&ms, e0, e1);
        }
        const double us_tile = ms * 1e3 / iters;
        printf("MFMA walk, run-time bounds, wave owns its output groups, 4 waves per CU: %.3f ms -> %.3f us per tile walk per CU "
               "(-> %.1f us per J/K build)\n", ms, us_tile, us_tile * 43.1);
    }
    return 0;
}
