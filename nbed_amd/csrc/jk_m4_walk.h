// The walk of one chunk of a 4 x 4-block packed tile on v_mfma_f64_4x4x4_4b_f64: shared by jk_m4.hip (N <= 148, a tile in
// four chunks) and jk_mx.hip (the sizes above, a tile in as many chunks as the LDS ring asks for).  G_ is the geometry
// class (jk_m4_layout.h: M4Geom<NB> / MxGeom<NB>): N, NB, NG and row0(k), the first block row of chunk k.
#pragma once
#include "jk_m4_layout.h"

namespace {

// ---------------------------------------------------------------------------------------------- the walk of one chunk
// buf: the chunk in LDS (block (T, C) at 16 (tri(T) - tri(RA) + C)); xs: X[N][4] of this tile; acc[G]: this wave's partial
// out rows 16 G + 4 b + a (D layout of the 4x4x4 product: lane 16 i + 4 b + j holds D_b[i][j]).
//
// The walking wave is bound by the number of instructions it issues (one per ~9 cycles, alone on its SIMD next to a
// loading wave), so the walk is written to issue few besides its reads and MFMAs:
//   * addresses: lane.rowg[G] = 64 G b + 16 (tri(b) + w4) + 4 (a ^ b ^ w4) + (c ^ a) holds everything of a row-part
//     operand's address that depends on the lane or the wave (ten registers for the whole kernel); the chunk's slot is
//     added once per row group and chunk and the rest of every address is a constant in the instruction's offset field;
//     the X operand likewise (lane.xrow + 64 j);
//   * masks: a block row outside the chunk (the chunk boundaries cut through groups of four rows) or past the matrix is
//     taken care of ONCE per row group and chunk by pointing that lane's base beyond the workgroup's LDS allocation
//     -- reads there return zero -- instead of a select per operand; what is left are the selects of the items that
//     touch the diagonal (C <= T, strict lower part of the diagonal blocks).
template <int NG_>
struct M4Lane {
    int a, b, c;      // lane = 16 a + 4 b + c
    int rowg[NG_];    // row part: see above
    int xrow;         // 4 a + c + 16 w4: the X element of block column C = 4 j + w4 is at xrow + 64 j
    int col0, cbx;    // column part: 16 b + (a ^ c), c ^ b
    int xlane;        // 4 a + c
};
// (doubles) beyond the LDS of a CU, with room for the offsets: out-of-range LDS reads return zero on gfx9 -- a HARDWARE
// contract, which nbx_ctx_create checks the premise of (LDS per CU <= 0x28000 bytes) and the LDS-poisoning test of
// tests/test_gpu_kernels.py the consequence of
constexpr int M4_LDS_OOB = 0x30000 / 8;

template <class G_, int K>
__device__ __forceinline__ void m4_walk_chunk(const double* __restrict__ buf, const double* __restrict__ xs, int w4,
                                              const M4Lane<G_::NG>& ln, double (&acc)[G_::NG],
                                              double (&bxr)[G_::NG]) {
    constexpr int NB = G_::NB;
    constexpr int RA = G_::row0(K), RB = G_::row0(K + 1), NG = G_::NG;
    constexpr int BASE = m4_tri(RA);
    constexpr int NJR = (NB + 3) / 4, NJC = (RB - RA + 3) / 4;
    const int a = ln.a, b = ln.b, c = ln.c;
    // ---- row part: items (G, C = 4 j + w4), C <= T = 4 G + b, T in [RA, RB)
    const double* bg[NG];
#pragma unroll
    for (int G = 0; G < NG; ++G) {
        if (4 * G + 3 < RA || 4 * G >= RB) continue;  // static: the group has no row in the chunk
        const bool whole = 4 * G >= RA && 4 * G + 3 < RB && 4 * G + 3 < NB;  // static
        const int T = 4 * G + b;
        bg[G] = buf + ((whole || (T >= RA && T < RB && T < NB)) ? ln.rowg[G] : M4_LDS_OOB);
    }
    const double* xr = xs + ln.xrow;
    auto row_live = [](int j) constexpr { return j < NJR && 4 * j < RB; };  // (static: a block of this column group lies in the chunk)
    // bxr[j] = X[4 (4 j + w4) + a][c], the B operand of the row part's items of block column 4 j + w4: the same for the
    // four chunks of a tile, read with the first (a quarter of all LDS reads of the walk were these, once per chunk)
    auto load_row = [&](int j, double (&av)[NG], double& bx) {
        // (a block column past the matrix -- 4 j + w4 >= NB, the last j of waves 1..3 -- has no X row: the read would land
        //  behind the X of this tile, in LDS that nothing has written yet when the first tile of a range starts, and a NaN
        //  left there by an earlier kernel times the zero of the masked operand is a NaN.  Zero, not whatever is there.)
        if (K == 0) bx = (4 * j + w4 < NB) ? xr[64 * j] : 0.0;
        if (!row_live(j)) return;
#pragma unroll
        for (int G = 0; G < NG; ++G) {
            if (G < j || 4 * G + 3 < RA || 4 * G >= RB) continue;  // static
            // block (T = 4 G + b, C): tri(4 G + b) = tri(4 G) + 4 G b + tri(b)
            av[G] = bg[G][16 * (m4_tri(4 * G) - BASE + 4 * j)];
        }
    };
    auto mma_row = [&](int j, const double (&av)[NG], double bx) {
        if (!row_live(j)) return;
#pragma unroll
        for (int G = 0; G < NG; ++G) {
            if (G < j || 4 * G + 3 < RA || 4 * G >= RB) continue;  // static
            double v = av[G];
            if (G == j) v = (b >= w4 && 4 * j + w4 < NB) ? v : 0.0;  // the diagonal group: C <= T, and C inside the matrix
            acc[G] = __builtin_amdgcn_mfma_f64_4x4x4f64(v, bx, acc[G], 0, 0, 0);
        }
    };
    // ---- column part: items (T = RA + 4 j + w4, H), block columns 4 H + b < T, or == T with the strict lower part
    auto load_col = [&](int j, double (&av)[NG], double& bt) {
        if (j >= NJC) return;
        const int Tu = RA + 4 * j + w4;                // (uniform)
        const bool t_ok = RA + 4 * j + 3 < RB || Tu < RB;
        const int T = t_ok ? Tu : RB - 1;
        bt = xs[16 * T + ln.xlane];
        const double* lt = (t_ok ? buf : buf + M4_LDS_OOB) + 16 * (m4_tri(T) - BASE) + ln.col0 + 4 * (ln.cbx ^ (T & 3));
#pragma unroll
        for (int H = 0; H < NG; ++H) {
            if (4 * H > RA + 4 * j + 3) continue;  // static: the whole group lies right of every T of this j
            av[H] = lt[64 * H];
        }
    };
    auto mma_col = [&](int j, const double (&av)[NG], double bt) {
        if (j >= NJC) return;
        const int T = RA + 4 * j + w4;
#pragma unroll
        for (int H = 0; H < NG; ++H) {
            if (4 * H > RA + 4 * j + 3) continue;  // static
            double v = av[H];
            if (!(4 * H + 3 < RA + 4 * j)) {  // (static: else every block column of the group is left of every T of this j)
                const int cb = 4 * H + b;
                v = (cb < T || (cb == T && c < a)) ? v : 0.0;
            }
            acc[H] = __builtin_amdgcn_mfma_f64_4x4x4f64(v, bt, acc[H], 0, 0, 0);
        }
    };
    // The four consumer waves start a chunk together (the barrier), and the chunk's operands are 25 KB of LDS reads per
    // wave: requested all at once and waited for before the first MFMA, the CU alternates between a phase in which the
    // LDS is saturated and the matrix pipe idle and one the other way round (measured with s_memtime: ~1900 cycles per
    // step where either phase alone is ~800 / ~600).  So the operands come in four batches -- the halves of the row
    // part, the halves of the column part --, two of them in flight, and a batch's MFMAs run while the next but one is
    // being read; the scheduler may not move anything across the batch boundaries (it would sort the MFMAs by
    // accumulator and wait for the last read first).
    constexpr int JR = (NJR + 1) / 2, JC = NJC / 2;
    double avr[NJR][NG], avc[NJC][NG], btc[NJC];
#pragma unroll
    for (int j = 0; j < JR; ++j) load_row(j, avr[j], bxr[j]);
#pragma unroll
    for (int j = JR; j < NJR; ++j) load_row(j, avr[j], bxr[j]);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int j = 0; j < JR; ++j) mma_row(j, avr[j], bxr[j]);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int j = 0; j < JC; ++j) load_col(j, avc[j], btc[j]);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int j = JR; j < NJR; ++j) mma_row(j, avr[j], bxr[j]);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int j = JC; j < NJC; ++j) load_col(j, avc[j], btc[j]);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int j = 0; j < JC; ++j) mma_col(j, avc[j], btc[j]);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int j = JC; j < NJC; ++j) mma_col(j, avc[j], btc[j]);
}

}  // namespace
