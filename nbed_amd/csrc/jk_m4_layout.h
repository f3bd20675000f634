// Layout of the 4 x 4-block packed tiles of jk_m4.hip and of its Dtot' weights table, shared with the kernel that
// prepares that table inside the SCF cycle (elementwise.hip huz_scalars_kernel).
#pragma once
#include <hip/hip_runtime.h>

#include <cstddef>
#include <cstdint>

namespace {

constexpr int M4_PROD_THREADS = 256, M4_NCH = 4;  // loading threads of a workgroup; chunks of a tile

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
    static constexpr int LPT = (max_blocks() * 128 + M4_PROD_THREADS * 16 - 1) / (M4_PROD_THREADS * 16);  // 16-byte loads per producer thread
    static constexpr int BUF = LPT * M4_PROD_THREADS * 2;                                            // doubles per LDS buffer
};

// Where the weight of the tile element (row, col <= row) sits in the table the main kernel keeps in registers:
// table[(k LPT + s) 512 + 2 tid + e] belongs to double e of the 16 bytes thread tid loads in slot s of chunk k, so an
// element at offset d of the tile that chunk k (block rows from row0(k)) holds is at k LPT 512 + d - 16 tri(row0(k)).
// Block (T, C) of the tile at 16 (tri(T) + C), element (i, k) at 4 (k ^ ((T ^ C) & 3)) + (i ^ k).
template <int NB>
__host__ __device__ __forceinline__ int m4_weight_index(int row, int col) {
    using G = M4Geom<NB>;
    static_assert(M4_NCH == 4, "three chunk boundaries below");
    constexpr int r1 = G::row0(1), r2 = G::row0(2), r3 = G::row0(3);
    const int bt = row >> 2, bc = col >> 2, ii = row & 3, kk = col & 3;
    const int k = (bt >= r1) + (bt >= r2) + (bt >= r3);
    const int rk = k == 0 ? 0 : k == 1 ? r1 : k == 2 ? r2 : r3;
    const int d = 16 * (m4_tri(bt) + bc) + 4 * (kk ^ ((bt ^ bc) & 3)) + (ii ^ kk);
    return k * G::LPT * M4_PROD_THREADS * 2 + d - 16 * m4_tri(rk);
}

// the same with the chunk boundaries (r1, r2, r3 = first block rows of chunks 1..3) and LPT as run-time values
// (nbx_jk_m4_weight_layout): the scalars kernel serves every size with one instance
__host__ __device__ __forceinline__ int m4_weight_index_rt(int r1, int r2, int r3, int lpt, int row, int col) {
    const int bt = row >> 2, bc = col >> 2, ii = row & 3, kk = col & 3;
    const int k = (bt >= r1) + (bt >= r2) + (bt >= r3);
    const int rk = k == 0 ? 0 : k == 1 ? r1 : k == 2 ? r2 : r3;
    const int d = 16 * (m4_tri(bt) + bc) + 4 * (kk ^ ((bt ^ bc) & 3)) + (ii ^ kk);
    return k * lpt * M4_PROD_THREADS * 2 + d - 16 * m4_tri(rk);
}

}  // namespace

// jk_m4.hip
void nbx_jk_m4_weight_layout(int64_t N, int out[4]);
bool nbx_jk_m4_covers(int64_t N);
size_t nbx_jk_m4_weights_bytes(int64_t N);
