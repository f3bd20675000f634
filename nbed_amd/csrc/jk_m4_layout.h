// Layout of the 4 x 4-block packed tiles of jk_m4.hip and of its Dtot' weights table, shared with the kernel that
// prepares that table inside the SCF cycle (elementwise.hip huz_scalars_kernel).
#pragma once
#include <hip/hip_runtime.h>

#include <cstddef>
#include <cstdint>

namespace {

constexpr int M4_PROD_THREADS = 256, M4_NCH = 4;  // loading threads of a workgroup; chunks of a tile

__host__ __device__ constexpr int m4_tri(int k) { return k * (k + 1) / 2; }

template <int NB_>
struct M4Geom {
    static constexpr int NB = NB_, N = 4 * NB, NG = (NB + 3) / 4, NBLK = m4_tri(NB), TILE = 16 * NBLK;
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

// ---- geometry of jk_mx.hip: the same tiles for the sizes whose tile does not fit the ring in four chunks (N > 148).
// A tile is cut into NCH chunks of whole block rows, as many as it takes for a chunk to fit one of the MX_RING buffers
// that the workgroup's LDS has room for next to its other arrays; a chunk is lpt(k) LDS-DMA instructions per loading
// wave (4 KB each).
constexpr int MX_RING = 5, MX_MAXCH = 96, MX_LDS_BYTES = 160 * 1024, MX_CUS = 256;

struct MxRows {
    int r[MX_MAXCH + 1];  // r[k]: first block row of chunk k; r[n] = NB
    int n;
};

template <int NB_>
struct MxGeom {
    static constexpr int NB = NB_, N = 4 * NB, NG = (NB + 3) / 4, NBLK = m4_tri(NB), TILE = 16 * NBLK;
    static constexpr int LMAX = (m4_tri(N) + MX_CUS - 1) / MX_CUS + 1;  // tiles of a workgroup's range, at most
    // doubles of LDS besides the ring: X of two tiles, the consumers' row-q / row-p partials, J partials, J of the range
    static constexpr int FIXED = 2 * 4 * N + 2 * 4 * NG * 32 + 16 + LMAX;
    static constexpr int lptm_room() {
        int l = (MX_LDS_BYTES - 8 * FIXED) / (MX_RING * 4096);
        return l > 6 ? 6 : l;
    }
    static constexpr MxRows greedy(int cap) {
        MxRows R{};
        int k = 0, t = 0;
        R.r[0] = 0;
        while (t < NB) {
            int blk = 0;
            while (t < NB && blk + t + 1 <= cap) blk += ++t;  // (row t has t + 1 blocks)
            R.r[++k] = t;
        }
        R.n = k;
        for (int i = k + 1; i <= MX_MAXCH; ++i) R.r[i] = NB;
        return R;
    }
    static constexpr MxRows make() {  // the fewest chunks the ring buffers allow, then the smallest cap that still gives that many
        int cap = 32 * lptm_room();
        const int n = greedy(cap).n;
        while (cap - 1 >= NB && greedy(cap - 1).n == n) --cap;
        return greedy(cap);
    }
    static constexpr MxRows ROWS = make();
    static constexpr int NCH = ROWS.n;
    static_assert(NCH >= 4 && NCH <= MX_MAXCH, "chunking");
    static constexpr int row0(int k) { return k <= 0 ? 0 : (k >= NCH ? NB : ROWS.r[k]); }
    static constexpr int blocks(int k) { return m4_tri(row0(k + 1)) - m4_tri(row0(k)); }
    static constexpr int lpt(int k) { return (blocks(k) + 31) / 32; }  // 16-byte loads per loading thread (256 x 16 B = 32 blocks)
    static constexpr int lptm() {
        int m = 0;
        for (int k = 0; k < NCH; ++k) m = lpt(k) > m ? lpt(k) : m;
        return m;
    }
    static constexpr int LPTM = lptm();
    static constexpr int BUF = LPTM * M4_PROD_THREADS * 2;  // doubles per ring buffer
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
