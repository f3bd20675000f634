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

// ---- geometry of jk_mx.hip: the same 4 x 4 blocks for the sizes whose tile does not fit the ring in four chunks
// (N > 148).  A tile is cut into NCH chunks, as many as it takes for a chunk to fit one of the MX_RING buffers that the
// workgroup's LDS has room for next to its other arrays; a chunk is lpt(k) LDS-DMA instructions per loading wave (4 KB
// each).  Two kinds of chunk:
//   WHOLE  block rows [ra, rb) in full, block (T, C) at start + tri(T) - tri(ra) + C -- the layout of jk_m4.hip; the top
//          of the triangle, where several rows fit a buffer;
//   BAND   the column groups [j0, j1) (block columns 4 j0 .. 4 j1 - 1, up to the diagonal) of the four block rows of
//          band g (rows 4 g .. 4 g + 3), row after row -- further down, where a buffer holds less than a band: every
//          MFMA of the walk then still works on four live block rows, and each walking wave has one row of the column
//          part (whole rows there would leave one row per chunk: a quarter of the row part's lanes and one wave of
//          four in the column part).  Used from the first band that does not fit a buffer, for sizes above MX_SPLIT_NB.
// Either way the chunks follow each other in the tile, which stays tri(NB) blocks long.
constexpr int MX_RING = 5, MX_MAXCH = 96, MX_LDS_BYTES = 160 * 1024, MX_CUS = 256;
#ifndef NBX_MX_SPLIT_NB
#define NBX_MX_SPLIT_NB 72
#endif
constexpr int MX_SPLIT_NB = NBX_MX_SPLIT_NB;  // sizes up to N = 288 keep whole rows all the way down (measured at 0.67-0.74 of 8 TB/s)

struct MxChunk {
    short band;        // 0: WHOLE, 1: BAND
    short ra, rb;      // block rows [ra, rb)
    short j0, j1;      // BAND: column groups [j0, j1) of band ra / 4 (j1 == ra / 4 + 1: the segment with the diagonal)
    int start, blocks; // first block of the chunk in the tile, number of blocks
};
struct MxChunks {
    MxChunk c[MX_MAXCH];
    int n;
};

// blocks of rows 4 g .. 4 g + 3 in column groups [j0, j1) (j1 <= g + 1)
__host__ __device__ constexpr int mx_band_blocks(int g, int j0, int j1) {
    int n = 0;
    for (int b = 0; b < 4; ++b) {
        const int last = 4 * j1 - 1 < 4 * g + b ? 4 * j1 - 1 : 4 * g + b;  // last block column of row 4 g + b in the segment
        n += last - 4 * j0 + 1;
    }
    return n;
}
// first block (relative to the segment) of row 4 g + b
__host__ __device__ constexpr int mx_band_rowstart(int g, int j0, int j1, int b) {
    return j1 == g + 1 ? b * 4 * (g - j0) + m4_tri(b) : b * 4 * (j1 - j0);
}

template <int NB_>
struct MxGeom {
    static constexpr int NB = NB_, N = 4 * NB, NG = (NB + 3) / 4, NBLK = m4_tri(NB), TILE = 16 * NBLK;
    static constexpr int LMAX = (m4_tri(N) + MX_CUS - 1) / MX_CUS + 1;  // tiles of a workgroup's range, at most
    // doubles of LDS besides the ring: X of two tiles, the consumers' partial rows (one buffer), J partials, J of the range
    static constexpr int FIXED = 2 * 4 * N + 4 * NG * 32 + 16 + LMAX;
    static constexpr int lptm_room() {
        int l = (MX_LDS_BYTES - 8 * FIXED) / (MX_RING * 4096);
        return l > 6 ? 6 : l;
    }
    // first band that is cut into segments (NB / 4: none)
    static constexpr int first_split_band(int cap) {
        if (NB <= MX_SPLIT_NB) return NB / 4 + 1;
        int g = 0;
        while (g < NB / 4 && 16 * g + 10 <= cap) ++g;
        return g;
    }
    static constexpr MxChunks build(int cap) {
        MxChunks R{};
        int k = 0;
        const int gs = first_split_band(cap);
        const int rs = 4 * gs < NB ? 4 * gs : NB;  // whole rows up to here
        auto whole = [&](int t0, int t1) {
            int t = t0;
            while (t < t1) {
                int blk = 0;
                const int ra = t;
                while (t < t1 && blk + t + 1 <= cap) blk += ++t;  // (row t has t + 1 blocks)
                if (k < MX_MAXCH) R.c[k] = MxChunk{0, (short)ra, (short)t, 0, 0, m4_tri(ra), blk};
                ++k;
            }
        };
        whole(0, rs);
        for (int g = gs; g < NB / 4; ++g) {
            int nseg = (16 * g + 10 + cap - 1) / cap;
            while (16 * ((g + 1 + nseg - 1) / nseg) > cap) ++nseg;  // (a segment of ceil((g + 1) / nseg) full groups must fit)
            int start = m4_tri(4 * g);
            for (int i = 0; i < nseg; ++i) {
                const int j0 = (g + 1) * i / nseg, j1 = (g + 1) * (i + 1) / nseg;
                const int blk = mx_band_blocks(g, j0, j1);
                if (k < MX_MAXCH) R.c[k] = MxChunk{1, (short)(4 * g), (short)(4 * g + 4), (short)j0, (short)j1, start, blk};
                ++k;
                start += blk;
            }
        }
        if (rs < NB && 4 * (NB / 4) < NB) whole(4 * (NB / 4), NB);  // (the rows of an incomplete last band)
        R.n = k;
        return R;
    }
    static constexpr MxChunks make() {  // the fewest chunks the ring buffers allow, then the smallest cap that still gives that many
        int cap = 32 * lptm_room();
        const int n = build(cap).n;
        while (cap - 1 >= NB && build(cap - 1).n == n) --cap;
        return build(cap);
    }
    static constexpr MxChunks CH = make();
    static constexpr int NCH = CH.n;
    static_assert(NCH >= 4 && NCH <= MX_MAXCH, "chunking");
    // (for the whole-row walk, jk_m4_walk.h: chunk k holds the block rows [row0(k), row0(k + 1)))
    static constexpr int row0(int k) { return k <= 0 ? 0 : (k >= NCH ? NB : CH.c[k].ra); }
    static constexpr int blocks(int k) { return CH.c[k].blocks; }
    static constexpr int lpt(int k) { return (blocks(k) + 31) / 32; }  // 16-byte loads per loading thread (256 x 16 B = 32 blocks)
    static constexpr int lptm() {
        int m = 0;
        for (int k = 0; k < NCH; ++k) m = lpt(k) > m ? lpt(k) : m;
        return m;
    }
    static constexpr int LPTM = lptm();
    static constexpr int BUF = LPTM * M4_PROD_THREADS * 2;  // doubles per ring buffer
};

// block (T, C <= T) of the tile -> its chunk and its block offset inside the chunk
__host__ __device__ inline void mx_locate(const MxChunks& ch, int T, int C, int& k, int& off) {
    for (k = 0; k < ch.n; ++k) {
        const MxChunk& c = ch.c[k];
        if (T < c.ra || T >= c.rb) continue;
        if (!c.band) {
            off = m4_tri(T) - m4_tri(c.ra) + C;
            return;
        }
        if (C >= 4 * c.j0 && C < 4 * c.j1) {
            off = mx_band_rowstart(c.ra / 4, c.j0, c.j1, T - c.ra) + C - 4 * c.j0;
            return;
        }
    }
    k = -1;
    off = 0;
}
// block offset `off` of chunk k -> (T, C)
__host__ __device__ inline void mx_block_of(const MxChunks& ch, int k, int off, int& T, int& C) {
    const MxChunk& c = ch.c[k];
    if (!c.band) {
        const int blk = m4_tri(c.ra) + off;
        T = c.ra;
        while (m4_tri(T + 1) <= blk) ++T;
        C = blk - m4_tri(T);
        return;
    }
    const int g = c.ra / 4;
    int b = 3;
    while (b > 0 && mx_band_rowstart(g, c.j0, c.j1, b) > off) --b;
    T = c.ra + b;
    C = 4 * c.j0 + off - mx_band_rowstart(g, c.j0, c.j1, b);
}

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

// jk_m8.hip (the 8-fold form: takes the size when it covers it)
bool nbx_jk_m8_covers(int64_t N);
void nbx_jk_m8_weight_layout(int64_t N, int out[4]);
size_t nbx_jk_m8_weights_bytes(int64_t N);
// jk_m4.hip
void nbx_jk_m4_weight_layout(int64_t N, int out[4]);
bool nbx_jk_m4_covers(int64_t N);
size_t nbx_jk_m4_weights_bytes(int64_t N);
