// Geometry of jk_m8.hip: the 8-fold packed form of (pq|rs) -- of tile (p, q <= p) only the elements (rs) <= (pq), i.e.
// the rows r < p of its (r, s <= r) triangle and the columns s <= q of row p -- in the 4 x 4 blocks, the swizzle and the
// block order of jk_m4_layout.h.
//
// A FULL tile is NCH chunks of whole block rows, each at most 32 LP blocks (LP LDS-DMA instructions per loading wave,
// the same number for every chunk: what is in flight is then known without knowing which chunks).  Tile (p, q) is stored
// as the chunks 0 .. nk(p) - 1, nk(p) - 1 = the chunk that holds block row p / 4 (the rows of that chunk below p are stored
// as zeros): tiles are of nk-dependent length, so rows p fall into NCH GROUPS of equal tile length, and both the
// address of a tile and the cost of the tiles before it are linear in the tile index T = p (p + 1) / 2 + q inside a group.
#pragma once
#include "jk_m4_layout.h"

namespace {

// (the loops over the chunk table are unrolled in device code -- constants instead of a table in memory; the host pass
//  is told nothing: it warns when it cannot)
#ifdef __HIP_DEVICE_COMPILE__
#define M8_UNROLL _Pragma("unroll")
#else
#define M8_UNROLL
#endif
constexpr int M8_MAXCH = 16, M8_LDS_BYTES = 160 * 1024, M8_CUS = 256;
#ifndef NBX_M8_TILE_COST
#define NBX_M8_TILE_COST 4
#endif
constexpr int M8_TILE_COST = NBX_M8_TILE_COST;  // cost of a tile of nk chunks in the split of the tile sequence: 4 nk + M8_TILE_COST

template <int NB_, int LP_>
struct M8Geom {
    static constexpr int NB = NB_, LP = LP_, N = 4 * NB, NG = (NB + 3) / 4, NBLK = m4_tri(NB), TILE = 16 * NBLK;
    static constexpr int CAP = 32 * LP;                        // blocks per chunk, at most
    static constexpr int BUF = LP * M4_PROD_THREADS * 2;       // doubles per ring buffer
    struct Table {
        int n;
        int row[M8_MAXCH + 1];  // chunk k = block rows [row[k], row[k + 1])
    };
    static constexpr Table build() {
        Table t{};
        int r = 0, k = 0;
        while (r < NB) {
            int blk = 0;
            if (k <= M8_MAXCH) t.row[k] = r;
            while (r < NB && blk + r + 1 <= CAP) blk += ++r;
            ++k;
        }
        t.n = k;
        for (int i = k; i <= M8_MAXCH; ++i) t.row[i] = NB;
        return t;
    }
    static constexpr Table TB = build();
    static constexpr int NCH = TB.n;
    static_assert(NB + 1 <= CAP, "a block row fits a chunk");
    static_assert(NCH >= 1 && NCH <= M8_MAXCH, "chunking");
    // (jk_m4_walk.h's interface)
    static constexpr int row0(int k) { return k <= 0 ? 0 : (k >= NCH ? NB : TB.row[k]); }
    static constexpr int blocks(int k) { return m4_tri(row0(k + 1)) - m4_tri(row0(k)); }
    static constexpr int start(int k) { return m4_tri(row0(k)); }        // first block of chunk k in the tile
    static constexpr int len(int nk) { return 16 * m4_tri(row0(nk)); }  // doubles of a tile of nk chunks
    // rows p of group g (g = 0 .. NCH - 1): [pb(g), pb(g + 1)), tiles of g + 1 chunks
    static constexpr int pb(int g) { return 4 * row0(g) < N ? 4 * row0(g) : N; }
    // LDS besides the ring (doubles): X of two tiles, three buffers of partial rows, J partials; the J of a range's tiles
    // comes on top (M8Plan)
    static constexpr int FIXED = 2 * 4 * N + 3 * 4 * NG * 32 + 16;
    static constexpr int LMAX_GUESS = 512;
    static constexpr int ring() {
        int r = (M8_LDS_BYTES - 8 * (FIXED + LMAX_GUESS)) / (8 * BUF);
        const int want = (96 * 1024 + 8 * BUF - 1) / (8 * BUF) + 1;  // ~96 KB in flight behind the chunk being walked
        return r < want ? r : want;
    }
    static constexpr int RING = ring();
    static_assert(RING >= 4 && (RING - 2) * LP <= 63, "ring / vmcnt");
};

// chunks of the tiles of row p
template <class G>
__host__ __device__ __forceinline__ int m8_nk(int p) {
    const int bp = p >> 2;
    int nk = 1;
M8_UNROLL
    for (int k = 1; k < G::NCH; ++k) nk += G::row0(k) <= bp ? 1 : 0;
    return nk;
}

// doubles from the first tile of the whole sequence (T = 0) to tile T
template <class G>
__host__ __device__ __forceinline__ int64_t m8_tile_offset(int64_t T) {
    int64_t off = 0;
M8_UNROLL
    for (int g = 0; g < G::NCH; ++g) {
        const int64_t t0 = m4_tri(G::pb(g)), t1 = m4_tri(G::pb(g + 1));
        const int64_t n = T <= t0 ? 0 : (T < t1 ? T - t0 : t1 - t0);
        off += n * G::len(g + 1);
    }
    return off;
}

// cost of the tiles before T
template <class G>
__host__ __device__ __forceinline__ int64_t m8_cost_before(int64_t T) {
    int64_t c = 0;
M8_UNROLL
    for (int g = 0; g < G::NCH; ++g) {
        const int64_t t0 = m4_tri(G::pb(g)), t1 = m4_tri(G::pb(g + 1));
        const int64_t n = T <= t0 ? 0 : (T < t1 ? T - t0 : t1 - t0);
        c += n * (4 * (g + 1) + M8_TILE_COST);
    }
    return c;
}

// The split of the tiles [t_begin, t_end) over W workgroups at equal cost: workgroup w has the tiles T with
// floor((cost_before(T) - cost_before(t_begin)) W / total) == w; m8_first_tile(w) = the first of them (t_end for w >= W).
template <class G>
__host__ __device__ __forceinline__ int64_t m8_first_tile(int64_t t_begin, int64_t t_end, int W, int w) {
    if (w <= 0) return t_begin;
    if (w >= W) return t_end;
    const int64_t c0 = m8_cost_before<G>(t_begin), total = m8_cost_before<G>(t_end) - c0;
    // the smallest T with (cost_before(T) - c0) W >= w total, i.e. cost_before(T) >= c0 + ceil(w total / W)
    const int64_t target = c0 + (w * total + W - 1) / W;
    int64_t T = t_end;
    int64_t cb = 0;  // cost before the group
    bool found = false;
M8_UNROLL
    for (int g = 0; g < G::NCH; ++g) {
        const int64_t t0 = m4_tri(G::pb(g)), t1 = m4_tri(G::pb(g + 1));
        const int64_t cg = 4 * (g + 1) + M8_TILE_COST;
        const int64_t cend = cb + (t1 - t0) * cg;
        if (!found && target <= cend) {
            const int64_t need = target > cb ? target - cb : 0;
            T = t0 + (need + cg - 1) / cg;
            found = true;
        }
        cb = cend;
    }
    T = T < t_begin ? t_begin : T;
    return T > t_end ? t_end : T;
}
template <class G>
__host__ __device__ __forceinline__ int m8_wg_of(int64_t t_begin, int64_t t_end, int W, int64_t T) {
    const int64_t c0 = m8_cost_before<G>(t_begin), total = m8_cost_before<G>(t_end) - c0;
    const int64_t w = (m8_cost_before<G>(T) - c0) * W / total;
    return (int)(w < W ? w : W - 1);
}

// where the element (row, col <= row) of a FULL tile sits in the staging order of the loading waves (the order of the
// Dtot' weights table and of the J partials): slot s of chunk k is thread ptid's two doubles (k LP + s) 512 + 2 ptid + e
template <class G>
__host__ __device__ __forceinline__ int m8_stage_index(int row, int col) {
    const int bt = row >> 2, bc = col >> 2, ii = row & 3, kk = col & 3;
    int k = 0;
M8_UNROLL
    for (int c = 1; c < G::NCH; ++c) k += G::row0(c) <= bt ? 1 : 0;
    int st = 0;
M8_UNROLL
    for (int c = 1; c < G::NCH; ++c) st = (c == k) ? G::start(c) : st;
    const int d = 16 * (m4_tri(bt) + bc - st) + 4 * (kk ^ ((bt ^ bc) & 3)) + (ii ^ kk);
    return k * G::LP * M4_PROD_THREADS * 2 + d;
}

}  // namespace
