// Geometry of jk_m8.hip: the 8-fold packed form of (pq|rs) -- of tile (p, q <= p) only the elements (rs) <= (pq), i.e.
// the rows r < p of its (r, s <= r) triangle and the columns s <= q of row p -- in the 4 x 4 blocks, the swizzle and the
// block order of jk_m4_layout.h.
//
// A FULL tile is NCH chunks of whole block rows, each at most 32 LP blocks (LP LDS-DMA instructions per loading wave,
// the same number for every chunk: what is in flight is then known without knowing which chunks).  Tile (p, q) is stored
// and streamed as the chunks 0 .. nk(p) - 1, nk(p) - 1 = the chunk that holds block row p / 4 (the rows of that chunk
// below p are stored as zeros).  Tiles are of nk-dependent length, the same for the four rows p of a block row, and the
// address of a tile is linear in the tile index T = p (p + 1) / 2 + q inside such a group of rows.
// (Cutting the stored tile at block row p / 4 -- the last chunk loaded up to the tile's end only, the X rows beyond zeroed,
// the J terms masked -- was built and measured: 13 % fewer bytes, no faster: a step costs its walk and its J pass over
// the whole chunk, whatever part of the chunk is the tile's.)
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

// chunks of a full tile of NB block rows at LP LDS-DMA instructions per loading wave and chunk (M8Geom::build's count)
constexpr int m8_nch(int NB, int LP) {
    int r = 0, k = 0;
    while (r < NB) {
        int blk = 0;
        while (r < NB && blk + r + 1 <= 32 * LP) blk += ++r;
        ++k;
    }
    return k;
}
// The chunk size of an instance: the smallest with at most four chunks per full tile -- a step then lasts as long as
// jk_m4.hip's (what a tile's hand-over costs is spread over steps of that length: measured at N = 148 with LP = 3 .. 7),
// and with exactly four chunks the Dtot' table has jk_m4.hip's order (nbx_jk_m8_weight_layout).
constexpr int m8_lp(int NB) {
    int lp = (NB + 1 + 31) / 32;
    while (m8_nch(NB, lp) > 4) ++lp;
    return lp;
}

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
    static constexpr int len_nk(int nk) { return 16 * m4_tri(row0(nk)); }  // doubles of a tile of nk chunks
    // LDS besides the ring (doubles): X of two tiles, three buffers of partial rows, J partials
    static constexpr int FIXED = 2 * 4 * N + 3 * 4 * NG * 32 + 16;
    static constexpr int ring() {
        int r = (M8_LDS_BYTES - 8 * FIXED) / (8 * BUF);
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

// doubles of the tiles of row p
template <class G>
__host__ __device__ __forceinline__ int m8_len(int p) {
    const int nk = m8_nk<G>(p);
    int ln = G::len_nk(1);
M8_UNROLL
    for (int k = 2; k <= G::NCH; ++k) ln = nk == k ? G::len_nk(k) : ln;
    return ln;
}

// doubles from the first tile of the whole sequence (T = 0) to tile T
template <class G>
__host__ __device__ __forceinline__ int64_t m8_tile_offset(int64_t T) {
    int64_t off = 0;
    for (int b = 0; b < G::NB; ++b) {  // the rows 4 b .. 4 b + 3
        const int64_t t0 = m4_tri(4 * b), t1 = m4_tri(4 * b + 4);
        const int64_t n = T <= t0 ? 0 : (T < t1 ? T - t0 : t1 - t0);
        off += n * m8_len<G>(4 * b);
    }
    return off;
}

// ---- the ORDER of the tiles.  The rows p come in BANDS of four (4 b .. 4 b + 3: one tile length), and inside a band the
// tiles are visited column group by column group: (lo, q), (lo + 1, q), .., (hi, q), then q + 1 -- of the rows of the band
// that exist in the slab [p0, p1) and have p >= q.  The four tiles of a group feed the SAME row q of K (their row-q sums
// stay in the walking waves' accumulators across the group: one hand-over per group, not per tile) and four different
// rows p (kept in four register sets, handed over when a row reaches its diagonal tile).  A band starts at the sequence
// number its first row has in the row-major numbering T = p (p + 1) / 2 + q, so the sequence number S of a tile differs
// from T inside a band only, and tile addresses (m8_tile_offset) are the same function of S as of T.
struct M8Band {
    int b, lo, hi;  // rows lo .. hi of band b exist
};
__host__ __device__ __forceinline__ M8Band m8_band_of_row(int p, int p0, int p1) {
    M8Band B;
    B.b = p >> 2;
    B.lo = 4 * B.b > p0 ? 4 * B.b : p0;
    B.hi = 4 * B.b + 3 < p1 - 1 ? 4 * B.b + 3 : p1 - 1;
    return B;
}
__host__ __device__ __forceinline__ int m8_tri_row_i(int64_t T) {
    int64_t p = (int64_t)((sqrt(8.0 * (double)T + 1.0) - 1.0) * 0.5);
    while (p * (p + 1) / 2 > T) --p;
    while ((p + 1) * (p + 2) / 2 <= T) ++p;
    return (int)p;
}
// sequence number S (absolute: m4_tri(p0) is the slab's first) -> (p, q)
__host__ __device__ __forceinline__ void m8_pq_of(int64_t S, int p0, int p1, int& p, int& q) {
    const M8Band B = m8_band_of_row(m8_tri_row_i(S), p0, p1);
    const int R = B.hi - B.lo + 1;
    int pos = (int)(S - m4_tri(B.lo));
    if (pos < (B.lo + 1) * R) {
        q = pos / R;
        p = B.lo + pos - q * R;
        return;
    }
    pos -= (B.lo + 1) * R;
    q = B.lo + 1;
    while (pos >= B.hi - q + 1) {
        pos -= B.hi - q + 1;
        ++q;
    }
    p = q + pos;
}
// (p, q) -> S
__host__ __device__ __forceinline__ int64_t m8_seq_of(int p, int q, int p0, int p1) {
    const M8Band B = m8_band_of_row(p, p0, p1);
    const int R = B.hi - B.lo + 1;
    if (q <= B.lo) return (int64_t)m4_tri(B.lo) + q * R + (p - B.lo);
    int pos = (B.lo + 1) * R;
    for (int qq = B.lo + 1; qq < q; ++qq) pos += B.hi - qq + 1;
    return (int64_t)m4_tri(B.lo) + pos + (p - q);
}
// the tile after (p, q)
__host__ __device__ __forceinline__ void m8_next(int& p, int& q, int p0, int p1) {
    const M8Band B = m8_band_of_row(p, p0, p1);
    if (p < B.hi) {
        ++p;
        return;
    }
    ++q;
    if (q <= B.hi) {
        p = q > B.lo ? q : B.lo;
        return;
    }
    p = B.hi + 1;  // the next band's first row (p1: the end)
    q = 0;
}
// the first sequence number of the column group (b, q) of the slab, counted in groups: group index for the row-q partials
__host__ __device__ __forceinline__ int m8_group_index(int b, int q, int p0, int p1) {
    const int b0 = p0 >> 2;
    int g = 0;
    for (int bb = b0; bb < b; ++bb) g += (4 * bb + 3 < p1 - 1 ? 4 * bb + 3 : p1 - 1) + 1;  // groups q = 0 .. hi of band bb
    return g + q;
}
__host__ __device__ __forceinline__ int m8_group_count(int p0, int p1) {
    return p1 > p0 ? m8_group_index((p1 - 1) >> 2, 0, p0, p1) + p1 : 0;  // (the last band's groups q = 0 .. p1 - 1)
}

// The split of the tiles [t_begin, t_end) over the workgroups: workgroup w has the tiles t_begin + first[w] .. first[w + 1] - 1
// (made on the host at equal cost, jk_m8.hip m8_ranges; a kernel ARGUMENT: 1 KB of scalar loads, nothing to allocate or copy)
struct M8Ranges {
    int first[M8_CUS + 1];
    int wmin[M8_MAXCH];  // the first workgroup whose range has a tile with chunk k (the J partials of chunk k exist from there on)
};
// the workgroup that has tile t_begin + trel
__device__ __forceinline__ int m8_wg_of(const M8Ranges& rg, int W, int trel) {
    int lo = 0, hi = W - 1;  // the largest w with first[w] <= trel
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (rg.first[mid] <= trel) lo = mid;
        else hi = mid - 1;
    }
    return lo;
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
