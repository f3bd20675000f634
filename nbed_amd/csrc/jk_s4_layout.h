// The packed J/K tile format of jk_s4.hip (include/nbx.h "J/K contraction, packed form") and the
// order in which its staging slots hold a chunk: shared with the kernel that prepares the Dtot'
// table for the next build (elementwise.hip, huz_scalars_kernel).
#pragma once
#include <cmath>
#include <cstdint>

#include <hip/hip_runtime.h>

__host__ __device__ __forceinline__ static int64_t s4_tri(int64_t k) { return k * (k + 1) / 2; }
__host__ __device__ __forceinline__ static int s4_tri_row(int64_t T) {
    int64_t p = (int64_t)((sqrt(8.0 * (double)T + 1.0) - 1.0) * 0.5);
    while (p * (p + 1) / 2 > T) --p;
    while ((p + 1) * (p + 2) / 2 <= T) ++p;
    return (int)p;
}

struct S4Geom {
    int N, NB, s, ls, tri;
    int E0, Er;  // doubles in the diagonal chunk / in a rectangle chunk (both even)
    int64_t M;   // doubles per tile
};

__host__ __device__ __forceinline__ static S4Geom s4_geom(int N, int NB) {
    S4Geom g;
    g.N = N;
    g.NB = NB;
    g.s = N / NB;
    g.ls = g.s | 1;
    g.tri = g.s * (g.s + 1) / 2;
    g.E0 = (NB * g.tri + 1) & ~1;
    g.Er = ((NB / 2) * g.s * g.ls + 1) & ~1;
    g.M = (int64_t)g.E0 + (int64_t)(NB - 1) * g.Er;
    return g;
}

// position of the rectangle (lo, lo ^ r), lo < lo ^ r, among the NB/2 rectangles of round r: in
// ascending order of the row block lo ^ r, so that the rows a <= p of a chunk are a PREFIX of it (the
// 8-fold form stores and reads only that prefix, jk_p8.hip)
__host__ __device__ __forceinline__ static int s4_slot(int lo, int r) {
    if (r == 3) return 1 - lo;  // NB = 4: (2,1) before (3,0)
    int hb = 0;
    while ((r >> (hb + 1)) != 0) ++hb;  // highest set bit of r: clear in lo
    return ((lo >> (hb + 1)) << hb) | (lo & ((1 << hb) - 1));
}

// offset of (a, b), b <= a, inside a tile
__host__ __device__ __forceinline__ static int64_t s4_flat(const S4Geom& g, int a, int b) {
    const int I = a / g.s, J = b / g.s, ai = a - I * g.s, bi = b - J * g.s;
    if (I == J) return (int64_t)I * g.tri + s4_tri(ai) + bi;
    const int r = I ^ J;
    return (int64_t)g.E0 + (int64_t)(r - 1) * g.Er + (int64_t)s4_slot(J, r) * g.s * g.ls + (int64_t)ai * g.ls + bi;
}

// Staging slots.  Wave w moves a chunk's pairs (16 bytes) in LPT loads of 64 pairs: slot (w, k)
// starts at pair 64 (LPT w + k); a slot that would run past the chunk end is pulled back to end
// exactly there: its first lanes repeat pairs of the slot before it -- same bytes to the same LDS
// address, and a zero weight in the J sum.  A slot that starts past the end (-1) reads a line that
// is always in cache (the head of the Dtot' table) into an LDS scratch area, also with weight 0.
// Everything is unconditional and wave-uniform: one address VGPR (the lane) serves every load.
__host__ __device__ __forceinline__ static int s4_slot_start(int ne, int lpt, int w, int k) {
    const int np2 = ne >> 1, ps = 64 * (lpt * w + k);
    return ps >= np2 ? -1 : (ps + 64 > np2 ? np2 - 64 : ps);
}


// Offset of (a, b), b <= a, in the Dtot' table nbx_jk_packed multiplies the staged tile with
// (doubles; the table is NB chunks of NB * lpt * 128): the staging order is the tile order except in
// the last, pulled-back slot of a chunk.
__host__ __device__ __forceinline__ static int64_t s4_dts_index(const S4Geom& g, int lpt, int a, int b) {
    const int64_t f = s4_flat(g, a, b);
    const int ch = f < g.E0 ? 0 : 1 + (int)((f - g.E0) / g.Er);
    const int fin = (int)(f - (ch == 0 ? 0 : g.E0 + (int64_t)(ch - 1) * g.Er));
    const int np2 = (ch == 0 ? g.E0 : g.Er) >> 1, pp = fin >> 1, sl = pp >> 6;
    const int ps = 64 * sl + 64 > np2 ? np2 - 64 : 64 * sl;
    return (int64_t)ch * g.NB * lpt * 128 + 2 * (64 * sl + (pp - ps)) + (fin & 1);
}

// four blocks when the chunks of N / 4 rows still fill a staging slot (128 doubles), else two
static inline int s4_nb(int64_t N) {
    if (N % 4 == 0) {
        const S4Geom g = s4_geom((int)N, 4);
        if (g.E0 >= 128 && g.Er >= 128) return 4;
    }
    return 2;
}

// template instances: loads per thread per chunk
static inline int s4_lpt_class(int lpt, int NB) {
    return lpt <= 2 ? 2 : lpt <= 6 ? 6 : lpt <= 10 ? 10 : (lpt <= 17 && NB == 4) ? 17 : 0;
}

static inline bool s4_supported(int64_t N) {
    if (N < 16 || N > 256 || N % 2 != 0) return false;  // a walk group is four steps: s >= 4
    const int NB = s4_nb(N);
    if (N % NB != 0 || N / NB > 64) return false;  // one wave walks a block
    const S4Geom g = s4_geom((int)N, NB);
    const int need = (int)(((g.E0 > g.Er ? g.E0 : g.Er) / 2 + NB * 64 - 1) / (NB * 64));
    return s4_lpt_class(need, NB) != 0 && g.E0 >= 128 && g.Er >= 128;  // a staging slot is 64 pairs
}


// Sizes the kernel has no instance for (odd N, N = 102, 150, ...) run as the next covered size with
// the extra rows and columns zero: at most S4_MAX_PAD of them, so that the padding stays a few per
// cent of the tile (0: no covered size that close).
constexpr int S4_MAX_PAD = 8;
static inline int64_t s4_padded(int64_t N) {
    if (N < 1) return 0;
    for (int64_t M = N; M <= N + S4_MAX_PAD && M <= 256; ++M)
        if (s4_supported(M)) return M;
    return 0;
}

// loads per thread per chunk of the kernel instance that serves N (0: not covered)
static inline int s4_lpt(int64_t N) {
    if (!s4_supported(N)) return 0;
    const int NB = s4_nb(N);
    const S4Geom g = s4_geom((int)N, NB);
    const int need = (int)(((g.E0 > g.Er ? g.E0 : g.Er) / 2 + NB * 64 - 1) / (NB * 64));
    return s4_lpt_class(need, NB);
}
