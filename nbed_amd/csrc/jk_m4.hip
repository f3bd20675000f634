// libnbx: the packed J/K contraction with its "walk" on the matrix cores (include/nbx.h "J/K contraction, packed form").
//
// jk_s4.hip streams the 4-fold packed tiles at 0.55 of the HBM roofline because the two symmetric matrix x vector
// products per spin that every tile feeds (K[p,:] += M d_q, K[q,:] += M d_p, M = the (r,s) matrix of the tile) are
// walked on the vector ALU: 36-50 instructions per 16 FMAs with 37 of 64 lanes live at N = 148.
// Here the same products run as  out[t][0:4] = sum_c M[t][c] X[c][0:4],  X[c] = (D^a_q, D^a_p, D^b_q, D^b_p)[c],  on
// v_mfma_f64_4x4x4_4b_f64: four independent 4 x 4 x 4 products per instruction, 256 multiply-adds, all of them useful
// (the 16x16x4 form would leave 12 of its 16 columns idle).  Measured in isolation (profiles/r03/
// jk_mfma_walk_probe.hip): 1.65 us per tile per CU with four waves against the 3.5 us at which tiles stream through a
// CU -- the walk fits under the stream.
//
// Layout (nbx_eri_pack makes it for the sizes this kernel serves): the tiles T(p,q) = p(p+1)/2 + q, q <= p, in
// sequence; a tile is the lower triangle of M in 4 x 4 BLOCKS, block (T, C <= T) at T(T+1)/2 + C, 16 doubles each,
// element (row i, column k) at 4 (k ^ ((T ^ C) & 3)) + (i ^ k), the upper part of the diagonal blocks stored as zeros.
// (The swizzle: an LDS pass serves 16 lanes = one value of a; with the plain order 4 k + i the row part's four 32-byte
// runs start at the same bank and the column part's 16 elements sit 32 bytes apart -- 4-way conflicts both, more than
// half of the LDS cycles of the kernel by the SQ counters, profiles/r03/jk_m4_sq_counters.txt.  With it the four runs
// of the row part land in four different quarters of the 128-byte bank row and the column part's 16 elements in 16
// different 8-byte slots.)  With lane = 16 a + 4 b + c the operands of one MFMA are
//   row part    item (G, C):  A = block (4 G + b, C) element (c, a)    B = X[4 C + a][c]   D -> out rows 16 G + 4 b + a
//   column part item (T, H):  A = block (T, 4 H + b) element (a, c)    B = X[4 T + a][c]   D -> out rows 16 H + 4 b + a
// -- the column part reads four consecutive blocks (512 contiguous bytes, conflict free), the row part four runs of
// 128 bytes.  (Both MFMA operands carry the contraction index in lane >> 4: one register can be contracted over one of
// its two indices only, so the column part is a second LDS read of the block in the transposed lane map; the triangle
// is read twice per tile.)  Wave w of the four takes block columns / block rows = w (mod 4), which makes every loop bound a
// compile-time constant: straight-line code, LDS reads batched ahead of the MFMAs by the compiler.
//
// Streaming, work distribution and reductions: persistent workgroups over equal contiguous ranges of the tile sequence,
// one per CU, eight waves in two roles (see the kernel); a tile arrives in four chunks (whole block rows, equal block
// counts) by LDS-DMA -- global_load_lds_dwordx4, HBM straight into a ring of five LDS buffers, three chunks in flight
// behind the one that has landed --, one barrier per chunk; J is a flat dot product of the chunk (read back from LDS by
// the loading waves) with a Dtot' table held in registers; row-q partials go out per tile, row-p partials when the range
// crosses a row, both summed in a fixed order by jk_sym_reduce_kernel (with the Fock epilogue), as in jk_s4.hip.
// This is the production kernel of the sizes nbx_jk_m4_covers() names (N = 100 .. 148 in steps of four, one instance each:
// the tile has to fit the five-buffer ring in four chunks and its Dtot' table a loading wave's registers); DESIGN.md section 9 has the
// measurements that led here and what is left (0.74 of the HBM roofline at N = 148 against jk_s4's 0.55; the ablation
// switches those measurements were made with live in tools/variants/jk_m4_r03_ablations.hip, not in this file).
#include <cstdlib>

#include "jk_m4_layout.h"
#include "jk_m4_walk.h"
#include "nbx_common.h"

// (m0 is named as a clobber of the LDS-DMA asm below; clang calls that a reserved register)
#pragma clang diagnostic ignored "-Winline-asm"

namespace {

constexpr int M4_THREADS = 512, M4_CUS = 256;  // 4 consumer + 4 producer waves (M4_PROD_THREADS, M4_NCH: jk_m4_layout.h)
constexpr int M4_PER_CU = 1;
constexpr int M4_RING = 5;  // chunk buffers in LDS: one being walked, one landed (its J taken), three in flight
typedef __attribute__((address_space(3))) void* m4_lds_vp;

// at most three chunks' worth of this wave's chunk loads (LPT each) still in flight
template <int LPT>
__device__ __forceinline__ void m4_wait_three_chunks() {
    static_assert(3 * LPT <= 63, "vmcnt is a 6-bit counter");
    if constexpr (LPT == 6) asm volatile("s_waitcnt vmcnt(18)" ::: "memory");
    else if constexpr (LPT == 5) asm volatile("s_waitcnt vmcnt(15)" ::: "memory");
    else if constexpr (LPT == 4) asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
    else if constexpr (LPT == 3) asm volatile("s_waitcnt vmcnt(9)" ::: "memory");
    else if constexpr (LPT == 2) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

typedef double m4_d2 __attribute__((ext_vector_type(2)));
// Row-q partials that wait for the end of the range (see the kernel): tiles [M4_HOLD_FIRST, M4_HOLD_FIRST + M4_HOLD_TILES) of a
// range in the loading waves' registers -- as many as fit next to the Dtot' table without a spill: check private_seg_size
// after touching this --, the M4_HOLD_LDS tiles before them in what is left of LDS (2 KB each); the first few go out at once.
constexpr int M4_HOLD_TILES = 34, M4_HOLD_FIRST = 10;
constexpr int M4_HOLD_LDS = 5;

__device__ __forceinline__ double2 m4_ldnt(const double* p) {
    const m4_d2 t = __builtin_nontemporal_load(reinterpret_cast<const m4_d2*>(p));
    return make_double2(t.x, t.y);
}

__host__ __device__ __forceinline__ int m4_tri_row(int64_t T) {
    int64_t p = (int64_t)((sqrt(8.0 * (double)T + 1.0) - 1.0) * 0.5);
    while (p * (p + 1) / 2 > T) --p;
    while ((p + 1) * (p + 2) / 2 <= T) ++p;
    return (int)p;
}

// ---------------------------------------------------------------------------------------------- pack and weights
// slab rows [p0, p0 + np) of the dense tensor -> block-major tiles; one workgroup per tile
// (nsrc <= N: the dense tensor's own size -- a size that is not a multiple of four runs as the next one that is, with the
//  extra rows and columns zero: only the tiles p < nsrc exist, and their elements beyond nsrc are stored as zeros)
template <int NB>
__global__ __launch_bounds__(256) void m4_pack_kernel(const double* __restrict__ eri, double* __restrict__ out, int p0,
                                                      int64_t t_begin, int nsrc) {
    using G = M4Geom<NB>;
    const int64_t T = t_begin + blockIdx.x;
    const int p = m4_tri_row(T), q = (int)(T - (int64_t)p * (p + 1) / 2);
    const double* src = eri + ((int64_t)(p - p0) * nsrc + q) * nsrc * nsrc;
    double* dst = out + (int64_t)blockIdx.x * G::TILE;
    for (int e = threadIdx.x; e < G::TILE; e += 256) {
        const int blk = e >> 4;
        const int bt = m4_tri_row(blk), bc = blk - m4_tri(bt);
        const int k = ((e >> 2) & 3) ^ ((bt ^ bc) & 3), i = (e & 3) ^ k;  // (the swizzle: see the layout note above)
        const int row = 4 * bt + i, col = 4 * bc + k;
        dst[e] = (col <= row && row < nsrc) ? src[(int64_t)row * nsrc + col] : 0.0;
    }
}

// Dtot' in the staging order of the main kernel: wt[(k LPT + s) M4_THREADS + tid] = the weights of the two doubles that
// thread tid holds in slot s of chunk k: Dtot[r][c] + Dtot[c][r] below the diagonal, Dtot[r][r] on it, 0 elsewhere
// (the zeros of the diagonal blocks, the clamped tail loads of a chunk)
template <int NB>
__global__ __launch_bounds__(256) void m4_weights_kernel(const double* __restrict__ dm, int ndm, double* __restrict__ wt) {
    using G = M4Geom<NB>;
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= M4_NCH * G::LPT * M4_PROD_THREADS) return;
    const int tid = i % M4_PROD_THREADS, s = (i / M4_PROD_THREADS) % G::LPT, k = (i / M4_PROD_THREADS) / G::LPT;
    const int d0 = 16 * m4_tri(G::row0(k)) + (s * M4_PROD_THREADS + tid) * 2, dend = 16 * m4_tri(G::row0(k + 1));
    const int64_t n2 = (int64_t)G::N * G::N;
    double out[2] = {0.0, 0.0};
    for (int e = 0; e < 2; ++e) {
        const int d = d0 + e;
        if (d >= dend) continue;
        const int blk = d >> 4;
        const int bt = m4_tri_row(blk), bc = blk - m4_tri(bt);
        const int kk = ((d >> 2) & 3) ^ ((bt ^ bc) & 3), ii = (d & 3) ^ kk;
        const int row = 4 * bt + ii, col = 4 * bc + kk;
        if (col > row) continue;
        double v = 0.0, vt = 0.0;
        for (int x = 0; x < ndm; ++x) {
            v += dm[x * n2 + (int64_t)row * G::N + col];
            vt += dm[x * n2 + (int64_t)col * G::N + row];
        }
        out[e] = row == col ? v : v + vt;
    }
    *reinterpret_cast<double2*>(wt + 2 * (int64_t)i) = make_double2(out[0], out[1]);
}

// ---------------------------------------------------------------------------------------------- the kernel
// One workgroup per CU, eight waves in two ROLES:
//   waves 0-3  CONSUMERS: walk chunk g from LDS (one wave per SIMD: the walk has a SIMD's issue slots to itself) and hold
//              the partial K rows in registers;
//   waves 4-7  PRODUCERS: bring chunk g + 1 from the staging registers into the other LDS buffer, take its J contribution
//              (Dtot' table in registers), re-issue the loads -- a whole tile in flight, and nothing but these few
//              instructions between a chunk's arrival and the reload of its registers --, fetch the next tile's X, and do
//              the consumers' end-of-tile reductions (row-q partial of the tile, row-p partial when the row changes, J).
// One barrier per chunk step, executed by both roles.  (With every wave doing everything -- two four-wave workgroups per
// CU, or one of eight -- the phases of a tile add up: ~1900 instructions per wave and tile, and a ring of loads that cannot
// be re-issued before its consumer is done; measured 227-265 us at N = 148 against 152 us for the stream alone.)
//
// kpart1[(w S + slot) NDM + x][N]: row-p partial of workgroup w for the slot-th row of its range;
// kpart2[(T - t_begin) NDM + x][N]: row-q partial of tile T (q < p); jfull (N, N): J[p][q] = J[q][p] of the tiles visited
template <int NB, int NDM>
__global__ __launch_bounds__(M4_THREADS, 1) void jk_m4_kernel(const double* __restrict__ packed, const double* __restrict__ dm,
                                                              const double* __restrict__ wtab, double* __restrict__ jfull,
                                                              double* __restrict__ kpart1, double* __restrict__ kpart2,
                                                              int64_t t_begin, int64_t t_end, int L, int S) {
    using G_ = M4Geom<NB>;
    constexpr int N = G_::N, NG = G_::NG, LPT = G_::LPT, BUF = G_::BUF, TILE = G_::TILE, PT = M4_PROD_THREADS;
    extern __shared__ __attribute__((aligned(16))) double smem[];
    double* buf0 = smem;                     // [M4_RING][BUF] chunk buffers
    double* xs0 = smem + M4_RING * BUF;      // [2][N][4] X of the current / next tile
    double* redq = xs0 + 2 * 4 * N;          // [4][NG][32] consumers' row-q halves (odd columns) of the tile just walked
    double* redp = redq + 4 * NG * 32;       // [4][NG][32] consumers' row-p halves (even columns) when the row ends
    double* jred = redp + 4 * NG * 32;       // [2][4] producers' J partials per tile parity
    double* jstage = jred + 16;              // [L] J of the tiles done, stored at the end of the range
    double* hold_lds = jstage + L;           // [M4_HOLD_LDS][PT] row-q partials of tiles M4_HOLD_FIRST - M4_HOLD_LDS .. M4_HOLD_FIRST - 1
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const bool producer = wave >= 4;
    const int ptid = tid - 256;  // producers: 0 .. 255
    const int64_t n2 = (int64_t)N * N;
    (void)n2;

    const int64_t T0 = t_begin + (int64_t)blockIdx.x * L;
    const int64_t T_end = min(t_end, T0 + L);
    if (T0 >= T_end) return;  // uniform for the whole workgroup
    const int p_first = m4_tri_row(T0);
    const int ntile = (int)(T_end - T0);
    const int nstep = M4_NCH * ntile;
    const double* tile0 = packed + (T0 - t_begin) * (int64_t)TILE;

    // ------------------------------------------------------------------ producer state
    double2 wt[M4_NCH][LPT];   // the thread's Dtot' entries (the same for every tile)
    double jacc = 0.0;
    // Chunk g = 4 t + k goes from HBM straight into ring slot g % M4_RING: global_load_lds_dwordx4, lane l of a wave
    // lands its 16 bytes at the instruction's LDS base + 16 l -- no staging registers, no ds_write; slot s of the chunk
    // is one instruction per producer wave, in the layout the register path had (thread ptid's two doubles of slot s at
    // (s PT + ptid) 2).  Every chunk is LPT instructions per wave whatever its length (the tail re-reads the chunk's last
    // 16 bytes, a chunk past the last tile re-reads the first tile: landed in a free slot, never read): a wait for
    // `vmcnt(3 LPT)` means "my part of the chunk three behind the newest has landed".
    auto issue = [&](int g) {
        const int k = g & (M4_NCH - 1);
        const bool real = g < nstep;
        const double* tile = tile0 + (real ? (int64_t)(g >> 2) * TILE : 0);
        const int begin = 16 * m4_tri(G_::row0(k)), end = 16 * m4_tri(G_::row0(k + 1));
        double* buf = buf0 + (g % M4_RING) * BUF;
        int pt_ = ptid;  // (opaque per chunk: the 24 clamped offsets of a tile are recomputed, not kept in registers)
        asm volatile("" : "+v"(pt_));
#pragma unroll
        for (int s = 0; s < LPT; ++s) {
            int d = begin + (s * PT + pt_) * 2;
            d = min(d, end - 2);
            const unsigned off = 8u * (unsigned)d;
            const unsigned lds_a = (unsigned)(size_t)(m4_lds_vp)(buf + (s * PT + (wave - 4) * 64) * 2);
            asm volatile("s_mov_b32 m0, %2\n\tglobal_load_lds_dwordx4 %0, %1 nt" : : "v"(off), "s"(tile), "s"(lds_a) : "memory", "m0");
        }
    };
    // the J contribution of a landed chunk: the thread reads back the 16 bytes its own wave brought in
    auto jpass = [&](int g) {
        const int k = g & (M4_NCH - 1);
        const double* buf = buf0 + (g % M4_RING) * BUF;
        // (in two halves: the read-back of a whole chunk at once is 24 registers, and every register pair this wave can
        //  spare holds one more tile's row partial until the range is done)
#pragma unroll
        for (int s = 0; s < LPT; ++s) {
            if (s == (LPT + 1) / 2) __builtin_amdgcn_sched_barrier(0);
            const double2 v = *reinterpret_cast<const double2*>(buf + (s * PT + ptid) * 2);
            const double2 w = k == 0 ? wt[0][s] : (k == 1 ? wt[1][s] : (k == 2 ? wt[2][s] : wt[3][s]));
            jacc = fma(v.y, w.y, fma(v.x, w.x, jacc));
        }
    };
    // X of a tile: xs[n][c] = D^{c / 2}[c & 1 ? p : q][n]  (NDM = 1: columns 2, 3 are zero).  Fetched and stored by the
    // CONSUMER waves: a load the compiler tracks would make a producer wait for its whole queue of chunk loads.
    constexpr int XU = (4 * N + PT - 1) / PT;
    auto fetch_x = [&](int pp, int qq, double (&v)[XU]) {
#pragma unroll
        for (int u = 0; u < XU; ++u) {
            const int e = tid + PT * u;  // element e = 4 n + c
            const int n = e >> 2, cc = e & 3, x = cc >> 1;
            const int off = x * N * N + ((cc & 1) ? pp : qq) * N + n;  // (32-bit: scalar base + one offset register)
            v[u] = (e < 4 * N && x < NDM) ? dm[off] : 0.0;
        }
    };
    auto store_x = [&](double* xs, const double (&v)[XU]) {
#pragma unroll
        for (int u = 0; u < XU; ++u) {
            const int e = tid + PT * u;
            if (e < 4 * N) xs[e] = v[u];
        }
    };
    // the consumers' partial rows of a finished tile: summed over the four consumer waves, in wave order
    // `last`: the last column that is stored.  K is symmetric (symmetric densities: include/nbx.h), so of the row-q
    // partial of a tile only the columns <= q are needed -- jk_sym_reduce_kernel(k_lower) mirrors the sums -- which is
    // a third of the 26 MB these per-tile rows come to at N = 148.  It is their memory traffic that costs: with the
    // stores suppressed the kernel runs at the pace of its read stream (160 against 188 us on one box), with every tile's
    // row written to the same cache-resident place as well (162); eight places per workgroup 176, forty 186.
    auto reduce_rows = [&](const double* red, int parity, double* dst, int last) {  // dst[x N + row]
        for (int e = ptid; e < NG * 32; e += PT) {
            const int g = e >> 5, l = 2 * (e & 31) + parity;
            const int row = 16 * g + 4 * ((l >> 2) & 3) + (l >> 4), x = (l & 3) >> 1;
            if (row <= last && x < NDM) dst[x * N + row] = (red[e] + red[NG * 32 + e]) + (red[2 * NG * 32 + e] + red[3 * NG * 32 + e]);
        }
    };

    // Row-q partials held back until the range is done.  Thread ptid owns element e = ptid of a tile's row (row groups
    // 0..7: rows < 128; the few rows above are stored at once), one register pair per tile for M4_HOLD_TILES tiles of the
    // range -- as many as the registers of a loading wave take next to its Dtot' table without a spill (the first
    // M4_HOLD_FIRST tiles of a range go out as they are finished).
    // Why: bytes written to fresh lines WHILE the chip streams cost nine times what a byte read costs; written when the
    // range's stream is over they cost what a write costs (profiles/r03/lds_dma_stream_probe.txt: +12 us against +2 for
    // all of them; the bursts have to come at the END -- twenty tiles' rows together in mid-stream cost what they cost
    // one by one).  176 -> 170 us.
    double hold0 = 0.0, hold1 = 0.0, hold2 = 0.0, hold3 = 0.0, hold4 = 0.0, hold5 = 0.0, hold6 = 0.0, hold7 = 0.0, hold8 = 0.0, hold9 = 0.0, hold10 = 0.0, hold11 = 0.0, hold12 = 0.0, hold13 = 0.0, hold14 = 0.0, hold15 = 0.0, hold16 = 0.0, hold17 = 0.0, hold18 = 0.0, hold19 = 0.0, hold20 = 0.0, hold21 = 0.0, hold22 = 0.0, hold23 = 0.0, hold24 = 0.0, hold25 = 0.0, hold26 = 0.0, hold27 = 0.0, hold28 = 0.0, hold29 = 0.0, hold30 = 0.0, hold31 = 0.0, hold32 = 0.0, hold33 = 0.0;
    auto hold_put = [&](int t, double v) {  // t uniform, 0 .. M4_HOLD_TILES - 1.  (Selects, not a switch: a switch is turned
        // into an indexed array in scratch memory, and a scratch access waits behind the whole load queue)
        hold0 = t == 0 ? v : hold0;
        hold1 = t == 1 ? v : hold1;
        hold2 = t == 2 ? v : hold2;
        hold3 = t == 3 ? v : hold3;
        hold4 = t == 4 ? v : hold4;
        hold5 = t == 5 ? v : hold5;
        hold6 = t == 6 ? v : hold6;
        hold7 = t == 7 ? v : hold7;
        hold8 = t == 8 ? v : hold8;
        hold9 = t == 9 ? v : hold9;
        hold10 = t == 10 ? v : hold10;
        hold11 = t == 11 ? v : hold11;
        hold12 = t == 12 ? v : hold12;
        hold13 = t == 13 ? v : hold13;
        hold14 = t == 14 ? v : hold14;
        hold15 = t == 15 ? v : hold15;
        hold16 = t == 16 ? v : hold16;
        hold17 = t == 17 ? v : hold17;
        hold18 = t == 18 ? v : hold18;
        hold19 = t == 19 ? v : hold19;
        hold20 = t == 20 ? v : hold20;
        hold21 = t == 21 ? v : hold21;
        hold22 = t == 22 ? v : hold22;
        hold23 = t == 23 ? v : hold23;
        hold24 = t == 24 ? v : hold24;
        hold25 = t == 25 ? v : hold25;
        hold26 = t == 26 ? v : hold26;
        hold27 = t == 27 ? v : hold27;
        hold28 = t == 28 ? v : hold28;
        hold29 = t == 29 ? v : hold29;
        hold30 = t == 30 ? v : hold30;
        hold31 = t == 31 ? v : hold31;
        hold32 = t == 32 ? v : hold32;
        hold33 = t == 33 ? v : hold33;
    };
    // element e of a row-q partial -> (row, spin) as in reduce_rows (parity 1)
    auto rowq_of = [&](int e, int& row, int& x) {
        const int g = e >> 5, l = 2 * (e & 31) + 1;
        row = 16 * g + 4 * ((l >> 2) & 3) + (l >> 4);
        x = (l & 3) >> 1;
    };
    // tile tt's row-q partial: summed as in reduce_rows; rows < 128 into the registers, the others stored now
    auto hold_rows = [&](int tt, int qq_, double* dst) {  // tt < 0: LDS slot tt + M4_HOLD_LDS
        const double mine = (redq[ptid] + redq[NG * 32 + ptid]) + (redq[2 * NG * 32 + ptid] + redq[3 * NG * 32 + ptid]);
        if (tt < 0) hold_lds[(tt + M4_HOLD_LDS) * PT + ptid] = mine;
        else hold_put(tt, mine);
        const int e = PT + ptid;
        if (e < NG * 32) {
            int row, x;
            rowq_of(e, row, x);
            if (row <= qq_ && row < N && x < NDM)
                dst[x * N + row] = (redq[e] + redq[NG * 32 + e]) + (redq[2 * NG * 32 + e] + redq[3 * NG * 32 + e]);
        }
    };

    // ------------------------------------------------------------------ consumer state
    double acc[NG], bxr[NG];  // (bxr: the row part's X operands of the tile being walked, m4_walk_chunk)
#pragma unroll
    for (int g = 0; g < NG; ++g) acc[g] = bxr[g] = 0.0;
    M4Lane<NG> ln;
    ln.a = lane >> 4;
    ln.b = (lane >> 2) & 3;
    ln.c = lane & 3;
    {
        const int w4 = wave & 3;
#pragma unroll
        for (int g = 0; g < NG; ++g)
            ln.rowg[g] = 64 * g * ln.b + 16 * (m4_tri(ln.b) + w4) + 4 * (ln.a ^ ln.b ^ w4) + (ln.c ^ ln.a);
        ln.xlane = 4 * ln.a + ln.c;
        ln.xrow = ln.xlane + 16 * w4;
        ln.col0 = 16 * ln.b + (ln.a ^ ln.c);
        ln.cbx = ln.c ^ ln.b;
    }

    // ------------------------------------------------------------------ prologue: four chunks in flight, chunk 0 and X in LDS
    int p = p_first, q = (int)(T0 - (int64_t)p * (p + 1) / 2);
    if (producer) {
#pragma unroll
        for (int k = 0; k < M4_NCH; ++k)
#pragma unroll
            for (int s = 0; s < LPT; ++s) wt[k][s] = *reinterpret_cast<const double2*>(wtab + 2 * ((k * LPT + s) * PT + ptid));
        if (ptid < 8) jred[ptid] = 0.0;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // (the table: nothing of the compiler's in the counter from here)
        issue(0);
        issue(1);
        issue(2);
        issue(3);
        m4_wait_three_chunks<LPT>();  // my part of chunk 0
    } else {
        double v[XU];
        fetch_x(p, q, v);
        store_x(xs0, v);
    }
    __syncthreads();

    // ------------------------------------------------------------------ the steps: tile t, chunk k -> step g = 4 t + k
    // Two loops, one per role (so that the producers' registers -- Dtot' table -- and the consumers' -- accumulators,
    // walk operands -- are never live together); both execute exactly one barrier per step.
    auto next_pq = [](int& pp, int& qq) {
        if (++qq > pp) {
            ++pp;
            qq = 0;
        }
    };
    if (producer) {
        for (int t = 0; t < ntile; ++t) {
            const int64_t T = T0 + t;
            int pn = p, qn = q;
            next_pq(pn, qn);
#pragma unroll
            for (int k = 0; k < M4_NCH; ++k) {
                const int g = M4_NCH * t + k;
                // step g: the consumers walk chunk g; chunk g + 4 goes into the slot they left at the last barrier
                // (chunk g - 1's), chunk g + 1 has landed and gives its J contribution
                issue(g + 4);  // (first: the stream is what the kernel is bound by)
                if (k == 0 && t > 0) {
                    // the consumers' rows of tile t - 1 (written at its last step, behind that step's barrier), and its J
                    int pp = p, qq = q - 1;
                    if (qq < 0) {
                        pp = p - 1;
                        qq = pp;
                    }
                    if (qq < pp) {
                        if (t - 1 >= M4_HOLD_FIRST - M4_HOLD_LDS && t - 1 < M4_HOLD_FIRST + M4_HOLD_TILES)
                            hold_rows(t - 1 - M4_HOLD_FIRST, qq, kpart2 + ((T - 1 - t_begin) * NDM) * (int64_t)N);
                        else
                            reduce_rows(redq, 1, kpart2 + ((T - 1 - t_begin) * NDM) * (int64_t)N, qq);
                    }
                    if (pp != p) reduce_rows(redp, 0, kpart1 + (((int64_t)blockIdx.x * S + (pp - p_first)) * NDM) * N, pp);
                    if (ptid == 0) {  // (J of the tile: kept in LDS until the range is done -- two scattered stores less per tile)
                        const double* jr = jred + ((t - 1) & 1) * 4;
                        jstage[t - 1] = (jr[0] + jr[1]) + (jr[2] + jr[3]);
                    }
                }
                // the J contribution of chunk g -- the one the consumers are walking: it landed a step ago, so nothing
                // but the wait for chunk g + 1 stands between this wave and the barrier once that chunk is there
                jpass(g);
                if (k == 3) {  // this wave's share of tile t's J
                    jacc = nbx_wave_sum_dpp(jacc);  // (lane moves, not six ds_bpermute round trips: jk_m8.hip)
                    if (lane == 0) jred[(t & 1) * 4 + (wave - 4)] = jacc;
                    jacc = 0.0;
                }
                m4_wait_three_chunks<LPT>();  // my part of chunk g + 1 (chunks g + 2 .. g + 4 may be in flight)
                __syncthreads();
            }
            p = pn;
            q = qn;
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // (the padding chunks: nothing may land after the workgroup has gone)
        // the last tile's rows ((p, q) has moved one past it)
        int pp = p, qq = q - 1;
        if (qq < 0) {
            pp = p - 1;
            qq = pp;
        }
        if (qq < pp) reduce_rows(redq, 1, kpart2 + ((T_end - 1 - t_begin) * NDM) * (int64_t)N, qq);
        reduce_rows(redp, 0, kpart1 + (((int64_t)blockIdx.x * S + (pp - p_first)) * NDM) * N, pp);
        {  // the rows held back: tiles 0 .. ntile - 2 of the range, (p, q) walked from its start
            int pt = p_first, qt = (int)(T0 - (int64_t)pt * (pt + 1) / 2);
            int row, x;
            rowq_of(ptid, row, x);
            for (int i = 0; i < M4_HOLD_FIRST; ++i) {
                const int sl = i - (M4_HOLD_FIRST - M4_HOLD_LDS);  // (this thread's own LDS entries: no barrier needed)
                if (sl >= 0 && i + 1 < ntile && qt < pt && row <= qt && x < NDM)
                    kpart2[((T0 + i - t_begin) * NDM) * (int64_t)N + x * N + row] = hold_lds[sl * PT + ptid];
                next_pq(pt, qt);
            }
            double* dst = kpart2 + ((T0 + M4_HOLD_FIRST - t_begin) * NDM) * (int64_t)N + x * N + row;
#define M4_FLUSH_HELD(t_)                                                      \
    if (M4_HOLD_FIRST + (t_) + 1 < ntile) {                                    \
        if (qt < pt && row <= qt && x < NDM) dst[(int64_t)(t_) * NDM * N] = hold##t_; \
        next_pq(pt, qt);                                                       \
    }
            M4_FLUSH_HELD(0);
            M4_FLUSH_HELD(1);
            M4_FLUSH_HELD(2);
            M4_FLUSH_HELD(3);
            M4_FLUSH_HELD(4);
            M4_FLUSH_HELD(5);
            M4_FLUSH_HELD(6);
            M4_FLUSH_HELD(7);
            M4_FLUSH_HELD(8);
            M4_FLUSH_HELD(9);
            M4_FLUSH_HELD(10);
            M4_FLUSH_HELD(11);
            M4_FLUSH_HELD(12);
            M4_FLUSH_HELD(13);
            M4_FLUSH_HELD(14);
            M4_FLUSH_HELD(15);
            M4_FLUSH_HELD(16);
            M4_FLUSH_HELD(17);
            M4_FLUSH_HELD(18);
            M4_FLUSH_HELD(19);
            M4_FLUSH_HELD(20);
            M4_FLUSH_HELD(21);
            M4_FLUSH_HELD(22);
            M4_FLUSH_HELD(23);
            M4_FLUSH_HELD(24);
            M4_FLUSH_HELD(25);
            M4_FLUSH_HELD(26);
            M4_FLUSH_HELD(27);
            M4_FLUSH_HELD(28);
            M4_FLUSH_HELD(29);
            M4_FLUSH_HELD(30);
            M4_FLUSH_HELD(31);
            M4_FLUSH_HELD(32);
            M4_FLUSH_HELD(33);
#undef M4_FLUSH_HELD
        }
        if (wave == 4) {  // J of every tile of the range: lane i stores tile i's (one wave: its LDS operations are in order)
            if (lane == 0) {  // the last tile's J (its partial sums were stored before the last barrier)
                const double* jr = jred + ((ntile - 1) & 1) * 4;
                jstage[ntile - 1] = (jr[0] + jr[1]) + (jr[2] + jr[3]);
            }
            for (int i = lane; i < ntile; i += 64) {
                const int64_t Ti = T0 + i;
                const int pi = m4_tri_row(Ti), qi = (int)(Ti - (int64_t)pi * (pi + 1) / 2);
                const double j = jstage[i];
                jfull[(int64_t)pi * N + qi] = j;
                jfull[(int64_t)qi * N + pi] = j;
            }
        }
    } else {
        int slot = 0;  // ring slot of the chunk being walked
        for (int t = 0; t < ntile; ++t) {
            int pn = p, qn = q;
            next_pq(pn, qn);
            const bool more = t + 1 < ntile;
            const bool row_ends = pn != p || !more;
            const double* xs = xs0 + (t & 1) * 4 * N;
            double xv[XU];
            fetch_x(more ? pn : p, more ? qn : q, xv);  // X of tile t + 1: fetched now, stored two steps on
#pragma unroll
            for (int k = 0; k < M4_NCH; ++k) {
                const double* buf = buf0 + slot * BUF;
                slot = slot + 1 == M4_RING ? 0 : slot + 1;
                if (k == 0) m4_walk_chunk<G_, 0>(buf, xs, wave, ln, acc, bxr);
                else if (k == 1) m4_walk_chunk<G_, 1>(buf, xs, wave, ln, acc, bxr);
                else if (k == 2) m4_walk_chunk<G_, 2>(buf, xs, wave, ln, acc, bxr);
                else m4_walk_chunk<G_, 3>(buf, xs, wave, ln, acc, bxr);
                if (k == 1 && more) store_x(xs0 + ((t + 1) & 1) * 4 * N, xv);
                if (k == M4_NCH - 1) {
                    // end of tile: the row-q halves (odd columns: they used D[p][:]) leave the registers; the row-p halves
                    // (even columns) stay until the row changes.  The producers sum them up during the next step.
                    const bool odd = lane & 1;
#pragma unroll
                    for (int g = 0; g < NG; ++g) {
                        if (odd) redq[(wave * NG + g) * 32 + (lane >> 1)] = acc[g];
                        else if (row_ends) redp[(wave * NG + g) * 32 + (lane >> 1)] = acc[g];
                        acc[g] = (odd || row_ends) ? 0.0 : acc[g];
                    }
                }
                // (the MFMAs are register-only, so the scheduler is free to sink them below the barrier -- and does: every
                // wave then waits for ALL its LDS reads, meets the others, and the four run their MFMAs at the same time
                // with the LDS idle, then read at the same time with the matrix pipe idle.  Pinned here, a wave's MFMAs
                // run as its operands arrive, under the other waves' reads)
                __builtin_amdgcn_sched_barrier(0);
                __syncthreads();
            }
            p = pn;
            q = qn;
        }
    }
}

size_t m4_align256(size_t x) { return (x + 255) & ~(size_t)255; }

struct M4Plan {
    int wgs, L, S;
    size_t wt_off, k1_off, k2_off, total, lds_bytes;
};

template <int NB>
M4Plan m4_plan_nb(int64_t p0, int64_t np, int64_t ndm) {
    using G = M4Geom<NB>;
    M4Plan pl;
    const int64_t ntiles = m4_tri((int)(p0 + np)) - m4_tri((int)p0);
    const int64_t slots = M4_CUS * M4_PER_CU;
    int64_t L = nbx_cdiv(ntiles, slots);
    if (L < 1) L = 1;
    pl.L = (int)L;
    pl.wgs = (int)nbx_cdiv(ntiles, L);
    pl.S = (int)sqrt(2.0 * (double)L) + 3;
    pl.lds_bytes = (size_t)(M4_RING * G::BUF + 2 * 4 * G::N + 2 * 4 * G::NG * 32 + 16 + pl.L + M4_HOLD_LDS * M4_PROD_THREADS) * sizeof(double);
    size_t off = 0;
    pl.wt_off = off; off += m4_align256((size_t)(M4_NCH * G::LPT * M4_PROD_THREADS * 2) * sizeof(double));
    pl.k1_off = off; off += m4_align256((size_t)((int64_t)pl.wgs * pl.S * ndm * G::N) * sizeof(double));
    pl.k2_off = off; off += m4_align256((size_t)(ntiles * ndm * G::N) * sizeof(double));
    pl.total = off;
    return pl;
}

template <int NB>
int m4_run(nbx_ctx* ctx, int64_t p0, int64_t p1, const double* d_packed, const double* d_dm, int64_t ndm, double* d_jk,
           void* d_work, const double* d_hv, double* d_fock, double* d_vhf, const double* d_wt_in) {
    using G = M4Geom<NB>;
    const int64_t np = p1 - p0, N = G::N, n2 = N * N;
    const M4Plan pl = m4_plan_nb<NB>(p0, np, ndm);
    char* base = static_cast<char*>(d_work);
    double* wt = reinterpret_cast<double*>(base + pl.wt_off);
    double* k1 = reinterpret_cast<double*>(base + pl.k1_off);
    double* k2 = reinterpret_cast<double*>(base + pl.k2_off);
    if (np < N) {  // J entries this slab does not own must read as zero
        const int rc = nbx_memset(ctx, d_jk, 0, (size_t)n2 * sizeof(double));
        if (rc != NBX_OK) return rc;
    }
    if (d_wt_in != nullptr) {  // the caller's table (left by nbx_huz_cycle_scalars_dts for this density)
        wt = const_cast<double*>(d_wt_in);
    } else {
        hipLaunchKernelGGL(m4_weights_kernel<NB>, dim3((unsigned)nbx_cdiv(M4_NCH * G::LPT * M4_PROD_THREADS, 256)), dim3(256), 0,
                           ctx->stream, d_dm, (int)ndm, wt);
        NBX_LAUNCH_CHECK();
    }
    const int64_t t_begin = m4_tri((int)p0), t_end = m4_tri((int)p1);
    {
        nbx_prof_scope prof(ctx, NBX_PROF_JK_DENSE);
        static bool attr_set = false;
        if (!attr_set) {
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&jk_m4_kernel<NB, 1>),
                                      hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&jk_m4_kernel<NB, 2>),
                                      hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
            attr_set = true;
        }
        if (ndm == 2)
            hipLaunchKernelGGL((jk_m4_kernel<NB, 2>), dim3((unsigned)pl.wgs), dim3(M4_THREADS), pl.lds_bytes, ctx->stream,
                               d_packed, d_dm, wt, d_jk, k1, k2, t_begin, t_end, pl.L, pl.S);
        else
            hipLaunchKernelGGL((jk_m4_kernel<NB, 1>), dim3((unsigned)pl.wgs), dim3(M4_THREADS), pl.lds_bytes, ctx->stream,
                               d_packed, d_dm, wt, d_jk, k1, k2, t_begin, t_end, pl.L, pl.S);
    }
    NBX_LAUNCH_CHECK();
    return nbx_jk_sym_reduce(ctx, k1, k2, d_jk + n2, N, p0, np, ndm, t_begin, pl.L, pl.S, d_jk, d_hv, d_fock, d_vhf, 1, 1);
}

}  // namespace

// The sizes this kernel has an instance for (N = 4 NB; every loop bound of the walk is a compile-time constant).
// NBX_JK_M4=0 in the environment (read once per process) hands them back to jk_s4.hip.
#ifndef NBX_M4_SIZES
#define NBX_M4_SIZES(X) X(25) X(26) X(27) X(28) X(29) X(30) X(31) X(32) X(33) X(34) X(35) X(36) X(37)  // N = 100 .. 148
#endif
#define M4_DISPATCH(N_, EXPR)            \
    switch ((int)((N_) / 4)) {           \
        NBX_M4_SIZES(M4_CASE_##EXPR)     \
        default: break;                  \
    }
bool nbx_jk_m4_covers(int64_t N) {
    static const bool on = getenv("NBX_JK_M4") == nullptr || atoi(getenv("NBX_JK_M4")) != 0;
    if (!on || N % 4 != 0) return false;
#define M4_CASE_covers(NB_) case NB_: return true;
    M4_DISPATCH(N, covers)
#undef M4_CASE_covers
    return false;
}

size_t nbx_jk_m4_packed_bytes(int64_t N, int64_t p0, int64_t p1) {
    const int64_t ntiles = m4_tri((int)p1) - m4_tri((int)p0);
    // (+ slack: the staging of the last tile prefetches nothing beyond)
#define M4_CASE_bytes(NB_) case NB_: return (size_t)(ntiles * M4Geom<NB_>::TILE) * sizeof(double) + 256;
    M4_DISPATCH(N, bytes)
#undef M4_CASE_bytes
    return 0;
}

size_t nbx_jk_m4_worksize(int64_t N, int64_t p0, int64_t p1, int64_t ndm) {
#define M4_CASE_work(NB_) case NB_: return m4_plan_nb<NB_>(p0, p1 - p0, ndm).total;
    M4_DISPATCH(N, work)
#undef M4_CASE_work
    return 0;
}

int nbx_jk_m4_pack(nbx_ctx* ctx, int64_t N, int64_t nsrc, int64_t p0, int64_t p1, const double* d_eri, double* d_packed) {
    NBX_CHECK_ARG(nbx_jk_m4_covers(N) && d_eri && d_packed && nsrc <= N && nsrc > N - 4 && p1 <= nsrc);
    const int64_t ntiles = m4_tri((int)p1) - m4_tri((int)p0);
#define M4_CASE_pack(NB_)                                                                                                  \
    case NB_:                                                                                                              \
        hipLaunchKernelGGL(m4_pack_kernel<NB_>, dim3((unsigned)ntiles), dim3(256), 0, ctx->stream, d_eri, d_packed, (int)p0, \
                           (int64_t)m4_tri((int)p0), (int)nsrc);                                                           \
        break;
    M4_DISPATCH(N, pack)
#undef M4_CASE_pack
    NBX_LAUNCH_CHECK();
    return NBX_OK;
}

// the Dtot' weights table of a density (m4_weight_index order; the entries nothing writes are zero weights)
size_t nbx_jk_m4_weights_bytes(int64_t N) {
#define M4_CASE_wbytes(NB_) case NB_: return (size_t)(M4_NCH * M4Geom<NB_>::LPT * M4_PROD_THREADS * 2) * sizeof(double);
    M4_DISPATCH(N, wbytes)
#undef M4_CASE_wbytes
    return 0;
}

// what huz_scalars_kernel needs to write that table for size N: the first block rows of chunks 1..3 and the slots per chunk
void nbx_jk_m4_weight_layout(int64_t N, int out[4]) {
    out[0] = out[1] = out[2] = out[3] = 0;
#define M4_CASE_wl(NB_)                     \
    case NB_:                               \
        out[0] = M4Geom<NB_>::row0(1);      \
        out[1] = M4Geom<NB_>::row0(2);      \
        out[2] = M4Geom<NB_>::row0(3);      \
        out[3] = M4Geom<NB_>::LPT;          \
        break;
    M4_DISPATCH(N, wl)
#undef M4_CASE_wl
}

// d_wt: NULL, or that table for d_dm (written by huz_scalars_kernel when it judged d_dm): saves the preparation launch
int nbx_jk_m4(nbx_ctx* ctx, int64_t N, int64_t p0, int64_t p1, const double* d_packed, const double* d_dm, int64_t ndm,
              double* d_jk, void* d_work, const double* d_hv, double* d_fock, double* d_vhf, const double* d_wt) {
    NBX_CHECK_ARG(nbx_jk_m4_covers(N));
#define M4_CASE_run(NB_) case NB_: return m4_run<NB_>(ctx, p0, p1, d_packed, d_dm, ndm, d_jk, d_work, d_hv, d_fock, d_vhf, d_wt);
    M4_DISPATCH(N, run)
#undef M4_CASE_run
    return NBX_E_UNSUPPORTED;
}
