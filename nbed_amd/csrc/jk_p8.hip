// libnbx: J/K contraction on EIGHT-fold packed integrals (include/nbx.h "J/K contraction, packed form").
//
//   (pq|rs) = (qp|rs) = (pq|sr) = (rs|pq): of the tiles of jk_s4.hip (pair (p, q <= p), lower triangle of
//   the (r, s) matrix) only the entries with (r, s) <= (p, q) in pair order are stored and read -- what
//   PySCF keeps in `mf._eri` and libcvhf contracts for the reference's get_veff (huzinaga_scf.py:156).
//
// Every stored entry L = w (pq|rs), w = 1 (rs < pq) or 1/2 (rs = pq), feeds
//       J_pq += L D'_rs   (forward: a dot product per tile, as in jk_s4.hip)
//       J_rs += L D'_pq   (backward: an axpy per tile into accumulators that live as long as the workgroup)
//       Kh_p[r] += L D_q[s], Kh_p[s] += L D_q[r], Kh_q[r] += L D_p[s], Kh_q[s] += L D_p[r]
// and K = Kh + Kh^T for a symmetric density (the four images (rs|pq), (sr|pq), (rs|qp), (sr|qp) of an
// entry are the transposes of the four above; the weight 1/2 keeps rs = pq from counting twice).  The
// Kh part is exactly the walk of jk_s4.hip on a tile whose rows a > p are absent.
//
// Tile format.  A tile keeps the chunk structure, the slot order and the Dtot' table of jk_s4.hip;
// rows a <= p are a PREFIX of every chunk (jk_s4_layout.h orders the rectangles of a round by row
// block), so tile (p, .) stores, chunk by chunk, only the 1 KB staging slots that start inside that
// prefix, entries past it (and entries of row p past q) zero.  A slot the tile does not store is
// loaded from a line of zeros instead: the loads in flight per wave never vary (the vector-memory
// counter is in order, so a skipped load would have to be waited for conservatively), LDS beyond the
// prefix always holds zeros, and nothing in the walk needs a mask.  N = 148: 0.509 GB against 0.973 GB.
//
// Registers.  The forward table (Dtot' at the thread's staged positions) and the backward accumulators are
// 2 x 48 doubles per thread for a whole tile in a four-wave workgroup -- twice what jk_s4.hip keeps.  This
// kernel runs ONE workgroup of EIGHT waves per compute unit instead: a thread stages three slots per chunk
// (wave W the slots 8 k + W, so that the slots of a truncated chunk spread over all waves), holds table and
// accumulators of its twelve slots (96 VGPRs), and the two waves of a row block share a walk, half of the
// steps each (their partial sums are added by the reduction).  Ranges are cut at equal COST (tiles grow with
// p; a tile costs at least what streams past in the latency of its loads).
//
// The walk feeds the density values by lane broadcast, not by scalar loads: a wave keeps the 2 NDM density rows
// of the block it walks in registers, element 16 g + (lane & 15) in every row of 16 lanes, and
// `v_fmac_f64_dpp acc, xr, tv row_newbcast:j` multiplies the lane's tile element by element j of its row.
#include <cstdlib>
#include <type_traits>
#include <utility>
#include <vector>

#include "jk_s4_device.h"

namespace {

constexpr int P8_NB = 4;
constexpr int P8_CUS = 256;

// pairs (16 bytes) of chunk ch that belong to rows a <= p
__host__ __device__ __forceinline__ int p8_prefix_pairs(const S4Geom& g, int p, int ch) {
    const int b = p / g.s, pl = p - b * g.s;
    int n;
    if (ch == 0) {
        n = b * g.tri + (pl + 1) * (pl + 2) / 2;
    } else {
        // rectangles of round ch in slot order: row blocks (1,3), (2,3), (2,3)
        const int hi0 = ch == 1 ? 1 : 2, hi1 = 3;
        n = (hi0 < b ? g.s * g.ls : hi0 == b ? (pl + 1) * g.ls : 0) + (hi1 < b ? g.s * g.ls : hi1 == b ? (pl + 1) * g.ls : 0);
    }
    return (n + 1) >> 1;
}

// staging slots of chunk ch a tile of row p stores: those whose first pair lies inside the prefix
__host__ __device__ __forceinline__ int p8_nslots(const S4Geom& g, int lpt, int p, int ch) {
    const int pp = p8_prefix_pairs(g, p, ch), ne = ch == 0 ? g.E0 : g.Er;
    int n = 0;
    for (int j = 0; j < P8_NB * lpt; ++j) {
        const int ps = s4_slot_start(ne, lpt, j / lpt, j % lpt);
        if (ps >= 0 && ps < pp) ++n;
    }
    return n;
}

// ---- the walk of one LDS-resident chunk with the density values broadcast across lanes -------------------
// jk_s4.hip feeds the wave-uniform density values D_q[u s + c], D_p[u s + c] of step c through scalar loads,
// and that walk is bound by their latency: a group of four steps cannot start before the loads issued one
// group earlier are back (~250 clocks for ~64 clocks of FMAs).  Here a wave holds the 2 NDM density rows of
// the block it walks in registers, element 16 g + (lane & 15) of the block in every row of 16 lanes, and
//       v_fmac_f64_dpp acc, xr, tv row_newbcast:j
// multiplies the lane's tile element by lane j of its own row of xr (probed on gfx950: full rate, semantics
// as documented for gfx90a; scratch/probe/dpp64.hip).  A step is one LDS read and 2 NDM FMAs, nothing else.
template <int J>
__device__ __forceinline__ void p8_fmac_bcast(double& acc, double xr, double tv) {
    asm volatile("v_fmac_f64_dpp %0, %1, %2 row_newbcast:%3 row_mask:0xf bank_mask:0xf" : "+v"(acc) : "v"(xr), "v"(tv), "n"(J));
}

typedef __attribute__((address_space(3))) const char* p8_lds_cp;
typedef __attribute__((address_space(3))) const double* p8_lds_dp;

// KIND 0: diagonal triangle (row part L[il][c], c <= il, else column part L[c][il]); 1: rectangle, row
// side (consecutive doubles from a0); 2: rectangle, column side (stride ls8 bytes from a0)
// A wave walks steps c0 + C, C = 0, 1, ...: the bases arrive shifted to step c0 (row part a0, rectangle
// sides), the column part of the triangle needs tri(c0 + C) = tri(c0) + c0 C + tri(C): ls8 carries 8 c0 there.
template <int KIND, int C>
__device__ __forceinline__ double p8_read(int a0, int a1, int il, int ls8) {
    int addr;
    if (KIND == 0) addr = il >= C ? a0 + 8 * C : a1 + C * ls8 + 8 * (C * (C + 1) / 2);
    else if (KIND == 1) addr = a0 + 8 * C;
    else addr = a0 + C * ls8;
    return *(p8_lds_dp)(size_t)addr;
}

// batches of BS steps: the reads of batch B + 1 are issued, then the FMAs of batch B run -- pinned by hand
// (left alone the scheduler hoists every read of the walk to its top and spills)
constexpr int P8_BS = 4;

template <int KIND, int B, int... J>
__device__ __forceinline__ void p8_readb(std::integer_sequence<int, J...>, double (&t)[P8_BS], int a0, int a1, int il, int ls8) {
    ((t[J] = p8_read<KIND, P8_BS * B + J>(a0, a1, il, ls8)), ...);
}

template <int NDM, int B, int J>
__device__ __forceinline__ void p8_fma1(double tv, const double (&x)[2 * NDM], double (&kp)[NDM], double (&kq)[NDM]) {
#pragma unroll
    for (int v = 0; v < NDM; ++v) {
        p8_fmac_bcast<(P8_BS * B + J) & 15>(kp[v], x[v], tv);
        p8_fmac_bcast<(P8_BS * B + J) & 15>(kq[v], x[NDM + v], tv);
    }
}

template <int NDM, int B, int... J>
__device__ __forceinline__ void p8_fmab(std::integer_sequence<int, J...>, bool whole, int nsteps, const double (&t)[P8_BS],
                                        const double (&x)[2 * NDM], double (&kp)[NDM], double (&kq)[NDM]) {
    if (whole) {
        (p8_fma1<NDM, B, J>(t[J], x, kp, kq), ...);
    } else {  // the last, partial batch: uniform guards
        ((P8_BS * B + J < nsteps ? p8_fma1<NDM, B, J>(t[J], x, kp, kq) : (void)0), ...);
    }
}

template <int NDM, int KIND, int NG, int B>
__device__ __forceinline__ void p8_stage(double (&cur)[P8_BS], double (&nxt)[P8_BS], int a0, int a1, int il, int ls8,
                                         const double (&xr)[NG][2 * NDM], int nsteps, double (&kp)[NDM], double (&kq)[NDM]) {
    using seq = std::make_integer_sequence<int, P8_BS>;
    constexpr int NBAT = 16 * NG / P8_BS;
    if (B < NBAT && P8_BS * B < nsteps) {
        if (B + 1 < NBAT && P8_BS * (B + 1) < nsteps) p8_readb<KIND, (B + 1 < NBAT ? B + 1 : 0)>(seq{}, nxt, a0, a1, il, ls8);
        __builtin_amdgcn_sched_barrier(0);
        p8_fmab<NDM, B>(seq{}, P8_BS * B + P8_BS <= nsteps, nsteps, cur, xr[(P8_BS * B / 16 < NG ? P8_BS * B / 16 : 0)], kp, kq);
        __builtin_amdgcn_sched_barrier(0);
    }
}

template <int NDM, int KIND, int NG, int... B2>
__device__ __forceinline__ void p8_stages(std::integer_sequence<int, B2...>, double (&ta)[P8_BS], double (&tb)[P8_BS], int a0,
                                          int a1, int il, int ls8, const double (&xr)[NG][2 * NDM], int nsteps, double (&kp)[NDM],
                                          double (&kq)[NDM]) {
    ((p8_stage<NDM, KIND, NG, 2 * B2>(ta, tb, a0, a1, il, ls8, xr, nsteps, kp, kq),
      p8_stage<NDM, KIND, NG, 2 * B2 + 1>(tb, ta, a0, a1, il, ls8, xr, nsteps, kp, kq)), ...);
}

// xr[g][v]: density rows of the walked block, NG groups of 16 steps; nsteps <= 16 NG
template <int NDM, int KIND, int NG>
__device__ __forceinline__ void p8_walk(int a0, int a1, int il, int ls8, const double (&xr)[NG][2 * NDM], int nsteps,
                                        double (&kp)[NDM], double (&kq)[NDM]) {
    double ta[P8_BS], tb[P8_BS];
    p8_readb<KIND, 0>(std::make_integer_sequence<int, P8_BS>{}, ta, a0, a1, il, ls8);
    p8_stages<NDM, KIND, NG>(std::make_integer_sequence<int, 8 * NG / P8_BS>{}, ta, tb, a0, a1, il, ls8, xr, nsteps, kp, kq);
}

// dense (N, N, N, N) -> 8-fold tiles.  One workgroup per tile.
__global__ __launch_bounds__(256) void p8_pack_kernel(const double* __restrict__ eri, double* __restrict__ out, int N,
                                                      int lpt, const int64_t* __restrict__ rowoff) {
    const S4Geom g = s4_geom(N, P8_NB);
    const int64_t T = blockIdx.x;
    const int p = s4_tri_row(T), q = (int)(T - s4_tri(p));
    const double* src = eri + ((int64_t)p * N + q) * (int64_t)N * N;
    const int64_t tl = (rowoff[p + 1] - rowoff[p]) / (p + 1);
    double* dst = out + rowoff[p] + (int64_t)q * tl;
    int base = 0;
    for (int ch = 0; ch < P8_NB; ++ch) {
        const int ns = p8_nslots(g, lpt, p, ch), ne = ch == 0 ? g.E0 : g.Er;
        for (int i = threadIdx.x; i < ns * 128; i += blockDim.x) {
            const int j = i >> 7, e = i & 127;
            const int ps = s4_slot_start(ne, lpt, j / lpt, j % lpt);
            int a, b;
            double v = 0.0;
            if (s4_unflat(g, ch, 2 * ps + e, a, b) && (a < p || (a == p && b <= q))) {
                v = src[(int64_t)a * N + b];
                if (a == p && b == q) v *= 0.5;
            }
            dst[(int64_t)(base + j) * 128 + e] = v;
        }
        base += ns;
    }
}

// NDM densities.  ONE workgroup of EIGHT waves per compute unit: wave W stages the slots 8 k + W, k < 3, of a
// chunk (so that a truncated chunk's slots spread over all waves) and walks half of the steps of block W / 2.
template <int NDM>
__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(2, 2))) void jk_p8_kernel(
    const double* __restrict__ eri, const double* __restrict__ dm, const double* __restrict__ dts,
    const double* __restrict__ zeros, const int64_t* __restrict__ rowoff, const int* __restrict__ rowtab,
    const int* __restrict__ wg_t0, double* __restrict__ jf, double* __restrict__ jb, double* __restrict__ kpart1,
    double* __restrict__ kpart2, int N, int S, int dbg) {
    constexpr int NB = P8_NB, NCH = NB, LPT = 3, LPT4 = 6, BUFD = LPT4 * NB * 128, NT = 512;
    constexpr int XV = 2 * NDM, XR = (XV * 156 + NT - 1) / NT, NG = 2;  // density rows per tile; loads per thread; 16-step groups
    extern __shared__ __attribute__((aligned(16))) double smem[];  // buf[2][BUFD] | slack[128] | jred[2][8] | xtab[2][XV][N]
    double* slack = smem + 2 * BUFD;
    double* jred = slack + 128;
    // the density rows of a tile, (D^x_q, x < NDM; D^x_p): staged by all threads at the start of the tile from
    // values fetched one tile earlier, read by every wave at the start of a chunk (two copies: tile parity)
    double* xtab = jred + 16;

    int64_t T = wg_t0[blockIdx.x];
    const int64_t T_end = wg_t0[blockIdx.x + 1];
    if (T >= T_end) {  // an empty range (uniform for the whole workgroup): its backward accumulators are zero
        for (int i = threadIdx.x; i < NCH * BUFD; i += NT) jb[(int64_t)blockIdx.x * (NCH * BUFD) + i] = 0.0;
        return;
    }
    int p = s4_tri_row(T);
    int q = (int)(T - s4_tri(p));
    const int p_first = p;

    const S4Geom g = s4_geom(N, NB);
    const int s = g.s, ls = g.ls;
    const int tid = threadIdx.x;
    const int W = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int w = W >> 1, hs = W & 1;  // block of rows; which half of a walk's steps
    const int lane = tid & 63;
    const bool live = lane < s;
    const int il = live ? lane : s - 1;  // idle lanes shadow the last row (their results are dropped)
    const int trow = w * s + il;
    const int64_t n2 = (int64_t)N * N;

    double2 stage[NCH][LPT];
    double2 dt[NCH][LPT];
    double2 jbacc[NCH][LPT];
    double xn[XR];
    // rowtab[p] = {slots stored of chunks 0..3, first slot of chunks 1..3 inside a tile, slots per tile}: kept
    // in scalar registers for the row of the NEXT tile (a scalar load and its wait per chunk cost more than
    // the chunk's arithmetic); reloaded when the row changes
    struct Row {
        int ns[4], c0[4], slots;
    };
    auto row_load = [&](int pp) {
        Row r;
        const int* t = rowtab + pp * 8;
        r.ns[0] = t[0]; r.ns[1] = t[1]; r.ns[2] = t[2]; r.ns[3] = t[3];
        r.c0[0] = 0; r.c0[1] = t[4]; r.c0[2] = t[5]; r.c0[3] = t[6];
        r.slots = t[7];
        return r;
    };
    auto issue = [&](double2(&st)[LPT], const double* tp, const Row& rw, int ch) {
        const int ns = rw.ns[ch];
        const double* cp = tp + (int64_t)rw.c0[ch] * 128;
#pragma unroll
        for (int k = 0; k < LPT; ++k) {
            // a slot the tile does not store still issues its load (the counts in flight stay static) from a line
            // of zeros (one line per slot; re-reading the head of the same tile instead measured 12 % slower)
            const int j = 8 * k + W;
            st[k] = s4_ldnt((j < ns ? cp : zeros) + 128 * j + 2 * lane);
        }
    };
    auto xfetch = [&](int pp, int qq) {
#pragma unroll
        for (int r = 0; r < XR; ++r) {
            const int e = min(tid + NT * r, XV * N - 1);
            const int v = e / N, c = e - v * N;
            xn[r] = dm[(int64_t)(v < NDM ? v : v - NDM) * n2 + (int64_t)(v < NDM ? qq : pp) * N + c];
        }
    };
    auto xstore = [&](double* dst) {
#pragma unroll
        for (int r = 0; r < XR; ++r) {
            const int e = tid + NT * r;
            if (e < XV * N) dst[e] = xn[r];
        }
    };
    Row rnext = row_load(p);  // row of the tile whose loads are issued next
    const double* tile = eri + rowoff[p] + (int64_t)q * ((int64_t)rnext.slots * 128);
#pragma unroll
    for (int ch = 0; ch < NCH; ++ch) {
        issue(stage[ch], tile, rnext, ch);
#pragma unroll
        for (int k = 0; k < LPT; ++k) {
            dt[ch][k] = *reinterpret_cast<const double2*>(dts + (ch * (NB * LPT4) + 8 * k + W) * 128 + 2 * lane);
            jbacc[ch][k] = make_double2(0.0, 0.0);
        }
    }
    xfetch(p, q);
    // the tables have arrived before the tile loop is entered: a load still pending at the loop header is
    // waited for INSIDE the loop (the wait must hold for the first iteration), and in order means that
    // wait drains the prefetch of every tile
#pragma unroll
    for (int ch = 0; ch < NCH; ++ch)
#pragma unroll
        for (int k = 0; k < LPT; ++k) asm volatile("" : "+v"(dt[ch][k].x), "+v"(dt[ch][k].y));
#pragma unroll
    for (int r = 0; r < XR; ++r) asm volatile("" : "+v"(xn[r]));
    const int lds0 = (int)(size_t)(p8_lds_cp)reinterpret_cast<const char*>(smem);  // LDS byte address of smem
    double kp[NDM];
#pragma unroll
    for (int x = 0; x < NDM; ++x) kp[x] = 0.0;
    auto flush_p = [&](int prow) {
        if (live) {
            double* kout = kpart1 + (((int64_t)blockIdx.x * S + (prow - p_first)) * 2 + hs) * NDM * N;
#pragma unroll
            for (int x = 0; x < NDM; ++x) kout[x * N + trow] = kp[x];
        }
#pragma unroll
        for (int x = 0; x < NDM; ++x) kp[x] = 0.0;
    };
    auto store_j = [&](int par, int pj, int qj) {  // thread 0, after a barrier that follows the jred writes
        double tot = 0.0;
#pragma unroll
        for (int v = 0; v < 8; ++v) tot += jred[par * 8 + v];
        jf[(int64_t)pj * N + qj] = tot;
    };

    int p_cur = p_first, par = 0;
    int pj = -1, qj = -1;  // the tile whose J partials sit in jred[par ^ 1]
    while (T < T_end) {
        if (p != p_cur) {
            flush_p(p_cur);
            p_cur = p;
        }
        // the next tile of the range (the last one re-reads itself: no tail case)
        int pn = p, qn = q + 1;
        const double* tile_next = tile + (int64_t)rnext.slots * 128;  // (rnext still describes row p here)
        if (T + 1 >= T_end) {
            tile_next = tile;
            qn = q;
        } else if (qn > pn) {
            ++pn;
            qn = 0;
            tile_next = eri + rowoff[pn];
            rnext = row_load(pn);
        }
        // D'_pq of the backward sum: (sum_x D_pq) + (sum_x D_qp), the association of the Dtot' table.
        // Scalar loads written out: as C++ they become vector loads, and a vector load issued behind
        // the streaming loads of the next tile drains the whole prefetch when its value is used.
        // (Not fetched a tile ahead: the compiler does not know an asm load is still in flight and may
        // copy or spill its destination before the data lands.)
        double dpq;
        {
            double dv[NDM], dvt[NDM];
            const int opq = (p * N + q) * 8, oqp = (q * N + p) * 8;
#pragma unroll
            for (int x = 0; x < NDM; ++x) {
                const double* dx = dm + x * n2;
                asm volatile("s_load_dwordx2 %0, %1, %2" : "=s"(dv[x]) : "s"(dx), "s"(opq));
                asm volatile("s_load_dwordx2 %0, %1, %2" : "=s"(dvt[x]) : "s"(dx), "s"(oqp));
            }
            __builtin_amdgcn_s_waitcnt(0xC07F);  // lgkmcnt(0): the compiler does not track these loads
            __builtin_amdgcn_sched_barrier(0);
            double v = 0.0, vt = 0.0;
#pragma unroll
            for (int x = 0; x < NDM; ++x) {
                v += dv[x];
                vt += dvt[x];
            }
            dpq = p == q ? v : v + vt;
        }
        double* xt = xtab + (int)(T & 1) * (XV * N);
        xstore(xt);  // ordered before the first walk by the barrier of the first chunk
        double kq[NDM];
#pragma unroll
        for (int x = 0; x < NDM; ++x) kq[x] = 0.0;
        const int bp = p / s, plp = p - bp * s;  // last row block of this tile and its last row
        // the lane's LDS addresses are the same for every tile: unless the lane is opaque here, every
        // address of every step is computed once, before the tile loop, and kept (hundreds of registers)
        int ilv = il, lanev = lane;
        asm volatile("" : "+v"(ilv));
        asm volatile("" : "+v"(lanev));  // (likewise the LDS addresses of the staging stores)
        // density rows of block u from step c0 on: element c0 + 16 g + (lane & 15) in every row of 16 lanes
        auto xload = [&](double(&xr)[NG][XV], int u, int c0) {
#pragma unroll
            for (int gq = 0; gq < NG; ++gq)
#pragma unroll
                for (int v = 0; v < XV; ++v) xr[gq][v] = xt[v * N + min(u * s + c0 + 16 * gq + (lanev & 15), N - 1)];
        };
        double jacc = 0.0;
#pragma unroll
        for (int ch = 0; ch < NCH; ++ch) {
            const int ne = ch == 0 ? g.E0 : g.Er;
            double* buf = smem + (ch & 1) * BUFD;
            double2(&st)[LPT] = stage[ch];
#pragma unroll
            for (int k = 0; k < LPT; ++k) {
                jacc = fma(st[k].x, dt[ch][k].x, fma(st[k].y, dt[ch][k].y, jacc));
                jbacc[ch][k].x = fma(st[k].x, dpq, jbacc[ch][k].x);
                jbacc[ch][k].y = fma(st[k].y, dpq, jbacc[ch][k].y);
                // done HERE: left to itself the compiler sinks these past the walk, the staged values stay
                // live, and the refill below needs another set of registers (and a wait for it every tile)
                asm volatile("" : "+v"(jbacc[ch][k].x), "+v"(jbacc[ch][k].y), "+v"(jacc));
                const int j = 8 * k + W;
                const int ps = s4_slot_start(ne, LPT4, j / LPT4, j % LPT4);
                *reinterpret_cast<double2*>((ps < 0 ? slack : buf + 2 * ps) + 2 * lanev) = st[k];
            }
            // refill the staging registers with the same chunk of the next tile (pinned after the
            // stores above: hoisted loads would need a second set of registers)
            __builtin_amdgcn_sched_barrier(0);
            issue(st, tile_next, rnext, ch);
            if (ch == 0) xfetch(pn, qn);  // (behind chunk 0's refill: the next tile needs both first)
            __builtin_amdgcn_sched_barrier(0);
            __syncthreads();
            if (ch == 0 && pj >= 0 && tid == 0) store_j(par ^ 1, pj, qj);

            // ---- the walk: rows a <= p only (a block past the last row block holds nothing; the last one
            // ends at row p: its row side needs no mask -- LDS holds zeros there --, its column side stops);
            // the two waves of a block take half of the steps each
            const int bufa = lds0 + (ch & 1) * BUFD * 8;
            if (dbg & 1) {
            } else if (ch == 0) {
                if (w <= bp) {
                    const int nst = w == bp ? plp + 1 : s, c0 = hs ? (nst + 1) >> 1 : 0, nh = hs ? nst - c0 : (nst + 1) >> 1;
                    double xr[NG][XV];
                    xload(xr, w, c0);
                    const int t0a = bufa + w * g.tri * 8;
                    p8_walk<NDM, 0, NG>(t0a + (ilv * (ilv + 1) / 2 + c0) * 8, t0a + (c0 * (c0 + 1) / 2 + ilv) * 8, ilv - c0,
                                        8 * c0, xr, nh, kp, kq);
                }
            } else {
                const int u = w ^ ch;
                if (max(w, u) <= bp) {
                    const int nst = (w > u || u != bp) ? s : plp + 1;
                    const int c0 = hs ? (nst + 1) >> 1 : 0, nh = hs ? nst - c0 : (nst + 1) >> 1;
                    double xr[NG][XV];
                    xload(xr, u, c0);
                    const int ra = bufa + s4_slot(min(w, u), ch) * s * ls * 8;
                    if (w > u) p8_walk<NDM, 1, NG>(ra + (ilv * ls + c0) * 8, 0, ilv, 0, xr, nh, kp, kq);
                    else p8_walk<NDM, 2, NG>(ra + (ilv + c0 * ls) * 8, 0, ilv, ls * 8, xr, nh, kp, kq);
                }
            }
        }
        // J partial of this tile (summed by thread 0 after the next barrier)
        jacc = nbx_wave_sum(jacc);
        if (lane == 0) jred[par * 8 + W] = jacc;
        pj = p;
        qj = q;
        par ^= 1;
        if (q < p && live && w <= bp) {
            double* k2 = kpart2 + (((T * 2 + hs) * NDM) * N) + trow;  // tile order: sequential stores
#pragma unroll
            for (int x = 0; x < NDM; ++x) k2[x * N] = kq[x];
        }
        ++T;
        tile = tile_next;
        p = pn;
        q = qn;
    }
    flush_p(p_cur);
    __syncthreads();
    if (tid == 0) store_j(par ^ 1, pj, qj);
    // backward J accumulators of this workgroup: [ch][slot][lane] pairs, the order of the Dtot' table
    double* jo = jb + (int64_t)blockIdx.x * (NCH * BUFD);
#pragma unroll
    for (int ch = 0; ch < NCH; ++ch)
#pragma unroll
        for (int k = 0; k < LPT; ++k)
            *reinterpret_cast<double2*>(jo + (int64_t)ch * BUFD + (8 * k + W) * 128 + 2 * lane) = jbacc[ch][k];
}

// Kh[x][row][b] = the row's row-p partials (kpart1, every workgroup whose range meets the row, both step
// halves) + the row-q partials of the tiles (p > row, row) (kpart2, both halves); blocks past a tile's last
// row block were not written and are skipped here too.  grid (N, NDM, ceil(N / 64)) x 256 threads.
__global__ __launch_bounds__(256) void p8_kh_kernel(const double* __restrict__ kpart1, const double* __restrict__ kpart2,
                                                    const int* __restrict__ wg_t0, int nrng, double* __restrict__ kh,
                                                    int N, int ndm, int S, int s) {
    __shared__ double part[4][64];
    const int row = blockIdx.x, x = blockIdx.y;
    const int lane = threadIdx.x & 63, chunk = threadIdx.x >> 6;
    const int b = blockIdx.z * 64 + lane;
    double t = 0.0;
    if (b < N) {
        const int wb = b / s;  // the waves that own column b write it only for tiles with p / s >= wb
        const int64_t stride = (int64_t)ndm * N;
        const double* src = kpart2 + (int64_t)x * N + b;
        int p = max(row + 1, wb * s) + chunk;
        for (; p + 12 < N; p += 16) {  // eight independent loads in flight
            double v[8];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int64_t T = (int64_t)(p + 4 * u) * (p + 4 * u + 1) / 2 + row;
                v[2 * u] = src[(T * 2) * stride];
                v[2 * u + 1] = src[(T * 2 + 1) * stride];
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) t += v[u];
        }
        for (; p < N; p += 4) {
            const int64_t T = (int64_t)p * (p + 1) / 2 + row;
            t += src[(T * 2) * stride] + src[(T * 2 + 1) * stride];
        }
    }
    part[chunk][lane] = t;
    __syncthreads();
    if (chunk == 0 && b < N) {
        double tot = part[0][lane] + part[1][lane] + part[2][lane] + part[3][lane];
        const int64_t t_lo = (int64_t)row * (row + 1) / 2, t_hi = t_lo + row;
        // the ranges are sorted: first range that ends past t_lo by bisection, then forward while they start <= t_hi
        int lo = 0, hi = nrng;
        while (lo < hi) {
            const int mid = (lo + hi) >> 1;
            if (wg_t0[mid + 1] <= t_lo) lo = mid + 1;
            else hi = mid;
        }
        for (int r = lo; r < nrng && wg_t0[r] <= t_hi; ++r) {
            const int64_t a = wg_t0[r], e = wg_t0[r + 1];
            if (a >= e) continue;
            const int slot = row - s4_tri_row(a);
            const double* k1 = kpart1 + ((((int64_t)r * S + slot) * 2) * ndm + x) * N + b;
            tot += k1[0] + k1[(int64_t)ndm * N];
        }
        kh[((int64_t)x * N + row) * N + b] = tot;
    }
}

// backward J: jbl[a][b], b <= a, = sum over the workgroups of their accumulator at the entry's staging
// position.  One wave per 64 entries x a quarter of the workgroups, fixed order.
__global__ __launch_bounds__(256) void p8_jb_kernel(const double* __restrict__ jb, int wgs, double* __restrict__ jbl, int N,
                                                    int lpt) {
    __shared__ double part[4][64];
    const int lane = threadIdx.x & 63, chunk = threadIdx.x >> 6;
    const int i = blockIdx.x * 64 + lane;
    const int a = i / N, b = i - a * N;
    double t = 0.0;
    if (i < N * N && b <= a) {
        const S4Geom g = s4_geom(N, P8_NB);
        const int64_t tile_doubles = (int64_t)P8_NB * lpt * P8_NB * 128;
        const double* src = jb + s4_dts_index(g, lpt, a, b);
        for (int wg = chunk; wg < wgs; wg += 4) t += src[(int64_t)wg * tile_doubles];
    }
    part[chunk][lane] = t;
    __syncthreads();
    if (chunk == 0 && i < N * N && b <= a) jbl[i] = (part[0][lane] + part[1][lane]) + (part[2][lane] + part[3][lane]);
}

// J = forward + backward, K = Kh + Kh^T, and the Fock epilogue of nbx_jk_packed_fock.
__global__ __launch_bounds__(256) void p8_finish_kernel(const double* __restrict__ jf, const double* __restrict__ jbl,
                                                        const double* __restrict__ kh, double* __restrict__ jk, int N,
                                                        int s, int ndm, const double* __restrict__ hv, double* __restrict__ fock,
                                                        double* __restrict__ vhf) {
    const int64_t n2 = (int64_t)N * N;
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n2) return;
    const int a = (int)(i / N), b = (int)(i - (int64_t)a * N);
    const int hi = max(a, b), lo = min(a, b);
    const int64_t o = (int64_t)hi * N + lo, ot = (int64_t)b * N + a;
    const double j = jf[o] + jbl[o];
    jk[i] = j;
    for (int x = 0; x < ndm; ++x) {
        const double k = kh[x * n2 + i] + kh[x * n2 + ot];
        jk[(1 + x) * n2 + i] = k;
        if (fock != nullptr) {
            const double v = j - k;
            fock[x * n2 + i] = hv[x * n2 + i] + v;
            if (vhf != nullptr) vhf[x * n2 + i] = v;
        }
    }
}

struct P8Plan {
    int lpt, wgs, nrng, S;
    size_t lds_bytes;
    // tail of the packed buffer: row offsets (N + 1 int64), range boundaries (nrng + 1 int), a line of zeros
    size_t tab_rowoff, tab_rowtab, tab_t0, tab_zeros, packed_total;
    size_t dts_off, jf_off, jb_off, jbl_off, kh_off, k1_off, k2_off, total;  // workspace
    std::vector<int64_t> rowoff;  // N + 1
    std::vector<int> t0;          // nrng + 1
    std::vector<int> rowtab;      // N x 8
    int64_t doubles;              // packed entries
};

size_t p8_align256(size_t x) { return (x + 255) & ~(size_t)255; }

void p8_rowoff(int64_t N, int lpt, std::vector<int64_t>& rowoff) {
    const S4Geom g = s4_geom((int)N, P8_NB);
    rowoff.assign(N + 1, 0);
    for (int p = 0; p < N; ++p) {
        int slots = 0;
        for (int ch = 0; ch < P8_NB; ++ch) slots += p8_nslots(g, lpt, p, ch);
        rowoff[p + 1] = rowoff[p] + (int64_t)(p + 1) * slots * 128;
    }
}

const P8Plan& p8_plan(int64_t N, int64_t ndm) {
    static P8Plan cache[2];
    static int64_t cache_n[2] = {0, 0};
    P8Plan& pl = cache[ndm - 1];
    if (cache_n[ndm - 1] == N) return pl;
    pl.lpt = s4_lpt(N);
    p8_rowoff(N, pl.lpt, pl.rowoff);
    pl.doubles = pl.rowoff[N];
    {
        const S4Geom g = s4_geom((int)N, P8_NB);
        pl.rowtab.assign((size_t)N * 8, 0);
        for (int p = 0; p < N; ++p) {
            int base = 0;
            for (int ch = 0; ch < P8_NB; ++ch) {
                const int n = p8_nslots(g, pl.lpt, p, ch);
                pl.rowtab[p * 8 + ch] = n;
                if (ch > 0) pl.rowtab[p * 8 + 3 + ch] = base;
                base += n;
            }
            pl.rowtab[p * 8 + 7] = base;
        }
    }
    pl.lds_bytes = (size_t)(2 * pl.lpt * P8_NB * 128 + 128 + 16 + 2 * 2 * ndm * N) * sizeof(double);
    pl.nrng = P8_CUS;  // one eight-wave workgroup (one range) per compute unit
    pl.wgs = pl.nrng;
    // ranges of equal COST, cut at tile boundaries: a tile costs its bytes, but not less than what streams
    // past in the latency of its loads (the prefetch is one tile deep: small tiles are latency bound)
    const int64_t ntiles = s4_tri(N);
    pl.t0.assign(pl.nrng + 1, 0);
    {
        static const double floor_bytes = [] {
            const char* e = getenv("NBX_P8_TILE_FLOOR");
            return e ? atof(e) : 49152.0;
        }();
        std::vector<double> cost(N);
        double total = 0.0;
        for (int p = 0; p < N; ++p) {
            const double bytes = 1024.0 * pl.rowtab[p * 8 + 7];
            cost[p] = bytes > floor_bytes ? bytes : floor_bytes;
            total += cost[p] * (p + 1);
        }
        int r = 1;
        int64_t T = 0;
        double acc = 0.0;
        for (int p = 0; p < N; ++p)
            for (int q = 0; q <= p; ++q, ++T) {
                acc += cost[p];
                while (r < pl.nrng && acc >= total * r / pl.nrng) pl.t0[r++] = (int)(T + 1);
            }
        for (; r <= pl.nrng; ++r) pl.t0[r] = (int)ntiles;
    }
    pl.S = 1;
    for (int r = 0; r < pl.nrng; ++r)
        if (pl.t0[r + 1] > pl.t0[r]) {
            const int span = s4_tri_row(pl.t0[r + 1] - 1) - s4_tri_row(pl.t0[r]) + 1;
            if (span > pl.S) pl.S = span;
        }
    size_t off = p8_align256((size_t)pl.doubles * sizeof(double));
    pl.tab_rowoff = off; off += p8_align256((size_t)(N + 1) * sizeof(int64_t));
    pl.tab_rowtab = off; off += p8_align256((size_t)N * 8 * sizeof(int));
    pl.tab_t0 = off; off += p8_align256((size_t)(pl.nrng + 1) * sizeof(int));
    pl.tab_zeros = off; off += (size_t)P8_NB * pl.lpt * 1024;
    pl.packed_total = off;
    off = 0;
    const size_t n2 = (size_t)(N * N) * sizeof(double);
    pl.dts_off = off; off += p8_align256((size_t)(P8_NB * P8_NB * pl.lpt * 128) * sizeof(double));
    pl.jf_off = off; off += p8_align256(n2);
    pl.jb_off = off; off += p8_align256((size_t)pl.wgs * P8_NB * pl.lpt * P8_NB * 128 * sizeof(double));
    pl.jbl_off = off; off += p8_align256(n2);
    pl.kh_off = off; off += p8_align256((size_t)ndm * n2);
    pl.k1_off = off; off += p8_align256((size_t)((int64_t)pl.wgs * pl.S * 2 * ndm * N) * sizeof(double));
    pl.k2_off = off; off += p8_align256((size_t)(ntiles * 2 * ndm * N) * sizeof(double));
    pl.total = off;
    cache_n[ndm - 1] = N;
    return pl;
}

}  // namespace

// EXPERIMENTAL: parity-tested, measured slower than jk_s4.hip (N = 148: 0.27 ms for the main kernel against
// 0.215) -- opt in with NBX_JK_P8=1.  Serves whole tensors of the four-block, six-loads
// sizes (N = 100 .. 156, N % 4 == 0).  (Packing and contraction must agree on the format: read once.)
bool nbx_jk_p8_covers(int64_t N, int64_t p0, int64_t p1) {
    static const bool on = [] {
        const char* e = getenv("NBX_JK_P8");
        return e != nullptr && e[0] == '1';
    }();
    return on && p0 == 0 && p1 == N && s4_supported(N) && s4_nb(N) == 4 && s4_lpt(N) == 6;
}

size_t nbx_jk_p8_packed_bytes(int64_t N) { return p8_plan(N, 1).packed_total; }
size_t nbx_jk_p8_worksize(int64_t N, int64_t ndm) { return p8_plan(N, ndm).total; }

int nbx_jk_p8_pack(nbx_ctx* ctx, int64_t N, const double* d_eri, double* d_packed) {
    const P8Plan& pl = p8_plan(N, 1);
    char* base = reinterpret_cast<char*>(d_packed);
    // the kernel's small tables live behind the tiles: written once, here
    NBX_HIP(hipMemcpyAsync(base + pl.tab_rowoff, pl.rowoff.data(), (size_t)(N + 1) * sizeof(int64_t),
                           hipMemcpyHostToDevice, ctx->stream));
    NBX_HIP(hipMemcpyAsync(base + pl.tab_rowtab, pl.rowtab.data(), (size_t)N * 8 * sizeof(int), hipMemcpyHostToDevice,
                           ctx->stream));
    NBX_HIP(hipMemcpyAsync(base + pl.tab_t0, pl.t0.data(), (size_t)(pl.nrng + 1) * sizeof(int), hipMemcpyHostToDevice,
                           ctx->stream));
    NBX_HIP(hipMemsetAsync(base + pl.tab_zeros, 0, (size_t)P8_NB * pl.lpt * 1024, ctx->stream));
    hipLaunchKernelGGL(p8_pack_kernel, dim3((unsigned)s4_tri(N)), dim3(256), 0, ctx->stream, d_eri, d_packed, (int)N,
                       pl.lpt, reinterpret_cast<const int64_t*>(base + pl.tab_rowoff));
    NBX_LAUNCH_CHECK();
    return NBX_OK;
}

int nbx_jk_p8(nbx_ctx* ctx, int64_t N, const double* d_packed, const double* d_dm, int64_t ndm, double* d_jk, void* d_work,
              const double* d_hv, double* d_fock, double* d_vhf, const double* d_dts_in) {
    const P8Plan& pl = p8_plan(N, ndm);
    const char* pk = reinterpret_cast<const char*>(d_packed);
    const int64_t* rowoff = reinterpret_cast<const int64_t*>(pk + pl.tab_rowoff);
    const int* t0 = reinterpret_cast<const int*>(pk + pl.tab_t0);
    const int* rowtab = reinterpret_cast<const int*>(pk + pl.tab_rowtab);
    const double* zeros = reinterpret_cast<const double*>(pk + pl.tab_zeros);
    char* base = static_cast<char*>(d_work);
    double* dtp = reinterpret_cast<double*>(base + pl.dts_off);
    double* jf = reinterpret_cast<double*>(base + pl.jf_off);
    double* jb = reinterpret_cast<double*>(base + pl.jb_off);
    double* jbl = reinterpret_cast<double*>(base + pl.jbl_off);
    double* kh = reinterpret_cast<double*>(base + pl.kh_off);
    double* k1 = reinterpret_cast<double*>(base + pl.k1_off);
    double* k2 = reinterpret_cast<double*>(base + pl.k2_off);
    const double* dts = d_dts_in;
    if (dts == nullptr) {
        const int rc = nbx_jk_s4_dtot(ctx, N, d_dm, ndm, dtp);
        if (rc != NBX_OK) return rc;
        dts = dtp;
    }
    static const int dbg = getenv("NBX_P8_DEBUG") ? atoi(getenv("NBX_P8_DEBUG")) : 0;  // ablation bits (timing only)
    {
        nbx_prof_scope prof(ctx, NBX_PROF_JK_DENSE);
#define NBX_P8_GO(NDM_)                                                                                                   \
    do {                                                                                                                  \
        static bool attr_set = false;                                                                                     \
        if (!attr_set) {                                                                                                  \
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&jk_p8_kernel<NDM_>),                                 \
                                      hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);                            \
            attr_set = true;                                                                                              \
        }                                                                                                                 \
        hipLaunchKernelGGL((jk_p8_kernel<NDM_>), dim3((unsigned)pl.wgs), dim3(512), pl.lds_bytes, ctx->stream,            \
                           d_packed, d_dm, dts, zeros, rowoff, rowtab, t0, jf, jb, k1, k2, (int)N, pl.S, dbg);                        \
    } while (0)
        if (ndm == 2) NBX_P8_GO(2);
        else NBX_P8_GO(1);
#undef NBX_P8_GO
    }
    NBX_LAUNCH_CHECK();
    const S4Geom g = s4_geom((int)N, P8_NB);
    hipLaunchKernelGGL(p8_kh_kernel, dim3((unsigned)N, (unsigned)ndm, (unsigned)nbx_cdiv(N, 64)), dim3(256), 0, ctx->stream,
                       k1, k2, t0, pl.nrng, kh, (int)N, (int)ndm, pl.S, g.s);
    NBX_LAUNCH_CHECK();
    hipLaunchKernelGGL(p8_jb_kernel, dim3((unsigned)nbx_cdiv(N * N, 64)), dim3(256), 0, ctx->stream, jb, pl.wgs, jbl,
                       (int)N, pl.lpt);
    NBX_LAUNCH_CHECK();
    hipLaunchKernelGGL(p8_finish_kernel, dim3((unsigned)nbx_cdiv(N * N, 256)), dim3(256), 0, ctx->stream, jf, jbl, kh, d_jk,
                       (int)N, g.s, (int)ndm, d_hv, d_fock, d_vhf);
    NBX_LAUNCH_CHECK();
    return NBX_OK;
}
