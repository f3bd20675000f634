// libnbx: the J/K contraction on the 8-FOLD packed (pq|rs) -- every integral read once per build (what stands behind
// get_veff of nbed/scf/huzinaga_scf.py:156; PySCF's own mf._eri is 8-fold packed too).
//
// jk_m4.hip streams the tiles T(p,q), q <= p, each the whole (r, s <= r) triangle: every (pq|rs) with (pq) != (rs) is in
// HBM twice, as element (rs) of tile (pq) and as element (pq) of tile (rs).  Here a tile keeps the elements (rs) <= (pq)
// only (jk_m8_layout.h: the block rows up to p / 4, in chunks), the element (rs) = (pq) halved, and each element does
// the work of both its copies:
//   K   the walk of jk_m4_walk.h gives Kp[p][r] += (pq|rs) D[q][s], Kp[p][s] += .. D[q][r], Kp[q][r] += .. D[p][s],
//       Kp[q][s] += .. D[p][r] as before (now r <= p: the row-q partial of a tile has columns up to p, not q); the four
//       terms of the mirrored copy are the transposes of these (symmetric D), so K = Kp + Kp^T -- no extra arithmetic;
//   J   J[pq] += (pq|rs) D'[rs] is the flat dot product of the tile with the Dtot' table in the loading waves' registers
//       as before; the mirrored copy's J[rs] += (pq|rs) D'[pq] is an AXPY of the tile into an accumulator of tile size in
//       the same registers (one more FMA per element, on the value the dot product has read back anyway), kept for the
//       whole range of the workgroup and summed over the workgroups in a fixed order by the reduction kernel.
// Tiles are of different lengths now, so (a) the chunk sequence of a range is produced by run-time iterators (which chunk
// of which tile goes into the ring next) while every chunk is the same number of LDS-DMA instructions -- the s_waitcnt
// immediates stay static --, (b) the code of a step is still specialised per chunk: the steps of a tile are unrolled
// over k with a uniform exit at k = nk(p), (c) the ranges of the persistent workgroups are cut at equal COST -- a tile costs
// its blocks plus a constant -- on the host (m8_ranges) and handed to the kernels as an argument, not at equal tile counts.
// Roles, ring, LDS-DMA, X operands, fixed-order partial rows: jk_m4.hip / jk_mx.hip.  Measurements, ablations and what was
// tried and dropped: profiles/r04/jk_m8_measurements.txt, DESIGN.md section 9 (4).
#include <cstdlib>
#include <type_traits>

#include "jk_m8_layout.h"
#include "jk_m4_walk.h"
#include "nbx_common.h"

#pragma clang diagnostic ignored "-Winline-asm"

namespace {

constexpr int M8_THREADS = 512;
typedef __attribute__((address_space(3))) void* m8_lds_vp;
typedef double m8_d2 __attribute__((ext_vector_type(2)));

__host__ __device__ __forceinline__ int m8_tri_row(int64_t T) {
    int64_t p = (int64_t)((sqrt(8.0 * (double)T + 1.0) - 1.0) * 0.5);
    while (p * (p + 1) / 2 > T) --p;
    while ((p + 1) * (p + 2) / 2 <= T) ++p;
    return (int)p;
}

template <int K0, int K1, class F>
__device__ __forceinline__ void m8_for(F&& f) {
    if constexpr (K0 < K1) {
        f(std::integral_constant<int, K0>{});
        m8_for<K0 + 1, K1>(f);
    }
}

template <int CNT>
__device__ __forceinline__ void m8_wait_vm() {
    static_assert(CNT >= 0 && CNT <= 63, "vmcnt is a 6-bit counter");
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(CNT) : "memory");
}

// ---------------------------------------------------------------------------------------------- pack and weights
// slab rows [p0, p0 + np) of the dense tensor -> 8-fold tiles; one workgroup per tile
// (nsrc <= N: the dense tensor's own size; the tiles p < nsrc exist, their elements beyond nsrc are stored as zeros)
template <class G>
__global__ __launch_bounds__(256) void m8_pack_kernel(const double* __restrict__ eri, double* __restrict__ out, int p0,
                                                      int64_t t_begin, int nsrc) {
    const int64_t T = t_begin + blockIdx.x;
    const int p = m8_tri_row(T), q = (int)(T - (int64_t)p * (p + 1) / 2);
    const double* src = eri + ((int64_t)(p - p0) * nsrc + q) * nsrc * nsrc;
    double* dst = out + (m8_tile_offset<G>(T) - m8_tile_offset<G>(t_begin));
    const int len = m8_len<G>(p);
    for (int e = threadIdx.x; e < len; e += 256) {
        const int blk = e >> 4;
        const int bt = m8_tri_row(blk), bc = blk - m4_tri(bt);
        const int k = ((e >> 2) & 3) ^ ((bt ^ bc) & 3), i = (e & 3) ^ k;  // (the swizzle: jk_m4.hip)
        const int row = 4 * bt + i, col = 4 * bc + k;
        const bool keep = col <= row && (row < p || (row == p && col <= q));  // (rs) <= (pq); row <= p < nsrc
        const double v = keep ? src[(int64_t)row * nsrc + col] : 0.0;
        dst[e] = (row == p && col == q) ? 0.5 * v : v;
    }
}

// Dtot' in the staging order (m8_stage_index): Dtot[r][c] + Dtot[c][r] below the diagonal, Dtot[r][r] on it, 0 elsewhere
template <class G>
__global__ __launch_bounds__(256) void m8_weights_kernel(const double* __restrict__ dm, int ndm, double* __restrict__ wt) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= G::NCH * G::LP * M4_PROD_THREADS) return;
    const int tid = i % M4_PROD_THREADS, s = (i / M4_PROD_THREADS) % G::LP, k = (i / M4_PROD_THREADS) / G::LP;
    const int d0 = (s * M4_PROD_THREADS + tid) * 2, dend = 16 * G::blocks(k);
    const int64_t n2 = (int64_t)G::N * G::N;
    double out[2] = {0.0, 0.0};
    for (int e = 0; e < 2; ++e) {
        const int d = d0 + e;
        if (d >= dend) continue;
        const int blk = G::start(k) + (d >> 4);
        const int bt = m8_tri_row(blk), bc = blk - m4_tri(bt);
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
// kpart1[(w S + slot) NDM + x][N]: row-p partial of workgroup w for the slot-th row of its range (columns <= p);
// kpart2[(T - t_begin) NDM + x][N]: row-q partial of tile T (q < p; columns <= p);
// jfull (N, N): J[p][q] = J[q][p] = the dot-product half of J for the tiles visited;
// jpart[w][NCH LP 512]: workgroup w's AXPY half of J, in the staging order
template <int NB, int NDM, int LP>
__global__ __launch_bounds__(M8_THREADS, 1) void jk_m8_kernel(const double* __restrict__ packed, const double* __restrict__ dm,
                                                              const double* __restrict__ wtab, double* __restrict__ jfull,
                                                              double* __restrict__ kpart1, double* __restrict__ kpart2,
                                                              double* __restrict__ jpart, int64_t t_begin, M8Ranges rg, int S) {
    using G_ = M8Geom<NB, LP>;
    constexpr int N = G_::N, NG = G_::NG, NCH = G_::NCH, BUF = G_::BUF, PT = M4_PROD_THREADS, RING = G_::RING, AHEAD = RING - 1;
    extern __shared__ __attribute__((aligned(16))) double smem[];
    double* buf0 = smem;                     // [RING][BUF] chunk buffers
    double* xs0 = smem + RING * BUF;         // [2][N][4] X of the current / next tile
    double* redq0 = xs0 + 2 * 4 * N;         // [2][4][NG][32] consumers' row-q halves (odd columns), by tile parity
    double* redp = redq0 + 2 * 4 * NG * 32;  // [4][NG][32] consumers' row-p halves (even columns) of a row that has ended
    double* jred = redp + 4 * NG * 32;       // [2][4] producers' J partials per tile parity
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const bool producer = wave >= 4;
    const int ptid = tid - 256;  // producers: 0 .. 255

    const int64_t T0 = t_begin + rg.first[blockIdx.x];
    const int64_t T_end = t_begin + rg.first[blockIdx.x + 1];
    if (T0 >= T_end) return;  // uniform for the whole workgroup
    const int p_first = m8_tri_row(T0);
    const int ntile = (int)(T_end - T0);
    const double* tile0 = packed + (m8_tile_offset<G_>(T0) - m8_tile_offset<G_>(t_begin));

    // X of a tile: xs[n][c] = D^{c / 2}[c & 1 ? p : q][n]  (NDM = 1: columns 2, 3 are zero); by the CONSUMER waves.  The two
    // q columns are new with every tile; the two p columns only when the row has changed since the buffer was last
    // filled (a vector-memory instruction costs its wave ~0.1 us of issue while the chip streams: one load per wave and
    // tile, not three).  Thread i of the 256 has the elements i, i + 256 of the 2 N of a column pair: (n, x) = (i >> 1, i & 1).
    constexpr int XH = (2 * N + PT - 1) / PT;
    auto fetch_xh = [&](int row, double (&v)[XH]) {
#pragma unroll
        for (int u = 0; u < XH; ++u) {
            const int i = tid + PT * u;
            const int n = i >> 1, x = i & 1;
            v[u] = (i < 2 * N && x < NDM) ? dm[x * N * N + row * N + n] : 0.0;
        }
    };
    auto store_xh = [&](double* xs, int col, const double (&v)[XH]) {  // col 0: the q columns, 1: the p columns
#pragma unroll
        for (int u = 0; u < XH; ++u) {
            const int i = tid + PT * u;
            if (i < 2 * N) xs[4 * (i >> 1) + 2 * (i & 1) + col] = v[u];
        }
    };
    // the consumers' partial rows of a finished tile: summed over the four consumer waves, in wave order; `last`: the
    // last column that is stored.  Loading thread 64 + u, u < 16 NG, has two neighbouring rows of one spin (lanes 2 j + parity and
    // 2 (j + 8) + parity of the D layout: rows 16 g + 4 a + b, b even and b + 1): one pass, one 16-byte store per thread
    // -- the partial rows of a tile are one store instruction each of three waves.
    auto reduce_rows = [&](const double* red, int parity, double* dst, int last) {  // dst[x N + row]
        // (the threads of loading waves 1 .. 3: wave 0 keeps the tiles' J -- 3.5 us of the kernel)
        static_assert(16 * NG <= 3 * 64, "the partial rows' pairs fit three waves");
        const int u = ptid - 64;
        if (u >= 0 && u < 16 * NG) {
            const int g = u >> 4, w = u & 15;
            const int x = w & 1, a4 = (w >> 1) & 3, b2 = (w >> 3) & 1;
            const int e0 = g * 32 + x + 2 * a4 + 16 * b2, e1 = e0 + 8;
            const int row = 16 * g + 4 * a4 + 2 * b2;
            (void)parity;  // (the buffers hold one parity each: the lane index is (lane >> 1))
            const double v0 = (red[e0] + red[NG * 32 + e0]) + (red[2 * NG * 32 + e0] + red[3 * NG * 32 + e0]);
            const double v1 = (red[e1] + red[NG * 32 + e1]) + (red[2 * NG * 32 + e1] + red[3 * NG * 32 + e1]);
            if (x < NDM) {
                // (non-temporal stores, here and for the J partials: -2 us in this kernel, within the noise of the build)
                if (row + 1 <= last) *reinterpret_cast<double2*>(dst + x * N + row) = make_double2(v0, v1);
                else if (row == last) dst[x * N + row] = v0;
            }
        }
    };
    auto next_pq = [](int& pp, int& qq) {
        if (++qq > pp) {
            ++pp;
            qq = 0;
        }
    };

    int p = p_first, q = (int)(T0 - (int64_t)p * (p + 1) / 2);
    if (producer) {
        // ------------------------------------------------------------------ the loading waves
        m8_d2 wres[NCH][LP];  // Dtot' of the whole tile (the same for every tile)
        double j2x[NCH][LP], j2y[NCH][LP];  // J[rs] += (pq|rs) D'[pq], summed over the tiles of the range
        double jacc = 0.0;
#pragma unroll
        for (int k = 0; k < NCH; ++k)
#pragma unroll
            for (int s = 0; s < LP; ++s) {
                wres[k][s] = *reinterpret_cast<const m8_d2*>(wtab + 2 * ((k * LP + s) * PT + ptid));
                j2x[k][s] = j2y[k][s] = 0.0;
            }
        if (ptid < 8) jred[ptid] = 0.0;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // (the table: nothing of the compiler's in the counter from here)
#pragma unroll
        for (int k = 0; k < NCH; ++k)
#pragma unroll
            for (int s = 0; s < LP; ++s) asm volatile("" : "+v"(wres[k][s]));  // (the compiler's own wait for the table: here, not behind the first chunk loads)
        // the chunk that goes into the ring next: chunk ik of tile it = (ip, iq), which has ink chunks and starts at itile
        int it = 0, ik = 0, ip = p, iq = q, ink = m8_nk<G_>(p);
        const double* itile = tile0;
        int islot = 0;
        // Chunk -> ring slot: global_load_lds_dwordx4, lane l of a wave lands its 16 bytes at the instruction's LDS base +
        // 16 l; slot s of the chunk is one instruction per producer wave (thread ptid's two doubles of slot s at
        // (s PT + ptid) 2).  Every chunk is LP instructions per wave (the tail re-reads the chunk's last 16 bytes; a chunk
        // past the last tile re-reads the first tile's first 16 bytes: landed in a free slot, never read -- leaving those loads out,
        // with a wait that counts what is really in flight, was measured 5 us SLOWER: the count costs every step).
        auto issue_next = [&]() {
            const bool real = it < ntile;
            const double* tile = real ? itile : tile0;
            const int kk = real ? ik : 0;
            int begin = 0, end = 16 * G_::blocks(0);
            m8_for<1, NCH>([&](auto kc) {
                constexpr int k = decltype(kc)::value;
                begin = kk == k ? 16 * G_::start(k) : begin;
                end = kk == k ? 16 * (G_::start(k) + G_::blocks(k)) : end;
            });
            end = real ? end : 2;  // (a chunk past the last tile: every lane the tile's first 16 bytes -- one cache line per instruction)
            double* buf = buf0 + islot * BUF;
            islot = islot + 1 == RING ? 0 : islot + 1;
            int pt_ = ptid;  // (opaque per chunk: the clamped offsets are recomputed, not kept in registers)
            asm volatile("" : "+v"(pt_));
#pragma unroll
            for (int s = 0; s < LP; ++s) {
                int d = begin + (s * PT + pt_) * 2;
                d = min(d, end - 2);
                const unsigned off = 8u * (unsigned)d;
                const unsigned lds_a = (unsigned)(size_t)(m8_lds_vp)(buf + (s * PT + (wave - 4) * 64) * 2);
                asm volatile("s_mov_b32 m0, %2\n\tglobal_load_lds_dwordx4 %0, %1 nt" : : "v"(off), "s"(tile), "s"(lds_a) : "memory", "m0");
            }
            if (real && ++ik == ink) {
                itile += m8_len<G_>(ip);
                ik = 0;
                ++it;
                next_pq(ip, iq);
                ink = m8_nk<G_>(ip);
            }
        };
#pragma unroll
        for (int c = 0; c < AHEAD; ++c) issue_next();
        m8_wait_vm<(AHEAD - 1) * LP>();  // my part of the first chunk
        __syncthreads();
        int jslot = 0;  // ring slot of the chunk the consumers walk at this step
        // J[pq] of the tiles done: sixty-four of them wait in one register of wave 4 (lane i: tile jbase + i of the range) --
        // two scattered stores per sixty-four tiles instead of per tile, and no LDS
        double jhold = 0.0;
        int jbase = 0;
        auto jflush = [&](double jv, int base, int count) {
            if (lane < count) {
                const int64_t Ti = T0 + base + lane;
                const int pi = m8_tri_row(Ti), qi = (int)(Ti - (int64_t)pi * (pi + 1) / 2);
                jfull[(int64_t)pi * N + qi] = jv;
                jfull[(int64_t)qi * N + pi] = jv;
            }
        };
        for (int t = 0; t < ntile; ++t) {
            const int64_t T = T0 + t;
            const int nk = m8_nk<G_>(p);
            int pn = p, qn = q;
            next_pq(pn, qn);
            // D'[pq] of this tile from its X (xs[n][c] = D^{c / 2}[c & 1 ? p : q][n])
            double dpq;
            {
                const double* xs = xs0 + (t & 1) * 4 * N;
                const double dp = xs[4 * q + 1] + xs[4 * q + 3];  // sum over spins of D[p][q]
                const double dq = xs[4 * p + 0] + xs[4 * p + 2];  //                    D[q][p]
                dpq = q < p ? dp + dq : dp;
            }
            m8_for<0, NCH>([&](auto kc) {
                constexpr int k = decltype(kc)::value;
                if (k < nk) {
                    // step (t, k): the consumers walk chunk k; the chunk AHEAD steps on goes into the slot they left at the
                    // last barrier; the next chunk has landed when this step ends
                    issue_next();  // (first: the stream is what the kernel is bound by)
                    if (k == 0 && t > 0) {
                        // the consumers' rows of tile t - 1 (written at its last step, behind that step's barrier), and its J
                        int pp = p, qq = q - 1;
                        if (qq < 0) {
                            pp = p - 1;
                            qq = pp;
                        }
                        if (qq < pp) reduce_rows(redq0 + ((t - 1) & 1) * 4 * NG * 32, 1, kpart2 + ((T - 1 - t_begin) * NDM) * (int64_t)N, pp);
                        if (pp != p) reduce_rows(redp, 0, kpart1 + (((int64_t)blockIdx.x * S + (pp - p_first)) * NDM) * N, pp);
                        if (wave == 4) {  // J of tile t - 1: lane (t - 1) & 63 of this wave keeps it
                            const double* jr = jred + ((t - 1) & 1) * 4;
                            const double js = (jr[0] + jr[1]) + (jr[2] + jr[3]);
                            jhold = lane == ((t - 1) & 63) ? js : jhold;
                            if (((t - 1) & 63) == 63) {
                                jflush(jhold, jbase, 64);
                                jbase += 64;
                            }
                        }
                        // (the one case in which the consumers write the row-p buffer in the step in which it is read: the
                        //  range ends with a single-step tile that opens a row -- the last tile always hands its row over)
                        if (pp != p && nk == 1 && t + 1 == ntile) __syncthreads();
                    }
                    {  // the two J contributions of chunk k (the one being walked: it landed a step ago)
                        const double* buf = buf0 + jslot * BUF;
                        jslot = jslot + 1 == RING ? 0 : jslot + 1;
#pragma unroll
                        for (int s = 0; s < LP; ++s) {
                            const double2 v = *reinterpret_cast<const double2*>(buf + (s * PT + ptid) * 2);
                            jacc = fma(v.y, wres[k][s].y, fma(v.x, wres[k][s].x, jacc));
                            j2x[k][s] = fma(v.x, dpq, j2x[k][s]);
                            j2y[k][s] = fma(v.y, dpq, j2y[k][s]);
                            asm volatile("" : "+v"(j2x[k][s]), "+v"(j2y[k][s]));  // (here: left alone, the compiler sinks the FMAs of every
                                                                                    //  step to the tile's end and keeps each chunk's read-back alive)
                        }
                        asm volatile("" : "+v"(jacc));
                    }
                    if (k == nk - 1) {  // this wave's share of tile t's J[pq]
                        jacc = nbx_wave_sum_dpp(jacc);  // (every lane active; not the LDS butterfly: six ds_bpermute round trips, 0.4 us, at every tile end)
                        if (lane == 0) jred[(t & 1) * 4 + (wave - 4)] = jacc;
                        jacc = 0.0;
                    }
                    m8_wait_vm<(AHEAD - 1) * LP>();  // my part of the next chunk
                    __syncthreads();
                }
            });
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
        if (qq < pp) reduce_rows(redq0 + ((ntile - 1) & 1) * 4 * NG * 32, 1, kpart2 + ((T_end - 1 - t_begin) * NDM) * (int64_t)N, pp);
        reduce_rows(redp, 0, kpart1 + (((int64_t)blockIdx.x * S + (pp - p_first)) * NDM) * N, pp);
        // the AXPY half of J of this range
        {
            double* dst = jpart + (int64_t)blockIdx.x * (NCH * LP * PT * 2) + 2 * ptid;
            const int nkmax = m8_nk<G_>(pp);  // (the last tile's: rows only grow along a range)
#pragma unroll
            for (int k = 0; k < NCH; ++k)
                if (k < nkmax) {
#pragma unroll
                    for (int s = 0; s < LP; ++s)
                        *reinterpret_cast<double2*>(dst + (k * LP + s) * PT * 2) = make_double2(j2x[k][s], j2y[k][s]);
                }
        }
        if (wave == 4) {  // J of the tiles not stored yet
            const double* jr = jred + ((ntile - 1) & 1) * 4;
            const double js = (jr[0] + jr[1]) + (jr[2] + jr[3]);
            jhold = lane == ((ntile - 1) & 63) ? js : jhold;
            jflush(jhold, jbase, ntile - jbase);
        }
    } else {
        // ------------------------------------------------------------------ the walking waves
        double acc[NG], bxr[NG];
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
        // X two tiles ahead: the X of tile t + 1 is stored at the first step of tile t (into the buffer tile t - 1 has left),
        // from registers that were loaded at the first step of tile t - 1 -- a tile may be a single step
        double xq[XH], xp[XH];
        int pbuf0 = p, pbuf1 = -1;  // the row whose p columns each buffer holds
        fetch_xh(q, xq);
        fetch_xh(p, xp);
        store_xh(xs0, 0, xq);
        store_xh(xs0, 1, xp);
        int px = p, qx = q;  // the tile whose X is fetched next
        if (ntile > 1) next_pq(px, qx);
        bool xp_new = ntile > 1;  // the registers hold p columns that the target buffer does not have yet
        fetch_xh(qx, xq);  // X of tile 1
        if (xp_new) {
            fetch_xh(px, xp);
            pbuf1 = px;
        }
        __syncthreads();
        int slot = 0;  // ring slot of the chunk being walked
        for (int t = 0; t < ntile; ++t) {
            const int nk = m8_nk<G_>(p);
            int pn = p, qn = q;
            next_pq(pn, qn);
            const bool more = t + 1 < ntile;
            const bool row_ends = pn != p || !more;
            const double* xs = xs0 + (t & 1) * 4 * N;
            m8_for<0, NCH>([&](auto kc) {
                constexpr int k = decltype(kc)::value;
                if (k < nk) {
                    const double* buf = buf0 + slot * BUF;
                    slot = slot + 1 == RING ? 0 : slot + 1;
                    m4_walk_chunk<G_, k>(buf, xs, wave, ln, acc, bxr);
                    if (k == 0) {
                        if (more) {
                            double* xn = xs0 + ((t + 1) & 1) * 4 * N;
                            store_xh(xn, 0, xq);
                            if (xp_new) store_xh(xn, 1, xp);
                        }
                        if (t + 2 < ntile) {  // the X of tile t + 2, for the buffer this tile is being walked from
                            next_pq(px, qx);
                            fetch_xh(qx, xq);
                            int& pb = (t & 1) ? pbuf1 : pbuf0;
                            xp_new = pb != px;
                            if (xp_new) {
                                fetch_xh(px, xp);
                                pb = px;
                            }
                        } else {
                            xp_new = false;
                        }
                    }
                    if (k == nk - 1) {
                        // end of tile: the row-q halves (odd columns: they used D[p][:]) leave the registers; the row-p halves
                        // (even columns) stay until the row changes.  The producers sum them up during the next step.
                        const bool odd = lane & 1;
                        if (t > 0 && q == 0 && nk == 1 && !more) __syncthreads();  // (see the loading waves: the row-p buffer is still being read)
                        // (one store region: a branch per accumulator and buffer is what the compiler makes of the two conditions)
                        const bool wr = odd || row_ends;
                        double* dst = (odd ? redq0 + (t & 1) * 4 * NG * 32 : redp) + wave * NG * 32 + (lane >> 1);
                        if (wr) {
#pragma unroll
                            for (int g = 0; g < NG; ++g) dst[g * 32] = acc[g];
                        }
#pragma unroll
                        for (int g = 0; g < NG; ++g) acc[g] = wr ? 0.0 : acc[g];
                    }
                    __builtin_amdgcn_sched_barrier(0);  // (the MFMAs stay above the barrier: jk_m4.hip)
                    __syncthreads();
                }
            });
            p = pn;
            q = qn;
        }
    }
}

// ---------------------------------------------------------------------------------------------- the reductions
// (1) kpf[x][r][c] = Kp[r][c], every column: the row-p partials of the workgroups whose range has tiles of row r (c <= r)
//     + the row-q partials of the tiles (P, r), P > r, P >= c.   N NDM ceil(N / 64) blocks, 512 threads = 8 interleaved
//     partial sums over P x 64 columns; fixed summation order.
// (2) (the blocks behind those, same launch) j2sum[e] = sum over the workgroups of jpart[w][e], e in the staging order
template <class G>
__global__ __launch_bounds__(512) void m8_reduce_kernel(const double* __restrict__ kpart1, const double* __restrict__ kpart2,
                                                        const double* __restrict__ jpart, double* __restrict__ kpf,
                                                        double* __restrict__ j2sum, int p0, int np, int ndm, int64_t t_begin,
                                                        M8Ranges rg, int W, int S) {
    constexpr int N = G::N, JCH = G::LP * M4_PROD_THREADS * 2, JLEN = G::NCH * JCH, CZ = (N + 63) / 64;
    static_assert(JCH % 64 == 0, "a block's 32 pairs lie in one chunk");
    __shared__ double part[16][64];
    const int nkb = N * ndm * CZ;  // blocks of the first kind
    if ((int)blockIdx.x >= nkb) {
        // 32 double pairs per block, the workgroups that have this chunk in sixteen interleaved groups
        const int blk = (int)blockIdx.x - nkb;
        const int l = threadIdx.x & 31, grp = threadIdx.x >> 5;
        const int e = (blk * 32 + l) * 2;
        const int w0 = rg.wmin[(blk * 64) / JCH];
        double sx = 0.0, sy = 0.0;
        int w = w0 + grp;
        // (a workgroup without tiles -- a tile dearer than a workgroup's share of a small slab -- has written nothing)
        auto at = [&](int v) {
            return rg.first[v + 1] > rg.first[v] ? *reinterpret_cast<const double2*>(jpart + (int64_t)v * JLEN + e) : make_double2(0.0, 0.0);
        };
        for (; w + 7 * 16 < W; w += 8 * 16) {
            double2 v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = at(w + 16 * u);
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                sx += v[u].x;
                sy += v[u].y;
            }
        }
        for (; w < W; w += 16) {
            const double2 v = at(w);
            sx += v.x;
            sy += v.y;
        }
        part[grp][l] = sx;
        part[grp][32 + l] = sy;
        __syncthreads();
        if (grp == 0) {
            double tx = part[0][l], ty = part[0][32 + l];
#pragma unroll
            for (int g = 1; g < 16; ++g) {
                tx += part[g][l];
                ty += part[g][32 + l];
            }
            *reinterpret_cast<double2*>(j2sum + e) = make_double2(tx, ty);
        }
        return;
    }
    const int row = (int)blockIdx.x % N, x = ((int)blockIdx.x / N) % ndm, bz = (int)blockIdx.x / (N * ndm);
    const int lane = threadIdx.x & 63, chunk = threadIdx.x >> 6;
    const int c = bz * 64 + lane;
    double t = 0.0;
    if (c < N) {
        const double* src = kpart2 + (int64_t)x * N + c;
        const int64_t stride = (int64_t)ndm * N;
        const int first = max(max(row + 1, c), p0);  // first global P
        auto at = [&](int P) { return src[((int64_t)P * (P + 1) / 2 + row - t_begin) * stride]; };
        int P = first + chunk;
        for (; P + 7 * 8 < p0 + np; P += 8 * 8) {
            double v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = at(P + 8 * u);
#pragma unroll
            for (int u = 0; u < 8; ++u) t += v[u];
        }
        for (; P < p0 + np; P += 8) t += at(P);
    }
    part[chunk][lane] = t;
    __syncthreads();
    if (chunk == 0 && c < N) {
        double tot = ((part[0][lane] + part[1][lane]) + (part[2][lane] + part[3][lane])) +
                     ((part[4][lane] + part[5][lane]) + (part[6][lane] + part[7][lane]));
        if (row >= p0 && row < p0 + np && c <= row) {
            const int trow = (int)(m4_tri(row) - t_begin);  // (first tile of the row, relative)
            const int w_lo = m8_wg_of(rg, W, trow), w_hi = m8_wg_of(rg, W, trow + row);
            for (int w = w_lo; w <= w_hi; ++w) {
                if (rg.first[w + 1] <= rg.first[w]) continue;  // (no tiles: nothing written)
                const int slot = row - m8_tri_row(t_begin + rg.first[w]);
                tot += kpart1[(((int64_t)w * S + slot) * ndm + x) * N + c];
            }
        }
        kpf[((int64_t)x * N + row) * N + c] = tot;
    }
}

// K = Kp + Kp^T, J = the dot-product half + the AXPY half (J[r][c] = J[c][r]), and the Fock epilogue F = hv + J - K
template <class G>
__global__ __launch_bounds__(256) void m8_finish_kernel(const double* __restrict__ kpf, const double* __restrict__ j2sum,
                                                        double* __restrict__ jout, double* __restrict__ kout, int ndm,
                                                        const double* __restrict__ hv, double* __restrict__ fock,
                                                        double* __restrict__ vhf) {
    constexpr int N = G::N;
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= N * N) return;
    const int r = i / N, c = i - r * N;
    const int hi = r > c ? r : c, lo = r > c ? c : r;
    const double j = jout[i] + j2sum[m8_stage_index<G>(hi, lo)];
    for (int x = 0; x < ndm; ++x) {
        const double k = kpf[((int64_t)x * N + r) * N + c] + kpf[((int64_t)x * N + c) * N + r];
        const int64_t o = (int64_t)x * N * N + i;
        kout[o] = k;
        if (fock != nullptr) {
            const double v = j - k;
            fock[o] = hv[o] + v;
            if (vhf != nullptr) vhf[o] = v;
        }
    }
    jout[i] = j;
}

size_t m8_align256(size_t x) { return (x + 255) & ~(size_t)255; }

struct M8Plan {
    int wgs, S, lmax;
    size_t wt_off, k1_off, k2_off, jp_off, js_off, kpf_off, total, lds_bytes;
};

// The ranges of the workgroups: contiguous in the tile sequence, cut at equal COST: a tile costs its blocks plus a
// constant for the hand-over of its rows (measured: 0.6 us per tile next to 1.05 us per chunk of 24 KB).
// (NBX_M8_TC in the environment, in blocks: for measurements; read once)
template <class G>
const M8Ranges& m8_ranges(int64_t p0, int64_t np, int* wgs_out) {
    static thread_local M8Ranges rg;
    static thread_local int64_t key_p0 = -1, key_np = -1;
    static thread_local int wgs = 0;
    if (key_p0 != p0 || key_np != np) {
        static const int tile_cost = getenv("NBX_M8_TC") ? atoi(getenv("NBX_M8_TC")) : 120;
        const int64_t t_begin = m4_tri((int)p0), t_end = m4_tri((int)(p0 + np));
        const int64_t ntiles = t_end - t_begin;
        wgs = (int)(ntiles < M8_CUS ? (ntiles > 0 ? ntiles : 1) : M8_CUS);
        auto cost_of_row = [&](int p) {
            return (int64_t)(m8_len<G>(p) / 16 + tile_cost);
        };
        int64_t total = 0;
        for (int p = (int)p0; p < (int)(p0 + np); ++p) total += (int64_t)(p + 1) * cost_of_row(p);
        // workgroup w starts at the first tile with (cost before it) W >= w total
        int w = 0;
        int64_t before = 0;
        int64_t trel = 0;
        for (int p = (int)p0; p < (int)(p0 + np); ++p) {
            const int64_t c = cost_of_row(p);
            for (int q = 0; q <= p; ++q, ++trel) {
                while (w < wgs && before * wgs >= (int64_t)w * total) rg.first[w++] = (int)trel;
                before += c;
            }
        }
        while (w <= M8_CUS) rg.first[w++] = (int)ntiles;
        for (int k = 0; k < M8_MAXCH; ++k) rg.wmin[k] = wgs;
        for (int v = wgs - 1; v >= 0; --v) {
            if (rg.first[v + 1] <= rg.first[v]) continue;
            const int nkmax = m8_nk<G>(m8_tri_row(t_begin + rg.first[v + 1] - 1));
            for (int k = 0; k < nkmax && k < M8_MAXCH; ++k) rg.wmin[k] = v;
        }
        key_p0 = p0;
        key_np = np;
    }
    *wgs_out = wgs;
    return rg;
}

template <class G>
M8Plan m8_plan(int64_t p0, int64_t np, int64_t ndm) {
    M8Plan pl;
    const int64_t t_begin = m4_tri((int)p0), t_end = m4_tri((int)(p0 + np));
    const int64_t ntiles = t_end - t_begin;
    const M8Ranges& rg = m8_ranges<G>(p0, np, &pl.wgs);
    int S = 1, lmax = 1;
    for (int w = 0; w < pl.wgs; ++w) {
        const int64_t a = t_begin + rg.first[w], b = t_begin + rg.first[w + 1];
        if (b <= a) continue;
        const int rows = m8_tri_row(b - 1) - m8_tri_row(a) + 1;
        S = rows > S ? rows : S;
        lmax = (int)(b - a) > lmax ? (int)(b - a) : lmax;
    }
    pl.S = S;
    pl.lmax = lmax;
    pl.lds_bytes = (size_t)(G::RING * G::BUF + G::FIXED) * sizeof(double);
    const size_t jlen = (size_t)G::NCH * G::LP * M4_PROD_THREADS * 2;
    size_t off = 0;
    pl.wt_off = off; off += m8_align256(jlen * sizeof(double));
    pl.k1_off = off; off += m8_align256((size_t)((int64_t)pl.wgs * pl.S * ndm * G::N) * sizeof(double));
    pl.k2_off = off; off += m8_align256((size_t)(ntiles * ndm * G::N) * sizeof(double));
    pl.jp_off = off; off += m8_align256((size_t)pl.wgs * jlen * sizeof(double));
    pl.js_off = off; off += m8_align256(jlen * sizeof(double));
    pl.kpf_off = off; off += m8_align256((size_t)(ndm * G::N * G::N) * sizeof(double));
    pl.total = off;
    return pl;
}

template <int NB, int LP>
int m8_run(nbx_ctx* ctx, int64_t p0, int64_t p1, const double* d_packed, const double* d_dm, int64_t ndm, double* d_jk,
           void* d_work, const double* d_hv, double* d_fock, double* d_vhf, const double* d_wt_in) {
    using G = M8Geom<NB, LP>;
    const int64_t np = p1 - p0, N = G::N, n2 = N * N;
    const M8Plan pl = m8_plan<G>(p0, np, ndm);
    int wgs_;
    const M8Ranges& rg = m8_ranges<G>(p0, np, &wgs_);
    if (pl.lds_bytes > (size_t)M8_LDS_BYTES) {
        nbx_set_error("nbx_jk_m8: %zu bytes of LDS for N = %lld", pl.lds_bytes, (long long)N);
        return NBX_E_UNSUPPORTED;
    }
    char* base = static_cast<char*>(d_work);
    double* wt = reinterpret_cast<double*>(base + pl.wt_off);
    double* k1 = reinterpret_cast<double*>(base + pl.k1_off);
    double* k2 = reinterpret_cast<double*>(base + pl.k2_off);
    double* jp = reinterpret_cast<double*>(base + pl.jp_off);
    double* js = reinterpret_cast<double*>(base + pl.js_off);
    double* kpf = reinterpret_cast<double*>(base + pl.kpf_off);
    if (np < N) {  // J entries this slab does not own must read as zero
        const int rc = nbx_memset(ctx, d_jk, 0, (size_t)n2 * sizeof(double));
        if (rc != NBX_OK) return rc;
    }
    constexpr int JSLOTS = G::NCH * G::LP * M4_PROD_THREADS;
    if (d_wt_in != nullptr) {
        wt = const_cast<double*>(d_wt_in);
    } else {
        hipLaunchKernelGGL(m8_weights_kernel<G>, dim3((unsigned)nbx_cdiv(JSLOTS, 256)), dim3(256), 0, ctx->stream, d_dm, (int)ndm, wt);
        NBX_LAUNCH_CHECK();
    }
    const int64_t t_begin = m4_tri((int)p0);
    {
        nbx_prof_scope prof(ctx, NBX_PROF_JK_DENSE);
        static bool attr_set = false;
        if (!attr_set) {
            const hipError_t e1 = hipFuncSetAttribute(reinterpret_cast<const void*>(&jk_m8_kernel<NB, 1, LP>),
                                                      hipFuncAttributeMaxDynamicSharedMemorySize, M8_LDS_BYTES);
            const hipError_t e2 = hipFuncSetAttribute(reinterpret_cast<const void*>(&jk_m8_kernel<NB, 2, LP>),
                                                      hipFuncAttributeMaxDynamicSharedMemorySize, M8_LDS_BYTES);
            if (e1 != hipSuccess || e2 != hipSuccess) {
                nbx_set_error("nbx_jk_m8: hipFuncSetAttribute(%d bytes of LDS): %s", M8_LDS_BYTES,
                              hipGetErrorString(e1 != hipSuccess ? e1 : e2));
                return NBX_E_HIP;
            }
            attr_set = true;
        }
        if (ndm == 2)
            hipLaunchKernelGGL((jk_m8_kernel<NB, 2, LP>), dim3((unsigned)pl.wgs), dim3(M8_THREADS), pl.lds_bytes, ctx->stream,
                               d_packed, d_dm, wt, d_jk, k1, k2, jp, t_begin, rg, pl.S);
        else
            hipLaunchKernelGGL((jk_m8_kernel<NB, 1, LP>), dim3((unsigned)pl.wgs), dim3(M8_THREADS), pl.lds_bytes, ctx->stream,
                               d_packed, d_dm, wt, d_jk, k1, k2, jp, t_begin, rg, pl.S);
    }
    NBX_LAUNCH_CHECK();
    {  // the two reductions in one launch
        const int jblocks = (int)nbx_cdiv(JSLOTS, 32);
        const int nkb = (int)(N * ndm * nbx_cdiv(N, 64));
        hipLaunchKernelGGL(m8_reduce_kernel<G>, dim3((unsigned)(nkb + jblocks)), dim3(512), 0, ctx->stream, k1, k2, jp, kpf, js,
                           (int)p0, (int)np, (int)ndm, t_begin, rg, pl.wgs, pl.S);
        NBX_LAUNCH_CHECK();
    }
    hipLaunchKernelGGL(m8_finish_kernel<G>, dim3((unsigned)nbx_cdiv(n2, 256)), dim3(256), 0, ctx->stream, kpf, js, d_jk, d_jk + n2,
                       (int)ndm, d_hv, d_fock, d_vhf);
    NBX_LAUNCH_CHECK();
    return NBX_OK;
}

}  // namespace

// The sizes this kernel has an instance for (N = 4 NB).  NBX_JK_M8=0 in the environment (read once per process) hands
// them back to jk_m4.hip's 4-fold form (the packed tensor is then the 4-fold one: the switch is read before packing).
#ifndef NBX_M8_SIZES
#define NBX_M8_SIZES(X) X(25) X(26) X(27) X(28) X(29) X(30) X(31) X(32) X(33) X(34) X(35) X(36) X(37)  // N = 100 .. 148
#endif
#ifdef NBX_M8_LP  // (measurements: one chunk size for every instance)
#define M8_LP(NB_) NBX_M8_LP
#else
#define M8_LP(NB_) m8_lp(NB_)
#endif
#define M8_DISPATCH(N_, EXPR)            \
    switch ((int)((N_) / 4)) {           \
        NBX_M8_SIZES(M8_CASE_##EXPR)     \
        default: break;                  \
    }
bool nbx_jk_m8_covers(int64_t N) {
    static const bool on = getenv("NBX_JK_M8") == nullptr || atoi(getenv("NBX_JK_M8")) != 0;
    if (!on || N % 4 != 0) return false;
#define M8_CASE_covers(NB_) case NB_: return true;
    M8_DISPATCH(N, covers)
#undef M8_CASE_covers
    return false;
}

size_t nbx_jk_m8_packed_bytes(int64_t N, int64_t p0, int64_t p1) {
    const int64_t t0 = m4_tri((int)p0), t1 = m4_tri((int)p1);
    // (+ slack: a chunk past the last tile of a range re-reads the range's first tile, nothing beyond the array)
#define M8_CASE_bytes(NB_) \
    case NB_: return (size_t)(m8_tile_offset<M8Geom<NB_, M8_LP(NB_)>>(t1) - m8_tile_offset<M8Geom<NB_, M8_LP(NB_)>>(t0)) * sizeof(double) + 256;
    M8_DISPATCH(N, bytes)
#undef M8_CASE_bytes
    return 0;
}

size_t nbx_jk_m8_worksize(int64_t N, int64_t p0, int64_t p1, int64_t ndm) {
#define M8_CASE_work(NB_) case NB_: return m8_plan<M8Geom<NB_, M8_LP(NB_)>>(p0, p1 - p0, ndm).total;
    M8_DISPATCH(N, work)
#undef M8_CASE_work
    return 0;
}

int nbx_jk_m8_pack(nbx_ctx* ctx, int64_t N, int64_t nsrc, int64_t p0, int64_t p1, const double* d_eri, double* d_packed) {
    NBX_CHECK_ARG(nbx_jk_m8_covers(N) && d_eri && d_packed && nsrc <= N && nsrc > N - 4 && p1 <= nsrc);  // (nsrc < N: zero rows and columns beyond)
    const int64_t ntiles = m4_tri((int)p1) - m4_tri((int)p0);
#define M8_CASE_pack(NB_)                                                                                                  \
    case NB_:                                                                                                              \
        hipLaunchKernelGGL((m8_pack_kernel<M8Geom<NB_, M8_LP(NB_)>>), dim3((unsigned)ntiles), dim3(256), 0, ctx->stream, d_eri, \
                           d_packed, (int)p0, (int64_t)m4_tri((int)p0), (int)nsrc);                                        \
        break;
    M8_DISPATCH(N, pack)
#undef M8_CASE_pack
    NBX_LAUNCH_CHECK();
    return NBX_OK;
}

// the Dtot' weights table of a density (m8_stage_index order; the entries nothing writes are zero weights)
size_t nbx_jk_m8_weights_bytes(int64_t N) {
#define M8_CASE_wbytes(NB_) \
    case NB_: return (size_t)(M8Geom<NB_, M8_LP(NB_)>::NCH * M8_LP(NB_) * M4_PROD_THREADS * 2) * sizeof(double);
    M8_DISPATCH(N, wbytes)
#undef M8_CASE_wbytes
    return 0;
}

// What huz_scalars_kernel needs to write that table for size N -- with at most four chunks the staging order is
// jk_m4.hip's with other chunk boundaries (m4_weight_index_rt; a chunk that does not exist begins at block row NB): the
// first block rows of chunks 1..3 and the slots per chunk; zeros: the instance has more chunks and prepares its table itself
void nbx_jk_m8_weight_layout(int64_t N, int out[4]) {
    out[0] = out[1] = out[2] = out[3] = 0;
#define M8_CASE_wl(NB_)                                   \
    case NB_: {                                           \
        using G = M8Geom<NB_, M8_LP(NB_)>;                 \
        if (G::NCH <= 4) {                                \
            out[0] = G::row0(1);                          \
            out[1] = G::row0(2);                          \
            out[2] = G::row0(3);                          \
            out[3] = G::LP;                               \
        }                                                 \
        break;                                            \
    }
    M8_DISPATCH(N, wl)
#undef M8_CASE_wl
}

// d_wt: NULL, or that table for d_dm: saves the preparation launch
int nbx_jk_m8(nbx_ctx* ctx, int64_t N, int64_t p0, int64_t p1, const double* d_packed, const double* d_dm, int64_t ndm,
              double* d_jk, void* d_work, const double* d_hv, double* d_fock, double* d_vhf, const double* d_wt) {
    NBX_CHECK_ARG(nbx_jk_m8_covers(N));
#define M8_CASE_run(NB_) \
    case NB_: return m8_run<NB_, M8_LP(NB_)>(ctx, p0, p1, d_packed, d_dm, ndm, d_jk, d_work, d_hv, d_fock, d_vhf, d_wt);
    M8_DISPATCH(N, run)
#undef M8_CASE_run
    return NBX_E_UNSUPPORTED;
}
