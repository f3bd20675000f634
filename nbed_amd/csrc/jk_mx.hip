// libnbx: the packed J/K contraction with its walk on the matrix cores for the sizes ABOVE jk_m4.hip's (N = 152 .. 288:
// what stands behind get_veff of nbed/scf/huzinaga_scf.py:156 at the sizes where a slab per GPU pays).
//
// Same tiles, same layout (jk_m4.hip: T(p,q) = p(p+1)/2 + q in sequence, a tile = the lower triangle of its (r,s)
// matrix in swizzled 4 x 4 blocks), same walk (jk_m4_walk.h), same persistent equal-range workgroups of four walking
// and four loading waves around a ring of five LDS buffers filled by LDS-DMA, same partial rows and fixed-order
// reduction.  What is different, because a tile of 95 .. 340 KB no longer goes through the ring in four chunks:
//   * a tile is NCH = 4 .. 20 chunks of whole block rows (MxGeom, jk_m4_layout.h), each as many 4 KB LDS-DMA
//     instructions per loading wave as it needs (lpt(k) of them: the counts are static per chunk, and so are the
//     `s_waitcnt vmcnt` immediates -- "my part of the chunk three behind the newest has landed");
//   * J: the flat dot product of a chunk with its Dtot' weights needs the weights of the WHOLE tile, which a loading
//     wave's registers hold for four chunks only.  The weights of chunks 0 .. 3 stay resident as in jk_m4.hip; those
//     of chunk c >= 4 are fetched from the table (L2-resident: every workgroup reads the same 0.1 .. 0.4 MB per tile)
//     four steps ahead of their use, into one of four rotating register sets, by inline-asm loads in the same
//     in-order queue as the LDS-DMA loads -- a load the compiler tracks would be waited for with a count that does not
//     know of the DMA loads around it, i.e. with the whole queue drained.  The life of such a register set begins
//     and ends inside the straight-line code of one tile.
// The per-tile row partials all go out as they are finished (no held-back rows here: a range is 47 .. 163 tiles and a
// build 0.2 .. 2.4 ms; the 10 us that buys at N = 148 is not what these sizes are short of).
#include <cstdlib>
#include <type_traits>

#include "jk_m4_layout.h"
#include "jk_m4_walk.h"
#include "nbx_common.h"

#pragma clang diagnostic ignored "-Winline-asm"

namespace {

constexpr int MX_THREADS = 512;
typedef __attribute__((address_space(3))) void* mx_lds_vp;
typedef double mx_d2 __attribute__((ext_vector_type(2)));

__host__ __device__ __forceinline__ int mx_tri_row(int64_t T) {
    int64_t p = (int64_t)((sqrt(8.0 * (double)T + 1.0) - 1.0) * 0.5);
    while (p * (p + 1) / 2 > T) --p;
    while ((p + 1) * (p + 2) / 2 <= T) ++p;
    return (int)p;
}

// static loop: f(std::integral_constant<int, K>) for K = K0 .. K1 - 1
template <int K0, int K1, class F>
__device__ __forceinline__ void mx_for(F&& f) {
    if constexpr (K0 < K1) {
        f(std::integral_constant<int, K0>{});
        mx_for<K0 + 1, K1>(f);
    }
}

template <int CNT>
__device__ __forceinline__ void mx_wait_vm() {
    static_assert(CNT >= 0 && CNT <= 63, "vmcnt is a 6-bit counter");
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(CNT) : "memory");
}

// ---------------------------------------------------------------------------------------------- pack and weights
// slab rows [p0, p0 + np) of the dense tensor -> tiles of 4 x 4 blocks in chunk order; one workgroup per tile
// (nsrc <= N: the dense tensor's own size; the tiles p < nsrc exist, their elements beyond nsrc are stored as zeros)
template <int NB>
__global__ __launch_bounds__(256) void mx_pack_kernel(const double* __restrict__ eri, double* __restrict__ out, int p0,
                                                      int64_t t_begin, int nsrc, MxChunks ch) {
    using G = MxGeom<NB>;
    const int64_t T = t_begin + blockIdx.x;
    const int p = mx_tri_row(T), q = (int)(T - (int64_t)p * (p + 1) / 2);
    const double* src = eri + ((int64_t)(p - p0) * nsrc + q) * nsrc * nsrc;
    double* dst = out + (int64_t)blockIdx.x * G::TILE;
    for (int e = threadIdx.x; e < G::TILE; e += 256) {
        const int blk = e >> 4;
        int k = 0;
        while (k + 1 < ch.n && ch.c[k + 1].start <= blk) ++k;
        int bt, bc;
        mx_block_of(ch, k, blk - ch.c[k].start, bt, bc);
        const int kk = ((e >> 2) & 3) ^ ((bt ^ bc) & 3), i = (e & 3) ^ kk;  // (the swizzle: jk_m4.hip)
        const int row = 4 * bt + i, col = 4 * bc + kk;
        dst[e] = (col <= row && row < nsrc) ? src[(int64_t)row * nsrc + col] : 0.0;
    }
}

// Dtot' in the staging order: wt[(k LPTM + s) 256 + ptid] (double2) = the weights of the two doubles that loading
// thread ptid brings in with slot s of chunk k: Dtot[r][c] + Dtot[c][r] below the diagonal, Dtot[r][r] on it, 0
// elsewhere (the zeros of the diagonal blocks, the clamped tail of a chunk, the slots a short chunk does not use)
template <int NB>
__global__ __launch_bounds__(256) void mx_weights_kernel(const double* __restrict__ dm, int ndm, double* __restrict__ wt,
                                                         MxChunks ch) {
    using G = MxGeom<NB>;
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= G::NCH * G::LPTM * M4_PROD_THREADS) return;
    const int tid = i % M4_PROD_THREADS, s = (i / M4_PROD_THREADS) % G::LPTM, k = (i / M4_PROD_THREADS) / G::LPTM;
    const int d0 = (s * M4_PROD_THREADS + tid) * 2, dend = 16 * ch.c[k].blocks;
    const int64_t n2 = (int64_t)G::N * G::N;
    double out[2] = {0.0, 0.0};
    for (int e = 0; e < 2; ++e) {
        const int d = d0 + e;
        if (d >= dend) continue;
        int bt, bc;
        mx_block_of(ch, k, d >> 4, bt, bc);
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

// ---------------------------------------------------------------------------------------------- the walk of a BAND chunk
// Chunk K = the column groups [J0, J1) of band G (block rows 4 G .. 4 G + 3), row after row in the buffer.  With
// lane = 16 a + 4 b + c as in jk_m4_walk.h:
//   row part    item j in [J0, J1):  A = block (4 G + b, 4 j + w4) element (c, a),  B = X[4 (4 j + w4) + a][c] (bxr[j])
//               -> acc[G]: every lane's block row is in the chunk, and so is every block left of the diagonal group;
//   column part item H in [J0, J1):  A = block (4 G + w4, 4 H + b) element (a, c),  B = X[4 (4 G + w4) + a][c]
//               -> acc[H]: each walking wave has one of the band's four rows.
// Only the segment that holds the diagonal (J1 = G + 1) has anything to mask: its last group.
template <class G_, int K>
__device__ __forceinline__ void mx_walk_band(const double* __restrict__ buf, const double* __restrict__ xs, int w4,
                                             const M4Lane<G_::NG>& ln, double (&acc)[G_::NG], const double (&bxr)[G_::NG]) {
    constexpr MxChunk ch = G_::CH.c[K];
    constexpr int G = ch.ra / 4, J0 = ch.j0, J1 = ch.j1, NJ = J1 - J0;
    constexpr bool DIAG = J1 == G + 1;
    const int a = ln.a, b = ln.b, c = ln.c;
    const int rs_b = DIAG ? b * 4 * (G - J0) + m4_tri(b) : b * 4 * NJ;     // first block of this lane's row in the segment
    const int rs_w = DIAG ? w4 * 4 * (G - J0) + m4_tri(w4) : w4 * 4 * NJ;  // first block of this wave's row (uniform)
    const double* rp = buf + 16 * (rs_b + w4) + 4 * (a ^ b ^ w4) + (c ^ a);
    const double* cp = buf + 16 * rs_w + ln.col0 + 4 * (ln.cbx ^ w4);
    const double bt = xs[16 * (4 * G + w4) + ln.xlane];
    double avr[NJ], avc[NJ];
#pragma unroll
    for (int i = 0; i < NJ; ++i) avr[i] = rp[64 * i];
#pragma unroll
    for (int i = 0; i < NJ; ++i) avc[i] = cp[64 * i];
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int i = 0; i < NJ; ++i) {
        double vr = avr[i], vc = avc[i];
        if (DIAG && J0 + i == G) {
            vr = b >= w4 ? vr : 0.0;                          // C = 4 G + w4 <= T = 4 G + b
            vc = (b < w4 || (b == w4 && c < a)) ? vc : 0.0;   // block column 4 G + b left of T = 4 G + w4, or the strict lower part
        }
        acc[G] = __builtin_amdgcn_mfma_f64_4x4x4f64(vr, bxr[J0 + i], acc[G], 0, 0, 0);
        acc[J0 + i] = __builtin_amdgcn_mfma_f64_4x4x4f64(vc, bt, acc[J0 + i], 0, 0, 0);
    }
}

// ---------------------------------------------------------------------------------------------- the kernel
// kpart1[(w S + slot) NDM + x][N]: row-p partial of workgroup w for the slot-th row of its range;
// kpart2[(T - t_begin) NDM + x][N]: row-q partial of tile T (q < p; columns <= q); jfull (N, N): J[p][q] = J[q][p]
template <int NB, int NDM>
__global__ __launch_bounds__(MX_THREADS, 1) void jk_mx_kernel(const double* __restrict__ packed, const double* __restrict__ dm,
                                                              const double* __restrict__ wtab, double* __restrict__ jfull,
                                                              double* __restrict__ kpart1, double* __restrict__ kpart2,
                                                              int64_t t_begin, int64_t t_end, int L, int S) {
    using G_ = MxGeom<NB>;
    constexpr int N = G_::N, NG = G_::NG, NCH = G_::NCH, LPTM = G_::LPTM, BUF = G_::BUF, TILE = G_::TILE, PT = M4_PROD_THREADS;
    constexpr int NRES = NCH < 4 ? NCH : 4;  // chunks whose weights stay in registers
    extern __shared__ __attribute__((aligned(16))) double smem[];
    double* buf0 = smem;                     // [MX_RING][BUF] chunk buffers
    double* xs0 = smem + MX_RING * BUF;      // [2][N][4] X of the current / next tile
    // ONE buffer for the consumers' partial rows: the row-q halves (odd columns) of the tile just walked are written at the
    // tile's last step and summed by the producers during step 0 of the next tile; the row-p halves (even columns) of a row
    // that has ended wait in the consumers' registers until the end of that step 1 and are summed during step 2.  (Two
    // buffers were 32 KB at N = 256 -- the difference between five and six 4 KB loads per chunk.)
    double* redq = xs0 + 2 * 4 * N;          // [4][NG][32]
    double* redp = redq;
    double* jred = redq + 4 * NG * 32;       // [2][4] producers' J partials per tile parity
    double* jstage = jred + 16;              // [L] J of the tiles done, stored at the end of the range
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const bool producer = wave >= 4;
    const int ptid = tid - 256;  // producers: 0 .. 255

    const int64_t T0 = t_begin + (int64_t)blockIdx.x * L;
    const int64_t T_end = min(t_end, T0 + L);
    if (T0 >= T_end) return;  // uniform for the whole workgroup
    const int p_first = mx_tri_row(T0);
    const int ntile = (int)(T_end - T0);
    const double* tile0 = packed + (T0 - t_begin) * (int64_t)TILE;

    // Chunk c of tile tt goes from HBM straight into ring slot `slot`: global_load_lds_dwordx4, lane l of a wave lands
    // its 16 bytes at the instruction's LDS base + 16 l; slot s of the chunk is one instruction per producer wave
    // (thread ptid's two doubles of slot s at (s PT + ptid) 2).  A chunk is lpt(c) instructions per wave (the tail
    // re-reads the chunk's last 16 bytes; a chunk past the last tile re-reads the first tile: landed in a free slot,
    // never read).
    auto issue = [&](auto cc, int tt, int slot) {
        constexpr int c = decltype(cc)::value;
        constexpr int begin = 16 * G_::CH.c[c].start, end = begin + 16 * G_::CH.c[c].blocks, LP = G_::lpt(c);
        const double* tile = tile0 + (tt < ntile ? (int64_t)tt * TILE : 0);
        double* buf = buf0 + slot * BUF;
        int pt_ = ptid;  // (opaque per chunk: the clamped offsets are recomputed, not kept in registers)
        asm volatile("" : "+v"(pt_));
#pragma unroll
        for (int s = 0; s < LP; ++s) {
            int d = begin + (s * PT + pt_) * 2;
            d = min(d, end - 2);
            const unsigned off = 8u * (unsigned)d;
            const unsigned lds_a = (unsigned)(size_t)(mx_lds_vp)(buf + (s * PT + (wave - 4) * 64) * 2);
            asm volatile("s_mov_b32 m0, %2\n\tglobal_load_lds_dwordx4 %0, %1 nt" : : "v"(off), "s"(tile), "s"(lds_a) : "memory", "m0");
        }
    };
    // X of a tile: xs[n][c] = D^{c / 2}[c & 1 ? p : q][n]  (NDM = 1: columns 2, 3 are zero); by the CONSUMER waves
    constexpr int XU = (4 * N + PT - 1) / PT;
    auto fetch_x = [&](int pp, int qq, double (&v)[XU]) {
#pragma unroll
        for (int u = 0; u < XU; ++u) {
            const int e = tid + PT * u;  // element e = 4 n + c
            const int n = e >> 2, cc = e & 3, x = cc >> 1;
            const int off = x * N * N + ((cc & 1) ? pp : qq) * N + n;
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
    // the consumers' partial rows of a finished tile: summed over the four consumer waves, in wave order; `last`: the
    // last column that is stored (K is symmetric: jk_sym_reduce_kernel(k_lower) mirrors the sums)
    auto reduce_rows = [&](const double* red, int parity, double* dst, int last) {  // dst[x N + row]
        for (int e = ptid; e < NG * 32; e += PT) {
            const int g = e >> 5, l = 2 * (e & 31) + parity;
            const int row = 16 * g + 4 * ((l >> 2) & 3) + (l >> 4), x = (l & 3) >> 1;
            if (row <= last && x < NDM) dst[x * N + row] = (red[e] + red[NG * 32 + e]) + (red[2 * NG * 32 + e] + red[3 * NG * 32 + e]);
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
        mx_d2 wres[NRES][LPTM];  // Dtot' of chunks 0 .. NRES - 1 (the same for every tile)
        double jacc = 0.0;
#pragma unroll
        for (int k = 0; k < NRES; ++k)
#pragma unroll
            for (int s = 0; s < LPTM; ++s)
                wres[k][s] = *reinterpret_cast<const mx_d2*>(wtab + 2 * ((k * LPTM + s) * PT + ptid));
        if (ptid < 8) jred[ptid] = 0.0;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // (the table: nothing of the compiler's in the counter from here)
        int islot = 0;  // ring slot of the next chunk to be issued
        mx_for<0, 4>([&](auto cc) {
            constexpr int c = decltype(cc)::value;
            issue(std::integral_constant<int, c % NCH>{}, c / NCH, islot);
            islot = islot + 1 == MX_RING ? 0 : islot + 1;
        });
        {  // my part of chunk 0
            constexpr int cnt = G_::lpt(1 % NCH) + G_::lpt(2 % NCH) + G_::lpt(3 % NCH);
            mx_wait_vm<cnt>();
        }
        __syncthreads();
        int jslot = 0;  // ring slot of the chunk the consumers walk at this step
        const unsigned wvoff = 16u * (unsigned)ptid;
        for (int t = 0; t < ntile; ++t) {
            const int64_t T = T0 + t;
            int pn = p, qn = q;
            next_pq(pn, qn);
            mx_d2 wstr[4][LPTM];  // Dtot' of a chunk c >= 4, set c & 3: loaded at step c - 4, used at step c
            mx_for<0, NCH>([&](auto kc) {
                constexpr int k = decltype(kc)::value;
                // step (t, k): the consumers walk chunk k; the chunk four ahead goes into the slot they left at the last
                // barrier, chunk k + 1 has landed (waited for at the end of this step)
                issue(std::integral_constant<int, (k + 4) % NCH>{}, t + (k + 4) / NCH, islot);  // (first: the stream is what the kernel is bound by)
                islot = islot + 1 == MX_RING ? 0 : islot + 1;
                if (k == 0 && t > 0) {
                    // the consumers' rows of tile t - 1 (written at its last step, behind that step's barrier), and its J
                    int pp = p, qq = q - 1;
                    if (qq < 0) {
                        pp = p - 1;
                        qq = pp;
                    }
                    if (qq < pp) reduce_rows(redq, 1, kpart2 + ((T - 1 - t_begin) * NDM) * (int64_t)N, qq);
                    if (ptid == 0) {
                        const double* jr = jred + ((t - 1) & 1) * 4;
                        jstage[t - 1] = (jr[0] + jr[1]) + (jr[2] + jr[3]);
                    }
                }
                if (k == 2 && t > 0) {  // the row-p halves of a row that ended with tile t - 1 (in the buffer since step 1)
                    const int pp = q == 0 ? p - 1 : p;
                    if (pp != p) reduce_rows(redp, 0, kpart1 + (((int64_t)blockIdx.x * S + (pp - p_first)) * NDM) * N, pp);
                }
                // the J contribution of chunk k (the one being walked: it landed a step ago)
                {
                    const double* buf = buf0 + jslot * BUF;
                    jslot = jslot + 1 == MX_RING ? 0 : jslot + 1;
                    constexpr int LP = G_::lpt(k);
#pragma unroll
                    for (int s = 0; s < LP; ++s) {
                        const double2 v = *reinterpret_cast<const double2*>(buf + (s * PT + ptid) * 2);
                        mx_d2 w;
                        if constexpr (k < NRES) w = wres[k][s];
                        else w = wstr[k & 3][s];
                        jacc = fma(v.y, w.y, fma(v.x, w.x, jacc));
                    }
                    asm volatile("" : "+v"(jacc));  // (here: left alone, the compiler sinks a tile's whole chain of FMAs to
                                                    //  the tile's end and keeps every chunk's read-back alive until then)
                }
                if (k == NCH - 1) {  // this wave's share of tile t's J
                    jacc = nbx_wave_sum_dpp(jacc);  // (lane moves, not six ds_bpermute round trips: jk_m8.hip)
                    if (lane == 0) jred[(t & 1) * 4 + (wave - 4)] = jacc;
                    jacc = 0.0;
                }
                // the weights of chunk k + 4 of this tile into the set chunk k has just been done with
                if constexpr (k + 4 < NCH) {
                    constexpr int c = k + 4, LP = G_::lpt(c);
                    // (one scalar base -- the kernel argument -- and the slot in the offset register, made afresh per step:
                    //  a base per (chunk, slot) is hoisted out of the tile loop, seventy pairs of scalar registers that end
                    //  up spilled into VGPR lanes, and a v_readlane right in front of an inline-asm load is a VALU write
                    //  of an SGPR that the load reads too early -- the hazard check does not look into asm statements)
                    unsigned wv = wvoff;
                    asm volatile("" : "+v"(wv));
#pragma unroll
                    for (int s = 0; s < LP; ++s)
                        asm volatile("global_load_dwordx4 %0, %1, %2" : "=v"(wstr[c & 3][s]) : "v"(wv + (unsigned)((c * LPTM + s) * PT * 16)), "s"(wtab) : "memory");
                }
                // my part of chunk k + 1 (and, for k + 1 >= 4, its weights): everything issued after them may still be
                // in flight -- the chunks k + 2 .. k + 4 and the weights of those among them that belong to this tile
                {
                    constexpr int cnt = [] {
                        int n = 0;
                        for (int i = 2; i <= 4; ++i) {
                            n += G_::lpt((k + i) % NCH);
                            if (k + i >= 4 && k + i < NCH) n += G_::lpt(k + i);
                        }
                        return n;
                    }();
#ifndef MX_WSLACK
#define MX_WSLACK 0
#endif
                    mx_wait_vm<(cnt > MX_WSLACK ? cnt - MX_WSLACK : 0)>();
                    if constexpr (k + 1 >= 4 && k + 1 < NCH) {
                        constexpr int LP1 = G_::lpt(k + 1);
#pragma unroll
                        for (int s = 0; s < LP1; ++s) asm volatile("" : "+v"(wstr[(k + 1) & 3][s]));  // (landed: not to be read above this line)
                    }
                }
                __syncthreads();
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
        if (qq < pp) reduce_rows(redq, 1, kpart2 + ((T_end - 1 - t_begin) * NDM) * (int64_t)N, qq);
        __syncthreads();  // (the consumers now put the last row's row-p halves into the buffer)
        __syncthreads();
        reduce_rows(redp, 0, kpart1 + (((int64_t)blockIdx.x * S + (pp - p_first)) * NDM) * N, pp);
        if (wave == 4) {  // J of every tile of the range: lane i stores tile i's (one wave: its LDS operations are in order)
            if (lane == 0) {
                const double* jr = jred + ((ntile - 1) & 1) * 4;
                jstage[ntile - 1] = (jr[0] + jr[1]) + (jr[2] + jr[3]);
            }
            for (int i = lane; i < ntile; i += 64) {
                const int64_t Ti = T0 + i;
                const int pi = mx_tri_row(Ti), qi = (int)(Ti - (int64_t)pi * (pi + 1) / 2);
                const double j = jstage[i];
                jfull[(int64_t)pi * N + qi] = j;
                jfull[(int64_t)qi * N + pi] = j;
            }
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
        {
            double v[XU];
            fetch_x(p, q, v);
            store_x(xs0, v);
        }
        __syncthreads();
        int slot = 0;  // ring slot of the chunk being walked
        double accsave[NG];  // row-p halves of a row that has ended, on their way to the buffer
        bool pend_p = false;
#pragma unroll
        for (int g = 0; g < NG; ++g) accsave[g] = 0.0;
        for (int t = 0; t < ntile; ++t) {
            int pn = p, qn = q;
            next_pq(pn, qn);
            const bool more = t + 1 < ntile;
            const bool row_ends = pn != p || !more;
            const double* xs = xs0 + (t & 1) * 4 * N;
            double xv[XU];
            fetch_x(more ? pn : p, more ? qn : q, xv);  // X of tile t + 1: fetched now, stored a step on
            mx_for<0, NCH>([&](auto kc) {
                constexpr int k = decltype(kc)::value;
                const double* buf = buf0 + slot * BUF;
                slot = slot + 1 == MX_RING ? 0 : slot + 1;
                if constexpr (G_::CH.c[k].band) mx_walk_band<G_, k>(buf, xs, wave, ln, acc, bxr);
                else m4_walk_chunk<G_, k>(buf, xs, wave, ln, acc, bxr);
                if (k == 1 && more) store_x(xs0 + ((t + 1) & 1) * 4 * N, xv);
                if (k == 1 && pend_p) {  // (uniform) the producers are done with the row-q halves that were in the buffer
                    if (!(lane & 1)) {
#pragma unroll
                        for (int g = 0; g < NG; ++g) redp[(wave * NG + g) * 32 + (lane >> 1)] = accsave[g];
                    }
                    pend_p = false;
                }
                if (k == NCH - 1) {
                    // end of tile: the row-q halves (odd columns: they used D[p][:]) leave the registers; the row-p halves
                    // (even columns) stay until the row changes, then wait in accsave for the buffer to be free.
                    const bool odd = lane & 1;
#pragma unroll
                    for (int g = 0; g < NG; ++g) {
                        if (odd) redq[(wave * NG + g) * 32 + (lane >> 1)] = acc[g];
                        else if (row_ends) accsave[g] = acc[g];
                        acc[g] = (odd || row_ends) ? 0.0 : acc[g];
                    }
                    pend_p = row_ends;
                }
                __builtin_amdgcn_sched_barrier(0);  // (the MFMAs stay above the barrier: jk_m4.hip)
                __syncthreads();
            });
            p = pn;
            q = qn;
        }
        // the last tile always ends its row: its row-p halves go into the buffer when the producers have read the row-q ones
        __syncthreads();
        if (!(lane & 1)) {
#pragma unroll
            for (int g = 0; g < NG; ++g) redp[(wave * NG + g) * 32 + (lane >> 1)] = accsave[g];
        }
        __syncthreads();
    }
}

size_t mx_align256(size_t x) { return (x + 255) & ~(size_t)255; }

struct MxPlan {
    int wgs, L, S;
    size_t wt_off, k1_off, k2_off, total, lds_bytes;
};

template <int NB>
MxPlan mx_plan_nb(int64_t p0, int64_t np, int64_t ndm) {
    using G = MxGeom<NB>;
    MxPlan pl;
    const int64_t ntiles = m4_tri((int)(p0 + np)) - m4_tri((int)p0);
    int64_t L = nbx_cdiv(ntiles, (int64_t)MX_CUS);
    if (L < 1) L = 1;
    pl.L = (int)L;
    pl.wgs = (int)nbx_cdiv(ntiles, L);
    pl.S = (int)sqrt(2.0 * (double)L) + 3;
    pl.lds_bytes = (size_t)(MX_RING * G::BUF + 2 * 4 * G::N + 4 * G::NG * 32 + 16 + pl.L) * sizeof(double);
    size_t off = 0;
    pl.wt_off = off; off += mx_align256((size_t)(G::NCH * G::LPTM * M4_PROD_THREADS * 2) * sizeof(double));
    pl.k1_off = off; off += mx_align256((size_t)((int64_t)pl.wgs * pl.S * ndm * G::N) * sizeof(double));
    pl.k2_off = off; off += mx_align256((size_t)(ntiles * ndm * G::N) * sizeof(double));
    pl.total = off;
    return pl;
}

template <int NB>
int mx_run(nbx_ctx* ctx, int64_t p0, int64_t p1, const double* d_packed, const double* d_dm, int64_t ndm, double* d_jk,
           void* d_work, const double* d_hv, double* d_fock, double* d_vhf) {
    using G = MxGeom<NB>;
    const int64_t np = p1 - p0, N = G::N, n2 = N * N;
    const MxPlan pl = mx_plan_nb<NB>(p0, np, ndm);
    static_assert((MX_RING * G::BUF + G::FIXED) * 8 <= MX_LDS_BYTES, "LDS of a workgroup");
    if (pl.lds_bytes > (size_t)MX_LDS_BYTES) {
        nbx_set_error("nbx_jk_mx: %zu bytes of LDS for N = %lld", pl.lds_bytes, (long long)N);
        return NBX_E_UNSUPPORTED;
    }
    char* base = static_cast<char*>(d_work);
    double* wt = reinterpret_cast<double*>(base + pl.wt_off);
    double* k1 = reinterpret_cast<double*>(base + pl.k1_off);
    double* k2 = reinterpret_cast<double*>(base + pl.k2_off);
    if (np < N) {  // J entries this slab does not own must read as zero
        const int rc = nbx_memset(ctx, d_jk, 0, (size_t)n2 * sizeof(double));
        if (rc != NBX_OK) return rc;
    }
    hipLaunchKernelGGL(mx_weights_kernel<NB>, dim3((unsigned)nbx_cdiv(G::NCH * G::LPTM * M4_PROD_THREADS, 256)), dim3(256), 0,
                       ctx->stream, d_dm, (int)ndm, wt, G::CH);
    NBX_LAUNCH_CHECK();
    const int64_t t_begin = m4_tri((int)p0), t_end = m4_tri((int)p1);
    {
        nbx_prof_scope prof(ctx, NBX_PROF_JK_DENSE);
        static bool attr_set = false;
        if (!attr_set) {
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&jk_mx_kernel<NB, 1>),
                                      hipFuncAttributeMaxDynamicSharedMemorySize, MX_LDS_BYTES);
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&jk_mx_kernel<NB, 2>),
                                      hipFuncAttributeMaxDynamicSharedMemorySize, MX_LDS_BYTES);
            attr_set = true;
        }
        if (ndm == 2)
            hipLaunchKernelGGL((jk_mx_kernel<NB, 2>), dim3((unsigned)pl.wgs), dim3(MX_THREADS), pl.lds_bytes, ctx->stream,
                               d_packed, d_dm, wt, d_jk, k1, k2, t_begin, t_end, pl.L, pl.S);
        else
            hipLaunchKernelGGL((jk_mx_kernel<NB, 1>), dim3((unsigned)pl.wgs), dim3(MX_THREADS), pl.lds_bytes, ctx->stream,
                               d_packed, d_dm, wt, d_jk, k1, k2, t_begin, t_end, pl.L, pl.S);
    }
    NBX_LAUNCH_CHECK();
    return nbx_jk_sym_reduce(ctx, k1, k2, d_jk + n2, N, p0, np, ndm, t_begin, pl.L, pl.S, d_jk, d_hv, d_fock, d_vhf, 1, 1);
}

}  // namespace

// The sizes this kernel has an instance for (N = 4 NB): every multiple of eight from 152 to 256, then 272 and every
// multiple of sixteen up to 400 (the sizes between run as the next one within eight, zero-padded; dense + packed
// tensor fit one GPU's HBM up to N = 400).  The instances are compiled in two translation units (this file for
// N <= 256, jk_mx_hi.hip -- which includes this one with another list -- for the rest) so that they build side by side.
// NBX_JK_MX=0 in the environment (read once per process) hands the sizes back to jk_s4.hip / jk_sym.hip.
#ifndef NBX_MX_SIZES
#define NBX_MX_SIZES(X) X(38) X(40) X(42) X(44) X(46) X(48) X(50) X(52) X(54) X(56) X(58) X(60) X(62) X(64)
#define MX_FN(name) name
#define MX_HAS_HI 1
#endif
#define MX_DISPATCH(N_, EXPR)            \
    switch ((int)((N_) / 4)) {           \
        NBX_MX_SIZES(MX_CASE_##EXPR)     \
        default: break;                  \
    }
#ifdef MX_HAS_HI
bool nbx_jk_mx_covers_hi(int64_t N);
size_t nbx_jk_mx_packed_bytes_hi(int64_t N, int64_t p0, int64_t p1);
size_t nbx_jk_mx_worksize_hi(int64_t N, int64_t p0, int64_t p1, int64_t ndm);
int nbx_jk_mx_pack_hi(nbx_ctx* ctx, int64_t N, int64_t nsrc, int64_t p0, int64_t p1, const double* d_eri, double* d_packed);
int nbx_jk_mx_hi(nbx_ctx* ctx, int64_t N, int64_t p0, int64_t p1, const double* d_packed, const double* d_dm, int64_t ndm,
                 double* d_jk, void* d_work, const double* d_hv, double* d_fock, double* d_vhf);
#define MX_ELSE_HI(call) return call;
#else
#define MX_ELSE_HI(call)
#endif

bool MX_FN(nbx_jk_mx_covers)(int64_t N) {
    static const bool on = getenv("NBX_JK_MX") == nullptr || atoi(getenv("NBX_JK_MX")) != 0;
    if (!on || N % 4 != 0) return false;
#define MX_CASE_covers(NB_) case NB_: return true;
    MX_DISPATCH(N, covers)
#undef MX_CASE_covers
    MX_ELSE_HI(nbx_jk_mx_covers_hi(N))
    return false;
}

#ifdef MX_HAS_HI
// the covered size that N runs as (N itself, or the next instance: extra rows and columns zero); 0: none
int64_t nbx_jk_mx_padded(int64_t N) {
    for (int64_t n = (N + 3) / 4 * 4; n <= N + 8; n += 4)
        if (nbx_jk_mx_covers(n)) return n;
    return 0;
}
#endif

size_t MX_FN(nbx_jk_mx_packed_bytes)(int64_t N, int64_t p0, int64_t p1) {
    const int64_t ntiles = m4_tri((int)p1) - m4_tri((int)p0);
#define MX_CASE_bytes(NB_) case NB_: return (size_t)(ntiles * MxGeom<NB_>::TILE) * sizeof(double) + 256;
    MX_DISPATCH(N, bytes)
#undef MX_CASE_bytes
    MX_ELSE_HI(nbx_jk_mx_packed_bytes_hi(N, p0, p1))
    return 0;
}

size_t MX_FN(nbx_jk_mx_worksize)(int64_t N, int64_t p0, int64_t p1, int64_t ndm) {
#define MX_CASE_work(NB_) case NB_: return mx_plan_nb<NB_>(p0, p1 - p0, ndm).total;
    MX_DISPATCH(N, work)
#undef MX_CASE_work
    MX_ELSE_HI(nbx_jk_mx_worksize_hi(N, p0, p1, ndm))
    return 0;
}

int MX_FN(nbx_jk_mx_pack)(nbx_ctx* ctx, int64_t N, int64_t nsrc, int64_t p0, int64_t p1, const double* d_eri, double* d_packed) {
    NBX_CHECK_ARG(MX_FN(nbx_jk_mx_covers)(N) && d_eri && d_packed && nsrc <= N && nsrc > N - 12 && p1 <= nsrc);
    const int64_t ntiles = m4_tri((int)p1) - m4_tri((int)p0);
#define MX_CASE_pack(NB_)                                                                                                  \
    case NB_:                                                                                                              \
        hipLaunchKernelGGL(mx_pack_kernel<NB_>, dim3((unsigned)ntiles), dim3(256), 0, ctx->stream, d_eri, d_packed, (int)p0, \
                           (int64_t)m4_tri((int)p0), (int)nsrc, MxGeom<NB_>::CH);                                          \
        NBX_LAUNCH_CHECK();                                                                                                \
        return NBX_OK;
    MX_DISPATCH(N, pack)
#undef MX_CASE_pack
    MX_ELSE_HI(nbx_jk_mx_pack_hi(ctx, N, nsrc, p0, p1, d_eri, d_packed))
    return NBX_E_UNSUPPORTED;
}

int MX_FN(nbx_jk_mx)(nbx_ctx* ctx, int64_t N, int64_t p0, int64_t p1, const double* d_packed, const double* d_dm, int64_t ndm,
                     double* d_jk, void* d_work, const double* d_hv, double* d_fock, double* d_vhf) {
    NBX_CHECK_ARG(MX_FN(nbx_jk_mx_covers)(N));
#define MX_CASE_run(NB_) case NB_: return mx_run<NB_>(ctx, p0, p1, d_packed, d_dm, ndm, d_jk, d_work, d_hv, d_fock, d_vhf);
    MX_DISPATCH(N, run)
#undef MX_CASE_run
    MX_ELSE_HI(nbx_jk_mx_hi(ctx, N, p0, p1, d_packed, d_dm, ndm, d_jk, d_work, d_hv, d_fock, d_vhf))
    return NBX_E_UNSUPPORTED;
}
