// libnbx: J/K contraction on 4-fold packed integrals (include/nbx.h "J/K contraction, packed form").
//
//   (pq|rs) = (qp|rs) = (pq|sr): only the pairs q <= p AND s <= r are stored and read -- a quarter
//   of the dense tensor, half of what jk_sym.hip reads.  (PySCF keeps `mf._eri` 8-fold packed and
//   libcvhf, which the reference calls through get_veff, contracts it in that form.)
//
// A tile is the lower triangle L[a][b], b <= a, of the (r,s) matrix of one pair (p,q); it feeds
//       J_pq = J_qp  = sum_{ab} Lsym_ab Dtot_ab
//       K^x_p[t]    += sum_c Lsym[t][c] D^x_q[c]     K^x_q[t] += sum_c Lsym[t][c] D^x_p[c]   (q < p)
// i.e. two symmetric matrix x vector products per spin with the matrix stored once.  Thread t owns
// output element t and walks the whole symmetric row t (the row part L[t][c], c <= t, and the
// column part L[c][t], c > t): every thread does N steps, nothing is reduced across threads, the
// density values of a step are wave-uniform (scalar loads, SGPR operands).
//
// Blocked triangle: the walk needs a tile resident in LDS, and a whole tile (N = 148: 88 KB)
// leaves no room to stream the next one.  The index range is cut into NB blocks of s = N / NB
// (one wave per block); a tile is stored as NB chunks,
//       chunk 0      : the NB diagonal triangles            (wave w walks its own triangle)
//       chunk r >= 1 : the NB/2 rectangles (I, J = I ^ r)   (wave w walks the rectangle it shares
//                      with wave w ^ r: one as the row side, the other as the column side)
// -- the rounds of a round-robin tournament: in every chunk every wave does exactly s steps.
// Chunks (N = 148: 22 KB) are double buffered in LDS; the next tile streams into registers with
// non-temporal 16-byte loads while the current one is walked.  Rectangle rows have an odd stride
// (s | 1) in the stored format so that the row-side walk is LDS-bank-conflict free.
//
// Work distribution, partial buffers and the final reduction are those of jk_sym.hip: persistent
// workgroups own equal contiguous ranges of the tile sequence T(p,q) = p(p+1)/2 + q; everything
// is summed in a fixed order (bitwise reproducible).  The slab interface is additive.
#include <cstdlib>

#include "nbx_common.h"
#include "jk_s4_layout.h"
#include "jk_s4_device.h"

// jk_p8.hip: the 8-fold form (truncated tiles) that serves whole tensors of the NB = 4 / six-loads sizes
bool nbx_jk_p8_covers(int64_t N, int64_t p0, int64_t p1);
size_t nbx_jk_p8_packed_bytes(int64_t N);
size_t nbx_jk_p8_worksize(int64_t N, int64_t ndm);
int nbx_jk_p8_pack(nbx_ctx* ctx, int64_t N, const double* d_eri, double* d_packed);
int nbx_jk_p8(nbx_ctx* ctx, int64_t N, const double* d_packed, const double* d_dm, int64_t ndm, double* d_jk, void* d_work,
              const double* d_hv, double* d_fock, double* d_vhf, const double* d_dts_in);

// jk_s8.hip: the eight-wave, matrix-pipe form of the NB = 4 / six-loads instance
bool nbx_jk_s8_covers(int64_t N);
void nbx_jk_s8_plan(int64_t ntiles, int* wgs, int* L, int* S);
int nbx_jk_s8_launch(nbx_ctx* ctx, int64_t N, int64_t p0, int64_t np, int64_t ndm, const double* d_packed,
                     const double* d_dm, const double* d_dts, double* d_j, double* k1, double* k2, int64_t t_begin,
                     int64_t t_end, int wgs, int L, int S);

// jk_s4d.hip: the NB = 4 / six-loads instance with the tiles streamed straight into LDS
bool nbx_jk_s4d_covers(int NB, int lpt);
size_t nbx_jk_s4d_lds_bytes(int NB, int lpt, int variant);
int nbx_jk_s4d_per_cu(int variant);
int nbx_jk_s4d_launch(nbx_ctx* ctx, int variant, int64_t N, int64_t p0, int64_t np, int64_t ndm, int lpt,
                      const double* d_packed, const double* d_dm, const double* d_dts, double* d_j, double* k1, double* k2,
                      int64_t t_begin, int64_t t_end, int wgs, int L, int S);

namespace {

constexpr int S4_CUS = 256;
constexpr size_t S4_LDS_PER_CU = 160 * 1024;

// dense slab rows [p0,p1) -> packed tiles T(p,q), q <= p (the buffer was zeroed: pads stay 0)
// N: size of the source tensor; NP >= N: size whose tile geometry is used (entries a >= N stay zero)
__global__ __launch_bounds__(256) void s4_pack_kernel(const double* __restrict__ eri, double* __restrict__ out, int N,
                                                      int NP, int NB, int p0, int64_t t_begin) {
    const S4Geom g = s4_geom(NP, NB);
    const int64_t T = t_begin + blockIdx.x;
    const int p = s4_tri_row(T), q = (int)(T - s4_tri(p));
    const double* src = eri + ((int64_t)(p - p0) * N + q) * (int64_t)N * N;
    double* dst = out + (int64_t)blockIdx.x * g.M;
    for (int a = 0; a < N; ++a)
        for (int b = threadIdx.x; b <= a; b += blockDim.x) dst[s4_flat(g, a, b)] = src[(int64_t)a * N + b];
}

// (nmat, N, N) -> (nmat, NP, NP) with zero rows/columns appended, and back
__global__ void s4_pad_square_kernel(const double* __restrict__ src, double* __restrict__ dst, int N, int NP, int nmat) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (int64_t)nmat * NP * NP) return;
    const int b = (int)(i % NP), a = (int)((i / NP) % NP), m = (int)(i / ((int64_t)NP * NP));
    dst[i] = (a < N && b < N) ? src[((int64_t)m * N + a) * N + b] : 0.0;
}
__global__ void s4_crop_square_kernel(const double* __restrict__ src, double* __restrict__ dst, int N, int NP, int nmat) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (int64_t)nmat * N * N) return;
    const int b = (int)(i % N), a = (int)((i / N) % N), m = (int)(i / ((int64_t)N * N));
    dst[i] = src[((int64_t)m * NP + a) * NP + b];
}

// Dtot' in the order the staging slots hold the tile: dts[ch][w][k][lane] = the pair of
//   sum_x (D^x_ab + D^x_ba) (a != b), sum_x D^x_aa   for the two entries the lane stages,
// 0 for pads, for the repeated lanes of a pulled-back slot and for empty slots -- the J sum then
// needs no masks.  One thread per (slot, lane).
__global__ __launch_bounds__(256) void s4_dtot_kernel(const double* __restrict__ dm, double* __restrict__ dts, int N,
                                                      int NB, int lpt, int ndm) {
    const S4Geom g = s4_geom(N, NB);
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= NB * NB * lpt * 64) return;
    const int lane = i & 63, k = (i >> 6) % lpt, w = ((i >> 6) / lpt) % NB, ch = (i >> 6) / (lpt * NB);
    const int ne = ch == 0 ? g.E0 : g.Er;
    const int ps = s4_slot_start(ne, lpt, w, k);
    const int64_t n2 = (int64_t)N * N;
    double out[2] = {0.0, 0.0};
    if (ps >= 0 && ps + lane >= 64 * (lpt * w + k)) {
        for (int e = 0; e < 2; ++e) {
            int a, b;
            if (!s4_unflat(g, ch, 2 * (ps + lane) + e, a, b)) continue;
            // (sum_x D_ab) + (sum_x D_ba): the association nbx_huz_cycle_scalars_dts uses too, so
            // that a build gives the same bits whichever kernel prepared its table
            double v = 0.0, vt = 0.0;
            for (int x = 0; x < ndm; ++x) {
                v += dm[x * n2 + (int64_t)a * N + b];
                vt += dm[x * n2 + (int64_t)b * N + a];
            }
            out[e] = a == b ? v : v + vt;
        }
    }
    *reinterpret_cast<double2*>(dts + 2 * (int64_t)i) = make_double2(out[0], out[1]);
}

// EXPERIMENTAL (compiled with -DNBX_S4_MFMA_WALK only; measured slower):
// the same walk on the matrix pipe.  One chunk asks of wave w the product
//       out[t][n] += sum_c M[t][c] X[c][n],   t, c in [0, s),
// M = the symmetric s x s matrix this wave sees in the chunk (its triangle, or its side of the
// rectangle it shares with block u), X[c] = (D^a_q, D^a_p, D^b_q, D^b_p)[u s + c]: a skinny GEMM.
// v_mfma_f64_4x4x4_4b_f64 multiplies four independent 4 x 4 blocks per instruction,
//       D_b[i][j] += sum_k A_b[i][k] B_b[k][j],   operand lanes: A_b[i][k] at 16 k + 4 b + i,
//       B_b[k][j] at 16 k + 4 b + j, D_b[i][j] at 16 i + 4 b + j   (measured: scratch/probe/mfma444.hip),
// i.e. a 16 x 4 block of M against a 4 x 4 block of X: lane (m = lane & 15, k = lane >> 4) reads
// M[16 i + m][4 j + k] straight from the LDS-resident chunk, whatever the stored orientation (row
// side, column side, packed triangle: the address arithmetic differs, nothing else), and X[4 j + k]
// [lane & 3] as B -- the 2 NDM columns of X fill the instruction (the 16x16x4 form would leave 12 of
// its 16 columns idle and costs 3.5x the matrix-pipe time per instruction).  ceil(s/16) * ceil(s/4)
// MFMAs and as many LDS reads replace s steps of ~10 VALU/scalar instructions each: the walk was
// instruction-issue bound.  acc[i] of lane l holds row 16 i + 4 ((l >> 2) & 3)
// + (l >> 4), column l & 3: K_p contributions in the even columns (kept while p is fixed), K_q ones
// in the odd columns.
// Rows >= s of the last strip and columns >= s of the last k-step: addresses are clamped to stay
// on initialised LDS, the products land in rows nobody stores / meet a zero in X.
// 16-row strips a block can have, from the kernel instance's loads per thread per chunk (which
// bound the chunk size, hence s: LPT 2 -> s <= 22, 6 -> s <= 39, 10 and 17 -> s <= 64)
constexpr int s4_strips(int lpt) { return lpt <= 2 ? 2 : lpt <= 6 ? 3 : 4; }

template <int KIND, int NSTR, int XW>
__device__ __forceinline__ void s4_walk_mfma(const double* lb, int ls, int s, const double* xs, int fr, int fk,
                                             double (&acc)[NSTR]) {
    // xs: this block's rows of the LDS-resident X table, XW doubles per row ([c][n]); lane reads
    // column fr (folded onto the XW that exist: the surplus columns of the product are not used).
    // One strip at a time (one A register, one address in flight: the kernel has no VGPRs to
    // spare, and a spill reload waits for every streaming load in flight); the operands of step
    // j + 1 are read from LDS while the MFMA of step j runs.
    const int nk = (s + 3) >> 2;
    const int xcol = fr & (XW - 1);
    // column of step 0 for this lane, clamped copies for the last (partial) step
    const int c_last = min(4 * (nk - 1) + fk, s - 1);
    const bool last_ok = 4 * (nk - 1) + fk < s;
#pragma unroll
    for (int i = 0; i < NSTR; ++i) {
        const int tt = 16 * i + fr;  // rows >= s: whatever LDS holds there lands in rows nobody stores
        auto aoff = [&](int cc) {
            if (KIND == 0) {
                const int hi = max(tt, cc), lo = min(tt, cc);
                return ((hi * (hi + 1)) >> 1) + lo;
            }
            return KIND == 1 ? tt * ls + cc : cc * ls + tt;
        };
        double a0 = lb[aoff(fk)], b0 = xs[fk * XW + xcol], a1, b1;
        double d = acc[i];
        int j = 0;
        for (; j + 2 < nk; j += 2) {  // full steps only (the last one is peeled below)
            const int c1 = 4 * (j + 1) + fk;
            a1 = lb[aoff(c1)];
            b1 = xs[c1 * XW + xcol];
            d = __builtin_amdgcn_mfma_f64_4x4x4f64(a0, b0, d, 0, 0, 0);
            const int c2 = 4 * (j + 2) + fk;
            const bool fin = j + 3 == nk;  // step j + 2 is the last one: clamp / zero
            const int c2c = fin ? c_last : c2;
            a0 = lb[aoff(c2c)];
            b0 = xs[c2c * XW + xcol];
            if (fin && !last_ok) b0 = 0.0;
            d = __builtin_amdgcn_mfma_f64_4x4x4f64(a1, b1, d, 0, 0, 0);
        }
        if (j + 2 == nk) {  // two steps left: a0/b0 hold step j (full), the last one is step j + 1
            a1 = lb[aoff(c_last)];
            b1 = last_ok ? xs[c_last * XW + xcol] : 0.0;
            d = __builtin_amdgcn_mfma_f64_4x4x4f64(a0, b0, d, 0, 0, 0);
            d = __builtin_amdgcn_mfma_f64_4x4x4f64(a1, b1, d, 0, 0, 0);
        } else {  // one step left: a0/b0 hold it (already clamped / zeroed above, or nk == 1)
            if (nk == 1 && !last_ok) b0 = 0.0;
            d = __builtin_amdgcn_mfma_f64_4x4x4f64(a0, b0, d, 0, 0, 0);
        }
        acc[i] = d;
    }
}

// NB waves, LPT 16-byte loads per thread per chunk, PD chunks of prefetch distance (PD == NB: one
// whole tile ahead), DT_REG: the thread's Dtot' entries (the same for every tile) live in registers.
template <int NDM, int NB, int LPT, int PD, bool DT_REG, int WV>
__global__ __launch_bounds__(NB * 64) __attribute__((amdgpu_waves_per_eu(WV, WV))) void jk_s4_kernel(const double* __restrict__ eri, const double* __restrict__ dm,
                                                        const double* __restrict__ dts, double* __restrict__ jfull,
                                                        double* __restrict__ kpart1, double* __restrict__ kpart2,
                                                        int N, int p0, int np, int64_t t_begin, int64_t t_end, int L,
                                                        int S) {
    constexpr int NCH = NB, BUFD = LPT * NB * 128;
    static_assert(PD == NCH || PD == 2 || PD == 1, "prefetch distance");
    extern __shared__ __attribute__((aligned(16))) double smem[];  // buf[2][BUFD] | slack[128] | jred[2][NB] | xtab[2][N][XW]
    double* slack = smem + 2 * BUFD;
    double* jred = slack + 128;
    // X table of the MFMA walk: xtab[tile parity][c][n] = (D^0_q, D^0_p, D^1_q, D^1_p)[c] (NDM = 1: two
    // columns), written for tile T + 1 at the end of tile T from values fetched at its start
#ifdef NBX_S4_MFMA_WALK
    constexpr int XW = 2 * NDM, XR = XW;  // N <= 64 NB (one wave walks a block): XW N <= XW threads-per-workgroup
    double* xtab = jred + 2 * NB + ((2 * NB) & 1);
#endif

    int64_t T = t_begin + (int64_t)blockIdx.x * L;
    const int64_t T_end = min(t_end, T + L);
    if (T >= T_end) return;  // uniform for the whole workgroup
    int p = s4_tri_row(T);
    int q = (int)(T - s4_tri(p));
    const int p_first = p;

    const S4Geom g = s4_geom(N, NB);
    const int s = g.s, ls = g.ls;
    const int tid = threadIdx.x;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lane = tid & 63;
    const bool live = lane < s;
    const int il = live ? lane : s - 1;  // idle lanes shadow the last row (their results are dropped)
    const int trow = w * s + il;
    const int tri_il = il * (il + 1) / 2;
    const int64_t n2 = (int64_t)N * N;

    auto chunk_off = [&](int ch) { return ch == 0 ? 0 : g.E0 + (ch - 1) * g.Er; };
    auto chunk_len = [&](int ch) { return ch == 0 ? g.E0 : g.Er; };

    // staging slots (s4_slot_start): uniform start + 2 * lane
    double2 stage[PD][LPT];
    double2 dt[DT_REG ? NCH : 1][LPT];
    auto issue = [&](double2(&st)[LPT], const double* tp, int ch) {
        const double* cp = tp + chunk_off(ch);
        const int ne = chunk_len(ch);
#pragma unroll
        for (int k = 0; k < LPT; ++k) {
            const int ps = s4_slot_start(ne, LPT, w, k);
            st[k] = s4_ldnt((ps < 0 ? dts : cp + 2 * ps) + 2 * lane);
        }
    };
    const double* dts_l = dts;
    auto issue_dt = [&](double2(&d)[LPT], int ch) {
        const double* cp = dts_l + ((ch * NB + w) * LPT) * 128;
#pragma unroll
        for (int k = 0; k < LPT; ++k) d[k] = *reinterpret_cast<const double2*>(cp + k * 128 + 2 * lane);
    };

    const double* tile = eri + (T - t_begin) * g.M;
#pragma unroll
    for (int ch = 0; ch < PD; ++ch) issue(stage[ch], tile, ch);
    if (DT_REG) {
#pragma unroll
        for (int ch = 0; ch < NCH; ++ch) issue_dt(dt[ch], ch);
    } else {
        issue_dt(dt[0], 0);
    }

#ifndef NBX_S4_MFMA_WALK
    double kp[NDM];
#pragma unroll
    for (int x = 0; x < NDM; ++x) kp[x] = 0.0;
    auto flush_p = [&](int prow) {
        if (live) {
            double* kout = kpart1 + ((int64_t)blockIdx.x * S + (prow - p_first)) * NDM * N;
#pragma unroll
            for (int x = 0; x < NDM; ++x) kout[x * N + trow] = kp[x];
        }
#pragma unroll
        for (int x = 0; x < NDM; ++x) kp[x] = 0.0;
    };
#else
    // MFMA walk: accumulator fragments (see s4_walk_mfma): column lane & 3 of the product, row
    // 4 ((lane >> 2) & 3) + (lane >> 4) of every 16-row strip
    const int fr = lane & 15, fk = lane >> 4;
    const int fcol = lane & 3, frow = 4 * ((lane >> 2) & 3) + fk;
    const bool col_used = fcol < 2 * NDM, col_q = (fcol & 1) != 0;
    const int col_x = (fcol >> 1) & (NDM - 1);
    constexpr int NSTR = s4_strips(LPT);
    double acc[NSTR];
#pragma unroll
    for (int i = 0; i < NSTR; ++i) acc[i] = 0.0;
    // store the K_p (want_q = false) or K_q columns of the fragments to dst[x * N + w s + row], zero them
    auto flush_cols = [&](double* dst, bool want_q, bool store) {
        const bool mine = col_q == want_q;
#pragma unroll
        for (int i = 0; i < NSTR; ++i) {
            const int row = 16 * i + frow;
            if (store && mine && col_used && row < s) dst[col_x * N + w * s + row] = acc[i];
            acc[i] = mine ? 0.0 : acc[i];
        }
    };
    auto flush_p = [&](int prow) {
        flush_cols(kpart1 + ((int64_t)blockIdx.x * S + (prow - p_first)) * NDM * N, false, true);
    };
    // entries e = tid + NB 64 r < XW N of a tile's X table: row c = e / XW, column n = e % XW
    double xn[XR];
    auto xfetch = [&](int pp, int qq) {
#pragma unroll
        for (int r = 0; r < XR; ++r) {
            const int e = tid + NB * 64 * r;
            const int c = min(e / XW, N - 1), n = e & (XW - 1);
            xn[r] = dm[(int64_t)(n >> 1) * n2 + (int64_t)((n & 1) ? pp : qq) * N + c];
        }
    };
    auto xstore = [&](double* dst) {
#pragma unroll
        for (int r = 0; r < XR; ++r) {
            const int e = tid + NB * 64 * r;
            if (e < XW * N) dst[e] = xn[r];
        }
    };
    xfetch(p, q);
    xstore(xtab + (T & 1) * (XW * N));  // ordered before the first walk by the barrier of chunk 0
#endif
    auto store_j = [&](int par, int pj, int qj) {  // thread 0, after a barrier that follows the jred writes
        double tot = 0.0;
#pragma unroll
        for (int v = 0; v < NB; ++v) tot += jred[par * NB + v];
        jfull[(int64_t)pj * N + qj] = tot;
        jfull[(int64_t)qj * N + pj] = tot;
    };

    int p_cur = p_first, par = 0;
    int pj = -1, qj = -1;  // the tile whose J partials sit in jred[par ^ 1]
    while (T < T_end) {
        if (p != p_cur) {
            flush_p(p_cur);
            p_cur = p;
        }
        const double* tile_next = T + 1 < T_end ? tile + g.M : tile;
        // Dtot' is the same for every tile: without this the loads are hoisted out of the tile loop
        // into NB * LPT double2 registers (which is DT_REG, for the sizes that can afford it)
        if (!DT_REG) asm volatile("" : "+s"(dts_l));
#ifndef NBX_S4_MFMA_WALK
        double kq[NDM];
#pragma unroll
        for (int x = 0; x < NDM; ++x) kq[x] = 0.0;
        const double* dq = dm + (int64_t)q * N;
        const double* dp = dm + (int64_t)p * N;
#else
        const double* xt = xtab + (T & 1) * (XW * N);  // written at the end of the previous tile
        // fetch the next tile's X entries now (they are written to LDS after this tile's walks)
        {
            int pn = p, qn = q + 1;
            if (qn > pn) {
                ++pn;
                qn = 0;
            }
            if (T + 1 >= T_end) {  // no next tile: re-read this one's (never used)
                pn = p;
                qn = q;
            }
            xfetch(pn, qn);
        }
#endif
        double jacc = 0.0;
#pragma unroll
        for (int ch = 0; ch < NCH; ++ch) {
            double* buf = smem + (ch & 1) * BUFD;
            double2(&st)[LPT] = stage[ch % PD];
            double2(&dd)[LPT] = dt[DT_REG ? ch : 0];
#pragma unroll
            for (int k = 0; k < LPT; ++k) {
                jacc = fma(st[k].x, dd[k].x, fma(st[k].y, dd[k].y, jacc));
                const int ps = s4_slot_start(chunk_len(ch), LPT, w, k);
                *reinterpret_cast<double2*>((ps < 0 ? slack : buf + 2 * ps) + 2 * lane) = st[k];
            }
            // refill the staging registers: chunk ch + PD of this tile, or of the next one (pinned
            // after the stores above: hoisted loads would need a second set of registers)
            __builtin_amdgcn_sched_barrier(0);
            // (the last tile of the range re-reads itself: no tail case)
            if (ch + PD < NCH) issue(st, tile, ch + PD);
            else issue(st, tile_next, ch + PD - NCH);
            if (!DT_REG) issue_dt(dd, (ch + 1) % NCH);
            __builtin_amdgcn_sched_barrier(0);
            __syncthreads();
            if (ch == 0 && pj >= 0 && tid == 0) store_j(par ^ 1, pj, qj);

            // ---- the walk: s steps, element Lsym[trow][u*s + c]
#ifndef NBX_S4_MFMA_WALK
            if (ch == 0) {
                s4_walk<NDM, 0>(buf + w * g.tri, 0, il, tri_il, dq + w * s, dp + w * s, n2, s, kp, kq);
            } else {
                const int u = w ^ ch;
                const double* rect = buf + s4_slot(min(w, u), ch) * s * ls;
                if (w > u) s4_walk<NDM, 1>(rect + il * ls, 1, il, tri_il, dq + u * s, dp + u * s, n2, s, kp, kq);
                else s4_walk<NDM, 2>(rect + il, ls, il, tri_il, dq + u * s, dp + u * s, n2, s, kp, kq);
            }
#else
            if (ch == 0) {
                s4_walk_mfma<0, NSTR, XW>(buf + w * g.tri, 0, s, xt + w * s * XW, fr, fk, acc);
            } else {
                const int u = w ^ ch;
                const double* rect = buf + s4_slot(min(w, u), ch) * s * ls;
                if (w > u) s4_walk_mfma<1, NSTR, XW>(rect, ls, s, xt + u * s * XW, fr, fk, acc);
                else s4_walk_mfma<2, NSTR, XW>(rect, ls, s, xt + u * s * XW, fr, fk, acc);
            }
#endif
        }
        // J partial of this tile (summed by thread 0 after the next barrier)
        jacc = nbx_wave_sum(jacc);
        if (lane == 0) jred[par * NB + w] = jacc;
        pj = p;
        qj = q;
        par ^= 1;
#ifndef NBX_S4_MFMA_WALK
        if (q < p && live) {
            double* k2 = kpart2 + ((T - t_begin) * NDM) * N + trow;  // tile order: sequential stores
#pragma unroll
            for (int x = 0; x < NDM; ++x) k2[x * N] = kq[x];
        }
#else
        // K_q columns: to the tile's row-q partial (q < p), dropped on the diagonal (q == p: the
        // K_p columns already carry the whole contribution)
        flush_cols(kpart2 + ((T - t_begin) * NDM) * N, true, q < p);
        xstore(xtab + ((T + 1) & 1) * (XW * N));
#endif
        ++T;
        tile += g.M;
        if (++q > p) {
            ++p;
            q = 0;
        }
    }
    flush_p(p_cur);
    __syncthreads();
    if (tid == 0) store_j(par ^ 1, pj, qj);
}

}  // namespace

// the Dtot' table of a density (shared with jk_p8.hip, which keeps the table's format)
int nbx_jk_s4_dtot(nbx_ctx* ctx, int64_t N, const double* d_dm, int64_t ndm, double* d_dts) {
    const int NB = s4_nb(N), lpt = s4_lpt(N);
    hipLaunchKernelGGL(s4_dtot_kernel, dim3((unsigned)nbx_cdiv(NB * NB * lpt * 64, 256)), dim3(256), 0, ctx->stream, d_dm,
                       d_dts, (int)N, NB, lpt, (int)ndm);
    NBX_LAUNCH_CHECK();
    return NBX_OK;
}

namespace {

bool s4_use_s8(int64_t N) {
    // jk_s8.hip is EXPERIMENTAL: parity-tested, but measured 5-15 % slower than the four-wave VALU kernel
    // below -- opt in with NBX_JK_S8=1
    static const bool on = getenv("NBX_JK_S8") != nullptr;
    return on && nbx_jk_s8_covers(N);
}

int s4_use_dma(int NB, int lpt) {  // 0: off; 1-3: the variants of jk_s4d.hip (A/B switch while they are measured)
    static const int v = getenv("NBX_JK_DMA") ? atoi(getenv("NBX_JK_DMA")) : 0;
    return (v >= 1 && v <= 3 && nbx_jk_s4d_covers(NB, lpt)) ? v : 0;
}

struct S4Plan {
    bool s8;
    int dma;
    int NB, lpt, wgs, L, S, per_cu;
    size_t lds_bytes, dtp_off, k1_off, k2_off, total;
    S4Geom g;
};

size_t s4_align256(size_t x) { return (x + 255) & ~(size_t)255; }

S4Plan s4_plan(int64_t N, int64_t p0, int64_t np, int64_t ndm) {
    S4Plan pl;
    pl.NB = s4_nb(N);
    pl.g = s4_geom((int)N, pl.NB);
    const int nt = pl.NB * 64;
    pl.lpt = s4_lpt_class((int)nbx_cdiv((pl.g.E0 > pl.g.Er ? pl.g.E0 : pl.g.Er) / 2, nt), pl.NB);
    // chunk buffers, slack, J partials, and the MFMA walk's X tables (2 x N x 2 NDM doubles)
#ifdef NBX_S4_MFMA_WALK
    pl.lds_bytes = (size_t)(2 * pl.lpt * nt * 2 + 128 + 2 * pl.NB + 2 + 2 * (int64_t)N * 2 * ndm) * sizeof(double);
#else
    pl.lds_bytes = (size_t)(2 * pl.lpt * nt * 2 + 128 + 2 * pl.NB) * sizeof(double);
#endif
    pl.dma = s4_use_dma(pl.NB, pl.lpt);
    if (pl.dma) pl.lds_bytes = nbx_jk_s4d_lds_bytes(pl.NB, pl.lpt, pl.dma);
    int64_t per_cu = (int64_t)(S4_LDS_PER_CU / (pl.lds_bytes + 256));
    // two waves per SIMD (the staging registers need the budget), one for the long-row instances
    const int64_t by_waves = pl.dma ? nbx_jk_s4d_per_cu(pl.dma) : (pl.lpt >= 10 ? 4 : 8) / pl.NB;
    if (per_cu > by_waves) per_cu = by_waves;
    if (per_cu < 1) per_cu = 1;
    pl.per_cu = (int)per_cu;
    const int64_t slots = S4_CUS * per_cu;
    const int64_t ntiles = s4_tri(p0 + np) - s4_tri(p0);
    int64_t L = nbx_cdiv(ntiles, slots);
    if (L < 1) L = 1;
    pl.L = (int)L;
    pl.wgs = (int)nbx_cdiv(ntiles, L);
    pl.S = (int)sqrt(2.0 * (double)L) + 3;
    pl.s8 = s4_use_s8(N);
    if (pl.s8) nbx_jk_s8_plan(ntiles, &pl.wgs, &pl.L, &pl.S);  // one eight-wave workgroup per CU
    size_t off = 0;
    pl.dtp_off = off; off += s4_align256((size_t)(pl.NB * pl.NB * pl.lpt * 128) * sizeof(double));
    pl.k1_off = off; off += s4_align256((size_t)((int64_t)pl.wgs * pl.S * ndm * N) * sizeof(double));
    pl.k2_off = off; off += s4_align256((size_t)(ntiles * ndm * N) * sizeof(double));  // one row-q partial per tile
    pl.total = off;
    return pl;
}

}  // namespace

// 1: a kernel instance serves N; 2: served as the next covered size with zero rows/columns; 0: no
extern "C" int nbx_jk_packed_supported(int64_t nao) {
    return s4_supported(nao) ? 1 : (s4_padded(nao) > 0 ? 2 : 0);
}

extern "C" size_t nbx_eri_packed_bytes(int64_t nao, int64_t p0, int64_t p1) {
    const int64_t NP = s4_padded(nao);
    if (NP == 0 || p0 < 0 || p1 < p0 || p1 > nao) return 0;
    if (nbx_jk_p8_covers(nao, p0, p1)) return nbx_jk_p8_packed_bytes(nao);
    const S4Geom g = s4_geom((int)NP, s4_nb(NP));
    return (size_t)((s4_tri(p1) - s4_tri(p0)) * g.M) * sizeof(double);
}

extern "C" int nbx_eri_pack(nbx_ctx* ctx, int64_t nao, int64_t p0, int64_t p1, const double* d_eri, double* d_packed) {
    NBX_CHECK_ARG(ctx);
    NBX_CHECK_ARG(nao > 0 && p0 >= 0 && p1 >= p0 && p1 <= nao);
    const int64_t NP = s4_padded(nao);  // the tiles (p, q) with p < nao of the padded tensor: the others are zero
    if (NP == 0) {
        nbx_set_error("nbx_eri_pack: N = %lld is not covered by the packed J/K kernel", (long long)nao);
        return NBX_E_UNSUPPORTED;
    }
    if (p0 == p1) return NBX_OK;
    NBX_CHECK_ARG(d_eri && d_packed);
    if (nbx_jk_p8_covers(nao, p0, p1)) return nbx_jk_p8_pack(ctx, nao, d_eri, d_packed);
    const int64_t ntiles = s4_tri(p1) - s4_tri(p0);
    int rc = nbx_memset(ctx, d_packed, 0, nbx_eri_packed_bytes(nao, p0, p1));
    if (rc != NBX_OK) return rc;
    hipLaunchKernelGGL(s4_pack_kernel, dim3((unsigned)ntiles), dim3(256), 0, ctx->stream, d_eri, d_packed, (int)nao,
                       (int)NP, s4_nb(NP), (int)p0, s4_tri(p0));
    NBX_LAUNCH_CHECK();
    return NBX_OK;
}

extern "C" size_t nbx_jk_packed_worksize(int64_t nao, int64_t p0, int64_t p1, int64_t ndm) {
    const int64_t NP = s4_padded(nao);
    if (NP == 0 || p0 < 0 || p1 < p0 || p1 > nao || ndm <= 0) return 0;
    if (nbx_jk_p8_covers(nao, p0, p1)) return nbx_jk_p8_worksize(nao, ndm);
    size_t total = s4_plan(NP, p0, p1 - p0, ndm).total;
    if (NP != nao) total += s4_align256((size_t)((1 + 2 * ndm) * NP * NP) * sizeof(double));  // padded D and J/K
    return total;
}

static int s4_jk(nbx_ctx* ctx, int64_t nao, int64_t p0, int64_t p1, const double* d_packed, const double* d_dm,
                 int64_t ndm, double* d_jk, void* d_work, size_t work_bytes, const double* d_hv, double* d_fock,
                 double* d_vhf, const double* d_dts = nullptr);
static int s4_jk_native(nbx_ctx* ctx, int64_t nao, int64_t p0, int64_t p1, const double* d_packed, const double* d_dm,
                        int64_t ndm, double* d_jk, void* d_work, size_t work_bytes, const double* d_hv,
                        double* d_fock, double* d_vhf, const double* d_dts);

extern "C" size_t nbx_jk_dts_bytes(int64_t nao) {
    if (!s4_supported(nao)) return 0;  // (zero-padded sizes build their table themselves)
    const int NB = s4_nb(nao);
    return (size_t)(NB * NB * s4_lpt(nao) * 128) * sizeof(double);
}

extern "C" int nbx_jk_dts_init(nbx_ctx* ctx, int64_t nao, double* d_dts) {
    NBX_CHECK_ARG(ctx && d_dts && s4_supported(nao));
    return nbx_memset(ctx, d_dts, 0, nbx_jk_dts_bytes(nao));
}

extern "C" int nbx_jk_packed(nbx_ctx* ctx, int64_t nao, int64_t p0, int64_t p1, const double* d_packed,
                             const double* d_dm, int64_t ndm, double* d_jk, void* d_work, size_t work_bytes) {
    return s4_jk(ctx, nao, p0, p1, d_packed, d_dm, ndm, d_jk, d_work, work_bytes, nullptr, nullptr, nullptr);
}

extern "C" int nbx_jk_packed_fock(nbx_ctx* ctx, int64_t nao, const double* d_packed, const double* d_dm,
                                  const double* d_hv, double* d_jk, double* d_fock, double* d_vhf, void* d_work,
                                  size_t work_bytes, const double* d_dts) {
    NBX_CHECK_ARG(d_hv && d_fock);
    return s4_jk(ctx, nao, 0, nao, d_packed, d_dm, 2, d_jk, d_work, work_bytes, d_hv, d_fock, d_vhf, d_dts);
}

static int s4_jk(nbx_ctx* ctx, int64_t nao, int64_t p0, int64_t p1, const double* d_packed, const double* d_dm,
                 int64_t ndm, double* d_jk, void* d_work, size_t work_bytes, const double* d_hv, double* d_fock,
                 double* d_vhf, const double* d_dts) {
    NBX_CHECK_ARG(ctx && d_dm && d_jk);
    NBX_CHECK_ARG(nao > 0 && p0 >= 0 && p1 >= p0 && p1 <= nao);
    NBX_CHECK_ARG(d_packed != nullptr || p0 == p1);
    NBX_CHECK_ARG(ndm == 1 || ndm == 2);
    const int64_t NPAD = s4_padded(nao);
    if (NPAD == 0) {
        nbx_set_error("nbx_jk_packed: N = %lld is not covered (N <= 256 within %d of an even size the kernel has an "
                      "instance for)", (long long)nao, S4_MAX_PAD);
        return NBX_E_UNSUPPORTED;
    }
    const size_t need = nbx_jk_packed_worksize(nao, p0, p1, ndm);
    if (d_work == nullptr || work_bytes < need) {
        nbx_set_error("nbx_jk_packed: workspace %zu < %zu bytes", work_bytes, need);
        return NBX_E_NOMEM;
    }
    NBX_CHECK_ARG((reinterpret_cast<uintptr_t>(d_packed) & 15) == 0 && (reinterpret_cast<uintptr_t>(d_work) & 15) == 0);
    if (p1 == p0) return nbx_memset(ctx, d_jk, 0, (size_t)((1 + ndm) * nao * nao) * sizeof(double));
    if (nbx_jk_p8_covers(nao, p0, p1)) return nbx_jk_p8(ctx, nao, d_packed, d_dm, ndm, d_jk, d_work, d_hv, d_fock, d_vhf, d_dts);
    if (NPAD != nao) {
        // Run as the NPAD x NPAD problem whose extra rows and columns are zero: the tiles (p, q) with
        // p >= nao vanish (never stored, never visited); D is padded on the way in, J/K cropped on the
        // way out.  The Fock assembly then is its own launch.
        char* tail = static_cast<char*>(d_work) + s4_plan(NPAD, p0, p1 - p0, ndm).total;
        double* dm_pad = reinterpret_cast<double*>(tail);
        double* jk_pad = dm_pad + ndm * NPAD * NPAD;
        const int64_t tin = ndm * NPAD * NPAD, tout = (1 + ndm) * nao * nao;
        hipLaunchKernelGGL(s4_pad_square_kernel, dim3((unsigned)nbx_cdiv(tin, 256)), dim3(256), 0, ctx->stream, d_dm,
                           dm_pad, (int)nao, (int)NPAD, (int)ndm);
        NBX_LAUNCH_CHECK();
        int rc = s4_jk_native(ctx, NPAD, p0, p1, d_packed, dm_pad, ndm, jk_pad, d_work, work_bytes, nullptr, nullptr,
                              nullptr, nullptr);
        if (rc != NBX_OK) return rc;
        hipLaunchKernelGGL(s4_crop_square_kernel, dim3((unsigned)nbx_cdiv(tout, 256)), dim3(256), 0, ctx->stream,
                           jk_pad, d_jk, (int)nao, (int)NPAD, (int)(1 + ndm));
        NBX_LAUNCH_CHECK();
        if (d_fock != nullptr) return nbx_fock_uhf(ctx, nao, d_hv, 3, nullptr, d_jk, d_fock, d_vhf);
        return NBX_OK;
    }
    return s4_jk_native(ctx, nao, p0, p1, d_packed, d_dm, ndm, d_jk, d_work, work_bytes, d_hv, d_fock, d_vhf, d_dts);
}

// the kernel proper, for a size it has an instance for (arguments checked by s4_jk)
static int s4_jk_native(nbx_ctx* ctx, int64_t nao, int64_t p0, int64_t p1, const double* d_packed, const double* d_dm,
                        int64_t ndm, double* d_jk, void* d_work, size_t work_bytes, const double* d_hv,
                        double* d_fock, double* d_vhf, const double* d_dts) {
    const int64_t np = p1 - p0, N = nao, n2 = N * N;
    (void)work_bytes;
    const S4Plan pl = s4_plan(N, p0, np, ndm);
    char* base = static_cast<char*>(d_work);
    double* dtp = reinterpret_cast<double*>(base + pl.dtp_off);
    double* k1 = reinterpret_cast<double*>(base + pl.k1_off);
    double* k2 = reinterpret_cast<double*>(base + pl.k2_off);
    if (np < N) {  // J entries this slab does not own must read as zero
        const int rc = nbx_memset(ctx, d_jk, 0, (size_t)n2 * sizeof(double));
        if (rc != NBX_OK) return rc;
    }
    if (d_dts != nullptr) {  // the caller's table (left by nbx_huz_cycle_scalars_dts for this density)
        dtp = const_cast<double*>(d_dts);
    } else {
        hipLaunchKernelGGL(s4_dtot_kernel, dim3((unsigned)nbx_cdiv(pl.NB * pl.NB * pl.lpt * 64, 256)), dim3(256), 0,
                           ctx->stream, d_dm, dtp, (int)N, pl.NB, pl.lpt, (int)ndm);
        NBX_LAUNCH_CHECK();
    }
    const int64_t t_begin = s4_tri(p0), t_end = s4_tri(p1);
    {
        nbx_prof_scope prof(ctx, NBX_PROF_JK_DENSE);
#define NBX_S4_GO(NDM_, NB_, LPT_, PD_, DT_, WV_)                                                                          \
    do {                                                                                                              \
        static bool attr_set = false;                                                                                 \
        if (!attr_set) {                                                                                              \
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&jk_s4_kernel<NDM_, NB_, LPT_, PD_, DT_, WV_>),        \
                                      hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);                        \
            attr_set = true;                                                                                          \
        }                                                                                                             \
        hipLaunchKernelGGL((jk_s4_kernel<NDM_, NB_, LPT_, PD_, DT_, WV_>), dim3((unsigned)pl.wgs), dim3(NB_ * 64),         \
                           pl.lds_bytes, ctx->stream, d_packed, d_dm, dtp, d_jk, k1, k2, (int)N, (int)p0, (int)np,    \
                           t_begin, t_end, pl.L, pl.S);                                                               \
    } while (0)
#define NBX_S4_NDM(NB_, LPT_, PD_, DT_, WV_)                                                                               \
    do {                                                                                                              \
        if (ndm == 2) NBX_S4_GO(2, NB_, LPT_, PD_, DT_, WV_);                                                              \
        else NBX_S4_GO(1, NB_, LPT_, PD_, DT_, WV_);                                                                       \
    } while (0)
        if (pl.dma && !pl.s8) {
            const int rcd = nbx_jk_s4d_launch(ctx, pl.dma, N, p0, np, ndm, pl.lpt, d_packed, d_dm, dtp, d_jk, k1, k2, t_begin,
                                              t_end, pl.wgs, pl.L, pl.S);
            if (rcd != NBX_OK) return rcd;
        } else if (pl.s8) {
            const int rc8 = nbx_jk_s8_launch(ctx, N, p0, np, ndm, d_packed, d_dm, dtp, d_jk, k1, k2, t_begin, t_end,
                                             pl.wgs, pl.L, pl.S);
            if (rc8 != NBX_OK) return rc8;
        } else if (pl.NB == 2) {
            if (pl.lpt == 2) NBX_S4_NDM(2, 2, 2, true, 2);
            else if (pl.lpt == 6) NBX_S4_NDM(2, 6, 2, true, 2);
            else NBX_S4_NDM(2, 10, 2, true, 1);
        } else {
            if (pl.lpt == 2) NBX_S4_NDM(4, 2, 4, true, 2);
            else if (pl.lpt == 6) {
                static const bool dt_stream = getenv("NBX_S4_DT_STREAM") != nullptr;  // probe: Dtot' re-read per chunk
                if (dt_stream) NBX_S4_NDM(4, 6, 2, false, 2);
                else NBX_S4_NDM(4, 6, 2, true, 2);
            }
            else if (pl.lpt == 10) NBX_S4_NDM(4, 10, 2, true, 1);   // one workgroup per CU (LDS): 512 registers
            else NBX_S4_NDM(4, 17, 1, false, 1);
        }
#undef NBX_S4_NDM
#undef NBX_S4_GO
    }
    NBX_LAUNCH_CHECK();
    return nbx_jk_sym_reduce(ctx, k1, k2, d_jk + n2, N, p0, np, ndm, t_begin, pl.L, pl.S, d_jk, d_hv, d_fock, d_vhf, 1);
}
