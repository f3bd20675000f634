// libnbx: J/K contraction on packed integrals, matrix-pipe walk (include/nbx.h "J/K contraction,
// packed form").  Same tile format, same persistent equal-range workgroups, partial buffers and
// reduction kernel as jk_s4.hip (read that file's header first); what differs is how a tile is
// consumed:
//
//   * ONE workgroup of EIGHT waves per CU (two waves per block of rows) instead of two workgroups of
//     four.  The whole of LDS belongs to it: besides the two chunk buffers it holds the Dtot' table
//     of the J sum (88 KB at N = 148), which jk_s4 keeps in 96 VGPRs per thread.  With those
//     registers free the kernel fits 2 waves/SIMD with no scratch -- which matters more than it
//     usually does: a spill reload is a vector-memory access, the vector-memory counter is in
//     order, so every reload waits for all the streaming loads in flight.
//   * the walk of a chunk runs on the matrix pipe.  Wave (w, h) owns rows [h ceil(s/2), ...) of
//     block w and needs   out[t][n] += sum_c M[t][c] X[c][n]   with M its side of the chunk and
//     X[c] = (D^a_q, D^a_p, D^b_q, D^b_p)[u s + c]:  v_mfma_f64_4x4x4_4b_f64 takes a 16 x 4 block of M
//     as A (lane 16 k + m reads M[r0 + m][4 j + k] from the LDS-resident chunk, whatever its stored
//     orientation) and a 4 x 4 block of X as B (lane 16 k + 4 b + n reads X[4 j + k][n] from an
//     LDS table written once per tile); D_b[i][n] lands in lane 16 i + 4 b + n.  ceil(s/4) MFMAs per
//     16-row strip replace s steps of ~10 VALU/scalar instructions each.
//     (Operand layout and issue rate measured with scratch/probe/mfma444.hip: 132 G instr/s, 3.5x
//     the 16x16x4 form, whose 16 columns this product could not fill.)
#include <cstdlib>

#include "nbx_common.h"
#include "jk_s4_layout.h"

namespace {

constexpr int S8_NB = 4;        // blocks of rows (as the jk_s4 instance it replaces)
constexpr int S8_WAVES = 8;     // two per block
constexpr int S8_THREADS = 64 * S8_WAVES;
constexpr int S8_LPT = 3;       // 16-byte loads per thread per chunk: 24 slots of 64 pairs, as 4 waves x 6
constexpr int S8_PD = 2;        // chunks of prefetch distance
constexpr int S8_BUFD = S8_LPT * S8_WAVES * 128;  // doubles per chunk buffer
constexpr int S8_NSTR = 2;      // 16-row strips per wave: ceil(s/2) <= 32 rows
constexpr int S8_CUS = 256;

__device__ __forceinline__ double2 s8_ldnt(const double* p) {
    typedef double nbx_d2 __attribute__((ext_vector_type(2)));
    const nbx_d2 t = __builtin_nontemporal_load(reinterpret_cast<const nbx_d2*>(p));
    return make_double2(t.x, t.y);
}

// KIND 0: packed triangle (lb = its start), 1: rectangle, this wave's rows are the stored rows,
// 2: rectangle, this wave's rows are the stored columns.  r0: first row of the wave inside the
// block; rows past the block and columns past s: addresses stay on the workgroup's LDS, the
// products land in rows nobody stores / meet a zero in X.
constexpr int S8_MAX_NK = 10;  // k-steps of four columns: s <= 39 in the instance this kernel replaces

// TRI: packed triangle (lb = its start); otherwise a rectangle addressed as row * sa + column * sb of
// THIS WAVE's matrix: (sa, sb) = (ls, 1) when its rows are the stored rows, (1, ls) when they are the
// stored columns -- one code path for both sides.
template <bool TRI, int XW, int S8_KG>
__device__ __forceinline__ void s8_walk(const double* lb, int sa, int sb, int s, int r0, const double* xs, int fr,
                                        int fk, double (&acc)[S8_NSTR]) {
    const int xcol = fr & (XW - 1);
    const int nk = (s + 3) >> 2;
    int tt[S8_NSTR], ta[S8_NSTR];
#pragma unroll
    for (int i = 0; i < S8_NSTR; ++i) {
        tt[i] = min(r0 + 16 * i + fr, s - 1);
        ta[i] = TRI ? (tt[i] * (tt[i] + 1)) >> 1 : tt[i] * sa;
    }
    // the operands of S8_KG steps are requested before the first of their MFMAs (LDS returns in
    // order: the MFMAs start as the first reads land)
#pragma unroll 1
    for (int j0 = 0; j0 < nk; j0 += S8_KG) {
        double a[S8_KG][S8_NSTR], b[S8_KG];
#pragma unroll
        for (int jj = 0; jj < S8_KG; ++jj) {
            const int c = 4 * (j0 + jj) + fk, cc = min(c, s - 1);  // steps past nk: X is zero there
#pragma unroll
            for (int i = 0; i < S8_NSTR; ++i) {
                int off;
                if (TRI) off = tt[i] >= cc ? ta[i] + cc : ((cc * (cc + 1)) >> 1) + tt[i];
                else off = ta[i] + cc * sb;
                a[jj][i] = lb[off];
            }
            const double v = xs[cc * XW + xcol];
            b[jj] = c < s ? v : 0.0;
        }
#pragma unroll
        for (int jj = 0; jj < S8_KG; ++jj) {
#pragma unroll
            for (int i = 0; i < S8_NSTR; ++i)
                acc[i] = __builtin_amdgcn_mfma_f64_4x4x4f64(a[jj][i], b[jj], acc[i], 0, 0, 0);
        }
    }
}

// DT_LDS: the Dtot' table lives in LDS (one workgroup per CU); otherwise each thread fetches the six
// weights of its next chunk from the L2-resident table right before the tile loads of chunk + PD
// (older than those in the in-order vector-memory queue, so consuming them waits for nothing the
// chunk itself does not wait for) and two workgroups share a CU.
template <int NDM, bool DT_LDS>
__global__ __launch_bounds__(S8_THREADS) __attribute__((amdgpu_waves_per_eu(DT_LDS ? 2 : 4, DT_LDS ? 2 : 4))) void jk_s8_kernel(
    const double* __restrict__ eri, const double* __restrict__ dm, const double* __restrict__ dts,
    double* __restrict__ jfull, double* __restrict__ kpart1, double* __restrict__ kpart2, int N, int p0, int np,
    int64_t t_begin, int64_t t_end, int L, int S, int dbg) {
    constexpr int NB = S8_NB, NCH = NB, LPT = S8_LPT, PD = S8_PD, BUFD = S8_BUFD, XW = 2 * NDM;
    constexpr int DTN = NCH * S8_WAVES * LPT * 128;  // doubles in the Dtot' table
    constexpr int KG = DT_LDS ? 5 : 2;               // k-steps whose walk operands are requested together
    // buf[2][BUFD] | slack[128] | jred[2][8] | xtab[2][N][XW] | dtab[DTN]
    extern __shared__ __attribute__((aligned(16))) double smem[];
    double* slack = smem + 2 * BUFD;
    double* jred = slack + 128;
    double* xtab = jred + 2 * S8_WAVES;
    double* dtab = xtab + ((2 * N * XW + 1) & ~1);  // (DT_LDS only)

    int64_t T = t_begin + (int64_t)blockIdx.x * L;
    const int64_t T_end = min(t_end, T + L);
    if (T >= T_end) return;  // uniform for the whole workgroup
    int p = s4_tri_row(T);
    int q = (int)(T - s4_tri(p));
    const int p_first = p;

    const S4Geom g = s4_geom(N, NB);
    const int s = g.s, ls = g.ls;
    const int tid = threadIdx.x;
    const int W = __builtin_amdgcn_readfirstlane(tid >> 6);  // wave: staging slots (W, k)
    const int w = W >> 1, h = W & 1;                         // block of rows, half of it
    const int lane = tid & 63;
    const int64_t n2 = (int64_t)N * N;
    const int hs = (s + 1) >> 1;             // rows per half
    const int r0 = h * hs;                   // this wave's first row inside the block
    const int rend = h ? s : hs;             // ... and the end of its rows

    auto chunk_off = [&](int ch) { return ch == 0 ? 0 : g.E0 + (ch - 1) * g.Er; };
    auto chunk_len = [&](int ch) { return ch == 0 ? g.E0 : g.Er; };

    // the Dtot' table (the same for every tile): global -> LDS once
    if (DT_LDS)
        for (int i = tid; i < DTN / 2; i += S8_THREADS)
            reinterpret_cast<double2*>(dtab)[i] = reinterpret_cast<const double2*>(dts)[i];
    double2 wts[LPT];  // (!DT_LDS) weights of the chunk about to be consumed
    auto issue_w = [&](int ch) {
        const double* dd = dts + ((ch * S8_WAVES + W) * LPT) * 128 + 2 * lane;
#pragma unroll
        for (int k = 0; k < LPT; ++k) wts[k] = *reinterpret_cast<const double2*>(dd + k * 128);
    };

    double2 stage[PD][LPT];
    auto issue = [&](double2(&st)[LPT], const double* tp, int ch) {
        const double* cp = tp + chunk_off(ch);
        const int ne = chunk_len(ch);
#pragma unroll
        for (int k = 0; k < LPT; ++k) {
            const int ps = s4_slot_start(ne, LPT, W, k);
            st[k] = s8_ldnt((ps < 0 ? dts : cp + 2 * ps) + 2 * lane);
        }
    };
    const double* tile = eri + (T - t_begin) * g.M;
#pragma unroll
    for (int ch = 0; ch < PD; ++ch) issue(stage[ch], tile, ch);
    if (!DT_LDS) issue_w(0);

    // accumulator fragments: column lane & 3 of the product, row 4 ((lane >> 2) & 3) + (lane >> 4) of a strip
    const int fr = lane & 15, fk = lane >> 4;
    const int fcol = lane & 3, frow = 4 * ((lane >> 2) & 3) + fk;
    const bool col_used = fcol < 2 * NDM, col_q = (fcol & 1) != 0;
    const int col_x = (fcol >> 1) & (NDM - 1);
    double acc[S8_NSTR];
#pragma unroll
    for (int i = 0; i < S8_NSTR; ++i) acc[i] = 0.0;
    // store the K_p (want_q = false) or K_q columns to dst[x * N + w s + row], then zero them
    auto flush_cols = [&](double* dst, bool want_q, bool store) {
        const bool mine = col_q == want_q;
#pragma unroll
        for (int i = 0; i < S8_NSTR; ++i) {
            const int row = r0 + 16 * i + frow;
            if (store && mine && col_used && 16 * i + frow < hs && row < rend) dst[col_x * N + w * s + row] = acc[i];
            acc[i] = mine ? 0.0 : acc[i];
        }
    };
    auto flush_p = [&](int prow) {
        flush_cols(kpart1 + ((int64_t)blockIdx.x * S + (prow - p_first)) * NDM * N, false, true);
    };
    auto store_j = [&](int par, int pj, int qj) {  // thread 0, after a barrier that follows the jred writes
        double tot = 0.0;
#pragma unroll
        for (int v = 0; v < S8_WAVES; ++v) tot += jred[par * S8_WAVES + v];
        jfull[(int64_t)pj * N + qj] = tot;
        jfull[(int64_t)qj * N + pj] = tot;
    };
    // X table: entries e = tid + 512 r < XW N; row c = e / XW, column n = e % XW
    constexpr int XR = 2;  // XW N <= 4 * 256
    double xn[XR];
    // byte offset of entry r inside the density array without the row term, and whether its row is p
    unsigned xoff[XR];
    bool xisp[XR];
#pragma unroll
    for (int r = 0; r < XR; ++r) {
        const int e = tid + S8_THREADS * r;
        const int c = min(e / XW, N - 1), n = e & (XW - 1);
        xoff[r] = 8u * (unsigned)((n >> 1) * (int)n2 + c);
        xisp[r] = (n & 1) != 0;
    }
    auto xfetch = [&](int pp, int qq) {
#pragma unroll
        for (int r = 0; r < XR; ++r) {
            const unsigned off = xoff[r] + 8u * (unsigned)((xisp[r] ? pp : qq) * N);
            // Not a load the compiler knows about: it would guard the use (one tile later, with loops in
            // between) with s_waitcnt vmcnt(0), i.e. with a wait for every streaming load in flight.  The
            // wait is written by hand in xwait(): the counter is in order and NCH * LPT vector loads are
            // always issued between this one and its use.  (Scalar base + 32-bit offset: no 64-bit
            // address pairs to keep alive across the tile.)
            asm volatile("global_load_dwordx2 %0, %1, %2" : "=v"(xn[r]) : "v"(off), "s"(dm) : "memory");
        }
    };
    auto xwait = [&]() {
        static_assert(XR == 2, "operand list below");
        asm volatile("s_waitcnt vmcnt(%2)" : "+v"(xn[0]), "+v"(xn[1]) : "n"(NCH * LPT));
    };
    auto xstore = [&](double* dst) {
#pragma unroll
        for (int r = 0; r < XR; ++r) {
            const int e = tid + S8_THREADS * r;
            if (e < XW * N) dst[e] = xn[r];
        }
    };
    xfetch(p, q);
    asm volatile("s_waitcnt vmcnt(0)" : "+v"(xn[0]), "+v"(xn[1]));
    xstore(xtab + (T & 1) * (XW * N));
    __syncthreads();  // dtab (and the first X table) complete before the first J sum

    int p_cur = p_first, par = 0;
    int pj = -1, qj = -1;  // the tile whose J partials sit in jred[par ^ 1]
    while (T < T_end) {
        if (p != p_cur) {
            flush_p(p_cur);
            p_cur = p;
        }
        const double* tile_next = T + 1 < T_end ? tile + g.M : tile;
        const double* xt = xtab + (T & 1) * (XW * N);  // written at the end of the previous tile
        {   // fetch the next tile's X entries now (written to LDS after this tile's walks)
            int pn = p, qn = q + 1;
            if (qn > pn) {
                ++pn;
                qn = 0;
            }
            if (T + 1 >= T_end) {
                pn = p;
                qn = q;
            }
            if (!(dbg & 4)) xfetch(pn, qn);
        }
        double jacc = 0.0;
#pragma unroll
        for (int ch = 0; ch < NCH; ++ch) {
            double* buf = smem + (ch & 1) * BUFD;
            double2(&st)[LPT] = stage[ch % PD];
            const double* dd = dtab + ((ch * S8_WAVES + W) * LPT) * 128 + 2 * lane;
#pragma unroll
            for (int k = 0; k < LPT; ++k) {
                if (!(dbg & 2)) {
                    const double2 d2 = DT_LDS ? *reinterpret_cast<const double2*>(dd + k * 128) : wts[k];
                    jacc = fma(st[k].x, d2.x, fma(st[k].y, d2.y, jacc));
                }
                const int ps = s4_slot_start(chunk_len(ch), LPT, W, k);
                *reinterpret_cast<double2*>((ps < 0 ? slack : buf + 2 * ps) + 2 * lane) = st[k];
            }
            __builtin_amdgcn_sched_barrier(0);
            if (!DT_LDS) issue_w((ch + 1) % NCH);
            if (ch + PD < NCH) issue(st, tile, ch + PD);
            else issue(st, tile_next, ch + PD - NCH);
            __builtin_amdgcn_sched_barrier(0);
            __syncthreads();
            if (ch == 0 && pj >= 0 && tid == 0) store_j(par ^ 1, pj, qj);

            if (dbg & 1) {
            } else if (ch == 0) {
                s8_walk<true, XW, KG>(buf + w * g.tri, 0, 0, s, r0, xt + w * s * XW, fr, fk, acc);
            } else {
                const int u = w ^ ch;
                const double* rect = buf + s4_slot(min(w, u), ch) * s * ls;
                const bool rows = w > u;  // this wave's rows are the rectangle's stored rows
                s8_walk<false, XW, KG>(rect, rows ? ls : 1, rows ? 1 : ls, s, r0, xt + u * s * XW, fr, fk, acc);
            }
        }
        jacc = nbx_wave_sum(jacc);
        if (lane == 0) jred[par * S8_WAVES + W] = jacc;
        pj = p;
        qj = q;
        par ^= 1;
        // K_q columns: the tile's row-q partial (q < p); on the diagonal the K_p columns carry it all
        if (!(dbg & 8)) flush_cols(kpart2 + ((T - t_begin) * NDM) * N, true, q < p);  // tile order
        if (!(dbg & 4)) {
            xwait();
            xstore(xtab + ((T + 1) & 1) * (XW * N));
        }
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

// sizes this kernel takes over from jk_s4.hip: its NB = 4, six-loads-per-thread instance
bool nbx_jk_s8_covers(int64_t N) {
    if (!s4_supported(N) || s4_nb(N) != S8_NB || s4_lpt(N) != 2 * S8_LPT || N / S8_NB > 4 * S8_MAX_NK) return false;
    const size_t lds = (size_t)(2 * S8_BUFD + 128 + 2 * S8_WAVES + ((2 * N * 4 + 1) & ~1ll) +
                                S8_NB * S8_WAVES * S8_LPT * 128) * sizeof(double);
    return lds <= 160 * 1024;
}

// workgroups / tiles per workgroup / row slots of the partial buffers for the tile range of a slab
void nbx_jk_s8_plan(int64_t ntiles, int* wgs, int* L, int* S) {
    const int per_cu = getenv("NBX_S8_ONE_WG") == nullptr ? 2 : 1;
    int64_t l = nbx_cdiv(ntiles, S8_CUS * per_cu);
    if (l < 1) l = 1;
    *L = (int)l;
    *wgs = (int)nbx_cdiv(ntiles, l);
    *S = (int)sqrt(2.0 * (double)l) + 3;
}

int nbx_jk_s8_launch(nbx_ctx* ctx, int64_t N, int64_t p0, int64_t np, int64_t ndm, const double* d_packed,
                     const double* d_dm, const double* d_dts, double* d_j, double* k1, double* k2, int64_t t_begin,
                     int64_t t_end, int wgs, int L, int S) {
    const int xw = 2 * (int)ndm;
    static const bool two = getenv("NBX_S8_ONE_WG") == nullptr;  // default: two workgroups per CU, weights from L2
    const size_t lds = (size_t)(2 * S8_BUFD + 128 + 2 * S8_WAVES + ((2 * N * xw + 1) & ~1ll) +
                                (two ? 0 : S8_NB * S8_WAVES * S8_LPT * 128)) * sizeof(double);
    static const int dbg = getenv("NBX_S8_DEBUG") ? atoi(getenv("NBX_S8_DEBUG")) : 0;  // timing ablations only
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&jk_s8_kernel<1, true>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&jk_s8_kernel<2, true>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&jk_s8_kernel<1, false>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&jk_s8_kernel<2, false>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024);
        attr_set = true;
    }
#define NBX_S8_GO(NDM_, DT_)                                                                                          \
    hipLaunchKernelGGL((jk_s8_kernel<NDM_, DT_>), dim3((unsigned)wgs), dim3(S8_THREADS), lds, ctx->stream, d_packed,  \
                       d_dm, d_dts, d_j, k1, k2, (int)N, (int)p0, (int)np, t_begin, t_end, L, S, dbg)
    if (ndm == 2) {
        if (two) NBX_S8_GO(2, false);
        else NBX_S8_GO(2, true);
    } else {
        if (two) NBX_S8_GO(1, false);
        else NBX_S8_GO(1, true);
    }
#undef NBX_S8_GO
    NBX_LAUNCH_CHECK();
    return NBX_OK;
}
