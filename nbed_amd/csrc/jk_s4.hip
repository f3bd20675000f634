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

// The EXPERIMENTAL forms of the packed kernel (parity green, none measurably faster than the
// kernel in this file) are linked only into a `make EXPERIMENTAL=1` build, where the environment
// selects them (NBX_JK_M4 / NBX_JK_P8 / NBX_JK_S8 / NBX_JK_DMA); the shipped library holds what runs.
// jk_m4.hip: the walk on the matrix cores (v_mfma_f64_4x4x4_4b_f64, block-major swizzled tiles streamed into LDS by the
// load unit, loading / walking waves): serves N = 148 (the size it is instantiated for) unless NBX_JK_M4=0 -- 0.87 of this
// file's kernel time there
bool nbx_jk_m4_covers(int64_t N);
size_t nbx_jk_m4_weights_bytes(int64_t N);
size_t nbx_jk_m4_packed_bytes(int64_t N, int64_t p0, int64_t p1);
size_t nbx_jk_m4_worksize(int64_t N, int64_t p0, int64_t p1, int64_t ndm);
int nbx_jk_m4_pack(nbx_ctx* ctx, int64_t N, int64_t nsrc, int64_t p0, int64_t p1, const double* d_eri, double* d_packed);
int nbx_jk_m4(nbx_ctx* ctx, int64_t N, int64_t p0, int64_t p1, const double* d_packed, const double* d_dm, int64_t ndm,
              double* d_jk, void* d_work, const double* d_hv, double* d_fock, double* d_vhf, const double* d_wt);
// jk_m8.hip: the 8-fold packed form (every integral once) for the sizes it has an instance for, unless NBX_JK_M8=0
bool nbx_jk_m8_covers(int64_t N);
void nbx_jk_m8_weight_layout(int64_t N, int out[4]);
size_t nbx_jk_m8_weights_bytes(int64_t N);
size_t nbx_jk_m8_packed_bytes(int64_t N, int64_t p0, int64_t p1);
size_t nbx_jk_m8_worksize(int64_t N, int64_t p0, int64_t p1, int64_t ndm);
int nbx_jk_m8_pack(nbx_ctx* ctx, int64_t N, int64_t nsrc, int64_t p0, int64_t p1, const double* d_eri, double* d_packed);
int nbx_jk_m8(nbx_ctx* ctx, int64_t N, int64_t p0, int64_t p1, const double* d_packed, const double* d_dm, int64_t ndm,
              double* d_jk, void* d_work, const double* d_hv, double* d_fock, double* d_vhf, const double* d_wt);
// jk_mx.hip: the same walk for the sizes above (N = 152 .. 288, a tile in 4 .. 20 chunks), unless NBX_JK_MX=0
bool nbx_jk_mx_covers(int64_t N);
int64_t nbx_jk_mx_padded(int64_t N);
size_t nbx_jk_mx_packed_bytes(int64_t N, int64_t p0, int64_t p1);
size_t nbx_jk_mx_worksize(int64_t N, int64_t p0, int64_t p1, int64_t ndm);
int nbx_jk_mx_pack(nbx_ctx* ctx, int64_t N, int64_t nsrc, int64_t p0, int64_t p1, const double* d_eri, double* d_packed);
int nbx_jk_mx(nbx_ctx* ctx, int64_t N, int64_t p0, int64_t p1, const double* d_packed, const double* d_dm, int64_t ndm,
              double* d_jk, void* d_work, const double* d_hv, double* d_fock, double* d_vhf);
#ifdef NBX_EXPERIMENTAL
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
#else
static inline bool nbx_jk_p8_covers(int64_t, int64_t, int64_t) { return false; }
static inline size_t nbx_jk_p8_packed_bytes(int64_t) { return 0; }
static inline size_t nbx_jk_p8_worksize(int64_t, int64_t) { return 0; }
static inline int nbx_jk_p8_pack(nbx_ctx*, int64_t, const double*, double*) { return NBX_E_UNSUPPORTED; }
static inline int nbx_jk_p8(nbx_ctx*, int64_t, const double*, const double*, int64_t, double*, void*, const double*, double*,
                            double*, const double*) { return NBX_E_UNSUPPORTED; }
static inline bool nbx_jk_s8_covers(int64_t) { return false; }
static inline void nbx_jk_s8_plan(int64_t, int*, int*, int*) {}
static inline int nbx_jk_s8_launch(nbx_ctx*, int64_t, int64_t, int64_t, int64_t, const double*, const double*, const double*,
                                   double*, double*, double*, int64_t, int64_t, int, int, int) { return NBX_E_UNSUPPORTED; }
static inline bool nbx_jk_s4d_covers(int, int) { return false; }
static inline size_t nbx_jk_s4d_lds_bytes(int, int, int) { return 0; }
static inline int nbx_jk_s4d_per_cu(int) { return 2; }
static inline int nbx_jk_s4d_launch(nbx_ctx*, int, int64_t, int64_t, int64_t, int64_t, int, const double*, const double*,
                                    const double*, double*, double*, double*, int64_t, int64_t, int, int, int) {
    return NBX_E_UNSUPPORTED;
}
#endif

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
    extern __shared__ __attribute__((aligned(16))) double smem[];  // buf[2][BUFD] | slack[128] | jred[2][NB]
    double* slack = smem + 2 * BUFD;
    double* jred = slack + 128;

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
        double kq[NDM];
#pragma unroll
        for (int x = 0; x < NDM; ++x) kq[x] = 0.0;
        const double* dq = dm + (int64_t)q * N;
        const double* dp = dm + (int64_t)p * N;
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
            if (ch == 0) {
                s4_walk<NDM, 0>(buf + w * g.tri, 0, il, tri_il, dq + w * s, dp + w * s, n2, s, kp, kq);
            } else {
                const int u = w ^ ch;
                const double* rect = buf + s4_slot(min(w, u), ch) * s * ls;
                if (w > u) s4_walk<NDM, 1>(rect + il * ls, 1, il, tri_il, dq + u * s, dp + u * s, n2, s, kp, kq);
                else s4_walk<NDM, 2>(rect + il, ls, il, tri_il, dq + u * s, dp + u * s, n2, s, kp, kq);
            }
        }
        // J partial of this tile (summed by thread 0 after the next barrier)
        jacc = nbx_wave_sum(jacc);
        if (lane == 0) jred[par * NB + w] = jacc;
        pj = p;
        qj = q;
        par ^= 1;
        if (q < p && live && trow <= q) {  // (columns <= q only: K is symmetric, jk_sym_reduce_kernel(k_lower) mirrors the sums --
                                           //  a third of the bytes these per-tile rows come to: DESIGN.md section 9 (xii))
            double* k2 = kpart2 + ((T - t_begin) * NDM) * N + trow;  // tile order: sequential stores
#pragma unroll
            for (int x = 0; x < NDM; ++x) k2[x * N] = kq[x];
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
    // chunk buffers, slack, J partials
    pl.lds_bytes = (size_t)(2 * pl.lpt * nt * 2 + 128 + 2 * pl.NB) * sizeof(double);
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

// The size jk_m4.hip runs nao as: nao itself or the next multiple of four (at most three zero rows / columns more), if that
// size has an instance; 0: jk_s4.hip's business
static int64_t m4_padded(int64_t nao) {
    const int64_t n4 = (nao + 3) / 4 * 4;
    return (nao > 0 && nbx_jk_m4_covers(n4)) ? n4 : 0;
}

// the same for jk_m8.hip (the 8-fold form of these sizes; it takes them when it has the instance)
static int64_t m8_padded(int64_t nao) {
    const int64_t n4 = (nao + 3) / 4 * 4;
    return (nao > 0 && nbx_jk_m8_covers(n4)) ? n4 : 0;
}

// the size jk_mx.hip runs nao as (nao itself or the next instance, at most eight zero rows / columns more); 0: none
static int64_t mx_padded(int64_t nao) { return (nao > 0 && !m4_padded(nao)) ? nbx_jk_mx_padded(nao) : 0; }

// 1: a kernel instance serves N; 2: served as the next covered size with zero rows/columns; 0: no
extern "C" int nbx_jk_packed_supported(int64_t nao) {
    if (const int64_t nx = mx_padded(nao)) return nx == nao ? 1 : 2;
    return s4_supported(nao) ? 1 : (s4_padded(nao) > 0 ? 2 : 0);
}

extern "C" int nbx_jk_packed_fold(int64_t nao) {
    if (nbx_jk_packed_supported(nao) == 0) return 0;
    return m8_padded(nao) ? 8 : 4;
}

extern "C" size_t nbx_eri_packed_bytes(int64_t nao, int64_t p0, int64_t p1) {
    if (p0 < 0 || p1 < p0 || p1 > nao) return 0;
    if (const int64_t nx = mx_padded(nao)) return nbx_jk_mx_packed_bytes(nx, p0, p1);
    const int64_t NP = s4_padded(nao);
    if (NP == 0) return 0;
    if (const int64_t n8 = m8_padded(nao)) return nbx_jk_m8_packed_bytes(n8, p0, p1);
    if (m4_padded(nao)) return nbx_jk_m4_packed_bytes(m4_padded(nao), p0, p1);
    if (nbx_jk_p8_covers(nao, p0, p1)) return nbx_jk_p8_packed_bytes(nao);
    const S4Geom g = s4_geom((int)NP, s4_nb(NP));
    return (size_t)((s4_tri(p1) - s4_tri(p0)) * g.M) * sizeof(double);
}

extern "C" int nbx_eri_pack(nbx_ctx* ctx, int64_t nao, int64_t p0, int64_t p1, const double* d_eri, double* d_packed) {
    NBX_CHECK_ARG(ctx);
    NBX_CHECK_ARG(nao > 0 && p0 >= 0 && p1 >= p0 && p1 <= nao);
    if (const int64_t nx = mx_padded(nao)) {
        if (p0 == p1) return NBX_OK;
        NBX_CHECK_ARG(d_eri && d_packed);
        return nbx_jk_mx_pack(ctx, nx, nao, p0, p1, d_eri, d_packed);
    }
    const int64_t NP = s4_padded(nao);  // the tiles (p, q) with p < nao of the padded tensor: the others are zero
    if (NP == 0) {
        nbx_set_error("nbx_eri_pack: N = %lld is not covered by the packed J/K kernel", (long long)nao);
        return NBX_E_UNSUPPORTED;
    }
    if (p0 == p1) return NBX_OK;
    NBX_CHECK_ARG(d_eri && d_packed);
    if (const int64_t n8 = m8_padded(nao)) return nbx_jk_m8_pack(ctx, n8, nao, p0, p1, d_eri, d_packed);
    if (m4_padded(nao)) return nbx_jk_m4_pack(ctx, m4_padded(nao), nao, p0, p1, d_eri, d_packed);
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
    if (p0 < 0 || p1 < p0 || p1 > nao || ndm <= 0) return 0;
    if (const int64_t nx = mx_padded(nao))  // (+ the padded densities and J/K of a size that is not an instance)
        return nbx_jk_mx_worksize(nx, p0, p1, ndm) + (nx != nao ? s4_align256((size_t)((1 + 2 * ndm) * nx * nx) * sizeof(double)) : 0);
    const int64_t NP = s4_padded(nao);
    if (NP == 0) return 0;
    if (const int64_t n8 = m8_padded(nao))  // (+ the padded densities and J/K of a size that is not a multiple of four)
        return nbx_jk_m8_worksize(n8, p0, p1, ndm) + (n8 != nao ? s4_align256((size_t)((1 + 2 * ndm) * n8 * n8) * sizeof(double)) : 0);
    if (const int64_t n4 = m4_padded(nao))  // (+ the padded densities and J/K of a size that is not a multiple of four)
        return nbx_jk_m4_worksize(n4, p0, p1, ndm) + (n4 != nao ? s4_align256((size_t)((1 + 2 * ndm) * n4 * n4) * sizeof(double)) : 0);
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
    if (mx_padded(nao)) return 0;  // (jk_mx.hip prepares its table itself: one small launch beside a build of 0.2 .. 2.4 ms)
    if (nbx_jk_m8_covers(nao)) {  // (jk_m8.hip: jk_m4.hip's staging order with its own chunk boundaries when it has four chunks)
        int wl[4];
        nbx_jk_m8_weight_layout(nao, wl);
        return wl[3] ? nbx_jk_m8_weights_bytes(nao) : 0;
    }
    if (nbx_jk_m4_covers(nao)) return nbx_jk_m4_weights_bytes(nao);  // (the same table in jk_m4.hip's staging order)
    if (!s4_supported(nao)) return 0;  // (zero-padded sizes build their table themselves)
    const int NB = s4_nb(nao);
    return (size_t)(NB * NB * s4_lpt(nao) * 128) * sizeof(double);
}

extern "C" int nbx_jk_dts_init(nbx_ctx* ctx, int64_t nao, double* d_dts) {
    NBX_CHECK_ARG(ctx && d_dts && s4_supported(nao) && !mx_padded(nao));
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
    const int64_t NX = mx_padded(nao);
    const int64_t NPAD = NX ? NX : s4_padded(nao);
    if (NPAD == 0) {
        nbx_set_error("nbx_jk_packed: N = %lld is not covered (N <= 288 within %d of a size the kernels have an "
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
    if (NX == nao) return nbx_jk_mx(ctx, nao, p0, p1, d_packed, d_dm, ndm, d_jk, d_work, d_hv, d_fock, d_vhf);
    if (NX) {
        // as the NX x NX problem whose extra rows and columns are zero (tiles with p >= nao are neither stored nor visited)
        char* tail = static_cast<char*>(d_work) + nbx_jk_mx_worksize(NX, p0, p1, ndm);
        double* dm_pad = reinterpret_cast<double*>(tail);
        double* jk_pad = dm_pad + ndm * NX * NX;
        const int64_t tin = ndm * NX * NX, tout = (1 + ndm) * nao * nao;
        hipLaunchKernelGGL(s4_pad_square_kernel, dim3((unsigned)nbx_cdiv(tin, 256)), dim3(256), 0, ctx->stream, d_dm,
                           dm_pad, (int)nao, (int)NX, (int)ndm);
        NBX_LAUNCH_CHECK();
        const int rc = nbx_jk_mx(ctx, NX, p0, p1, d_packed, dm_pad, ndm, jk_pad, d_work, nullptr, nullptr, nullptr);
        if (rc != NBX_OK) return rc;
        hipLaunchKernelGGL(s4_crop_square_kernel, dim3((unsigned)nbx_cdiv(tout, 256)), dim3(256), 0, ctx->stream,
                           jk_pad, d_jk, (int)nao, (int)NX, (int)(1 + ndm));
        NBX_LAUNCH_CHECK();
        if (d_fock != nullptr) return nbx_fock_uhf(ctx, nao, d_hv, 3, nullptr, d_jk, d_fock, d_vhf);
        return NBX_OK;
    }
    if (nbx_jk_m8_covers(nao)) return nbx_jk_m8(ctx, nao, p0, p1, d_packed, d_dm, ndm, d_jk, d_work, d_hv, d_fock, d_vhf, d_dts);
    if (const int64_t n8 = m8_padded(nao)) {
        // (n8 > nao: as the n8 x n8 problem whose extra rows and columns are zero, like the jk_m4.hip branch below)
        char* tail = static_cast<char*>(d_work) + nbx_jk_m8_worksize(n8, p0, p1, ndm);
        double* dm_pad = reinterpret_cast<double*>(tail);
        double* jk_pad = dm_pad + ndm * n8 * n8;
        const int64_t tin = ndm * n8 * n8, tout = (1 + ndm) * nao * nao;
        hipLaunchKernelGGL(s4_pad_square_kernel, dim3((unsigned)nbx_cdiv(tin, 256)), dim3(256), 0, ctx->stream, d_dm,
                           dm_pad, (int)nao, (int)n8, (int)ndm);
        NBX_LAUNCH_CHECK();
        const int rc = nbx_jk_m8(ctx, n8, p0, p1, d_packed, dm_pad, ndm, jk_pad, d_work, nullptr, nullptr, nullptr, nullptr);
        if (rc != NBX_OK) return rc;
        hipLaunchKernelGGL(s4_crop_square_kernel, dim3((unsigned)nbx_cdiv(tout, 256)), dim3(256), 0, ctx->stream,
                           jk_pad, d_jk, (int)nao, (int)n8, (int)(1 + ndm));
        NBX_LAUNCH_CHECK();
        if (d_fock != nullptr) return nbx_fock_uhf(ctx, nao, d_hv, 3, nullptr, d_jk, d_fock, d_vhf);
        return NBX_OK;
    }
    if (nbx_jk_m4_covers(nao)) return nbx_jk_m4(ctx, nao, p0, p1, d_packed, d_dm, ndm, d_jk, d_work, d_hv, d_fock, d_vhf, d_dts);
    if (const int64_t n4 = m4_padded(nao)) {
        // as the n4 x n4 problem whose extra rows and columns are zero (tiles with p >= nao are neither stored nor visited):
        // D padded on the way in, J/K cropped on the way out, the Fock assembly its own launch
        char* tail = static_cast<char*>(d_work) + nbx_jk_m4_worksize(n4, p0, p1, ndm);
        double* dm_pad = reinterpret_cast<double*>(tail);
        double* jk_pad = dm_pad + ndm * n4 * n4;
        const int64_t tin = ndm * n4 * n4, tout = (1 + ndm) * nao * nao;
        hipLaunchKernelGGL(s4_pad_square_kernel, dim3((unsigned)nbx_cdiv(tin, 256)), dim3(256), 0, ctx->stream, d_dm,
                           dm_pad, (int)nao, (int)n4, (int)ndm);
        NBX_LAUNCH_CHECK();
        const int rc = nbx_jk_m4(ctx, n4, p0, p1, d_packed, dm_pad, ndm, jk_pad, d_work, nullptr, nullptr, nullptr, nullptr);
        if (rc != NBX_OK) return rc;
        hipLaunchKernelGGL(s4_crop_square_kernel, dim3((unsigned)nbx_cdiv(tout, 256)), dim3(256), 0, ctx->stream,
                           jk_pad, d_jk, (int)nao, (int)n4, (int)(1 + ndm));
        NBX_LAUNCH_CHECK();
        if (d_fock != nullptr) return nbx_fock_uhf(ctx, nao, d_hv, 3, nullptr, d_jk, d_fock, d_vhf);
        return NBX_OK;
    }
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
    return nbx_jk_sym_reduce(ctx, k1, k2, d_jk + n2, N, p0, np, ndm, t_begin, pl.L, pl.S, d_jk, d_hv, d_fock, d_vhf, 1, 1);
}
