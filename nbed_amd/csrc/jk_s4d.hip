// libnbx: the packed J/K kernel of jk_s4.hip with the tiles streamed STRAIGHT INTO LDS
// (global_load_lds_dwordx4: lane l of a wave lands its 16 bytes at the wave's LDS base + 16 l, which is exactly
// the staging-slot layout of jk_s4_layout.h), for the NB = 4 instances.
//
// jk_s4_kernel moves a chunk global -> registers -> LDS: 2 x LPT double2 staging registers per thread (48 VGPRs at
// LPT = 6), LPT ds_write_b128 per chunk, and the J partial taken from the staged registers.  Here the loads carry
// no registers: chunk c + 2 is in flight into a ring of three LDS buffers while chunk c is walked, the J partial
// reads the thread's own slots back from LDS.  Same tile format, Dtot' table, partial buffers, reduction and
// work distribution (s4_plan) as jk_s4_kernel; same sums in the same order: bit-identical results.
#include "nbx_common.h"
#include "jk_s4_layout.h"
#include "jk_s4_device.h"

// (m0 is named as a clobber of the LDS-DMA asm below; clang calls that a reserved register)
#pragma clang diagnostic ignored "-Winline-asm"

namespace {

typedef __attribute__((address_space(3))) void* s4d_lds_vp;

template <int NDM, int NB, int LPT, int WV, int NBUF, bool DT_REG>
__global__ __launch_bounds__(NB * 64) __attribute__((amdgpu_waves_per_eu(WV, WV))) void jk_s4d_kernel(
    const double* __restrict__ eri, const double* __restrict__ dm, const double* __restrict__ dts,
    double* __restrict__ jfull, double* __restrict__ kpart1, double* __restrict__ kpart2, int N, int p0, int np,
    int64_t t_begin, int64_t t_end, int L, int S) {
    constexpr int NCH = NB, BUFD = LPT * NB * 128, PDD = NBUF - 1;  // chunks in flight ahead of the walk
    static_assert(NBUF == 2 || NBUF == 3, "ring of two or three chunk buffers");
    extern __shared__ __attribute__((aligned(16))) double smem[];  // buf[NBUF][BUFD] | slack[128] | jred[2][NB]
    double* slack = smem + NBUF * BUFD;
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

    const unsigned lane16 = 16u * (unsigned)lane;
    // chunk `ch` of the tile at `tp` into ring buffer `b`: LPT loads of 64 x 16 bytes per wave, no registers
    auto issue = [&](const double* tp, int ch, int b) {
        const double* cp = tp + chunk_off(ch);
        const int ne = chunk_len(ch);
        double* buf = smem + b * BUFD;
#pragma unroll
        for (int k = 0; k < LPT; ++k) {
            const int ps = s4_slot_start(ne, LPT, w, k);
            const double* src = ps < 0 ? dts : cp + 2 * ps;  // wave-uniform; the lane's 16 bytes: + lane16
            double* dst = ps < 0 ? slack : buf + 2 * ps;     // wave-uniform: the lane offset is the hardware's
            // (inline asm, not __builtin_amdgcn_global_load_lds: the compiler cannot tell which LDS reads a
            // pending load aliases and puts s_waitcnt vmcnt(0) in front of the next ds_read -- the prefetch of two
            // chunks would be waited for at once; the waits are the explicit ones in the chunk loop)
            const unsigned lds_a = (unsigned)(size_t)(s4d_lds_vp)dst;
            asm volatile("s_mov_b32 m0, %2\n\tglobal_load_lds_dwordx4 %0, %1 nt"
                         :
                         : "v"(lane16), "s"(src), "s"(lds_a)
                         : "memory", "m0");
        }
    };
    // the thread's Dtot' entries (the same for every tile): in registers (DT_REG), or fetched chunk by chunk from the
    // L2-resident table behind the chunk's own loads (one chunk in flight only: NBUF == 2)
    static_assert(DT_REG || NBUF == 2, "per-chunk Dtot' loads are waited for with the chunk");
    double2 dt[DT_REG ? NCH : 1][LPT];
    const double* dts_l = dts;
    auto issue_dt = [&](double2(&d)[LPT], int ch) {
        const double* cp = dts_l + ((ch * NB + w) * LPT) * 128;
#pragma unroll
        for (int k = 0; k < LPT; ++k) d[k] = *reinterpret_cast<const double2*>(cp + k * 128 + 2 * lane);
    };
    if (DT_REG) {
#pragma unroll
        for (int ch = 0; ch < NCH; ++ch) issue_dt(dt[ch], ch);
    }

    const double* tile = eri + (T - t_begin) * g.M;
    int b = 0;  // ring position of the chunk about to be walked
    issue(tile, 0, 0);
    if (PDD == 2) issue(tile, 1, 1);
    double2 dnext[DT_REG ? 1 : LPT];  // Dtot' of the chunk in flight
    if constexpr (!DT_REG) issue_dt(dnext, 0);

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
        const double* tile_next = T + 1 < T_end ? tile + g.M : tile;  // (the last tile re-reads itself: no tail case)
        double kq[NDM];
#pragma unroll
        for (int x = 0; x < NDM; ++x) kq[x] = 0.0;
        const double* dq = dm + (int64_t)q * N;
        const double* dp = dm + (int64_t)p * N;
        double jacc = 0.0;
#pragma unroll
        for (int ch = 0; ch < NCH; ++ch) {
            double* buf = smem + b * BUFD;
            // this wave's loads of chunk c have landed once at most the LPT loads of chunk c + 1 (and whatever stores
            // were issued since) are outstanding; after the barrier that holds for every wave's loads, and every wave
            // has finished walking chunk c - 1, whose buffer chunk c + 2 now streams into
            constexpr int AHEAD = (PDD - 1) * LPT;  // loads that may stay outstanding
            __builtin_amdgcn_s_waitcnt(0x0F70 | (AHEAD & 15) | ((AHEAD >> 4) << 14));
            __syncthreads();
            if constexpr (!DT_REG) {
#pragma unroll
                for (int k = 0; k < LPT; ++k) dt[0][k] = dnext[k];
                asm volatile("" : "+s"(dts_l));  // (or the table's loads are hoisted out of the tile loop: DT_REG)
            }
            {
                const int b2 = b + PDD >= NBUF ? b + PDD - NBUF : b + PDD;
                if (ch + PDD < NCH) issue(tile, ch + PDD, b2);
                else issue(tile_next, ch + PDD - NCH, b2);
                if constexpr (!DT_REG) issue_dt(dnext, (ch + 1) % NCH);
            }
            if (ch == 0 && pj >= 0 && tid == 0) store_j(par ^ 1, pj, qj);
            // J partial from the thread's own slots (weights of repeated / overhanging slots are zero in Dtot')
#pragma unroll
            for (int k = 0; k < LPT; ++k) {
                const int ps = s4_slot_start(chunk_len(ch), LPT, w, k);
                const double2 st = *reinterpret_cast<const double2*>((ps < 0 ? slack : buf + 2 * ps) + 2 * lane);
                jacc = fma(st.x, dt[DT_REG ? ch : 0][k].x, fma(st.y, dt[DT_REG ? ch : 0][k].y, jacc));
            }
            // ---- the walk: s steps, element Lsym[trow][u*s + c]
            if (ch == 0) {
                s4_walk<NDM, 0>(buf + w * g.tri, 0, il, tri_il, dq + w * s, dp + w * s, n2, s, kp, kq);
            } else {
                const int u = w ^ ch;
                const double* rect = buf + s4_slot(min(w, u), ch) * s * ls;
                if (w > u) s4_walk<NDM, 1>(rect + il * ls, 1, il, tri_il, dq + u * s, dp + u * s, n2, s, kp, kq);
                else s4_walk<NDM, 2>(rect + il, ls, il, tri_il, dq + u * s, dp + u * s, n2, s, kp, kq);
            }
            b = b + 1 >= NBUF ? 0 : b + 1;
        }
        // J partial of this tile (summed by thread 0 after the next barrier)
        jacc = nbx_wave_sum(jacc);
        if (lane == 0) jred[par * NB + w] = jacc;
        pj = p;
        qj = q;
        par ^= 1;
        if (q < p && live) {
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
    __builtin_amdgcn_s_waitcnt(0x0F70);  // the two chunks still streaming into LDS: land before the workgroup ends
    __syncthreads();
    if (tid == 0) store_j(par ^ 1, pj, qj);
}

}  // namespace

// variant 1: ring of three buffers (two chunks in flight), two workgroups per CU; 2: two buffers (one chunk in
// flight), two workgroups per CU; 3: two buffers, THREE workgroups per CU (the staging registers are gone)
size_t nbx_jk_s4d_lds_bytes(int NB, int lpt, int variant) {
    return (size_t)((variant == 1 ? 3 : 2) * lpt * NB * 64 * 2 + 128 + 2 * NB) * sizeof(double);
}
int nbx_jk_s4d_per_cu(int variant) { return variant == 3 ? 3 : 2; }

bool nbx_jk_s4d_covers(int NB, int lpt) { return NB == 4 && lpt == 6; }

int nbx_jk_s4d_launch(nbx_ctx* ctx, int variant, int64_t N, int64_t p0, int64_t np, int64_t ndm, int lpt,
                      const double* d_packed, const double* d_dm, const double* d_dts, double* d_j, double* k1, double* k2,
                      int64_t t_begin, int64_t t_end, int wgs, int L, int S) {
    NBX_CHECK_ARG(nbx_jk_s4d_covers(4, lpt) && (ndm == 1 || ndm == 2) && variant >= 1 && variant <= 3);
    const size_t lds = nbx_jk_s4d_lds_bytes(4, lpt, variant);
#define NBX_S4D_GO(NDM_, WV_, NBUF_, DT_)                                                                                 \
    do {                                                                                                              \
        static bool attr_set = false;                                                                                 \
        if (!attr_set) {                                                                                              \
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&jk_s4d_kernel<NDM_, 4, 6, WV_, NBUF_, DT_>),           \
                                      hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);                        \
            attr_set = true;                                                                                          \
        }                                                                                                             \
        hipLaunchKernelGGL((jk_s4d_kernel<NDM_, 4, 6, WV_, NBUF_, DT_>), dim3((unsigned)wgs), dim3(256), lds, ctx->stream,  \
                           d_packed, d_dm, d_dts, d_j, k1, k2, (int)N, (int)p0, (int)np, t_begin, t_end, L, S);       \
    } while (0)
#define NBX_S4D_NDM(WV_, NBUF_, DT_)               \
    do {                                           \
        if (ndm == 2) NBX_S4D_GO(2, WV_, NBUF_, DT_); \
        else NBX_S4D_GO(1, WV_, NBUF_, DT_);       \
    } while (0)
    if (variant == 1) NBX_S4D_NDM(2, 3, true);
    else if (variant == 2) NBX_S4D_NDM(2, 2, true);
    else NBX_S4D_NDM(3, 2, false);
#undef NBX_S4D_NDM
#undef NBX_S4D_GO
    NBX_LAUNCH_CHECK();
    return NBX_OK;
}
