// Device pieces of the packed J/K kernels shared by jk_s4.hip (4-fold tiles) and jk_p8.hip (8-fold,
// truncated tiles): the non-temporal 16-byte load, the inverse of the tile layout, and the walk of
// one LDS-resident chunk.
#pragma once
#include "jk_s4_layout.h"
#include "nbx_common.h"

int nbx_jk_s4_dtot(nbx_ctx* ctx, int64_t N, const double* d_dm, int64_t ndm, double* d_dts);  // jk_s4.hip

namespace {

__device__ __forceinline__ double2 s4_ldnt(const double* p) {
    typedef double nbx_d2 __attribute__((ext_vector_type(2)));
    const nbx_d2 t = __builtin_nontemporal_load(reinterpret_cast<const nbx_d2*>(p));
    return make_double2(t.x, t.y);
}

// (a, b) of the entry at offset f of chunk ch; false for a pad
__device__ __forceinline__ bool s4_unflat(const S4Geom& g, int ch, int f, int& a, int& b) {
    if (ch == 0) {
        if (f >= g.NB * g.tri) return false;
        const int I = f / g.tri, rem = f - I * g.tri, ai = s4_tri_row(rem);
        a = I * g.s + ai;
        b = I * g.s + rem - (int)s4_tri(ai);
        return true;
    }
    const int slot = f / (g.s * g.ls), rem = f - slot * g.s * g.ls, ai = rem / g.ls, bi = rem - ai * g.ls;
    if (slot >= g.NB / 2 || bi >= g.s) return false;
    int hb = 0;
    while ((ch >> (hb + 1)) != 0) ++hb;
    const int lo = ch == 3 ? 1 - slot : ((slot >> hb) << (hb + 1)) | (slot & ((1 << hb) - 1));  // inverse of s4_slot
    a = (lo ^ ch) * g.s + ai;
    b = lo * g.s + bi;
    return true;
}

// The walk of one chunk: s steps, thread-private element tv = Lsym[trow][u*s + c] from LDS and the
// wave-uniform density values D_q[u*s + c], D_p[u*s + c] through scalar loads.  A wavefront issues
// at most one instruction every four cycles, whatever its kind, so the walk costs what its
// instruction count costs: the scalar loads are written as s_load_dwordx8 with one running SGPR
// offset (no pointer arithmetic per load), rows of the row side are read with immediate offsets.
// LDS and scalar loads share one counter and scalar loads return out of order, so a wait for
// either is a wait for everything: the loop is software pipelined by hand in groups of four steps
// -- wait, issue the next group's loads, then do this group's FMAs -- with two register sets.
// (The compiler does not track the inline-asm loads: every use below follows an explicit wait.)
// The last group starts at s - 4 and masks the steps an earlier group has done.
// KIND 0: diagonal triangle (per-lane row / column select), 1: rectangle, row side (consecutive
// doubles), 2: rectangle, column side (stride `step` doubles).
typedef double s4_v4d __attribute__((ext_vector_type(4)));

template <int NDM, int KIND>
__device__ __forceinline__ void s4_walk(const double* lb, int step, int il, int tri_il, const double* xq,
                                        const double* xp, int64_t n2, int s, double (&kp)[NDM], double (&kq)[NDM]) {
    const int ng = (s + 3) >> 2;
    const double* xqs[NDM];
    const double* xps[NDM];
#pragma unroll
    for (int x = 0; x < NDM; ++x) {
        xqs[x] = xq + x * n2;
        xps[x] = xp + x * n2;
    }
    const char* lbc = reinterpret_cast<const char*>(lb);
    // diagonal triangle: LDS byte addresses of the lane's row (L[il][0]) and of its column's head
    // (L[0][il] if the rows above il were full); opaque, so that the uniform parts of the per-step
    // addresses stay in scalar registers
    typedef __attribute__((address_space(3))) const char* s4_lds_cp;
    typedef __attribute__((address_space(3))) const double* s4_lds_dp;
    int row_a = (int)(size_t)(s4_lds_cp)lbc + tri_il * 8, col_a = (int)(size_t)(s4_lds_cp)lbc + il * 8;
    if (KIND == 0) {
        asm volatile("" : "+v"(row_a));
        asm volatile("" : "+v"(col_a));
    }
    auto load = [&](int g, s4_v4d(&a)[NDM], s4_v4d(&b)[NDM], double(&t)[4]) {
        const int c0 = min(4 * g, s - 4);
        const int off = c0 * 8;
#pragma unroll
        for (int x = 0; x < NDM; ++x) {
            asm volatile("s_load_dwordx8 %0, %1, %2" : "=s"(a[x]) : "s"(xqs[x]), "s"(off));
            asm volatile("s_load_dwordx8 %0, %1, %2" : "=s"(b[x]) : "s"(xps[x]), "s"(off));
        }
        if (KIND == 0) {
            // row part (c <= il): L[il][c] at tri(il) + c; column part: L[c][il] at tri(c) + il
            const int rb = row_a + off;
            int tric = (int)(((unsigned)c0 * (unsigned)(c0 + 1)) >> 1);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int sj = (tric - j) * 8;  // minus j: the read below adds 8 j to both forms
                const int ab = (il >= c0 + j) ? rb : col_a + sj;
                t[j] = *(s4_lds_dp)(size_t)(ab + 8 * j);
                tric += c0 + j + 1;
            }
        } else if (KIND == 1) {
#pragma unroll
            for (int j = 0; j < 4; ++j) t[j] = *reinterpret_cast<const double*>(lbc + off + 8 * j);
        } else {
            const int sb = step * 8;
            int ab = c0 * sb;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                t[j] = *reinterpret_cast<const double*>(lbc + ab);
                ab += sb;
            }
        }
    };
    auto fmas = [&](const s4_v4d(&a)[NDM], const s4_v4d(&b)[NDM], const double(&t)[4], int skip) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const double tj = j >= skip ? t[j] : 0.0;
#pragma unroll
            for (int x = 0; x < NDM; ++x) {
                kp[x] = fma(tj, a[x][j], kp[x]);
                kq[x] = fma(tj, b[x][j], kq[x]);
            }
        }
    };
#define S4_WAIT_THEN(LOADS)                  \
    __builtin_amdgcn_s_waitcnt(0xC07F);      \
    __builtin_amdgcn_sched_barrier(0);       \
    LOADS;                                   \
    __builtin_amdgcn_sched_barrier(0)
    s4_v4d a0[NDM], b0[NDM], a1[NDM], b1[NDM];
    double t0[4], t1[4];
    load(0, a0, b0, t0);
    int g = 0;
    for (; g + 2 < ng; g += 2) {
        S4_WAIT_THEN(load(g + 1, a1, b1, t1));
        fmas(a0, b0, t0, 0);
        S4_WAIT_THEN(load(g + 2, a0, b0, t0));
        fmas(a1, b1, t1, 0);
    }
    const int skip = 4 * ng - s;  // steps of the last group that belong to the one before it
    if (g + 1 < ng) {  // two groups left: set 0 holds a full one
        S4_WAIT_THEN(load(g + 1, a1, b1, t1));
        fmas(a0, b0, t0, 0);
        S4_WAIT_THEN((void)0);
        fmas(a1, b1, t1, skip);
    } else {
        S4_WAIT_THEN((void)0);
        fmas(a0, b0, t0, skip);
    }
#undef S4_WAIT_THEN
}

}  // namespace
