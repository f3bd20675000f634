// libnbx: AO -> MO four-index transform (include/nbx.h "four-index transform").
//
//   (ij|kl) = sum_pqrs C1_pi C2_qj C3_rk C4_sl (pq|rs)          i in [i0, i1)
//
// Four sequential quarter transforms, each one (batched) fp64 MFMA GEMM (gemm.hip):
//   Q1  X1[i,(qrs)]   = C1^T (ni x N)   . ERI  (N x N^3)                 2 ni N^4      flop
//   Q2  X2[i][j,(rs)] = C2^T (n2 x N)   . X1[i] (N x N^2)    batch ni     2 ni n2 N^3
//   Q3  W[(ijr),l]    = X2 (ni n2 N x N) . C4 (N x n4)                    2 ni n2 n4 N^2
//   Q4  out[ij][k,l]  = C3^T (n3 x N)   . W[ij] (N x n4)     batch ni*n2  2 ni n2 n3 n4 N
// (s is contracted before r: the batched product then has n4 columns -- one full 128-wide tile
// for n = 128 -- instead of N = 148 columns, of which a second 128-wide tile would be 84 % empty.)
// No permutational symmetry is used (SURVEY.md section 8d flop count).  The contiguous ERI
// index s stays the fastest index of every intermediate, so all GEMM operand loads are
// coalesced.  The outer MO index i is the multi-GPU shard axis: a rank transforms its own
// i-slab against the full ERI and the slabs are all-gathered by the host.
#include "nbx_common.h"

namespace {
size_t align256(size_t x) { return (x + 255) & ~(size_t)255; }

struct Ao2moPlan {
    size_t a_doubles, b_doubles;
};

Ao2moPlan plan(int64_t N, int64_t ni, int64_t n2, int64_t n4) {
    Ao2moPlan p;
    const size_t x1 = (size_t)ni * N * N * N;
    const size_t x3 = (size_t)ni * n2 * N * n4;  // W
    p.a_doubles = x1 > x3 ? x1 : x3;
    p.b_doubles = (size_t)ni * n2 * N * N;
    return p;
}
}  // namespace

extern "C" size_t nbx_ao2mo_worksize(int64_t nao, int64_t ni, int64_t n2, int64_t n3, int64_t n4) {
    (void)n3;
    if (nao <= 0 || ni <= 0 || n2 <= 0 || n4 < 0) return 0;
    const Ao2moPlan p = plan(nao, ni, n2, n4);
    return align256(p.a_doubles * sizeof(double)) + align256(p.b_doubles * sizeof(double));
}

extern "C" int nbx_ao2mo(nbx_ctx* ctx, int64_t nao, const double* d_eri, const double* d_c1, int64_t n1,
                         int64_t i0, int64_t i1, const double* d_c2, int64_t n2, const double* d_c3, int64_t n3,
                         const double* d_c4, int64_t n4, double* d_out, void* d_work, size_t work_bytes) {
    return nbx_ao2mo_pair(ctx, nao, d_eri, d_c1, n1, i0, i1, d_c2, n2, d_c3, n3, d_c4, n4, d_out, nullptr, 0, nullptr,
                          0, nullptr, d_work, work_bytes);
}

extern "C" size_t nbx_ao2mo_pair_worksize(int64_t nao, int64_t ni, int64_t n2, int64_t n4, int64_t n6) {
    return nbx_ao2mo_worksize(nao, ni, n2, 0, n4 > n6 ? n4 : n6);
}

extern "C" int nbx_ao2mo_pair(nbx_ctx* ctx, int64_t nao, const double* d_eri, const double* d_c1, int64_t n1,
                              int64_t i0, int64_t i1, const double* d_c2, int64_t n2, const double* d_c3, int64_t n3,
                              const double* d_c4, int64_t n4, double* d_out, const double* d_c5, int64_t n5,
                              const double* d_c6, int64_t n6, double* d_out2, void* d_work, size_t work_bytes) {
    const bool pair = d_out2 != nullptr;
    NBX_CHECK_ARG(ctx != nullptr && i0 >= 0 && i1 >= i0);
    if (i1 == i0) return NBX_OK;  // an empty outer-index slab has no output storage
    NBX_CHECK_ARG(d_eri && d_c1 && d_c2 && d_c3 && d_c4 && d_out);
    NBX_CHECK_ARG(!pair || (d_c5 && d_c6 && n5 > 0 && n6 > 0));
    if (!pair) n5 = 0;
    NBX_CHECK_ARG(nao > 0 && n1 > 0 && n2 > 0 && n3 > 0 && n4 > 0);
    NBX_CHECK_ARG(i0 >= 0 && i1 >= i0 && i1 <= n1);
    const int64_t ni = i1 - i0;
    if (ni == 0) return NBX_OK;
    const int64_t N = nao;
    if (N * N * N >= (1ll << 31)) {
        nbx_set_error("nbx_ao2mo: dense path needs N^3 < 2^31 (N=%lld)", (long long)N);
        return NBX_E_UNSUPPORTED;
    }
    const size_t need = nbx_ao2mo_pair_worksize(N, ni, n2, n4, pair ? n6 : 0);
    if (d_work == nullptr || work_bytes < need) {
        nbx_set_error("nbx_ao2mo: workspace %zu < %zu bytes", work_bytes, need);
        return NBX_E_NOMEM;
    }
    const Ao2moPlan p = plan(N, ni, n2, (pair && n6 > n4) ? n6 : n4);
    double* bufA = static_cast<double*>(d_work);
    double* bufB = reinterpret_cast<double*>(static_cast<char*>(d_work) + align256(p.a_doubles * sizeof(double)));
    const int64_t N2 = N * N, N3 = N2 * N;
    int rc;
    nbx_prof_scope prof_all(ctx, NBX_PROF_AO2MO);
    // Q1
    {
        nbx_prof_scope prof_q1(ctx, NBX_PROF_AO2MO_Q1);
        rc = nbx_gemm(ctx, 'T', 'N', ni, N3, N, 1.0, d_c1 + i0, n1, 0, d_eri, N3, 0, 0.0, bufA, N3, 0, 1);
    }
    if (rc != NBX_OK) return rc;
    // Q2
    rc = nbx_gemm(ctx, 'T', 'N', n2, N2, N, 1.0, d_c2, n2, 0, bufA, N2, N3, 0.0, bufB, N2, n2 * N2, ni);
    if (rc != NBX_OK) return rc;
    // Q3: s -> l
    rc = nbx_gemm(ctx, 'N', 'N', ni * n2 * N, n4, N, 1.0, bufB, N, 0, d_c4, n4, 0, 0.0, bufA, n4, 0, 1);
    if (rc != NBX_OK) return rc;
    // Q4: r -> k, batched over (i,j)
    rc = nbx_gemm(ctx, 'T', 'N', n3, n4, N, 1.0, d_c3, n3, 0, bufA, n4, N * n4, 0.0, d_out, n4, n3 * n4, ni * n2);
    if (rc != NBX_OK || !pair) return rc;
    // second tensor of the pair: quarters 3 and 4 again from the X2 still sitting in bufB
    rc = nbx_gemm(ctx, 'N', 'N', ni * n2 * N, n6, N, 1.0, bufB, N, 0, d_c6, n6, 0, 0.0, bufA, n6, 0, 1);
    if (rc != NBX_OK) return rc;
    return nbx_gemm(ctx, 'T', 'N', n5, n6, N, 1.0, d_c5, n5, 0, bufA, n6, N * n6, 0.0, d_out2, n6, n5 * n6, ni * n2);
}

// ---------------------------------------------------------------------------------------------
// The same transform(s) when C1 and C2 are THE SAME matrix and the whole outer range is done here:
// (ij|kl) = (ji|kl), so quarters 3 and 4 run on the pairs j <= i only (a "triangular batch" of
// the GEMM: batch entry i has (i+1) N rows, results stored compactly) and the finished (n3 x n4)
// blocks are copied to both (i,j) and (j,i).  All three spin blocks of the unrestricted
// Hamiltonian qualify (nbed/ham_builder.py:127-133: C1 = C2 in every ao2mo.kernel call).  Not
// bit-identical to nbx_ao2mo_pair: there (i,j) and (j,i) are computed separately and agree to
// rounding; here they are equal by construction.  Slabs (multi-GPU) use nbx_ao2mo_pair.
namespace {
// out[(p,q)][T(r,s)] = eri[p][q][r][s], s <= r, T = r(r+1)/2 + s
__global__ __launch_bounds__(256) void eri_pack_rs_kernel(const double* __restrict__ eri, double* __restrict__ out, int N) {
    const int64_t pq = blockIdx.x, nt = (int64_t)N * (N + 1) / 2;
    const double* src = eri + pq * N * N;
    double* dst = out + pq * nt;
    // 16 rows per workgroup (grid.y), one 64-lane wave per row at a time
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    for (int rr = wave; rr < 16; rr += 4) {
        const int r = blockIdx.y * 16 + rr;
        if (r >= N) break;
        for (int s = lane; s <= r; s += 64) dst[(int64_t)r * (r + 1) / 2 + s] = src[(int64_t)r * N + s];
    }
}

struct Ao2moSymPlan {
    size_t a_doubles, b_doubles, c_doubles;
};
Ao2moSymPlan sym_plan(int64_t N, int64_t n, int64_t n4, int64_t n6) {
    Ao2moSymPlan p;
    const size_t npairs = (size_t)(n * (n + 1) / 2);
    const size_t x1 = (size_t)n * N * N * N, w = npairs * N * (size_t)n4;
    p.a_doubles = x1 > w ? x1 : w;
    p.b_doubles = (size_t)n * n * N * N;
    p.c_doubles = npairs * N * (size_t)n6;
    return p;
}
}  // namespace

extern "C" size_t nbx_ao2mo_pair_sym_rs_worksize(int64_t nao, int64_t n, int64_t n4, int64_t n6) {
    if (nao <= 0 || n <= 0 || n4 <= 0 || n6 < 0) return 0;
    const Ao2moSymPlan p = sym_plan(nao, n, n4, n6);
    // + the packed X2 (n^2 N(N+1)/2); the packed X1 is smaller than the dense one the plan sizes bufA for
    return align256(p.a_doubles * sizeof(double)) + align256(p.b_doubles * sizeof(double)) +
           align256(p.c_doubles * sizeof(double)) + align256((size_t)(n * n * (nao * (nao + 1) / 2)) * sizeof(double));
}

extern "C" size_t nbx_ao2mo_pair_sym_worksize(int64_t nao, int64_t n, int64_t n4, int64_t n6) {
    if (nao <= 0 || n <= 0 || n4 <= 0 || n6 < 0) return 0;
    const Ao2moSymPlan p = sym_plan(nao, n, n4, n6);
    return align256(p.a_doubles * sizeof(double)) + align256(p.b_doubles * sizeof(double)) +
           align256(p.c_doubles * sizeof(double));
}

static int ao2mo_pair_sym_impl(nbx_ctx* ctx, int64_t nao, const double* d_eri, bool rs_packed, const double* d_c12,
                               int64_t n, const double* d_c3, int64_t n3, const double* d_c4, int64_t n4, double* d_out,
                               const double* d_c5, int64_t n5, const double* d_c6, int64_t n6, double* d_out2,
                               void* d_work, size_t work_bytes);

extern "C" int nbx_ao2mo_pair_sym(nbx_ctx* ctx, int64_t nao, const double* d_eri, const double* d_c12, int64_t n,
                                  const double* d_c3, int64_t n3, const double* d_c4, int64_t n4, double* d_out,
                                  const double* d_c5, int64_t n5, const double* d_c6, int64_t n6, double* d_out2,
                                  void* d_work, size_t work_bytes) {
    return ao2mo_pair_sym_impl(ctx, nao, d_eri, false, d_c12, n, d_c3, n3, d_c4, n4, d_out, d_c5, n5, d_c6, n6, d_out2,
                               d_work, work_bytes);
}

extern "C" int nbx_ao2mo_pair_sym_rs(nbx_ctx* ctx, int64_t nao, const double* d_eri_rs, const double* d_c12, int64_t n,
                                     const double* d_c3, int64_t n3, const double* d_c4, int64_t n4, double* d_out,
                                     const double* d_c5, int64_t n5, const double* d_c6, int64_t n6, double* d_out2,
                                     void* d_work, size_t work_bytes) {
    return ao2mo_pair_sym_impl(ctx, nao, d_eri_rs, true, d_c12, n, d_c3, n3, d_c4, n4, d_out, d_c5, n5, d_c6, n6, d_out2,
                               d_work, work_bytes);
}

extern "C" size_t nbx_eri_rs_bytes(int64_t nao) {
    return nao > 0 ? (size_t)(nao * nao * (nao * (nao + 1) / 2)) * sizeof(double) : 0;
}

extern "C" int nbx_eri_pack_rs(nbx_ctx* ctx, int64_t nao, const double* d_eri, double* d_out) {
    NBX_CHECK_ARG(ctx && d_eri && d_out && nao > 0 && nao * nao < (1ll << 31));
    hipLaunchKernelGGL(eri_pack_rs_kernel, dim3((unsigned)(nao * nao), (unsigned)nbx_cdiv(nao, 16)), dim3(256), 0,
                       ctx->stream, d_eri, d_out, (int)nao);
    NBX_LAUNCH_CHECK();
    return NBX_OK;
}

static int ao2mo_pair_sym_impl(nbx_ctx* ctx, int64_t nao, const double* d_eri, bool rs_packed, const double* d_c12,
                               int64_t n, const double* d_c3, int64_t n3, const double* d_c4, int64_t n4, double* d_out,
                               const double* d_c5, int64_t n5, const double* d_c6, int64_t n6, double* d_out2,
                               void* d_work, size_t work_bytes) {
    const bool pair = d_out2 != nullptr;
    NBX_CHECK_ARG(ctx && d_eri && d_c12 && d_c3 && d_c4 && d_out);
    NBX_CHECK_ARG(nao > 0 && n > 0 && n3 > 0 && n4 > 0);
    NBX_CHECK_ARG(!pair || (d_c5 && d_c6 && n5 > 0 && n6 > 0));
    if (n * (n + 1) / 2 > 65535) {  // one grid dimension holds the pairs
        nbx_set_error("nbx_ao2mo_pair_sym: n = %lld gives more than 65535 pairs; use nbx_ao2mo_pair", (long long)n);
        return NBX_E_UNSUPPORTED;
    }
    const int64_t N = nao;
    if (N * N * N >= (1ll << 31)) {
        nbx_set_error("nbx_ao2mo_pair_sym: dense path needs N^3 < 2^31 (N=%lld)", (long long)N);
        return NBX_E_UNSUPPORTED;
    }
    const size_t need = rs_packed ? nbx_ao2mo_pair_sym_rs_worksize(N, n, n4, pair ? n6 : 0)
                                  : nbx_ao2mo_pair_sym_worksize(N, n, n4, pair ? n6 : 0);
    if (d_work == nullptr || work_bytes < need) {
        nbx_set_error("nbx_ao2mo_pair_sym: workspace %zu < %zu bytes", work_bytes, need);
        return NBX_E_NOMEM;
    }
    const Ao2moSymPlan p = sym_plan(N, n, n4, pair ? n6 : 0);
    char* base = static_cast<char*>(d_work);
    double* bufA = reinterpret_cast<double*>(base);
    double* bufB = reinterpret_cast<double*>(base + align256(p.a_doubles * sizeof(double)));
    double* bufC = reinterpret_cast<double*>(base + align256(p.a_doubles * sizeof(double)) +
                                             align256(p.b_doubles * sizeof(double)));
    const int64_t N2 = N * N, N3 = N2 * N;
    int rc;
    nbx_prof_scope prof_all(ctx, NBX_PROF_AO2MO);
    if (rs_packed) {
        // (pq|rs) = (pq|sr): the integrals come with (r, s <= r) packed (nbx_eri_pack_rs), quarters 1
        // and 2 run on N(N+1)/2 columns instead of N^2, and quarter 3 reads the packed X2 through the
        // GEMM's symmetric-packed A operand (no unpacked copy is ever made)
        const int64_t Nt = N * (N + 1) / 2;
        double* bufT = reinterpret_cast<double*>(reinterpret_cast<char*>(bufC) + align256(p.c_doubles * sizeof(double)));
        {
            nbx_prof_scope prof_q1(ctx, NBX_PROF_AO2MO_Q1);
            rc = nbx_gemm(ctx, 'T', 'N', n, N * Nt, N, 1.0, d_c12, n, 0, d_eri, N * Nt, 0, 0.0, bufA, N * Nt, 0, 1);
        }
        if (rc != NBX_OK) return rc;
        rc = nbx_gemm(ctx, 'T', 'N', n, Nt, N, 1.0, d_c12, n, 0, bufA, Nt, N * Nt, 0.0, bufT, Nt, n * Nt, n);
        if (rc != NBX_OK) return rc;
        // quarter 3 reads the packed X2 directly (symmetric-packed A operand of the GEMM)
        rc = nbx_gemm_tri(ctx, N, 0, n, n4, N, bufT, N, n * Nt, d_c4, n4, bufA, n4, N);
        if (rc != NBX_OK) return rc;
        if (pair) {
            rc = nbx_gemm_tri(ctx, N, 0, n, n6, N, bufT, N, n * Nt, d_c6, n6, bufC, n6, N);
            if (rc != NBX_OK) return rc;
        }
    } else {
        {
            nbx_prof_scope prof_q1(ctx, NBX_PROF_AO2MO_Q1);
            rc = nbx_gemm(ctx, 'T', 'N', n, N3, N, 1.0, d_c12, n, 0, d_eri, N3, 0, 0.0, bufA, N3, 0, 1);
        }
        if (rc != NBX_OK) return rc;
        rc = nbx_gemm(ctx, 'T', 'N', n, N2, N, 1.0, d_c12, n, 0, bufA, N2, N3, 0.0, bufB, N2, n * N2, n);
        if (rc != NBX_OK) return rc;
    }
    // Q3 on the pairs j <= i: W[(i,j),r,l] = sum_s X2[i,j,r,s] C4[s,l]
    if (!rs_packed) {
        rc = nbx_gemm_tri(ctx, N, 0, n, n4, N, bufB, N, n * N2, d_c4, n4, bufA, n4);
        if (rc != NBX_OK) return rc;
        if (pair) {
            rc = nbx_gemm_tri(ctx, N, 0, n, n6, N, bufB, N, n * N2, d_c6, n6, bufC, n6);
            if (rc != NBX_OK) return rc;
        }
    }
    // Q4 per pair, each (n3 x n4) result stored at (i,j) and (j,i) by the GEMM's epilogue
    rc = nbx_gemm_pair_scatter(ctx, n, n3, n4, N, d_c3, n3, bufA, n4, N * n4, d_out);
    if (rc != NBX_OK || !pair) return rc;
    return nbx_gemm_pair_scatter(ctx, n, n5, n6, N, d_c5, n5, bufC, n6, N * n6, d_out2);
}
