// libnbx: AO -> MO four-index transform of the SYNTHETIC (pq|rs) without ever storing it
// (include/nbx.h "streamed transform"; the N_AO = 2000 configuration of BASELINE.json).
//
//   out[i,j,k,l] = sum_{r in [r0,r1)} sum_{pqs} C1_pi C2_qj C3_rk C4_sl (pq|rs)
//
// The tensor is produced slab by slab from the counter hash (synth.hip) and consumed at once.
// (pq|rs) = (pq|sr) is used: only the pairs s <= r are generated and half-transformed, which
// halves quarters 1 and 2 -- 94 % of the work at N = 2000, n = 128 -- (PySCF's ao2mo, which the
// reference calls, exploits the same symmetry); the pair (r, s < r) then feeds both
// C3[r,k] C4[s,l] and C3[s,k] C4[r,l]:
//   for r:                                                        flop (summed over r and s <= r)
//     for s-chunks:  Y[s] = C1^T M[s]   (n1 x N)      GEMM       n1 N^4       M[s][p,q] = (pq|rs)
//                                                     generated in the GEMM's B-operand registers
//                    Z[s] = Y[s] C2     (n1 x n2)     GEMM       n1 n2 N^3
//     U[r] = C4[:r+1]^T Z[:r+1]   (n4 x n1 n2)        GEMM       n1 n2 n4 N^2
//     W[r] = C3[:r]^T   Z[:r]     (n3 x n1 n2)        GEMM       n1 n2 n3 N^2
//   every RB r's:  acc[k][l]   += C3[rb]^T U[rb]      GEMM       2 n1 n2 n3 n4 N
//                  acc[k][l]   += C4[rb]^T W[rb][k]   GEMM (batched over k)   2 n1 n2 n3 n4 N
//   out = acc^T   ((k,l),(i,j)) -> ((i,j),(k,l))
// with O(N^2 n^2 / N) memory instead of O(N^4).  The r range is the multi-GPU shard axis here (partial sums are
// all-reduced by the host): sharding the MO index i would make every rank regenerate the
// whole tensor.  The integrals never exist in memory: gemm_f64_kernel<..., B_GEN> evaluates the
// hash for the 16 x 128 B tile it is about to stage (4 values per thread per k-step, hidden
// behind the 32 MFMAs per wave of that step).
#include "nbx_common.h"
#include "synth_device.h"

namespace {

size_t align256(size_t x) { return (x + 255) & ~(size_t)255; }

struct SynthPlan {
    int64_t sc, rb;
    size_t m_off, y_off, z_off, u_off, w_off, acc_off, u2_off, w2_off, acc2_off, total;
};

// n5, n6 > 0: a second tensor (C1 C2|C5 C6) accumulated beside the first one
SynthPlan plan(int64_t N, int64_t n1, int64_t n2, int64_t n3, int64_t n4, int64_t n5 = 0, int64_t n6 = 0) {
    SynthPlan p;
    int64_t sc = (int64_t)(4.0e9 / (8.0 * (double)N * (double)n1));  // ~4 GB of quarter-1 output
    if (sc < 1) sc = 1;
    if (sc > N) sc = N;
    p.sc = sc;
    p.rb = 8;
    size_t off = 0;
    p.m_off = off;  // (no generated slab any more)
    p.y_off = off; off += align256((size_t)(sc * n1 * N) * sizeof(double));
    p.z_off = off; off += align256((size_t)(N * n1 * n2) * sizeof(double));
    p.u_off = off; off += align256((size_t)(p.rb * n4 * n1 * n2) * sizeof(double));
    p.w_off = off; off += align256((size_t)(p.rb * n3 * n1 * n2) * sizeof(double));
    p.acc_off = off; off += align256((size_t)(n3 * n4 * n1 * n2) * sizeof(double));
    p.u2_off = off; off += align256((size_t)(p.rb * n6 * n1 * n2) * sizeof(double));
    p.w2_off = off; off += align256((size_t)(p.rb * n5 * n1 * n2) * sizeof(double));
    p.acc2_off = off; off += align256((size_t)(n5 * n6 * n1 * n2) * sizeof(double));
    p.total = off;
    return p;
}

}  // namespace

extern "C" size_t nbx_ao2mo_synth_worksize(int64_t nao, int64_t n1, int64_t n2, int64_t n3, int64_t n4) {
    if (nao <= 0 || n1 <= 0 || n2 <= 0 || n3 <= 0 || n4 <= 0) return 0;
    return plan(nao, n1, n2, n3, n4).total;
}

extern "C" int nbx_ao2mo_synth(nbx_ctx* ctx, int64_t nao, uint64_t seed, int64_t r0, int64_t r1, const double* d_c1,
                               int64_t n1, const double* d_c2, int64_t n2, const double* d_c3, int64_t n3,
                               const double* d_c4, int64_t n4, double* d_out, void* d_work, size_t work_bytes) {
    return nbx_ao2mo_synth_pair(ctx, nao, seed, r0, r1, d_c1, n1, d_c2, n2, d_c3, n3, d_c4, n4, d_out, nullptr, 0,
                                nullptr, 0, nullptr, d_work, work_bytes);
}

extern "C" size_t nbx_ao2mo_synth_pair_worksize(int64_t nao, int64_t n1, int64_t n2, int64_t n3, int64_t n4,
                                                int64_t n5, int64_t n6) {
    if (nao <= 0 || n1 <= 0 || n2 <= 0 || n3 <= 0 || n4 <= 0 || n5 < 0 || n6 < 0) return 0;
    return plan(nao, n1, n2, n3, n4, n5, n6).total;
}

extern "C" int nbx_ao2mo_synth_pair(nbx_ctx* ctx, int64_t nao, uint64_t seed, int64_t r0, int64_t r1,
                                    const double* d_c1, int64_t n1, const double* d_c2, int64_t n2, const double* d_c3,
                                    int64_t n3, const double* d_c4, int64_t n4, double* d_out, const double* d_c5,
                                    int64_t n5, const double* d_c6, int64_t n6, double* d_out2, void* d_work,
                                    size_t work_bytes) {
    const bool pair = d_out2 != nullptr;
    NBX_CHECK_ARG(ctx && d_c1 && d_c2 && d_c3 && d_c4 && d_out);
    NBX_CHECK_ARG(!pair || (d_c5 && d_c6 && n5 > 0 && n6 > 0));
    if (!pair) n5 = n6 = 0;
    NBX_CHECK_ARG(nao > 0 && n1 > 0 && n2 > 0 && n3 > 0 && n4 > 0 && r0 >= 0 && r1 >= r0 && r1 <= nao);
    const int64_t N = nao, n12 = n1 * n2;
    NBX_CHECK_ARG(n4 * n12 < (1ll << 31) && n6 * n12 < (1ll << 31) && N * N < (1ll << 31));
    const SynthPlan pl = plan(N, n1, n2, n3, n4, n5, n6);
    if (d_work == nullptr || work_bytes < pl.total) {
        nbx_set_error("nbx_ao2mo_synth: workspace %zu < %zu bytes", work_bytes, pl.total);
        return NBX_E_NOMEM;
    }
    char* base = static_cast<char*>(d_work);
    double* Y = reinterpret_cast<double*>(base + pl.y_off);
    double* Z = reinterpret_cast<double*>(base + pl.z_off);
    double* U = reinterpret_cast<double*>(base + pl.u_off);
    double* acc = reinterpret_cast<double*>(base + pl.acc_off);
    double* W = reinterpret_cast<double*>(base + pl.w_off);
    double* W2 = reinterpret_cast<double*>(base + pl.w2_off);
    double* U2 = reinterpret_cast<double*>(base + pl.u2_off);
    double* acc2 = reinterpret_cast<double*>(base + pl.acc2_off);
    nbx_prof_scope prof_all(ctx, NBX_PROF_AO2MO);
    int rc = nbx_memset(ctx, acc, 0, (size_t)(n3 * n4 * n12) * sizeof(double));
    if (rc != NBX_OK) return rc;
    if (pair) {
        rc = nbx_memset(ctx, acc2, 0, (size_t)(n5 * n6 * n12) * sizeof(double));
        if (rc != NBX_OK) return rc;
    }
    int64_t rb0 = r0;  // first r of the U block being filled
    for (int64_t r = r0; r < r1; ++r) {
        const int64_t nsr = r + 1;  // s <= r only
        for (int64_t s0 = 0; s0 < nsr; s0 += pl.sc) {
            const int64_t ns = (nsr - s0) < pl.sc ? (nsr - s0) : pl.sc;
            {
                nbx_prof_scope prof_q1(ctx, NBX_PROF_AO2MO_Q1);
                rc = nbx_gemm_q1_synth(ctx, n1, N, N, d_c1, n1, seed, 1.0 / (double)N, r, s0, Y, N, n1 * N, ns);
            }
            if (rc != NBX_OK) return rc;
            rc = nbx_gemm(ctx, 'N', 'N', n1, n2, N, 1.0, Y, N, n1 * N, d_c2, n2, 0, 0.0, Z + s0 * n12, n2, n12, ns);
            if (rc != NBX_OK) return rc;
        }
        // U[r - rb0] (n4 x n12) = C4[:r+1]^T . Z[:r+1]        (pairs (r, s <= r) as C3[r,k] C4[s,l])
        // W[r - rb0] (n3 x n12) = C3[:r]^T   . Z[:r]          (pairs (r, s <  r) as C3[s,k] C4[r,l])
        rc = nbx_gemm(ctx, 'T', 'N', n4, n12, nsr, 1.0, d_c4, n4, 0, Z, n12, 0, 0.0, U + (r - rb0) * n4 * n12, n12, 0, 1);
        if (rc != NBX_OK) return rc;
        rc = nbx_gemm(ctx, 'T', 'N', n3, n12, r, 1.0, d_c3, n3, 0, Z, n12, 0, 0.0, W + (r - rb0) * n3 * n12, n12, 0, 1);
        if (rc != NBX_OK) return rc;
        if (pair) {  // the second tensor reuses Z: only quarters 3 and 4 are repeated
            rc = nbx_gemm(ctx, 'T', 'N', n6, n12, nsr, 1.0, d_c6, n6, 0, Z, n12, 0, 0.0, U2 + (r - rb0) * n6 * n12, n12, 0,
                          1);
            if (rc != NBX_OK) return rc;
            rc = nbx_gemm(ctx, 'T', 'N', n5, n12, r, 1.0, d_c5, n5, 0, Z, n12, 0, 0.0, W2 + (r - rb0) * n5 * n12, n12, 0, 1);
            if (rc != NBX_OK) return rc;
        }
        if (r - rb0 + 1 == pl.rb || r + 1 == r1) {
            const int64_t nr = r - rb0 + 1;
            // acc (n3 x n4 n12) += C3[rb0:rb0+nr]^T (n3 x nr) . U (nr x n4 n12)
            rc = nbx_gemm(ctx, 'T', 'N', n3, n4 * n12, nr, 1.0, d_c3 + rb0 * n3, n3, 0, U, n4 * n12, 0, 1.0, acc,
                          n4 * n12, 0, 1);
            if (rc != NBX_OK) return rc;
            // acc[k] (n4 x n12) += C4[rb0:rb0+nr]^T (n4 x nr) . W[:, k] (nr x n12), batched over k
            rc = nbx_gemm(ctx, 'T', 'N', n4, n12, nr, 1.0, d_c4 + rb0 * n4, n4, 0, W, n3 * n12, n12, 1.0, acc, n12,
                          n4 * n12, n3);
            if (rc != NBX_OK) return rc;
            if (pair) {
                rc = nbx_gemm(ctx, 'T', 'N', n5, n6 * n12, nr, 1.0, d_c5 + rb0 * n5, n5, 0, U2, n6 * n12, 0, 1.0, acc2,
                              n6 * n12, 0, 1);
                if (rc != NBX_OK) return rc;
                rc = nbx_gemm(ctx, 'T', 'N', n6, n12, nr, 1.0, d_c6 + rb0 * n6, n6, 0, W2, n5 * n12, n12, 1.0, acc2, n12,
                              n6 * n12, n5);
                if (rc != NBX_OK) return rc;
            }
            rb0 = r + 1;
        }
    }
    // acc[(k,l)][(i,j)] -> out[(i,j)][(k,l)]
    rc = nbx_transpose(ctx, n3 * n4, n12, 1, acc, d_out);
    if (rc != NBX_OK || !pair) return rc;
    return nbx_transpose(ctx, n5 * n6, n12, 1, acc2, d_out2);
}
