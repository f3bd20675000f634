/*
 * CPU oracle (TEST INFRASTRUCTURE, see oracle/__init__.py): plain C restatement of the
 * two O(N^4)/O(N^5) contractions of the path, used (a) to cross-check the numpy oracle and
 * (b) as the "port" CPU baseline that bench.py times on the host cores.
 *
 *   jk_ref    : J_pq = sum_rs (pq|rs) Dtot_rs ; K^x_pr = sum_qs (pq|rs) D^x_qs
 *               -- what PySCF's get_jk computes for the reference at
 *               nbed/scf/huzinaga_scf.py:156 (restated from the definition; PySCF's libcvhf
 *               is not part of /root/reference)
 *   ao2mo_ref : (ij|kl) = sum_pqrs C1_pi C2_qj C3_rk C4_sl (pq|rs), four quarter
 *               transforms -- what ao2mo.kernel + restore(1) compute at nbed/ham_builder.py:127-131
 *
 * One pass over the dense (N,N,N,N) tensor, OpenMP over the first index.
 */
#include <stddef.h>
#include <stdlib.h>
#include <string.h>

/* out: (1+ndm, np, n) -- same layout as nbx_jk_dense */
void jk_ref(const double* eri, const double* dm, int ndm, int n, int p0, int p1, double* out) {
    const size_t n2 = (size_t)n * n;
    const int np = p1 - p0;
    double* dtot = (double*)malloc(n2 * sizeof(double));
    for (size_t i = 0; i < n2; ++i) {
        double t = dm[i];
        for (int x = 1; x < ndm; ++x) t += dm[(size_t)x * n2 + i];
        dtot[i] = t;
    }
    memset(out, 0, (size_t)(1 + ndm) * np * n * sizeof(double));
#pragma omp parallel for schedule(dynamic, 1)
    for (int pl = 0; pl < np; ++pl) {
        const double* ep = eri + (size_t)pl * n * n2;
        double* jrow = out + (size_t)pl * n;
        for (int q = 0; q < n; ++q) {
            const double* t = ep + (size_t)q * n2;
            double jacc = 0.0;
#pragma omp simd reduction(+ : jacc)
            for (size_t i = 0; i < n2; ++i) jacc += t[i] * dtot[i];
            jrow[q] = jacc;
            for (int x = 0; x < ndm; ++x) {
                const double* dq = dm + (size_t)x * n2 + (size_t)q * n;
                double* krow = out + ((size_t)(1 + x) * np + pl) * n;
                for (int r = 0; r < n; ++r) {
                    const double* tr = t + (size_t)r * n;
                    double acc = 0.0;
#pragma omp simd reduction(+ : acc)
                    for (int s = 0; s < n; ++s) acc += tr[s] * dq[s];
                    krow[r] += acc;
                }
            }
        }
    }
    free(dtot);
}

/* C[m x n] = A^T[m x k] * B[k x n]  (A stored k x m), row-major, accumulate into zeroed C */
static void gemm_tn(int m, int n, int k, const double* a, int lda, const double* b, int ldb, double* c, int ldc) {
    for (int i = 0; i < m; ++i) memset(c + (size_t)i * ldc, 0, (size_t)n * sizeof(double));
    for (int p = 0; p < k; ++p) {
        const double* bp = b + (size_t)p * ldb;
        for (int i = 0; i < m; ++i) {
            const double aip = a[(size_t)p * lda + i];
            double* ci = c + (size_t)i * ldc;
#pragma omp simd
            for (int j = 0; j < n; ++j) ci[j] += aip * bp[j];
        }
    }
}

/* out (n1,n2,n3,n4); work: n1*N^3 + n1*n2*N^2 + n1*n2*n3*N doubles */
void ao2mo_ref(const double* eri, int n, const double* c1, int n1, const double* c2, int n2, const double* c3,
               int n3, const double* c4, int n4, double* out, double* work) {
    const size_t N = (size_t)n, N2 = N * N, N3 = N2 * N;
    double* x1 = work;
    double* x2 = x1 + (size_t)n1 * N3;
    double* x3 = x2 + (size_t)n1 * n2 * N2;
    /* Q1: x1[i,(qrs)] = sum_p c1[p,i] eri[p,(qrs)], parallel over column blocks */
#pragma omp parallel for schedule(static)
    for (long blk = 0; blk < (long)N2; ++blk) {
        gemm_tn(n1, n, n, c1, n1, eri + (size_t)blk * N, (int)N3, x1 + (size_t)blk * N, (int)N3);
    }
    /* Q2: x2[i][j,(rs)] = sum_q c2[q,j] x1[i][q,(rs)] */
#pragma omp parallel for schedule(static)
    for (int i = 0; i < n1; ++i) gemm_tn(n2, (int)N2, n, c2, n2, x1 + (size_t)i * N3, (int)N2, x2 + (size_t)i * n2 * N2, (int)N2);
    /* Q3: x3[ij][k,s] = sum_r c3[r,k] x2[ij][r,s] */
#pragma omp parallel for schedule(static)
    for (long ij = 0; ij < (long)n1 * n2; ++ij) gemm_tn(n3, n, n, c3, n3, x2 + (size_t)ij * N2, n, x3 + (size_t)ij * n3 * N, n);
    /* Q4: out[(ijk),l] = sum_s x3[(ijk),s] c4[s,l] */
#pragma omp parallel for schedule(static)
    for (long ijk = 0; ijk < (long)n1 * n2 * n3; ++ijk) {
        const double* xs = x3 + (size_t)ijk * N;
        double* o = out + (size_t)ijk * n4;
        for (int l = 0; l < n4; ++l) o[l] = 0.0;
        for (int s = 0; s < n; ++s) {
            const double v = xs[s];
            const double* cs = c4 + (size_t)s * n4;
#pragma omp simd
            for (int l = 0; l < n4; ++l) o[l] += v * cs[l];
        }
    }
}
