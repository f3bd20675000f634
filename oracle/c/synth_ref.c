/*
 * CPU oracle (TEST INFRASTRUCTURE, see oracle/__init__.py): the synthetic (pq|rs) of SURVEY.md
 * section 8d in plain C, and two contractions that consume it WITHOUT storing it, so that the
 * sizes the benchmark quotes (N_AO = 148 dense, N_AO = 2000 streamed) can be checked on the host
 * in seconds.  Same counter hash as oracle/synth.py (which tests/test_oracle_golden.py holds this
 * file to, bit for bit):
 *
 *     u(k)  = splitmix64(((stream << 48) | k) XOR seed),  stream 0 for the ERI
 *     val   = (u >> 11) * 2^-53 * 2 - 1
 *     (pq|rs) = val(tri(tri(p,q), tri(r,s))) * (1 / N)
 *
 * The reference has no generator of its own: these integrals stand in for mol.intor('int2e')
 * reached through get_veff (nbed/scf/huzinaga_scf.py:156) and ao2mo.kernel
 * (nbed/ham_builder.py:128,138).
 */
#include <stddef.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

static inline uint64_t splitmix64(uint64_t x) {
    uint64_t z = x + 0x9E3779B97F4A7C15ULL;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
    return z ^ (z >> 31);
}

static inline uint64_t tri(uint64_t a, uint64_t b) {
    const uint64_t hi = a > b ? a : b, lo = a > b ? b : a;
    return hi * (hi + 1) / 2 + lo;
}

static inline double eri_val(uint64_t pq, uint64_t rs, uint64_t seed, double scale) {
    const uint64_t u = splitmix64(tri(pq, rs) ^ seed);
    return ((double)(u >> 11) * 0x1.0p-53 * 2.0 - 1.0) * scale;
}

/* out[(p-p0), q, r, s], rows p in [p0, p1) of the dense C-order tensor (oracle.synth.eri_block) */
void synth_eri_ref(int n, int p0, int p1, uint64_t seed, double* out) {
    const double scale = 1.0 / (double)n;
    const size_t n2 = (size_t)n * n;
#pragma omp parallel for collapse(2) schedule(static)
    for (int p = p0; p < p1; ++p) {
        for (int q = 0; q < n; ++q) {
            const uint64_t pq = tri((uint64_t)p, (uint64_t)q);
            double* t = out + ((size_t)(p - p0) * n + q) * n2;
            for (int r = 0; r < n; ++r)
                for (int s = 0; s < n; ++s) t[(size_t)r * n + s] = eri_val(pq, tri((uint64_t)r, (uint64_t)s), seed, scale);
        }
    }
}

/*
 * The ADDITIVE symmetric J/K form of nbx_jk_dense_sym / nbx_jk_packed / nbx_jk_synth_sym
 * (include/nbx.h): out ((1+ndm), N, N) holds the contribution of the pairs (p, q <= p), p in
 * [p0, p1), and of their mirror images:
 *     J[p,q] = J[q,p] = sum_rs (pq|rs) Dtot[r,s]
 *     K^x[p,r] += sum_s (pq|rs) D^x[q,s];   q < p:  K^x[q,r] += sum_s (pq|rs) D^x[p,s]
 * Integrals generated on the fly.  Parallel over r (each thread owns column r of K).
 */
void jk_synth_sym_ref(int n, int p0, int p1, uint64_t seed, const double* dm, int ndm, double* out) {
    const double scale = 1.0 / (double)n;
    const size_t n2 = (size_t)n * n;
    double* dtot = (double*)malloc(n2 * sizeof(double));
    for (size_t i = 0; i < n2; ++i) {
        double t = dm[i];
        for (int x = 1; x < ndm; ++x) t += dm[(size_t)x * n2 + i];
        dtot[i] = t;
    }
    memset(out, 0, (size_t)(1 + ndm) * n2 * sizeof(double));
    for (int p = p0; p < p1; ++p) {
        for (int q = 0; q <= p; ++q) {
            const uint64_t pq = tri((uint64_t)p, (uint64_t)q);
            double jacc = 0.0;
#pragma omp parallel for schedule(static) reduction(+ : jacc)
            for (int r = 0; r < n; ++r) {
                double kp[2] = {0.0, 0.0}, kq[2] = {0.0, 0.0};
                for (int s = 0; s < n; ++s) {
                    const double v = eri_val(pq, tri((uint64_t)r, (uint64_t)s), seed, scale);
                    jacc += v * dtot[(size_t)r * n + s];
                    for (int x = 0; x < ndm; ++x) {
                        kp[x] += v * dm[(size_t)x * n2 + (size_t)q * n + s];
                        kq[x] += v * dm[(size_t)x * n2 + (size_t)p * n + s];
                    }
                }
                for (int x = 0; x < ndm; ++x) {
                    out[(size_t)(1 + x) * n2 + (size_t)p * n + r] += kp[x];
                    if (q < p) out[(size_t)(1 + x) * n2 + (size_t)q * n + r] += kq[x];
                }
            }
            out[(size_t)p * n + q] = jacc;
            out[(size_t)q * n + p] = jacc;
        }
    }
    free(dtot);
}

/*
 * Quarters 1-2 of the four-index transform for ONE value of r and m chosen column pairs:
 *     y[k][s] = sum_pq a[k][p] b[k][q] (pq|rs),   s in [0, r]   (y is (m, r+1))
 * -- the half-transformed integrals nbx_ao2mo_synth builds for the pairs s <= r of its r-slab
 * (nbed/ham_builder.py:128: ao2mo.kernel; the host finishes quarters 3-4 of the sampled elements).
 * a, b are (m, N) row-major: row k = column i_k of C1, column j_k of C2.
 */
void half_transform_rs_ref(int n, uint64_t seed, int r, const double* a, const double* b, int m, double* y) {
    const double scale = 1.0 / (double)n;
#pragma omp parallel for schedule(dynamic, 4)
    for (int s = 0; s <= r; ++s) {
        const uint64_t rs = tri((uint64_t)r, (uint64_t)s);
        double acc[64];
        double* t = (double*)malloc((size_t)m * n * sizeof(double)); /* t[k][p] = sum_q b[k][q] (pq|rs) */
        for (int p = 0; p < n; ++p) {
            for (int k = 0; k < m; ++k) acc[k] = 0.0;
            for (int q = 0; q < n; ++q) {
                const double v = eri_val(tri((uint64_t)p, (uint64_t)q), rs, seed, scale);
                for (int k = 0; k < m; ++k) acc[k] += v * b[(size_t)k * n + q];
            }
            for (int k = 0; k < m; ++k) t[(size_t)k * n + p] = acc[k];
        }
        for (int k = 0; k < m; ++k) {
            double sum = 0.0;
            for (int p = 0; p < n; ++p) sum += a[(size_t)k * n + p] * t[(size_t)k * n + p];
            y[(size_t)k * (r + 1) + s] = sum;
        }
        free(t);
    }
}
