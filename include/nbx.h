/*
 * nbx.h -- C ABI of libnbx: MI355X (gfx950) kernels for the embedded-SCF hot
 * path of UCL-CCS/Nbed (Huzinaga / mu-shift Fock build, J/K contraction,
 * projector products, symmetric eigensolve, SPADE / concentric-localisation
 * SVDs, AO->active-MO four-index transform, spin-orbital scatter).
 *
 * The reference is pure Python and has no FFI of its own; what this library
 * replaces are the calls the reference makes into PySCF's C libraries and
 * into NumPy/SciPy LAPACK on that path.  Each entry point cites the reference
 * call site(s) it stands in for (paths relative to the reference repository).
 *
 * Conventions (binding):
 *   - extern "C", every function returns int: 0 = ok, < 0 = error
 *     (NBX_E_*); nbx_last_error() returns a thread-local message.
 *   - No exceptions cross the boundary.  Caller allocates every output.
 *   - Matrices are row-major (C order) double precision.  "d_" pointers are
 *     DEVICE pointers (hipMalloc / nbx_malloc / torch tensor data_ptr);
 *     "h_" pointers are host pointers.
 *   - Sizes are int64_t.  Work runs asynchronously on the context's HIP
 *     stream; nbx_sync() or a d2h copy waits for it.
 *   - A context is bound to one device and one stream and must not be used
 *     from two threads at once; the library keeps no other global state.
 *   - Multi-GPU: one process (context) per GPU.  Entry points that shard take
 *     an index range [i0, i1) and compute that slab only; the exchange step
 *     (all-gather over RCCL) is done by the host (torch.distributed).
 */
#ifndef NBX_H
#define NBX_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ABI version: bumped whenever the meaning or the size of a buffer behind an existing entry point changes (a caller
 * built against an older header would otherwise get silent out-of-bounds writes, not an error).  Callers compare
 * nbx_version() with the NBX_VERSION they were built against and refuse to run on a mismatch (nbed_amd/_nbx.py does).
 *   2 (round 4): nbx_huz_cycle's h_out is 7 doubles (was 6); nbx_huz_cycle_scalars_dev/_dts write a ready word at
 *     d_out[4 + tail_n]; nbx_diis_update*'s d_coef holds nbx_diis_coef_doubles(space) zeroed doubles; nbx_huz_state
 *     gained jk_kind, jk_p0, jk_p1, d_eri and (jk_p0, jk_p1) = (0, 0) now means an EMPTY slab; nbx_xc_density /
 *     nbx_xc_half are gone; nbx_mu_cycle* are new.
 *   3 (round 4): the packed tensor of N = 97 .. 148 is the 8-fold form (nbx_eri_packed_bytes / nbx_jk_packed_worksize /
 *     nbx_jk_dts_bytes return other sizes for it: a packed buffer kept from a version-2 library is not this one's);
 *     nbx_jk_packed_fold and nbx_sym_pow_ns* are new; nbx_eigh_tridiag_worksize grew by one status slot.          */
#define NBX_VERSION 3

#define NBX_OK 0
#define NBX_E_INVALID (-1)  /* bad argument (null pointer, negative size, shape mismatch) */
#define NBX_E_HIP (-2)      /* a HIP runtime call failed (message has the HIP error string) */
#define NBX_E_NOMEM (-3)    /* workspace too small / allocation failed */
#define NBX_E_NOCONV (-4)   /* iterative kernel hit its sweep limit (results still written) */
#define NBX_E_UNSUPPORTED (-5)

typedef struct nbx_ctx nbx_ctx;

/* ------------------------------------------------------------------ context */
int nbx_version(void);
/* 1 for a `make EXPERIMENTAL=1` build (the J/K experiments of DESIGN.md section 9 are linked in and selectable
 * through the environment), 0 for the shipped library. */
int nbx_experimental(void);
const char* nbx_last_error(void);
int nbx_device_count(int* count);
/* stream: the hipStream_t to launch on (e.g. torch's current stream; NULL = the device's
 * default stream).  private_stream != 0: ignore `stream` and create a private non-blocking
 * stream owned by the context. */
int nbx_ctx_create(int device, void* stream, int private_stream, nbx_ctx** ctx);
int nbx_ctx_destroy(nbx_ctx* ctx);
int nbx_ctx_set_stream(nbx_ctx* ctx, void* stream);
int nbx_sync(nbx_ctx* ctx);
int nbx_malloc(nbx_ctx* ctx, size_t bytes, void** d_ptr);
int nbx_free(nbx_ctx* ctx, void* d_ptr);
int nbx_memcpy_h2d(nbx_ctx* ctx, void* d_dst, const void* h_src, size_t bytes);
int nbx_memcpy_d2h(nbx_ctx* ctx, void* h_dst, const void* d_src, size_t bytes); /* synchronises */
/* count <= 8 device arrays of doubles, one after the other, into h_dst: PINNED (device-mapped) host memory the
 * kernel stores to directly -- one launch (the results of an SCF run).  h_dst: sum(n_doubles) + 1 doubles; the
 * last one receives 1.0 once everything before it is visible to the host.  wait != 0: synchronises the stream;
 * wait = 0: returns after the launch -- the caller cleared that word beforehand and polls it. */
int nbx_gather_to_host(nbx_ctx* ctx, int64_t count, const double* const* d_src, const int64_t* n_doubles,
                       double* h_dst, int wait);
int nbx_memcpy_d2d(nbx_ctx* ctx, void* d_dst, const void* d_src, size_t bytes);
int nbx_memset(nbx_ctx* ctx, void* d_ptr, int value, size_t bytes);

/* ------------------------------------------------------------------ density-fitted J/K (an extra)
 * (pq|rs) ~ sum_L B_L[p][q] B_L[r][s] with symmetric B_L -- what PySCF contracts
 * behind get_veff (huzinaga_scf.py:156, driver.py:344,847) when the mean-field object is built with
 * .density_fit().  The reference never asks for that; this entry point is NOT on the parity path of the exact
 * integrals.  It is the GEMM-shaped form of J/K, the only one there is at N_AO = 2000:
 *   J = sum_L B_L <B_L, D_a + D_b>,   K^x = sum_L (B_L C^x)(B_L C^x)^T,   D^x = C^x C^x^T.
 * d_b: (naux, N, N) the auxiliary functions this rank holds (results are additive over slabs of L);
 * d_c: (ndm, N, N) orbital coefficients whose first nocc[x] COLUMNS are the occupied orbitals of spin x
 *      (ndm = 1: closed shell, D_a = D_b); d_jk: (1 + ndm, N, N) = J, K_a[, K_b].
 * nbx_df_synth: d_b[l - l0][p][q] = scale * val(stream 9, l N(N+1)/2 + tri(p, q)), l in [l0, l1).               */
size_t nbx_jk_df_worksize(int64_t nao, int64_t ndm, int64_t nocc_max);
int nbx_jk_df(nbx_ctx* ctx, int64_t nao, int64_t naux, const double* d_b, int64_t ndm, const double* d_c,
              const int64_t* nocc, double* d_jk, void* d_work, size_t work_bytes);
int nbx_df_synth(nbx_ctx* ctx, int64_t nao, int64_t l0, int64_t l1, uint64_t seed, double scale, double* d_b);

/* ------------------------------------------------------------------ in-library kernel timing
 * While enabled, the listed launches are bracketed by HIP events recorded on the context's
 * stream; nbx_profile_read() waits for them and returns the accumulated milliseconds and
 * launch count of a slot.  bench.py uses this for the roofline numbers.                    */
#define NBX_PROF_JK_DENSE 0    /* jk_dense_kernel (the streaming kernel of nbx_jk_dense)     */
#define NBX_PROF_AO2MO_Q1 1    /* first quarter-transform GEMM of nbx_ao2mo                  */
#define NBX_PROF_AO2MO 2       /* all four quarter transforms of one nbx_ao2mo call          */
#define NBX_PROF_EIGH 3        /* eigh_jacobi_kernel                                         */
#define NBX_PROF_SVD 4         /* svd_jacobi_kernel                                          */
#define NBX_PROF_GEMM 5        /* every gemm_f64_kernel launch                               */
/* on = 0: off; on = 1: every slot; on = 2 | (mask << 2): only the slots whose bit is set in
 * mask (an event pair costs a few microseconds of stream time, which matters inside an SCF
 * cycle: bench.py brackets the J/K kernel only).                                             */
int nbx_profile_enable(nbx_ctx* ctx, int on);
/* Bracket only one launch in `every` of each enabled slot (the first after nbx_profile_reset, then every
 * `every`-th; default 1 = all of them).  An event pair holds the stream for ~11 us (two marker packets of ~6 us
 * each, measured on a rocprofv3 kernel trace with and without them: profiles/r03/README.md), 4 % of a settled
 * N_AO = 148 cycle: bench.py samples the J/K launch of every fourth cycle of its timed region.           */
int nbx_profile_sample(nbx_ctx* ctx, int every);
/* Test support: fills the LDS of every CU with `value` (tests pass a NaN) so that a kernel reading LDS it has not
 * written fails its parity test instead of depending on what ran on the CU before (tests/test_gpu_kernels.py).   */
int nbx_debug_fill_lds(nbx_ctx* ctx, double value);
int nbx_profile_read(nbx_ctx* ctx, int slot, double* ms_sum, int64_t* count);
int nbx_profile_reset(nbx_ctx* ctx);

/* ------------------------------------------------------------------ synthetic inputs
 * Counter-hash (pq|rs) of SURVEY.md section 8d, rows p in [p0,p1) of the dense
 * C-order (N,N,N,N) tensor: d_eri[(p-p0),q,r,s] = val(canon(p,q,r,s)) / N.
 * Stands in for mol.intor('int2e') (reached through PySCF get_veff / ao2mo.kernel,
 * nbed/scf/huzinaga_scf.py:156, nbed/ham_builder.py:128). */
int nbx_synth_eri(nbx_ctx* ctx, int64_t nao, int64_t p0, int64_t p1, uint64_t seed, double* d_eri);

/* ------------------------------------------------------------------ J/K contraction
 * Replaces PySCF UHF.get_jk / get_veff (libcvhf) at nbed/scf/huzinaga_scf.py:156,
 * nbed/scf/embedded_hcore_funcs.py:34, nbed/driver.py:344-345,847-849,
 * nbed/localizers/virtual/concentric.py:104,109.
 *   d_eri : slab rows p in [p0,p1) of the dense (N,N,N,N) chemist-order ERI
 *   d_dm  : (ndm,N,N) density matrices (ndm = 1 or 2), symmetric
 *   d_jk  : out, ((1+ndm), p1-p0, N):  [0] = J rows from sum_x dm[x],
 *           [1+x] = K rows from dm[x]:  J_pq = sum_rs (pq|rs) D_rs,
 *           K_pr = sum_qs (pq|rs) D_qs   (uses (pq|rs) = (pq|sr))
 *   d_work: nbx_jk_dense_worksize() bytes                                  */
size_t nbx_jk_dense_worksize(int64_t nao, int64_t np, int64_t ndm);
int nbx_jk_dense(nbx_ctx* ctx, int64_t nao, int64_t p0, int64_t p1, const double* d_eri,
                 const double* d_dm, int64_t ndm, double* d_jk, void* d_work, size_t work_bytes);

/* Symmetric form: (pq|rs) = (qp|rs), so only the tiles q <= p of the slab are read -- half the
 * bytes of nbx_jk_dense, which is HBM bound (PySCF's libcvhf, which the reference calls, works on
 * 8-fold packed integrals).  The interface is ADDITIVE over slabs:
 *   d_jk : out, ((1+ndm), N, N) full-size matrices holding the contribution of slab rows
 *          p in [p0,p1): all pairs (p, q <= p) and their mirror images; [0] = J, [1+x] = K of
 *          dm[x].  With p0 = 0, p1 = N they are the result; across GPUs the slabs' outputs are
 *          summed (all-reduce).  The work of a slab grows with p: equal-work slabs are not
 *          equal-height.
 * Sizes the symmetric kernel does not cover (odd N, N > 512) go through nbx_jk_dense inside.  */
size_t nbx_jk_dense_sym_worksize(int64_t nao, int64_t p0, int64_t p1, int64_t ndm);
int nbx_jk_dense_sym(nbx_ctx* ctx, int64_t nao, int64_t p0, int64_t p1, const double* d_eri,
                     const double* d_dm, int64_t ndm, double* d_jk, void* d_work, size_t work_bytes);

/* Packed form: (pq|rs) = (qp|rs) = (pq|sr), so only q <= p and s <= r are stored and read -- a
 * quarter of the dense tensor (PySCF holds `mf._eri` 8-fold packed and libcvhf, which the
 * reference reaches through get_veff at nbed/scf/huzinaga_scf.py:156, contracts it packed).
 * nbx_eri_pack converts slab rows [p0,p1) of the dense tensor ONCE (the integrals do not change
 * during an SCF); nbx_jk_packed then reads nbx_eri_packed_bytes() per call instead of
 * 8 N^4 (p1-p0)/N.  Storage: the tiles T(p,q) = p(p+1)/2 + q, q <= p, in sequence; a tile is the
 * lower triangle of its (r,s) matrix cut into NB = 2 (N <= 128) or 4 blocks of s = N/NB rows:
 * first the NB diagonal triangles (row-major, packed), then for r = 1..NB-1 the rectangles
 * (I, J = I ^ r), I > J, ordered by the row block I, row-major with row stride s|1 (zero pad); each
 * of the NB chunks padded to an even number of doubles.  (With NBX_JK_P8=1 in the environment whole
 * tensors of N = 100..156, N % 4 == 0 are stored 8-fold instead -- of every tile only the entries
 * (r,s) <= (p,q), csrc/jk_p8.hip; same entry points and results, experimental: DESIGN.md section 9.)
 * N = 100, 104, ..., 148 have a layout and a kernel of their own (csrc/jk_m4.hip, one instance per size, the contraction on
 * v_mfma_f64_4x4x4_4b_f64; N = 97 .. 147 that are not multiples of four run as the next multiple with the extra rows and
 * columns zero -- the padding is internal, as for jk_s4.hip's padded sizes): a tile is the
 * lower triangle in 4 x 4 blocks, block (T, C <= T) at T(T+1)/2 + C, element (i, k) at 4 (k ^ ((T ^ C) & 3)) + (i ^ k),
 * zeros above the diagonal of the diagonal blocks; NBX_JK_M4=0 in the environment keeps the layout above.  Either way the
 * packed buffer is opaque to the caller: nbx_eri_packed_bytes / nbx_eri_pack / nbx_jk_packed agree on it per process.
 * That kernel uses K = K^T (d_dm symmetric, as the interface asks): its per-slab outputs are symmetrised partials.
 * N = 149 .. 400 run the same walk with a tile in more chunks (csrc/jk_mx.hip: instances at every multiple of eight from
 * 152 to 256, then 272, 288 and every multiple of sixteen from 304 to 400; the same 4 x 4 blocks, stored chunk after
 * chunk -- whole block rows at the top of the triangle, from N = 304 the lower bands of four block rows in segments of
 * column groups; a size between two instances runs as the next one within eight, zero-padded as above;
 * NBX_JK_MX=0 hands N <= 256 back to the layout above and the sizes beyond it to nbx_jk_dense_sym).  These sizes make
 * their Dtot' table themselves (nbx_jk_dts_bytes = 0: nothing to hand over).  Dense + packed tensor must both be
 * resident while nbx_eri_pack runs, which ends at N = 400 on one 288 GB device.
 * N = 97 .. 148 (round 4; csrc/jk_m8.hip, an instance per multiple of four as for jk_m4.hip, unless NBX_JK_M8=0) are stored
 * and contracted 8-FOLD packed -- of tile (p, q) only the elements (rs) <= (pq), the element (rs) = (pq) halved: every
 * integral is in HBM once, 0.63 of the 4-fold form's bytes with the tiles cut at chunk boundaries -- with the same blocks and
 * the same walk; K = Kp + Kp^T, the J term of the mirrored
 * copy as an AXPY in the loading waves' registers (csrc/jk_m8.hip).  nbx_jk_packed_fold says which form a size has.
 *   nbx_jk_packed_fold      : 8 = the packed tensor of this size holds the 8-fold unique integrals, 4 = the 4-fold ones
 *            (q <= p, s <= r), 0 = no packed form
 *   nbx_jk_packed_supported : 1 = a kernel instance serves N (even N <= 256 with N % NB == 0; the sizes above);
 *            2 = N is served as the next such size (at most 8 more) with the extra rows and
 *            columns zero -- odd N, N = 102, 150, 300, ...: the same entry points, the padding is internal
 *            (tiles with p >= N are neither stored nor visited, D is padded on the way in, J/K cropped
 *            on the way out); 0 = not covered (use nbx_jk_dense_sym)
 *   d_jk   : out, ((1+ndm), N, N): ADDITIVE over slabs exactly as nbx_jk_dense_sym
 *   d_work : nbx_jk_packed_worksize() bytes                                                  */
int nbx_jk_packed_supported(int64_t nao);
int nbx_jk_packed_fold(int64_t nao);
size_t nbx_eri_packed_bytes(int64_t nao, int64_t p0, int64_t p1);
int nbx_eri_pack(nbx_ctx* ctx, int64_t nao, int64_t p0, int64_t p1, const double* d_eri, double* d_packed);
size_t nbx_jk_packed_worksize(int64_t nao, int64_t p0, int64_t p1, int64_t ndm);
int nbx_jk_packed(nbx_ctx* ctx, int64_t nao, int64_t p0, int64_t p1, const double* d_packed,
                  const double* d_dm, int64_t ndm, double* d_jk, void* d_work, size_t work_bytes);

/* nbx_jk_packed over the whole tensor (p0 = 0, p1 = N, two spin densities) with the Fock
 * assembly of nbx_fock_uhf done by its reduction kernel (one launch less per SCF cycle):
 *   d_hv : (2,N,N) hcore + V_emb;  d_fock[x] = d_hv[x] + J - K[x];  d_vhf[x] = J - K[x] (may be
 *   NULL);  d_jk as nbx_jk_packed.  nbed/scf/huzinaga_scf.py:156-160.
 *   d_dts : NULL, or the Dtot' table of d_dm that nbx_huz_cycle_scalars_dts left behind when it
 *   judged this density (saves the build its preparation launch): nbx_jk_dts_bytes() bytes, zeroed
 *   ONCE with nbx_jk_dts_init (the entries it never writes are zero weights).               */
size_t nbx_jk_dts_bytes(int64_t nao);
int nbx_jk_dts_init(nbx_ctx* ctx, int64_t nao, double* d_dts);
int nbx_jk_packed_fock(nbx_ctx* ctx, int64_t nao, const double* d_packed, const double* d_dm,
                       const double* d_hv, double* d_jk, double* d_fock, double* d_vhf, void* d_work,
                       size_t work_bytes, const double* d_dts);

/* Same contraction with the synthetic (pq|rs) of nbx_synth_eri GENERATED in registers instead of
 * read from HBM (the N_AO = 2000 configuration: a dense tensor would be 128 TB).  Workspace as
 * nbx_jk_dense_worksize().  ALU-bound (one 64-bit counter hash per integral).               */
int nbx_jk_synth(nbx_ctx* ctx, int64_t nao, int64_t p0, int64_t p1, uint64_t seed, const double* d_dm,
                 int64_t ndm, double* d_jk, void* d_work, size_t work_bytes);

/* nbx_jk_synth in the additive, symmetric form of nbx_jk_dense_sym: only the tiles q <= p are
 * generated (half the hash evaluations; the kernel is VALU bound).  Even N <= 2048.  d_jk: out,
 * ((1+ndm), N, N) full-size contributions of slab rows [p0,p1), summed across slabs by the host. */
size_t nbx_jk_synth_sym_worksize(int64_t nao, int64_t p0, int64_t p1, int64_t ndm);
int nbx_jk_synth_sym(nbx_ctx* ctx, int64_t nao, int64_t p0, int64_t p1, uint64_t seed,
                     const double* d_dm, int64_t ndm, double* d_jk, void* d_work, size_t work_bytes);

/* ------------------------------------------------------------------ dense products (MFMA fp64)
 * C[b] = alpha * op(A[b]) * op(B[b]) + beta * C[b], row-major, op = 'N' or 'T'.
 * Replaces the OpenBLAS dgemm behind numpy matmul/einsum at
 * nbed/scf/huzinaga_scf.py:78-86,132-134,144-146,166-169; nbed/driver.py:439,446;
 * nbed/localizers/system.py:34-36; nbed/ham_builder.py:76-79;
 * nbed/localizers/occupied/spade.py:98-99,132-134;
 * nbed/localizers/virtual/concentric.py:146-153,175-176,205-207,228,235.      */
int nbx_gemm(nbx_ctx* ctx, char trans_a, char trans_b, int64_t m, int64_t n, int64_t k,
             double alpha, const double* d_a, int64_t lda, int64_t stride_a,
             const double* d_b, int64_t ldb, int64_t stride_b, double beta,
             double* d_c, int64_t ldc, int64_t stride_c, int64_t batch);

/* ------------------------------------------------------------------ Fock assembly & reductions */
/* F[x] = hcore[x or 0] + vemb[x] + J - K[x]  (x < 2); d_jk as written by nbx_jk_dense with
 * p0=0,p1=N,ndm=2.  hcore_ndim = 2 -> (N,N) shared, 3 -> (2,N,N).  d_vemb may be NULL.
 * nbed/scf/huzinaga_scf.py:157 (get_veff: vhf = J_a + J_b - K_x).                          */
int nbx_fock_uhf(nbx_ctx* ctx, int64_t nao, const double* d_hcore, int hcore_ndim,
                 const double* d_vemb, const double* d_jk, double* d_fock, double* d_vhf);
/* Huzinaga operator, occupied part (nbed/scf/huzinaga_scf.py:78-80):
 * given FDS[b] = F[b] @ (D_env[b] @ S):  Hz[b] = -kappa (FDS[b] + FDS[b]^T); if d_fock_io
 * is not NULL also F[b] += Hz[b] (:160).                                                    */
int nbx_huzinaga_sym(nbx_ctx* ctx, int64_t nao, int64_t batch, const double* d_fds, double kappa,
                     double* d_hz, double* d_fock_io);
/* out[b] = sum_ij A[b,i,j] * B[b,j,i]   (einsum "ij,ji->"; huzinaga_scf.py:185,
 * embedded_hcore_funcs.py:38-42, driver.py:957-962).  h_out: host, synchronises.          */
/* The same operator from F and DS = D_env S directly, product and symmetrisation in one launch
 * (huzinaga_scf.py:78-81 as one kernel): d_hz[b] = -kappa (F[b] DS[b] + (F[b] DS[b])^T) and, if
 * d_fock_out is not NULL, d_fock_out[b] = F[b] + d_hz[b] (out of place: must not alias d_f).  */
int nbx_huzinaga_fused(nbx_ctx* ctx, int64_t nao, int64_t batch, const double* d_f, const double* d_ds, double kappa,
                       double* d_hz, double* d_fock_out);
int nbx_trace_prod(nbx_ctx* ctx, int64_t nao, int64_t batch, const double* d_a, const double* d_b,
                   double* h_out);
/* Per-cycle scalars of the Huzinaga HF branch (huzinaga_scf.py:181-194):
 *   h_out[x]   = tr[(hcore + vemb[x] + 0.5 vhf[x] + hz[x]) D[x]]      x = 0,1
 *   h_out[2+x] = || D[x] - Dold[x] ||_F                                                     */
int nbx_huz_cycle_scalars(nbx_ctx* ctx, int64_t nao, const double* d_hcore, int hcore_ndim,
                          const double* d_vemb, const double* d_vhf, const double* d_hz,
                          const double* d_dm, const double* d_dm_old, double* h_out);
/* Same scalars (square roots applied) with no synchronisation, so that the host can queue the
 * next SCF cycle before it reads them.  One launch: the workgroup that arrives last does the
 * second stage.  d_out: 4 + tail_n + 1 doubles of device memory or pinned (device-mapped) host
 * memory -- stored to host memory directly the values need no copy.  The LAST word,
 * d_out[4 + tail_n], receives 1.0 after everything before it is visible system-wide: a host that
 * cleared it before the call can poll it instead of recording an event on the stream (an event
 * between two cycles holds the next cycle's first kernel back by ~10 us on MI355X).
 * d_tail (optional): tail_n <= 64 device ints appended as doubles, e.g. the eigensolver status
 * words, so that they reach the host together.                                                */
int nbx_huz_cycle_scalars_dev(nbx_ctx* ctx, int64_t nao, const double* d_hcore, int hcore_ndim,
                              const double* d_vemb, const double* d_vhf, const double* d_hz,
                              const double* d_dm, const double* d_dm_old, double* d_out,
                              const int* d_tail, int64_t tail_n);
/* nbx_huz_cycle_scalars_dev that also prepares the NEXT J/K build: with d_dts (see
 * nbx_jk_packed_fock) the kernel, which reads D anyway, leaves Dtot' of d_dm in the table.     */
int nbx_huz_cycle_scalars_dts(nbx_ctx* ctx, int64_t nao, const double* d_hcore, int hcore_ndim,
                              const double* d_vemb, const double* d_vhf, const double* d_hz,
                              const double* d_dm, const double* d_dm_old, double* d_out, const int* d_tail,
                              int64_t tail_n, double* d_dts);
/* One pyscf.lib.diis.DIIS.update step (behind huzinaga_scf.py:130,164) with nothing leaving
 * the device.  State: d_xs, d_es = (space, n) trial / error vectors, d_h = the
 * (space+1)x(space+1) Pulay matrix (row 0 / column 0 = 1, H[0][0] = 0; the caller initialises
 * it once), d_xprev = the vector returned by the previous update.  The call stores x in slot
 * `slot`, e = x - xprev beside it, fills row/column slot+1 of H with <e_slot|e_k> (k < nd, nd
 * counts the vectors held including this one), solves H[:nd+1,:nd+1] c = (1,0,...) with
 * PySCF's rule (modes with |eigenvalue| < 1e-14 are dropped when there are any, otherwise an
 * LU solve), and overwrites d_xprev with sum_k c[k+1] xs[k]: the extrapolated vector.
 * d_coef: nbx_diis_coef_doubles(space) doubles, ZERO-INITIALISED by the caller before the first
 * update of a ring and left alone afterwards: c[1:] in the first `space` of them, behind them
 * the solver's own state (the eigenvector basis its Jacobi sweeps ended in, from which the next
 * update's solve starts: one row / column of H changes per update).  1 <= nd <= space <= 16.   */
size_t nbx_diis_coef_doubles(int64_t space);
int nbx_diis_update(nbx_ctx* ctx, int64_t n, int64_t space, int64_t slot, int64_t nd,
                    const double* d_x, double* d_xprev, double* d_xs, double* d_es, double* d_h,
                    double* d_coef);
/* The same step with the error vector supplied by the caller (d_err, n doubles) instead of
 * x - xprev: pyscf.scf.diis.CDIIS (error S D F - F D S, space 8) behind scf.hf.kernel, which the
 * mu-shift path runs (nbed/driver.py:533).  d_xprev only receives the extrapolated vector.     */
int nbx_diis_update_err(nbx_ctx* ctx, int64_t n, int64_t space, int64_t slot, int64_t nd,
                        const double* d_x, const double* d_err, double* d_xprev, double* d_xs,
                        double* d_es, double* d_h, double* d_coef);
/* d_out[x] = sum_{a >= nocc_x, i < nocc_x} fmo[x][a][i]^2 for the (2,N,N) MO-basis Fock matrix
 * C^T F C: the squared norm of the orbital gradient PySCF's scf.hf.kernel tests for convergence
 * (get_grad + norm/sqrt(size), behind nbed/driver.py:533).  Device output, no synchronisation.  */
int nbx_vo_sumsq(nbx_ctx* ctx, int64_t nao, const double* d_fmo, int64_t nocc_a, int64_t nocc_b,
                 double* d_out);
/* y = a*x + b*y over n doubles. */
int nbx_axpby(nbx_ctx* ctx, int64_t n, double a, const double* d_x, double b, double* d_y);
/* out = sum_k coef[k] * vecs[k] (k < nvec; vecs[k] = d_vecs + k*stride), DIIS extrapolation
 * (pyscf.lib.diis.DIIS.extrapolate behind huzinaga_scf.py:164).                             */
int nbx_lincomb(nbx_ctx* ctx, int64_t n, int64_t nvec, const double* h_coef, const double* d_vecs,
                int64_t stride, double* d_out);
/* h_out[k] = <x, vecs[k]>, k < nvec (DIIS B-matrix row). Synchronises. */
int nbx_dots(nbx_ctx* ctx, int64_t n, int64_t nvec, const double* d_x, const double* d_vecs,
             int64_t stride, double* h_out);
/* B[b] = A[b]^T for (rows x cols) matrices. */
int nbx_transpose(nbx_ctx* ctx, int64_t rows, int64_t cols, int64_t batch, const double* d_a,
                  double* d_b);
/* A[:, j] *= s[j]  (columns scaled; row-major (rows x cols)). */
int nbx_scale_cols(nbx_ctx* ctx, int64_t rows, int64_t cols, int64_t batch, const double* d_s,
                   double* d_a);

/* ------------------------------------------------------------------ symmetric eigensolver
 * Replaces np.linalg.eigh (LAPACK dsyevd) at nbed/scf/huzinaga_scf.py:145,168 and the
 * eigensolve inside fractional_matrix_power at :128 / spade.py:99.
 * d_a: (batch,N,N) symmetric, preserved.  d_w: (batch,N) ascending.  d_v: (batch,N,N),
 * columns = eigenvectors.  Cyclic two-sided Jacobi (parallel ordering) for N < 64; Householder reduction,
 * Sturm multisection, inverse iteration and back-transformation above (csrc/eigh_tridiag.hip: the matrix in one
 * workgroup's registers up to N = 198; csrc/eigh_grid.hip beyond: the matrix in the LDS of up to 256 workgroups,
 * one grid-wide hand-over per Householder step -- a COOPERATIVE launch, so the call needs the device's CUs to itself
 * for its duration -- and the back-transformation in compact-WY blocks on the GEMM; N <= 2048), Jacobi as the
 * polisher when a cluster defeats inverse iteration.
 * d_work: nbx_eigh_worksize() bytes.  Returns NBX_E_NOCONV if max sweeps hit.             */
size_t nbx_eigh_worksize(int64_t n, int64_t batch);
int nbx_eigh(nbx_ctx* ctx, int64_t n, int64_t batch, const double* d_a, double* d_w, double* d_v,
             void* d_work, size_t work_bytes);
/* Warm start: d_v0 (batch,N,N) holds approximate eigenvectors (orthonormal columns, e.g. the
 * previous SCF cycle's).  For N <= 196 the pair is first refined with GEMMs only (Ogita-Aishima
 * iteration, 3 steps, quadratically convergent); a matrix the refinement does not bring to
 * max|E| < 3e-8 -- or that has coupled near-degenerate eigenvalues -- falls through, on the
 * device and with no host round trip, to Jacobi sweeps on V0^T A V0 (nearly diagonal: 1-4
 * sweeps instead of 8-10).  d_v0 == NULL is nbx_eigh.  Same outputs/workspace.               */
int nbx_eigh_warm(nbx_ctx* ctx, int64_t n, int64_t batch, const double* d_a, const double* d_v0,
                  double* d_w, double* d_v, void* d_work, size_t work_bytes);
/* Same with the number of refinement iterations queued before the Jacobi fallback chosen by the
 * caller (0 = none, 3 = the default of nbx_eigh_warm, 6 = most).  Results do not depend on it -- a matrix the queued
 * iterations do not finish is solved by the sweeps -- only the number of launches does: an SCF
 * driver that saw the last cycles accepted after one iteration queues one.                   */
int nbx_eigh_warm_ex(nbx_ctx* ctx, int64_t n, int64_t batch, const double* d_a, const double* d_v0,
                     double* d_w, double* d_v, void* d_work, size_t work_bytes, int refine_iters);
/* Byte offset inside the nbx_eigh workspace of the int[batch] status words nbx_eigh_status reads,
 * for callers that fetch them with their own stream-ordered copy instead of synchronising.   */
size_t nbx_eigh_status_offset(int64_t n, int64_t batch);
/* Reads back the sweep counts of the last nbx_eigh on this workspace (synchronises):
 * h_sweeps[b] > 0 = sweeps used (1000 + k: accepted by the warm-start refinement after k
 * iterations, no sweeps); returns NBX_E_NOCONV if any matrix hit the sweep limit.          */
int nbx_eigh_status(nbx_ctx* ctx, int64_t n, int64_t batch, const void* d_work, int* h_sweeps);
/* Eigenpairs to inverse-iteration accuracy with nothing read back: Householder reduction, multisection,
 * inverse iteration, back-transformation and one Newton-Schulz orthonormalisation step, all queued (the cold
 * route of nbx_eigh without its two host decisions).  A START for nbx_eigh_warm_ex / nbx_huz_cycle mode 0 on
 * the same or a nearby matrix -- e.g. solved on a second stream beside SCF cycles that do not need orbitals
 * (nbx_huz_cycle mode 2) -- not a result: d_status[b] = 1 the vectors are orthonormal to ~1e-12 (their
 * residual is whatever inverse iteration left, typically 1e-10 ||A||); -1 clustered levels left nearly
 * dependent vectors (or NaN): do not use them.  n <= 2048.                                          */
size_t nbx_eigh_approx_worksize(int64_t n, int64_t batch);
int nbx_eigh_approx(nbx_ctx* ctx, int64_t n, int64_t batch, const double* d_a, double* d_w, double* d_v,
                    void* d_work, size_t work_bytes, int* d_status);
/* out = S^p for symmetric positive definite S (N,N): U diag(w^p) U^T.
 * scipy.linalg.fractional_matrix_power(S, -0.5 / +0.5) at nbed/scf/huzinaga_scf.py:128,
 * nbed/localizers/occupied/spade.py:99; np.linalg.inv(S_AA) at
 * nbed/localizers/virtual/concentric.py:147 (p = -1).                                      */
size_t nbx_sym_pow_worksize(int64_t n);
/* The same power, p in {-1/2, +1/2, -1}, of a symmetric POSITIVE DEFINITE S by the coupled Newton-Schulz iteration
 * (three GEMMs per step, ~0.2 ms at N = 148 where the eigen route takes ~1-3 ms): c = a bound of the spectrum (the
 * caller's ||S||_inf), convergence (||I - Z Y||_F) read back every `check_every` steps.  *h_iters = steps taken, or -1
 * when it did not converge within max_iter (not positive definite, condition number beyond ~1e6): d_out is then
 * untouched and the caller takes nbx_sym_pow.  Agrees with fractional_matrix_power to ~cond(S) 1e-16.  Synchronises. */
size_t nbx_sym_pow_ns_worksize(int64_t n);
int nbx_sym_pow_ns(nbx_ctx* ctx, int64_t n, const double* d_s, double p, double c, double* d_out, void* d_work,
                   size_t work_bytes, int max_iter, int check_every, int* h_iters);
int nbx_sym_pow(nbx_ctx* ctx, int64_t n, const double* d_s, double p, double* d_out,
                void* d_work, size_t work_bytes);

/* ------------------------------------------------------------------ SVD (right vectors)
 * Replaces scipy.linalg.svd / np.linalg.svd (LAPACK dgesdd) at
 * nbed/localizers/occupied/spade.py:101 and nbed/localizers/virtual/concentric.py:151,205: A (m x n) -> s (min(m,n)) descending, Vt (n x n) with rows the
 * right singular vectors (full_matrices=True semantics).  One-sided Jacobi.              */
size_t nbx_svd_worksize(int64_t m, int64_t n);
int nbx_svd_right(nbx_ctx* ctx, int64_t m, int64_t n, const double* d_a, double* d_s, double* d_vt,
                  void* d_work, size_t work_bytes);
/* Generalised symmetric eigenproblem F C = S C eps by iterative refinement of the PREVIOUS SCF
 * cycle's solution (Ogita-Aishima on the pencil: G = C^T S C, S~ = C^T F C, C <- C (I + E)), all
 * GEMMs -- what scipy.linalg.eigh(F, S) / the Loewdin route of nbed/scf/huzinaga_scf.py:166-169
 * computes, once an SCF is under way.  batch matrices: d_f, d_s (the overlap, repeated per batch
 * entry), d_c0 (S-orthonormal start vectors), all (batch, n, n) row-major.
 *   d_status[b] = 1000 + iterations used: accepted, d_w[b] (ascending) and d_c[b] written;
 *               <= 0: not converged within max_iter (1..6), near-degenerate cluster coupled, or
 *                     (max_iter = 1, where no sorting pass is queued) the eigenvalue order has
 *                     changed -- outputs not valid.  There is NO fallback solver behind this entry: the
 *                     caller checks the status (the SCF loops do, one cycle late, and redo the run
 *                     on nbx_eigh_warm_ex if it ever fails).
 * Nothing synchronises.  d_work: nbx_geig_refine_worksize() bytes.                              */
size_t nbx_geig_refine_worksize(int64_t n, int64_t batch);
int nbx_geig_refine(nbx_ctx* ctx, int64_t n, int64_t batch, const double* d_f, const double* d_s,
                    const double* d_c0, double* d_w, double* d_c, int* d_status, void* d_work,
                    size_t work_bytes, int max_iter);

/* nbx_ao2mo_pair for C1 = C2 (one matrix d_c12, n columns) over the whole outer range:
 * (ij|kl) = (ji|kl), so quarters 3 and 4 run on the pairs j <= i only and each finished (k,l)
 * block is stored at (i,j) and (j,i).  Every ao2mo.kernel call of nbed/ham_builder.py:127-133
 * has C1 = C2.  Same outputs as nbx_ao2mo_pair up to rounding ((i,j) and (j,i) are equal by
 * construction here); d_out2 == NULL: one tensor.  Outer-index slabs, or n > 361 (more than 65535
 * pairs: NBX_E_UNSUPPORTED): use nbx_ao2mo_pair.                                               */
size_t nbx_ao2mo_pair_sym_worksize(int64_t nao, int64_t n, int64_t n4, int64_t n6);
/* ... and with (pq|rs) = (pq|sr) too: nbx_eri_pack_rs stores (r, s <= r) packed once per molecule
 * (out[(p,q)][r(r+1)/2 + s], nbx_eri_rs_bytes() bytes -- PySCF's "s2kl" form of ao2mo's input);
 * quarters 1 and 2 of nbx_ao2mo_pair_sym_rs then run on N(N+1)/2 columns instead of N^2 and
 * quarter 3 reads the packed X2 in place.  Same results as nbx_ao2mo_pair_sym up to rounding.  */
size_t nbx_eri_rs_bytes(int64_t nao);
int nbx_eri_pack_rs(nbx_ctx* ctx, int64_t nao, const double* d_eri, double* d_out);
size_t nbx_ao2mo_pair_sym_rs_worksize(int64_t nao, int64_t n, int64_t n4, int64_t n6);
int nbx_ao2mo_pair_sym_rs(nbx_ctx* ctx, int64_t nao, const double* d_eri_rs, const double* d_c12, int64_t n,
                          const double* d_c3, int64_t n3, const double* d_c4, int64_t n4, double* d_out,
                          const double* d_c5, int64_t n5, const double* d_c6, int64_t n6, double* d_out2,
                          void* d_work, size_t work_bytes);
int nbx_ao2mo_pair_sym(nbx_ctx* ctx, int64_t nao, const double* d_eri, const double* d_c12, int64_t n,
                       const double* d_c3, int64_t n3, const double* d_c4, int64_t n4, double* d_out,
                       const double* d_c5, int64_t n5, const double* d_c6, int64_t n6, double* d_out2,
                       void* d_work, size_t work_bytes);

/* Sweep count of the last nbx_svd_right on this workspace (synchronises); NBX_E_NOCONV if the
 * sweep limit was hit. */
int nbx_svd_status(nbx_ctx* ctx, int64_t m, int64_t n, const void* d_work, int* h_sweeps);

/* ------------------------------------------------------------------ four-index transform
 * Replaces pyscf.ao2mo.kernel + ao2mo.restore(1, ...) at nbed/ham_builder.py:127-131:
 *   out[i-i0,j,k,l] = sum_pqrs C1[p,i] C2[q,j] C3[r,k] C4[s,l] (pq|rs),  i in [i0,i1)
 * d_eri dense (N,N,N,N); Cx are (N,nx) row-major.  Four quarter transforms (MFMA GEMMs).
 * The outer MO index i is the multi-GPU shard axis.                                       */
size_t nbx_ao2mo_worksize(int64_t nao, int64_t ni, int64_t n2, int64_t n3, int64_t n4);
int nbx_ao2mo(nbx_ctx* ctx, int64_t nao, const double* d_eri, const double* d_c1, int64_t n1,
              int64_t i0, int64_t i1, const double* d_c2, int64_t n2, const double* d_c3,
              int64_t n3, const double* d_c4, int64_t n4, double* d_out, void* d_work,
              size_t work_bytes);
/* Two tensors that share their first two MO indices in one pass,
 *   out [i-i0,j,k,l] = (C1 C2 | C3 C4),   out2[i-i0,j,k,l] = (C1 C2 | C5 C6):
 * quarters 1 and 2 are computed once.  The unrestricted Hamiltonian needs (aa|aa), (bb|bb) and
 * (aa|bb) (nbed/ham_builder.py:127-133, three independent ao2mo.kernel calls in the reference):
 * (aa|aa) and (aa|bb) are such a pair, which removes a third of the quarter-1 work.  Results are
 * bitwise those of two nbx_ao2mo calls.  d_out2 == NULL: plain nbx_ao2mo.                    */
size_t nbx_ao2mo_pair_worksize(int64_t nao, int64_t ni, int64_t n2, int64_t n4, int64_t n6);
int nbx_ao2mo_pair(nbx_ctx* ctx, int64_t nao, const double* d_eri, const double* d_c1, int64_t n1,
                   int64_t i0, int64_t i1, const double* d_c2, int64_t n2, const double* d_c3,
                   int64_t n3, const double* d_c4, int64_t n4, double* d_out, const double* d_c5,
                   int64_t n5, const double* d_c6, int64_t n6, double* d_out2, void* d_work,
                   size_t work_bytes);
/* Streamed transform of the synthetic (pq|rs) of nbx_synth_eri, never stored (N_AO = 2000).
 * (pq|rs) = (pq|sr) is used: only the pairs s <= r are generated and half-transformed (the
 * symmetry PySCF's ao2mo, which the reference calls, uses as well).  The call returns the part
 * of the sum that belongs to r in [r0,r1), i.e. all pairs (r, s <= r) and their mirror images:
 *   out[i,j,k,l] = sum_{r in [r0,r1)} sum_{s<=r} sum_pq C1[p,i] C2[q,j] (pq|rs)
 *                                   (C3[r,k] C4[s,l] + [s<r] C3[s,k] C4[r,l])
 * [r0,r1) is the multi-GPU shard axis: the partial tensors of the ranks are SUMMED (all-reduce)
 * by the host; the work of a range grows with r, so equal-work ranges are not equal-length.   */
size_t nbx_ao2mo_synth_worksize(int64_t nao, int64_t n1, int64_t n2, int64_t n3, int64_t n4);
/* ... and the pair form of it (see nbx_ao2mo_pair): out2 = (C1 C2|C5 C6) beside out, the generated
 * integrals and quarters 1-2 shared.  d_out2 == NULL: nbx_ao2mo_synth.                        */
size_t nbx_ao2mo_synth_pair_worksize(int64_t nao, int64_t n1, int64_t n2, int64_t n3, int64_t n4,
                                     int64_t n5, int64_t n6);
int nbx_ao2mo_synth_pair(nbx_ctx* ctx, int64_t nao, uint64_t seed, int64_t r0, int64_t r1,
                         const double* d_c1, int64_t n1, const double* d_c2, int64_t n2,
                         const double* d_c3, int64_t n3, const double* d_c4, int64_t n4, double* d_out,
                         const double* d_c5, int64_t n5, const double* d_c6, int64_t n6, double* d_out2,
                         void* d_work, size_t work_bytes);
int nbx_ao2mo_synth(nbx_ctx* ctx, int64_t nao, uint64_t seed, int64_t r0, int64_t r1, const double* d_c1,
                    int64_t n1, const double* d_c2, int64_t n2, const double* d_c3, int64_t n3,
                    const double* d_c4, int64_t n4, double* d_out, void* d_work, size_t work_bytes);
/* out[a,c,d,b] = in[a,b,c,d]: chemist (ij|kl) -> the reference's physicist-ordered block
 * eri.transpose(0,2,3,1) (ham_builder.py:133).                                            */
int nbx_chem_to_phys(nbx_ctx* ctx, int64_t n1, int64_t n2, int64_t n3, int64_t n4,
                     const double* d_in, double* d_out);
/* Spin-orbital scatter + truncation (nbed/ham_builder.py:158-216) and the 0.5 factor of
 * build() (:254):  h1 (2n,2n), h2 (2n,2n,2n,2n) from one_body (2,n,n), two_body (4,n,n,n,n);
 * entries with |x| < tol are zeroed; h2 is multiplied by h2_scale afterwards.            */
int nbx_spinorb_scatter(nbx_ctx* ctx, int64_t n, const double* d_one_body, const double* d_two_body,
                        double tol, double h2_scale, double* d_h1, double* d_h2);
/* Elements [idx0, idx0 + count) of the flattened (2n)^4 tensor only, into d_h2_part[0..count): lets
 * a host that wants the result in its own memory stream it out piece by piece instead of holding
 * the whole tensor (60 GB at n = 147) on the device first; nbx_spinorb_scatter_h1 is the one-body
 * part alone (nbed/ham_builder.py:180-214, as nbx_spinorb_scatter).                             */
int nbx_spinorb_scatter_h1(nbx_ctx* ctx, int64_t n, const double* d_one_body, double tol, double* d_h1);
int nbx_spinorb_scatter_range(nbx_ctx* ctx, int64_t n, const double* d_two_body, double tol,
                              double h2_scale, int64_t idx0, int64_t count, double* d_h2_part);

/* ------------------------------------------------------------------ fused SCF cycle
 * One Huzinaga-projected UHF cycle of nbed/scf/huzinaga_scf.py:154-201 queued by ONE call (the chain is
 * 12-40 launches; issued one by one from the host language the first cycles of an SCF are host bound).
 * The caller fills the state once per SCF -- every pointer a device pointer it owns -- and rotates the
 * per-cycle result buffers itself; nothing here allocates or synchronises.  Same kernels, order and
 * operands as the step-by-step entry points: bit-identical results.
 *   mode 1 ("tracked"): nbx_geig_refine from d_c_in = the previous cycle's S-orthonormal C; status in
 *          d_status_out.  mode 0 ("guarded"): X F X, nbx_eigh_warm_ex warm-started from d_c_in = the previous
 *          cycle's orthonormal-basis vectors (NULL: cold), C = X V; d_v_out receives V.
 *          mode 2 ("purified"): no eigenvectors at all -- D = X P X with P the projector on the occupied
 *          levels of X F X from nbx_purify (refine_iters caps its steps, <= 0: the limit); d_c_out and
 *          d_w_out are NOT written, d_v_out (if given) receives X F X -- the matrix an eigensolver would
 *          have been given, so that the cycle's orbitals can be had later or beside the next cycles on
 *          another stream; the status words are nbx_purify's.  For the first cycles of a run, whose
 *          Fock matrix moves too much for a warm-started eigensolver.
 *   diis_mode 0: no DIIS; 1: pyscf.lib.diis' first update (only remembers F); 2: nbx_diis_update with
 *          (diis_slot, diis_nd) -- the ring bookkeeping stays with the caller.
 *   dts_ready: st->d_dts holds the Dtot' table of d_dm_in (left by the previous call).
 *   h_out: 7 doubles of pinned (device-mapped) host memory: E_alpha, E_beta, |dD_alpha|, |dD_beta|, the
 *          eigensolver's two status words, then 1.0 -- stored last, once the six are visible to the host, which
 *          can clear that word before the call and poll it (see nbx_huz_cycle_scalars_dev).          */
typedef struct nbx_huz_state {
    int64_t nao, nocc_a, nocc_b;
    const double* d_packed; /* nbx_eri_pack of the whole tensor / of this rank's slab (NBX_HUZ_JK_PACKED) */
    const double* d_hv;     /* (2,N,N) hcore + V_emb */
    const double* d_ds;     /* (2,N,N) D_env S */
    const double* d_sb;     /* (2,N,N) the overlap, once per spin (tracked mode) */
    const double* d_x;      /* (N,N) S^-1/2 (guarded mode) */
    double* d_dts;          /* nbx_jk_dts_bytes() table or NULL */
    double* d_jk;           /* (3,N,N) */
    double* d_fock;         /* (2,N,N) h + V + vhf */
    double* d_vhf;          /* (2,N,N) */
    double* d_fock2;        /* (2,N,N) ... + Hz */
    double* d_tmp;          /* (2,N,N) guarded mode */
    double* d_fo;           /* (2,N,N) guarded mode */
    void* d_jk_work;
    size_t jk_work_bytes;   /* nbx_jk_packed_worksize / nbx_jk_dense_sym_worksize (nao, p0, p1, 2) */
    void* d_eig_work;
    size_t eig_work_bytes;  /* nbx_eigh_worksize(nao, 2) */
    void* d_geig_work;
    size_t geig_work_bytes; /* nbx_geig_refine_worksize(nao, 2) */
    int64_t diis_space;
    double* d_diis_xs;      /* (space, 2 N^2) */
    double* d_diis_es;
    double* d_diis_h;       /* (space+1)^2, initialised by the caller (row/column 0 = 1) */
                            /* d_diis_coef: nbx_diis_coef_doubles(space) doubles, zeroed by the caller */
    double* d_diis_coef;
    double* d_diis_xprev;   /* 2 N^2 */
    /* which J/K kernel builds the Fock matrix, and on which rows of (pq|rs) */
    int64_t jk_kind;        /* NBX_HUZ_JK_PACKED: d_packed (nbx_jk_packed[_fock]); NBX_HUZ_JK_SYM: d_eri, the dense
                               tensor / row slab (nbx_jk_dense_sym: every N, falls to nbx_jk_dense inside) */
    int64_t jk_p0, jk_p1;   /* this rank's slab rows [p0, p1) of the first AO index: 0, nao = the whole tensor;
                               p0 == p1 = an empty slab (contributes zero) */
    const double* d_eri;    /* NBX_HUZ_JK_SYM: slab rows [p0, p1) of the dense (N,N,N,N) tensor */
} nbx_huz_state;
#define NBX_HUZ_JK_PACKED 0
#define NBX_HUZ_JK_SYM 1
int nbx_huz_cycle(nbx_ctx* ctx, const nbx_huz_state* st, const double* d_dm_in, const double* d_c_in,
                  double* d_dm_out, double* d_c_out, double* d_v_out, double* d_w_out, double* d_hz_out, int mode,
                  int refine_iters, int diis_mode, int diis_slot, int diis_nd, int dts_ready, double* h_out,
                  int* d_status_out);
/* The same cycle in two calls for runs over several GPUs (one process per GPU): the J/K
 * contribution of this rank's slab of (pq|rs) rows into st->d_jk ((3,N,N), additive over slabs) --
 *     nbx_huz_cycle_jk;   all-reduce (sum) of st->d_jk queued on the same stream by the caller (RCCL);
 *     nbx_huz_cycle_post: Fock assembly from the summed J/K (nbx_fock_uhf) and the rest of the cycle exactly
 * as nbx_huz_cycle queues it -- so every rank keeps the one-cycle look-ahead, the purified early cycles and the
 * tracked eigensolver, and (the replicated part being deterministic) holds bitwise the same matrices.
 * nbx_huz_cycle itself takes this route, without the collective, whenever the state is not "packed kernel on the
 * whole tensor" (N < 100, sizes without a packed instance: NBX_HUZ_JK_SYM).  st->d_dts is not used by _jk. */
int nbx_huz_cycle_jk(nbx_ctx* ctx, const nbx_huz_state* st, const double* d_dm_in);
int nbx_huz_cycle_post(nbx_ctx* ctx, const nbx_huz_state* st, const double* d_dm_in, const double* d_c_in,
                       double* d_dm_out, double* d_c_out, double* d_v_out, double* d_w_out, double* d_hz_out, int mode,
                       int refine_iters, int diis_mode, int diis_slot, int diis_nd, double* h_out, int* d_status_out);

/* ------------------------------------------------------------------ fused mu-shift SCF cycle
 * One cycle of the SCF the mu-shift embedding runs -- `embedded_scf.kernel()` at nbed/driver.py:533 on the hcore
 * patched with mu S D_env S + V_emb (:500-538), i.e. PySCF's scf.hf.kernel: CDIIS from cycle 1, eig, aufbau
 * density, get_veff, energy_tot, |dE| and orbital-gradient test -- queued by ONE call, nothing synchronises.
 * Also the global mean field's kernel() (driver.py:112-191).  The state block is nbx_huz_state with
 *   d_hv        = the kernel's h1e per spin (2,N,N)  (hcore + mu P + V_emb, or hcore twice);
 *   d_sb, d_x, d_jk, d_tmp, d_fo, d_fock2 (scratch), the J/K, eigensolver and refinement workspaces as for
 *   nbx_huz_cycle; d_ds, d_fock, d_vhf, d_dts unused (may be NULL);
 *   the DIIS ring for CDIIS: diis_space = 8, d_diis_xs / d_diis_es (8, 2 N^2), d_diis_h (9 x 9, row/column 0 = 1),
 *   d_diis_coef zeroed, d_diis_xprev (2 N^2) receives the extrapolated Fock matrix.
 * Cycle:  [diis_on: err = F D S - S D F of (d_dm_in, d_fock_in) per spin, pyscf.lib.diis.DIIS.update(F, err) with the
 *   caller's ring position (diis_slot, diis_nd)]  ->  F C = S C eps (mode 0 guarded: Loewdin step + nbx_eigh_warm_ex
 *   warm-started from d_c_in = the previous cycle's d_v_out, NULL = cold; mode 1 tracked: nbx_geig_refine from
 *   d_c_in = the previous cycle's d_c_out, status in d_status_out)  ->  d_dm_out = C_occ C_occ^T  ->  J/K of d_dm_out,
 *   d_fock_out = d_hv + J - K[x], d_vhf_out = J - K[x]  ->  [want_grad: sum of squares of the virtual-occupied block of
 *   C^T F_out C per spin]  ->  scalars.
 * h_out (pinned, device-mapped host memory; the caller clears the last word and polls it):
 *   tr[(h + vhf/2) D] alpha, beta (their sum + E_nuc = energy_tot), |D - D_in|_F alpha, beta, the eigensolver's two
 *   status words (as nbx_huz_cycle), [want_grad: the two gradient sums], 1.0  -- 7 or 9 doubles.
 * Same kernels, order and operands as issuing the steps one by one: bit-identical results.                        */
int nbx_mu_cycle(nbx_ctx* ctx, const nbx_huz_state* st, const double* d_dm_in, const double* d_fock_in,
                 const double* d_c_in, double* d_dm_out, double* d_fock_out, double* d_vhf_out, double* d_c_out,
                 double* d_v_out, double* d_w_out, int mode, int refine_iters, int diis_on, int diis_slot, int diis_nd,
                 int want_grad, double* h_out, int* d_status_out);
/* The pieces.  nbx_mu_cycle_solve: CDIIS -> eigenproblem -> density (no J/K).  nbx_mu_cycle_fock: J/K + Fock matrix of
 * d_dm on this rank's WHOLE tensor, gradient if d_c != NULL, scalars (mode: whose status words travel with them --
 * 0 the guarded solver's, 1 d_status_tracked, -1 none: h_out is then 5 or 7 doubles) -- alone it is the build of the
 * starting density and of the conv_check cycle.  Several ranks (SURVEY 8e, J/K row slabs): nbx_mu_cycle_solve,
 * nbx_huz_cycle_jk on the new density, the caller's all-reduce of st->d_jk, nbx_mu_cycle_fock_post.              */
int nbx_mu_cycle_solve(nbx_ctx* ctx, const nbx_huz_state* st, const double* d_dm_in, const double* d_fock_in,
                       const double* d_c_in, double* d_dm_out, double* d_c_out, double* d_v_out, double* d_w_out,
                       int mode, int refine_iters, int diis_on, int diis_slot, int diis_nd, int* d_status_out);
int nbx_mu_cycle_fock(nbx_ctx* ctx, const nbx_huz_state* st, const double* d_dm, const double* d_dm_old,
                      const double* d_c, double* d_fock_out, double* d_vhf_out, int mode, const int* d_status_tracked,
                      double* h_out);
int nbx_mu_cycle_fock_post(nbx_ctx* ctx, const nbx_huz_state* st, const double* d_dm, const double* d_dm_old,
                           const double* d_c, double* d_fock_out, double* d_vhf_out, int mode,
                           const int* d_status_tracked, double* h_out);

/* ------------------------------------------------------------------ exchange-correlation evaluation
 * (E_xc, v_xc) of a two-spin density matrix on the stored grid arrays -- what the reference gets from PySCF's
 * numint + libxc behind dft.UKS.get_veff (nbed/driver.py:155-191, 315-431, 845-852, 1138-1231): three passes,
 * csrc/xc.hip.  d_ao (npts, nao) AO values, d_dao (3, npts, nao) their gradients (nbx_eval_ao, transformed to the
 * working AOs by the caller), row-major; nothing is read back by these calls.
 * nbx_xc_rho:  d_rho (2, npts) and d_grad (2, 3, npts) of d_dm (2, nao, nao; symmetric): rho_x = diag(ao D_x ao^T),
 *   grad rho_x = 2 diag(ao D_x dao^T); c = ao D_x is formed tile by tile on the matrix cores for both spins and
 *   reduced against ao / dao in the epilogue (each array read once).  nao <= 640.
 * nbx_xc_functional:  energy density and first derivatives, written out analytically, of `code` (the semi-local part:
 *   NBX_XC_SLATER; _LDA_VWN_RPA = Slater + VWN(RPA); _LDA_VWN5 = Slater + VWN5, PySCF's default "lda,vwn"; _B3LYP =
 *   0.08 Slater + 0.72 B88 + 0.19 VWN(RPA) + 0.81 LYP, libxc's HYB_GGA_XC_B3LYP without its 0.2 exact exchange), d_w
 *   (npts) the quadrature weights.  Densities are clamped from below at rho_floor / 2 and points with
 *   rho_a + rho_b <= rho_floor carry no weight.  Out: d_vr (2, npts) = w dE/drho_x; d_vec (2, 3, npts) =
 *   w (2 dE/dsigma_xx grad rho_x + dE/dsigma_ab grad rho_other); d_sums[0] = E_xc, d_sums[1] = the integrated
 *   electron count (summed in a fixed order).  d_work: nbx_xc_functional_worksize() bytes.
 * nbx_xc_vmat:  d_vxc (2, nao, nao) = V + V^T, V_x[m][n] = sum_g ao[g][m] (d_vr[x][g] / 2 ao[g][n] + sum_a
 *   d_vec[x][a][g] dao[a][g][n]); the second factor is built on the fly as an operand of the product.  Split over
 *   chunks of grid points whose partial matrices are added in a fixed order.  d_work: nbx_xc_vmat_worksize() bytes. */
#define NBX_XC_SLATER 0
#define NBX_XC_LDA_VWN_RPA 1
#define NBX_XC_LDA_VWN5 2
#define NBX_XC_B3LYP 3
int nbx_xc_rho(nbx_ctx* ctx, int64_t npts, int64_t nao, const double* d_ao, const double* d_dao, const double* d_dm,
               double* d_rho, double* d_grad);
size_t nbx_xc_functional_worksize(int64_t npts);
int nbx_xc_functional(nbx_ctx* ctx, int code, int64_t npts, const double* d_rho, const double* d_grad, const double* d_w,
                      double rho_floor, double* d_vr, double* d_vec, double* d_sums, void* d_work, size_t work_bytes);
size_t nbx_xc_vmat_worksize(int64_t npts, int64_t nao);
int nbx_xc_vmat(nbx_ctx* ctx, int64_t npts, int64_t nao, const double* d_ao, const double* d_dao, const double* d_vr,
                const double* d_vec, double* d_vxc, void* d_work, size_t work_bytes);

/* ------------------------------------------------------------------ quadrature grid producers
 * What the reference gets from PySCF behind scf.UKS(...) (nbed/driver.py:86-104,315-431): `dft.gen_grid`'s Becke
 * partition and `numint.eval_ao`.  Inputs of the exchange-correlation part of V_emb (driver.py:845-852), not of
 * the SCF hot path; both one thread per grid point.
 * nbx_becke_share: d_share[g] = w_owner / sum_i w_i of Becke's cell functions (three smoothing iterations, size
 *   adjustment a_ij given by the caller) at the points d_pts (npts,3); d_centres (natm,3), d_aij and
 *   d_inv_dist = 1 / |R_i - R_j| (any finite value on the diagonal) are (natm,natm) (kept in LDS up to 96 atoms).
 * nbx_eval_ao: values d_out (npts, ncart) and, if d_dout != NULL, gradients (3, npts, ncart) of the contracted
 *   Cartesian Gaussians x^l y^m z^n sum_k c_k exp(-a_k r^2), l + m + n <= 3.  d_shell_i (nshell,4) =
 *   {first component, components, first primitive, primitives}; d_shell_centre (nshell,3); d_comp_lmn (ncart,4)
 *   = {l, m, n, offset of the component's coefficients in d_coefs}; max_prim = the largest primitive count
 *   (<= 24, NBX_E_UNSUPPORTED beyond).  The spherical transform is the caller's GEMM.               */
int nbx_becke_share(nbx_ctx* ctx, int64_t npts, const double* d_pts, int64_t natm, const double* d_centres,
                    const double* d_aij, const double* d_inv_dist, int64_t owner, double* d_share);
int nbx_eval_ao(nbx_ctx* ctx, int64_t npts, const double* d_pts, int64_t nshell, const int* d_shell_i,
                const double* d_shell_centre, const int* d_comp_lmn, const double* d_exps, const double* d_coefs,
                int64_t ncart, int64_t max_prim, double* d_out, double* d_dout);
/* ------------------------------------------------------------------ density by purification
 * The projector P on the nocc LOWEST eigenvectors of each symmetric matrix d_f (batch, n, n; an orthonormal
 * basis), i.e. C_occ C_occ^T of `eigh` + aufbau occupation (nbed/scf/huzinaga_scf.py:166-174) without the
 * eigenvectors: trace-correcting purification (SP2), one (n x n) product per step, everything decided on
 * the device.  For SCF cycles whose Fock matrix still moves too much for a warm-started eigensolver.
 *   nocc_a, nocc_b : occupied levels of matrix 0 and of the others (batch = 2: alpha, beta); a batch of more
 *                    than two needs nocc_a == nocc_b (NBX_E_INVALID otherwise)
 *   d_p            : out (batch, n, n)
 *   d_work         : nbx_purify_worksize() bytes;  max_iter <= 0: the limit (72 steps)
 *   d_status[b]    : > 0 steps taken; < 0 no gap between levels nocc and nocc + 1 was resolved (or the
 *                    matrix is not finite): d_p is then meaningless                                      */
size_t nbx_purify_worksize(int64_t n, int64_t batch);
int nbx_purify(nbx_ctx* ctx, int64_t n, int64_t batch, const double* d_f, int64_t nocc_a, int64_t nocc_b, double* d_p,
               void* d_work, size_t work_bytes, int max_iter, int* d_status);

/* d_x[i] <- (|d_x[i]| < tol ? 0 : d_x[i]) * scale over n doubles: the 1e-8 truncation
 * (nbed/ham_builder.py:213-214) and the 1/2 of build() (:254) applied to a SPATIAL block, for callers
 * that keep the three unique spin blocks (aaaa, bbbb, aabb) instead of the 16x larger scattered
 * spin-orbital tensor.                                                                          */
int nbx_threshold_scale(nbx_ctx* ctx, int64_t n, double tol, double scale, double* d_x);

/* ------------------------------------------------------------------ AO integrals of a real molecule (HOST)
 * (pq|rs) over contracted Gaussian shells of angular momentum <= 3, dense (nao, nao, nao, nao) in HOST
 * memory: the producer the reference reaches through PySCF/libcint (gto.Mole.intor("int2e"), implied by
 * scf.UKS(mol).kernel() at nbed/driver.py:155-191 and ao2mo at nbed/ham_builder.py:139-170).  Input of
 * the hot path, produced once per molecule: host threads, no GPU, no context.
 *   ang, nprim, nfunc (nshell)   angular momentum, primitive count, AO count (2l+1, or (l+1)(l+2)/2
 *                                to keep the Cartesian components) of each shell
 *   centres (nshell, 3)          Bohr
 *   exps, coefs                  primitives of all shells, concatenated; coefs carry the radial
 *                                normalisation (common to the components of a shell)
 *   sph                          per shell a (nfunc, ncart) matrix from the Cartesian components
 *                                (xx xy xz yy yz zz; xxx xxy xxz xyy xyz xzz yyy yyz yzz zzz)
 *                                to its AOs, concatenated; only read for l >= 2
 *   cutoff                       Schwarz bound below which a shell quartet is skipped (1e-16: below fp64
 *                                resolution of the O(1) integrals)
 *   nthreads                     <= 0: all hardware threads                                              */
int nbx_host_eri(int nshell, const int* ang, const int* nprim, const int* nfunc, const double* centres,
                 const double* exps, const double* coefs, const double* sph, double cutoff, int nthreads,
                 double* out);
/* Overlap, kinetic-energy and nuclear-attraction matrices (nao, nao) of the same shells, HOST memory:
 * intor("int1e_ovlp"), ("int1e_kin"), ("int1e_nuc") behind get_ovlp() / get_hcore() (nbed/driver.py:155-191).
 *   charges (natm), atom_xyz (natm, 3): the point charges of V (nuclei; Bohr)                      */
int nbx_host_1e(int nshell, const int* ang, const int* nprim, const int* nfunc, const double* centres,
                const double* exps, const double* coefs, const double* sph, int natm, const double* charges,
                const double* atom_xyz, int nthreads, double* s_out, double* t_out, double* v_out);

#ifdef __cplusplus
}
#endif
#endif /* NBX_H */
