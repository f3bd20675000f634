// libnbx: one Huzinaga-projected UHF SCF cycle as ONE call (include/nbx.h "fused SCF cycle"), and below it one
// cycle of the mu-shift SCF (PySCF's scf.hf.kernel control flow) the same way.
//
// The cycle of nbed/scf/huzinaga_scf.py:154-201 is a chain of 12 (settled) to 40 (first cycles, guarded
// eigensolver) launches.  Issued from Python one by one -- a ctypes call with 10-17 marshalled arguments
// and a handful of tensor allocations each -- the host needs 0.3-0.8 ms per early cycle, more than the
// GPU does, and the GPU idles 100-300 us per cycle until the eigensolver settles
// (tools/trace_cycle.py).  Here the chain is queued by one C call from a state block the host fills
// once per SCF; all arithmetic stays in the kernels the step-by-step path uses (same entry points,
// same order, same operands): results are bit-identical to that path.
#include <cstdlib>

#include "nbx_common.h"

// This rank's J/K contribution from the density d_dm_in into st->d_jk: the whole tensor or the slab rows
// [jk_p0, jk_p1) of it, additive over slabs (nbx_jk_packed / nbx_jk_dense_sym) -- the caller sums the ranks'
// (3,N,N) partials with one all-reduce queued on the same stream, then calls nbx_huz_cycle_post.
static int huz_jk_slab(nbx_ctx* ctx, const nbx_huz_state* st, const double* d_dm_in, int64_t p0, int64_t p1) {
    const int64_t N = st->nao;
    if (st->jk_kind == NBX_HUZ_JK_PACKED) {
        NBX_CHECK_ARG(st->d_packed || p0 == p1);
        return nbx_jk_packed(ctx, N, p0, p1, st->d_packed, d_dm_in, 2, st->d_jk, st->d_jk_work, st->jk_work_bytes);
    }
    NBX_CHECK_ARG(st->jk_kind == NBX_HUZ_JK_SYM);
    NBX_CHECK_ARG(st->d_eri || p0 == p1);
    return nbx_jk_dense_sym(ctx, N, p0, p1, st->d_eri, d_dm_in, 2, st->d_jk, st->d_jk_work, st->jk_work_bytes);
}

// (jk_p0 == jk_p1 is an EMPTY slab -- a rank that holds no rows, n < world -- whose contribution is zero: not the
//  whole tensor)
static bool huz_whole_tensor(const nbx_huz_state* st) { return st->jk_p0 == 0 && st->jk_p1 == st->nao; }

static int huz_cycle_rest(nbx_ctx* ctx, const nbx_huz_state* st, const double* d_dm_in, const double* d_c_in,
                          double* d_dm_out, double* d_c_out, double* d_v_out, double* d_w_out, double* d_hz_out,
                          int mode, int refine_iters, int diis_mode, int diis_slot, int diis_nd, double* h_out,
                          int* d_status_out);

// D[x] = C[x][:, :nocc_x] C[x][:, :nocc_x]^T: the solvers return ascending eigenvalues, so the aufbau-occupied
// orbitals (get_occ: the n_alpha / n_beta lowest) are the leading columns
static int cycle_density(nbx_ctx* ctx, int64_t N, int64_t nocc_a, int64_t nocc_b, const double* d_c, double* d_dm) {
    const int64_t n2 = N * N;
    int rc = NBX_OK;
    if (nocc_a == nocc_b && nocc_a > 0)
        return nbx_gemm(ctx, 'N', 'T', N, N, nocc_a, 1.0, d_c, N, n2, d_c, N, n2, 0.0, d_dm, N, n2, 2);
    const int64_t nocc[2] = {nocc_a, nocc_b};
    for (int x = 0; x < 2; ++x) {
        if (nocc[x] > 0) {
            rc = nbx_gemm(ctx, 'N', 'T', N, N, nocc[x], 1.0, d_c + x * n2, N, 0, d_c + x * n2, N, 0, 0.0, d_dm + x * n2,
                          N, 0, 1);
        } else {
            rc = nbx_memset(ctx, d_dm + x * n2, 0, (size_t)n2 * sizeof(double));
        }
        if (rc != NBX_OK) return rc;
    }
    return rc;
}

extern "C" int nbx_huz_cycle_jk(nbx_ctx* ctx, const nbx_huz_state* st, const double* d_dm_in) {
    NBX_CHECK_ARG(ctx && st && d_dm_in);
    NBX_CHECK_ARG(st->nao > 0 && st->d_jk && st->d_jk_work);
    const int64_t p0 = st->jk_p0, p1 = st->jk_p1;
    NBX_CHECK_ARG(p0 >= 0 && p0 <= p1 && p1 <= st->nao);
    return huz_jk_slab(ctx, st, d_dm_in, p0, p1);
}

extern "C" int nbx_huz_cycle_post(nbx_ctx* ctx, const nbx_huz_state* st, const double* d_dm_in, const double* d_c_in,
                                  double* d_dm_out, double* d_c_out, double* d_v_out, double* d_w_out, double* d_hz_out,
                                  int mode, int refine_iters, int diis_mode, int diis_slot, int diis_nd, double* h_out,
                                  int* d_status_out) {
    NBX_CHECK_ARG(ctx && st && d_dm_in && d_dm_out && d_c_out && d_w_out && d_hz_out && h_out);
    NBX_CHECK_ARG(st->nao > 0 && st->d_hv && st->d_ds && st->d_jk && st->d_fock && st->d_vhf && st->d_fock2);
    // ---- Fock assembly (:157-160) from the summed J/K: F[x] = (h + V_emb)[x] + J - K[x]
    int rc = nbx_fock_uhf(ctx, st->nao, st->d_hv, 3, nullptr, st->d_jk, st->d_fock, st->d_vhf);
    if (rc != NBX_OK) return rc;
    return huz_cycle_rest(ctx, st, d_dm_in, d_c_in, d_dm_out, d_c_out, d_v_out, d_w_out, d_hz_out, mode, refine_iters,
                          diis_mode, diis_slot, diis_nd, h_out, d_status_out);
}

extern "C" int nbx_huz_cycle(nbx_ctx* ctx, const nbx_huz_state* st, const double* d_dm_in, const double* d_c_in,
                             double* d_dm_out, double* d_c_out, double* d_v_out, double* d_w_out, double* d_hz_out,
                             int mode, int refine_iters, int diis_mode, int diis_slot, int diis_nd, int dts_ready,
                             double* h_out, int* d_status_out) {
    NBX_CHECK_ARG(ctx && st && d_dm_in && d_dm_out && d_c_out && d_w_out && d_hz_out && h_out);
    NBX_CHECK_ARG(st->nao > 0 && st->d_hv && st->d_ds && st->d_jk && st->d_fock && st->d_vhf && st->d_fock2 &&
                  st->d_jk_work);
    const int64_t N = st->nao;
    int rc;

    // ---- Fock build (:156-160)
    if (st->jk_kind == NBX_HUZ_JK_PACKED && huz_whole_tensor(st)) {
        // J/K on the packed integrals with the Fock assembly in its reduction
        NBX_CHECK_ARG(st->d_packed);
        rc = nbx_jk_packed_fock(ctx, N, st->d_packed, d_dm_in, st->d_hv, st->d_jk, st->d_fock, st->d_vhf, st->d_jk_work,
                                st->jk_work_bytes, (dts_ready && st->d_dts) ? st->d_dts : nullptr);
        if (rc != NBX_OK) return rc;
    } else {
        // the symmetric kernel on the dense tensor (N < 97; sizes the packed kernel has no instance for) with the Fock assembly
        // in its reduction where it has the whole tensor; else (odd N, a one-rank "slab" run): J/K, then the assembly as its
        // own launch
        rc = NBX_E_UNSUPPORTED;
        if (st->jk_kind == NBX_HUZ_JK_SYM && huz_whole_tensor(st) && st->d_eri)
            rc = nbx_jk_dense_sym_fock(ctx, N, st->d_eri, d_dm_in, st->d_hv, st->d_jk, st->d_fock, st->d_vhf, st->d_jk_work,
                                       st->jk_work_bytes);
        if (rc == NBX_E_UNSUPPORTED) {
            rc = nbx_huz_cycle_jk(ctx, st, d_dm_in);
            if (rc != NBX_OK) return rc;
            rc = nbx_fock_uhf(ctx, N, st->d_hv, 3, nullptr, st->d_jk, st->d_fock, st->d_vhf);
        }
        if (rc != NBX_OK) return rc;
    }
    return huz_cycle_rest(ctx, st, d_dm_in, d_c_in, d_dm_out, d_c_out, d_v_out, d_w_out, d_hz_out, mode, refine_iters,
                          diis_mode, diis_slot, diis_nd, h_out, d_status_out);
}

// Everything of a cycle after the Fock matrices st->d_fock / st->d_vhf exist.
static int huz_cycle_rest(nbx_ctx* ctx, const nbx_huz_state* st, const double* d_dm_in, const double* d_c_in,
                          double* d_dm_out, double* d_c_out, double* d_v_out, double* d_w_out, double* d_hz_out,
                          int mode, int refine_iters, int diis_mode, int diis_slot, int diis_nd, double* h_out,
                          int* d_status_out) {
    NBX_CHECK_ARG(mode == 0 || mode == 1 || mode == 2);
    NBX_CHECK_ARG(diis_mode >= 0 && diis_mode <= 2);
    const int64_t N = st->nao, n2 = N * N;
    int rc;
    // Huzinaga operator and F += Hz in one launch (:159-160); the operator of the pre-DIIS Fock matrix is
    // what the loop returns (:206)
    rc = nbx_huzinaga_fused(ctx, N, 2, st->d_fock, st->d_ds, 1.0, d_hz_out, st->d_fock2);
    if (rc != NBX_OK) return rc;

    // ---- DIIS (:162-164): pyscf.lib.diis.DIIS.update -- the first call only remembers F
    const double* f_use = st->d_fock2;
    if (diis_mode == 1) {
        NBX_CHECK_ARG(st->d_diis_xprev);
        rc = nbx_memcpy_d2d(ctx, st->d_diis_xprev, st->d_fock2, (size_t)(2 * n2) * sizeof(double));
        if (rc != NBX_OK) return rc;
    } else if (diis_mode == 2) {
        NBX_CHECK_ARG(st->d_diis_xprev && st->d_diis_xs && st->d_diis_es && st->d_diis_h && st->d_diis_coef);
        rc = nbx_diis_update(ctx, 2 * n2, st->diis_space, diis_slot, diis_nd, st->d_fock2, st->d_diis_xprev,
                             st->d_diis_xs, st->d_diis_es, st->d_diis_h, st->d_diis_coef);
        if (rc != NBX_OK) return rc;
        f_use = st->d_diis_xprev;  // the extrapolated matrix
    }

    // ---- eigenproblem F C = S C eps (:166-169)
    const int* d_status = nullptr;
    if (mode == 2) {
        // density without eigenvectors: the projector on the nocc lowest levels of X F X by purification
        // (purify.hip), D = X P X.  For cycles whose Fock matrix still moves too much for a warm start to help;
        // d_c_out / d_w_out are NOT written (this cycle has no orbitals).
        NBX_CHECK_ARG(st->d_x && st->d_eig_work && st->d_tmp && st->d_fo && d_status_out);
        NBX_CHECK_ARG(st->eig_work_bytes >= nbx_purify_worksize(N, 2));
        // d_v_out given: X F X is left there (an eigensolve of this cycle's matrix can then be run later, or
        // beside the following cycles on another stream: the orbitals of a purified cycle on demand)
        double* fo = d_v_out ? d_v_out : st->d_fo;
        rc = nbx_gemm(ctx, 'N', 'N', N, N, N, 1.0, st->d_x, N, 0, f_use, N, n2, 0.0, st->d_tmp, N, n2, 2);
        if (rc != NBX_OK) return rc;
        rc = nbx_gemm(ctx, 'N', 'N', N, N, N, 1.0, st->d_tmp, N, n2, st->d_x, N, 0, 0.0, fo, N, n2, 2);
        if (rc != NBX_OK) return rc;
        rc = nbx_purify(ctx, N, 2, fo, st->nocc_a, st->nocc_b, st->d_tmp, st->d_eig_work, st->eig_work_bytes,
                        refine_iters, d_status_out);
        if (rc != NBX_OK) return rc;
        rc = nbx_gemm(ctx, 'N', 'N', N, N, N, 1.0, st->d_x, N, 0, st->d_tmp, N, n2, 0.0, st->d_fo, N, n2, 2);
        if (rc != NBX_OK) return rc;
        rc = nbx_gemm(ctx, 'N', 'N', N, N, N, 1.0, st->d_fo, N, n2, st->d_x, N, 0, 0.0, d_dm_out, N, n2, 2);
        if (rc != NBX_OK) return rc;
        return nbx_huz_cycle_scalars_dts(ctx, N, st->d_hv, 3, nullptr, st->d_vhf, d_hz_out, d_dm_out, d_dm_in, h_out,
                                         d_status_out, 2, st->d_dts);
    }
    if (mode == 1) {  // tracked: refine the previous cycle's (eps, C) on the pencil (F, S), no fallback queued
        NBX_CHECK_ARG(d_c_in && st->d_sb && st->d_geig_work && d_status_out);
        rc = nbx_geig_refine(ctx, N, 2, f_use, st->d_sb, d_c_in, d_w_out, d_c_out, d_status_out, st->d_geig_work,
                             st->geig_work_bytes, refine_iters);
        if (rc != NBX_OK) return rc;
        d_status = d_status_out;
    } else {  // guarded: Loewdin step, warm-started eigensolver (refinement, Jacobi behind it on the device)
        NBX_CHECK_ARG(st->d_x && st->d_eig_work && st->d_tmp && st->d_fo && d_v_out);
        rc = nbx_gemm(ctx, 'N', 'N', N, N, N, 1.0, st->d_x, N, 0, f_use, N, n2, 0.0, st->d_tmp, N, n2, 2);
        if (rc != NBX_OK) return rc;
        rc = nbx_gemm(ctx, 'N', 'N', N, N, N, 1.0, st->d_tmp, N, n2, st->d_x, N, 0, 0.0, st->d_fo, N, n2, 2);
        if (rc != NBX_OK) return rc;
        // d_c_in: the previous cycle's ORTHONORMAL-basis vectors (NULL: cold start)
        rc = nbx_eigh_warm_ex(ctx, N, 2, st->d_fo, d_c_in, d_w_out, d_v_out, st->d_eig_work, st->eig_work_bytes,
                              refine_iters);
        if (rc != NBX_OK) return rc;
        rc = nbx_gemm(ctx, 'N', 'N', N, N, N, 1.0, st->d_x, N, 0, d_v_out, N, n2, 0.0, d_c_out, N, n2, 2);
        if (rc != NBX_OK) return rc;
        d_status = reinterpret_cast<const int*>(static_cast<const char*>(st->d_eig_work) +
                                                nbx_eigh_status_offset(N, 2));
    }

    // ---- density of the aufbau occupation (:170-174) and the energy and convergence scalars (:181-194) with the
    // eigensolver's status words, stored by the kernel into (pinned) host memory; it also leaves the Dtot' table of the
    // new density for the next build.  One launch where the occupied block is at most 64 orbitals wide (the same bits
    // as the two launches it stands for: elementwise.hip), else the product and the scalars kernel.
    static const bool fused_density = getenv("NBX_FUSED_DENSITY") == nullptr || atoi(getenv("NBX_FUSED_DENSITY")) != 0;
    if (fused_density) {
        rc = nbx_density_scalars_launch(ctx, N, st->d_hv, 3, nullptr, st->d_vhf, d_hz_out, d_c_out, st->nocc_a, st->nocc_b,
                                        d_dm_out, d_dm_in, h_out, d_status, 2, st->d_dts);
        if (rc != NBX_E_UNSUPPORTED) return rc;
    }
    rc = cycle_density(ctx, N, st->nocc_a, st->nocc_b, d_c_out, d_dm_out);
    if (rc != NBX_OK) return rc;
    return nbx_huz_cycle_scalars_dts(ctx, N, st->d_hv, 3, nullptr, st->d_vhf, d_hz_out, d_dm_out, d_dm_in, h_out,
                                     d_status, 2, st->d_dts);
}

// ===================================================================================================== mu-shift
// One cycle of the SCF behind nbed/driver.py:533 (`embedded_scf.kernel()` on the hcore patched with mu P + V_emb:
// PySCF's scf.hf.kernel) as one call.  The state block is nbx_huz_state with d_hv = the kernel's
// h1e per spin (d_ds, d_fock, d_vhf unused: the Fock matrix and vhf of a cycle belong to its result set, because
// the NEXT cycle's CDIIS step reads them), the DIIS ring sized for CDIIS (space 8) and d_diis_xprev receiving the
// extrapolated Fock matrix.  Same kernels, order and operands as GpuUHF's step-by-step loop: bit-identical results
// in guarded mode.

// CDIIS -> eigenproblem -> aufbau density
extern "C" int nbx_mu_cycle_solve(nbx_ctx* ctx, const nbx_huz_state* st, const double* d_dm_in, const double* d_fock_in,
                                  const double* d_c_in, double* d_dm_out, double* d_c_out, double* d_v_out,
                                  double* d_w_out, int mode, int refine_iters, int diis_on, int diis_slot, int diis_nd,
                                  int* d_status_out) {
    NBX_CHECK_ARG(ctx && st && d_dm_in && d_fock_in && d_dm_out && d_c_out && d_w_out);
    NBX_CHECK_ARG(st->nao > 0 && st->d_sb && st->d_tmp && st->d_fo && st->d_fock2);
    NBX_CHECK_ARG(mode == 0 || mode == 1);
    const int64_t N = st->nao, n2 = N * N;
    int rc;
    const double* f_use = d_fock_in;
    if (diis_on) {
        // pyscf.scf.diis.CDIIS.update(s, d, f): error vector (S D F)^T - S D F = F D S - S D F per spin, then
        // lib.diis.DIIS.update(f, xerr=err) -- ring bookkeeping (slot, nd) with the caller
        NBX_CHECK_ARG(st->d_diis_xprev && st->d_diis_xs && st->d_diis_es && st->d_diis_h && st->d_diis_coef);
        rc = nbx_gemm(ctx, 'N', 'N', N, N, N, 1.0, st->d_sb, N, 0, d_dm_in, N, n2, 0.0, st->d_tmp, N, n2, 2);
        if (rc != NBX_OK) return rc;
        rc = nbx_gemm(ctx, 'N', 'N', N, N, N, 1.0, st->d_tmp, N, n2, d_fock_in, N, n2, 0.0, st->d_fo, N, n2, 2);
        if (rc != NBX_OK) return rc;
        // (the error vector A^T - A of A = S D F is formed inside the push kernel: the same numbers as nbx_antisym's)
        rc = nbx_diis_update_anti(ctx, 2 * n2, st->diis_space, diis_slot, diis_nd, d_fock_in, st->d_fo, N,
                                  st->d_diis_xprev, st->d_diis_xs, st->d_diis_es, st->d_diis_h, st->d_diis_coef);
        if (rc != NBX_OK) return rc;
        f_use = st->d_diis_xprev;
    }
    if (mode == 1) {  // tracked: the previous cycle's (eps, C) refined on the pencil (F, S); no fallback queued
        NBX_CHECK_ARG(d_c_in && st->d_geig_work && d_status_out);
        rc = nbx_geig_refine(ctx, N, 2, f_use, st->d_sb, d_c_in, d_w_out, d_c_out, d_status_out, st->d_geig_work,
                             st->geig_work_bytes, refine_iters);
        if (rc != NBX_OK) return rc;
    } else {  // guarded: scipy.linalg.eigh(F, S) by the Loewdin step, warm-started from d_c_in (orthonormal basis)
        NBX_CHECK_ARG(st->d_x && st->d_eig_work && d_v_out);
        rc = nbx_gemm(ctx, 'N', 'N', N, N, N, 1.0, st->d_x, N, 0, f_use, N, n2, 0.0, st->d_tmp, N, n2, 2);
        if (rc != NBX_OK) return rc;
        rc = nbx_gemm(ctx, 'N', 'N', N, N, N, 1.0, st->d_tmp, N, n2, st->d_x, N, 0, 0.0, st->d_fo, N, n2, 2);
        if (rc != NBX_OK) return rc;
        rc = nbx_eigh_warm_ex(ctx, N, 2, st->d_fo, d_c_in, d_w_out, d_v_out, st->d_eig_work, st->eig_work_bytes,
                              refine_iters);
        if (rc != NBX_OK) return rc;
        rc = nbx_gemm(ctx, 'N', 'N', N, N, N, 1.0, st->d_x, N, 0, d_v_out, N, n2, 0.0, d_c_out, N, n2, 2);
        if (rc != NBX_OK) return rc;
    }
    return cycle_density(ctx, N, st->nocc_a, st->nocc_b, d_c_out, d_dm_out);
}

// What follows the Fock matrix of the new density: the orbital gradient (PySCF get_grad: the virtual-occupied block
// of C^T F C) and the cycle's scalars stored into pinned host memory --
//   h_out = E1+E2 alpha, beta ( tr[(h + vhf/2) D] per spin ), |dD| alpha, beta, [2 status words], [2 gradient sums], 1.0
static int mu_cycle_tail(nbx_ctx* ctx, const nbx_huz_state* st, const double* d_dm, const double* d_dm_old,
                         const double* d_c, const double* d_fock, const double* d_vhf, int mode,
                         const int* d_status_tracked, double* h_out) {
    const int64_t N = st->nao, n2 = N * N;
    int rc;
    const double* dtail = nullptr;
    if (d_c != nullptr) {
        rc = nbx_gemm(ctx, 'T', 'N', N, N, N, 1.0, d_c, N, n2, d_fock, N, n2, 0.0, st->d_tmp, N, n2, 2);
        if (rc != NBX_OK) return rc;
        rc = nbx_gemm(ctx, 'N', 'N', N, N, N, 1.0, st->d_tmp, N, n2, d_c, N, n2, 0.0, st->d_fo, N, n2, 2);
        if (rc != NBX_OK) return rc;
        rc = nbx_vo_sumsq(ctx, N, st->d_fo, st->nocc_a, st->nocc_b, st->d_tmp);  // (d_tmp is free again)
        if (rc != NBX_OK) return rc;
        dtail = st->d_tmp;
    }
    const int* d_status = nullptr;
    if (mode == 1) {
        NBX_CHECK_ARG(d_status_tracked);
        d_status = d_status_tracked;
    } else if (mode == 0) {
        NBX_CHECK_ARG(st->d_eig_work);
        d_status = reinterpret_cast<const int*>(static_cast<const char*>(st->d_eig_work) + nbx_eigh_status_offset(N, 2));
    }
    return nbx_cycle_scalars_launch(ctx, N, st->d_hv, 3, nullptr, d_vhf, nullptr, d_dm, d_dm_old, h_out, d_status,
                                    d_status ? 2 : 0, nullptr, dtail, dtail ? 2 : 0);
}

// Fock matrix of d_dm (J/K on this rank's whole tensor), gradient with d_c (NULL: none), scalars.  mode: whose
// status words travel with the scalars -- 0 the guarded solver's (in st->d_eig_work), 1 d_status_tracked, -1 none.
// (st->d_dts is not used: the build of a cycle contracts the density the same call has just made, so no earlier
//  kernel could have left its Dtot' table -- the packed build prepares its own.)
extern "C" int nbx_mu_cycle_fock(nbx_ctx* ctx, const nbx_huz_state* st, const double* d_dm, const double* d_dm_old,
                                 const double* d_c, double* d_fock_out, double* d_vhf_out, int mode,
                                 const int* d_status_tracked, double* h_out) {
    NBX_CHECK_ARG(ctx && st && d_dm && d_dm_old && d_fock_out && d_vhf_out && h_out);
    NBX_CHECK_ARG(st->nao > 0 && st->d_hv && st->d_jk && st->d_jk_work && st->d_tmp && st->d_fo);
    NBX_CHECK_ARG(mode >= -1 && mode <= 1);
    const int64_t N = st->nao;
    int rc;
    if (st->jk_kind == NBX_HUZ_JK_PACKED && huz_whole_tensor(st)) {
        NBX_CHECK_ARG(st->d_packed);
        rc = nbx_jk_packed_fock(ctx, N, st->d_packed, d_dm, st->d_hv, st->d_jk, d_fock_out, d_vhf_out, st->d_jk_work,
                                st->jk_work_bytes, nullptr);
        if (rc != NBX_OK) return rc;
    } else {
        rc = NBX_E_UNSUPPORTED;
        if (st->jk_kind == NBX_HUZ_JK_SYM && huz_whole_tensor(st) && st->d_eri)
            rc = nbx_jk_dense_sym_fock(ctx, N, st->d_eri, d_dm, st->d_hv, st->d_jk, d_fock_out, d_vhf_out, st->d_jk_work,
                                       st->jk_work_bytes);
        if (rc == NBX_E_UNSUPPORTED) {
            rc = nbx_huz_cycle_jk(ctx, st, d_dm);
            if (rc != NBX_OK) return rc;
            rc = nbx_fock_uhf(ctx, N, st->d_hv, 3, nullptr, st->d_jk, d_fock_out, d_vhf_out);
        }
        if (rc != NBX_OK) return rc;
    }
    return mu_cycle_tail(ctx, st, d_dm, d_dm_old, d_c, d_fock_out, d_vhf_out, mode, d_status_tracked, h_out);
}

// Several ranks: nbx_mu_cycle_solve; nbx_huz_cycle_jk on the new density; the caller's all-reduce of st->d_jk; this.
extern "C" int nbx_mu_cycle_fock_post(nbx_ctx* ctx, const nbx_huz_state* st, const double* d_dm, const double* d_dm_old,
                                      const double* d_c, double* d_fock_out, double* d_vhf_out, int mode,
                                      const int* d_status_tracked, double* h_out) {
    NBX_CHECK_ARG(ctx && st && d_dm && d_dm_old && d_fock_out && d_vhf_out && h_out);
    NBX_CHECK_ARG(st->nao > 0 && st->d_hv && st->d_jk && st->d_tmp && st->d_fo);
    NBX_CHECK_ARG(mode >= -1 && mode <= 1);
    const int rc = nbx_fock_uhf(ctx, st->nao, st->d_hv, 3, nullptr, st->d_jk, d_fock_out, d_vhf_out);
    if (rc != NBX_OK) return rc;
    return mu_cycle_tail(ctx, st, d_dm, d_dm_old, d_c, d_fock_out, d_vhf_out, mode, d_status_tracked, h_out);
}

extern "C" int nbx_mu_cycle(nbx_ctx* ctx, const nbx_huz_state* st, const double* d_dm_in, const double* d_fock_in,
                            const double* d_c_in, double* d_dm_out, double* d_fock_out, double* d_vhf_out,
                            double* d_c_out, double* d_v_out, double* d_w_out, int mode, int refine_iters, int diis_on,
                            int diis_slot, int diis_nd, int want_grad, double* h_out, int* d_status_out) {
    const int rc = nbx_mu_cycle_solve(ctx, st, d_dm_in, d_fock_in, d_c_in, d_dm_out, d_c_out, d_v_out, d_w_out, mode,
                                      refine_iters, diis_on, diis_slot, diis_nd, d_status_out);
    if (rc != NBX_OK) return rc;
    return nbx_mu_cycle_fock(ctx, st, d_dm_out, d_dm_in, want_grad ? d_c_out : nullptr, d_fock_out, d_vhf_out, mode,
                             d_status_out, h_out);
}
