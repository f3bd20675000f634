// Internal helpers shared by the libnbx translation units (gfx950 only).
#pragma once

#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <vector>

#include "nbx.h"

constexpr int NBX_PROF_SLOTS = 8;      // see include/nbx.h NBX_PROF_*
constexpr int NBX_PROF_MAX_EVENTS = 4096;

struct nbx_prof_slot {
    std::vector<hipEvent_t> start, stop;  // recorded pairs not yet read
    double ms_sum = 0.0;
    int64_t count = 0;
    int64_t seen = 0;  // launches of the slot since nbx_profile_reset (bracketed or not)
};

struct nbx_ctx {
    int device;
    hipStream_t stream;
    bool own_stream;
    double* d_scratch;      // small device scratch for reductions (NBX_SCRATCH_DOUBLES)
    double* h_pinned;       // pinned host mirror of the scratch
    int* d_counters;        // NBX_COUNTERS zero-initialised ints: arrival counters of kernels that
                            // synchronise across workgroups (each kernel leaves them zero again)
    bool profiling = false;
    unsigned prof_mask = ~0u;
    int prof_every = 1;     // nbx_profile_sample: one launch in this many is bracketed
    nbx_prof_slot prof[NBX_PROF_SLOTS];
};

// eigh_lds.hip
bool nbx_eigh_lds_supported(int64_t n);
size_t nbx_eigh_lds_worksize(int64_t n, int64_t batch);
int nbx_eigh_lds(nbx_ctx* ctx, int64_t n, int64_t batch, const double* d_a, const double* d_v0, double* d_w,
                 double* d_v, void* d_work, size_t work_bytes, int refine_iters, const int* d_skip = nullptr);
const int* nbx_eigh_lds_status_ptr(int64_t n, int64_t batch, const void* d_work);
size_t nbx_eigh_lds_status_offset(int64_t n, int64_t batch);
int* nbx_eigh_lds_skip_ptr(int64_t n, int64_t batch, void* d_work);
int nbx_apply_rotation_log_t(nbx_ctx* ctx, int n, int np_even, int steps, const void* d_rot, const int* d_flags,
                             const int* d_nsteps, const int* d_rank, double* d_vt);

// jk_sym.hip
int nbx_jk_dense_sym_fock(nbx_ctx* ctx, int64_t nao, const double* d_eri, const double* d_dm, const double* d_hv, double* d_jk,
                          double* d_fock, double* d_vhf, void* d_work, size_t work_bytes);

// gemm.hip
int nbx_gemm_q1_synth(nbx_ctx* ctx, int64_t m, int64_t n, int64_t k, const double* d_a, int64_t lda, uint64_t seed,
                      double scale, int64_t r, int64_t s0, double* d_y, int64_t ldy, int64_t stride_y, int64_t batch);
bool nbx_gemm_small_supported(int64_t m, int64_t n, int64_t k, int64_t batch);
int nbx_gemm_pair_scatter(nbx_ctx* ctx, int64_t pair_n, int64_t m, int64_t n, int64_t k, const double* d_a, int64_t lda,
                          const double* d_b, int64_t ldb, int64_t stride_b, double* d_c);
int nbx_gemm_tri(nbx_ctx* ctx, int64_t unit, int64_t tri_m0, int64_t nbatch, int64_t n, int64_t k, const double* d_a,
                 int64_t lda, int64_t stride_a, const double* d_b, int64_t ldb, double* d_c, int64_t ldc,
                 int64_t a_symn = 0);
int nbx_gemm_gated(nbx_ctx* ctx, char trans_a, char trans_b, int64_t m, int64_t n, int64_t k, double alpha,
                   const double* d_a, int64_t lda, int64_t stride_a, const double* d_b, int64_t ldb, int64_t stride_b,
                   double beta, double* d_c, int64_t ldc, int64_t stride_c, int64_t batch, const int* d_gate,
                   int gate_a, int gate_b);
int nbx_gemm_small_gated(nbx_ctx* ctx, char trans_a, char trans_b, int64_t m, int64_t n, int64_t k, double alpha,
                         const double* d_a, int64_t lda, int64_t stride_a, const double* d_b, int64_t ldb,
                         int64_t stride_b, double beta, double* d_c, int64_t ldc, int64_t stride_c, int64_t batch,
                         const int* d_gate, int gate_a, int gate_b, const double* d_b2 = nullptr,
                         double* d_c2 = nullptr, double* d_norm_part = nullptr);
// doubles of d_norm_part for an (m x n) pair product over `batch` entries (16 x 16 tiles)
inline int64_t nbx_gemm_small_norm_doubles(int64_t m, int64_t n, int64_t batch) {
    return batch * 2 * ((m + 15) / 16) * ((n + 15) / 16) * 2;
}

// elementwise.hip
// The launch behind nbx_huz_cycle_scalars_dev / _dts with two more options: d_hz == NULL (no Huzinaga operator in the
// energy) and d_dtail (dtail_n <= 64 device doubles stored behind the status words, ahead of the ready word).
int nbx_density_scalars_launch(nbx_ctx* ctx, int64_t nao, const double* d_hcore, int hcore_ndim, const double* d_vemb,
                               const double* d_vhf, const double* d_hz, const double* d_c, int64_t nocc_a, int64_t nocc_b,
                               double* d_dm_out, const double* d_dm_old, double* d_out, const int* d_tail, int64_t tail_n,
                               double* d_dts);
int nbx_diis_update_anti(nbx_ctx* ctx, int64_t n, int64_t space, int64_t slot, int64_t nd, const double* d_x,
                         const double* d_err, int64_t anti_n, double* d_xprev, double* d_xs, double* d_es, double* d_h,
                         double* d_coef);
int nbx_cycle_scalars_launch(nbx_ctx* ctx, int64_t nao, const double* d_hcore, int hcore_ndim, const double* d_vemb,
                             const double* d_vhf, const double* d_hz, const double* d_dm, const double* d_dm_old,
                             double* d_out, const int* d_tail, int64_t tail_n, double* d_dts, const double* d_dtail,
                             int64_t dtail_n);
// out[b] = A[b]^T - A[b], (batch, nao, nao), out of place
int nbx_antisym(nbx_ctx* ctx, int64_t nao, int64_t batch, const double* d_a, double* d_out);

// jk_sym.hip
bool nbx_jk_sym_supported(int64_t nao);
int nbx_jk_sym_reduce(nbx_ctx* ctx, const double* k1, const double* k2, double* d_k, int64_t N, int64_t p0, int64_t np,
                      int64_t ndm, int64_t t_begin, int L, int S, const double* d_j = nullptr,
                      const double* d_hv = nullptr, double* d_fock = nullptr, double* d_vhf = nullptr,
                      int k2_tile_order = 0, int k_lower = 0);

// eigh_refine.hip
bool nbx_eigh_refine_supported(int64_t n, int64_t batch);
size_t nbx_eigh_refine_worksize(int64_t n, int64_t batch);
// Queues the refinement of (A, V0); returns in *d_status_out the device int[batch] that is > 0
// for every matrix whose eigenpairs were accepted and written to d_w / d_v.
int nbx_eigh_refine(nbx_ctx* ctx, int64_t n, int64_t batch, const double* d_a, const double* d_v0, double* d_w,
                    double* d_v, void* d_work, int* d_jacobi_status, const int** d_status_out, int max_iter);
constexpr int NBX_EIGH_REFINE_ITERS = 3;  // default of nbx_eigh_warm
constexpr int NBX_EIGH_REFINE_MAX = 6;    // most a caller can queue (eigh_refine.hip RF_MAX_ITER)

// eigh_tridiag.hip
size_t nbx_eigh_tridiag_worksize(int64_t n, int64_t batch);
int nbx_eigh_tridiag(nbx_ctx* ctx, int64_t n, int64_t batch, const double* d_a, double* d_w, double* d_v,
                     void* d_work, size_t work_bytes, double* h_quality);
int nbx_eigh_tridiag_dev(nbx_ctx* ctx, int64_t n, int64_t batch, const double* d_a, double* d_w, double* d_v,
                         void* d_work, size_t work_bytes, int* d_status, int* d_skip);

// HIP-event bracket around a launch, active only while profiling is enabled.
struct nbx_prof_scope {
    nbx_ctx* ctx;
    int slot;
    hipEvent_t stop = nullptr;
    nbx_prof_scope(nbx_ctx* c, int s) : ctx(c), slot(s) {
        if (!ctx->profiling || !((ctx->prof_mask >> slot) & 1u) ||
            (int)ctx->prof[slot].start.size() >= NBX_PROF_MAX_EVENTS)
            return;
        if (ctx->prof[slot].seen++ % ctx->prof_every != 0) return;
        hipEvent_t a, b;
        if (hipEventCreate(&a) != hipSuccess) return;
        if (hipEventCreate(&b) != hipSuccess) { (void)hipEventDestroy(a); return; }
        (void)hipEventRecord(a, ctx->stream);
        ctx->prof[slot].start.push_back(a);
        ctx->prof[slot].stop.push_back(b);
        stop = b;
    }
    ~nbx_prof_scope() {
        if (stop) (void)hipEventRecord(stop, ctx->stream);
    }
};

constexpr int NBX_SCRATCH_DOUBLES = 4096;
constexpr int NBX_COUNTERS = 1024;

void nbx_set_error(const char* fmt, ...);

#define NBX_CHECK_ARG(cond)                                                     \
    do {                                                                        \
        if (!(cond)) {                                                          \
            nbx_set_error("%s: invalid argument: %s", __func__, #cond);         \
            return NBX_E_INVALID;                                               \
        }                                                                       \
    } while (0)

#define NBX_HIP(call)                                                           \
    do {                                                                        \
        hipError_t e_ = (call);                                                 \
        if (e_ != hipSuccess) {                                                 \
            nbx_set_error("%s: %s -> %s", __func__, #call, hipGetErrorString(e_)); \
            return NBX_E_HIP;                                                   \
        }                                                                       \
    } while (0)

#define NBX_LAUNCH_CHECK() NBX_HIP(hipGetLastError())

static inline int64_t nbx_cdiv(int64_t a, int64_t b) { return (a + b - 1) / b; }

// ---------------------------------------------------------------- device helpers
__device__ __forceinline__ double nbx_wave_sum(double v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}

// The same sums without the LDS crossbar: a ds_bpermute round trip is ~100 cycles and the butterfly above makes
// six of them in a row, which is most of what a one-wavefront kernel on the SCF's critical path spends.  DPP
// moves take the neighbour's value inside the VALU (quad permutes, then the mirrored half rows of 8 and 16 lanes:
// a sum does not care which lane a term came from), and the four 16-lane rows meet through v_readlane.
// EVERY lane of the wavefront must be active.  The order of the additions differs from nbx_wave_sum's, so the
// two agree to rounding, not bit for bit; every lane returns the same bits.
template <int CTRL>
__device__ __forceinline__ double nbx_dpp_f64(double v) {
    const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), CTRL, 0xf, 0xf, false);
    const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), CTRL, 0xf, 0xf, false);
    return __hiloint2double(hi, lo);
}
constexpr int NBX_DPP_XOR1 = 0xB1, NBX_DPP_XOR2 = 0x4E, NBX_DPP_HALF_MIRROR = 0x141, NBX_DPP_MIRROR = 0x140;

__device__ __forceinline__ double nbx_readlane_f64(double v, int l) {
    return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), l),
                            __builtin_amdgcn_readlane(__double2loint(v), l));
}

// sum over the aligned group of 8 lanes this lane belongs to (valid in all 8)
__device__ __forceinline__ double nbx_sum8_dpp(double v) {
    v += nbx_dpp_f64<NBX_DPP_XOR1>(v);
    v += nbx_dpp_f64<NBX_DPP_XOR2>(v);
    v += nbx_dpp_f64<NBX_DPP_HALF_MIRROR>(v);
    return v;
}

__device__ __forceinline__ double nbx_wave_sum_dpp(double v) {
    v = nbx_sum8_dpp(v);
    v += nbx_dpp_f64<NBX_DPP_MIRROR>(v);
    return (nbx_readlane_f64(v, 0) + nbx_readlane_f64(v, 16)) + (nbx_readlane_f64(v, 32) + nbx_readlane_f64(v, 48));
}

__device__ __forceinline__ double nbx_wave_min_dpp(double v) {
    v = fmin(v, nbx_dpp_f64<NBX_DPP_XOR1>(v));
    v = fmin(v, nbx_dpp_f64<NBX_DPP_XOR2>(v));
    v = fmin(v, nbx_dpp_f64<NBX_DPP_HALF_MIRROR>(v));
    v = fmin(v, nbx_dpp_f64<NBX_DPP_MIRROR>(v));
    return fmin(fmin(nbx_readlane_f64(v, 0), nbx_readlane_f64(v, 16)),
                fmin(nbx_readlane_f64(v, 32), nbx_readlane_f64(v, 48)));
}

// Sum over a workgroup of up to 1024 threads; result valid in every thread.
// `smem` needs 17 doubles.
__device__ __forceinline__ double nbx_block_sum(double v, double* smem) {
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const int nwave = (blockDim.x + 63) >> 6;
    v = nbx_wave_sum(v);
    __syncthreads();
    if (lane == 0) smem[wave] = v;
    __syncthreads();
    if (threadIdx.x == 0) {
        double t = 0.0;
        for (int w = 0; w < nwave; ++w) t += smem[w];
        smem[16] = t;
    }
    __syncthreads();
    return smem[16];
}
