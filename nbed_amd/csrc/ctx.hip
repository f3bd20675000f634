// libnbx: context, memory and error plumbing (include/nbx.h, "context" section).
#include "nbx_common.h"

static thread_local char g_err[512] = "";

void nbx_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

extern "C" {

int nbx_version(void) { return NBX_VERSION; }

int nbx_experimental(void) {
#ifdef NBX_EXPERIMENTAL
    return 1;
#else
    return 0;
#endif
}

const char* nbx_last_error(void) { return g_err; }

int nbx_device_count(int* count) {
    NBX_CHECK_ARG(count != nullptr);
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess) {
        *count = 0;
        nbx_set_error("nbx_device_count: %s", hipGetErrorString(e));
        (void)hipGetLastError();
        return NBX_E_HIP;
    }
    *count = n;
    return NBX_OK;
}

int nbx_ctx_create(int device, void* stream, int private_stream, nbx_ctx** out) {
    NBX_CHECK_ARG(out != nullptr);
    NBX_CHECK_ARG(device >= 0);
    int n = 0;
    NBX_HIP(hipGetDeviceCount(&n));
    if (device >= n) {
        nbx_set_error("nbx_ctx_create: device %d not present (%d visible)", device, n);
        return NBX_E_INVALID;
    }
    NBX_HIP(hipSetDevice(device));
    {
        // jk_m4.hip masks the block rows a chunk does not hold by pointing their LDS reads BEYOND the LDS of a CU, where
        // gfx9 returns zero (M4_LDS_OOB = 0x30000 bytes past the buffer): a hardware contract, not a language one -- it
        // holds while a CU's LDS ends below that address.  Checked here, once per context: a device that fails it (not
        // gfx950: 160 KB) is refused rather than given a J/K build that might read live memory.
        int lds_per_cu = 0;
        NBX_HIP(hipDeviceGetAttribute(&lds_per_cu, hipDeviceAttributeMaxSharedMemoryPerMultiprocessor, device));
        if (lds_per_cu <= 0 || lds_per_cu > 0x28000) {
            nbx_set_error("nbx_ctx_create: device %d reports %d bytes of LDS per CU; libnbx is built for gfx950 (163840) and "
                          "its packed J/K kernel relies on reads beyond 0x30000 returning zero", device, lds_per_cu);
            return NBX_E_UNSUPPORTED;
        }
    }
    nbx_ctx* c = new nbx_ctx();
    c->device = device;
    c->own_stream = (private_stream != 0);
    if (c->own_stream) {
        hipError_t e = hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking);
        if (e != hipSuccess) {
            delete c;
            nbx_set_error("nbx_ctx_create: hipStreamCreate -> %s", hipGetErrorString(e));
            return NBX_E_HIP;
        }
    } else {
        c->stream = reinterpret_cast<hipStream_t>(stream);
    }
    c->d_scratch = nullptr;
    c->h_pinned = nullptr;
    c->d_counters = nullptr;
    if (hipMalloc(reinterpret_cast<void**>(&c->d_scratch), NBX_SCRATCH_DOUBLES * sizeof(double)) != hipSuccess ||
        hipHostMalloc(reinterpret_cast<void**>(&c->h_pinned), NBX_SCRATCH_DOUBLES * sizeof(double), 0) != hipSuccess ||
        hipMalloc(reinterpret_cast<void**>(&c->d_counters), NBX_COUNTERS * sizeof(int)) != hipSuccess ||
        hipMemsetAsync(c->d_counters, 0, NBX_COUNTERS * sizeof(int), c->stream) != hipSuccess) {
        nbx_set_error("nbx_ctx_create: scratch allocation failed");
        if (c->d_scratch) (void)hipFree(c->d_scratch);
        if (c->h_pinned) (void)hipHostFree(c->h_pinned);
        if (c->d_counters) (void)hipFree(c->d_counters);
        if (c->own_stream) (void)hipStreamDestroy(c->stream);
        delete c;
        return NBX_E_NOMEM;
    }
    *out = c;
    return NBX_OK;
}

int nbx_ctx_destroy(nbx_ctx* ctx) {
    if (ctx == nullptr) return NBX_OK;
    (void)hipSetDevice(ctx->device);
    (void)hipStreamSynchronize(ctx->stream);
    (void)nbx_profile_reset(ctx);
    if (ctx->d_scratch) (void)hipFree(ctx->d_scratch);
    if (ctx->h_pinned) (void)hipHostFree(ctx->h_pinned);
    if (ctx->d_counters) (void)hipFree(ctx->d_counters);
    if (ctx->own_stream) (void)hipStreamDestroy(ctx->stream);
    delete ctx;
    return NBX_OK;
}

int nbx_ctx_set_stream(nbx_ctx* ctx, void* stream) {
    NBX_CHECK_ARG(ctx != nullptr);
    if (ctx->own_stream) {
        (void)hipStreamSynchronize(ctx->stream);
        (void)hipStreamDestroy(ctx->stream);
        ctx->own_stream = false;
    }
    ctx->stream = reinterpret_cast<hipStream_t>(stream);
    return NBX_OK;
}

int nbx_sync(nbx_ctx* ctx) {
    NBX_CHECK_ARG(ctx != nullptr);
    NBX_HIP(hipStreamSynchronize(ctx->stream));
    return NBX_OK;
}

int nbx_malloc(nbx_ctx* ctx, size_t bytes, void** d_ptr) {
    NBX_CHECK_ARG(ctx != nullptr && d_ptr != nullptr);
    NBX_HIP(hipSetDevice(ctx->device));
    hipError_t e = hipMalloc(d_ptr, bytes ? bytes : 8);
    if (e != hipSuccess) {
        nbx_set_error("nbx_malloc(%zu bytes): %s", bytes, hipGetErrorString(e));
        (void)hipGetLastError();
        return NBX_E_NOMEM;
    }
    return NBX_OK;
}

int nbx_free(nbx_ctx* ctx, void* d_ptr) {
    NBX_CHECK_ARG(ctx != nullptr);
    if (d_ptr == nullptr) return NBX_OK;
    NBX_HIP(hipFree(d_ptr));
    return NBX_OK;
}

int nbx_memcpy_h2d(nbx_ctx* ctx, void* d_dst, const void* h_src, size_t bytes) {
    NBX_CHECK_ARG(ctx != nullptr && (bytes == 0 || (d_dst != nullptr && h_src != nullptr)));
    if (bytes == 0) return NBX_OK;
    NBX_HIP(hipMemcpyAsync(d_dst, h_src, bytes, hipMemcpyHostToDevice, ctx->stream));
    // pageable source: make the call safe to return from
    NBX_HIP(hipStreamSynchronize(ctx->stream));
    return NBX_OK;
}

int nbx_memcpy_d2h(nbx_ctx* ctx, void* h_dst, const void* d_src, size_t bytes) {
    NBX_CHECK_ARG(ctx != nullptr && (bytes == 0 || (h_dst != nullptr && d_src != nullptr)));
    if (bytes == 0) return NBX_OK;
    NBX_HIP(hipMemcpyAsync(h_dst, d_src, bytes, hipMemcpyDeviceToHost, ctx->stream));
    NBX_HIP(hipStreamSynchronize(ctx->stream));
    return NBX_OK;
}

namespace {
struct GatherArgs {
    const double* src[8];
    int64_t n[8];
    int count;
};
__global__ __launch_bounds__(256) void gather_kernel(GatherArgs a, double* __restrict__ dst, int* __restrict__ counter) {
    __shared__ int last;
    int64_t off = 0;
    for (int i = 0; i < a.count; ++i) {
        const double* __restrict__ s = a.src[i];
        for (int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; j < a.n[i]; j += (int64_t)gridDim.x * blockDim.x)
            dst[off + j] = s[j];
        off += a.n[i];
    }
    // the word behind the data says it is all there (as huz_scalars_kernel does): the workgroup that arrives last
    // stores it, system scope, after every workgroup's stores have been made visible
    __threadfence_system();
    __syncthreads();
    if (threadIdx.x == 0) {
        const int prev = __hip_atomic_fetch_add(counter, 1, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
        last = prev == (int)gridDim.x - 1;
        if (last) __hip_atomic_store(counter, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    __syncthreads();
    if (last && threadIdx.x == 0) {
        __threadfence_system();
        __hip_atomic_store(dst + off, 1.0, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}
}  // namespace

// Up to 8 device arrays of doubles, one after the other, into `h_dst` -- pinned (device-mapped) host memory the
// kernel stores to directly: one launch for the results of an SCF run (C, eps, D, Hz), where a concatenation kernel, a
// copy and their first-use costs took 0.3 ms of an 8 ms run.  h_dst holds sum(n_doubles) + 1 doubles: the last one
// receives 1.0 after everything before it is visible to the host, which -- with wait = 0 -- can clear it before the
// call, do other work and poll it.  wait != 0: synchronises the stream before returning.
int nbx_gather_to_host(nbx_ctx* ctx, int64_t count, const double* const* d_src, const int64_t* n_doubles, double* h_dst,
                       int wait) {
    NBX_CHECK_ARG(ctx && d_src && n_doubles && h_dst && count >= 1 && count <= 8);
    GatherArgs a{};
    a.count = (int)count;
    int64_t total = 0;
    for (int i = 0; i < count; ++i) {
        NBX_CHECK_ARG(n_doubles[i] >= 0 && (n_doubles[i] == 0 || d_src[i] != nullptr));
        a.src[i] = d_src[i];
        a.n[i] = n_doubles[i];
        total += n_doubles[i];
    }
    const unsigned blocks = (unsigned)((total + 2047) / 2048 < 256 ? (total + 2047) / 2048 : 256);
    hipLaunchKernelGGL(gather_kernel, dim3(blocks ? blocks : 1), dim3(256), 0, ctx->stream, a, h_dst,
                       ctx->d_counters + NBX_COUNTERS - 2);
    NBX_LAUNCH_CHECK();
    if (wait) NBX_HIP(hipStreamSynchronize(ctx->stream));
    return NBX_OK;
}

int nbx_memcpy_d2d(nbx_ctx* ctx, void* d_dst, const void* d_src, size_t bytes) {
    NBX_CHECK_ARG(ctx != nullptr && (bytes == 0 || (d_dst != nullptr && d_src != nullptr)));
    if (bytes == 0) return NBX_OK;
    NBX_HIP(hipMemcpyAsync(d_dst, d_src, bytes, hipMemcpyDeviceToDevice, ctx->stream));
    return NBX_OK;
}

int nbx_memset(nbx_ctx* ctx, void* d_ptr, int value, size_t bytes) {
    NBX_CHECK_ARG(ctx != nullptr && (bytes == 0 || d_ptr != nullptr));
    if (bytes == 0) return NBX_OK;
    NBX_HIP(hipMemsetAsync(d_ptr, value, bytes, ctx->stream));
    return NBX_OK;
}

int nbx_profile_enable(nbx_ctx* ctx, int on) {
    NBX_CHECK_ARG(ctx != nullptr);
    ctx->profiling = (on != 0);
    ctx->prof_mask = (on == 1 || on == 0) ? ~0u : ((unsigned)on >> 2);
    return NBX_OK;
}

// Test support: every CU's LDS filled with `value` (a NaN, say), so that a kernel that reads LDS it has not written shows
// it in its results instead of depending on what ran before it.
__global__ __launch_bounds__(256) void fill_lds_kernel(double value, double* sink) {
    extern __shared__ double lds_all[];
    for (int i = threadIdx.x; i < 160 * 1024 / 8; i += 256) lds_all[i] = value;
    __syncthreads();
    if (lds_all[(threadIdx.x * 37) % (160 * 1024 / 8)] == 1.2345e300) *sink = 0.0;  // (keeps the stores alive)
}

int nbx_debug_fill_lds(nbx_ctx* ctx, double value) {
    NBX_CHECK_ARG(ctx != nullptr);
    static bool attr_set = false;
    if (!attr_set) {
        NBX_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&fill_lds_kernel),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        attr_set = true;
    }
    // one workgroup per CU at a time (it takes the whole LDS); eight rounds over the chip
    hipLaunchKernelGGL(fill_lds_kernel, dim3(2048), dim3(256), 160 * 1024, ctx->stream, value, ctx->d_scratch);
    NBX_LAUNCH_CHECK();
    return NBX_OK;
}

int nbx_profile_sample(nbx_ctx* ctx, int every) {
    NBX_CHECK_ARG(ctx != nullptr && every >= 1);
    ctx->prof_every = every;
    return NBX_OK;
}

static int prof_drain(nbx_ctx* ctx, int slot) {
    nbx_prof_slot& ps = ctx->prof[slot];
    for (size_t i = 0; i < ps.start.size(); ++i) {
        float ms = 0.f;
        NBX_HIP(hipEventSynchronize(ps.stop[i]));
        NBX_HIP(hipEventElapsedTime(&ms, ps.start[i], ps.stop[i]));
        ps.ms_sum += ms;
        ps.count += 1;
        (void)hipEventDestroy(ps.start[i]);
        (void)hipEventDestroy(ps.stop[i]);
    }
    ps.start.clear();
    ps.stop.clear();
    return NBX_OK;
}

int nbx_profile_read(nbx_ctx* ctx, int slot, double* ms_sum, int64_t* count) {
    NBX_CHECK_ARG(ctx != nullptr && slot >= 0 && slot < NBX_PROF_SLOTS && ms_sum && count);
    const int rc = prof_drain(ctx, slot);
    if (rc != NBX_OK) return rc;
    *ms_sum = ctx->prof[slot].ms_sum;
    *count = ctx->prof[slot].count;
    return NBX_OK;
}

int nbx_profile_reset(nbx_ctx* ctx) {
    NBX_CHECK_ARG(ctx != nullptr);
    for (int s = 0; s < NBX_PROF_SLOTS; ++s) {
        const int rc = prof_drain(ctx, s);
        if (rc != NBX_OK) return rc;
        ctx->prof[s].ms_sum = 0.0;
        ctx->prof[s].count = 0;
        ctx->prof[s].seen = 0;
    }
    return NBX_OK;
}

}  // extern "C"
