// ctx.hip -- context, flat device buffers and matrix descriptors of the C-ABI (include/glf.h).
#include "glf_internal.hpp"

#include <cmath>
#include <cstdlib>

extern "C" {

const char *glf_strerror(int status)
{
    switch (status) {
    case GLF_OK: return "ok";
    case GLF_ERR_INVALID: return "invalid argument";
    case GLF_ERR_NOMEM: return "out of memory";
    case GLF_ERR_HIP: return "HIP runtime error";
    case GLF_ERR_NODEVICE: return "no usable gfx950 device";
    case GLF_ERR_COMM: return "collective callback failed";
    case GLF_ERR_NOCONV: return "eigensolver did not converge";
    case GLF_ERR_IO: return "image I/O error";
    case GLF_ERR_UNSUPPORTED: return "unsupported";
    default: return "unknown status";
    }
}

int glf_ctx_create(glf_ctx **out, int device, void *hip_stream)
{
    if (!out) return GLF_ERR_INVALID;
    *out = nullptr;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return GLF_ERR_NODEVICE;
    if (device < 0 || device >= ndev) return GLF_ERR_INVALID;
    if (hipSetDevice(device) != hipSuccess) return GLF_ERR_NODEVICE;
    glf_ctx *ctx = new (std::nothrow) glf_ctx();
    if (!ctx) return GLF_ERR_NOMEM;
    ctx->device = device;
    if (hipGetDeviceProperties(&ctx->prop, device) != hipSuccess) {
        delete ctx;
        return GLF_ERR_NODEVICE;
    }
    // The kernels are compiled for gfx950 only (MFMA f32 32x32x2, wave64): fail
    // loudly on anything else instead of letting a launch die later.
    if (std::strncmp(ctx->prop.gcnArchName, "gfx950", 6) != 0) {
        fprintf(stderr, "glf: device %d is %s, this library is built for gfx950 only\n", device,
                ctx->prop.gcnArchName);
        delete ctx;
        return GLF_ERR_NODEVICE;
    }
    if (hip_stream) {
        ctx->stream = reinterpret_cast<hipStream_t>(hip_stream);
        ctx->owns_stream = false;
    } else {
        if (hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking) != hipSuccess) {
            delete ctx;
            return GLF_ERR_HIP;
        }
        ctx->owns_stream = true;
    }
    for (auto &e : ctx->ev)
        if (hipEventCreate(&e) != hipSuccess) {
            delete ctx;
            return GLF_ERR_HIP;
        }
    for (auto &row : ctx->mv_ev)
        for (auto &e : row)
            if (hipEventCreate(&e) != hipSuccess) {
                delete ctx;
                return GLF_ERR_HIP;
            }
    ctx->comm.rank = 0;
    ctx->comm.size = 1;
    if (const char *mode = std::getenv("GLF_CONTRACTION")) {
        if (std::strcmp(mode, "f32") == 0) ctx->contraction = GLF_CONTRACT_F32_MFMA;
        else if (std::strcmp(mode, "f16s") == 0) ctx->contraction = GLF_CONTRACT_F16_SPLIT;
        else fprintf(stderr, "glf: ignoring GLF_CONTRACTION=%s (expected f32 or f16s)\n", mode);
    }
    if (const char *dbg = std::getenv("GLF_POOL_DEBUG")) ctx->pool_debug = dbg[0] && dbg[0] != '0';
    for (const char *key : {"NYS_PATH", "DEG_PATH", "MV_PATH", "ROWPASS", "ROWPASS_OP", "SWEEP_COLPASS", "COLPASS", "NYS_NO_LUT", "NO_ECR", "NO_NARROW", "NO_FUSED_FILTER", "EIG_SHARD", "ZMFMA_GROUPS", "GS", "RESIDUAL", "VERBOSE"}) {
        char name[32];
        std::snprintf(name, sizeof(name), "GLF_%s", key);
        if (const char *v = std::getenv(name))
            if (glf_ctx_set_tuning(ctx, key, v) != GLF_OK) fprintf(stderr, "glf: ignoring %s=%s\n", name, v);
    }
    *out = ctx;
    return GLF_OK;
}

int glf_ctx_set_tuning(glf_ctx *ctx, const char *key, const char *value)
{
    if (!ctx || !key) return GLF_ERR_INVALID;
    const bool unset = !value || !value[0] || !std::strcmp(value, "auto") || !std::strcmp(value, "default");
    auto is = [&](const char *s) { return value && !std::strcmp(value, s); };
    auto flag = [&]() { return !unset && !is("0"); };
    glf_tuning &t = ctx->tune;
    if (!std::strcmp(key, "NYS_PATH") || !std::strcmp(key, "DEG_PATH")) {
        const bool nys = key[0] == 'N';
        if (!unset && !is("grid") && !is("direct") && !(nys && (is("rank") || is("band")))) return GLF_ERR_INVALID;
        (nys ? t.nys_path : t.deg_path) = unset ? 0 : is("grid") ? 1 : is("rank") ? 3 : is("band") ? 4 : 2;
    } else if (!std::strcmp(key, "MV_PATH")) {
        if (!unset && !is("grid") && !is("dense") && !is("rank") && !is("band")) return GLF_ERR_INVALID;
        t.mv_path = unset ? 0 : is("grid") ? 1 : is("rank") ? 3 : is("band") ? 4 : 2;
    } else if (!std::strcmp(key, "ROWPASS")) {
        if (!unset && !is("rt") && !is("v1")) return GLF_ERR_INVALID;
        t.rowpass = is("v1") ? 1 : 0;
    } else if (!std::strcmp(key, "ROWPASS_OP")) {
        if (!unset && !is("rt") && !is("v1")) return GLF_ERR_INVALID;
        t.rowpass_op = is("rt") ? 1 : 0;
    } else if (!std::strcmp(key, "SWEEP_COLPASS")) { // rank-form L_A sweeps: the value-grouped column pass on the samples | k_rank_samples
        if (!unset && !is("segments") && !is("samples")) return GLF_ERR_INVALID;
        t.sweep_samples = is("samples");
    } else if (!std::strcmp(key, "COLPASS")) {
        if (!unset && !is("ws") && !is("v1")) return GLF_ERR_INVALID;
        t.colpass = is("v1") ? 1 : 0;
    } else if (!std::strcmp(key, "NYS_NO_LUT")) t.nys_no_lut = flag();
    else if (!std::strcmp(key, "NO_ECR")) t.no_ecr = flag();
    else if (!std::strcmp(key, "NO_NARROW")) t.no_narrow = flag();
    else if (!std::strcmp(key, "NO_FUSED_FILTER")) t.no_fused_filter = flag();
    else if (!std::strcmp(key, "ZMFMA_GROUPS")) t.zmfma_groups = flag();
    else if (!std::strcmp(key, "EIG_SHARD")) t.eig_shard = unset ? 0 : is("0") ? 2 : 1;
    else if (!std::strcmp(key, "GS")) {
        if (!unset && !is("seq") && !is("gram")) return GLF_ERR_INVALID;
        t.gs_seq = is("seq");
    } else if (!std::strcmp(key, "RESIDUAL")) {
        if (!unset && !is("sweep") && !is("derived")) return GLF_ERR_INVALID;
        t.residual_sweep = is("sweep");
    } else if (!std::strcmp(key, "VERBOSE")) t.verbose = flag();
    else return GLF_ERR_INVALID;
    return GLF_OK;
}

size_t glf_ctx_cached_bytes(const glf_ctx *ctx)
{
    size_t total = 0;
    if (ctx)
        for (const glf_pool_block &b : ctx->pool)
            if (!b.in_use) total += b.bytes;
    return total;
}

int glf_ctx_debug_violations(const glf_ctx *ctx)
{
    if (!ctx || !ctx->pool_debug) return -1;
    return ctx->pool_violations;
}

int glf_ctx_destroy(glf_ctx *ctx)
{
    if (!ctx) return GLF_OK;
    (void)hipSetDevice(ctx->device);
    (void)hipStreamSynchronize(ctx->stream);
    for (auto &e : ctx->ev)
        if (e) (void)hipEventDestroy(e);
    for (auto &row : ctx->mv_ev)
        for (auto &e : row)
            if (e) (void)hipEventDestroy(e);
    for (auto &row : ctx->cp_ev)
        for (auto &e : row)
            if (e) (void)hipEventDestroy(e);
    glf::native_comm_release(ctx);
    if (ctx->mv_scratch) (void)hipFree(ctx->mv_scratch);
    if (ctx->x0_block) (void)hipFree(ctx->x0_block);
    glf::band_cache_free(ctx);
    if (ctx->rank_ftab) (void)hipFree(ctx->rank_ftab);
    if (ctx->rank_ff) (void)hipFree(ctx->rank_ff);
    if (ctx->pinned) (void)hipHostFree(ctx->pinned);
    glf::pool_free_all(ctx, false);
    if (ctx->owns_stream && ctx->stream) (void)hipStreamDestroy(ctx->stream);
    delete ctx;
    return GLF_OK;
}

int glf_ctx_synchronize(glf_ctx *ctx)
{
    if (!ctx) return GLF_ERR_INVALID;
    GLF_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return GLF_OK;
}

const char *glf_ctx_last_error(const glf_ctx *ctx) { return ctx ? ctx->last_error : "null context"; }

int glf_ctx_device_info(const glf_ctx *ctx, char *name, size_t name_len, int *num_cus, size_t *total_mem)
{
    if (!ctx) return GLF_ERR_INVALID;
    if (name && name_len) snprintf(name, name_len, "%s (%s)", ctx->prop.name, ctx->prop.gcnArchName);
    if (num_cus) *num_cus = ctx->prop.multiProcessorCount;
    if (total_mem) *total_mem = ctx->prop.totalGlobalMem;
    return GLF_OK;
}

int glf_ctx_set_comm(glf_ctx *ctx, const glf_comm *comm)
{
    if (!ctx) return GLF_ERR_INVALID;
    glf::native_comm_release(ctx); // a communicator the library created itself (glf_ctx_set_comm_rccl) is replaced
    ctx->force_comm = false;
    if (!comm || comm->size <= 1) {
        ctx->comm = glf_comm{};
        ctx->comm.size = 1;
        ctx->has_comm = false;
        return GLF_OK;
    }
    if (comm->rank < 0 || comm->rank >= comm->size || !comm->allreduce_sum_f64 || !comm->allreduce_sum_f32)
        return glf::set_error(ctx, GLF_ERR_INVALID, "glf_comm needs rank < size and both allreduce callbacks");
    ctx->comm = *comm;
    ctx->has_comm = true;
    return GLF_OK;
}

int glf_ctx_set_contraction(glf_ctx *ctx, int mode)
{
    if (!ctx || (mode != GLF_CONTRACT_F32_MFMA && mode != GLF_CONTRACT_F16_SPLIT)) return GLF_ERR_INVALID;
    ctx->contraction = mode;
    return GLF_OK;
}

int glf_malloc(glf_ctx *ctx, void **dptr, size_t bytes)
{
    if (!ctx || !dptr) return GLF_ERR_INVALID;
    *dptr = nullptr;
    if (bytes == 0) return GLF_OK;
    GLF_ENTER(ctx);
    GLF_HIP(ctx, hipMalloc(dptr, bytes));
    return GLF_OK;
}

int glf_free(glf_ctx *ctx, void *dptr)
{
    if (!ctx) return GLF_ERR_INVALID;
    GLF_ENTER(ctx);
    if (dptr) GLF_HIP(ctx, hipFree(dptr));
    return GLF_OK;
}

int glf_memcpy_h2d(glf_ctx *ctx, void *dst, const void *src, size_t bytes)
{
    if (!ctx || (bytes && (!dst || !src))) return GLF_ERR_INVALID;
    GLF_ENTER(ctx);
    GLF_HIP(ctx, hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, ctx->stream));
    GLF_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return GLF_OK;
}

int glf_memcpy_d2h(glf_ctx *ctx, void *dst, const void *src, size_t bytes)
{
    if (!ctx || (bytes && (!dst || !src))) return GLF_ERR_INVALID;
    GLF_ENTER(ctx);
    GLF_HIP(ctx, hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, ctx->stream));
    GLF_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return GLF_OK;
}

int glf_memset(glf_ctx *ctx, void *dst, int value, size_t bytes)
{
    if (!ctx || (bytes && !dst)) return GLF_ERR_INVALID;
    GLF_ENTER(ctx);
    GLF_HIP(ctx, hipMemsetAsync(dst, value, bytes, ctx->stream));
    return GLF_OK;
}

int glf_mat_create_dense(glf_ctx *ctx, glf_mat *mat, int64_t rows, int64_t cols, int64_t ld)
{
    if (!ctx || !mat || rows < 0 || cols < 0) return GLF_ERR_INVALID;
    if (ld <= 0) ld = cols;
    if (ld < cols) return GLF_ERR_INVALID;
    std::memset(mat, 0, sizeof(*mat));
    mat->kind = GLF_MAT_DENSE;
    mat->rows = rows;
    mat->cols = cols;
    mat->ld = ld;
    void *d = nullptr;
    GLF_TRY(glf_malloc(ctx, &d, sizeof(float) * (size_t)rows * (size_t)ld));
    mat->data = static_cast<float *>(d);
    mat->owns_data = 1;
    if (d) GLF_HIP(ctx, hipMemsetAsync(d, 0, sizeof(float) * (size_t)rows * (size_t)ld, ctx->stream));
    return GLF_OK;
}

int glf_mat_create_diag(glf_ctx *ctx, glf_mat *mat, int64_t n)
{
    if (!ctx || !mat || n < 0) return GLF_ERR_INVALID;
    std::memset(mat, 0, sizeof(*mat));
    mat->kind = GLF_MAT_DIAG;
    mat->rows = mat->cols = n;
    mat->ld = 1;
    void *d = nullptr;
    GLF_TRY(glf_malloc(ctx, &d, sizeof(float) * (size_t)n));
    mat->data = static_cast<float *>(d);
    mat->owns_data = 1;
    return GLF_OK;
}

int glf_mat_destroy(glf_ctx *ctx, glf_mat *mat)
{
    if (!ctx || !mat) return GLF_ERR_INVALID;
    GLF_ENTER(ctx);
    if (mat->owns_data && mat->data) (void)hipFree(mat->data);
    if (mat->owns_desc) {
        if (mat->samples) (void)hipFree(const_cast<float *>(mat->samples));
        if (mat->mask) (void)hipFree(const_cast<uint8_t *>(mat->mask));
        if (mat->idx) (void)hipFree(const_cast<uint32_t *>(mat->idx));
        if (mat->degree) (void)hipFree(mat->degree);
    }
    std::memset(mat, 0, sizeof(*mat));
    return GLF_OK;
}

int glf_mat_get_column(glf_ctx *ctx, const glf_mat *mat, int64_t col, float *host_out)
{
    if (!ctx || !mat || !host_out || mat->kind != GLF_MAT_DENSE || col < 0 || col >= mat->cols) return GLF_ERR_INVALID;
    GLF_ENTER(ctx);
    GLF_HIP(ctx, hipMemcpy2DAsync(host_out, sizeof(float), mat->data + col, sizeof(float) * (size_t)mat->ld, sizeof(float),
                                  (size_t)mat->rows, hipMemcpyDeviceToHost, ctx->stream));
    GLF_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return GLF_OK;
}

void glf_options_default(glf_options *opt)
{
    if (!opt) return;
    std::memset(opt, 0, sizeof(*opt));
    opt->struct_size = sizeof(glf_options);
    opt->num_samples = 0;
    opt->sample_frac = 0.01;   // hpc/image_processing.c:187
    opt->num_eigvals = 0;      // -> p - 1, :96-108
    opt->opti_gs = 1;          // :128-140
    opt->epsilon = 0.1;        // :151
    opt->inner_rtol = 1e-5;    // PETSc KSP default rtol
    opt->max_outer = 100000;
    opt->seed = 1;
    opt->gain = 3.0f;          // hpc/display.c:73
    opt->h_loc = 40.0f;        // hpc/affinity.c:118
    opt->h_val = 30.0f;        // hpc/affinity.c:117
    opt->kernel = GLF_KERNEL_BILATERAL;
    opt->filter_pow = 1;       // MatPow no-op, hpc/utils.c:721
    opt->filter_mode = GLF_FILTER_REFERENCE;
    opt->filter_beta = 1.5f;
    opt->skip_exact_zeros = 0; // evaluate every entry, as the reference does
    opt->sampling = GLF_SAMPLING_UNIFORM; // hpc/sampling.c:6-23 (the PoC's default too: python/image_processing.py:249)
    opt->sampling_seed = 1;
}

void glf_host_free(void *ptr) { std::free(ptr); }

} // extern "C"

namespace glf {

// debug pool: the guard zone of a block (POOL_GUARD_BYTES of POOL_CANARY after its usable bytes) must be intact
static void pool_check_guard(glf_ctx *ctx, const glf_pool_block &b, const char *when)
{
    unsigned char h[POOL_GUARD_BYTES];
    (void)hipStreamSynchronize(ctx->stream);
    if (hipMemcpy(h, static_cast<char *>(b.p) + b.bytes, POOL_GUARD_BYTES, hipMemcpyDeviceToHost) != hipSuccess) {
        (void)hipGetLastError();
        return;
    }
    size_t bad = 0, first = 0;
    for (size_t i = 0; i < POOL_GUARD_BYTES; ++i)
        if (h[i] != (unsigned char)POOL_CANARY) {
            if (!bad) first = i;
            ++bad;
        }
    if (bad) {
        ++ctx->pool_violations;
        fprintf(stderr, "glf debug pool: %zu guard bytes overwritten after a %zu-byte work buffer (first at +%zu) %s\n", bad, b.bytes,
                b.bytes + first, when);
        set_error(ctx, GLF_ERR_HIP, "debug pool: write past the end of a %zu-byte work buffer", b.bytes);
    }
}

void pool_age(glf_ctx *ctx);

void *pool_get(glf_ctx *ctx, size_t bytes, bool poison_nan)
{
    (void)hipSetDevice(ctx->device);
    const size_t need = (size_t)round_up((int64_t)bytes, 256);
    if (ctx->pool_debug) { // exact size + guard zone, never reused, no stale contents
        void *p = nullptr;
        if (hipMalloc(&p, need + POOL_GUARD_BYTES) != hipSuccess) {
            (void)hipGetLastError();
            set_error(ctx, GLF_ERR_NOMEM, "hipMalloc(%zu bytes, debug pool)", need + POOL_GUARD_BYTES);
            return nullptr;
        }
        (void)hipMemsetAsync(p, poison_nan ? 0xFF : 0x00, need, ctx->stream); // 0xFF...: NaN as f16, f32 and f64
        (void)hipMemsetAsync(static_cast<char *>(p) + need, POOL_CANARY, POOL_GUARD_BYTES, ctx->stream);
        ctx->pool.push_back(glf_pool_block{p, need, true, ctx->call_no});
        return p;
    }
    int best = -1;
    for (int i = 0; i < (int)ctx->pool.size(); ++i) {
        const glf_pool_block &b = ctx->pool[i];
        if (b.in_use || b.bytes < need || b.bytes > need + need / 2 + (1u << 20)) continue;
        if (best < 0 || b.bytes < ctx->pool[best].bytes) best = i;
    }
    if (best >= 0) {
        ctx->pool[best].in_use = true;
        ctx->pool[best].last_call = ctx->call_no;
        return ctx->pool[best].p;
    }
    pool_age(ctx); // a miss: before growing, give back what no call has taken for a while
    void *p = nullptr;
    hipError_t e = hipMalloc(&p, need);
    if (e != hipSuccess) { // make room: drop every cached block nobody is using, then retry once
        (void)hipGetLastError();
        (void)hipStreamSynchronize(ctx->stream);
        pool_free_all(ctx, true);
        e = hipMalloc(&p, need);
    }
    if (e != hipSuccess) {
        (void)hipGetLastError();
        set_error(ctx, GLF_ERR_NOMEM, "hipMalloc(%zu bytes) -> %s", need, hipGetErrorString(e));
        return nullptr;
    }
    ctx->pool.push_back(glf_pool_block{p, need, true, ctx->call_no});
    return p;
}

// Returns the cached blocks no call has taken for POOL_KEEP_CALLS public calls to the driver: a process that walks through many
// image sizes would otherwise keep every size's work buffers for the life of the context. Called on every pool miss and at
// the start of the whole-path entry points.
void pool_age(glf_ctx *ctx)
{
    if (ctx->pool_debug) return; // (the debug pool caches nothing)
    for (size_t i = 0; i < ctx->pool.size();) {
        const glf_pool_block &b = ctx->pool[i];
        if (!b.in_use && ctx->call_no - b.last_call > POOL_KEEP_CALLS) {
            (void)hipFree(b.p); // (waits for work that may still read it)
            ctx->pool.erase(ctx->pool.begin() + (long)i);
        } else {
            ++i;
        }
    }
}

void pool_put(glf_ctx *ctx, void *ptr)
{
    if (!ctx || !ptr) return;
    for (size_t i = 0; i < ctx->pool.size(); ++i) {
        glf_pool_block &b = ctx->pool[i];
        if (b.p != ptr) continue;
        if (ctx->pool_debug) { // verify the guard, then give the block back to the driver (no reuse)
            pool_check_guard(ctx, b, "(at release)");
            (void)hipFree(b.p);
            ctx->pool.erase(ctx->pool.begin() + (long)i);
        } else {
            b.in_use = false;
        }
        return;
    }
    (void)hipFree(ptr); // not ours: plain allocation
}

void pool_forget(glf_ctx *ctx, void *ptr)
{
    for (size_t i = 0; i < ctx->pool.size(); ++i)
        if (ctx->pool[i].p == ptr) {
            if (ctx->pool_debug) pool_check_guard(ctx, ctx->pool[i], "(ownership handed to the caller)");
            ctx->pool.erase(ctx->pool.begin() + (long)i);
            return;
        }
}

void pool_free_all(glf_ctx *ctx, bool only_unused)
{
    std::vector<glf_pool_block> keep;
    for (auto &b : ctx->pool) {
        if (only_unused && b.in_use) keep.push_back(b);
        else {
            if (ctx->pool_debug) pool_check_guard(ctx, b, "(at context teardown)");
            (void)hipFree(b.p);
        }
    }
    ctx->pool.swap(keep);
}

KernelCoef make_coef(int kernel, float h_loc, float h_val)
{
    const double log2e = 1.4426950408889634;
    KernelCoef c;
    c.s_loc = (kernel == GLF_KERNEL_PHOTOMETRIC || kernel == GLF_KERNEL_NLM) ? 0.0f : (float)(log2e / ((double)h_loc * (double)h_loc));
    c.s_val = (kernel == GLF_KERNEL_SPATIAL) ? 0.0f : (float)(log2e / ((double)h_val * (double)h_val));
    c.kernel = kernel;
    return c;
}

} // namespace glf
