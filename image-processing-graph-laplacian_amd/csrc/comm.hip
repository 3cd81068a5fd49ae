// comm.hip -- native collectives behind glf_comm, and the one-process multi-GPU driver (include/glf.h, "multi-GPU").
//
// The reference is an MPI program: every rank runs main, PETSc's MPIDENSE matrices are row-distributed and VecDot /
// VecNorm / MatMult reduce or gather over PETSC_COMM_WORLD (hpc/image_processing.c:30-76, hpc/gram_schmidt.c:14-15,59,
// hpc/inverse_power_it.c:167). Here the collectives are RCCL calls on the context's own stream, issued by the library:
//   * glf_ctx_set_comm_rccl   one process per GPU (bench.py under torch.distributed.run): ncclCommInitRank from a
//                             unique id the host program distributes;
//   * glf_multi_create        ONE process, one context + one host thread per GPU (host/image_processing.c -ngpu N):
//                             ncclCommInitAll over the listed devices; or, for tests on a single GPU, the LOOPBACK
//                             backend: the same collectives staged through host memory between the rank threads in
//                             fixed rank order (ranks may share a device; RCCL refuses that).
// RCCL is loaded with dlopen at first use -- libglf.so has no link-time dependency on it (a process that already
// carries an RCCL, e.g. under torch, keeps using that one).
#include "glf_internal.hpp"

#include <dlfcn.h>
#include <rccl/rccl.h>

#include <atomic>
#include <condition_variable>
#include <memory>
#include <mutex>
#include <thread>

namespace glf {

// ---- RCCL through dlopen ---------------------------------------------------------------------------------------------
struct RcclApi {
    void *handle = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId *) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t *, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommInitAll)(ncclComm_t *, int, const int *) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*AllReduce)(const void *, void *, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*AllGather)(const void *, void *, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
    const char *(*GetErrorString)(ncclResult_t) = nullptr;
    // optional: what the communicator itself reports, and the way out when a rank fails outside a collective
    ncclResult_t (*CommCount)(const ncclComm_t, int *) = nullptr;
    ncclResult_t (*CommUserRank)(const ncclComm_t, int *) = nullptr;
    ncclResult_t (*CommAbort)(ncclComm_t) = nullptr;
    bool ok = false;
    char load_error[256] = {0}; // dlerror() text of the last failed dlopen (captured once: dlerror() clears itself)
};

static RcclApi g_rccl_api_storage;
static RcclApi *rccl_api()
{
    RcclApi &api = g_rccl_api_storage;
    static std::once_flag once;
    std::call_once(once, [&api] {
        const char *names[] = {"librccl.so", "librccl.so.1", "/opt/rocm/lib/librccl.so.1"};
        for (const char *n : names) // an RCCL the process already carries (torch's) first
            if ((api.handle = dlopen(n, RTLD_NOW | RTLD_NOLOAD))) break;
        for (int i = 0; !api.handle && i < 3; ++i) {
            api.handle = dlopen(names[i], RTLD_NOW | RTLD_GLOBAL);
            if (!api.handle) {
                const char *e = dlerror();
                std::snprintf(api.load_error, sizeof(api.load_error), "%s", e ? e : "not found");
            }
        }
        if (!api.handle) return;
#define GLF_SYM(field, name) api.field = reinterpret_cast<decltype(api.field)>(dlsym(api.handle, name))
        GLF_SYM(GetUniqueId, "ncclGetUniqueId");
        GLF_SYM(CommInitRank, "ncclCommInitRank");
        GLF_SYM(CommInitAll, "ncclCommInitAll");
        GLF_SYM(CommDestroy, "ncclCommDestroy");
        GLF_SYM(AllReduce, "ncclAllReduce");
        GLF_SYM(AllGather, "ncclAllGather");
        GLF_SYM(GetErrorString, "ncclGetErrorString");
        GLF_SYM(CommCount, "ncclCommCount");
        GLF_SYM(CommUserRank, "ncclCommUserRank");
        GLF_SYM(CommAbort, "ncclCommAbort");
#undef GLF_SYM
        api.ok = api.GetUniqueId && api.CommInitRank && api.CommInitAll && api.CommDestroy && api.AllReduce && api.AllGather;
    });
    return api.ok ? &api : nullptr;
}

// user pointer of the native glf_comm callbacks
struct NativeComm {
    glf_ctx *ctx = nullptr;
    ncclComm_t nccl = nullptr;     // RCCL backend
    struct Loopback *loop = nullptr; // loopback backend
    int rank = 0, size = 1;
    // collectives issued since the counters were last read (glf_ctx_comm_counters): calls and payload bytes of this rank
    unsigned long long n_allreduce = 0, n_allgather = 0, bytes_allreduce = 0, bytes_allgather = 0;
};

static int rccl_allreduce(NativeComm *nc, void *buf, size_t count, ncclDataType_t dt)
{
    RcclApi *api = rccl_api();
    if (!api) return 1;
    if (!nc->nccl) return 1; // aborted
    ++nc->n_allreduce;
    nc->bytes_allreduce += count * (dt == ncclFloat64 ? 8 : 4);
    const ncclResult_t r = api->AllReduce(buf, buf, count, dt, ncclSum, nc->nccl, nc->ctx->stream);
    if (r != ncclSuccess) {
        set_error(nc->ctx, GLF_ERR_COMM, "ncclAllReduce -> %s", api->GetErrorString ? api->GetErrorString(r) : "error");
        return 1;
    }
    return 0;
}
static int rccl_allreduce_f32(void *user, float *d, size_t n) { return rccl_allreduce(static_cast<NativeComm *>(user), d, n, ncclFloat32); }
static int rccl_allreduce_f64(void *user, double *d, size_t n) { return rccl_allreduce(static_cast<NativeComm *>(user), d, n, ncclFloat64); }
static int rccl_allgather_f32(void *user, float *d, size_t count_per_rank)
{
    NativeComm *nc = static_cast<NativeComm *>(user);
    RcclApi *api = rccl_api();
    if (!api) return 1;
    if (!nc->nccl) return 1; // aborted
    ++nc->n_allgather;
    nc->bytes_allgather += count_per_rank * sizeof(float) * (size_t)nc->size; // (what lands in this rank's buffer)
    // in place: rank r's block already sits at offset r * count_per_rank of the receive buffer
    const ncclResult_t r = api->AllGather(d + (size_t)nc->rank * count_per_rank, d, count_per_rank, ncclFloat32, nc->nccl, nc->ctx->stream);
    if (r != ncclSuccess) {
        set_error(nc->ctx, GLF_ERR_COMM, "ncclAllGather -> %s", api->GetErrorString ? api->GetErrorString(r) : "error");
        return 1;
    }
    return 0;
}

// ---- loopback: the rank threads of one process meet at a barrier and exchange through host memory -------------------
struct Loopback {
    int size = 1;
    std::mutex mu;
    std::condition_variable cv;
    int waiting = 0;
    unsigned long generation = 0;
    bool broken = false;
    std::vector<std::vector<char>> slot; // one staging buffer per rank
    void barrier()
    {
        std::unique_lock<std::mutex> lk(mu);
        const unsigned long gen = generation;
        if (++waiting == size) {
            waiting = 0;
            ++generation;
            cv.notify_all();
        } else {
            cv.wait(lk, [&] { return generation != gen || broken; });
        }
    }
    void abort_all() // a rank failed outside a collective: release the others (their next collective returns an error)
    {
        std::lock_guard<std::mutex> lk(mu);
        broken = true;
        cv.notify_all();
    }
};

template <typename T>
static int loop_allreduce(NativeComm *nc, T *d, size_t n)
{
    Loopback *L = nc->loop;
    if (L->broken) return 1;
    ++nc->n_allreduce;
    nc->bytes_allreduce += n * sizeof(T);
    std::vector<char> &mine = L->slot[nc->rank];
    mine.resize(n * sizeof(T));
    if (hipMemcpyAsync(mine.data(), d, n * sizeof(T), hipMemcpyDeviceToHost, nc->ctx->stream) != hipSuccess) return 1;
    if (hipStreamSynchronize(nc->ctx->stream) != hipSuccess) return 1;
    L->barrier(); // every slot is filled
    if (L->broken) return 1;
    std::vector<T> sum(n, T(0));
    for (int r = 0; r < L->size; ++r) { // fixed rank order: identical bits on every rank
        if (L->slot[r].size() != n * sizeof(T)) return 1;
        const T *src = reinterpret_cast<const T *>(L->slot[r].data());
        for (size_t i = 0; i < n; ++i) sum[i] += src[i];
    }
    if (hipMemcpyAsync(d, sum.data(), n * sizeof(T), hipMemcpyHostToDevice, nc->ctx->stream) != hipSuccess) return 1;
    if (hipStreamSynchronize(nc->ctx->stream) != hipSuccess) return 1;
    L->barrier(); // every rank has read the slots: they may be overwritten
    return L->broken ? 1 : 0;
}
static int loop_allreduce_f32(void *user, float *d, size_t n) { return loop_allreduce(static_cast<NativeComm *>(user), d, n); }
static int loop_allreduce_f64(void *user, double *d, size_t n) { return loop_allreduce(static_cast<NativeComm *>(user), d, n); }
static int loop_allgather_f32(void *user, float *d, size_t count_per_rank)
{
    NativeComm *nc = static_cast<NativeComm *>(user);
    Loopback *L = nc->loop;
    if (L->broken) return 1;
    const size_t bytes = count_per_rank * sizeof(float);
    ++nc->n_allgather;
    nc->bytes_allgather += bytes * (size_t)L->size;
    std::vector<char> &mine = L->slot[nc->rank];
    mine.resize(bytes);
    if (hipMemcpyAsync(mine.data(), d + (size_t)nc->rank * count_per_rank, bytes, hipMemcpyDeviceToHost, nc->ctx->stream) != hipSuccess) return 1;
    if (hipStreamSynchronize(nc->ctx->stream) != hipSuccess) return 1;
    L->barrier();
    if (L->broken) return 1;
    for (int r = 0; r < L->size; ++r) {
        if (r == nc->rank) continue;
        if (L->slot[r].size() != bytes) return 1;
        if (hipMemcpyAsync(d + (size_t)r * count_per_rank, L->slot[r].data(), bytes, hipMemcpyHostToDevice, nc->ctx->stream) != hipSuccess) return 1;
    }
    if (hipStreamSynchronize(nc->ctx->stream) != hipSuccess) return 1;
    L->barrier();
    return L->broken ? 1 : 0;
}

static void install(glf_ctx *ctx, NativeComm *nc, bool rccl)
{
    glf_comm c{};
    c.rank = nc->rank;
    c.size = nc->size;
    c.allreduce_sum_f32 = rccl ? rccl_allreduce_f32 : loop_allreduce_f32;
    c.allreduce_sum_f64 = rccl ? rccl_allreduce_f64 : loop_allreduce_f64;
    c.allgather_f32 = rccl ? rccl_allgather_f32 : loop_allgather_f32;
    c.user = nc;
    ctx->comm = c;
    ctx->has_comm = nc->size > 1 || ctx->force_comm;
}

} // namespace glf

using namespace glf;

// the native communicator a context owns (released by glf_ctx_set_comm(NULL) / glf_ctx_destroy through native_comm_release)
struct glf_native_comm {
    NativeComm nc;
    bool owns_nccl = false;
};

namespace glf {
void native_comm_release(glf_ctx *ctx)
{
    glf_native_comm *n = ctx->native;
    if (!n) return;
    if (n->owns_nccl && n->nc.nccl) {
        if (RcclApi *api = rccl_api()) (void)api->CommDestroy(n->nc.nccl);
    }
    delete n;
    ctx->native = nullptr;
}
} // namespace glf

struct glf_multi {
    int n = 0, backend = GLF_MULTI_RCCL;
    std::vector<glf_ctx *> ctxs;
    std::unique_ptr<Loopback> loop;
    // per-rank device buffers of glf_multi_image_processing, kept between calls (one image size at a time)
    struct Buffers {
        uint8_t *d_img = nullptr, *d_out = nullptr;
        float *d_zf = nullptr;
        size_t npix = 0;
        bool has_zf = false;
    };
    std::vector<Buffers> buf;
    bool broken = false; // RCCL backend: a rank failed and the communicators were aborted -- the world cannot be used again
    char last_error[512] = {0};
};

extern "C" {

int glf_rccl_unique_id(void *id_out, size_t bytes)
{
    if (!id_out || bytes < sizeof(ncclUniqueId)) return GLF_ERR_INVALID;
    RcclApi *api = rccl_api();
    if (!api) return GLF_ERR_UNSUPPORTED;
    ncclUniqueId id;
    if (api->GetUniqueId(&id) != ncclSuccess) return GLF_ERR_COMM;
    std::memset(id_out, 0, bytes);
    std::memcpy(id_out, &id, sizeof(id));
    return GLF_OK;
}

int glf_ctx_set_comm_rccl(glf_ctx *ctx, int rank, int size, const void *unique_id, size_t bytes, int force)
{
    if (!ctx || size < 1 || rank < 0 || rank >= size || !unique_id || bytes < sizeof(ncclUniqueId)) return GLF_ERR_INVALID;
    GLF_ENTER(ctx);
    RcclApi *api = rccl_api();
    if (!api) return set_error(ctx, GLF_ERR_UNSUPPORTED, "librccl.so could not be loaded: %s", g_rccl_api_storage.load_error);
    native_comm_release(ctx);
    ncclUniqueId id;
    std::memcpy(&id, unique_id, sizeof(id));
    std::unique_ptr<glf_native_comm> n(new glf_native_comm);
    const ncclResult_t r = api->CommInitRank(&n->nc.nccl, size, id, rank);
    if (r != ncclSuccess)
        return set_error(ctx, GLF_ERR_COMM, "ncclCommInitRank(rank %d of %d) -> %s", rank, size, api->GetErrorString ? api->GetErrorString(r) : "error");
    n->owns_nccl = true;
    n->nc.ctx = ctx;
    n->nc.rank = rank;
    n->nc.size = size;
    ctx->force_comm = force != 0;
    ctx->native = n.release();
    install(ctx, &ctx->native->nc, true);
    return GLF_OK;
}

/* What the communicator of a context reports: info = {rank, size, backend (0 none / caller's callbacks, 1 RCCL, 2 loopback),
 * ranks the RCCL communicator itself counts (ncclCommCount; 0 when not RCCL)}. */
int glf_ctx_comm_info(glf_ctx *ctx, int info[4])
{
    if (!ctx || !info) return GLF_ERR_INVALID;
    info[0] = ctx->comm.rank;
    info[1] = ctx->comm.size;
    info[2] = 0;
    info[3] = 0;
    if (glf_native_comm *n = ctx->native) {
        info[2] = n->nc.loop ? 2 : 1;
        if (!n->nc.loop && n->nc.nccl)
            if (RcclApi *api = rccl_api())
                if (api->CommCount) (void)api->CommCount(n->nc.nccl, &info[3]);
    }
    return GLF_OK;
}

/* Collectives this rank issued through the library's own communicator since the last reset:
 * out = {all-reduce calls, all-reduce bytes, all-gather calls, all-gather bytes (received)}. */
int glf_ctx_comm_counters(glf_ctx *ctx, unsigned long long out[4], int reset)
{
    if (!ctx || !out) return GLF_ERR_INVALID;
    out[0] = out[1] = out[2] = out[3] = 0;
    if (glf_native_comm *n = ctx->native) {
        out[0] = n->nc.n_allreduce;
        out[1] = n->nc.bytes_allreduce;
        out[2] = n->nc.n_allgather;
        out[3] = n->nc.bytes_allgather;
        if (reset) n->nc.n_allreduce = n->nc.bytes_allreduce = n->nc.n_allgather = n->nc.bytes_allgather = 0;
    }
    return GLF_OK;
}

int glf_multi_create(glf_multi **out, int n, const int *devices, int backend)
{
    if (!out || n < 1 || n > 64 || (backend != GLF_MULTI_RCCL && backend != GLF_MULTI_LOOPBACK)) return GLF_ERR_INVALID;
    *out = nullptr;
    std::unique_ptr<glf_multi> w(new glf_multi);
    w->n = n;
    w->backend = backend;
    w->buf.resize(n);
    std::vector<int> devs(n);
    for (int r = 0; r < n; ++r) devs[r] = devices ? devices[r] : r;
    int rc = GLF_OK;
    if (backend == GLF_MULTI_RCCL) // RCCL cannot run two ranks on one device: refuse here rather than inside ncclCommInitAll
        for (int r = 0; r < n; ++r)
            for (int q = 0; q < r; ++q)
                if (devs[q] == devs[r]) return GLF_ERR_INVALID;
    for (int r = 0; r < n && rc == GLF_OK; ++r) {
        glf_ctx *c = nullptr;
        rc = glf_ctx_create(&c, devs[r], nullptr);
        if (rc == GLF_OK) w->ctxs.push_back(c);
    }
    std::vector<ncclComm_t> comms(n, nullptr);
    if (rc == GLF_OK && backend == GLF_MULTI_RCCL) {
        RcclApi *api = rccl_api();
        if (!api) rc = GLF_ERR_UNSUPPORTED;
        else if (api->CommInitAll(comms.data(), n, devs.data()) != ncclSuccess) rc = GLF_ERR_COMM; // (refuses a device listed twice)
    }
    if (rc == GLF_OK && backend == GLF_MULTI_LOOPBACK) {
        w->loop.reset(new Loopback);
        w->loop->size = n;
        w->loop->slot.resize(n);
    }
    if (rc != GLF_OK) {
        for (glf_ctx *c : w->ctxs) glf_ctx_destroy(c);
        return rc;
    }
    for (int r = 0; r < n; ++r) {
        glf_native_comm *nat = new glf_native_comm;
        nat->nc.ctx = w->ctxs[r];
        nat->nc.rank = r;
        nat->nc.size = n;
        nat->nc.nccl = comms[r];
        nat->owns_nccl = backend == GLF_MULTI_RCCL;
        nat->nc.loop = w->loop.get();
        w->ctxs[r]->native = nat;
        w->ctxs[r]->force_comm = true; // a one-rank world still runs its collectives (tests the plumbing on one GPU)
        install(w->ctxs[r], &nat->nc, backend == GLF_MULTI_RCCL);
    }
    *out = w.release();
    return GLF_OK;
}

int glf_multi_destroy(glf_multi *w)
{
    if (!w) return GLF_OK;
    for (int r = 0; r < w->n; ++r) {
        glf_ctx *c = w->ctxs[r];
        (void)hipSetDevice(c->device);
        glf_multi::Buffers &b = w->buf[r];
        if (b.d_img) (void)hipFree(b.d_img);
        if (b.d_out) (void)hipFree(b.d_out);
        if (b.d_zf) (void)hipFree(b.d_zf);
        glf_ctx_destroy(c); // releases the native communicator
    }
    delete w;
    return GLF_OK;
}

int glf_multi_size(const glf_multi *w) { return w ? w->n : 0; }
glf_ctx *glf_multi_ctx(glf_multi *w, int rank) { return (w && rank >= 0 && rank < w->n) ? w->ctxs[rank] : nullptr; }
const char *glf_multi_last_error(const glf_multi *w) { return w ? w->last_error : "null glf_multi"; }

// One rank thread per GPU: replicate the image (hpc/image_processing.c:45-76 broadcasts it to every rank), run the sharded
// path, copy this rank's pixel rows of the result back (hpc/utils.c:502-527 gathers to rank 0).
int glf_multi_image_processing(glf_multi *w, const glf_options *opt, const uint8_t *h_img, int width, int height, uint8_t *h_out,
                               float *h_zf, double *eigvals_out, glf_stats *stats)
{
    if (!w || !h_img || !h_out || width <= 0 || height <= 0) return GLF_ERR_INVALID;
    if (w->broken) {
        std::snprintf(w->last_error, sizeof(w->last_error), "the communicators of this world were aborted after a rank failed; create a new one");
        return GLF_ERR_COMM;
    }
    const size_t N = (size_t)width * height;
    std::vector<int> rcs(w->n, GLF_OK);
    std::mutex abort_mu;
    auto rank_main = [&](int r) {
        glf_ctx *ctx = w->ctxs[r];
        glf_multi::Buffers &b = w->buf[r];
        int rc = GLF_OK;
        auto step = [&](hipError_t e) {
            if (e != hipSuccess && rc == GLF_OK) rc = set_error(ctx, GLF_ERR_HIP, "glf_multi rank %d: %s", r, hipGetErrorString(e));
        };
        step(hipSetDevice(ctx->device));
        if (rc == GLF_OK && (b.npix != N || (h_zf && !b.has_zf))) {
            if (b.d_img) (void)hipFree(b.d_img);
            if (b.d_out) (void)hipFree(b.d_out);
            if (b.d_zf) (void)hipFree(b.d_zf);
            b = glf_multi::Buffers{};
            step(hipMalloc(reinterpret_cast<void **>(&b.d_img), N));
            step(hipMalloc(reinterpret_cast<void **>(&b.d_out), N));
            if (h_zf) step(hipMalloc(reinterpret_cast<void **>(&b.d_zf), N * sizeof(float)));
            if (rc == GLF_OK) {
                b.npix = N;
                b.has_zf = h_zf != nullptr;
            }
        }
        if (rc == GLF_OK) {
            step(hipMemcpyAsync(b.d_img, h_img, N, hipMemcpyHostToDevice, ctx->stream));
            step(hipStreamSynchronize(ctx->stream));
        }
        glf_stats st{};
        if (rc == GLF_OK) rc = glf_image_processing(ctx, opt, b.d_img, width, height, b.d_out, h_zf ? b.d_zf : nullptr,
                                                    r == 0 ? eigvals_out : nullptr, &st);
        if (rc == GLF_OK) {
            const size_t o = (size_t)st.row0 * width, len = (size_t)(st.row1 - st.row0) * width;
            if (len) step(hipMemcpyAsync(h_out + o, b.d_out + o, len, hipMemcpyDeviceToHost, ctx->stream));
            if (len && h_zf) step(hipMemcpyAsync(h_zf + o, b.d_zf + o, len * sizeof(float), hipMemcpyDeviceToHost, ctx->stream));
            step(hipStreamSynchronize(ctx->stream));
            if (stats) stats[r] = st;
        }
        if (rc != GLF_OK && w->loop) w->loop->abort_all(); // the other ranks must not wait for this one
        if (rc != GLF_OK && !w->loop) {
            // RCCL: the peers may sit in a collective this rank will never join (it failed outside one: out of memory, a launch
            // error, no convergence on this rank only) and their streams would never drain. Abort every communicator of the
            // world: the pending collectives return an error, the rank threads come back, the world is marked unusable.
            std::lock_guard<std::mutex> lk(abort_mu);
            if (!w->broken) {
                w->broken = true;
                if (RcclApi *api = rccl_api())
                    if (api->CommAbort)
                        for (int q = 0; q < w->n; ++q) {
                            glf_native_comm *nat = w->ctxs[q]->native;
                            if (nat && nat->nc.nccl) {
                                (void)api->CommAbort(nat->nc.nccl);
                                nat->nc.nccl = nullptr;
                                nat->owns_nccl = false;
                            }
                        }
            }
        }
        rcs[r] = rc;
    };
    if (w->loop) {
        std::lock_guard<std::mutex> lk(w->loop->mu);
        w->loop->broken = false;
        w->loop->waiting = 0;
    }
    std::vector<std::thread> threads;
    for (int r = 1; r < w->n; ++r) threads.emplace_back(rank_main, r);
    rank_main(0);
    for (auto &t : threads) t.join();
    for (int r = 0; r < w->n; ++r)
        if (rcs[r] != GLF_OK) {
            std::snprintf(w->last_error, sizeof(w->last_error), "rank %d: %s", r, w->ctxs[r]->last_error);
            return rcs[r];
        }
    return GLF_OK;
}

} // extern "C"
