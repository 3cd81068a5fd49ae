// glf_internal.hpp -- shared declarations of the HIP implementation behind include/glf.h.
// gfx950 (MI355X / CDNA4) only; wave = 64 lanes.
#pragma once

#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <type_traits>
#include <vector>

#include "../../include/glf.h"

// Workspace pool of a context: device blocks released by DevBuf are kept and handed out again
// (same stream => ordering is safe). hipMalloc / hipFree of the multi-GB buffers on every call cost
// sporadic 1.7 s stalls (driver unmap of a 29 GB block) and a device-wide sync per hipFree.
struct glf_pool_block {
    void *p;
    size_t bytes;     // usable bytes (debug pool: the exact request rounded up to 256; a guard zone follows)
    bool in_use;
    unsigned last_call = 0; // glf_ctx::call_no of the public call that last took the block
};

// Which of several equivalent kernels implements a stage. Defaults (0 / false) = chosen from the problem size; every choice
// computes the same sums and is tested against the oracle and against the others. Set with glf_ctx_set_tuning, initialised at
// context creation from the environment variables GLF_<KEY> (read once, never per call).
struct glf_tuning {
    int nys_path = 0;            // NYS_PATH: 0 auto, 1 grid (factored Nystroem, all 256 grey levels), 2 direct (entry by entry), 3 rank (factored, photometric table as a rank-R expansion), 4 band (entry by entry, only the samples within the kernel's radius)
    int deg_path = 0;            // DEG_PATH: likewise for the degree
    int mv_path = 0;             // MV_PATH: 0 auto, 1 grid (L_A applied in factored form), 2 dense (stored L_A), 3 rank (factored, rank-R photometric table), 4 band (entry by entry within the radius, L_A never stored)
    int rowpass = 0;             // ROWPASS: row pass of the Nystroem passes: 0 / rt = row-tile form, 1 / v1 = one image row per wave
    int rowpass_op = 0;          // ROWPASS_OP: row pass of the L_A sweeps: 0 / v1, 1 / rt
    bool sweep_samples = false;  // SWEEP_COLPASS: rank-form L_A sweeps: false / segments = k_rank_colpass on the samples, true / samples = k_rank_samples
    int colpass = 0;             // COLPASS: rank-form column pass: 0 / ws = waves split into forming and contracting roles, 1 / v1 = every wave does both
    bool nys_no_lut = false;     // NYS_NO_LUT: direct Nystroem kernel generates entries with v_exp_f32 instead of LDS tables
    bool no_ecr = false;         // NO_ECR: column pass reads the per-column Ec fragment table instead of the compact one
    bool gs_seq = false;         // GS=seq: Gram-Schmidt as the column-by-column sweep instead of the Gram-matrix form
    bool residual_sweep = false; // RESIDUAL=sweep: residual from an explicit L_A sweep instead of the PCG state
    bool zmfma_groups = false;   // ZMFMA_GROUPS: degree row contraction in groups of five m-tiles (two n-tiles per wave) even when ten fit one wave
    int eig_shard = 0;           // EIG_SHARD: 0 auto (row-sharded eigen-solve unless the operator is in band form), 1 sharded, 2 replicated
    bool no_fused_filter = false; // NO_FUSED_FILTER: band form writes Phi and the filter runs as its own stage (k_apply_filter)
    bool no_narrow = false;      // NO_NARROW: block PCG applies the operator to all columns of the block even when few still iterate
    bool verbose = false;        // VERBOSE: log every outer iteration on stderr (the reference does, hpc/inverse_power_it.c:164-181)
};

struct glf_native_comm; // comm.hip: RCCL / loopback communicator owned by a context

struct glf_ctx {
    std::vector<glf_pool_block> pool;
    int device = 0;
    hipStream_t stream = nullptr;
    bool owns_stream = false;
    glf_comm comm{};
    bool has_comm = false;
    glf_native_comm *native = nullptr; // set by glf_ctx_set_comm_rccl / glf_multi_create: the library's own collectives
    bool force_comm = false;           // run the collectives even on a one-rank world (tests of the N > 1 plumbing)
    char last_error[512] = {0};
    hipDeviceProp_t prop{};
    // reusable events for stage timing
    hipEvent_t ev[8] = {};
    void *mv_scratch = nullptr; // split-f16 X fragments of the block mat-vec
    size_t mv_scratch_bytes = 0;
    // timing of the L_A sweeps (the HBM-bound kernel): a ring of event pairs, summed by mv_collect()
    static constexpr int MV_RING = 64;
    hipEvent_t mv_ev[2][MV_RING] = {};
    static constexpr int CP_RING = 16; // event pairs around the rank-form column-pass launches of one call (created on first use)
    hipEvent_t cp_ev[2][CP_RING] = {};
    int mv_pending = 0;
    unsigned call_no = 0; // public entry points so far (ages the cached work buffers: pool_get)
    int mv_count = 0;
    int narrow_sweeps = 0; // L_A sweeps applied to a packed block of the still-active columns (block PCG)
    double mv_ms = 0.0, mv_bytes = 0.0;
    // the seeded random start block of the eigensolver (hpc/inverse_power_it.c:12-47) as a device [p64][ld] f32 block:
    // it depends on (p, m, ld, seed) only, so images of one size reuse it (17 ms of host time at cfg4)
    float *x0_block = nullptr;
    unsigned x0_p = 0, x0_m = 0, x0_ld = 0;
    unsigned long long x0_seed = 0;
    int contraction = GLF_CONTRACT_F16_SPLIT; // how glf_Nystroem / glf_image_processing contract K_B^T Psi
    // one page of pinned, device-visible host memory the eigensolver's kernels write their flags and norms into (read
    // after a stream synchronise; no D2H copy launches, and no hipHostMalloc per image): see glf::ctx_pinned()
    void *pinned = nullptr;
    // rank form of the grid-factored contractions (nystroem_rank.inc): the factor F of the photometric table for one scale
    // band form (nystroem_band.inc): the geometry tables of the last (sample grid, kernel, image size), pixels / samples as targets
    // -- they depend on neither the image nor the operand, so consecutive images of one size reuse them (a plan)
    void *band_cache[2] = {nullptr, nullptr};
    bool rank_valid = false;
    float rank_s_val = 0.f;
    int rank_R = 0;             // terms of the expansion (0: the table is not low-rank enough, the exact form runs)
    void *rank_ftab = nullptr;  // f32 [R][256]
    void *rank_ff = nullptr;    // split-f16 A fragments of 2^15 F
    // debug pool (GLF_POOL_DEBUG=1): exact-size blocks + guard zone, NaN-filled floating-point buffers, no reuse
    bool pool_debug = false;
    int pool_violations = 0;
    glf_tuning tune;
};

namespace glf {

constexpr int WAVE = 64;
constexpr int VEC_PAD = 64; // rows of vector blocks / lda of L_A are padded to a multiple of this (zeros)
constexpr int NYS_PAD = 64; // sample table and Psi are zero-padded to a multiple of this many rows

// pinned page layout (bytes): [0] PCG active-column counter (int), [64] Gram-Schmidt fallback flag (int),
// [128 .. 128 + 8 * 256) residual column sums (double[ld <= 256]), [3072] largest segment count of a row (rank form)
constexpr size_t PINNED_BYTES = 4096, PINNED_NACTIVE = 0, PINNED_GSFLAG = 64, PINNED_SUMS = 128, PINNED_RANKSEG = 3072, PINNED_BANDEVAL = 3136;
inline char *ctx_pinned(glf_ctx *ctx)
{
    if (!ctx->pinned && hipHostMalloc(&ctx->pinned, PINNED_BYTES, hipHostMallocDefault) != hipSuccess) ctx->pinned = nullptr;
    return static_cast<char *>(ctx->pinned);
}

inline int set_error(glf_ctx *ctx, int status, const char *fmt, ...)
{
    if (ctx) {
        va_list ap;
        va_start(ap, fmt);
        vsnprintf(ctx->last_error, sizeof(ctx->last_error), fmt, ap);
        va_end(ap);
    }
    return status;
}

#define GLF_HIP(ctx, call)                                                                    \
    do {                                                                                      \
        hipError_t e__ = (call);                                                              \
        if (e__ != hipSuccess)                                                                \
            return glf::set_error((ctx), e__ == hipErrorOutOfMemory ? GLF_ERR_NOMEM : GLF_ERR_HIP, \
                                  "%s:%d %s -> %s", __FILE__, __LINE__, #call, hipGetErrorString(e__)); \
    } while (0)

#define GLF_TRY(expr)                 \
    do {                              \
        int s__ = (expr);             \
        if (s__ != GLF_OK) return s__; \
    } while (0)

#define GLF_LAUNCH_CHECK(ctx) GLF_HIP(ctx, hipGetLastError())
// every public entry point: the calling thread's current device becomes the context's (two contexts on different GPUs in
// one process, or one host thread per GPU in glf_multi_*, would otherwise allocate and launch on the wrong device)
#define GLF_ENTER(ctx)                                 \
    do {                                               \
        GLF_HIP(ctx, hipSetDevice((ctx)->device));     \
        ++(ctx)->call_no;                              \
    } while (0)

inline int64_t round_up(int64_t x, int64_t q) { return (x + q - 1) / q * q; }
inline int64_t ceil_div(int64_t x, int64_t q) { return (x + q - 1) / q; }
// Leading dimension of vector blocks: m rounded up to a power of two in {32,64,128,256}
// (the column-reduction kernels map 256 threads onto 256/ld row lanes x ld columns).
inline unsigned ld_for(unsigned m) { unsigned ld = 32; while (ld < m) ld <<= 1; return ld; }
inline bool valid_ld(unsigned ld) { return ld == 32 || ld == 64 || ld == 128 || ld == 256; }
// More than 256 vectors: row-major matrices of the C-ABI carry ld = m rounded up to a multiple of 256; internally the
// vectors are processed as panels of 256 columns, each a contiguous [rows][256] block (eigen.hip, "panels").
constexpr unsigned PANEL_COLS = 256;
inline bool wide_ld(unsigned ld) { return ld > PANEL_COLS && ld % PANEL_COLS == 0; }
inline unsigned ld_total_for(unsigned m) { return m <= PANEL_COLS ? ld_for(m) : (unsigned)round_up(m, PANEL_COLS); }

constexpr size_t POOL_GUARD_BYTES = 4096;            // debug pool: canary bytes after every block
constexpr unsigned POOL_KEEP_CALLS = 64;             // an unused cached work buffer survives this many public calls without being taken
constexpr int POOL_CANARY = 0xA5;
// poison: the block holds floating-point data (debug pool: handed out filled with NaN; integer blocks with zeros)
void native_comm_release(glf_ctx *ctx);              // comm.hip
void *pool_get(glf_ctx *ctx, size_t bytes, bool poison_nan = false); // nullptr on failure (last_error set)
void pool_put(glf_ctx *ctx, void *ptr);              // back to the pool
void pool_forget(glf_ctx *ctx, void *ptr);           // ownership leaves the pool (caller hipFree's it)
void pool_free_all(glf_ctx *ctx, bool only_unused);
void pool_age(glf_ctx *ctx);                         // frees the cached blocks no call has taken for POOL_KEEP_CALLS public calls

// RAII device buffer from the context's workspace pool (returned to the pool at scope exit).
template <typename T>
struct DevBuf {
    T *p = nullptr;
    size_t n = 0;
    glf_ctx *owner = nullptr;
    DevBuf() = default;
    DevBuf(const DevBuf &) = delete;
    DevBuf &operator=(const DevBuf &) = delete;
    ~DevBuf() { release(); }
    int alloc(glf_ctx *ctx, size_t count)
    {
        release();
        n = count;
        owner = ctx;
        if (count == 0) return GLF_OK;
        p = static_cast<T *>(pool_get(ctx, count * sizeof(T), std::is_floating_point<T>::value || std::is_same<T, _Float16>::value ||
                                                                  std::is_same<T, float4>::value));
        return p ? GLF_OK : GLF_ERR_NOMEM;
    }
    void release()
    {
        if (p) pool_put(owner, p);
        p = nullptr;
        n = 0;
    }
    T *take() // the block becomes a plain hipMalloc'd allocation owned by the caller
    {
        T *q = p;
        if (q) pool_forget(owner, q);
        p = nullptr;
        n = 0;
        return q;
    }
};

// Kernel-side description of the Gaussian kernel (hpc/affinity.c:59-121): the
// exponent is  -(dr^2+dc^2)/h_loc^2 - dv^2/h_val^2; we evaluate it as
// exp2(-(q*s_loc + u*s_val)) with s = log2(e)/h^2 so one v_exp_f32 does both
// reference exps. dr, dc, dv are exact small integers in f32.
struct KernelCoef {
    float s_loc; // log2(e) / h_loc^2 (0 for the photometric kernel)
    float s_val; // log2(e) / h_val^2 (0 for the spatial kernel); NLM: log2(e) / h^2 of the patch distance
    int kernel;  // GLF_KERNEL_*: NLM takes its own kernels (nlm.hip), the others share the positional ones
};
KernelCoef make_coef(int kernel, float h_loc, float h_val);
// host_util.cpp: P[v][w] = exp2(-s_val (v - w)^2) ~= F F^T in f64, F [256][rank] (strongest term first)
bool photometric_factor(double s_val, int max_chol, std::vector<double> &F, int &rank_out);
double photometric_factor_error(double s_val, const std::vector<double> &F, int rank, int R);

__device__ __forceinline__ float kernel_eval(float dr, float dc, float dv, float s_loc, float s_val)
{
    const float q = fmaf(dc, dc, dr * dr);
    const float t = fmaf(dv * dv, s_val, q * s_loc);
    return __builtin_amdgcn_exp2f(-t);
}

// ---- LDS-DMA staging ---------------------------------------------------------------------------
// global_load_lds_dwordx4: 64 lanes x 16 B land at LDS byte offset (wave-uniform base) + lane * 16, no
// VGPRs. Issued from inline asm on purpose: with the builtin, hipcc (ROCm 7.2) cannot prove that the DMA
// into one staging buffer does not alias the ds_reads of the other and puts s_waitcnt vmcnt(0) in front
// of the MFMA loop; staged through a register array instead, the array became a stack object (scratch:
// tens of GB of HBM writes per launch). The issuing wave must call lds_dma_drain() before the barrier that
// hands the buffer to its readers.
__device__ __forceinline__ void lds_dma_16B(const void *gptr, unsigned lds_byte_offset)
{
    asm volatile("s_mov_b32 m0, %1\n\t"
                 "s_nop 0\n\t"
                 "global_load_lds_dwordx4 %0, off"
                 :
                 : "v"(gptr), "s"(lds_byte_offset)
                 : "memory");
}
__device__ __forceinline__ void lds_dma_drain() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
// f32 at an absolute LDS byte address (ds_read_b32 with a computed address)
__device__ __forceinline__ float lds_f32(unsigned lds_byte_addr)
{
    return *reinterpret_cast<const __attribute__((address_space(3))) float *>(lds_byte_addr);
}
__device__ __forceinline__ unsigned lds_offset_of(const void *lds_ptr)
{
    return (unsigned)(uintptr_t)(__attribute__((address_space(3))) const char *)(const char *)lds_ptr;
}
// copy `pieces` KiB-sized pieces global -> LDS, piece i handled by wave (i % NW) of an NW-wave workgroup
template <int NW = 4>
__device__ __forceinline__ void lds_dma_copy(const void *gsrc, void *ldst, int pieces, int wave, int lane)
{
    const char *g = reinterpret_cast<const char *>(gsrc);
    const unsigned base = (unsigned)__builtin_amdgcn_readfirstlane((int)lds_offset_of(ldst)); // (wave-uniform: goes to m0)
    const int uw = __builtin_amdgcn_readfirstlane(wave);
    for (int i = uw; i < pieces; i += NW) lds_dma_16B(g + (size_t)i * 1024 + lane * 16, base + (unsigned)i * 1024u);
}

// ---- stage implementations (device pointers, all on ctx->stream) -----------------

// Sample table: float4 {row, col, value, 0} per sample + mask + device idx.
struct SampleTables {
    DevBuf<float4> samples;
    DevBuf<uint8_t> mask;
    DevBuf<uint32_t> idx;
};
int build_sample_tables(glf_ctx *ctx, const uint8_t *d_img, int width, int height, unsigned p,
                        const unsigned *h_idx, SampleTables &out);

// Partial degree D[i] = sum over pixels in rows [row0,row1) of K(sample i, pixel).
int degree_rows(glf_ctx *ctx, const uint8_t *d_img, int width, int height, int row0, int row1,
                const float4 *d_samples, unsigned p, KernelCoef coef, double *d_degree);

// Same sums, skipping pixels whose kernel entry underflows to exactly 0 in f32 for every sample of a block
// (h_idx: host sample indices, needed for the tile-major sample order). evaluated: entries executed.
// sums the pending mat-vec event pairs into ctx->mv_ms (synchronises on the last one)
int mv_collect(glf_ctx *ctx);
// device pointer to the cached start block X0 [round_up(p,64)][ld] for (p, m, ld, seed)
int start_block_cached(glf_ctx *ctx, unsigned p, unsigned m, unsigned ld, unsigned long long seed, const float **d_block);
int degree_rows_auto(glf_ctx *ctx, const uint8_t *d_img, int width, int height, int row0, int row1, const float4 *d_samples,
                     unsigned p, const unsigned *h_idx, KernelCoef coef, double *d_degree, int window, double *evaluated,
                     const uint32_t *d_idx = nullptr, double *d_ysum = nullptr, bool *have_ysum = nullptr);
int degree_rows_windowed(glf_ctx *ctx, const uint8_t *d_img, int width, int height, int row0, int row1,
                         const float4 *d_samples, unsigned p, const unsigned *h_idx, KernelCoef coef,
                         double *d_degree, double *evaluated);

// -no_approx: z = clamp(y - L y) with the full N x N Laplacian, never stored (affinity.hip)
int entire_computation(glf_ctx *ctx, const uint8_t *d_img, int width, int height, KernelCoef coef, uint8_t *d_out, float *d_zf,
                       double *alpha_out);

// K_A (scale = 1, diag untouched) or L_A (scale = -alpha, diagonal alpha * D_i).
// columns [col0, col0 + ncols) only (ncols = 0: all p columns); out is [p][ld] with local column index
int build_sample_matrix(glf_ctx *ctx, const float4 *d_samples, unsigned p, KernelCoef coef,
                        float *d_out, int64_t ld, bool laplacian, double alpha, const double *d_degree,
                        unsigned col0 = 0, unsigned ncols = 0, const uint8_t *d_img = nullptr, int width = 0, int height = 0,
                        const uint32_t *d_idx = nullptr); // (image and device indices: needed by the NLM kernel only)
// non-local-means affinity (nlm.hip): same contracts as degree_rows / build_sample_matrix / nystroem_contract
int nlm_degree_rows(glf_ctx *ctx, const uint8_t *d_img, int width, int height, int row0, int row1, const uint32_t *d_idx, unsigned p,
                    KernelCoef coef, double *d_degree);
int nlm_sample_matrix(glf_ctx *ctx, const uint8_t *d_img, int width, int height, const uint32_t *d_idx, unsigned p, KernelCoef coef,
                      float *d_out, int64_t ld, bool laplacian, double alpha, const double *d_degree, unsigned col0, unsigned ncols);
int nlm_nystroem(glf_ctx *ctx, const uint8_t *d_img, int width, int height, int64_t pix0, int64_t pix1, const uint8_t *d_mask,
                 const uint32_t *d_idx, unsigned p, KernelCoef coef, const float *d_psi, unsigned ld, float *d_phi, int raster,
                 double *d_c, float *kernel_ms);
int laplacian_from_KA(glf_ctx *ctx, const float *d_KA, int64_t ldk, unsigned p, float *d_LA, int64_t ld,
                      double alpha, const double *d_degree);

// Eigensolver pieces (eigen.hip)
// Row sharding of the symmetric p x p operator over the ranks of ctx->comm: this rank computes output
// rows [row0,row1) of A X and holds only the matching COLUMN block A[:, row0:row1) (= the transposed row
// block), stored [p][lda] with element (k, row) at A[k * lda + (row - row0)]. rows_per_rank is the
// all-gather block (a multiple of 64 >= ceil(p / size)); 0 = not sharded (A is the full matrix).
// the row-pass kernel of the grid-factored Nystroem contraction, timed launch by launch
struct RowpassStats {
    int launches = 0;
    float ms = 0.f;
    double flops = 0.0;
    // rank form: the fused T' + column-pass kernel, and the terms of the photometric expansion
    int col_launches = 0;
    float col_ms = 0.f;
    double col_flops = 0.0;
    int rank_R = 0;
};
struct GridOp; // L_A = alpha (D - K_A) applied in grid-factored form, never stored (nystroem_grid.inc)
int grid_op_create(glf_ctx *ctx, const float4 *d_samples, const unsigned *h_idx, unsigned p, int width, int height,
                   KernelCoef coef, GridOp **out); // GLF_ERR_UNSUPPORTED: not a tensor grid (or not the split-f16 mode)
void grid_op_destroy(GridOp *op);
unsigned grid_op_rows_per_rank(const GridOp *op, int size); // all-gather block: whole grid rows
void band_cache_free(glf_ctx *ctx);
// Filter applied in the epilogue of the band-form Nystroem kernel: z = y + gain Phi[px] . w - ysub y straight from the
// accumulators, Phi itself never written (hpc/display.c:60-78 fused into hpc/nystroem.c:41-42)
struct BandFilter {
    const float *w = nullptr; // [ld] filter weights x c (device)
    float gain = 0.f, ysub = 0.f;
    uint8_t *out = nullptr;   // [N] (absolute pixel index)
    float *zf = nullptr;      // [N] or null
    float *corr = nullptr;    // [pix1 - pix0] or null
};
// GLF_ERR_UNSUPPORTED: the band form does not apply (not a tensor grid, radius too large, or auto mode prefers a factored form)
int nystroem_band_filter(glf_ctx *ctx, const uint8_t *d_img, int width, int height, int64_t pix0, int64_t pix1, const float4 *d_samples,
                         const uint8_t *d_mask, const uint32_t *d_idx, unsigned p, KernelCoef coef, const float *d_psi, unsigned ld,
                         const BandFilter &flt, float *kernel_ms, uint64_t *entries_evaluated, double *mfma_flops, int *path,
                         RowpassStats *stats, const unsigned *h_idx = nullptr); // h_idx: the host copy of d_idx when the caller has it
// c = Psi^T (ysum - t) + Phi_A^T y_A  (ysum: the degree stage's value-weighted sums over all pixels; t = K_A y_A takes the
// sample pixels out again): Phi^T y without Phi. d_c [ld] f64.
int c_from_ysum(glf_ctx *ctx, const float *d_psi, const float *d_phiA, const double *d_ysum, const float *d_t, unsigned t_ld,
                const float4 *d_samples, unsigned p, unsigned ld, double *d_c);
// the sample pixels' outputs from their rows of Phi_A (the band kernel filters every pixel with its extended row)
int filter_sample_rows(glf_ctx *ctx, const float *d_phiA, unsigned n, unsigned ld, const uint32_t *d_idx, const uint8_t *d_img,
                       const float *d_w, float gain, float ysub, uint8_t *d_out, float *d_zf, float *d_corr, int64_t pix0);
int grid_op_path(const GridOp *op);                          // glf_stats.matvec_path: 1 exact grid form, 3 rank form
int grid_op_apply(glf_ctx *ctx, GridOp *op, const float *X, float *Y, unsigned ld, double alpha, const double *d_degree,
                  unsigned row0, unsigned row1, int window);

struct MatShard {
    unsigned row0 = 0, row1 = 0, rows_per_rank = 0;
    // when set, the operator is applied through the grid-factored form and `A` is not used (may be null)
    GridOp *grid = nullptr;
    double grid_alpha = 0.0;
    const double *grid_degree = nullptr;
    int grid_window = 0;
    // Exact-zero tile skipping of the split-f16 mat-vec (optional): kbox[c] = bounding box {rmin, rmax,
    // cmin, cmax} of samples [64 c, 64 c + 64); a (128-row block, 64-k tile) pair whose boxes are more
    // than `radius` pixels apart holds only entries with |2^10 A| < 2^-25, i.e. f16 hi = lo = 0.
    const int4 *kbox = nullptr;
    int radius = -1;
};
inline unsigned shard_rows_per_rank(unsigned p, int size) { return (unsigned)round_up(ceil_div(p, size), VEC_PAD); }
// rows to allocate for a p x ld vector block (zero padded) so that the all-gather blocks fit
inline unsigned vec_rows(unsigned p, const MatShard *sh, int size)
{
    const unsigned base = (unsigned)round_up(p, VEC_PAD);
    if (!sh || !sh->rows_per_rank) return base;
    const unsigned g = sh->rows_per_rank * (unsigned)size;
    return g > base ? g : base;
}
int block_matvec(glf_ctx *ctx, const float *A, int64_t lda, unsigned p, const float *X, float *Y,
                 unsigned mld, const MatShard *shard = nullptr);
int orthonormalise(glf_ctx *ctx, float *X, unsigned n, unsigned m, unsigned ld, double *h_norms);
int normalise(glf_ctx *ctx, float *X, unsigned n, unsigned m, unsigned ld, double *h_norms);
int inverse_power_iteration(glf_ctx *ctx, const float *A, int64_t lda, unsigned p, unsigned m, unsigned ld,
                            const double *h_X0, int opti_gs, double epsilon, double inner_rtol, int max_outer,
                            float *d_eigvecs, double *h_eigvals, glf_eig_stats *stats,
                            const MatShard *shard = nullptr, const float *d_dinv = nullptr,
                            const float *d_X0_block = nullptr);

// m > 256 (the reference default m = p - 1): panel-major blocks, see eigen.hip
int inverse_power_iteration_panels(glf_ctx *ctx, const float *A, int64_t lda, unsigned p, unsigned m, const double *h_X0,
                                   unsigned long long seed, int opti_gs, double epsilon, double inner_rtol, int max_outer,
                                   float *d_eigvecs, double *h_eigvals, glf_eig_stats *stats, const MatShard *shard,
                                   const float *d_dinv);
int orthonormalise_panels(glf_ctx *ctx, float *X, unsigned n, unsigned m, double *h_norms, bool normalise_only);

// Nystroem contraction (nystroem.hip): Phi[pix][j] = sum_i scale*K(sample i, pix) * Psi[i][j]
// for pixels [pix0, pix1). raster != 0: row = pix; else sample-first (rows of sample pixels skipped).
// cpart (optional): m doubles, += sum over NON-sample pixels of Phi[pix][j] * y[pix].
int nystroem_contract(glf_ctx *ctx, const uint8_t *d_img, int width, int height, int64_t pix0, int64_t pix1,
                      const float4 *d_samples, const uint8_t *d_mask, const uint32_t *d_idx, unsigned p,
                      KernelCoef coef, float scale, const float *d_psi, unsigned m, unsigned ld,
                      float *d_phi, int raster, double *d_c, float *kernel_ms, int window = 0,
                      uint64_t *entries_evaluated = nullptr, double *mfma_flops = nullptr, int *path = nullptr,
                      RowpassStats *rowpass = nullptr);
// box[c] = {rmin, rmax, cmin, cmax} of samples [64 c, 64 c + 64) (nystroem.hip)
int chunk_boxes(glf_ctx *ctx, const float4 *d_samples, unsigned p, int4 *d_box);
// Phi rows of the sample pixels <- Phi_A rows (hpc/nystroem.c:25-34 + hpc/utils.c:149-152)
int scatter_sample_rows(glf_ctx *ctx, const float *d_phiA, unsigned p, unsigned ld, const uint32_t *d_idx,
                        float *d_phi, int raster, const uint8_t *d_img, double *d_c, unsigned m);
int permute_rows(glf_ctx *ctx, const float *d_in, float *d_out, int64_t N, unsigned ld,
                 const uint32_t *d_idx, unsigned p);

// filter.hip
int phi_t_y(glf_ctx *ctx, const float *d_phi, const uint8_t *d_img, int64_t pix0, int64_t pix1, unsigned m,
            unsigned ld, double *d_c);
int phi_gram(glf_ctx *ctx, const float *d_phi, int64_t pix0, int64_t pix1, unsigned ld, double *d_G); // Phi^T Phi (f64 [ld][ld])
int apply_filter(glf_ctx *ctx, const uint8_t *d_img, const float *d_phi, int64_t pix0, int64_t pix1,
                 unsigned m, unsigned ld, const float *d_w, float gain, float ysub, uint8_t *d_out, float *d_zf,
                 float *d_corr = nullptr);
// the same filter panel by panel (m > 256): acc[px - pix0] (+)= sum_j Phi[px][j] w[j], then z = (1 - ysub) y + gain * acc
int filter_accumulate(glf_ctx *ctx, const float *d_phi, int64_t pix0, int64_t pix1, unsigned ld, const float *d_w, float *d_acc,
                      bool first);
int filter_finish(glf_ctx *ctx, const uint8_t *d_img, const float *d_acc, int64_t pix0, int64_t pix1, float gain, float ysub,
                  uint8_t *d_out, float *d_zf);

} // namespace glf
