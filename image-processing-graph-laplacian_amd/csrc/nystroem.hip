// nystroem.hip -- Nystroem eigen-extension (hpc/nystroem.c:5-69) as a dense f32 MFMA
// contraction with the K_B operand generated on the fly, plus the row permutation
// (hpc/utils.c:134-173).
//
//   Phi[pix][j] = sum_i  K(sample i, pix) * Psi[i][j],   Psi = -alpha * Phi_A * Pi^-1
//
// i.e. lower = L_B^T (phi_A Pi_A_Inv) (hpc/nystroem.c:41-42) with L_B = -alpha K_B
// (hpc/laplacian.c:37-38) never stored. GEMM view: C[N x m] = Kt[N x p] * Psi[p x m];
// the "A" operand Kt is computed in registers directly in the MFMA A-fragment layout
// (v_mfma_f32_32x32x2_f32: lane l supplies A[i = l & 31][k = l >> 5], so a lane owns
// one pixel and walks the samples), Psi tiles are staged through LDS and shared by
// the 4 waves of a workgroup. Algorithmic work 2 (N - p) p m flops; HBM traffic is
// the Phi write (4 N m bytes) -- the kernel is MFMA-bound for m >= 64 and
// VALU/transcendental-bound (kernel generation) below.
#include "glf_internal.hpp"

#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <vector>

namespace glf {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int NYS_KC = 64; // samples per LDS chunk

// number of samples with index < px (binary search in the ascending idx table)
__device__ __forceinline__ unsigned samples_before(const uint32_t *__restrict__ idx, unsigned p, uint32_t px)
{
    unsigned lo = 0, hi = p;
    while (lo < hi) {
        const unsigned mid = (lo + hi) >> 1;
        if (idx[mid] < px) lo = mid + 1;
        else hi = mid;
    }
    return lo;
}


// ---------------------------------------------------------------------------------------------
// Exact-zero skipping. A sample whose row or column distance to every pixel of the workgroup
// exceeds `radius` has exp2(-t) below the smallest value the contraction can represent (radius is
// chosen on the host from the arithmetic in use), so its whole chunk of 64 samples contributes
// exactly +0 to every accumulator: skipping it leaves the output bit-identical. Each workgroup
// compacts, in ascending order (accumulation order is preserved), the chunks whose bounding box
// comes within `radius` of its pixels' bounding box.
// ---------------------------------------------------------------------------------------------
constexpr int NYS_MAXCH = 4096; // chunks a workgroup can list (p <= 262 144); more -> dense on the host side
constexpr int NYS_MAXCH_LUT = 2048; // the LUT kernel has less LDS to spare
constexpr int NYS_LUT_RMAX = 511;   // largest half-width of the spatial factor table

// box[chunk] = {rmin, rmax, cmin, cmax} over the chunk's valid samples
__global__ void k_chunk_boxes(const float4 *__restrict__ samples, unsigned p, int4 *__restrict__ box)
{
    const unsigned ch = blockIdx.x * blockDim.x + threadIdx.x;
    if (ch * 64 >= p) return;
    int rmin = 1 << 30, rmax = -1, cmin = 1 << 30, cmax = -1;
    for (unsigned s = ch * 64; s < min(ch * 64 + 64, p); ++s) {
        const float4 v = samples[s];
        rmin = min(rmin, (int)v.x);
        rmax = max(rmax, (int)v.x);
        cmin = min(cmin, (int)v.y);
        cmax = max(cmax, (int)v.y);
    }
    box[ch] = make_int4(rmin, rmax, cmin, cmax);
}

// Returns the number of listed chunks (all of them, identity order, when radius < 0).
__device__ __forceinline__ int build_chunk_list(const int4 *__restrict__ box, int nchunks, int radius, int width,
                                                int64_t wg_first, int64_t wg_last, unsigned short *clist,
                                                int *scratch /* [257] */)
{
    if (radius < 0) return nchunks;
    const int t = threadIdx.x;
    const bool lister = t < 256; // workgroups may be wider than the 256 listing threads
    const int r_lo = (int)(wg_first / width), r_hi = (int)(wg_last / width);
    const int c_lo = (r_lo == r_hi) ? (int)(wg_first % width) : 0;
    const int c_hi = (r_lo == r_hi) ? (int)(wg_last % width) : width - 1;
    const int per = (nchunks + 255) / 256;
    unsigned bits = 0;
    int count = 0;
    for (int i = 0; i < per; ++i) {
        const int ch = t * per + i;
        if (lister && ch < nchunks) {
            const int4 b = box[ch];
            const bool rel = b.y >= r_lo - radius && b.x <= r_hi + radius && b.w >= c_lo - radius && b.z <= c_hi + radius;
            bits |= (unsigned)rel << i;
            count += rel;
        }
    }
    if (lister) scratch[t] = count;
    __syncthreads();
    int off = 0;
    for (int k = 0; k < min(t, 256); ++k) off += scratch[k];
    if (t == 255) scratch[256] = off + count;
    for (int i = 0; i < per; ++i)
        if (bits & (1u << i)) clist[off++] = (unsigned short)(t * per + i);
    __syncthreads();
    return scratch[256];
}

template <int MB, int PB, bool SKIP> // MB = ld / 32 column blocks; PB = 32-pixel blocks per wave; SKIP: chunk list
__global__ __launch_bounds__(256) void k_nystroem(const uint8_t *__restrict__ img, int width, int64_t pix0, int64_t pix1,
                                                   const float4 *__restrict__ samples, unsigned p, float s_loc,
                                                   float s_val, const float *__restrict__ psi,
                                                   float *__restrict__ phi, int raster,
                                                   const uint8_t *__restrict__ mask, const uint32_t *__restrict__ idx,
                                                   double *__restrict__ cpartial, const int4 *__restrict__ chunk_box,
                                                   int radius, unsigned *__restrict__ visited)
{
    constexpr int LD = MB * 32;
    constexpr int KC = NYS_KC;
    __shared__ unsigned short clist[SKIP ? NYS_MAXCH : 1];
    __shared__ int cscratch[SKIP ? 257 : 1];
    // one array for everything (guide: a second __shared__ object can de-pipeline LDS staging)
    __shared__ __attribute__((aligned(16))) float lds[2 * (NYS_KC * 4 + NYS_KC * MB * 32)];
    constexpr int BUF = KC * 4 + KC * LD; // floats per buffer: sample table then Psi tile

    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int half = lane >> 5, l31 = lane & 31;
    const int64_t wbase = pix0 + ((int64_t)blockIdx.x * 4 + wave) * (32 * PB);

    // this lane's pixels (one per 32-pixel block): exact integer coordinates in f32
    float pr[PB], pc[PB], pv[PB];
#pragma unroll
    for (int b = 0; b < PB; ++b) {
        int64_t px = wbase + 32 * b + l31;
        if (px >= pix1) px = pix1 - 1; // clamp loads; stores are guarded
        pr[b] = (float)(px / width);
        pc[b] = (float)(px % width);
        pv[b] = (float)img[px];
    }

    f32x16 acc[PB][MB];
#pragma unroll
    for (int b = 0; b < PB; ++b)
#pragma unroll
        for (int j = 0; j < MB; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[b][j][r] = 0.f;

    const int nchunks = (int)((p + KC - 1) / KC);
    const int64_t wg_first = pix0 + (int64_t)blockIdx.x * (128 * PB);
    const int64_t wg_last = min(wg_first + 128 * PB, pix1) - 1;
    const int nlist = SKIP ? build_chunk_list(chunk_box, nchunks, radius, width, wg_first, wg_last, clist, cscratch) : nchunks;
    if (visited && threadIdx.x == 0) visited[blockIdx.x] = (unsigned)nlist;
    auto chunk_at = [&](int i) { return SKIP ? clist[i] : i; };
    auto stage = [&](int chunk, int buf) {
        const unsigned s0 = (unsigned)chunk * KC;
        if (threadIdx.x < KC) {
            const unsigned s = s0 + threadIdx.x;
            reinterpret_cast<float4 *>(lds + buf * BUF)[threadIdx.x] = (s < p) ? samples[s] : make_float4(0.f, 0.f, 0.f, 0.f);
        }
        for (int e = threadIdx.x * 4; e < KC * LD; e += 256 * 4) {
            const unsigned s = s0 + e / LD;
            const float4 v = (s < p) ? *reinterpret_cast<const float4 *>(&psi[(size_t)s0 * LD + e])
                                     : make_float4(0.f, 0.f, 0.f, 0.f); // tail: K * 0 = 0
            *reinterpret_cast<float4 *>(lds + buf * BUF + KC * 4 + e) = v;
        }
    };
    if (nlist > 0) stage(chunk_at(0), 0);
    __syncthreads();
    for (int ch = 0; ch < nlist; ++ch) {
        const int buf = ch & 1;
        if (ch + 1 < nlist) stage(chunk_at(ch + 1), buf ^ 1);
        const float4 *stb = reinterpret_cast<const float4 *>(lds + buf * BUF) + half * (KC / 2);
        const float *psb = lds + buf * BUF + KC * 4 + (half * (KC / 2)) * LD + l31;
#pragma unroll 4
        for (int kk = 0; kk < KC / 2; ++kk) {
            const float4 s = stb[kk]; // two addresses per wave: broadcast within each half
            float a[PB];
#pragma unroll
            for (int b = 0; b < PB; ++b) a[b] = kernel_eval(pr[b] - s.x, pc[b] - s.y, pv[b] - s.z, s_loc, s_val);
#pragma unroll
            for (int j = 0; j < MB; ++j) {
                const float bf = psb[kk * LD + 32 * j];
#pragma unroll
                for (int b = 0; b < PB; ++b)
                    acc[b][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[b], bf, acc[b][j], 0, 0, 0);
            }
        }
        __syncthreads();
    }

    // ---- epilogue: store Phi rows, accumulate c_j += Phi[pix][j] * y[pix] over non-sample pixels
    // C/D layout: col = lane & 31, row = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5)
    float csum[MB];
#pragma unroll
    for (int j = 0; j < MB; ++j) csum[j] = 0.f;
#pragma unroll
    for (int b = 0; b < PB; ++b) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int64_t px = wbase + 32 * b + (r & 3) + 8 * (r >> 2) + 4 * half;
            if (px >= pix1) continue;
            const bool is_sample = mask[px] != 0;
            int64_t dst;
            if (raster) dst = px;
            else {
                if (is_sample) continue; // sample rows come from Phi_A (hpc/nystroem.c:25-34)
                dst = (int64_t)p + px - (int64_t)samples_before(idx, p, (uint32_t)px);
            }
            const float y = is_sample ? 0.f : (float)img[px];
#pragma unroll
            for (int j = 0; j < MB; ++j) {
                const float v = acc[b][j][r];
                phi[(size_t)dst * LD + 32 * j + l31] = v;
                csum[j] = fmaf(v, y, csum[j]);
            }
        }
    }
    if (cpartial) {
        __syncthreads(); // all waves are done with the staging buffers: reuse as scratch
        float *red = lds; // [4 waves][LD]
#pragma unroll
        for (int j = 0; j < MB; ++j) {
            float v = csum[j] + __shfl_xor(csum[j], 32, 64);
            if (half == 0) red[wave * LD + 32 * j + l31] = v;
        }
        __syncthreads();
        if (threadIdx.x < LD)
            cpartial[(size_t)blockIdx.x * LD + threadIdx.x] =
                ((double)red[threadIdx.x] + (double)red[LD + threadIdx.x]) +
                ((double)red[2 * LD + threadIdx.x] + (double)red[3 * LD + threadIdx.x]);
    }
}

// out[c] (+)= sum_rows in[row][c]   two-level, fixed order
__global__ __launch_bounds__(256) void k_rows_sum_lvl1(const double *__restrict__ in, int64_t nrows, unsigned ld,
                                                        double *__restrict__ out)
{
    __shared__ double sh[256];
    const int col = threadIdx.x % ld, rl = threadIdx.x / ld, nrl = 256 / ld;
    const int64_t base = (int64_t)blockIdx.x * 1024;
    double s = 0.0;
    for (int64_t r = rl; r < 1024; r += nrl) {
        if (base + r >= nrows) break;
        s += in[(size_t)(base + r) * ld + col];
    }
    sh[threadIdx.x] = s;
    __syncthreads();
    if (threadIdx.x < ld) {
        double t = 0.0;
        for (int r = 0; r < nrl; ++r) t += sh[r * ld + col];
        out[(size_t)blockIdx.x * ld + col] = t;
    }
}

__global__ void k_rows_sum_lvl2(const double *__restrict__ in, int nrows, unsigned ld, double *__restrict__ out, int accumulate)
{
    const int c = threadIdx.x;
    if (c >= (int)ld) return;
    double s = accumulate ? out[c] : 0.0;
    for (int r = 0; r < nrows; ++r) s += in[(size_t)r * ld + c];
    out[c] = s;
}

static int sum_rows(glf_ctx *ctx, const double *d_in, int64_t nrows, unsigned ld, double *d_out, bool accumulate)
{
    const int n1 = (int)ceil_div(nrows, 1024);
    DevBuf<double> tmp;
    GLF_TRY(tmp.alloc(ctx, (size_t)n1 * ld));
    hipLaunchKernelGGL(k_rows_sum_lvl1, dim3(n1), dim3(256), 0, ctx->stream, d_in, nrows, ld, tmp.p);
    hipLaunchKernelGGL(k_rows_sum_lvl2, dim3(1), dim3(256), 0, ctx->stream, tmp.p, n1, ld, d_out, accumulate ? 1 : 0);
    GLF_LAUNCH_CHECK(ctx);
    GLF_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return GLF_OK;
}

int chunk_boxes(glf_ctx *ctx, const float4 *d_samples, unsigned p, int4 *d_box)
{
    const unsigned nchunks = (unsigned)ceil_div(p, 64);
    hipLaunchKernelGGL(k_chunk_boxes, dim3((nchunks + 63) / 64), dim3(64), 0, ctx->stream, d_samples, p, d_box);
    GLF_LAUNCH_CHECK(ctx);
    return GLF_OK;
}

// Host side of the exact-zero skipping: radius from the contraction's arithmetic, chunk boxes, and
// the per-workgroup "chunks visited" counters (summed for the executed-work accounting).
struct NysWindow {
    DevBuf<int4> box;
    DevBuf<unsigned> visited;
    int radius = -1;
    int init(glf_ctx *ctx, const float4 *d_samples, unsigned p, KernelCoef coef, int window, double t_zero, int64_t nwg,
             int max_chunks = NYS_MAXCH)
    {
        const unsigned nchunks = (unsigned)ceil_div(p, 64);
        radius = -1;
        // K < 2^-t_zero is exactly zero for the contraction; t >= s_loc * d^2 for a row or column distance d
        if (window && coef.s_loc > 0.f && nchunks <= (unsigned)max_chunks)
            radius = (int)std::floor(std::sqrt(t_zero / (double)coef.s_loc)) + 1;
        GLF_TRY(box.alloc(ctx, nchunks));
        GLF_TRY(visited.alloc(ctx, (size_t)nwg));
        hipLaunchKernelGGL(k_chunk_boxes, dim3((nchunks + 63) / 64), dim3(64), 0, ctx->stream, d_samples, p, box.p);
        GLF_LAUNCH_CHECK(ctx);
        return GLF_OK;
    }
    int total(glf_ctx *ctx, int64_t nwg, uint64_t *out)
    {
        std::vector<unsigned> h((size_t)nwg);
        GLF_HIP(ctx, hipMemcpyAsync(h.data(), visited.p, sizeof(unsigned) * (size_t)nwg, hipMemcpyDeviceToHost, ctx->stream));
        GLF_HIP(ctx, hipStreamSynchronize(ctx->stream));
        uint64_t s = 0;
        for (unsigned v : h) s += v;
        *out = s;
        return GLF_OK;
    }
};

template <int MB, int PB>
static int launch_nystroem(glf_ctx *ctx, const uint8_t *d_img, int width, int64_t pix0, int64_t pix1,
                           const float4 *d_samples, const uint8_t *d_mask, const uint32_t *d_idx, unsigned p,
                           KernelCoef coef, const float *d_psi, float *d_phi, int raster, double *d_c, float *kernel_ms,
                           int window, uint64_t *entries_evaluated)
{
    constexpr int LD = MB * 32;
    const int64_t npix = pix1 - pix0;
    const int64_t nwg = ceil_div(npix, 4 * 32 * PB);
    DevBuf<double> cpart;
    if (d_c) GLF_TRY(cpart.alloc(ctx, (size_t)nwg * LD));
    // f32 operands: K underflows to exactly 0 below 2^-149 (t > 150)
    NysWindow win;
    GLF_TRY(win.init(ctx, d_samples, p, coef, window, 151.0, nwg));
    if (kernel_ms) GLF_HIP(ctx, hipEventRecord(ctx->ev[6], ctx->stream));
    if (win.radius >= 0)
        hipLaunchKernelGGL((k_nystroem<MB, PB, true>), dim3((unsigned)nwg), dim3(256), 0, ctx->stream, d_img, width, pix0, pix1,
                           d_samples, p, coef.s_loc, coef.s_val, d_psi, d_phi, raster, d_mask, d_idx, d_c ? cpart.p : nullptr,
                           win.box.p, win.radius, win.visited.p);
    else
        hipLaunchKernelGGL((k_nystroem<MB, PB, false>), dim3((unsigned)nwg), dim3(256), 0, ctx->stream, d_img, width, pix0, pix1,
                           d_samples, p, coef.s_loc, coef.s_val, d_psi, d_phi, raster, d_mask, d_idx, d_c ? cpart.p : nullptr,
                           win.box.p, win.radius, win.visited.p);
    GLF_LAUNCH_CHECK(ctx);
    if (kernel_ms) GLF_HIP(ctx, hipEventRecord(ctx->ev[7], ctx->stream));
    if (entries_evaluated) { // listed chunks x 64 samples x the workgroup's pixels
        GLF_TRY(win.total(ctx, nwg, entries_evaluated));
        *entries_evaluated *= 64ull * (4 * 32 * PB);
    }
    if (d_c) GLF_TRY(sum_rows(ctx, cpart.p, nwg, LD, d_c, true));
    if (kernel_ms) {
        GLF_HIP(ctx, hipEventSynchronize(ctx->ev[7]));
        GLF_HIP(ctx, hipEventElapsedTime(kernel_ms, ctx->ev[6], ctx->ev[7]));
    }
    return GLF_OK;
}

// =====================================================================================================
// Split-f16 contraction: the same Phi = K^T Psi, but on the f16 matrix pipe so that the MFMAs overlap the
// VALU kernel generation (the f32-input MFMA shares the f32 FMA pipe with the VALU and cannot: see
// tools/mfma_probe.hip). Both operands are split into an f16 (hi, lo) pair with hi + lo carrying 22
// significant bits:  K' = 2^15 K,  Psi' = T_j Psi (T_j a power of two per column, |Psi'| < 2^14);
//   K' Psi' ~= Khi Phi_hi + Khi Plo + Klo Phi_hi     (the dropped lo*lo term is 2^-22 relative)
// accumulated in f32 by v_mfma_f32_32x32x16_f16 and rescaled by 2^-15 / T_j (exact) in the epilogue.
// A lane owns one pixel per 32-pixel block and generates K for 8 consecutive samples per 16-sample
// MFMA step (A fragment: lane l holds A[row l&31][k = 8 (l>>5) + j], j = 0..7).
// =====================================================================================================

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

constexpr float NYS_F16_KSCALE_LOG2 = 15.0f; // K' = 2^15 K  (max 32768 < 65504)

// Psi [p_pad][ld] f32  ->  per chunk of 64 samples: [step t (4)][jb (MB)][piece q (2)][lane (64)][8 halves]
// (every (t, jb, q) fragment block is 1 KiB, lane-linear: one conflict-free ds_read_b128 per lane).
__global__ __launch_bounds__(256) void k_psi_split_f16(const float *__restrict__ psi, unsigned p_pad, unsigned ld,
                                                        const float *__restrict__ colscale, _Float16 *__restrict__ out)
{
    const size_t e = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (e >= (size_t)p_pad * ld) return;
    const unsigned s = (unsigned)(e / ld), c = (unsigned)(e % ld);
    const float v = psi[e] * colscale[c];
    const _Float16 hi = (_Float16)v;                 // round to nearest
    const _Float16 lo = (_Float16)(v - (float)hi);
    const unsigned mb = ld / 32;
    const unsigned chunk = s / 64, t = (s % 64) / 16, h = (s % 16) / 8, j = s % 8, jb = c / 32, r = c % 32;
    const size_t frag = (((size_t)chunk * 4 + t) * mb + jb) * 2; // + q
    const size_t lane = h * 32 + r;
    out[((frag + 0) * 64 + lane) * 8 + j] = hi;
    out[((frag + 1) * 64 + lane) * 8 + j] = lo;
}

// sample records {row, col, value, 0} x p_pad  ->  per chunk of 64: rows[64] cols[64] vals[64] pad[64]
__global__ void k_samples_soa(const float4 *__restrict__ samples, unsigned p_pad, float *__restrict__ out)
{
    const unsigned s = blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= p_pad) return;
    const float4 v = samples[s];
    float *o = out + (size_t)(s / 64) * 256 + (s % 64);
    o[0] = v.x;
    o[64] = v.y;
    o[128] = v.z;
    o[192] = __int_as_float(128 * (int)v.z); // value * (32 banks * 4 B): row stride of the LUT kernel's photometric table
}

// out[c] = max_i |psi[i][c]|: coalesced row reads, 128 rows per workgroup, then one workgroup over the block maxima
constexpr int CAM_ROWS = 128;
__global__ __launch_bounds__(256) void k_col_absmax_part(const float *__restrict__ psi, unsigned p, unsigned ld,
                                                          float *__restrict__ part)
{
    __shared__ float sh[256];
    const unsigned col = threadIdx.x % ld, rl = threadIdx.x / ld, nrl = 256 / ld;
    const unsigned base = blockIdx.x * CAM_ROWS;
    float m = 0.f;
    for (unsigned r = rl; r < CAM_ROWS; r += nrl) {
        const unsigned i = base + r;
        if (i >= p) break;
        m = fmaxf(m, fabsf(psi[(size_t)i * ld + col]));
    }
    sh[threadIdx.x] = m;
    __syncthreads();
    if (threadIdx.x < ld) {
        for (unsigned r = 1; r < nrl; ++r) m = fmaxf(m, sh[r * ld + col]);
        part[(size_t)blockIdx.x * ld + col] = m;
    }
}
__global__ __launch_bounds__(256) void k_col_absmax_fin(const float *__restrict__ part, int nblk, unsigned ld, float *__restrict__ out)
{
    __shared__ float sh[256];
    const unsigned col = threadIdx.x % ld, pt = threadIdx.x / ld, npt = 256 / ld;
    float m = 0.f;
    int b = (int)pt;
    for (; b + 7 * (int)npt < nblk; b += 8 * (int)npt) { // eight loads in flight at a time
        float x[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) x[u] = part[(size_t)(b + u * (int)npt) * ld + col];
#pragma unroll
        for (int u = 0; u < 8; ++u) m = fmaxf(m, x[u]);
    }
    for (; b < nblk; b += (int)npt) m = fmaxf(m, part[(size_t)b * ld + col]);
    sh[threadIdx.x] = m;
    __syncthreads();
    if (threadIdx.x < ld) {
        for (unsigned r = 1; r < npt; ++r) m = fmaxf(m, sh[r * ld + col]);
        out[col] = m;
    }
}
// scratch: ceil(p / CAM_ROWS) * ld floats
static void col_absmax(hipStream_t st, const float *psi, unsigned p, unsigned ld, float *scratch, float *out)
{
    const int nblk = (int)ceil_div(p, CAM_ROWS);
    hipLaunchKernelGGL(k_col_absmax_part, dim3(nblk), dim3(256), 0, st, psi, p, ld, scratch);
    hipLaunchKernelGGL(k_col_absmax_fin, dim3(1), dim3(256), 0, st, scratch, nblk, ld, out);
}

// |a - b| + c in one instruction (hipcc expands __usad into max/min/sub/add)
__device__ __forceinline__ unsigned sad_u32(unsigned a, unsigned b, unsigned c)
{
    unsigned d;
    asm("v_sad_u32 %0, %1, %2, %3" : "=v"(d) : "v"(a), "v"(b), "v"(c));
    return d;
}

// LUT = true generates K' without transcendental instructions. The bilateral kernel factors as
//   K' = [2^15 exp2(-s_loc dr^2)] * exp2(-s_loc dc^2) * exp2(-s_val dv^2) = Er(dr) * Ec(dc) * P(|dv|)
// and, with u8 pixel values and integer coordinates, each factor takes few distinct values:
//   P   256 entries, gathered per entry from a copy replicated over the 32 ds_read_b32 banks
//       (lane l reads bank l % 32: conflict-free whatever the pixel values are);
//       -- the address is one v_sad_u32: |128 pv - 128 sv| + (plut + 4 (l % 32));
//   Ec  2 (R + 32) + 1 entries, R = the distance at which the factor is exactly 0 in f32. A block of 32
//       consecutive pixels lies in one image row (width % 32 == 0), so its lanes read 32 consecutive
//       entries from a per-(block, sample) base; the base is clamped once per (wave, block, chunk) into
//       the zero margins of the table, and the per-entry address is one v_add_u32;
//   Er  the same factor with the 2^15 folded in, gathered once per (wave, block, chunk) likewise.
// The tables are correctly rounded from f64 on the host, so K' carries three roundings (< 2 ulp).
// Per entry: 4 VALU + 2 ds_read_b32 instead of 9 VALU + v_exp_f32, then the same 2 VALU f16 split.
// NW = waves per workgroup: 4, or 6 for the LUT kernel (its 74 KiB of LDS allow two workgroups per CU; six
// waves each make that three waves per SIMD, which the gather latency needs).
template <int MB, int PB, bool SKIP, bool LUT, int NW>
__global__ __launch_bounds__(NW * 64) __attribute__((amdgpu_waves_per_eu(NW == 8 ? 4 : 1, NW == 8 ? 4 : 8))) void k_nystroem_f16s(const uint8_t *__restrict__ img, int width, int64_t pix0, int64_t pix1,
                                                        const float *__restrict__ soa, unsigned p, float s_loc, float s_val,
                                                        const _Float16 *__restrict__ psi16, const float *__restrict__ invscale,
                                                        float *__restrict__ phi, int raster,
                                                        const uint8_t *__restrict__ mask, const uint32_t *__restrict__ idx,
                                                        double *__restrict__ cpartial, const int4 *__restrict__ chunk_box,
                                                        int radius, unsigned *__restrict__ visited,
                                                        const float *__restrict__ lut, int lut_r)
{
    constexpr int LD = MB * 32;
    __shared__ unsigned short clist[SKIP ? (LUT ? NYS_MAXCH_LUT : NYS_MAXCH) : 1];
    __shared__ int cscratch[SKIP ? 257 : 1];
    // LUT generation (see the header comment of this kernel)
    __shared__ float plut[LUT ? 256 * 32 : 1];
    __shared__ float elut[LUT ? 2 * (NYS_LUT_RMAX + 32 * PB) + 1 : 1];
    __shared__ __attribute__((aligned(16))) float erw[LUT ? NW * 64 : 4];
    __shared__ __attribute__((aligned(16))) unsigned cbw[LUT ? NW * 64 : 4];
    constexpr int PSI_F4 = 4 * MB * 2 * 64;      // float4 (16 B) words of one chunk's Psi fragments
    constexpr int BUF_F4 = 64 + PSI_F4;          // + 1 KiB sample SoA
    __shared__ __attribute__((aligned(16))) float4 lds[2 * BUF_F4];

    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int half = lane >> 5;
    const int64_t wbase = pix0 + ((int64_t)blockIdx.x * NW + wave) * (32 * PB);

    float pr[PB], pc[PB], pv[PB];
    int pv128[PB]; // LUT: 128 * value
#pragma unroll
    for (int b = 0; b < PB; ++b) {
        int64_t px = wbase + 32 * b + (lane & 31);
        if (px >= pix1) px = pix1 - 1;
        pr[b] = (float)(px / width);
        pc[b] = (float)(px % width);
        pv[b] = (float)img[px];
        pv128[b] = 128 * (int)img[px];
    }
    // LUT: the wave's 32 PB consecutive pixels lie in one image row (host-checked): first column and row
    const int64_t wfirst = min(wbase, pix1 - 1);
    const int pcol0 = (int)(wfirst % width), prow = (int)(wfirst / width);
    const int lut_w = lut_r + 32 * PB, lut_n = 2 * lut_w + 1;
    const unsigned lane4 = (lane & 31) * 4;
    const unsigned pbase = LUT ? lds_offset_of(plut) + lane4 : 0;
    const unsigned ebase = LUT ? lds_offset_of(elut) : 0;
    if (LUT) {
        for (int i = threadIdx.x; i < 256 * 32; i += NW * 64) plut[i] = lut[i >> 5];
        for (int i = threadIdx.x; i < lut_n; i += NW * 64) elut[i] = lut[256 + i];
        // visible to all waves after the barrier that follows the first stage()
    }
    const float *lut_er = lut + 256 + lut_n;
    f32x16 acc[PB][MB];
#pragma unroll
    for (int b = 0; b < PB; ++b)
#pragma unroll
        for (int j = 0; j < MB; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[b][j][r] = 0.f;

    const int nchunks = (int)((p + 63) / 64);
    const int64_t wg_first = pix0 + (int64_t)blockIdx.x * (NW * 32 * PB);
    const int64_t wg_last = min(wg_first + NW * 32 * PB, pix1) - 1;
    const int nlist = SKIP ? build_chunk_list(chunk_box, nchunks, radius, width, wg_first, wg_last, clist, cscratch) : nchunks;
    if (visited && threadIdx.x == 0) visited[blockIdx.x] = (unsigned)nlist;
    auto chunk_at = [&](int i) { return SKIP ? clist[i] : i; };
    const float4 *gsoa = reinterpret_cast<const float4 *>(soa);
    const float4 *gpsi = reinterpret_cast<const float4 *>(psi16);
    // Double-buffered LDS-DMA staging: a chunk is 1 KiB of sample SoA + PSI_F4 * 16 B of Psi fragments,
    // contiguous in both tables, copied with no VGPRs while the previous chunk is being contracted.
    constexpr int PIECES = PSI_F4 * 16 / 1024;
    auto stage = [&](int chunk, int buf) {
        float4 *dst = lds + buf * BUF_F4;
        if (wave == 0) lds_dma_16B(reinterpret_cast<const char *>(gsoa + (size_t)chunk * 64) + lane * 16, lds_offset_of(dst));
        lds_dma_copy<NW>(gpsi + (size_t)chunk * PSI_F4, dst + 64, PIECES, wave, lane);
    };
    if (nlist > 0) stage(chunk_at(0), 0);
    lds_dma_drain();
    __syncthreads();
    for (int ch = 0; ch < nlist; ++ch) {
        const int buf = ch & 1;
        if (ch + 1 < nlist) stage(chunk_at(ch + 1), buf ^ 1); // in flight during the MFMA/VALU sweep below
        const float *ssoa = reinterpret_cast<const float *>(lds + buf * BUF_F4);
        const f16x8 *sfrag = reinterpret_cast<const f16x8 *>(lds + buf * BUF_F4 + 64);
        if (LUT) {
            // Er of sample `lane` for each pixel block of this wave (wave-private LDS slots: the wave
            // executes its LDS operations in order, no barrier needed)
            const int srow = (int)ssoa[lane], scol = (int)ssoa[64 + lane];
            const int d = min(max(prow - srow + lut_w, 0), 2 * lut_w);
            erw[wave * 64 + lane] = lut_er[d];
            // lanes add 0 .. 32 PB - 1 to the base: clamped so that they stay inside the table's zero margins
            const int cb = min(max(pcol0 - scol, -lut_w), lut_r + 1);
            cbw[wave * 64 + lane] = ebase + 4u * (unsigned)(cb + lut_w);
            __builtin_amdgcn_wave_barrier();
        }
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            // this lane's 8 samples of the step: 16 t + 8 half + (0..7)
            float sr[8], sc[8], sv[8];
            unsigned sv128[8], cb[8];
            float er[8];
            if (LUT) {
                const uint4 *qi = reinterpret_cast<const uint4 *>(ssoa + 192 + 16 * t + 8 * half);
                const uint4 v0 = qi[0], v1 = qi[1];
                sv128[0] = v0.x; sv128[1] = v0.y; sv128[2] = v0.z; sv128[3] = v0.w;
                sv128[4] = v1.x; sv128[5] = v1.y; sv128[6] = v1.z; sv128[7] = v1.w;
                const float4 *qe = reinterpret_cast<const float4 *>(erw + wave * 64 + 16 * t + 8 * half);
                const float4 e0 = qe[0], e1 = qe[1];
                er[0] = e0.x; er[1] = e0.y; er[2] = e0.z; er[3] = e0.w;
                er[4] = e1.x; er[5] = e1.y; er[6] = e1.z; er[7] = e1.w;
                const uint4 *qc = reinterpret_cast<const uint4 *>(cbw + wave * 64 + 16 * t + 8 * half);
                const uint4 c0 = qc[0], c1 = qc[1];
                cb[0] = c0.x; cb[1] = c0.y; cb[2] = c0.z; cb[3] = c0.w;
                cb[4] = c1.x; cb[5] = c1.y; cb[6] = c1.z; cb[7] = c1.w;
            } else {
                const float4 *q = reinterpret_cast<const float4 *>(ssoa + 16 * t + 8 * half);
                const float4 r0 = q[0], r1 = q[1], c0 = q[16], c1 = q[17], v0 = q[32], v1 = q[33];
                sr[0] = r0.x; sr[1] = r0.y; sr[2] = r0.z; sr[3] = r0.w; sr[4] = r1.x; sr[5] = r1.y; sr[6] = r1.z; sr[7] = r1.w;
                sc[0] = c0.x; sc[1] = c0.y; sc[2] = c0.z; sc[3] = c0.w; sc[4] = c1.x; sc[5] = c1.y; sc[6] = c1.z; sc[7] = c1.w;
                sv[0] = v0.x; sv[1] = v0.y; sv[2] = v0.z; sv[3] = v0.w; sv[4] = v1.x; sv[5] = v1.y; sv[6] = v1.z; sv[7] = v1.w;
            }
            f16x8 ah[PB], al[PB];
#pragma unroll
            for (int b = 0; b < PB; ++b) {
#pragma unroll
                for (int e = 0; e < 8; e += 2) {
                    f32x2 y;
#pragma unroll
                    for (int u = 0; u < 2; ++u) {
                        if (LUT) {
                            const float ec = lds_f32(cb[e + u] + (lane4 + 128u * b));
                            const float pp = lds_f32(sad_u32((unsigned)pv128[b], sv128[e + u], pbase));
                            y[u] = (er[e + u] * ec) * pp;
                        } else {
                            const float dr = pr[b] - sr[e + u], dc = pc[b] - sc[e + u], dv = pv[b] - sv[e + u];
                            const float q = fmaf(dc, dc, dr * dr);
                            y[u] = __builtin_amdgcn_exp2f(NYS_F16_KSCALE_LOG2 - fmaf(dv * dv, s_val, q * s_loc));
                        }
                    }
                    const f16x2 h2 = __builtin_convertvector(y, f16x2);            // round to nearest
                    // scalar subtractions on purpose: v_pk_add_f32 is slower than two v_sub_f32 beside MFMAs
                    f32x2 res;
                    res[0] = y[0] - (float)h2[0]; // exact in f32
                    res[1] = y[1] - (float)h2[1];
                    const f16x2 l2 = __builtin_convertvector(res, f16x2);
                    ah[b][e] = h2[0];
                    ah[b][e + 1] = h2[1];
                    al[b][e] = l2[0];
                    al[b][e + 1] = l2[1];
                }
            }
            // The two waves of a SIMD drift into lockstep (both generating, then both contracting) unless the
            // one in its MFMA burst wins the issue arbitration: the other then fills the gaps with its VALU.
            __builtin_amdgcn_s_setprio(1);
#pragma unroll
            for (int j = 0; j < MB; ++j) {
                const f16x8 bh = sfrag[((t * MB + j) * 2 + 0) * 64 + lane];
                const f16x8 bl = sfrag[((t * MB + j) * 2 + 1) * 64 + lane];
#pragma unroll
                for (int b = 0; b < PB; ++b) {
                    acc[b][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[b], bh, acc[b][j], 0, 0, 0);
                    acc[b][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[b], bl, acc[b][j], 0, 0, 0);
                    acc[b][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[b], bh, acc[b][j], 0, 0, 0);
                }
            }
            __builtin_amdgcn_s_setprio(0);
        }
        lds_dma_drain(); // this wave's pieces of the next chunk have landed ...
        __syncthreads(); // ... and so have everybody else's; buffer `buf` is free again
    }

    // ---- epilogue (same as k_nystroem, plus the exact power-of-two rescale per column)
    const int l31 = lane & 31;
    float inv[MB], csum[MB];
#pragma unroll
    for (int j = 0; j < MB; ++j) {
        inv[j] = invscale[32 * j + l31];
        csum[j] = 0.f;
    }
#pragma unroll
    for (int b = 0; b < PB; ++b) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int64_t px = wbase + 32 * b + (r & 3) + 8 * (r >> 2) + 4 * half;
            if (px >= pix1) continue;
            const bool is_sample = mask[px] != 0;
            int64_t dst;
            if (raster) dst = px;
            else {
                if (is_sample) continue;
                dst = (int64_t)p + px - (int64_t)samples_before(idx, p, (uint32_t)px);
            }
            const float y = is_sample ? 0.f : (float)img[px];
#pragma unroll
            for (int j = 0; j < MB; ++j) {
                const float v = acc[b][j][r] * inv[j];
                phi[(size_t)dst * LD + 32 * j + l31] = v;
                csum[j] = fmaf(v, y, csum[j]);
            }
        }
    }
    if (cpartial) {
        __syncthreads();
        float *red = reinterpret_cast<float *>(lds); // [NW waves][LD]
#pragma unroll
        for (int j = 0; j < MB; ++j) {
            float v = csum[j] + __shfl_xor(csum[j], 32, 64);
            if (half == 0) red[wave * LD + 32 * j + l31] = v;
        }
        __syncthreads();
        if (threadIdx.x < LD) {
            double tot = 0.0;
#pragma unroll
            for (int w = 0; w < NW; w += 2)
                tot += (double)red[w * LD + threadIdx.x] + (double)red[(w + 1) * LD + threadIdx.x];
            cpartial[(size_t)blockIdx.x * LD + threadIdx.x] = tot;
        }
    }
}

template <int MB, int PB>
static int launch_nystroem_f16s(glf_ctx *ctx, const uint8_t *d_img, int width, int64_t pix0, int64_t pix1,
                                const float4 *d_samples, const uint8_t *d_mask, const uint32_t *d_idx, unsigned p,
                                KernelCoef coef, const float *d_psi, float *d_phi, int raster, double *d_c, float *kernel_ms,
                                int window, uint64_t *entries_evaluated)
{
    constexpr unsigned LD = MB * 32;
    const unsigned p_pad = (unsigned)round_up(p, NYS_PAD);
    hipStream_t st = ctx->stream;
    // per-column power-of-two scale T_j with |Psi| T_j < 2^14
    DevBuf<float> colmax, colscale, invscale, soa;
    DevBuf<_Float16> psi16;
    GLF_TRY(colmax.alloc(ctx, LD));
    GLF_TRY(colscale.alloc(ctx, LD));
    GLF_TRY(invscale.alloc(ctx, LD));
    GLF_TRY(soa.alloc(ctx, (size_t)p_pad * 4));
    GLF_TRY(psi16.alloc(ctx, (size_t)p_pad * LD * 2));
    DevBuf<float> camx;
    GLF_TRY(camx.alloc(ctx, (size_t)ceil_div(p, CAM_ROWS) * LD));
    col_absmax(st, d_psi, p, LD, camx.p, colmax.p);
    GLF_LAUNCH_CHECK(ctx);
    std::vector<float> hmax(LD), hscale(LD), hinv(LD);
    GLF_HIP(ctx, hipMemcpyAsync(hmax.data(), colmax.p, sizeof(float) * LD, hipMemcpyDeviceToHost, st));
    GLF_HIP(ctx, hipStreamSynchronize(st));
    for (unsigned j = 0; j < LD; ++j) {
        int e = 0;
        if (hmax[j] > 0.f && std::isfinite(hmax[j])) {
            (void)std::frexp(hmax[j], &e);   // hmax = f * 2^e, f in [0.5, 1)
            e = 14 - e;                       // hmax * 2^e in [2^13, 2^14)
        }
        if (e > 100) e = 100;
        if (e < -100) e = -100;
        hscale[j] = std::ldexp(1.0f, e);
        hinv[j] = std::ldexp(1.0f, -e - (int)NYS_F16_KSCALE_LOG2);
    }
    GLF_HIP(ctx, hipMemcpyAsync(colscale.p, hscale.data(), sizeof(float) * LD, hipMemcpyHostToDevice, st));
    GLF_HIP(ctx, hipMemcpyAsync(invscale.p, hinv.data(), sizeof(float) * LD, hipMemcpyHostToDevice, st));
    hipLaunchKernelGGL(k_psi_split_f16, dim3((unsigned)ceil_div((int64_t)p_pad * LD, 256)), dim3(256), 0, st, d_psi, p_pad, LD,
                       colscale.p, psi16.p);
    hipLaunchKernelGGL(k_samples_soa, dim3((p_pad + 255) / 256), dim3(256), 0, st, d_samples, p_pad, soa.p);
    GLF_LAUNCH_CHECK(ctx);

    // Table-driven generation (k_nystroem_f16s<.., LUT = true>): needs each wave's 32 PB pixels inside one image
    // row, a spatial factor that reaches exact zero within NYS_LUT_RMAX, and LDS for two workgroups per CU.
    constexpr int NW_LUT = 8;
    int lut_r = 0;
    if (MB <= 2 && coef.s_loc > 0.f && width % (32 * PB) == 0 && pix0 % (32 * PB) == 0 && !ctx->tune.nys_no_lut) {
        // exp2(-s d^2) < 2^-150 rounds to +0 in f32
        lut_r = (int)std::floor(std::sqrt(150.5 / (double)coef.s_loc)) + 1;
        if (lut_r > NYS_LUT_RMAX) lut_r = 0;
    }
    const bool use_lut = lut_r > 0;
    const int nw = use_lut ? NW_LUT : 4;
    const int64_t npix = pix1 - pix0;
    const int64_t nwg = ceil_div(npix, nw * 32 * PB);
    DevBuf<double> cpart;
    if (d_c) GLF_TRY(cpart.alloc(ctx, (size_t)nwg * LD));
    std::vector<float> hlut;
    DevBuf<float> dlut;
    if (use_lut) {
        const int lut_w = lut_r + 32 * PB, ne = 2 * lut_w + 1; // 32 PB zero entries of margin on both sides
        hlut.resize(256 + 2 * (size_t)ne);
        for (int e = 0; e < 256; ++e) hlut[e] = (float)std::exp2(-(double)coef.s_val * e * e);
        for (int i = 0; i < ne; ++i) {
            const double t = (double)coef.s_loc * (double)(i - lut_w) * (double)(i - lut_w);
            hlut[256 + i] = (float)std::exp2(-t);
            hlut[256 + ne + i] = (float)std::exp2((double)NYS_F16_KSCALE_LOG2 - t);
        }
        GLF_TRY(dlut.alloc(ctx, hlut.size()));
        GLF_HIP(ctx, hipMemcpyAsync(dlut.p, hlut.data(), sizeof(float) * hlut.size(), hipMemcpyHostToDevice, st));
    }
    // K' = 2^15 K rounds to zero in f16 (hi and lo) below 2^-25: t > 40; 40.5 covers the f32 rounding of t
    NysWindow win;
    GLF_TRY(win.init(ctx, d_samples, p, coef, window, 40.5, nwg, use_lut ? NYS_MAXCH_LUT : NYS_MAXCH));
    if (kernel_ms) GLF_HIP(ctx, hipEventRecord(ctx->ev[6], st));
#define GLF_NYS_LAUNCH(SKIP_, LUT_, NW_)                                                                                     \
    hipLaunchKernelGGL((k_nystroem_f16s<MB, PB, SKIP_, LUT_, NW_>), dim3((unsigned)nwg), dim3(NW_ * 64), 0, st, d_img, width, \
                       pix0, pix1, soa.p, p, coef.s_loc, coef.s_val, psi16.p, invscale.p, d_phi, raster, d_mask, d_idx,      \
                       d_c ? cpart.p : nullptr, win.box.p, win.radius, win.visited.p, dlut.p, lut_r)
    if constexpr (MB <= 2) {
        if (use_lut) {
            if (win.radius >= 0) GLF_NYS_LAUNCH(true, true, NW_LUT);
            else GLF_NYS_LAUNCH(false, true, NW_LUT);
        }
    }
    if (!use_lut) {
        if (win.radius >= 0) GLF_NYS_LAUNCH(true, false, 4);
        else GLF_NYS_LAUNCH(false, false, 4);
    }
#undef GLF_NYS_LAUNCH
    GLF_LAUNCH_CHECK(ctx);
    if (kernel_ms) GLF_HIP(ctx, hipEventRecord(ctx->ev[7], st));
    if (entries_evaluated) { // listed chunks x 64 samples x the workgroup's pixels
        GLF_TRY(win.total(ctx, nwg, entries_evaluated));
        *entries_evaluated *= 64ull * (uint64_t)(nw * 32 * PB);
    }
    if (d_c) GLF_TRY(sum_rows(ctx, cpart.p, nwg, LD, d_c, true));
    GLF_HIP(ctx, hipStreamSynchronize(st)); // hscale/hinv and the DevBufs go out of scope
    if (kernel_ms) GLF_HIP(ctx, hipEventElapsedTime(kernel_ms, ctx->ev[6], ctx->ev[7]));
    return GLF_OK;
}

#include "grid_common.inc"
#include "nystroem_grid.inc"

int nystroem_contract(glf_ctx *ctx, const uint8_t *d_img, int width, int height, int64_t pix0, int64_t pix1,
                      const float4 *d_samples, const uint8_t *d_mask, const uint32_t *d_idx, unsigned p,
                      KernelCoef coef, float /*scale folded into psi*/, const float *d_psi, unsigned m, unsigned ld,
                      float *d_phi, int raster, double *d_c, float *kernel_ms, int window, uint64_t *entries_evaluated,
                      double *mfma_flops, int *path, RowpassStats *rowpass)
{
    uint64_t entries_local = 0;
    if (!entries_evaluated) entries_evaluated = &entries_local;
    if (path) *path = 0;
    if (mfma_flops) *mfma_flops = 0.0;
    const int64_t N = (int64_t)width * height;
    if (pix0 < 0 || pix1 > N || pix0 > pix1 || !valid_ld(ld) || m > ld)
        return set_error(ctx, GLF_ERR_INVALID, "nystroem_contract: bad range or ld=%u", ld);
    if (kernel_ms) *kernel_ms = 0.f;
    if (pix0 == pix1) return GLF_OK;
    if (coef.kernel == GLF_KERNEL_NLM) { // patch distances generated on the vector pipe, f32 MFMA contraction (nlm.hip)
        *entries_evaluated = (uint64_t)p * (uint64_t)(pix1 - pix0);
        if (mfma_flops) *mfma_flops = 2.0 * (double)*entries_evaluated * ld;
        return nlm_nystroem(ctx, d_img, width, height, pix0, pix1, d_mask, d_idx, p, coef, d_psi, ld, d_phi, raster, d_c, kernel_ms);
    }
    {
        // a tensor-grid sample set (hpc/sampling.c always yields one) takes the factored contraction
        const int rc = nystroem_contract_grid(ctx, d_img, width, height, pix0, pix1, d_samples, d_mask, d_idx, p, coef, d_psi, ld,
                                              d_phi, raster, d_c, kernel_ms, window, entries_evaluated, mfma_flops, rowpass, path);
        if (rc != GLF_ERR_UNSUPPORTED) return rc;
        if (path) *path = 0;
    }
    int rc = GLF_ERR_UNSUPPORTED;
    if (ctx->contraction == GLF_CONTRACT_F16_SPLIT) {
        switch (ld) {
        case 32: rc = launch_nystroem_f16s<1, 2>(ctx, d_img, width, pix0, pix1, d_samples, d_mask, d_idx, p, coef, d_psi, d_phi, raster, d_c, kernel_ms, window, entries_evaluated); break;
        case 64: rc = launch_nystroem_f16s<2, 2>(ctx, d_img, width, pix0, pix1, d_samples, d_mask, d_idx, p, coef, d_psi, d_phi, raster, d_c, kernel_ms, window, entries_evaluated); break;
        case 128: rc = launch_nystroem_f16s<4, 1>(ctx, d_img, width, pix0, pix1, d_samples, d_mask, d_idx, p, coef, d_psi, d_phi, raster, d_c, kernel_ms, window, entries_evaluated); break;
        case 256: rc = launch_nystroem_f16s<8, 1>(ctx, d_img, width, pix0, pix1, d_samples, d_mask, d_idx, p, coef, d_psi, d_phi, raster, d_c, kernel_ms, window, entries_evaluated); break;
        }
        if (rc == GLF_OK && mfma_flops) *mfma_flops = 6.0 * (double)*entries_evaluated * ld;
        return rc;
    }
    switch (ld) {
    case 32:
        rc = launch_nystroem<1, 2>(ctx, d_img, width, pix0, pix1, d_samples, d_mask, d_idx, p, coef, d_psi, d_phi, raster, d_c, kernel_ms, window, entries_evaluated); break;
    case 64:
        rc = launch_nystroem<2, 2>(ctx, d_img, width, pix0, pix1, d_samples, d_mask, d_idx, p, coef, d_psi, d_phi, raster, d_c, kernel_ms, window, entries_evaluated); break;
    case 128:
        rc = launch_nystroem<4, 2>(ctx, d_img, width, pix0, pix1, d_samples, d_mask, d_idx, p, coef, d_psi, d_phi, raster, d_c, kernel_ms, window, entries_evaluated); break;
    case 256:
        rc = launch_nystroem<8, 1>(ctx, d_img, width, pix0, pix1, d_samples, d_mask, d_idx, p, coef, d_psi, d_phi, raster, d_c, kernel_ms, window, entries_evaluated); break;
    }
    if (rc == GLF_OK && mfma_flops) *mfma_flops = 2.0 * (double)*entries_evaluated * ld;
    return rc;
}

// ---- sample rows: Phi[row(i)] = Phi_A[i]; c += Phi_A^T y_A -------------------------------------

__global__ __launch_bounds__(256) void k_scatter_sample_rows(const float *__restrict__ phiA, unsigned p, unsigned ld,
                                                              const uint32_t *__restrict__ idx, float *__restrict__ phi,
                                                              int raster, const uint8_t *__restrict__ img,
                                                              double *__restrict__ partial)
{
    __shared__ double sh[256];
    const int col = threadIdx.x % ld, rl = threadIdx.x / ld, nrl = 256 / ld;
    double s = 0.0;
    const unsigned base = blockIdx.x * 512;
    for (unsigned r = rl; r < 512; r += nrl) {
        const unsigned i = base + r;
        if (i >= p) break;
        const float v = phiA[(size_t)i * ld + col];
        const uint32_t px = idx[i];
        const size_t dst = raster ? (size_t)px : (size_t)i; // hpc/utils.c:149-152 / hpc/nystroem.c:25-34
        phi[dst * ld + col] = v;
        if (partial) s += (double)v * (double)img[px];
    }
    if (partial) {
        sh[threadIdx.x] = s;
        __syncthreads();
        if (threadIdx.x < ld) {
            double t = 0.0;
            for (int r = 0; r < nrl; ++r) t += sh[r * ld + col];
            partial[(size_t)blockIdx.x * ld + col] = t;
        }
    }
}

int scatter_sample_rows(glf_ctx *ctx, const float *d_phiA, unsigned p, unsigned ld, const uint32_t *d_idx, float *d_phi,
                        int raster, const uint8_t *d_img, double *d_c, unsigned /*m*/)
{
    const int nblk = (int)ceil_div(p, 512);
    DevBuf<double> part;
    if (d_c) GLF_TRY(part.alloc(ctx, (size_t)nblk * ld));
    hipLaunchKernelGGL(k_scatter_sample_rows, dim3(nblk), dim3(256), 0, ctx->stream, d_phiA, p, ld, d_idx, d_phi, raster,
                       d_img, d_c ? part.p : nullptr);
    GLF_LAUNCH_CHECK(ctx);
    if (d_c) {
        hipLaunchKernelGGL(k_rows_sum_lvl2, dim3(1), dim3(256), 0, ctx->stream, part.p, nblk, ld, d_c, 1);
        GLF_LAUNCH_CHECK(ctx);
        GLF_HIP(ctx, hipStreamSynchronize(ctx->stream));
    }
    return GLF_OK;
}

// ---- Permutation (hpc/utils.c:134-173): out[raster pixel] = in[sample-first position] ------------

__global__ __launch_bounds__(256) void k_permute_rows(const float *__restrict__ in, float *__restrict__ out, int64_t N,
                                                       unsigned ld, const uint32_t *__restrict__ idx, unsigned p)
{
    const int lanes_per_row = ld / 4; // float4 per lane
    const int64_t px = (int64_t)blockIdx.x * (256 / lanes_per_row) + threadIdx.x / lanes_per_row;
    if (px >= N) return;
    const int q = threadIdx.x % lanes_per_row;
    const unsigned before = samples_before(idx, p, (uint32_t)px);
    const bool is_sample = before < p && idx[before] == (uint32_t)px;
    const size_t src = is_sample ? (size_t)before : (size_t)p + (size_t)px - before;
    reinterpret_cast<float4 *>(out + (size_t)px * ld)[q] = reinterpret_cast<const float4 *>(in + src * ld)[q];
}

int permute_rows(glf_ctx *ctx, const float *d_in, float *d_out, int64_t N, unsigned ld, const uint32_t *d_idx, unsigned p)
{
    if (!valid_ld(ld)) return set_error(ctx, GLF_ERR_INVALID, "permute_rows: ld=%u", ld);
    const int rows_per_block = 256 / (ld / 4);
    hipLaunchKernelGGL(k_permute_rows, dim3((unsigned)ceil_div(N, rows_per_block)), dim3(256), 0, ctx->stream, d_in, d_out,
                       N, ld, d_idx, p);
    GLF_LAUNCH_CHECK(ctx);
    return GLF_OK;
}

} // namespace glf

#ifdef RK_STAMP
extern "C" int glf_debug_rank_stamps(unsigned long long *out)
{
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(glf::g_rk_stamps), sizeof(unsigned long long) * 8 * 256) == hipSuccess ? 0 : -1;
}
#endif
