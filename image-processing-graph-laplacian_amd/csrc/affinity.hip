// affinity.hip -- bilateral affinity kernels: sample tables, degree (row sums of
// [K_A K_B] with K_B generated on the fly), K_A / L_A materialisation.
//
// Replaces ComputeAffinityMatrices (hpc/affinity.c:129-262) and the row-sum half
// of ComputeLaplacianMatrix (hpc/laplacian.c:18-29). K_B (p x (N-p)) is never
// stored: 5.3 TiB at 4096^2 / 0.5 %.
#include "glf_internal.hpp"

#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <cstring>

namespace glf {

// ---- sample tables ------------------------------------------------------------------

__global__ void k_sample_tables(const uint8_t *__restrict__ img, int width, int64_t N, unsigned p,
                                const uint32_t *__restrict__ idx, float4 *__restrict__ samples,
                                uint8_t *__restrict__ mask)
{
    const unsigned i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= p) return;
    const uint32_t px = idx[i];
    if ((int64_t)px >= N) return; // validated on the host as well
    // num2x / num2y, hpc/utils.c:11-19: x = row, y = column
    samples[i] = make_float4((float)(px / (uint32_t)width), (float)(px % (uint32_t)width), (float)img[px], 0.f);
    mask[px] = 1;
}

int build_sample_tables(glf_ctx *ctx, const uint8_t *d_img, int width, int height, unsigned p,
                        const unsigned *h_idx, SampleTables &out)
{
    const int64_t N = (int64_t)width * height;
    if (p == 0 || !h_idx) return set_error(ctx, GLF_ERR_INVALID, "empty sample set");
    for (unsigned i = 0; i < p; ++i) {
        if ((int64_t)h_idx[i] >= N || (i && h_idx[i] <= h_idx[i - 1]))
            return set_error(ctx, GLF_ERR_INVALID, "sample_indices must be ascending and < width*height (i=%u)", i);
    }
    const size_t p_pad = (size_t)round_up(p, NYS_PAD); // zero records past p (Nystroem LDS staging)
    GLF_TRY(out.samples.alloc(ctx, p_pad));
    GLF_HIP(ctx, hipMemsetAsync(out.samples.p, 0, sizeof(float4) * p_pad, ctx->stream));
    GLF_TRY(out.mask.alloc(ctx, (size_t)N));
    GLF_TRY(out.idx.alloc(ctx, p));
    GLF_HIP(ctx, hipMemcpyAsync(out.idx.p, h_idx, sizeof(uint32_t) * p, hipMemcpyHostToDevice, ctx->stream));
    GLF_HIP(ctx, hipMemsetAsync(out.mask.p, 0, (size_t)N, ctx->stream));
    hipLaunchKernelGGL(k_sample_tables, dim3((p + 255) / 256), dim3(256), 0, ctx->stream, d_img, width, N, p,
                       out.idx.p, out.samples.p, out.mask.p);
    GLF_LAUNCH_CHECK(ctx);
    // h_idx is pageable host memory: make sure the async copy has consumed it
    GLF_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return GLF_OK;
}

// ---- degree: D[i] = sum_pixels K(sample i, pixel) --------------------------------------
//
// lane = sample (its row/col/value live in VGPRs); a workgroup (256 samples) sweeps a chunk of 16
// image rows. The Gaussian is separable in the two coordinates and the photometric part:
//     K = Er * Ec * P,  Er = exp2(-s_loc (r_i - r)^2)  (16 registers per chunk),
//                       Ec = exp2(-s_loc (c_i - c)^2)  (once per column, 16 rows),
//                       P  = exp2(-s_val dv^2)         (per pixel).
// Two implementations share the sweep: degree_chunk<KY> evaluates P with v_exp_f32 (4 VALU + 1
// transcendental per entry; used by the full-matrix mode, which also needs sum K y), and
// degree_chunk_lut gathers P from an LDS table (the D_A path, see below). Pixel values are staged in a
// 16-row LDS tile and read back as wave-wide broadcasts (ds_read_b128 = 4 pixels). HBM traffic is
// ~N bytes per sample block: the kernels are bound by the VALU/transcendental pipes or by the LDS.
// Accumulation: f32 over one 4-column strip, f64 across strips, chunks and the final reduction, all in
// a fixed order => bitwise reproducible, and identical between the dense sweep and the
// exact-zero-skipping sweep (which visits a 4-column / 16-row aligned sub-range).

constexpr int DEG_THREADS = 256;
constexpr int DEG_ROWS = 16;   // rows per chunk (fixed: part of the accumulation order)
constexpr int DEG_SEGW = 512;  // columns per staged LDS tile

// Sum over rows [r_begin, r_end) (at most DEG_ROWS) and columns [c_begin, c_end), c_begin % 4 == 0.
// KY: also accumulate sum K * pixel value into *ky (the K y product of the full-matrix mode).
template <bool KY>
__device__ __forceinline__ double degree_chunk(const uint8_t *__restrict__ img, int width, int r_begin, int r_end, int c_begin,
                                               int c_end, float4 s, float s_loc, float s_val, float *tile /* [16][SEGW] */,
                                               double *ky = nullptr)
{
    const int nr = r_end - r_begin;
    float a[DEG_ROWS];
#pragma unroll
    for (int r = 0; r < DEG_ROWS; ++r) {
        const float dr = s.x - (float)(r_begin + r);
        a[r] = dr * dr * s_loc;
    }
    double total = 0.0, total_ky = 0.0;
    for (int c0 = c_begin; c0 < c_end; c0 += DEG_SEGW) {
        const int seg = min(DEG_SEGW, c_end - c0);
        const int seg4 = (seg + 3) & ~3;
        __syncthreads(); // previous tile fully consumed
        for (int e = threadIdx.x; e < nr * seg4; e += DEG_THREADS) {
            const int r = e / seg4, c = e % seg4;
            tile[r * DEG_SEGW + c] = (c < seg) ? (float)img[(size_t)(r_begin + r) * width + c0 + c] : 0.f;
        }
        __syncthreads();
        float dc = s.y - (float)c0; // exact integer, decremented per column
        for (int c = 0; c < seg4; c += 4) {
            float ec[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                // columns past the end of the range (padding of the last strip) get weight 0
                ec[u] = (c + u < seg) ? __builtin_amdgcn_exp2f(-(dc * dc * s_loc)) : 0.f;
                dc -= 1.f;
            }
            float acc = 0.f, acc_ky = 0.f;
#pragma unroll
            for (int r = 0; r < DEG_ROWS; ++r) {
                if (r < nr) { // uniform
                    const float4 v = *reinterpret_cast<const float4 *>(&tile[r * DEG_SEGW + c]);
                    const float pv[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        const float dv = s.z - pv[u];
                        if (KY) {
                            const float k = ec[u] * __builtin_amdgcn_exp2f(-fmaf(dv * dv, s_val, a[r]));
                            acc += k;
                            acc_ky = fmaf(k, pv[u], acc_ky);
                        } else {
                            acc = fmaf(ec[u], __builtin_amdgcn_exp2f(-fmaf(dv * dv, s_val, a[r])), acc);
                        }
                    }
                }
            }
            total += (double)acc;
            if (KY) total_ky += (double)acc_ky;
        }
    }
    if (KY) *ky = total_ky;
    return total;
}

// ---- the same sweep with the photometric factor from a table (the D_A path) ---------------------
// K = Er * Ec * P(|v_i - pixel|): the sample value v_i is a u8 fixed per lane and the pixel value is a
// u8 broadcast to the wave, so P is one of 256 numbers. It is gathered from a copy of the table
// replicated over the 32 ds_read_b32 banks (lane l reads bank l % 32: conflict-free), addressed by a
// single v_sad_u32: |128 v_i - 128 pixel| + (table + 4 (l % 32)). Per entry: v_sad_u32 + ds_read_b32 +
// v_fma_f32 (+ 1/4 ds_read_b128 of four staged pixels, + 1/4 v_fma for the row factor) instead of
// 4 VALU + v_exp_f32: the kernel moves from the transcendental pipe to the LDS (3 LDS cycles per
// entry-wave, shared by the CU's 4 SIMDs). Er (16 per chunk) and Ec (4 per strip) still use v_exp_f32;
// P is correctly rounded from f64 on the host. Same strips, chunks and accumulation order as above.
// A workgroup is 512 threads: both halves hold the same 256 samples and take 8 of the chunk's 16 rows each
// (their f64 partial sums are added at the end), so that three workgroups per CU give 6 waves per SIMD to
// hide the gather latency.
constexpr int DEGL_SEGW = 256; // 16 KiB tile + 32 KiB table: three workgroups per CU
constexpr int DEGL_THREADS = 2 * DEG_THREADS;
constexpr int DEGL_HROWS = DEG_ROWS / 2;

__device__ __forceinline__ unsigned deg_sad_u32(unsigned a, unsigned b, unsigned c)
{
    unsigned d;
    asm("v_sad_u32 %0, %1, %2, %3" : "=v"(d) : "v"(a), "v"(b), "v"(c));
    return d;
}

__device__ __forceinline__ void degree_lut_init(const float *__restrict__ ptab, float *plut /* [256][32] */)
{
    for (int e = threadIdx.x; e < 256 * 32; e += DEGL_THREADS) plut[e] = ptab[e >> 5];
    // made visible by the first barrier of degree_chunk_lut
}

__device__ __forceinline__ double degree_chunk_lut(const uint8_t *__restrict__ img, int width, int r_begin, int r_end,
                                                   int c_begin, int c_end, float4 s, float s_loc,
                                                   unsigned *tile /* [16][DEGL_SEGW] of 128 * pixel */, const float *plut,
                                                   double *comb /* [DEG_THREADS] */)
{
    const int nr = r_end - r_begin;
    const int rsel = __builtin_amdgcn_readfirstlane((int)(threadIdx.x / DEG_THREADS)) * DEGL_HROWS; // this half's first row
    float er[DEGL_HROWS];
#pragma unroll
    for (int r = 0; r < DEGL_HROWS; ++r) {
        const float dr = s.x - (float)(r_begin + rsel + r);
        er[r] = __builtin_amdgcn_exp2f(-(dr * dr * s_loc));
    }
    const unsigned sv128 = 128u * (unsigned)s.z;
    const unsigned pbase = lds_offset_of(plut) + 4u * (threadIdx.x & 31);
    double total = 0.0;
    for (int c0 = c_begin; c0 < c_end; c0 += DEGL_SEGW) {
        const int seg = min(DEGL_SEGW, c_end - c0);
        const int seg4 = (seg + 3) & ~3;
        __syncthreads(); // previous tile fully consumed
        for (int e = threadIdx.x; e < nr * seg4; e += DEGL_THREADS) {
            const int r = e / seg4, c = e % seg4;
            tile[r * DEGL_SEGW + c] = (c < seg) ? 128u * (unsigned)img[(size_t)(r_begin + r) * width + c0 + c] : 0u;
        }
        __syncthreads();
        float dc = s.y - (float)c0; // exact integer, decremented per column
        for (int c = 0; c < seg4; c += 4) {
            float ec[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                // columns past the end of the range (padding of the last strip) get weight 0
                ec[u] = (c + u < seg) ? __builtin_amdgcn_exp2f(-(dc * dc * s_loc)) : 0.f;
                dc -= 1.f;
            }
            float acc = 0.f;
#pragma unroll
            for (int r = 0; r < DEGL_HROWS; ++r) {
                if (rsel + r < nr) { // uniform
                    const uint4 v = *reinterpret_cast<const uint4 *>(&tile[(rsel + r) * DEGL_SEGW + c]);
                    float row = ec[0] * lds_f32(deg_sad_u32(sv128, v.x, pbase));
                    row = fmaf(ec[1], lds_f32(deg_sad_u32(sv128, v.y, pbase)), row);
                    row = fmaf(ec[2], lds_f32(deg_sad_u32(sv128, v.z, pbase)), row);
                    row = fmaf(ec[3], lds_f32(deg_sad_u32(sv128, v.w, pbase)), row);
                    acc = fmaf(er[r], row, acc);
                }
            }
            total += (double)acc;
        }
    }
    // rows 0-7 + rows 8-15, in that order
    __syncthreads();
    if (rsel) comb[threadIdx.x - DEG_THREADS] = total;
    __syncthreads();
    if (!rsel) total += comb[threadIdx.x];
    return total; // valid in the first half of the workgroup
}

struct DegLutLds {
    unsigned tile[DEG_ROWS * DEGL_SEGW];
    float plut[256 * 32];
    double comb[DEG_THREADS];
};

__global__ __launch_bounds__(DEGL_THREADS) void k_degree(const uint8_t *__restrict__ img, int width, int row0,
                                                         int row1, const float4 *__restrict__ samples, unsigned p,
                                                         float s_loc, const float *__restrict__ ptab,
                                                         double *__restrict__ partial)
{
    __shared__ __attribute__((aligned(16))) DegLutLds sh;
    degree_lut_init(ptab, sh.plut);
    const unsigned i = blockIdx.x * DEG_THREADS + threadIdx.x % DEG_THREADS;
    const bool live = i < p;
    const float4 s = live ? samples[i] : make_float4(0.f, 0.f, 0.f, 0.f);
    const int r_begin = row0 + (int)blockIdx.y * DEG_ROWS;
    const int r_end = min(r_begin + DEG_ROWS, row1);
    const double total = degree_chunk_lut(img, width, r_begin, r_end, 0, width, s, s_loc, sh.tile, sh.plut, sh.comb);
    if (live && threadIdx.x < DEG_THREADS) partial[(size_t)blockIdx.y * p + i] = total;
}

// P(e) = exp2(-s_val e^2), e = 0..255, correctly rounded; freed by the caller's DevBuf after its stream sync
static int upload_photometric_table(glf_ctx *ctx, KernelCoef coef, DevBuf<float> &d_tab, std::vector<float> &h_tab)
{
    h_tab.resize(256);
    for (int e = 0; e < 256; ++e) h_tab[e] = (float)std::exp2(-(double)coef.s_val * e * e);
    GLF_TRY(d_tab.alloc(ctx, 256));
    GLF_HIP(ctx, hipMemcpyAsync(d_tab.p, h_tab.data(), sizeof(float) * 256, hipMemcpyHostToDevice, ctx->stream));
    return GLF_OK;
}

__global__ void k_reduce_partials(const double *__restrict__ partial, unsigned p, int nchunks,
                                  double *__restrict__ out)
{
    const unsigned i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= p) return;
    double s = 0.0;
    for (int k = 0; k < nchunks; ++k) s += partial[(size_t)k * p + i];
    out[i] = s;
}

int degree_rows(glf_ctx *ctx, const uint8_t *d_img, int width, int height, int row0, int row1,
                const float4 *d_samples, unsigned p, KernelCoef coef, double *d_degree)
{
    if (row0 < 0 || row1 > height || row0 > row1) return set_error(ctx, GLF_ERR_INVALID, "bad row range");
    if (row0 == row1) {
        GLF_HIP(ctx, hipMemsetAsync(d_degree, 0, sizeof(double) * p, ctx->stream));
        return GLF_OK;
    }
    const int nsb = (int)ceil_div(p, DEG_THREADS);
    const int nchunks = (int)ceil_div(row1 - row0, DEG_ROWS);
    if (nchunks > 65535) return set_error(ctx, GLF_ERR_UNSUPPORTED, "image too tall for one degree launch");
    DevBuf<double> partial;
    GLF_TRY(partial.alloc(ctx, (size_t)nchunks * p));
    DevBuf<float> d_ptab;
    std::vector<float> h_ptab;
    GLF_TRY(upload_photometric_table(ctx, coef, d_ptab, h_ptab));
    hipLaunchKernelGGL(k_degree, dim3(nsb, nchunks), dim3(DEGL_THREADS), 0, ctx->stream, d_img, width, row0, row1,
                       d_samples, p, coef.s_loc, d_ptab.p, partial.p);
    GLF_LAUNCH_CHECK(ctx);
    hipLaunchKernelGGL(k_reduce_partials, dim3((p + 255) / 256), dim3(256), 0, ctx->stream, partial.p, p, nchunks,
                       d_degree);
    GLF_LAUNCH_CHECK(ctx);
    GLF_HIP(ctx, hipStreamSynchronize(ctx->stream)); // partial is freed at scope exit
    return GLF_OK;
}

// ---- degree with exact-zero skipping -----------------------------------------------------------
// exp2(-t) underflows to exactly 0 in f32 once s_loc * d^2 > 150 for a row or column distance d, so a
// block of 256 spatially compact samples only needs the pixels within `radius` of its bounding box; the
// rest would add +0 (or denormals that cannot change a sum >= 1). Samples are taken in a tile-major
// order (perm) so that a block is compact in both directions. Same accumulation structure as k_degree.
__global__ __launch_bounds__(DEGL_THREADS) void k_degree_win(const uint8_t *__restrict__ img, int width, int row0, int row1,
                                                             const float4 *__restrict__ samples, unsigned p,
                                                             const uint32_t *__restrict__ perm,
                                                             const int4 *__restrict__ blk_box, int radius, float s_loc,
                                                             const float *__restrict__ ptab, double *__restrict__ partial)
{
    __shared__ __attribute__((aligned(16))) DegLutLds sh;
    degree_lut_init(ptab, sh.plut);
    const uint32_t i = perm[blockIdx.x * DEG_THREADS + threadIdx.x % DEG_THREADS]; // 0xFFFFFFFF = padding
    const bool live = i < p;
    const float4 s = live ? samples[i] : make_float4(0.f, 0.f, 0.f, 0.f);
    const int4 box = blk_box[blockIdx.x]; // {rmin, rmax, cmin, cmax} of the block's samples
    // window, snapped to the dense sweep's 16-row chunks and 4-column strips (same accumulation order)
    const int rw0 = row0 + ((max(row0, box.x - radius) - row0) / DEG_ROWS) * DEG_ROWS;
    const int rw1 = min(row1, box.y + radius + 1);
    const int cw0 = max(0, box.z - radius) & ~3, cw1 = min(width, box.w + radius + 1);
    const int r_begin = rw0 + (int)blockIdx.y * DEG_ROWS;
    const int r_end = min(r_begin + DEG_ROWS, min(rw1 + DEG_ROWS, row1)); // whole chunk, as the dense sweep
    double total = 0.0;
    if (r_begin < rw1) total = degree_chunk_lut(img, width, r_begin, r_end, cw0, cw1, s, s_loc, sh.tile, sh.plut, sh.comb); // uniform branch
    if (live && threadIdx.x < DEG_THREADS) partial[(size_t)blockIdx.y * p + i] = total;
}

int degree_rows_windowed(glf_ctx *ctx, const uint8_t *d_img, int width, int height, int row0, int row1,
                         const float4 *d_samples, unsigned p, const unsigned *h_idx, KernelCoef coef, double *d_degree,
                         double *evaluated)
{
    if (row0 < 0 || row1 > height || row0 > row1) return set_error(ctx, GLF_ERR_INVALID, "bad row range");
    if (evaluated) *evaluated = 0.0;
    if (row0 == row1) {
        GLF_HIP(ctx, hipMemsetAsync(d_degree, 0, sizeof(double) * p, ctx->stream));
        return GLF_OK;
    }
    const int radius = (int)std::floor(std::sqrt(151.0 / (double)coef.s_loc)) + 1; // t > 150 => exp2(-t) == 0 in f32
    // tile-major sample order: tiles of about 256 samples
    const int64_t N = (int64_t)width * height;
    const int tile = std::max(8, (int)std::ceil(16.0 * std::sqrt((double)N / (double)p)));
    const int ntc = (int)ceil_div(width, tile);
    std::vector<uint32_t> perm(p);
    for (unsigned i = 0; i < p; ++i) perm[i] = i;
    std::vector<uint32_t> key(p);
    for (unsigned i = 0; i < p; ++i) key[i] = (uint32_t)((h_idx[i] / (unsigned)width) / (unsigned)tile * (unsigned)ntc + (h_idx[i] % (unsigned)width) / (unsigned)tile);
    std::stable_sort(perm.begin(), perm.end(), [&](uint32_t a, uint32_t b) { return key[a] < key[b]; });
    const int nsb = (int)ceil_div(p, DEG_THREADS);
    perm.resize((size_t)nsb * DEG_THREADS, 0xFFFFFFFFu);
    std::vector<int4> box(nsb);
    int max_rows = 1;
    double evals = 0.0;
    for (int b = 0; b < nsb; ++b) {
        int rmin = 1 << 30, rmax = -1, cmin = 1 << 30, cmax = -1;
        for (int t = 0; t < DEG_THREADS; ++t) {
            const uint32_t i = perm[(size_t)b * DEG_THREADS + t];
            if (i >= p) continue;
            const int r = (int)(h_idx[i] / (unsigned)width), c = (int)(h_idx[i] % (unsigned)width);
            rmin = std::min(rmin, r); rmax = std::max(rmax, r); cmin = std::min(cmin, c); cmax = std::max(cmax, c);
        }
        box[b] = make_int4(rmin, rmax, cmin, cmax);
        const int rw0 = row0 + ((std::max(row0, rmin - radius) - row0) / DEG_ROWS) * DEG_ROWS;
        const int rw1 = std::min(row1, rmax + radius + 1);
        const int rows = std::max(0, std::min(row1, rw0 + (int)ceil_div(std::max(0, rw1 - rw0), DEG_ROWS) * DEG_ROWS) - rw0);
        const int cols = std::min(width, cmax + radius + 1) - (std::max(0, cmin - radius) & ~3);
        max_rows = std::max(max_rows, rows);
        evals += (double)DEG_THREADS * rows * cols;
    }
    if (evaluated) *evaluated = evals;
    const int nchunks = (int)ceil_div(max_rows, DEG_ROWS);
    DevBuf<uint32_t> d_perm;
    DevBuf<int4> d_box;
    DevBuf<double> partial;
    GLF_TRY(d_perm.alloc(ctx, perm.size()));
    GLF_TRY(d_box.alloc(ctx, box.size()));
    GLF_TRY(partial.alloc(ctx, (size_t)nchunks * p));
    GLF_HIP(ctx, hipMemcpyAsync(d_perm.p, perm.data(), sizeof(uint32_t) * perm.size(), hipMemcpyHostToDevice, ctx->stream));
    GLF_HIP(ctx, hipMemcpyAsync(d_box.p, box.data(), sizeof(int4) * box.size(), hipMemcpyHostToDevice, ctx->stream));
    DevBuf<float> d_ptab;
    std::vector<float> h_ptab;
    GLF_TRY(upload_photometric_table(ctx, coef, d_ptab, h_ptab));
    hipLaunchKernelGGL(k_degree_win, dim3(nsb, nchunks), dim3(DEGL_THREADS), 0, ctx->stream, d_img, width, row0, row1,
                       d_samples, p, d_perm.p, d_box.p, radius, coef.s_loc, d_ptab.p, partial.p);
    GLF_LAUNCH_CHECK(ctx);
    hipLaunchKernelGGL(k_reduce_partials, dim3((p + 255) / 256), dim3(256), 0, ctx->stream, partial.p, p, nchunks, d_degree);
    GLF_LAUNCH_CHECK(ctx);
    GLF_HIP(ctx, hipStreamSynchronize(ctx->stream)); // host vectors and DevBufs are released at scope exit
    return GLF_OK;
}

#include "grid_common.inc"
#include "affinity_grid.inc"

// D_A over the image rows [row0, row1): the grid-factored form when the samples are a tensor grid, else the
// direct sweep (windowed when exact zeros may be skipped). *evaluated = kernel entries represented.
int degree_rows_auto(glf_ctx *ctx, const uint8_t *d_img, int width, int height, int row0, int row1, const float4 *d_samples,
                     unsigned p, const unsigned *h_idx, KernelCoef coef, double *d_degree, int window, double *evaluated,
                     const uint32_t *d_idx, double *d_ysum, bool *have_ysum)
{
    if (have_ysum) *have_ysum = false;
    if (coef.kernel == GLF_KERNEL_NLM) { // patch distances: no factored form, its own sweep (nlm.hip)
        if (!d_idx) return set_error(ctx, GLF_ERR_INVALID, "NLM degree needs the device sample indices");
        if (evaluated) *evaluated = (double)p * (double)(row1 - row0) * (double)width;
        return nlm_degree_rows(ctx, d_img, width, height, row0, row1, d_idx, p, coef, d_degree);
    }
    const int rc = degree_rows_grid(ctx, d_img, width, height, row0, row1, d_samples, p, h_idx, coef, d_degree, window, evaluated, d_ysum);
    if (rc != GLF_ERR_UNSUPPORTED) {
        if (have_ysum) *have_ysum = rc == GLF_OK && d_ysum != nullptr; // (only the grid-factored degree has the value-weighted sums)
        return rc;
    }
    if (window && coef.s_loc > 0.f)
        return degree_rows_windowed(ctx, d_img, width, height, row0, row1, d_samples, p, h_idx, coef, d_degree, evaluated);
    if (evaluated) *evaluated = (double)p * (double)(row1 - row0) * (double)width;
    return degree_rows(ctx, d_img, width, height, row0, row1, d_samples, p, coef, d_degree);
}

// ---- full-matrix mode (-no_approx): z = clamp(y - L y), L = alpha (D - K) over ALL pixels ---------------
// hpc/image_processing.c:155-181 (EntireComputation): ComputeEntireAffinityMatrix (hpc/affinity.c:264-336),
// ComputeEntireLaplacianMatrix (hpc/laplacian.c:44-65), ComputeResultFromEntireLaplacian (hpc/display.c:128-149).
// The N x N matrices are never stored: (L y)_i = alpha (D_i y_i - (K y)_i) needs only the two row sums, which
// the degree sweep produces with every pixel as a "sample".
__global__ void k_pixel_table(const uint8_t *__restrict__ img, int width, int64_t N, float4 *__restrict__ table)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < N) table[i] = make_float4((float)(i / width), (float)(i % width), (float)img[i], 0.f);
}

__global__ __launch_bounds__(DEG_THREADS) void k_degree_ky(const uint8_t *__restrict__ img, int width, int height,
                                                            const float4 *__restrict__ table, unsigned n, float s_loc,
                                                            float s_val, double *__restrict__ part_d,
                                                            double *__restrict__ part_ky)
{
    __shared__ __attribute__((aligned(16))) float tile[DEG_ROWS * DEG_SEGW];
    const unsigned i = blockIdx.x * DEG_THREADS + threadIdx.x;
    const bool live = i < n;
    const float4 s = live ? table[i] : make_float4(0.f, 0.f, 0.f, 0.f);
    const int r_begin = (int)blockIdx.y * DEG_ROWS;
    const int r_end = min(r_begin + DEG_ROWS, height);
    double ky = 0.0;
    const double d = degree_chunk<true>(img, width, r_begin, r_end, 0, width, s, s_loc, s_val, tile, &ky);
    if (live) {
        part_d[(size_t)blockIdx.y * n + i] = d;
        part_ky[(size_t)blockIdx.y * n + i] = ky;
    }
}

__global__ void k_entire_result(const uint8_t *__restrict__ img, const double *__restrict__ D, const double *__restrict__ Ky,
                                int64_t N, double alpha, uint8_t *__restrict__ out, float *__restrict__ zf)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= N) return;
    const double y = (double)img[i];
    double z = y - alpha * (D[i] * y - Ky[i]); // MatAXPY(z, -1, Lapl_y), hpc/display.c:136
    if (zf) zf[i] = (float)z;
    z = z > 255.0 ? 255.0 : z;                 // AboveXSetY, :139
    z = z > 0.0 ? z : 0.0;                     // SetNegativesToZero, :141
    out[i] = (uint8_t)z;                       // (png_byte) cast, hpc/utils.c:525
}

int entire_computation(glf_ctx *ctx, const uint8_t *d_img, int width, int height, KernelCoef coef, uint8_t *d_out, float *d_zf,
                       double *alpha_out)
{
    const int64_t N = (int64_t)width * height;
    if (N > (int64_t)1 << 22) return set_error(ctx, GLF_ERR_UNSUPPORTED, "-no_approx is O(N^2): limited to 4 Mpixel images");
    const unsigned n = (unsigned)N;
    const int nsb = (int)ceil_div(n, DEG_THREADS), nchunks = (int)ceil_div(height, DEG_ROWS);
    if (nchunks > 65535) return set_error(ctx, GLF_ERR_UNSUPPORTED, "image too tall");
    DevBuf<float4> table;
    DevBuf<double> part_d, part_ky, D, Ky;
    GLF_TRY(table.alloc(ctx, n));
    GLF_TRY(part_d.alloc(ctx, (size_t)nchunks * n));
    GLF_TRY(part_ky.alloc(ctx, (size_t)nchunks * n));
    GLF_TRY(D.alloc(ctx, n));
    GLF_TRY(Ky.alloc(ctx, n));
    hipStream_t st = ctx->stream;
    hipLaunchKernelGGL(k_pixel_table, dim3((unsigned)ceil_div(N, 256)), dim3(256), 0, st, d_img, width, N, table.p);
    hipLaunchKernelGGL(k_degree_ky, dim3(nsb, nchunks), dim3(DEG_THREADS), 0, st, d_img, width, height, table.p, n, coef.s_loc,
                       coef.s_val, part_d.p, part_ky.p);
    hipLaunchKernelGGL(k_reduce_partials, dim3((n + 255) / 256), dim3(256), 0, st, part_d.p, n, nchunks, D.p);
    hipLaunchKernelGGL(k_reduce_partials, dim3((n + 255) / 256), dim3(256), 0, st, part_ky.p, n, nchunks, Ky.p);
    GLF_LAUNCH_CHECK(ctx);
    // alpha = 1 / mean(D), hpc/laplacian.c:57 + hpc/utils.c:378-388
    std::vector<double> h(n);
    GLF_HIP(ctx, hipMemcpyAsync(h.data(), D.p, sizeof(double) * n, hipMemcpyDeviceToHost, st));
    GLF_HIP(ctx, hipStreamSynchronize(st));
    double sum = 0.0;
    for (unsigned i = 0; i < n; ++i) sum += h[i];
    const double alpha = 1.0 / (sum / (double)n);
    if (alpha_out) *alpha_out = alpha;
    hipLaunchKernelGGL(k_entire_result, dim3((unsigned)ceil_div(N, 256)), dim3(256), 0, st, d_img, D.p, Ky.p, N, alpha, d_out,
                       d_zf);
    GLF_LAUNCH_CHECK(ctx);
    GLF_HIP(ctx, hipStreamSynchronize(st));
    return GLF_OK;
}

// ---- K_A / L_A (p x p) -----------------------------------------------------------------
//
// out[i][j] = scale * K(sample i, sample j); Laplacian form (hpc/laplacian.c:31-35):
// L_A = alpha * (diag(D) - K_A)  => off-diagonal -alpha*K, diagonal alpha*(D_i - K_ii).
// Write-bound: 4 p^2 bytes (29 GB at p = 85 264).

__global__ __launch_bounds__(256) void k_sample_matrix(const float4 *__restrict__ samples, unsigned p, float s_loc,
                                                        float s_val, float *__restrict__ out, int64_t ld,
                                                        int laplacian, double alpha,
                                                        const double *__restrict__ degree, unsigned col0,
                                                        unsigned ncols)
{
    const unsigned jl = blockIdx.x * 64 + (threadIdx.x & 63); // local column
    const unsigned i0 = (blockIdx.y * 4 + (threadIdx.x >> 6)) * 16;
    if (jl >= (unsigned)ld) return;
    if (jl >= ncols) { // padding columns up to the leading dimension: zeros (no separate memset of the 4 p ld bytes)
        for (unsigned ii = 0; ii < 16 && i0 + ii < p; ++ii) out[(size_t)(i0 + ii) * ld + jl] = 0.f;
        return;
    }
    const unsigned j = col0 + jl;
    const float4 sj = samples[j];
    const float fscale = laplacian ? (float)(-alpha) : 1.0f;
#pragma unroll 4
    for (unsigned ii = 0; ii < 16; ++ii) {
        const unsigned i = i0 + ii;
        if (i >= p) break;
        const float4 si = samples[i]; // wave-uniform -> scalar load
        float k = kernel_eval(si.x - sj.x, si.y - sj.y, si.z - sj.z, s_loc, s_val);
        float v = fscale * k;
        if (laplacian && i == j) v = (float)(alpha * (degree[i] - (double)k));
        out[(size_t)i * ld + jl] = v;
    }
}

int build_sample_matrix(glf_ctx *ctx, const float4 *d_samples, unsigned p, KernelCoef coef, float *d_out,
                        int64_t ld, bool laplacian, double alpha, const double *d_degree, unsigned col0, unsigned ncols,
                        const uint8_t *d_img, int width, int height, const uint32_t *d_idx)
{
    if (coef.kernel == GLF_KERNEL_NLM) {
        if (!d_img || !d_idx) return set_error(ctx, GLF_ERR_INVALID, "NLM sample matrix needs the image and the device sample indices");
        return nlm_sample_matrix(ctx, d_img, width, height, d_idx, p, coef, d_out, ld, laplacian, alpha, d_degree, col0, ncols);
    }
    if (ncols == 0) {
        col0 = 0;
        ncols = p;
    }
    if (col0 + ncols > p) return set_error(ctx, GLF_ERR_INVALID, "build_sample_matrix: column range");
    dim3 grid((unsigned)((ld + 63) / 64), (p + 63) / 64); // covers the padding columns too
    hipLaunchKernelGGL(k_sample_matrix, grid, dim3(256), 0, ctx->stream, d_samples, p, coef.s_loc, coef.s_val, d_out,
                       ld, laplacian ? 1 : 0, alpha, d_degree, col0, ncols);
    GLF_LAUNCH_CHECK(ctx);
    return GLF_OK;
}

__global__ __launch_bounds__(256) void k_laplacian_from_KA(const float *__restrict__ KA, int64_t ldk, unsigned p,
                                                            float *__restrict__ LA, int64_t ld, double alpha,
                                                            const double *__restrict__ degree)
{
    const unsigned j = blockIdx.x * 256 + threadIdx.x;
    if (j >= p) return;
    for (unsigned i = blockIdx.y; i < p; i += gridDim.y) {
        const float k = KA[(size_t)i * ldk + j];
        // MatAYPX(L_A, -1, D_A) then MatScale(alpha), hpc/laplacian.c:33-35
        LA[(size_t)i * ld + j] = (i == j) ? (float)(alpha * (degree[i] - (double)k)) : (float)(-alpha) * k;
    }
}

int laplacian_from_KA(glf_ctx *ctx, const float *d_KA, int64_t ldk, unsigned p, float *d_LA, int64_t ld,
                      double alpha, const double *d_degree)
{
    hipLaunchKernelGGL(k_laplacian_from_KA, dim3((p + 255) / 256, p < 16384 ? p : 16384), dim3(256), 0, ctx->stream, d_KA, ldk, p, d_LA,
                       ld, alpha, d_degree);
    GLF_LAUNCH_CHECK(ctx);
    return GLF_OK;
}

} // namespace glf
