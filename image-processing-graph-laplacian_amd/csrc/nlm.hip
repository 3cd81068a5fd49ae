// nlm.hip -- the non-local-means affinity (python/affinity_methods/NLM.py:9-34) behind the same stage
// API as the positional kernels (glf_options.kernel = GLF_KERNEL_NLM):
//
//   K(i, j) = exp(-|| G o (patch_i - patch_j) ||^2 / h^2),  7 x 7 patches of the symmetrically padded image (np.pad
//   'symmetric', :16), G the 7 x 7 Gaussian mask of sigma 1.2 normalised to sum 1 (:17-19) multiplying the patch values
//   (:21-22, :29), h = h_val (the PoC fixes h = 3, :12). Pixel indices are raster indices throughout (the PoC's
//   transposed column layout is documented in oracle/glf_oracle.c).
//
// The kernel does not factor over rows, columns and values, so the grid-factored forms do not apply: every entry is a
// 49-term weighted patch distance, generated on the vector pipe from patch features f[k] = G[k] * padded(r + a, c + b)
// -- the "LDS-staged sample PATCHES" of north_star, literally: one side of each pair lives in 49 VGPRs of its lane, the
// other is staged as a tile of feature rows in LDS and read back as wave-wide broadcasts. The features are computed from
// the image on the fly (49 byte loads per pixel), never stored. Three kernels mirror the positional ones:
//   k_nlm_degree     D[i] = sum over pixels K(sample i, pixel)      lane = sample, pixel tiles in LDS   (k_degree)
//   k_nlm_matrix     K_A / L_A                                      lane = column sample                (k_sample_matrix)
//   k_nlm_nystroem   Phi = K_B^T Psi, f32 MFMA, K generated in the A-fragment layout (lane = pixel, sample tiles +
//                    Psi tiles in LDS) with the epilogue of k_nystroem (Permutation folded in, Phi^T y partials)
// Cost: ~100 VALU operations per kernel entry (1.8 s per pass over p x N entries at 4096^2 -- a next-row feature, not the
// headline path; an MFMA formulation of the patch distances through || a ||^2 + || b ||^2 - 2 a.b is the obvious next
// step and loses ~1e-5 relative accuracy to cancellation, which is why the first version keeps the differences).
#include "glf_internal.hpp"

#include <cmath>
#include <vector>

namespace glf {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int NLM_R = 3, NLM_K = 49, NLM_F = 52; // patch radius, features, feature row stride in LDS (13 float4)

struct NlmMask {
    float g[NLM_K];
};

// matlab_style_gauss2D((7, 7), 1.2) normalised to sum 1 (python/utils.py:17-31, NLM.py:17-19), rounded to f32 at the end
static NlmMask nlm_mask()
{
    double G[NLM_K], sum = 0.0, mx = 0.0;
    for (int a = -NLM_R; a <= NLM_R; ++a)
        for (int b = -NLM_R; b <= NLM_R; ++b) {
            const double g = std::exp(-(double)(a * a + b * b) / (2. * 1.2 * 1.2));
            G[(a + NLM_R) * 7 + (b + NLM_R)] = g;
            mx = std::max(mx, g);
        }
    for (int k = 0; k < NLM_K; ++k) {
        if (G[k] < 2.220446049250313e-16 * mx) G[k] = 0.0;
        sum += G[k];
    }
    for (int k = 0; k < NLM_K; ++k) G[k] /= sum;
    sum = 0.0;
    for (int k = 0; k < NLM_K; ++k) sum += G[k];
    NlmMask m;
    for (int k = 0; k < NLM_K; ++k) m.g[k] = (float)(G[k] / sum);
    return m;
}

// np.pad 'symmetric', one reflection: exact for n >= NLM_R = 3 (smaller images are refused at the entry points, pipeline.hip)
__device__ __forceinline__ int nlm_reflect(int i, int n) { return i < 0 ? -i - 1 : (i >= n ? 2 * n - i - 1 : i); }

// the 49 weighted patch values of pixel (r, c)
__device__ __forceinline__ void nlm_patch(const uint8_t *__restrict__ img, int width, int height, int r, int c, const NlmMask &G,
                                          float (&f)[NLM_K])
{
#pragma unroll
    for (int a = 0; a < 7; ++a) {
        const uint8_t *row = img + (size_t)nlm_reflect(r + a - NLM_R, height) * width;
#pragma unroll
        for (int b = 0; b < 7; ++b) f[a * 7 + b] = G.g[a * 7 + b] * (float)row[nlm_reflect(c + b - NLM_R, width)];
    }
}

// feature rows of `count` pixels (pixel index px0 + t, or list[t]) into LDS [count][NLM_F]; rows past `valid` are zero
__device__ __forceinline__ void nlm_stage(const uint8_t *__restrict__ img, int width, int height, const NlmMask &G, int64_t px0,
                                          const uint32_t *__restrict__ list, int count, int valid, float *tile, int nthreads)
{
    for (int e = threadIdx.x; e < count * NLM_F; e += nthreads) {
        const int t = e / NLM_F, k = e - t * NLM_F;
        float v = 0.f;
        if (t < valid && k < NLM_K) {
            const int64_t px = list ? (int64_t)list[t] : px0 + t;
            const int r = (int)(px / width), c = (int)(px % width);
            const int a = k / 7, b = k - a * 7;
            v = G.g[k] * (float)img[(size_t)nlm_reflect(r + a - NLM_R, height) * width + nlm_reflect(c + b - NLM_R, width)];
        }
        tile[e] = v;
    }
}

// || f - row ||^2 with `row` a feature row in LDS (wave-wide broadcast reads of 13 float4). Two terms per instruction
// (v_pk_add_f32 / v_pk_fma_f32 on register pairs, two accumulators: even and odd terms): 52 vector instructions per entry
// instead of 98. It bought 3 %: v_pk_add_f32 / v_pk_fma_f32 issue at half the rate of their scalar forms on this part, so the
// passes stay bound by the vector pipe at ~100 issue slots per entry (5.6e9 entries x 100 x 4 cycles / 1024 SIMDs = 14 ms at
// 1024^2, what the degree pass takes); sharing a feature row between two patches per lane (half the LDS reads) changed nothing
// either. Halving the work takes the dot-product form || a ||^2 + || b ||^2 - 2 a.b, whose f32 cancellation (~5e-5 absolute
// in d^2, 6e-6 relative in K) sits at the degree test's 3e-6 bound: not taken.
typedef float nlm_f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ float nlm_dist(const float (&f)[NLM_K], const float *row)
{
    nlm_f32x2 d = {0.f, 0.f};
#pragma unroll
    for (int q = 0; q < 12; ++q) {
        const float4 x = *reinterpret_cast<const float4 *>(row + 4 * q);
        nlm_f32x2 a0 = {f[4 * q], f[4 * q + 1]}, a1 = {f[4 * q + 2], f[4 * q + 3]};
        const nlm_f32x2 x0 = {x.x, x.y}, x1 = {x.z, x.w};
        a0 -= x0;
        a1 -= x1;
        d = __builtin_elementwise_fma(a0, a0, d);
        d = __builtin_elementwise_fma(a1, a1, d);
    }
    const float e = f[48] - row[48];
    return fmaf(e, e, d[0] + d[1]);
}

// ---- degree -------------------------------------------------------------------------------------------------------------
constexpr int NLM_TILE = 64;      // pixels per staged tile
constexpr int NLM_CHUNK = 8192;   // pixels per workgroup (one f64 partial per sample and chunk)

__global__ __launch_bounds__(256) void k_nlm_degree(const uint8_t *__restrict__ img, int width, int height, int64_t pix0, int64_t pix1,
                                                     const uint32_t *__restrict__ idx, unsigned p, float s_val, NlmMask G,
                                                     double *__restrict__ partial)
{
    __shared__ __attribute__((aligned(16))) float tile[NLM_TILE * NLM_F];
    const unsigned i = blockIdx.x * 256 + threadIdx.x;
    const bool live = i < p;
    float f[NLM_K];
    {
        const uint32_t px = live ? idx[i] : 0u;
        nlm_patch(img, width, height, (int)(px / (uint32_t)width), (int)(px % (uint32_t)width), G, f);
    }
    const int64_t c0 = pix0 + (int64_t)blockIdx.y * NLM_CHUNK, c1 = min(c0 + NLM_CHUNK, pix1);
    double total = 0.0;
    for (int64_t t0 = c0; t0 < c1; t0 += NLM_TILE) {
        const int valid = (int)min((int64_t)NLM_TILE, c1 - t0);
        __syncthreads();
        nlm_stage(img, width, height, G, t0, nullptr, NLM_TILE, valid, tile, 256);
        __syncthreads();
        float acc = 0.f;
        for (int t = 0; t < valid; ++t) acc += __builtin_amdgcn_exp2f(-(nlm_dist(f, tile + t * NLM_F) * s_val));
        total += (double)acc;
    }
    if (live) partial[(size_t)blockIdx.y * p + i] = total;
}

__global__ void k_nlm_reduce(const double *__restrict__ partial, unsigned p, int nchunks, double *__restrict__ out)
{
    const unsigned i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= p) return;
    double s = 0.0;
    for (int k = 0; k < nchunks; ++k) s += partial[(size_t)k * p + i];
    out[i] = s;
}

int nlm_degree_rows(glf_ctx *ctx, const uint8_t *d_img, int width, int height, int row0, int row1, const uint32_t *d_idx, unsigned p,
                    KernelCoef coef, double *d_degree)
{
    if (row0 < 0 || row1 > height || row0 > row1) return set_error(ctx, GLF_ERR_INVALID, "bad row range");
    if (row0 == row1) {
        GLF_HIP(ctx, hipMemsetAsync(d_degree, 0, sizeof(double) * p, ctx->stream));
        return GLF_OK;
    }
    const int64_t pix0 = (int64_t)row0 * width, pix1 = (int64_t)row1 * width;
    const int nchunks = (int)ceil_div(pix1 - pix0, NLM_CHUNK);
    if (nchunks > 65535) return set_error(ctx, GLF_ERR_UNSUPPORTED, "image too large for one NLM degree launch");
    DevBuf<double> partial;
    GLF_TRY(partial.alloc(ctx, (size_t)nchunks * p));
    hipLaunchKernelGGL(k_nlm_degree, dim3((unsigned)ceil_div(p, 256), nchunks), dim3(256), 0, ctx->stream, d_img, width, height, pix0, pix1,
                       d_idx, p, coef.s_val, nlm_mask(), partial.p);
    hipLaunchKernelGGL(k_nlm_reduce, dim3((p + 255) / 256), dim3(256), 0, ctx->stream, partial.p, p, nchunks, d_degree);
    GLF_LAUNCH_CHECK(ctx);
    GLF_HIP(ctx, hipStreamSynchronize(ctx->stream)); // partial is released at scope exit
    return GLF_OK;
}

// ---- K_A / L_A ---------------------------------------------------------------------------------------------------------------
// out[i][jl] = K(sample i, sample col0 + jl) (scale 1) or the Laplacian form alpha (D_i delta_ij - K); lane = column sample,
// 64 row samples staged per tile. Same contract as build_sample_matrix (padding columns up to ld are zeroed here).
__global__ __launch_bounds__(256) void k_nlm_matrix(const uint8_t *__restrict__ img, int width, int height,
                                                     const uint32_t *__restrict__ idx, unsigned p, float s_val, NlmMask G,
                                                     float *__restrict__ out, int64_t ld, int laplacian, double alpha,
                                                     const double *__restrict__ degree, unsigned col0, unsigned ncols)
{
    __shared__ __attribute__((aligned(16))) float tile[NLM_TILE * NLM_F];
    const unsigned jl = blockIdx.x * 256 + threadIdx.x;
    const bool col_ok = jl < ncols;
    const unsigned j = col0 + (col_ok ? jl : 0u);
    float f[NLM_K];
    {
        const uint32_t px = idx[j < p ? j : 0];
        nlm_patch(img, width, height, (int)(px / (uint32_t)width), (int)(px % (uint32_t)width), G, f);
    }
    const unsigned i0 = blockIdx.y * NLM_TILE;
    const int valid = (int)min((unsigned)NLM_TILE, p - i0);
    nlm_stage(img, width, height, G, 0, idx + i0, NLM_TILE, valid, tile, 256);
    __syncthreads();
    if (jl >= (unsigned)ld) return;
    const float fscale = laplacian ? (float)(-alpha) : 1.0f;
    for (int t = 0; t < valid; ++t) {
        const unsigned i = i0 + t;
        float v = 0.f;
        if (col_ok) {
            const float k = __builtin_amdgcn_exp2f(-(nlm_dist(f, tile + t * NLM_F) * s_val));
            v = fscale * k;
            if (laplacian && i == j) v = (float)(alpha * (degree[i] - (double)k));
        }
        out[(size_t)i * ld + jl] = v;
    }
}

int nlm_sample_matrix(glf_ctx *ctx, const uint8_t *d_img, int width, int height, const uint32_t *d_idx, unsigned p, KernelCoef coef,
                      float *d_out, int64_t ld, bool laplacian, double alpha, const double *d_degree, unsigned col0, unsigned ncols)
{
    if (ncols == 0) {
        col0 = 0;
        ncols = p;
    }
    if (col0 + ncols > p) return set_error(ctx, GLF_ERR_INVALID, "nlm_sample_matrix: column range");
    hipLaunchKernelGGL(k_nlm_matrix, dim3((unsigned)ceil_div(ld, 256), (unsigned)ceil_div(p, NLM_TILE)), dim3(256), 0, ctx->stream, d_img,
                       width, height, d_idx, p, coef.s_val, nlm_mask(), d_out, ld, laplacian ? 1 : 0, alpha, d_degree, col0, ncols);
    GLF_LAUNCH_CHECK(ctx);
    return GLF_OK;
}

// ---- Nystroem contraction ------------------------------------------------------------------------------------------------------
// Phi[pix][j] = sum_i K(sample i, pix) Psi[i][j] as in k_nystroem (nystroem.hip): v_mfma_f32_32x32x2_f32, lane l supplies
// A[pixel l & 31][sample of its half-chunk]; the A operand is the NLM entry generated from the lane's 49 pixel features and the
// sample's feature row in LDS. One 32-pixel block per wave, 4 waves, 64 samples per LDS chunk.
constexpr int NLM_KC = 64;

__device__ __forceinline__ unsigned nlm_samples_before(const uint32_t *__restrict__ idx, unsigned p, uint32_t px)
{
    unsigned lo = 0, hi = p;
    while (lo < hi) {
        const unsigned mid = (lo + hi) >> 1;
        if (idx[mid] < px) lo = mid + 1;
        else hi = mid;
    }
    return lo;
}

template <int MB>
__global__ __launch_bounds__(256) void k_nlm_nystroem(const uint8_t *__restrict__ img, int width, int height, int64_t pix0, int64_t pix1,
                                                       const uint32_t *__restrict__ idx, unsigned p, float s_val, NlmMask G,
                                                       const float *__restrict__ psi, float *__restrict__ phi, int raster,
                                                       const uint8_t *__restrict__ mask, double *__restrict__ cpartial)
{
    constexpr int LD = MB * 32;
    __shared__ __attribute__((aligned(16))) float feat[NLM_KC * NLM_F];
    __shared__ __attribute__((aligned(16))) float pst[NLM_KC * LD];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, half = lane >> 5, l31 = lane & 31;
    const int64_t wbase = pix0 + ((int64_t)blockIdx.x * 4 + wave) * 32;
    float f[NLM_K];
    {
        int64_t px = wbase + l31;
        if (px >= pix1) px = pix1 - 1; // clamp loads; stores are guarded
        nlm_patch(img, width, height, (int)(px / width), (int)(px % width), G, f);
    }
    f32x16 acc[MB];
#pragma unroll
    for (int j = 0; j < MB; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
    const int nchunks = (int)((p + NLM_KC - 1) / NLM_KC);
    for (int ch = 0; ch < nchunks; ++ch) {
        const unsigned s0 = (unsigned)ch * NLM_KC;
        const int valid = (int)min((unsigned)NLM_KC, p - s0);
        __syncthreads(); // the previous chunk is consumed
        nlm_stage(img, width, height, G, 0, idx + s0, NLM_KC, valid, feat, 256);
        for (int e = threadIdx.x * 4; e < NLM_KC * LD; e += 256 * 4) {
            const unsigned s = s0 + e / LD;
            const float4 v = (s < p) ? *reinterpret_cast<const float4 *>(&psi[(size_t)s0 * LD + e]) : make_float4(0.f, 0.f, 0.f, 0.f);
            *reinterpret_cast<float4 *>(pst + e) = v;
        }
        __syncthreads();
        const float *fb = feat + (half * (NLM_KC / 2)) * NLM_F;
        const float *psb = pst + (half * (NLM_KC / 2)) * LD + l31;
        for (int kk = 0; kk < NLM_KC / 2; ++kk) {
            // (samples past p: zero feature row, zero Psi row -> K * 0 = 0)
            const float a = __builtin_amdgcn_exp2f(-(nlm_dist(f, fb + kk * NLM_F) * s_val));
#pragma unroll
            for (int j = 0; j < MB; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, psb[kk * LD + 32 * j], acc[j], 0, 0, 0);
        }
    }
    // epilogue as k_nystroem: Phi rows (raster or sample-first), c_j += Phi[pix][j] * y[pix] over non-sample pixels
    float csum[MB];
#pragma unroll
    for (int j = 0; j < MB; ++j) csum[j] = 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int64_t px = wbase + (r & 3) + 8 * (r >> 2) + 4 * half;
        if (px >= pix1) continue;
        const bool is_sample = mask[px] != 0;
        int64_t dst;
        if (raster) dst = px;
        else {
            if (is_sample) continue; // sample rows come from Phi_A (hpc/nystroem.c:25-34)
            dst = (int64_t)p + px - (int64_t)nlm_samples_before(idx, p, (uint32_t)px);
        }
        const float y = is_sample ? 0.f : (float)img[px];
#pragma unroll
        for (int j = 0; j < MB; ++j) {
            const float v = acc[j][r];
            phi[(size_t)dst * LD + 32 * j + l31] = v;
            csum[j] = fmaf(v, y, csum[j]);
        }
    }
    if (cpartial) {
        __syncthreads();
        float *red = pst; // [4 waves][LD]
#pragma unroll
        for (int j = 0; j < MB; ++j) {
            const float v = csum[j] + __shfl_xor(csum[j], 32, 64);
            if (half == 0) red[wave * LD + 32 * j + l31] = v;
        }
        __syncthreads();
        if (threadIdx.x < LD)
            cpartial[(size_t)blockIdx.x * LD + threadIdx.x] = ((double)red[threadIdx.x] + (double)red[LD + threadIdx.x]) +
                                                              ((double)red[2 * LD + threadIdx.x] + (double)red[3 * LD + threadIdx.x]);
    }
}

__global__ __launch_bounds__(256) void k_nlm_rows_sum(const double *__restrict__ in, int64_t nrows, unsigned ld, double *__restrict__ out)
{
    // out[c] += sum_rows in[row][c]: one workgroup, column = t % ld, fixed order
    __shared__ double sh[256];
    const int col = threadIdx.x % ld, rl = threadIdx.x / ld, nrl = 256 / ld;
    double s = 0.0;
    for (int64_t r = rl; r < nrows; r += nrl) s += in[(size_t)r * ld + col];
    sh[threadIdx.x] = s;
    __syncthreads();
    if (threadIdx.x < ld) {
        double t = out[col];
        for (int r = 0; r < nrl; ++r) t += sh[r * ld + col];
        out[col] = t;
    }
}

int nlm_nystroem(glf_ctx *ctx, const uint8_t *d_img, int width, int height, int64_t pix0, int64_t pix1, const uint8_t *d_mask,
                 const uint32_t *d_idx, unsigned p, KernelCoef coef, const float *d_psi, unsigned ld, float *d_phi, int raster,
                 double *d_c, float *kernel_ms)
{
    const int64_t nwg = ceil_div(pix1 - pix0, 128);
    DevBuf<double> cpart;
    if (d_c) GLF_TRY(cpart.alloc(ctx, (size_t)nwg * ld));
    hipStream_t st = ctx->stream;
    if (kernel_ms) GLF_HIP(ctx, hipEventRecord(ctx->ev[6], st));
    const NlmMask G = nlm_mask();
#define GLF_NLM_GO(MB_)                                                                                                            \
    hipLaunchKernelGGL((k_nlm_nystroem<MB_>), dim3((unsigned)nwg), dim3(256), 0, st, d_img, width, height, pix0, pix1, d_idx, p,   \
                       coef.s_val, G, d_psi, d_phi, raster, d_mask, d_c ? cpart.p : nullptr)
    switch (ld) {
    case 32: GLF_NLM_GO(1); break;
    case 64: GLF_NLM_GO(2); break;
    case 128: GLF_NLM_GO(4); break;
    case 256: GLF_NLM_GO(8); break;
    default: return set_error(ctx, GLF_ERR_INVALID, "nlm_nystroem: ld=%u", ld);
    }
#undef GLF_NLM_GO
    GLF_LAUNCH_CHECK(ctx);
    if (kernel_ms) GLF_HIP(ctx, hipEventRecord(ctx->ev[7], st));
    if (d_c) {
        hipLaunchKernelGGL(k_nlm_rows_sum, dim3(1), dim3(256), 0, st, cpart.p, nwg, ld, d_c);
        GLF_LAUNCH_CHECK(ctx);
    }
    GLF_HIP(ctx, hipStreamSynchronize(st));
    if (kernel_ms) GLF_HIP(ctx, hipEventElapsedTime(kernel_ms, ctx->ev[6], ctx->ev[7]));
    return GLF_OK;
}

} // namespace glf
