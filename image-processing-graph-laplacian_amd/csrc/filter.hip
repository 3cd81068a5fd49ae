// filter.hip -- spectral filter reconstruction (hpc/display.c:58-83):
//   right = Phi^T y                (m reductions over all pixels)
//   z     = y + gain * Phi (f(Pi) right);  z > 255 -> 255;  (png_byte) cast
// Both kernels stream Phi (N x ld floats) once: HBM-bandwidth bound,
// 4 N ld + N bytes each.
#include "glf_internal.hpp"

#include <algorithm>

namespace glf {

// partial[blk][j] = sum_{pix in blk} Phi[pix][j] * y[pix]
__global__ __launch_bounds__(256) void k_phi_t_y(const float *__restrict__ phi, const uint8_t *__restrict__ img, int64_t pix0,
                                                  int64_t pix1, unsigned ld, double *__restrict__ partial)
{
    __shared__ double sh[256];
    const int col = threadIdx.x % ld, rl = threadIdx.x / ld, nrl = 256 / ld;
    const int64_t base = pix0 + (int64_t)blockIdx.x * 1024;
    float acc = 0.f;
    double s = 0.0;
    int cnt = 0;
    for (int64_t r = rl; r < 1024; r += nrl) {
        const int64_t px = base + r;
        if (px >= pix1) break;
        acc = fmaf(phi[(size_t)px * ld + col], (float)img[px], acc);
        if (++cnt == 32) { // bounded f32 chains, f64 across them
            s += (double)acc;
            acc = 0.f;
            cnt = 0;
        }
    }
    s += (double)acc;
    sh[threadIdx.x] = s;
    __syncthreads();
    if (threadIdx.x < ld) {
        double t = 0.0;
        for (int r = 0; r < nrl; ++r) t += sh[r * ld + col];
        partial[(size_t)blockIdx.x * ld + col] = t;
    }
}

__global__ void k_cols_sum(const double *__restrict__ in, int nrows, unsigned ld, double *__restrict__ out)
{
    // one workgroup per column chunk: 256 threads stride the rows, tree-reduce in LDS (fixed order)
    __shared__ double sh[256];
    const unsigned c = blockIdx.x;
    double s = 0.0;
    for (int r = threadIdx.x; r < nrows; r += 256) s += in[(size_t)r * ld + c];
    sh[threadIdx.x] = s;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if ((int)threadIdx.x < o) sh[threadIdx.x] += sh[threadIdx.x + o];
        __syncthreads();
    }
    if (threadIdx.x == 0) out[c] = sh[0];
}

int phi_t_y(glf_ctx *ctx, const float *d_phi, const uint8_t *d_img, int64_t pix0, int64_t pix1, unsigned /*m*/, unsigned ld,
            double *d_c)
{
    if (!valid_ld(ld) || pix0 > pix1) return set_error(ctx, GLF_ERR_INVALID, "phi_t_y: ld=%u", ld);
    if (pix0 == pix1) {
        GLF_HIP(ctx, hipMemsetAsync(d_c, 0, sizeof(double) * ld, ctx->stream));
        return GLF_OK;
    }
    const int nblk = (int)ceil_div(pix1 - pix0, 1024);
    DevBuf<double> part;
    GLF_TRY(part.alloc(ctx, (size_t)nblk * ld));
    hipLaunchKernelGGL(k_phi_t_y, dim3(nblk), dim3(256), 0, ctx->stream, d_phi, d_img, pix0, pix1, ld, part.p);
    hipLaunchKernelGGL(k_cols_sum, dim3(ld), dim3(256), 0, ctx->stream, part.p, nblk, ld, d_c);
    GLF_LAUNCH_CHECK(ctx);
    GLF_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return GLF_OK;
}

// ---- c = Phi^T y without Phi (band form with the filter in the Nystroem kernel's epilogue) ---------------------------------
// Phi's rows are Phi_A at the sample pixels and sum_s K(px, s) Psi[s] elsewhere, so
//   c[n] = sum_s Phi_A[s][n] y_s + sum_s Psi[s][n] u_s,   u_s = sum over the NON-sample pixels of K(px, s) y[px]
// and u = (the degree stage's value-weighted sums over all pixels) - K_A y_A. f64 throughout; 128 samples per workgroup.
__global__ __launch_bounds__(256) void k_c_from_ysum(const float *__restrict__ psi, const float *__restrict__ phiA,
                                                      const double *__restrict__ ysum, const float *__restrict__ t, unsigned t_ld,
                                                      const float4 *__restrict__ samples, unsigned p, unsigned ld,
                                                      double *__restrict__ partial)
{
    __shared__ double sh[256];
    const int col = threadIdx.x % ld, rl = threadIdx.x / ld, nrl = 256 / ld;
    double acc = 0.0;
    const unsigned base = blockIdx.x * 128u;
    for (unsigned r = rl; r < 128u; r += nrl) {
        const unsigned s = base + r;
        if (s >= p) break;
        const double u = ysum[s] - (double)t[(size_t)s * t_ld];
        acc = fma((double)psi[(size_t)s * ld + col], u, acc);
        acc = fma((double)phiA[(size_t)s * ld + col], (double)samples[s].z, acc);
    }
    sh[threadIdx.x] = acc;
    __syncthreads();
    if (threadIdx.x < ld) {
        double tot = 0.0;
        for (int r = 0; r < nrl; ++r) tot += sh[r * ld + col];
        partial[(size_t)blockIdx.x * ld + col] = tot;
    }
}

int c_from_ysum(glf_ctx *ctx, const float *d_psi, const float *d_phiA, const double *d_ysum, const float *d_t, unsigned t_ld,
                const float4 *d_samples, unsigned p, unsigned ld, double *d_c)
{
    if (!valid_ld(ld)) return set_error(ctx, GLF_ERR_INVALID, "c_from_ysum: ld=%u", ld);
    const int nblk = (int)ceil_div(p, 128u);
    DevBuf<double> part;
    GLF_TRY(part.alloc(ctx, (size_t)nblk * ld));
    hipLaunchKernelGGL(k_c_from_ysum, dim3(nblk), dim3(256), 0, ctx->stream, d_psi, d_phiA, d_ysum, d_t, t_ld, d_samples, p, ld, part.p);
    hipLaunchKernelGGL(k_cols_sum, dim3(ld), dim3(256), 0, ctx->stream, part.p, nblk, ld, d_c);
    GLF_LAUNCH_CHECK(ctx);
    GLF_HIP(ctx, hipStreamSynchronize(ctx->stream)); // (part goes out of scope)
    return GLF_OK;
}

// partial[blk][i][j] = sum over the pixels of block blk of Phi[px][i] Phi[px][j] (blocks stride the range): the Gram matrix of
// the extended eigenvectors, which are not orthonormal -- what sits between the factors of the PoC's sharpening filter
// (python/image_processing.py:231-235). f32 products, f64 across chains of 32.
__global__ __launch_bounds__(256) void k_phi_gram(const float *__restrict__ phi, int64_t pix0, int64_t pix1, unsigned ld,
                                                   double *__restrict__ partial)
{
    extern __shared__ float tile[]; // [64 pixels][ld]
    const unsigned npair = ld * ld, e0 = blockIdx.y * 4096u; // this workgroup's 4096 pairs (i, j): 16 per thread
    double acc[16];
#pragma unroll
    for (int q = 0; q < 16; ++q) acc[q] = 0.0;
    for (int64_t base = pix0 + (int64_t)blockIdx.x * 64; base < pix1; base += (int64_t)gridDim.x * 64) {
        const int npx = (int)((pix1 - base) < 64 ? (pix1 - base) : 64);
        __syncthreads();
        for (unsigned e = threadIdx.x; e < 64 * ld; e += 256) tile[e] = e < (unsigned)npx * ld ? phi[(size_t)base * ld + e] : 0.f;
        __syncthreads();
#pragma unroll
        for (int q = 0; q < 16; ++q) {
            const unsigned e = e0 + q * 256 + threadIdx.x;
            if (e < npair) {
                const unsigned i = e / ld, j = e % ld;
                float a = 0.f;
                for (int r = 0; r < 64; ++r) a = fmaf(tile[r * ld + i], tile[r * ld + j], a);
                acc[q] += (double)a;
            }
        }
    }
#pragma unroll
    for (int q = 0; q < 16; ++q) {
        const unsigned e = e0 + q * 256 + threadIdx.x;
        if (e < npair) partial[(size_t)blockIdx.x * npair + e] = acc[q];
    }
}

// d_G[ld][ld] (f64) = Phi^T Phi over the pixels [pix0, pix1) of this rank
int phi_gram(glf_ctx *ctx, const float *d_phi, int64_t pix0, int64_t pix1, unsigned ld, double *d_G)
{
    if (!valid_ld(ld) || pix0 > pix1) return set_error(ctx, GLF_ERR_INVALID, "phi_gram: ld=%u", ld);
    const int nblk = (int)std::max<int64_t>(1, std::min<int64_t>(512, ceil_div(pix1 - pix0, 64)));
    DevBuf<double> part;
    GLF_TRY(part.alloc(ctx, (size_t)nblk * ld * ld));
    hipLaunchKernelGGL(k_phi_gram, dim3(nblk, (unsigned)ceil_div((int64_t)ld * ld, 4096)), dim3(256), 64 * ld * sizeof(float), ctx->stream, d_phi,
                       pix0, pix1, ld, part.p);
    hipLaunchKernelGGL(k_cols_sum, dim3(ld * ld), dim3(256), 0, ctx->stream, part.p, nblk, ld * ld, d_G);
    GLF_LAUNCH_CHECK(ctx);
    GLF_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return GLF_OK;
}

// AboveXSetY(z, 255, 255) (hpc/display.c:76), negative -> 0 (survey quirk Q4) and the truncating (png_byte) cast
// (hpc/utils.c:525) of the reference's fp64 z = y + c, evaluated WITHOUT rounding the sum to f32 first: y is an integer, so
// trunc(y + c) = y + floor(c) wherever y + c >= 0. At 4096^2 |c| ~ 1e-3 grey levels: the f32 sum rounds y - 2e-6 up to y
// and 4 % of the pixels then miss the reference's y - 1.
__device__ __forceinline__ uint8_t filter_output(int y, float c)
{
    int zi = y + (int)floorf(fminf(fmaxf(c, -1.0e6f), 1.0e6f));
    zi = zi > 255 ? 255 : zi;
    zi = (zi < 0 || !(c == c)) ? 0 : zi; // (NaN -> 0)
    return (uint8_t)zi;
}

// z[pix] = y + gain * sum_j Phi[pix][j] w[j]; LD/4 lanes per pixel, float4 each.
template <int LD>
__global__ __launch_bounds__(256) void k_apply_filter(const uint8_t *__restrict__ img, const float *__restrict__ phi,
                                                       int64_t pix0, int64_t pix1, const float *__restrict__ w, float gain, float ysub,
                                                       uint8_t *__restrict__ out, float *__restrict__ zf, float *__restrict__ corr)
{
    constexpr int LPP = LD / 4;       // lanes per pixel (8 .. 64)
    constexpr int PPB = 256 / LPP;    // pixels per block pass
    const int q = threadIdx.x % LPP, pl = threadIdx.x / LPP;
    const float4 wq = reinterpret_cast<const float4 *>(w)[q];
    for (int64_t px = pix0 + (int64_t)blockIdx.x * PPB + pl; px < pix1; px += (int64_t)gridDim.x * PPB) {
        const float4 f = reinterpret_cast<const float4 *>(phi + (size_t)px * LD)[q];
        float s = f.x * wq.x + f.y * wq.y + f.z * wq.z + f.w * wq.w;
#pragma unroll
        for (int o = LPP / 2; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
        if (q == 0) {
            const int y = (int)img[px];
            // the correction 3.0 * Lapl_y, hpc/display.c:64-73; ysub = 1 for the filters without a y term (z = Phi f Phi^T y):
            // y is an integer, so y + floor(c - y) is the truncation of their z
            const float c = gain * s - ysub * (float)y;
            if (zf) zf[px] = (float)y + c;           // MatAXPY(z, 3.0, Lapl_y) as a float (resolves c to ulp(z) only)
            if (corr) corr[px - pix0] = c;
            out[px] = filter_output(y, c);
        }
    }
}

// the same for n sample pixels, each from its row of Phi_A: px = idx[i] (k_apply_filter's arithmetic)
template <int LD>
__global__ __launch_bounds__(256) void k_filter_sample_rows(const uint8_t *__restrict__ img, const float *__restrict__ phiA, unsigned n,
                                                             const uint32_t *__restrict__ idx, const float *__restrict__ w, float gain,
                                                             float ysub, uint8_t *__restrict__ out, float *__restrict__ zf,
                                                             float *__restrict__ corr, int64_t pix0)
{
    constexpr int LPP = LD / 4, PPB = 256 / LPP;
    const int q = threadIdx.x % LPP, pl = threadIdx.x / LPP;
    const float4 wq = reinterpret_cast<const float4 *>(w)[q];
    const unsigned i = blockIdx.x * PPB + pl;
    const bool live = i < n;
    const float4 f = live ? reinterpret_cast<const float4 *>(phiA + (size_t)i * LD)[q] : make_float4(0.f, 0.f, 0.f, 0.f);
    float s = f.x * wq.x + f.y * wq.y + f.z * wq.z + f.w * wq.w;
#pragma unroll
    for (int o = LPP / 2; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
    if (q == 0 && live) {
        const int64_t px = (int64_t)idx[i];
        const int y = (int)img[px];
        const float c = gain * s - ysub * (float)y;
        if (zf) zf[px] = (float)y + c;
        if (corr) corr[px - pix0] = c;
        out[px] = filter_output(y, c);
    }
}

int filter_sample_rows(glf_ctx *ctx, const float *d_phiA, unsigned n, unsigned ld, const uint32_t *d_idx, const uint8_t *d_img,
                       const float *d_w, float gain, float ysub, uint8_t *d_out, float *d_zf, float *d_corr, int64_t pix0)
{
    if (n == 0) return GLF_OK;
    if (ld != 32 && ld != 64) return set_error(ctx, GLF_ERR_INVALID, "filter_sample_rows: ld=%u", ld);
    const unsigned ppb = 256 / (ld / 4);
    const dim3 grid((unsigned)ceil_div(n, ppb)), block(256);
    if (ld == 32) hipLaunchKernelGGL(k_filter_sample_rows<32>, grid, block, 0, ctx->stream, d_img, d_phiA, n, d_idx, d_w, gain, ysub, d_out, d_zf, d_corr, pix0);
    else hipLaunchKernelGGL(k_filter_sample_rows<64>, grid, block, 0, ctx->stream, d_img, d_phiA, n, d_idx, d_w, gain, ysub, d_out, d_zf, d_corr, pix0);
    GLF_LAUNCH_CHECK(ctx);
    return GLF_OK;
}

int apply_filter(glf_ctx *ctx, const uint8_t *d_img, const float *d_phi, int64_t pix0, int64_t pix1, unsigned /*m*/,
                 unsigned ld, const float *d_w, float gain, float ysub, uint8_t *d_out, float *d_zf, float *d_corr)
{
    if (!valid_ld(ld) || pix0 > pix1) return set_error(ctx, GLF_ERR_INVALID, "apply_filter: ld=%u", ld);
    if (pix0 == pix1) return GLF_OK;
    const int ppb = 256 / (ld / 4);
    int64_t nblk = ceil_div(pix1 - pix0, ppb);
    if (nblk > 8192) nblk = 8192; // grid-stride the rest
    dim3 grid((unsigned)nblk), block(256);
    switch (ld) {
    case 32: hipLaunchKernelGGL(k_apply_filter<32>, grid, block, 0, ctx->stream, d_img, d_phi, pix0, pix1, d_w, gain, ysub, d_out, d_zf, d_corr); break;
    case 64: hipLaunchKernelGGL(k_apply_filter<64>, grid, block, 0, ctx->stream, d_img, d_phi, pix0, pix1, d_w, gain, ysub, d_out, d_zf, d_corr); break;
    case 128: hipLaunchKernelGGL(k_apply_filter<128>, grid, block, 0, ctx->stream, d_img, d_phi, pix0, pix1, d_w, gain, ysub, d_out, d_zf, d_corr); break;
    case 256: hipLaunchKernelGGL(k_apply_filter<256>, grid, block, 0, ctx->stream, d_img, d_phi, pix0, pix1, d_w, gain, ysub, d_out, d_zf, d_corr); break;
    }
    GLF_LAUNCH_CHECK(ctx);
    return GLF_OK;
}

// ---- the same filter for more than 256 eigenpairs: one 256-column panel of Phi at a time ------------------------------
__global__ __launch_bounds__(256) void k_filter_accum(const float *__restrict__ phi, int64_t pix0, int64_t pix1,
                                                       const float *__restrict__ w, float *__restrict__ acc, int first)
{
    constexpr int LD = 256, LPP = LD / 4, PPB = 256 / LPP;
    const int q = threadIdx.x % LPP, pl = threadIdx.x / LPP;
    const float4 wq = reinterpret_cast<const float4 *>(w)[q];
    for (int64_t px = pix0 + (int64_t)blockIdx.x * PPB + pl; px < pix1; px += (int64_t)gridDim.x * PPB) {
        const float4 f = reinterpret_cast<const float4 *>(phi + (size_t)(px - pix0) * LD)[q];
        float s = f.x * wq.x + f.y * wq.y + f.z * wq.z + f.w * wq.w;
#pragma unroll
        for (int o = LPP / 2; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
        if (q == 0) acc[px - pix0] = first ? s : acc[px - pix0] + s;
    }
}

__global__ __launch_bounds__(256) void k_filter_finish(const uint8_t *__restrict__ img, const float *__restrict__ acc, int64_t pix0,
                                                        int64_t pix1, float gain, float ysub, uint8_t *__restrict__ out,
                                                        float *__restrict__ zf)
{
    const int64_t px = pix0 + (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (px >= pix1) return;
    const int y = (int)img[px];
    const float c = gain * acc[px - pix0] - ysub * (float)y;
    if (zf) zf[px] = (float)y + c;
    out[px] = filter_output(y, c);
}

// d_phi: this panel's rows for the pixels [pix0, pix1), [pix1 - pix0][256]
int filter_accumulate(glf_ctx *ctx, const float *d_phi, int64_t pix0, int64_t pix1, unsigned ld, const float *d_w, float *d_acc, bool first)
{
    if (ld != PANEL_COLS || pix0 > pix1) return set_error(ctx, GLF_ERR_INVALID, "filter_accumulate: ld=%u", ld);
    if (pix0 == pix1) return GLF_OK;
    int64_t nblk = ceil_div(pix1 - pix0, 4);
    if (nblk > 8192) nblk = 8192;
    hipLaunchKernelGGL(k_filter_accum, dim3((unsigned)nblk), dim3(256), 0, ctx->stream, d_phi, pix0, pix1, d_w, d_acc, first ? 1 : 0);
    GLF_LAUNCH_CHECK(ctx);
    return GLF_OK;
}

int filter_finish(glf_ctx *ctx, const uint8_t *d_img, const float *d_acc, int64_t pix0, int64_t pix1, float gain, float ysub, uint8_t *d_out,
                  float *d_zf)
{
    if (pix0 >= pix1) return GLF_OK;
    hipLaunchKernelGGL(k_filter_finish, dim3((unsigned)ceil_div(pix1 - pix0, 256)), dim3(256), 0, ctx->stream, d_img, d_acc, pix0, pix1, gain,
                       ysub, d_out, d_zf);
    GLF_LAUNCH_CHECK(ctx);
    return GLF_OK;
}

} // namespace glf
