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

template <int MB, int PB> // MB = ld / 32 column blocks; PB = 32-pixel blocks per wave
__global__ __launch_bounds__(256) void k_nystroem(const uint8_t *__restrict__ img, int width, int64_t pix0, int64_t pix1,
                                                   const float4 *__restrict__ samples, unsigned p, float s_loc,
                                                   float s_val, const float *__restrict__ psi,
                                                   float *__restrict__ phi, int raster,
                                                   const uint8_t *__restrict__ mask, const uint32_t *__restrict__ idx,
                                                   double *__restrict__ cpartial)
{
    constexpr int LD = MB * 32;
    constexpr int KC = NYS_KC;
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
    stage(0, 0);
    __syncthreads();
    for (int ch = 0; ch < nchunks; ++ch) {
        const int buf = ch & 1;
        if (ch + 1 < nchunks) stage(ch + 1, buf ^ 1);
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

template <int MB, int PB>
static int launch_nystroem(glf_ctx *ctx, const uint8_t *d_img, int width, int64_t pix0, int64_t pix1,
                           const float4 *d_samples, const uint8_t *d_mask, const uint32_t *d_idx, unsigned p,
                           KernelCoef coef, const float *d_psi, float *d_phi, int raster, double *d_c, float *kernel_ms)
{
    constexpr int LD = MB * 32;
    const int64_t npix = pix1 - pix0;
    const int64_t nwg = ceil_div(npix, 4 * 32 * PB);
    DevBuf<double> cpart;
    if (d_c) GLF_TRY(cpart.alloc(ctx, (size_t)nwg * LD));
    if (kernel_ms) GLF_HIP(ctx, hipEventRecord(ctx->ev[6], ctx->stream));
    hipLaunchKernelGGL((k_nystroem<MB, PB>), dim3((unsigned)nwg), dim3(256), 0, ctx->stream, d_img, width, pix0, pix1,
                       d_samples, p, coef.s_loc, coef.s_val, d_psi, d_phi, raster, d_mask, d_idx, d_c ? cpart.p : nullptr);
    GLF_LAUNCH_CHECK(ctx);
    if (kernel_ms) GLF_HIP(ctx, hipEventRecord(ctx->ev[7], ctx->stream));
    if (d_c) GLF_TRY(sum_rows(ctx, cpart.p, nwg, LD, d_c, true));
    if (kernel_ms) {
        GLF_HIP(ctx, hipEventSynchronize(ctx->ev[7]));
        GLF_HIP(ctx, hipEventElapsedTime(kernel_ms, ctx->ev[6], ctx->ev[7]));
    }
    return GLF_OK;
}

int nystroem_contract(glf_ctx *ctx, const uint8_t *d_img, int width, int height, int64_t pix0, int64_t pix1,
                      const float4 *d_samples, const uint8_t *d_mask, const uint32_t *d_idx, unsigned p,
                      KernelCoef coef, float /*scale folded into psi*/, const float *d_psi, unsigned m, unsigned ld,
                      float *d_phi, int raster, double *d_c, float *kernel_ms)
{
    const int64_t N = (int64_t)width * height;
    if (pix0 < 0 || pix1 > N || pix0 > pix1 || !valid_ld(ld) || m > ld)
        return set_error(ctx, GLF_ERR_INVALID, "nystroem_contract: bad range or ld=%u", ld);
    if (kernel_ms) *kernel_ms = 0.f;
    if (pix0 == pix1) return GLF_OK;
    switch (ld) {
    case 32:
        return launch_nystroem<1, 2>(ctx, d_img, width, pix0, pix1, d_samples, d_mask, d_idx, p, coef, d_psi, d_phi, raster, d_c, kernel_ms);
    case 64:
        return launch_nystroem<2, 2>(ctx, d_img, width, pix0, pix1, d_samples, d_mask, d_idx, p, coef, d_psi, d_phi, raster, d_c, kernel_ms);
    case 128:
        return launch_nystroem<4, 2>(ctx, d_img, width, pix0, pix1, d_samples, d_mask, d_idx, p, coef, d_psi, d_phi, raster, d_c, kernel_ms);
    case 256:
        return launch_nystroem<8, 1>(ctx, d_img, width, pix0, pix1, d_samples, d_mask, d_idx, p, coef, d_psi, d_phi, raster, d_c, kernel_ms);
    }
    return GLF_ERR_UNSUPPORTED;
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
