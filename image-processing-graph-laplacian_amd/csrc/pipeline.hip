// pipeline.hip -- the C-ABI stage entry points (mirror of hpc/*.h) and the whole
// approximate path glf_image_processing (hpc/image_processing.c:183-277).
#include "glf_internal.hpp"

#include <cmath>
#include <cstring>
#include <cstdlib>
#include <atomic>
#include <thread>

namespace glf {

static void shard_rows(const glf_ctx *ctx, int height, int *row0, int *row1)
{
    const int G = ctx->has_comm ? ctx->comm.size : 1, g = ctx->has_comm ? ctx->comm.rank : 0;
    (void)glf_shard_rows(height, g, G, row0, row1);
}

static int allreduce_f64(glf_ctx *ctx, double *d, size_t n)
{
    if (!ctx->has_comm) return GLF_OK;
    if (ctx->comm.allreduce_sum_f64(ctx->comm.user, d, n) != 0)
        return set_error(ctx, GLF_ERR_COMM, "allreduce_sum_f64 callback failed");
    return GLF_OK;
}

// Psi[i][j] = scale * Phi_A[i][j] * pinv[j]   (part_lower = phi_A * Pi^-1, hpc/nystroem.c:41; scale = -alpha)
// X[s][0] = y_s (the sample's pixel value) in a block of `ld` columns (the others stay zero)
__global__ void k_sample_values_column(const float4 *__restrict__ samples, unsigned p, unsigned ld, float *__restrict__ X)
{
    const unsigned s = blockIdx.x * blockDim.x + threadIdx.x;
    if (s < p) X[(size_t)s * ld] = samples[s].z;
}

// t[s][0] = D_s y_s - t[s][0] / alpha: K_A y_A from (L_A y_A) with L_A = alpha (D - K_A)
__global__ void k_ka_from_la(const float4 *__restrict__ samples, const double *__restrict__ degree, unsigned p, unsigned ld, double inv_alpha,
                             float *__restrict__ T)
{
    const unsigned s = blockIdx.x * blockDim.x + threadIdx.x;
    if (s < p) T[(size_t)s * ld] = (float)(degree[s] * (double)samples[s].z - (double)T[(size_t)s * ld] * inv_alpha);
}

__global__ void k_make_psi(const float *__restrict__ phiA, const float *__restrict__ pinv, unsigned p, unsigned ld, unsigned m,
                           float scale, float *__restrict__ psi)
{
    const size_t e = (size_t)blockIdx.x * 256 + threadIdx.x;
    const unsigned i = (unsigned)(e / ld), j = (unsigned)(e % ld);
    if (i >= p) return;
    psi[e] = (j < m) ? scale * (phiA[e] * pinv[j]) : 0.f;
}

// Jacobi preconditioner of L_A without the matrix: 1 / (alpha (D_i - K_ii)), K_ii = 1 (hpc/laplacian.c:31-35)
__global__ void k_dinv_from_degree(const double *__restrict__ degree, unsigned p, double alpha, float *__restrict__ dinv)
{
    const unsigned i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < p) dinv[i] = 1.0f / (float)(alpha * (degree[i] - 1.0));
}

__global__ void k_diag_inverse(const float *__restrict__ x, unsigned n, float *__restrict__ y)
{
    const unsigned i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) y[i] = 1.0f / x[i]; // InverseDiagMat, hpc/utils.c:578-579
}

// 256-column panel q of a row-major [rows][ld_total] matrix <-> a contiguous [rows][256] block (m > 256, see eigen.hip "panels")
static int pack_panel(glf_ctx *ctx, const float *rm, int64_t ld_total, int64_t rows, unsigned q, float *panel)
{
    GLF_HIP(ctx, hipMemcpy2DAsync(panel, sizeof(float) * PANEL_COLS, rm + (size_t)q * PANEL_COLS, sizeof(float) * (size_t)ld_total,
                                  sizeof(float) * PANEL_COLS, (size_t)rows, hipMemcpyDeviceToDevice, ctx->stream));
    return GLF_OK;
}
static int unpack_panel(glf_ctx *ctx, const float *panel, int64_t ld_total, int64_t rows, unsigned q, float *rm)
{
    GLF_HIP(ctx, hipMemcpy2DAsync(rm + (size_t)q * PANEL_COLS, sizeof(float) * (size_t)ld_total, panel, sizeof(float) * PANEL_COLS,
                                  sizeof(float) * PANEL_COLS, (size_t)rows, hipMemcpyDeviceToDevice, ctx->stream));
    return GLF_OK;
}

static int sum_host(glf_ctx *ctx, const double *d_v, unsigned n, double *out)
{
    std::vector<double> h(n);
    GLF_HIP(ctx, hipMemcpyAsync(h.data(), d_v, sizeof(double) * n, hipMemcpyDeviceToHost, ctx->stream));
    GLF_HIP(ctx, hipStreamSynchronize(ctx->stream));
    double s = 0.0;
    for (unsigned i = 0; i < n; ++i) s += h[i];
    *out = s;
    return GLF_OK;
}

// OrthonormaliseVecs / NormaliseVecs for more than 256 vectors: row-major in, panels inside, row-major out
static int orthonormalise_wide(glf_ctx *ctx, glf_mat *X, double *norms, bool normalise_only)
{
    const unsigned n = (unsigned)X->rows, m = (unsigned)X->cols, ld = (unsigned)X->ld;
    if (!wide_ld(ld) || m > ld) return set_error(ctx, GLF_ERR_INVALID, "more than 256 vectors need ld = a multiple of 256 (ld = %u)", ld);
    const unsigned npan = (unsigned)ceil_div(m, PANEL_COLS), n32 = (unsigned)round_up(n, VEC_PAD);
    DevBuf<float> panels;
    GLF_TRY(panels.alloc(ctx, (size_t)npan * n32 * PANEL_COLS));
    GLF_HIP(ctx, hipMemsetAsync(panels.p, 0, sizeof(float) * (size_t)npan * n32 * PANEL_COLS, ctx->stream));
    for (unsigned q = 0; q < npan; ++q) GLF_TRY(pack_panel(ctx, X->data, ld, n, q, panels.p + (size_t)q * n32 * PANEL_COLS));
    GLF_TRY(orthonormalise_panels(ctx, panels.p, n, m, norms, normalise_only));
    for (unsigned q = 0; q < npan; ++q) GLF_TRY(unpack_panel(ctx, panels.p + (size_t)q * n32 * PANEL_COLS, ld, n, q, X->data));
    GLF_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return GLF_OK;
}

} // namespace glf

using namespace glf;

extern "C" {

// ------------------------------------------------------------------------------------------
// Stage API
// ------------------------------------------------------------------------------------------
// Throughput mode: a batch of equally sized tiles (BASELINE config 5)
// ------------------------------------------------------------------------------------------
// Every tile goes through glf_image_processing unchanged; nctx contexts (one stream and one host thread each) work
// concurrently and take the tiles from a shared counter, so the launch- and latency-bound stages of one tile (the
// eigensolver at 1024 x 1024: dozens of small kernels and a host round trip per outer iteration) overlap the wide
// kernels of the others. Replicas only: no data-path exchange between tiles, contexts must not carry a comm.
int glf_image_processing_batch(glf_ctx *const *ctxs, int nctx, const glf_options *opt, const uint8_t *d_imgs, int width, int height,
                               int ntiles, uint8_t *d_outs, float *d_zfs, glf_stats *stats)
{
    if (!ctxs || nctx < 1 || !ctxs[0]) return GLF_ERR_INVALID;
    if (ntiles < 0 || width <= 0 || height <= 0 || (ntiles > 0 && (!d_imgs || !d_outs)))
        return set_error(ctxs[0], GLF_ERR_INVALID, "glf_image_processing_batch: %d tiles of %dx%d", ntiles, width, height);
    for (int c = 0; c < nctx; ++c) {
        if (!ctxs[c]) return set_error(ctxs[0], GLF_ERR_INVALID, "glf_image_processing_batch: context %d is NULL", c);
        if (ctxs[c]->has_comm) return set_error(ctxs[0], GLF_ERR_INVALID, "glf_image_processing_batch: context %d has a comm (tiles are replicas)", c);
        for (int d = 0; d < c; ++d)
            if (ctxs[d] == ctxs[c]) return set_error(ctxs[0], GLF_ERR_INVALID, "glf_image_processing_batch: context %d listed twice", c);
    }
    const size_t N = (size_t)width * height;
    std::atomic<int> next{0}, status{GLF_OK}, failed{-1};
    auto worker = [&](int c) {
        if (hipSetDevice(ctxs[c]->device) != hipSuccess) {
            int expected = GLF_OK;
            if (status.compare_exchange_strong(expected, GLF_ERR_NODEVICE)) failed.store(c);
            return;
        }
        for (;;) {
            const int t = next.fetch_add(1);
            if (t >= ntiles || status.load() != GLF_OK) break;
            const int rc = glf_image_processing(ctxs[c], opt, d_imgs + (size_t)t * N, width, height, d_outs + (size_t)t * N,
                                                d_zfs ? d_zfs + (size_t)t * N : nullptr, nullptr, stats ? stats + t : nullptr);
            if (rc != GLF_OK) {
                int expected = GLF_OK;
                if (status.compare_exchange_strong(expected, rc)) failed.store(c);
                break;
            }
        }
    };
    std::vector<std::thread> threads;
    const int nworkers = std::min(nctx, std::max(1, ntiles));
    for (int c = 1; c < nworkers; ++c) threads.emplace_back(worker, c);
    worker(0);
    for (auto &th : threads) th.join();
    const int rc = status.load(), fc = failed.load();
    if (rc != GLF_OK && fc > 0) std::snprintf(ctxs[0]->last_error, sizeof(ctxs[0]->last_error), "context %d: %s", fc, ctxs[fc]->last_error);
    return rc;
}

// ------------------------------------------------------------------------------------------

int glf_ComputeAffinityMatrices(glf_ctx *ctx, glf_mat *K_A, glf_mat *K_B, const uint8_t *d_img, int width, int height,
                                unsigned sample_size, const unsigned *sample_indices, int kernel, float h_loc,
                                float h_val)
{
    if (!ctx || !K_B || !d_img || width <= 0 || height <= 0) return GLF_ERR_INVALID;
    if (kernel < GLF_KERNEL_BILATERAL || kernel > GLF_KERNEL_NLM) return set_error(ctx, GLF_ERR_INVALID, "kernel %d", kernel);
    if (kernel == GLF_KERNEL_NLM && (width < 3 || height < 3)) // (nlm.hip reflects an out-of-image patch index once: valid from 3 pixels on)
        return set_error(ctx, GLF_ERR_UNSUPPORTED, "non-local-means kernel: the image must be at least 3 x 3 pixels (%d x %d)", width, height);
    GLF_ENTER(ctx);
    const unsigned p = sample_size;
    SampleTables tb;
    GLF_TRY(build_sample_tables(ctx, d_img, width, height, p, sample_indices, tb));
    const KernelCoef coef = make_coef(kernel, h_loc, h_val);
    DevBuf<double> deg;
    GLF_TRY(deg.alloc(ctx, p));
    int row0, row1;
    shard_rows(ctx, height, &row0, &row1);
    GLF_TRY(degree_rows_auto(ctx, d_img, width, height, row0, row1, tb.samples.p, p, sample_indices, coef, deg.p, 0, nullptr, tb.idx.p));
    GLF_TRY(allreduce_f64(ctx, deg.p, p));
    if (K_A) {
        const int64_t lda = round_up(p, VEC_PAD);
        GLF_TRY(glf_mat_create_dense(ctx, K_A, p, p, lda)); // zero-filled incl. padding
        GLF_TRY(build_sample_matrix(ctx, tb.samples.p, p, coef, K_A->data, lda, false, 0.0, nullptr, 0, 0, d_img, width, height, tb.idx.p));
    }
    std::memset(K_B, 0, sizeof(*K_B));
    K_B->kind = GLF_MAT_KERNEL_B;
    K_B->rows = p;
    K_B->cols = (int64_t)width * height - p;
    K_B->img = d_img;
    K_B->width = width;
    K_B->height = height;
    K_B->p = p;
    K_B->scale = 1.0f;
    K_B->h_loc = h_loc;
    K_B->h_val = h_val;
    K_B->kernel = kernel;
    K_B->samples = reinterpret_cast<const float *>(tb.samples.take());
    K_B->mask = tb.mask.take();
    K_B->idx = tb.idx.take();
    K_B->degree = deg.take();
    K_B->owns_desc = 1;
    GLF_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return GLF_OK;
}

int glf_ComputeLaplacianMatrix(glf_ctx *ctx, glf_mat *L_A, glf_mat *L_B, const glf_mat *K_A, const glf_mat *K_B,
                               double *alpha_out)
{
    if (!ctx || !L_A || !K_B || K_B->kind != GLF_MAT_KERNEL_B || !K_B->degree) return GLF_ERR_INVALID;
    GLF_ENTER(ctx);
    const unsigned p = K_B->p;
    if (K_A && (K_A->kind != GLF_MAT_DENSE || K_A->rows != p || K_A->cols != p))
        return set_error(ctx, GLF_ERR_INVALID, "K_A must be dense p x p");
    // alpha = 1 / VecMean(D_A), hpc/laplacian.c:29 + hpc/utils.c:378-388
    double sum = 0.0;
    GLF_TRY(sum_host(ctx, K_B->degree, p, &sum));
    const double alpha = 1.0 / (sum / (double)p);
    const int64_t lda = round_up(p, VEC_PAD);
    GLF_TRY(glf_mat_create_dense(ctx, L_A, p, p, lda));
    const KernelCoef coef = make_coef(K_B->kernel, K_B->h_loc, K_B->h_val);
    if (K_A)
        GLF_TRY(laplacian_from_KA(ctx, K_A->data, K_A->ld, p, L_A->data, lda, alpha, K_B->degree));
    else
        GLF_TRY(build_sample_matrix(ctx, reinterpret_cast<const float4 *>(K_B->samples), p, coef, L_A->data, lda, true,
                                    alpha, K_B->degree, 0, 0, K_B->img, K_B->width, K_B->height, K_B->idx));
    if (L_B) { // L_B = -alpha K_B, hpc/laplacian.c:37-38: same generator, other scale (shares tables)
        *L_B = *K_B;
        L_B->scale = (float)(-alpha);
        L_B->owns_desc = 0;
    }
    if (alpha_out) *alpha_out = alpha;
    GLF_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return GLF_OK;
}

int glf_InversePowerIteration(glf_ctx *ctx, const glf_mat *A, unsigned m, glf_mat *eigenvectors, glf_mat *eigenvalues,
                              int optiGramSchmidt, double epsilon, double inner_rtol, int max_outer, const double *X0,
                              glf_eig_stats *stats)
{
    if (!ctx || !A || A->kind != GLF_MAT_DENSE || A->rows != A->cols) return GLF_ERR_INVALID;
    GLF_ENTER(ctx);
    const unsigned p = (unsigned)A->rows;
    if (m == 0 || m >= p) return set_error(ctx, GLF_ERR_INVALID, "need 0 < m < p (m=%u p=%u)", m, p);
    const unsigned ld = ld_total_for(m);
    const unsigned p32 = (unsigned)round_up(p, VEC_PAD);
    if (m > PANEL_COLS) { // panels of 256 vectors (the reference default m = p - 1, hpc/image_processing.c:96-108)
        const unsigned npan = ld / PANEL_COLS;
        glf_mat wide;
        GLF_TRY(glf_mat_create_dense(ctx, &wide, p32, m, ld));
        wide.rows = p;
        DevBuf<float> panels;
        GLF_TRY(panels.alloc(ctx, (size_t)npan * p32 * PANEL_COLS));
        std::vector<double> lamw(m);
        int rcw = inverse_power_iteration_panels(ctx, A->data, A->ld, p, m, X0, 1, optiGramSchmidt, epsilon, inner_rtol,
                                                 max_outer > 0 ? max_outer : 100000, panels.p, lamw.data(), stats, nullptr, nullptr);
        if (rcw != GLF_OK && rcw != GLF_ERR_NOCONV) {
            glf_mat_destroy(ctx, &wide);
            return rcw;
        }
        for (unsigned q = 0; q < npan; ++q) GLF_TRY(unpack_panel(ctx, panels.p + (size_t)q * p32 * PANEL_COLS, ld, p32, q, wide.data));
        GLF_HIP(ctx, hipStreamSynchronize(ctx->stream));
        if (eigenvalues) {
            GLF_TRY(glf_mat_create_diag(ctx, eigenvalues, m));
            std::vector<float> lf(m);
            for (unsigned j = 0; j < m; ++j) lf[j] = (float)lamw[j];
            GLF_TRY(glf_memcpy_h2d(ctx, eigenvalues->data, lf.data(), sizeof(float) * m));
        }
        if (eigenvectors) *eigenvectors = wide;
        else glf_mat_destroy(ctx, &wide);
        return rcw;
    }
    glf_mat vecs;
    GLF_TRY(glf_mat_create_dense(ctx, &vecs, p32, m, ld));
    vecs.rows = p;
    std::vector<double> lam(m);
    int rc = inverse_power_iteration(ctx, A->data, A->ld, p, m, ld, X0, optiGramSchmidt, epsilon, inner_rtol,
                                     max_outer > 0 ? max_outer : 100000, vecs.data, lam.data(), stats);
    if (rc != GLF_OK && rc != GLF_ERR_NOCONV) {
        glf_mat_destroy(ctx, &vecs);
        return rc;
    }
    if (eigenvalues) {
        int rc2 = glf_mat_create_diag(ctx, eigenvalues, m);
        if (rc2 != GLF_OK) return rc2;
        std::vector<float> lf(m);
        for (unsigned j = 0; j < m; ++j) lf[j] = (float)lam[j];
        GLF_TRY(glf_memcpy_h2d(ctx, eigenvalues->data, lf.data(), sizeof(float) * m));
    }
    if (eigenvectors) *eigenvectors = vecs;
    else glf_mat_destroy(ctx, &vecs);
    return rc;
}

int glf_OrthonormaliseVecs(glf_ctx *ctx, glf_mat *X, double *norms)
{
    if (!ctx || !X || X->kind != GLF_MAT_DENSE) return GLF_ERR_INVALID;
    GLF_ENTER(ctx);
    if (X->cols > PANEL_COLS) return orthonormalise_wide(ctx, X, norms, false);
    return orthonormalise(ctx, X->data, (unsigned)X->rows, (unsigned)X->cols, (unsigned)X->ld, norms);
}

int glf_NormaliseVecs(glf_ctx *ctx, glf_mat *X, double *norms)
{
    if (!ctx || !X || X->kind != GLF_MAT_DENSE) return GLF_ERR_INVALID;
    GLF_ENTER(ctx);
    if (X->cols > PANEL_COLS) return orthonormalise_wide(ctx, X, norms, true);
    return normalise(ctx, X->data, (unsigned)X->rows, (unsigned)X->cols, (unsigned)X->ld, norms);
}

int glf_InverseDiagMat(glf_ctx *ctx, const glf_mat *x, glf_mat *inv)
{
    if (!ctx || !x || !inv || x->kind != GLF_MAT_DIAG) return GLF_ERR_INVALID;
    GLF_ENTER(ctx);
    GLF_TRY(glf_mat_create_diag(ctx, inv, x->rows));
    const unsigned n = (unsigned)x->rows;
    hipLaunchKernelGGL(k_diag_inverse, dim3((n + 255) / 256), dim3(256), 0, ctx->stream, x->data, n, inv->data);
    GLF_LAUNCH_CHECK(ctx);
    return GLF_OK;
}

int glf_Nystroem(glf_ctx *ctx, const glf_mat *B, const glf_mat *phi_A, const glf_mat *Pi_A_Inv, glf_mat *phi)
{
    if (!ctx || !B || !phi_A || !Pi_A_Inv || !phi) return GLF_ERR_INVALID;
    GLF_ENTER(ctx);
    if (B->kind != GLF_MAT_KERNEL_B || phi_A->kind != GLF_MAT_DENSE || Pi_A_Inv->kind != GLF_MAT_DIAG)
        return set_error(ctx, GLF_ERR_INVALID, "Nystroem: B must be a KERNEL_B descriptor, phi_A dense, Pi_A_Inv diagonal");
    const unsigned p = B->p, m = (unsigned)phi_A->cols, ld = (unsigned)phi_A->ld;
    if (phi_A->rows != p || Pi_A_Inv->rows != m || !(valid_ld(ld) || wide_ld(ld)) || m > ld)
        return set_error(ctx, GLF_ERR_INVALID, "Nystroem: shape mismatch (p=%u, phi_A %lld x %u ld %u)", p,
                         (long long)phi_A->rows, m, ld);
    const int64_t N = (int64_t)B->width * B->height;
    if (wide_ld(ld)) { // more than 256 eigenvectors: one 256-column panel of Phi at a time (columns are independent)
        const unsigned npan = (unsigned)ceil_div(m, PANEL_COLS), p64 = (unsigned)round_up(p, NYS_PAD);
        DevBuf<float> pa, psiw, phip;
        GLF_TRY(pa.alloc(ctx, (size_t)p64 * PANEL_COLS));
        GLF_TRY(psiw.alloc(ctx, (size_t)p64 * PANEL_COLS));
        GLF_TRY(phip.alloc(ctx, (size_t)N * PANEL_COLS));
        GLF_TRY(glf_mat_create_dense(ctx, phi, N, m, ld));
        phi->row_order = GLF_ROWS_SAMPLE_FIRST;
        const KernelCoef coefw = make_coef(B->kernel, B->h_loc, B->h_val);
        for (unsigned q = 0; q < npan; ++q) {
            const unsigned mq = std::min(PANEL_COLS, m - q * PANEL_COLS);
            GLF_HIP(ctx, hipMemsetAsync(pa.p, 0, sizeof(float) * (size_t)p64 * PANEL_COLS, ctx->stream));
            GLF_HIP(ctx, hipMemsetAsync(psiw.p, 0, sizeof(float) * (size_t)p64 * PANEL_COLS, ctx->stream));
            GLF_TRY(pack_panel(ctx, phi_A->data, ld, p, q, pa.p));
            hipLaunchKernelGGL(k_make_psi, dim3((unsigned)ceil_div((int64_t)p * PANEL_COLS, 256)), dim3(256), 0, ctx->stream, pa.p,
                               Pi_A_Inv->data + (size_t)q * PANEL_COLS, p, PANEL_COLS, mq, B->scale, psiw.p);
            GLF_LAUNCH_CHECK(ctx);
            GLF_TRY(nystroem_contract(ctx, B->img, B->width, B->height, 0, N, reinterpret_cast<const float4 *>(B->samples), B->mask, B->idx, p,
                                      coefw, B->scale, psiw.p, mq, PANEL_COLS, phip.p, 0, nullptr, nullptr));
            GLF_TRY(scatter_sample_rows(ctx, pa.p, p, PANEL_COLS, B->idx, phip.p, 0, B->img, nullptr, mq));
            GLF_TRY(unpack_panel(ctx, phip.p, ld, N, q, phi->data));
        }
        GLF_HIP(ctx, hipStreamSynchronize(ctx->stream));
        return GLF_OK;
    }
    DevBuf<float> psi;
    GLF_TRY(psi.alloc(ctx, (size_t)round_up(p, NYS_PAD) * ld));
    GLF_HIP(ctx, hipMemsetAsync(psi.p, 0, sizeof(float) * (size_t)round_up(p, NYS_PAD) * ld, ctx->stream));
    hipLaunchKernelGGL(k_make_psi, dim3((unsigned)ceil_div((int64_t)p * ld, 256)), dim3(256), 0, ctx->stream, phi_A->data,
                       Pi_A_Inv->data, p, ld, m, B->scale, psi.p);
    GLF_LAUNCH_CHECK(ctx);
    GLF_TRY(glf_mat_create_dense(ctx, phi, N, m, ld));
    phi->row_order = GLF_ROWS_SAMPLE_FIRST;
    const KernelCoef coef = make_coef(B->kernel, B->h_loc, B->h_val);
    GLF_TRY(nystroem_contract(ctx, B->img, B->width, B->height, 0, N, reinterpret_cast<const float4 *>(B->samples), B->mask,
                              B->idx, p, coef, B->scale, psi.p, m, ld, phi->data, 0, nullptr, nullptr));
    GLF_TRY(scatter_sample_rows(ctx, phi_A->data, p, ld, B->idx, phi->data, 0, B->img, nullptr, m));
    GLF_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return GLF_OK;
}

int glf_Permutation(glf_ctx *ctx, const glf_mat *in, const unsigned *sample_indices, unsigned num, glf_mat *out)
{
    if (!ctx || !in || !out || in->kind != GLF_MAT_DENSE || !sample_indices) return GLF_ERR_INVALID;
    GLF_ENTER(ctx);
    DevBuf<uint32_t> idx;
    GLF_TRY(idx.alloc(ctx, num));
    GLF_TRY(glf_memcpy_h2d(ctx, idx.p, sample_indices, sizeof(uint32_t) * num));
    GLF_TRY(glf_mat_create_dense(ctx, out, in->rows, in->cols, in->ld));
    out->row_order = GLF_ROWS_RASTER;
    if (wide_ld((unsigned)in->ld)) { // 256-column panels
        DevBuf<float> a, b;
        GLF_TRY(a.alloc(ctx, (size_t)in->rows * PANEL_COLS));
        GLF_TRY(b.alloc(ctx, (size_t)in->rows * PANEL_COLS));
        for (unsigned q = 0; q < (unsigned)(in->ld / PANEL_COLS); ++q) {
            GLF_TRY(pack_panel(ctx, in->data, in->ld, in->rows, q, a.p));
            GLF_TRY(permute_rows(ctx, a.p, b.p, in->rows, PANEL_COLS, idx.p, num));
            GLF_TRY(unpack_panel(ctx, b.p, in->ld, in->rows, q, out->data));
        }
        GLF_HIP(ctx, hipStreamSynchronize(ctx->stream));
        return GLF_OK;
    }
    GLF_TRY(permute_rows(ctx, in->data, out->data, in->rows, (unsigned)in->ld, idx.p, num));
    GLF_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return GLF_OK;
}

int glf_ComputeResultFromLaplacian(glf_ctx *ctx, const uint8_t *d_img, const glf_mat *phi, const glf_mat *Pi, unsigned width,
                                   unsigned height, float gain, uint8_t *d_out, float *d_zf)
{
    if (!ctx || !d_img || !phi || !Pi || !d_out) return GLF_ERR_INVALID;
    GLF_ENTER(ctx);
    const int64_t N = (int64_t)width * height;
    const unsigned m = (unsigned)phi->cols, ld = (unsigned)phi->ld;
    if (phi->kind != GLF_MAT_DENSE || phi->rows != N || Pi->kind != GLF_MAT_DIAG || Pi->rows != m || !(valid_ld(ld) || wide_ld(ld)))
        return set_error(ctx, GLF_ERR_INVALID, "ComputeResultFromLaplacian: shape mismatch");
    if (phi->row_order == GLF_ROWS_SAMPLE_FIRST)
        return set_error(ctx, GLF_ERR_INVALID, "phi is in sample-first order: call glf_Permutation first (hpc/image_processing.c:250)");
    if (wide_ld(ld)) { // 256-column panels: right = phi^T z panel by panel, then the correction accumulated over the panels
        const unsigned npan = (unsigned)ceil_div(m, PANEL_COLS);
        DevBuf<float> pan, wq, acc;
        DevBuf<double> cq;
        GLF_TRY(pan.alloc(ctx, (size_t)N * PANEL_COLS));
        GLF_TRY(wq.alloc(ctx, PANEL_COLS));
        GLF_TRY(acc.alloc(ctx, (size_t)N));
        GLF_TRY(cq.alloc(ctx, PANEL_COLS));
        std::vector<float> hp(m);
        GLF_TRY(glf_memcpy_d2h(ctx, hp.data(), Pi->data, sizeof(float) * m));
        for (unsigned q = 0; q < npan; ++q) {
            const unsigned mq = std::min(PANEL_COLS, m - q * PANEL_COLS);
            GLF_TRY(pack_panel(ctx, phi->data, ld, N, q, pan.p));
            GLF_TRY(phi_t_y(ctx, pan.p, d_img, 0, N, mq, PANEL_COLS, cq.p));
            std::vector<double> hc(PANEL_COLS);
            std::vector<float> hw(PANEL_COLS, 0.f);
            GLF_TRY(glf_memcpy_d2h(ctx, hc.data(), cq.p, sizeof(double) * PANEL_COLS));
            for (unsigned j = 0; j < mq; ++j) hw[j] = (float)((double)hp[q * PANEL_COLS + j] * hc[j]);
            GLF_TRY(glf_memcpy_h2d(ctx, wq.p, hw.data(), sizeof(float) * PANEL_COLS));
            GLF_TRY(filter_accumulate(ctx, pan.p, 0, N, PANEL_COLS, wq.p, acc.p, q == 0));
        }
        GLF_TRY(filter_finish(ctx, d_img, acc.p, 0, N, gain, 0.f, d_out, d_zf));
        GLF_HIP(ctx, hipStreamSynchronize(ctx->stream));
        return GLF_OK;
    }
    DevBuf<double> c;
    GLF_TRY(c.alloc(ctx, ld));
    GLF_TRY(phi_t_y(ctx, phi->data, d_img, 0, N, m, ld, c.p)); // right = phi^T z, hpc/display.c:66
    std::vector<double> hc(ld);
    std::vector<float> hp(m), hw(ld, 0.f);
    GLF_TRY(glf_memcpy_d2h(ctx, hc.data(), c.p, sizeof(double) * ld));
    GLF_TRY(glf_memcpy_d2h(ctx, hp.data(), Pi->data, sizeof(float) * m));
    for (unsigned j = 0; j < m; ++j) hw[j] = (float)((double)hp[j] * hc[j]); // left = phi Pi, :64
    DevBuf<float> w;
    GLF_TRY(w.alloc(ctx, ld));
    GLF_TRY(glf_memcpy_h2d(ctx, w.p, hw.data(), sizeof(float) * ld));
    GLF_TRY(apply_filter(ctx, d_img, phi->data, 0, N, m, ld, w.p, gain, 0.f, d_out, d_zf));
    GLF_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return GLF_OK;
}

int glf_EntireComputation(glf_ctx *ctx, const uint8_t *d_img, int width, int height, int kernel, float h_loc, float h_val,
                          uint8_t *d_out, float *d_zf, double *alpha_out)
{
    if (!ctx || !d_img || !d_out || width <= 0 || height <= 0) return GLF_ERR_INVALID;
    if (kernel < GLF_KERNEL_BILATERAL || kernel > GLF_KERNEL_SPATIAL) return set_error(ctx, GLF_ERR_INVALID, "kernel %d", kernel);
    GLF_ENTER(ctx);
    return entire_computation(ctx, d_img, width, height, make_coef(kernel, h_loc, h_val), d_out, d_zf, alpha_out);
}

// ------------------------------------------------------------------------------------------
// Whole approximate path
// ------------------------------------------------------------------------------------------

int glf_image_processing(glf_ctx *ctx, const glf_options *opt_in, const uint8_t *d_img, int width, int height,
                         uint8_t *d_out, float *d_zf, double *eigvals_out, glf_stats *stats)
{
    return glf_image_processing_capture(ctx, opt_in, d_img, width, height, d_out, d_zf, eigvals_out, stats, nullptr);
}

int glf_image_processing_capture(glf_ctx *ctx, const glf_options *opt_in, const uint8_t *d_img, int width, int height,
                                 uint8_t *d_out, float *d_zf, double *eigvals_out, glf_stats *stats, glf_capture *cap)
{
    if (!ctx || !d_img || !d_out || width <= 0 || height <= 0) return GLF_ERR_INVALID;
    if (cap && cap->struct_size != sizeof(glf_capture))
        return set_error(ctx, GLF_ERR_INVALID, "glf_capture.struct_size %u != %zu", cap->struct_size, sizeof(glf_capture));
    GLF_ENTER(ctx);
    pool_age(ctx);
    glf_options opt;
    glf_options_default(&opt);
    if (opt_in) {
        if (opt_in->struct_size != sizeof(glf_options))
            return set_error(ctx, GLF_ERR_INVALID, "glf_options.struct_size %u != %zu", opt_in->struct_size, sizeof(glf_options));
        opt = *opt_in;
    }
    if (opt.kernel < GLF_KERNEL_BILATERAL || opt.kernel > GLF_KERNEL_NLM) return set_error(ctx, GLF_ERR_INVALID, "kernel %d", opt.kernel);
    if (opt.kernel == GLF_KERNEL_NLM && (width < 3 || height < 3))
        return set_error(ctx, GLF_ERR_UNSUPPORTED, "non-local-means kernel: the image must be at least 3 x 3 pixels (%d x %d)", width, height);
    if (opt.filter_mode < GLF_FILTER_REFERENCE || opt.filter_mode > GLF_FILTER_SHARPEN) return set_error(ctx, GLF_ERR_INVALID, "filter_mode %d", opt.filter_mode);
    // f(Pi), the gain and the y term of the filter z = ysub' y + gain Phi f(Pi) Phi^T y:
    //   reference: z = y + gain Phi Pi^k Phi^T y (MatPow is a no-op there, hpc/utils.c:721 => k = 1);
    //   PoC: z = y - Phi diag(mu + 5) Phi^T y (python/image_processing.py:304-305);
    //   smoothing / sharpening (:197-241): z = Phi f(s) Phi^T y with s = 1 - mu the eigenvalues of W = I - L, f(s) = s or
    //   (1 + beta) s^2 - beta s^3 -- no y term: the kernels subtract it from the correction (filter_ysub).
    const bool ref_filter = opt.filter_mode == GLF_FILTER_REFERENCE;
    const float filter_gain = ref_filter ? opt.gain : 1.0f;
    const float filter_ysub = opt.filter_mode >= GLF_FILTER_SMOOTH ? 1.0f : 0.0f;
    auto filter_weight = [&](double lambda) {
        const double s1 = 1.0 - lambda, beta = (double)opt.filter_beta;
        switch (opt.filter_mode) {
        case GLF_FILTER_POC: return -(lambda + 5.0);
        case GLF_FILTER_SMOOTH: return s1;
        case GLF_FILTER_SHARPEN: return (1.0 + beta) * s1 * s1 - beta * s1 * s1 * s1;
        default: return std::pow(lambda, (double)(opt.filter_pow > 0 ? opt.filter_pow : 1));
        }
    };
    const int64_t N = (int64_t)width * height;
    if (N >= (int64_t)1 << 31) return set_error(ctx, GLF_ERR_UNSUPPORTED, "image too large");
    hipStream_t st = ctx->stream;
    glf_stats S{};

    // p = width*height*0.01 (truncating), hpc/image_processing.c:187; Sampling rewrites it, :191-193
    unsigned p = opt.num_samples ? opt.num_samples : (unsigned)((double)N * opt.sample_frac);
    unsigned *h_idx = nullptr;
    {
        int rc = opt.sampling == GLF_SAMPLING_RANDOM ? glf_RandomSampling(width, height, &p, &h_idx, opt.sampling_seed)
                                                       : glf_Sampling(width, height, &p, &h_idx);
        if (rc != GLF_OK || p < 2) {
            std::free(h_idx);
            return set_error(ctx, GLF_ERR_INVALID, "sampling failed (requested %u samples on %dx%d)", p, width, height);
        }
    }
    struct FreeIdx {
        unsigned *q;
        ~FreeIdx() { std::free(q); }
    } free_idx{h_idx};
    // GetNumberEigenvalues, hpc/image_processing.c:96-108
    unsigned m = opt.num_eigvals;
    if (m == 0 || m >= p) m = p - 1;
    const bool wide = m > PANEL_COLS; // more than 256 eigenpairs (the reference default m = p - 1): 256-column panels, see below
    if (wide && opt.filter_mode == GLF_FILTER_SHARPEN) // (its Gram matrix Phi^T Phi would couple the panels)
        return set_error(ctx, GLF_ERR_UNSUPPORTED, "the sharpening filter takes at most %u eigenpairs (%u asked for)", PANEL_COLS, m);
    if (wide && cap) return set_error(ctx, GLF_ERR_UNSUPPORTED, "glf_capture with more than 256 eigenpairs");
    const unsigned ld = wide ? PANEL_COLS : ld_for(m);
    const unsigned p32 = (unsigned)round_up(p, VEC_PAD);
    const KernelCoef coef = make_coef(opt.kernel, opt.h_loc, opt.h_val);
    int row0, row1;
    shard_rows(ctx, height, &row0, &row1);
    const int64_t pix0 = (int64_t)row0 * width, pix1 = (int64_t)row1 * width;
    S.p = p;
    S.m = m;
    S.row0 = row0;
    S.row1 = row1;

    GLF_HIP(ctx, hipEventRecord(ctx->ev[0], st));
    // ---- affinity: sample tables + degree (K_B generated on the fly) -------------------------
    SampleTables tb;
    GLF_TRY(build_sample_tables(ctx, d_img, width, height, p, h_idx, tb));
    DevBuf<double> deg; // D_A [p], then (grid-factored degree only) the value-weighted sums U[s] = sum_px K(s, px) y[px] [p]
    GLF_TRY(deg.alloc(ctx, 2 * (size_t)p));
    bool have_ysum = false;
    GLF_TRY(degree_rows_auto(ctx, d_img, width, height, row0, row1, tb.samples.p, p, h_idx, coef, deg.p, opt.skip_exact_zeros,
                             &S.degree_evaluated, tb.idx.p, deg.p + p, &have_ysum));
    GLF_TRY(allreduce_f64(ctx, deg.p, have_ysum ? 2 * (size_t)p : p));
    GLF_HIP(ctx, hipEventRecord(ctx->ev[1], st));
    if (cap) {
        cap->ld = ld;
        if (cap->h_degree) {
            GLF_HIP(ctx, hipMemcpyAsync(cap->h_degree, deg.p, sizeof(double) * p, hipMemcpyDeviceToHost, st));
            GLF_HIP(ctx, hipStreamSynchronize(st));
        }
        if ((cap->d_phi_A && cap->phi_A_floats < (size_t)p32 * ld) || (cap->d_phi && cap->phi_floats < (size_t)(pix1 - pix0) * ld) ||
            (cap->d_corr && cap->corr_floats < (size_t)(pix1 - pix0)))
            return set_error(ctx, GLF_ERR_INVALID, "glf_capture: phi_A needs %zu floats, phi %zu", (size_t)p32 * ld, (size_t)(pix1 - pix0) * ld);
    }
    // ---- Laplacian ---------------------------------------------------------------------------
    double dsum = 0.0;
    GLF_TRY(sum_host(ctx, deg.p, p, &dsum));
    const double alpha = 1.0 / (dsum / (double)p);
    S.alpha = alpha;
    // With a communicator that can all-gather, L_A is sharded: rank g holds the column block
    // L_A[:, r0:r1) (= its row block transposed; L_A is symmetric) and computes rows [r0,r1) of every
    // mat-vec. Otherwise (single GPU, f32 contraction, or no allgather callback) L_A is whole.
    MatShard shard;
    bool shard_eig = ctx->has_comm && ctx->comm.allgather_f32 && ctx->contraction == GLF_CONTRACT_F16_SPLIT && !wide;
    // For a tensor-grid sample set L_A is applied in grid-factored form and never stored (GLF_MV_PATH = grid | dense | auto;
    // auto: from 16 384 samples on -- below, streaming a small stored L_A is cheaper than the factored sweep's fixed cost)
    struct GridOpGuard {
        GridOp *op = nullptr;
        ~GridOpGuard() { grid_op_destroy(op); }
    } gop;
    {
        const bool want = (ctx->tune.mv_path == 1 || ctx->tune.mv_path == 3 || ctx->tune.mv_path == 4) ? true : ctx->tune.mv_path == 2 ? false : p >= 16384;
        if (want && opt.kernel != GLF_KERNEL_NLM) {
            const int rc = grid_op_create(ctx, tb.samples.p, h_idx, p, width, height, coef, &gop.op);
            if (rc != GLF_OK && rc != GLF_ERR_UNSUPPORTED) return rc;
        }
    }
    S.matvec_path = gop.op ? grid_op_path(gop.op) : 0;
    // With the band form a sweep costs less (0.19 ms at cfg4) than the all-gather of its 21.8 MB operand block over xGMI
    // (~0.2-0.4 ms), and the whole eigen-solve (4.8 ms) less than the sharded one's 12 all-gathers + ~46 small all-reduces:
    // every rank then runs the eigen-solve on all rows (bit-identical to one GPU, no collective) and only the pixel rows
    // -- degree, extension, filter -- are sharded. EIG_SHARD=1 forces the row-sharded solve, 0 the replicated one.
    if (shard_eig && ((S.matvec_path == 4 && ctx->tune.eig_shard != 1) || ctx->tune.eig_shard == 2)) shard_eig = false;
    S.eigen_sharded = shard_eig ? 1 : 0;
    if (gop.op) {
        shard.grid = gop.op;
        shard.grid_alpha = alpha;
        shard.grid_degree = deg.p;
        shard.grid_window = opt.skip_exact_zeros;
    }
    if (shard_eig) {
        shard.rows_per_rank = gop.op ? grid_op_rows_per_rank(gop.op, ctx->comm.size) : shard_rows_per_rank(p, ctx->comm.size);
        const uint64_t b = (uint64_t)shard.rows_per_rank * (unsigned)ctx->comm.rank;
        shard.row0 = (unsigned)(b < p ? b : p);
        shard.row1 = (unsigned)(b + shard.rows_per_rank < p ? b + shard.rows_per_rank : p);
    }
    // Exact-zero tile skipping of the mat-vec: |2^10 L_A[i][j]| = 2^10 alpha K < 2^-25 once t > 35 + log2(alpha)
    DevBuf<int4> kbox;
    if (!gop.op && opt.skip_exact_zeros && coef.s_loc > 0.f && opt.kernel != GLF_KERNEL_NLM && ctx->contraction == GLF_CONTRACT_F16_SPLIT) {
        const double t_zero = 35.5 + std::log2(alpha);
        GLF_TRY(kbox.alloc(ctx, (size_t)ceil_div(p, 64)));
        GLF_TRY(chunk_boxes(ctx, tb.samples.p, p, kbox.p));
        shard.kbox = kbox.p;
        shard.radius = t_zero > 0.0 ? (int)std::floor(std::sqrt(t_zero / (double)coef.s_loc)) + 1 : 0;
    }
    const unsigned la_cols = shard_eig ? shard.row1 - shard.row0 : p;
    const int64_t lda = shard_eig ? round_up(la_cols ? la_cols : 1, VEC_PAD) : (int64_t)p32;
    DevBuf<float> LA, dinv;
    if (!gop.op) {
        GLF_TRY(LA.alloc(ctx, (size_t)p * lda));
        if (la_cols) // writes every element of the p x lda block, padding columns included
            GLF_TRY(build_sample_matrix(ctx, tb.samples.p, p, coef, LA.p, lda, true, alpha, deg.p, shard_eig ? shard.row0 : 0u, la_cols, d_img,
                                        width, height, tb.idx.p));
    }
    GLF_TRY(dinv.alloc(ctx, p));
    hipLaunchKernelGGL(k_dinv_from_degree, dim3((p + 255) / 256), dim3(256), 0, st, deg.p, p, alpha, dinv.p);
    GLF_LAUNCH_CHECK(ctx);
    GLF_HIP(ctx, hipEventRecord(ctx->ev[2], st));
    // ---- eigenpairs --------------------------------------------------------------------------
    if (wide) {
        // Panels of 256 vectors: the eigen-solve once (eigen.hip "panels"), then Nystroem, Phi^T y and the filter's correction
        // one panel at a time -- only one [pixels][256] block of Phi exists at any moment.
        const unsigned npan = (unsigned)ceil_div(m, PANEL_COLS);
        const size_t pstride = (size_t)p32 * PANEL_COLS;
        DevBuf<float> vecs;
        GLF_TRY(vecs.alloc(ctx, pstride * npan));
        std::vector<double> lamw(m);
        int rcw = inverse_power_iteration_panels(ctx, LA.p, lda, p, m, nullptr, opt.seed, opt.opti_gs, opt.epsilon, opt.inner_rtol,
                                                 opt.max_outer > 0 ? opt.max_outer : 100000, vecs.p, lamw.data(), &S.eig,
                                                 shard.grid ? &shard : nullptr, dinv.p);
        if (rcw != GLF_OK) return rcw;
        LA.release();
        if (eigvals_out)
            for (unsigned j = 0; j < m; ++j) eigvals_out[j] = lamw[j];
        GLF_HIP(ctx, hipEventRecord(ctx->ev[3], st));
        const int64_t npixw = pix1 - pix0;
        DevBuf<float> psiw, pinvw, phiw, ww, acc;
        DevBuf<double> cw;
        const size_t psi_n = (size_t)round_up(p, NYS_PAD) * PANEL_COLS;
        GLF_TRY(psiw.alloc(ctx, psi_n));
        GLF_TRY(pinvw.alloc(ctx, PANEL_COLS));
        GLF_TRY(ww.alloc(ctx, PANEL_COLS));
        GLF_TRY(cw.alloc(ctx, PANEL_COLS));
        GLF_TRY(phiw.alloc(ctx, (size_t)npixw * PANEL_COLS));
        GLF_TRY(acc.alloc(ctx, (size_t)npixw));
        float *phi_basew = phiw.p - (size_t)pix0 * PANEL_COLS;
        float kms_total = 0.f;
        for (unsigned q = 0; q < npan; ++q) {
            const unsigned mq = std::min(PANEL_COLS, m - q * PANEL_COLS);
            const float *vq = vecs.p + q * pstride;
            std::vector<float> hp(PANEL_COLS, 0.f);
            for (unsigned j = 0; j < mq; ++j) hp[j] = (float)(1.0 / lamw[q * PANEL_COLS + j]); // InverseDiagMat, hpc/utils.c:559-586
            GLF_HIP(ctx, hipMemcpyAsync(pinvw.p, hp.data(), sizeof(float) * PANEL_COLS, hipMemcpyHostToDevice, st));
            GLF_HIP(ctx, hipStreamSynchronize(st));
            GLF_HIP(ctx, hipMemsetAsync(psiw.p, 0, sizeof(float) * psi_n, st));
            hipLaunchKernelGGL(k_make_psi, dim3((unsigned)ceil_div((int64_t)p * PANEL_COLS, 256)), dim3(256), 0, st, vq, pinvw.p, p, PANEL_COLS, mq,
                               (float)(-alpha), psiw.p);
            GLF_LAUNCH_CHECK(ctx);
            GLF_HIP(ctx, hipMemsetAsync(cw.p, 0, sizeof(double) * PANEL_COLS, st));
            float kms = 0.f;
            uint64_t evaluated = 0;
            GLF_TRY(nystroem_contract(ctx, d_img, width, height, pix0, pix1, tb.samples.p, tb.mask.p, tb.idx.p, p, coef, (float)(-alpha), psiw.p,
                                      mq, PANEL_COLS, phi_basew, 1, cw.p, &kms, opt.skip_exact_zeros, &evaluated, nullptr, &S.nystroem_path, nullptr));
            kms_total += kms;
            S.nystroem_evaluated += (double)evaluated;
            {
                unsigned i0 = 0, i1 = 0;
                while (i0 < p && (int64_t)h_idx[i0] < pix0) ++i0;
                i1 = i0;
                while (i1 < p && (int64_t)h_idx[i1] < pix1) ++i1;
                if (i1 > i0)
                    GLF_TRY(scatter_sample_rows(ctx, vq + (size_t)i0 * PANEL_COLS, i1 - i0, PANEL_COLS, tb.idx.p + i0, phi_basew, 1, d_img, cw.p, mq));
            }
            GLF_TRY(allreduce_f64(ctx, cw.p, PANEL_COLS)); // right = phi^T y over all ranks' pixels (this panel's columns)
            std::vector<double> hc(PANEL_COLS);
            GLF_HIP(ctx, hipMemcpyAsync(hc.data(), cw.p, sizeof(double) * PANEL_COLS, hipMemcpyDeviceToHost, st));
            GLF_HIP(ctx, hipStreamSynchronize(st));
            std::vector<float> hw(PANEL_COLS, 0.f);
            for (unsigned j = 0; j < mq; ++j)
                hw[j] = (float)(filter_weight(lamw[q * PANEL_COLS + j]) * hc[j]);
            GLF_HIP(ctx, hipMemcpyAsync(ww.p, hw.data(), sizeof(float) * PANEL_COLS, hipMemcpyHostToDevice, st));
            GLF_HIP(ctx, hipStreamSynchronize(st));
            GLF_TRY(filter_accumulate(ctx, phiw.p, pix0, pix1, PANEL_COLS, ww.p, acc.p, q == 0));
        }
        GLF_HIP(ctx, hipEventRecord(ctx->ev[4], st));
        GLF_TRY(filter_finish(ctx, d_img, acc.p, pix0, pix1, filter_gain, filter_ysub, d_out, d_zf));
        GLF_HIP(ctx, hipEventRecord(ctx->ev[5], st));
        GLF_HIP(ctx, hipEventSynchronize(ctx->ev[5]));
        S.nystroem_launches = (int)npan;
        S.nystroem_kernel_ms = kms_total;
        S.contraction = ctx->contraction;
        S.skip_exact_zeros = opt.skip_exact_zeros;
        GLF_HIP(ctx, hipEventElapsedTime(&S.ms_affinity, ctx->ev[0], ctx->ev[1]));
        GLF_HIP(ctx, hipEventElapsedTime(&S.ms_laplacian, ctx->ev[1], ctx->ev[2]));
        GLF_HIP(ctx, hipEventElapsedTime(&S.ms_eigen, ctx->ev[2], ctx->ev[3]));
        GLF_HIP(ctx, hipEventElapsedTime(&S.ms_nystroem, ctx->ev[3], ctx->ev[4]));
        GLF_HIP(ctx, hipEventElapsedTime(&S.ms_filter, ctx->ev[4], ctx->ev[5]));
        GLF_HIP(ctx, hipEventElapsedTime(&S.ms_total, ctx->ev[0], ctx->ev[5]));
        if (stats) *stats = S;
        return GLF_OK;
    }
    DevBuf<float> phiA;
    GLF_TRY(phiA.alloc(ctx, (size_t)p32 * ld));
    std::vector<double> lam(m);
    {
        const float *d_x0 = nullptr; // seeded start block (hpc/inverse_power_it.c:12-47), cached per (p, m, ld, seed)
        GLF_TRY(start_block_cached(ctx, p, m, ld, opt.seed, &d_x0));
        int rc = inverse_power_iteration(ctx, LA.p, lda, p, m, ld, nullptr, opt.opti_gs, opt.epsilon, opt.inner_rtol,
                                         opt.max_outer > 0 ? opt.max_outer : 100000, phiA.p, lam.data(), &S.eig,
                                         (shard_eig || shard.kbox || shard.grid) ? &shard : nullptr, dinv.p, d_x0);
        if (rc != GLF_OK) return rc;
    }
    if (shard_eig || gop.op) LA.release(); // (a whole stored L_A is used once more below: K_A y_A for the fused filter)
    if (cap && cap->d_phi_A) GLF_HIP(ctx, hipMemcpyAsync(cap->d_phi_A, phiA.p, sizeof(float) * (size_t)p32 * ld, hipMemcpyDeviceToDevice, st));
    if (eigvals_out)
        for (unsigned j = 0; j < m; ++j) eigvals_out[j] = lam[j];
    GLF_HIP(ctx, hipEventRecord(ctx->ev[3], st));
    // ---- Nystroem extension, written straight in raster order (Permutation folded in) ----------
    DevBuf<float> psi, pinv, phi, w;
    DevBuf<double> c;
    GLF_TRY(psi.alloc(ctx, (size_t)round_up(p, NYS_PAD) * ld));
    GLF_HIP(ctx, hipMemsetAsync(psi.p, 0, sizeof(float) * (size_t)round_up(p, NYS_PAD) * ld, st));
    GLF_TRY(pinv.alloc(ctx, ld));
    GLF_TRY(w.alloc(ctx, ld));
    GLF_TRY(c.alloc(ctx, ld));
    {
        std::vector<float> hp(ld, 0.f);
        for (unsigned j = 0; j < m; ++j) hp[j] = (float)(1.0 / lam[j]); // InverseDiagMat, hpc/utils.c:559-586
        GLF_HIP(ctx, hipMemcpyAsync(pinv.p, hp.data(), sizeof(float) * ld, hipMemcpyHostToDevice, st));
        GLF_HIP(ctx, hipStreamSynchronize(st));
    }
    hipLaunchKernelGGL(k_make_psi, dim3((unsigned)ceil_div((int64_t)p * ld, 256)), dim3(256), 0, st, phiA.p, pinv.p, p, ld, m,
                       (float)(-alpha), psi.p);
    GLF_LAUNCH_CHECK(ctx);
    const int64_t npix = pix1 - pix0;
    float kms = 0.f;
    uint64_t evaluated = 0;
    RowpassStats rps;
    auto nystroem_stats = [&]() {
        S.nystroem_rowpass_launches = rps.launches;
        S.nystroem_rowpass_ms = rps.ms;
        S.nystroem_rowpass_flops = rps.flops;
        S.nystroem_colpass_launches = rps.col_launches;
        S.nystroem_colpass_ms = rps.col_ms;
        S.nystroem_colpass_flops = rps.col_flops;
        S.rank_terms = rps.rank_R;
        S.nystroem_launches = 1;
        S.nystroem_kernel_ms = kms;
        S.contraction = ctx->contraction;
        S.skip_exact_zeros = opt.skip_exact_zeros;
        S.nystroem_evaluated = (double)evaluated; // kernel entries generated (listed chunks x 64 x workgroup pixels)
    };
    unsigned si0 = 0, si1 = 0; // the samples among this shard's pixels
    while (si0 < p && (int64_t)h_idx[si0] < pix0) ++si0;
    si1 = si0;
    while (si1 < p && (int64_t)h_idx[si1] < pix1) ++si1;
    // ---- band form with the filter in the kernel's epilogue: Phi is never written -----------------------------------------
    // c = Phi^T y is needed before the extension then: it follows from the degree stage's value-weighted sums (c_from_ysum).
    bool fused = false;
    if (have_ysum && (gop.op || LA.p) && ld <= 64 && opt.filter_mode != GLF_FILTER_SHARPEN && !(cap && cap->d_phi) &&
        ctx->contraction == GLF_CONTRACT_F16_SPLIT && (ctx->tune.nys_path == 0 || ctx->tune.nys_path == 4) && !ctx->tune.no_fused_filter) {
        // t = K_A y_A (the sample pixels' share of the sums over all pixels): one 32-column application of the operator
        DevBuf<float> ya, t;
        DevBuf<double> zdeg;
        GLF_TRY(ya.alloc(ctx, (size_t)p32 * 32));
        GLF_TRY(t.alloc(ctx, (size_t)p32 * 32));
        GLF_TRY(zdeg.alloc(ctx, p));
        GLF_HIP(ctx, hipMemsetAsync(ya.p, 0, sizeof(float) * (size_t)p32 * 32, st));
        GLF_HIP(ctx, hipMemsetAsync(t.p, 0, sizeof(float) * (size_t)p32 * 32, st));
        GLF_HIP(ctx, hipMemsetAsync(zdeg.p, 0, sizeof(double) * p, st));
        hipLaunchKernelGGL(k_sample_values_column, dim3((p + 255) / 256), dim3(256), 0, st, tb.samples.p, p, 32u, ya.p);
        GLF_LAUNCH_CHECK(ctx);
        if (gop.op) GLF_TRY(grid_op_apply(ctx, gop.op, ya.p, t.p, 32, -1.0, zdeg.p, 0, p, 0)); // -(0 X - K_A X) = K_A y_A
        else { // stored L_A = alpha (D - K_A): K_A y_A = D y_A - L_A y_A / alpha
            GLF_TRY(block_matvec(ctx, LA.p, lda, p, ya.p, t.p, 32, shard.kbox ? &shard : nullptr));
            hipLaunchKernelGGL(k_ka_from_la, dim3((p + 255) / 256), dim3(256), 0, st, tb.samples.p, deg.p, p, 32u, 1.0 / alpha, t.p);
            GLF_LAUNCH_CHECK(ctx);
        }
        GLF_TRY(c_from_ysum(ctx, psi.p, phiA.p, deg.p + p, t.p, 32, tb.samples.p, p, ld, c.p));
        std::vector<double> hc(ld);
        GLF_HIP(ctx, hipMemcpyAsync(hc.data(), c.p, sizeof(double) * ld, hipMemcpyDeviceToHost, st));
        GLF_HIP(ctx, hipStreamSynchronize(st));
        if (cap && cap->h_c) std::memcpy(cap->h_c, hc.data(), sizeof(double) * ld);
        std::vector<float> hw(ld, 0.f);
        for (unsigned j = 0; j < m; ++j) hw[j] = (float)(filter_weight(lam[j]) * hc[j]);
        GLF_HIP(ctx, hipMemcpyAsync(w.p, hw.data(), sizeof(float) * ld, hipMemcpyHostToDevice, st));
        GLF_HIP(ctx, hipStreamSynchronize(st));
        BandFilter flt;
        flt.w = w.p;
        flt.gain = filter_gain;
        flt.ysub = filter_ysub;
        flt.out = d_out;
        flt.zf = d_zf;
        flt.corr = cap ? cap->d_corr : nullptr;
        // the sample pixels from their rows of Phi_A first (k_band leaves them alone: the host work between the two launches --
        // grid detection, cached tables -- then overlaps a kernel instead of following the long one)
        GLF_TRY(filter_sample_rows(ctx, phiA.p + (size_t)si0 * ld, si1 - si0, ld, tb.idx.p + si0, d_img, w.p, filter_gain, filter_ysub, d_out, d_zf,
                                   cap ? cap->d_corr : nullptr, pix0));
        const int rc = nystroem_band_filter(ctx, d_img, width, height, pix0, pix1, tb.samples.p, tb.mask.p, tb.idx.p, p, coef, psi.p, ld, flt, &kms,
                                            &evaluated, &S.nystroem_mfma_flops, &S.nystroem_path, &rps, h_idx);
        if (rc == GLF_OK) {
            nystroem_stats();
            S.filter_fused = 1;
            GLF_HIP(ctx, hipEventRecord(ctx->ev[4], st));
            fused = true;
        } else if (rc != GLF_ERR_UNSUPPORTED) return rc;
    }
    LA.release();
    if (!fused) {
    GLF_TRY(phi.alloc(ctx, (size_t)npix * ld));
    float *phi_base = phi.p - (size_t)pix0 * ld; // rows addressed by absolute pixel index
    GLF_HIP(ctx, hipMemsetAsync(c.p, 0, sizeof(double) * ld, st));
    GLF_TRY(nystroem_contract(ctx, d_img, width, height, pix0, pix1, tb.samples.p, tb.mask.p, tb.idx.p, p, coef, (float)(-alpha),
                              psi.p, m, ld, phi_base, 1, c.p, &kms, opt.skip_exact_zeros, &evaluated, &S.nystroem_mfma_flops,
                              &S.nystroem_path, &rps));
    nystroem_stats();
    // sample rows of this shard <- Phi_A, and their share of c
    if (si1 > si0)
        GLF_TRY(scatter_sample_rows(ctx, phiA.p + (size_t)si0 * ld, si1 - si0, ld, tb.idx.p + si0, phi_base, 1, d_img, c.p, m));
    GLF_TRY(allreduce_f64(ctx, c.p, ld)); // right = phi^T y over all ranks' pixels
    GLF_HIP(ctx, hipEventRecord(ctx->ev[4], st));
    // ---- filter ------------------------------------------------------------------------------
    {
        std::vector<double> hc(ld);
        GLF_HIP(ctx, hipMemcpyAsync(hc.data(), c.p, sizeof(double) * ld, hipMemcpyDeviceToHost, st));
        if (cap && cap->d_phi) GLF_HIP(ctx, hipMemcpyAsync(cap->d_phi, phi.p, sizeof(float) * (size_t)npix * ld, hipMemcpyDeviceToDevice, st));
        GLF_HIP(ctx, hipStreamSynchronize(st));
        if (cap && cap->h_c) std::memcpy(cap->h_c, hc.data(), sizeof(double) * ld);
        std::vector<float> hw(ld, 0.f);
        for (unsigned j = 0; j < m; ++j) hw[j] = (float)(filter_weight(lam[j]) * hc[j]);
        if (opt.filter_mode == GLF_FILTER_SHARPEN) {
            // (1 + beta) W^2 y - beta W^3 y with W = Phi L Phi^T applied factor by factor as the PoC does
            // (python/image_processing.py:231-235): the extended eigenvectors are not orthonormal, so G = Phi^T Phi sits
            // between the factors -- z = Phi w, w = (1 + beta) L G L c - beta L G L G L c, L = 1 - mu, c = Phi^T y
            DevBuf<double> G;
            GLF_TRY(G.alloc(ctx, (size_t)ld * ld));
            GLF_TRY(phi_gram(ctx, phi_base, pix0, pix1, ld, G.p));
            GLF_TRY(allreduce_f64(ctx, G.p, (size_t)ld * ld));
            std::vector<double> hG((size_t)ld * ld), t(m), u(m), v(m);
            GLF_HIP(ctx, hipMemcpyAsync(hG.data(), G.p, sizeof(double) * ld * ld, hipMemcpyDeviceToHost, st));
            GLF_HIP(ctx, hipStreamSynchronize(st));
            const double beta = (double)opt.filter_beta;
            auto LG = [&](const std::vector<double> &x, std::vector<double> &y) { // y = L (G x)
                for (unsigned i = 0; i < m; ++i) {
                    double a = 0.0;
                    for (unsigned j = 0; j < m; ++j) a += hG[(size_t)i * ld + j] * x[j];
                    y[i] = (1.0 - lam[i]) * a;
                }
            };
            for (unsigned j = 0; j < m; ++j) t[j] = (1.0 - lam[j]) * hc[j];
            LG(t, u);
            LG(u, v);
            for (unsigned j = 0; j < m; ++j) hw[j] = (float)((1.0 + beta) * u[j] - beta * v[j]);
        }
        GLF_HIP(ctx, hipMemcpyAsync(w.p, hw.data(), sizeof(float) * ld, hipMemcpyHostToDevice, st));
        GLF_HIP(ctx, hipStreamSynchronize(st));
    }
    GLF_TRY(apply_filter(ctx, d_img, phi_base, pix0, pix1, m, ld, w.p, filter_gain, filter_ysub, d_out, d_zf, cap ? cap->d_corr : nullptr));
    } // (!fused)
    GLF_HIP(ctx, hipEventRecord(ctx->ev[5], st));
    GLF_HIP(ctx, hipEventSynchronize(ctx->ev[5]));
    GLF_HIP(ctx, hipEventElapsedTime(&S.ms_affinity, ctx->ev[0], ctx->ev[1]));
    GLF_HIP(ctx, hipEventElapsedTime(&S.ms_laplacian, ctx->ev[1], ctx->ev[2]));
    GLF_HIP(ctx, hipEventElapsedTime(&S.ms_eigen, ctx->ev[2], ctx->ev[3]));
    GLF_HIP(ctx, hipEventElapsedTime(&S.ms_nystroem, ctx->ev[3], ctx->ev[4]));
    GLF_HIP(ctx, hipEventElapsedTime(&S.ms_filter, ctx->ev[4], ctx->ev[5]));
    GLF_HIP(ctx, hipEventElapsedTime(&S.ms_total, ctx->ev[0], ctx->ev[5]));
    if (stats) *stats = S;
    return GLF_OK;
}

} // extern "C"
