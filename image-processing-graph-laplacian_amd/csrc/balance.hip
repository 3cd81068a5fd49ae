// balance.hip -- the PoC's balancing steps of the approximated affinity (python/image_processing.py):
//   glf_Sinkhorn          :90-107   alternating scalings r, c of K = Phi diag(Pi) Phi^T, K never formed: every one of the
//                                   200 products K x is Phi (Pi o (Phi^T x)) -- one reduction over the rows of Phi to m
//                                   numbers and one row-wise dot product; rows of W_AB = diag(r) K diag(c) on demand
//   glf_Orthogonalisation :110-127  V = [A ; B^T] A^-1/2 phi_Q Pi_Q^-1/2, Q = A + A^-1/2 B B^T A^-1/2 for a dense symmetric
//                                   A (n x n) and a dense B (n x q): two symmetric eigen-decompositions (cyclic Jacobi in f64,
//                                   one workgroup, n <= 512) and a handful of f64 products
// Both are inactive experiments in the PoC (commented out at their call sites, :171-186) and dense by construction there
// (W_AB is n x N); they are here for completeness of the f4 row, in f64 throughout, not tuned: coverage, not speed.
#include "glf_internal.hpp"

#include <algorithm>
#include <cmath>
#include <vector>

namespace glf {

// partial[blk][j] = sum over the rows of block blk of Phi[row][j] x[row] (f64)
__global__ __launch_bounds__(256) void k_phi_t_vec(const float *__restrict__ phi, const double *__restrict__ x, int64_t nrows, unsigned ld,
                                                    double *__restrict__ partial)
{
    __shared__ double sh[256];
    const int col = threadIdx.x % ld, rl = threadIdx.x / ld, nrl = 256 / ld;
    double s = 0.0;
    for (int64_t r = (int64_t)blockIdx.x * 1024 + rl; r < min(nrows, ((int64_t)blockIdx.x + 1) * 1024); r += nrl)
        s += (double)phi[(size_t)r * ld + col] * x[r];
    sh[threadIdx.x] = s;
    __syncthreads();
    if (threadIdx.x < ld) {
        double t = 0.0;
        for (int q = 0; q < nrl; ++q) t += sh[q * ld + col];
        partial[(size_t)blockIdx.x * ld + col] = t;
    }
}

__global__ void k_sum_blocks(const double *__restrict__ partial, int nblk, unsigned ld, const float *__restrict__ pi, unsigned m,
                             double *__restrict__ t)
{
    // t[j] = Pi[j] * sum over the blocks (fixed order); 0 beyond m
    const unsigned j = threadIdx.x;
    if (j >= ld) return;
    double s = 0.0;
    for (int b = 0; b < nblk; ++b) s += partial[(size_t)b * ld + j];
    t[j] = j < m ? (double)pi[j] * s : 0.0;
}

// out[row] = nan_to_num(1 / (Phi[row] . t)): the PoC's np.nan_to_num(1. / x) -- 1/0 = inf becomes the largest finite double
__global__ __launch_bounds__(256) void k_phi_vec_recip(const float *__restrict__ phi, const double *__restrict__ t, int64_t nrows, unsigned ld,
                                                        unsigned m, double *__restrict__ out)
{
    const int64_t r = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (r >= nrows) return;
    double s = 0.0;
    for (unsigned j = 0; j < m; ++j) s += (double)phi[(size_t)r * ld + j] * t[j];
    double v = 1.0 / s;
    if (v != v) v = 0.0;
    else if (v > 1.7976931348623157e308) v = 1.7976931348623157e308;
    else if (v < -1.7976931348623157e308) v = -1.7976931348623157e308;
    out[r] = v;
}

__global__ __launch_bounds__(256) void k_fill_f64(double *__restrict__ x, int64_t n, double v)
{
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n) x[i] = v;
}

// W_AB[i][j] = r[row0 + i] c[j] sum_k Phi[row0 + i][k] Pi[k] Phi[j][k]   (python/image_processing.py:99-102)
__global__ __launch_bounds__(256) void k_sinkhorn_rows(const float *__restrict__ phi, const float *__restrict__ pi, const double *__restrict__ r,
                                                        const double *__restrict__ c, int64_t nrows, unsigned ld, unsigned m, int64_t row0,
                                                        double *__restrict__ out)
{
    const int64_t j = (int64_t)blockIdx.x * 256 + threadIdx.x, i = row0 + blockIdx.y;
    if (j >= nrows) return;
    double s = 0.0;
    for (unsigned k = 0; k < m; ++k) s += (double)phi[(size_t)i * ld + k] * (double)pi[k] * (double)phi[(size_t)j * ld + k];
    out[(size_t)blockIdx.y * nrows + j] = r[i] * c[j] * s;
}

// ---- dense f64 helpers of the orthogonalisation ---------------------------------------------------------------------------
// C[i][j] = sum_k opA(A)[i][k] opB(B)[k][j]; row-major, leading dimensions lda / ldb / ldc; one thread per entry
__global__ __launch_bounds__(256) void k_gemm_f64(const double *__restrict__ A, int lda, int ta, const double *__restrict__ B, int ldb, int tb,
                                                   int M, int N, int K, double *__restrict__ C, int ldc)
{
    const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (e >= (int64_t)M * N) return;
    const int i = (int)(e / N), j = (int)(e % N);
    double s = 0.0;
    for (int k = 0; k < K; ++k) s += (ta ? A[(size_t)k * lda + i] : A[(size_t)i * lda + k]) * (tb ? B[(size_t)j * ldb + k] : B[(size_t)k * ldb + j]);
    C[(size_t)i * ldc + j] = s;
}

// Cyclic Jacobi eigen-decomposition of a symmetric n x n matrix (f64, in global memory), one workgroup: per round n / 2
// disjoint pairs (p, q) from the round-robin tournament are annihilated together -- rotations from the current entries, rows
// p, q of A for all pairs, then columns p, q of A and of the accumulated vectors V. A is overwritten (diagonal = eigenvalues),
// V holds the eigenvectors in its columns.
__global__ __launch_bounds__(1024) void k_jacobi_eigh(double *__restrict__ A, double *__restrict__ V, int n, int sweeps)
{
    __shared__ int pp[256], qq[256];
    __shared__ double cs[256], sn[256];
    __shared__ double off2, diag2;
    const int t = threadIdx.x, T = blockDim.x;
    for (int e = t; e < n * n; e += T) V[e] = (e / n == e % n) ? 1.0 : 0.0;
    const int ne = (n + 1) & ~1, npairs = ne / 2; // players 0 .. ne - 1 (ne - 1 = a bye when n is odd)
    __syncthreads();
    for (int sweep = 0; sweep < sweeps; ++sweep) {
        if (t == 0) off2 = diag2 = 0.0;
        __syncthreads();
        {
            double o = 0.0, d = 0.0;
            for (int e = t; e < n * n; e += T) {
                const double a = A[e];
                if (e / n == e % n) d += a * a;
                else o += a * a;
            }
            atomicAdd(&off2, o);
            atomicAdd(&diag2, d);
        }
        __syncthreads();
        const bool done = off2 <= 1e-30 * diag2; // (uniform: shared values)
        __syncthreads();
        if (done) break;
        for (int round = 0; round < ne - 1; ++round) {
            if (t < npairs) { // circle method: player ne - 1 fixed, the others rotate
                int a = t == 0 ? ne - 1 : (round + t) % (ne - 1), b = (round + ne - 1 - t) % (ne - 1);
                if (a > b) {
                    const int x = a;
                    a = b;
                    b = x;
                }
                double c = 1.0, s = 0.0;
                if (b < n && a != b) {
                    const double apq = A[(size_t)a * n + b];
                    if (apq != 0.0) {
                        const double theta = (A[(size_t)b * n + b] - A[(size_t)a * n + a]) / (2.0 * apq);
                        const double tt = (theta >= 0.0 ? 1.0 : -1.0) / (fabs(theta) + sqrt(theta * theta + 1.0));
                        c = 1.0 / sqrt(tt * tt + 1.0);
                        s = tt * c;
                    }
                } else {
                    b = -1; // bye
                }
                pp[t] = a;
                qq[t] = b;
                cs[t] = c;
                sn[t] = s;
            }
            __syncthreads();
            for (int e = t; e < npairs * n; e += T) { // rows p, q of A
                const int k = e / n, j = e % n, p = pp[k], q = qq[k];
                if (q < 0) continue;
                const double ap = A[(size_t)p * n + j], aq = A[(size_t)q * n + j];
                A[(size_t)p * n + j] = cs[k] * ap - sn[k] * aq;
                A[(size_t)q * n + j] = sn[k] * ap + cs[k] * aq;
            }
            __syncthreads();
            for (int e = t; e < npairs * n; e += T) { // columns p, q of A and of V
                const int k = e / n, i = e % n, p = pp[k], q = qq[k];
                if (q < 0) continue;
                const double ap = A[(size_t)i * n + p], aq = A[(size_t)i * n + q];
                A[(size_t)i * n + p] = cs[k] * ap - sn[k] * aq;
                A[(size_t)i * n + q] = sn[k] * ap + cs[k] * aq;
                const double vp = V[(size_t)i * n + p], vq = V[(size_t)i * n + q];
                V[(size_t)i * n + p] = cs[k] * vp - sn[k] * vq;
                V[(size_t)i * n + q] = sn[k] * vp + cs[k] * vq;
            }
            __syncthreads();
        }
    }
}

// eigenpairs of the symmetric d_A (n x n f64, destroyed) in descending order: h_w[n], d_V columns
static int eigh_desc(glf_ctx *ctx, double *d_A, int n, std::vector<double> &h_w, double *d_V)
{
    hipStream_t st = ctx->stream;
    DevBuf<double> V0;
    GLF_TRY(V0.alloc(ctx, (size_t)n * n));
    hipLaunchKernelGGL(k_jacobi_eigh, dim3(1), dim3(1024), 0, st, d_A, V0.p, n, 60);
    GLF_LAUNCH_CHECK(ctx);
    std::vector<double> hA((size_t)n * n), hV((size_t)n * n), hV2((size_t)n * n);
    GLF_HIP(ctx, hipMemcpyAsync(hA.data(), d_A, sizeof(double) * n * n, hipMemcpyDeviceToHost, st));
    GLF_HIP(ctx, hipMemcpyAsync(hV.data(), V0.p, sizeof(double) * n * n, hipMemcpyDeviceToHost, st));
    GLF_HIP(ctx, hipStreamSynchronize(st));
    std::vector<int> order(n);
    for (int i = 0; i < n; ++i) order[i] = i;
    std::sort(order.begin(), order.end(), [&](int a, int b) { return hA[(size_t)a * n + a] > hA[(size_t)b * n + b]; });
    h_w.resize(n);
    for (int k = 0; k < n; ++k) { // (a column permutation: bookkeeping, not arithmetic)
        h_w[k] = hA[(size_t)order[k] * n + order[k]];
        for (int i = 0; i < n; ++i) hV2[(size_t)i * n + k] = hV[(size_t)i * n + order[k]];
    }
    GLF_HIP(ctx, hipMemcpyAsync(d_V, hV2.data(), sizeof(double) * n * n, hipMemcpyHostToDevice, st));
    GLF_HIP(ctx, hipStreamSynchronize(st));
    return GLF_OK;
}

static void gemm(hipStream_t st, const double *A, int lda, int ta, const double *B, int ldb, int tb, int M, int N, int K, double *C, int ldc)
{
    hipLaunchKernelGGL(k_gemm_f64, dim3((unsigned)ceil_div((int64_t)M * N, 256)), dim3(256), 0, st, A, lda, ta, B, ldb, tb, M, N, K, C, ldc);
}

__global__ __launch_bounds__(256) void k_scale_cols_f64(double *__restrict__ X, int rows, int cols, int ld, const double *__restrict__ s)
{
    const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (e < (int64_t)rows * cols) X[(size_t)(e / cols) * ld + e % cols] *= s[e % cols];
}
__global__ __launch_bounds__(256) void k_add_f64(double *__restrict__ X, const double *__restrict__ Y, int64_t n)
{
    const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (e < n) X[e] += Y[e];
}

} // namespace glf

using namespace glf;

extern "C" {

int glf_Sinkhorn(glf_ctx *ctx, const glf_mat *phi, const glf_mat *Pi, int iterations, double *d_r, double *d_c)
{
    if (!ctx || !phi || !Pi || !d_r || !d_c || iterations < 1) return GLF_ERR_INVALID;
    GLF_ENTER(ctx);
    if (phi->kind != GLF_MAT_DENSE || Pi->kind != GLF_MAT_DIAG || Pi->rows != phi->cols || !valid_ld((unsigned)phi->ld))
        return set_error(ctx, GLF_ERR_INVALID, "Sinkhorn: phi dense N x m (ld 32 .. 256), Pi diagonal m");
    hipStream_t st = ctx->stream;
    const int64_t N = phi->rows;
    const unsigned ld = (unsigned)phi->ld, m = (unsigned)phi->cols;
    const int nblk = (int)ceil_div(N, 1024);
    DevBuf<double> part, t;
    GLF_TRY(part.alloc(ctx, (size_t)nblk * ld));
    GLF_TRY(t.alloc(ctx, ld));
    hipLaunchKernelGGL(k_fill_f64, dim3((unsigned)ceil_div(N, 256)), dim3(256), 0, st, d_r, N, 1.0);
    auto scale = [&](const double *x, double *out) { // out = nan_to_num(1 / (Phi (Pi o (Phi^T x))))
        hipLaunchKernelGGL(k_phi_t_vec, dim3(nblk), dim3(256), 0, st, phi->data, x, N, ld, part.p);
        hipLaunchKernelGGL(k_sum_blocks, dim3(1), dim3(256), 0, st, part.p, nblk, ld, Pi->data, m, t.p);
        hipLaunchKernelGGL(k_phi_vec_recip, dim3((unsigned)ceil_div(N, 256)), dim3(256), 0, st, phi->data, t.p, N, ld, m, out);
    };
    for (int it = 0; it < iterations; ++it) { // python/image_processing.py:94-98
        scale(d_r, d_c);
        scale(d_c, d_r);
    }
    GLF_LAUNCH_CHECK(ctx);
    GLF_HIP(ctx, hipStreamSynchronize(st));
    return GLF_OK;
}

int glf_SinkhornRows(glf_ctx *ctx, const glf_mat *phi, const glf_mat *Pi, const double *d_r, const double *d_c, int64_t row0, int nrows,
                     double *d_out)
{
    if (!ctx || !phi || !Pi || !d_r || !d_c || !d_out || nrows < 1 || row0 < 0) return GLF_ERR_INVALID;
    GLF_ENTER(ctx);
    if (phi->kind != GLF_MAT_DENSE || Pi->kind != GLF_MAT_DIAG || Pi->rows != phi->cols || row0 + nrows > phi->rows) return GLF_ERR_INVALID;
    hipLaunchKernelGGL(k_sinkhorn_rows, dim3((unsigned)ceil_div(phi->rows, 256), (unsigned)nrows), dim3(256), 0, ctx->stream, phi->data, Pi->data,
                       d_r, d_c, phi->rows, (unsigned)phi->ld, (unsigned)phi->cols, row0, d_out);
    GLF_LAUNCH_CHECK(ctx);
    GLF_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return GLF_OK;
}

int glf_Orthogonalisation(glf_ctx *ctx, const double *d_A, int n, const double *d_B, int q, double *d_V, double *h_Pi)
{
    if (!ctx || !d_A || !d_B || !d_V || !h_Pi || n < 1 || q < 0) return GLF_ERR_INVALID;
    GLF_ENTER(ctx);
    if (n > 512) return set_error(ctx, GLF_ERR_UNSUPPORTED, "Orthogonalisation: n = %d (the dense Jacobi eigensolver takes n <= 512)", n);
    hipStream_t st = ctx->stream;
    const size_t nn = (size_t)n * n;
    DevBuf<double> W, P, S, BBt, T1, Q, PQ, Mx;
    for (DevBuf<double> *b : {&W, &P, &S, &BBt, &T1, &Q, &PQ, &Mx}) GLF_TRY(b->alloc(ctx, nn));
    DevBuf<double> sc;
    GLF_TRY(sc.alloc(ctx, n));
    std::vector<double> w, hs(n);
    // phi, Pi = svd(A): A is symmetric positive semi-definite, its SVD is its eigen-decomposition           (:113)
    GLF_HIP(ctx, hipMemcpyAsync(W.p, d_A, sizeof(double) * nn, hipMemcpyDeviceToDevice, st));
    GLF_TRY(eigh_desc(ctx, W.p, n, w, P.p));
    // A_sqrt_inv = (phi / sqrt(Pi)) phi^T                                                                     (:114-115)
    for (int k = 0; k < n; ++k) hs[k] = 1.0 / std::sqrt(w[k]);
    GLF_HIP(ctx, hipMemcpyAsync(sc.p, hs.data(), sizeof(double) * n, hipMemcpyHostToDevice, st));
    GLF_HIP(ctx, hipMemcpyAsync(T1.p, P.p, sizeof(double) * nn, hipMemcpyDeviceToDevice, st));
    hipLaunchKernelGGL(k_scale_cols_f64, dim3((unsigned)ceil_div((int64_t)nn, 256)), dim3(256), 0, st, T1.p, n, n, n, sc.p);
    gemm(st, T1.p, n, 0, P.p, n, 1, n, n, n, S.p, n);
    // Q = A + A_sqrt_inv B B^T A_sqrt_inv                                                                      (:117)
    gemm(st, d_B, q, 0, d_B, q, 1, n, n, q, BBt.p, n);
    gemm(st, S.p, n, 0, BBt.p, n, 0, n, n, n, T1.p, n);
    gemm(st, T1.p, n, 0, S.p, n, 0, n, n, n, Q.p, n);
    hipLaunchKernelGGL(k_add_f64, dim3((unsigned)ceil_div((int64_t)nn, 256)), dim3(256), 0, st, Q.p, d_A, (int64_t)nn);
    GLF_LAUNCH_CHECK(ctx);
    GLF_HIP(ctx, hipStreamSynchronize(st)); // hs is reused below
    // phi_Q, Pi_Q = svd(Q)                                                                                     (:118)
    GLF_TRY(eigh_desc(ctx, Q.p, n, w, PQ.p));
    // V = [A ; B^T] A_sqrt_inv phi_Q Pi_Q^-1/2                                                                  (:120-122)
    for (int k = 0; k < n; ++k) hs[k] = 1.0 / std::sqrt(w[k]);
    GLF_HIP(ctx, hipMemcpyAsync(sc.p, hs.data(), sizeof(double) * n, hipMemcpyHostToDevice, st));
    gemm(st, S.p, n, 0, PQ.p, n, 0, n, n, n, Mx.p, n);
    hipLaunchKernelGGL(k_scale_cols_f64, dim3((unsigned)ceil_div((int64_t)nn, 256)), dim3(256), 0, st, Mx.p, n, n, n, sc.p);
    gemm(st, d_A, n, 0, Mx.p, n, 0, n, n, n, d_V, n);
    if (q) gemm(st, d_B, q, 1, Mx.p, n, 0, q, n, n, d_V + nn, n);
    GLF_LAUNCH_CHECK(ctx);
    GLF_HIP(ctx, hipStreamSynchronize(st));
    for (int k = 0; k < n; ++k) h_Pi[k] = std::min(w[k], 1.0); // Pi[Pi > 1] = 1                                (:123-124)
    return GLF_OK;
}

} // extern "C"
