/*
 * glf_oracle.c -- TEST INFRASTRUCTURE ONLY (see glf_oracle.h for the rules).
 *
 * fp64 restatement of the reference's approximate path. Citations are
 * file:line in the reference tree (hpc/...).
 */
#include "glf_oracle.h"

#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>
#ifdef _OPENMP
#include <omp.h>
#endif

static double now_s(void)
{
    struct timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec;
}

int orc_num_threads(void)
{
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}

void orc_default_params(orc_params *prm)
{
    prm->h_loc = 40.0; /* hpc/affinity.c:118 */
    prm->h_val = 30.0; /* hpc/affinity.c:117 */
    prm->kernel = ORC_BILATERAL; /* hpc/affinity.c:121 */
}

void orc_free(void *ptr) { free(ptr); }

/* ---------------------------------------------------------------- sampling */

/* hpc/sampling.c:6-23. The arithmetic keeps the reference's C types: the
 * quotient is an unsigned integer division (int * int converted to unsigned by
 * the unsigned divisor) before sqrt (survey quirk Q15); the last row and column
 * are never sampled (i < height-1, quirk Q16). */
int orc_sampling(int width, int height, unsigned *sample_size, unsigned **sample_indices)
{
    if (width <= 0 || height <= 0 || !sample_size || *sample_size == 0) return -1;
    const unsigned quotient = (unsigned)(width * height) / (*sample_size);
    const unsigned sample_dist = (unsigned)sqrt((double)quotient);
    if (sample_dist == 0) return -1;
    const unsigned xy0 = sample_dist / 2;
    const unsigned size_x_span = (unsigned)ceil(((unsigned)height - 1u - xy0) / (double)sample_dist);
    const unsigned size_y_span = (unsigned)ceil(((unsigned)width - 1u - xy0) / (double)sample_dist);
    *sample_size = size_x_span * size_y_span;
    *sample_indices = (unsigned *)malloc(sizeof(unsigned) * (size_t)(*sample_size ? *sample_size : 1));
    if (!*sample_indices) return -1;
    unsigned c = 0;
    for (unsigned i = xy0; i < (unsigned)height - 1u; i += sample_dist)
        for (unsigned j = xy0; j < (unsigned)width - 1u; j += sample_dist)
            (*sample_indices)[c++] = (unsigned)width * i + j;
    return (c == *sample_size) ? 0 : -1;
}

/* ---------------------------------------------------------------- affinity */

/* hpc/affinity.c:59-113 (bilateral), :8-17 (photometric), :19-57 (spatial).
 * Same operation order as the Vec pipeline: shift, square, add, scale by
 * -(1/h^2), exp; two exps multiplied for the bilateral kernel. */
double orc_kernel_entry(const orc_params *prm, double r0, double c0, double v0,
                        double r1, double c1, double v1)
{
    const double dx = r1 - r0, dy = c1 - c0;
    const double loc = exp(-(1. / (prm->h_loc * prm->h_loc)) * (dx * dx + dy * dy));
    const double dv = fabs(v1 - v0);
    const double val = exp(-(1. / (prm->h_val * prm->h_val)) * (dv * dv));
    switch (prm->kernel) {
    case ORC_PHOTOMETRIC: return val;
    case ORC_SPATIAL: return loc;
    default: return loc * val;
    }
}

/* ---- non-local-means kernel (python/affinity_methods/NLM.py:9-34; the C reference has no NLM) --------------------
 * K(i, j) = exp(-|| G o (patch_i - patch_j) ||^2 / h^2): 7 x 7 patches of the symmetrically padded image (np.pad 'symmetric',
 * :16), G the 7 x 7 Gaussian mask of sigma 1.2 normalised to sum 1 (matlab_style_gauss2D + :18-19) multiplying the patch VALUES
 * (:21-22, :29), h = prm->h_val (the PoC fixes h = 3, :12). The PoC lays its kernel row out with the pixel (row j, col i) at
 * column i*M + j (im2col of the TRANSPOSED image, :21) and then indexes it with raster indices (python/image_processing.py:
 * 59-64) -- consistent only where row == col; here pixel indices are raster indices throughout, and tools/gen_golden_nlm.py
 * maps the PoC's columns back before comparing. */
#define ORC_NLM_R 3
#define ORC_NLM_K 49
static void nlm_mask(double *G)
{
    double sum = 0.0, mx = 0.0;
    for (int a = -ORC_NLM_R; a <= ORC_NLM_R; ++a)
        for (int b = -ORC_NLM_R; b <= ORC_NLM_R; ++b) {
            const double g = exp(-(double)(a * a + b * b) / (2. * 1.2 * 1.2));
            G[(a + ORC_NLM_R) * 7 + (b + ORC_NLM_R)] = g;
            if (g > mx) mx = g;
        }
    for (int k = 0; k < ORC_NLM_K; ++k) { /* h[h < eps * max] = 0, then h /= sum (python/utils.py:24-29) */
        if (G[k] < 2.220446049250313e-16 * mx) G[k] = 0.0;
        sum += G[k];
    }
    for (int k = 0; k < ORC_NLM_K; ++k) G[k] /= sum;
    sum = 0.0;
    for (int k = 0; k < ORC_NLM_K; ++k) sum += G[k]; /* G / np.sum(G), NLM.py:19 */
    for (int k = 0; k < ORC_NLM_K; ++k) G[k] /= sum;
}
static int reflect_index(int i, int n) { return i < 0 ? -i - 1 : (i >= n ? 2 * n - i - 1 : i); } /* np.pad 'symmetric' */

/* F[px][k] = G[k] * padded(r + a, c + b), k = 7 (a + 3) + (b + 3) */
static double *nlm_features(const uint8_t *img, int width, int height)
{
    const size_t N = (size_t)width * (size_t)height;
    double G[ORC_NLM_K];
    nlm_mask(G);
    double *F = (double *)malloc(sizeof(double) * N * ORC_NLM_K);
    if (!F) return NULL;
#pragma omp parallel for schedule(static)
    for (long px = 0; px < (long)N; ++px) {
        const int r = (int)((size_t)px / (size_t)width), c = (int)((size_t)px % (size_t)width);
        for (int a = -ORC_NLM_R; a <= ORC_NLM_R; ++a)
            for (int b = -ORC_NLM_R; b <= ORC_NLM_R; ++b) {
                const int k = (a + ORC_NLM_R) * 7 + (b + ORC_NLM_R);
                F[(size_t)px * ORC_NLM_K + k] =
                    G[k] * (double)img[(size_t)reflect_index(r + a, height) * width + reflect_index(c + b, width)];
            }
    }
    return F;
}

/* One kernel entry between two PIXELS of an image, whatever the kernel: the positional kernels go through
 * orc_kernel_entry (same doubles as before), NLM through the patch features. */
typedef struct {
    const orc_params *prm;
    const uint8_t *img;
    int width;
    double *F; /* NLM patch features, NULL otherwise */
} orc_pairs;
static int pairs_init(orc_pairs *pc, const orc_params *prm, const uint8_t *img, int width, int height)
{
    pc->prm = prm;
    pc->img = img;
    pc->width = width;
    pc->F = NULL;
    if (prm->kernel == ORC_NLM) {
        if (height <= 0) return -1;
        pc->F = nlm_features(img, width, height);
        if (!pc->F) return -1;
    }
    return 0;
}
static void pairs_free(orc_pairs *pc) { free(pc->F); pc->F = NULL; }
static inline double pair_entry(const orc_pairs *pc, size_t a, size_t b)
{
    if (pc->F) {
        const double *fa = pc->F + a * ORC_NLM_K, *fb = pc->F + b * ORC_NLM_K;
        double d = 0.0;
        for (int k = 0; k < ORC_NLM_K; ++k) d += (fa[k] - fb[k]) * (fa[k] - fb[k]);
        return exp(-d / (pc->prm->h_val * pc->prm->h_val)); /* NLM.py:31 */
    }
    const unsigned w = (unsigned)pc->width;
    return orc_kernel_entry(pc->prm, (double)(a / w), (double)(a % w), (double)pc->img[a], (double)(b / w), (double)(b % w),
                            (double)pc->img[b]);
}

static uint8_t *build_sample_mask(size_t N, unsigned p, const unsigned *idx)
{
    uint8_t *mask = (uint8_t *)calloc(N, 1);
    if (!mask) return NULL;
    for (unsigned i = 0; i < p; ++i) {
        if (idx[i] >= N) { free(mask); return NULL; }
        mask[idx[i]] = 1;
    }
    return mask;
}

/* hpc/affinity.c:129-262 */
int orc_affinity(const orc_params *prm, const uint8_t *img, int width, int height,
                 unsigned p, const unsigned *idx, double *K_A, double *K_B)
{
    const size_t N = (size_t)width * (size_t)height;
    /* sample (x, y, value) vectors, hpc/affinity.c:152-178: num2x / num2y (hpc/utils.c:11-19) inside pair_entry */
    orc_pairs pc;
    if (pairs_init(&pc, prm, img, width, height) != 0) return -1;
    if (K_A) {
#pragma omp parallel for schedule(static)
        for (long i = 0; i < (long)p; ++i) /* hpc/affinity.c:181-193 */
            for (unsigned j = 0; j < p; ++j) K_A[(size_t)i * p + j] = pair_entry(&pc, idx[i], idx[j]);
    }
    if (K_B) {
        /* remaining pixels in raster order, hpc/affinity.c:215-235; the
         * tmp_idx < p guard is survey quirk Q2 (OOB read at :222). */
        const size_t R = N - p;
        unsigned *rem = (unsigned *)malloc(sizeof(unsigned) * (R ? R : 1));
        if (!rem) { pairs_free(&pc); return -1; }
        size_t tmp_idx = 0, k = 0;
        for (size_t j = 0; j < N; ++j) {
            if (tmp_idx < p && j == idx[tmp_idx]) ++tmp_idx;
            else rem[k++] = (unsigned)j;
        }
#pragma omp parallel for schedule(static)
        for (long i = 0; i < (long)p; ++i) /* hpc/affinity.c:239-250 */
            for (size_t c = 0; c < R; ++c) K_B[(size_t)i * R + c] = pair_entry(&pc, idx[i], rem[c]);
        free(rem);
    }
    pairs_free(&pc);
    return 0;
}

/* D_A = rowsum(K_A) + rowsum(K_B), hpc/laplacian.c:18-20, K_B streamed. */
int orc_degree(const orc_params *prm, const uint8_t *img, int width, int height,
               int row0, int row1, unsigned p, const unsigned *idx, double *D)
{
    const size_t N = (size_t)width * (size_t)height;
    if (row0 < 0 || row1 > height || row0 > row1) return -1;
    uint8_t *mask = build_sample_mask(N, p, idx);
    if (!mask) return -1;
    orc_pairs pc;
    if (pairs_init(&pc, prm, img, width, height) != 0) { free(mask); return -1; }
#pragma omp parallel for schedule(dynamic, 8)
    for (long i = 0; i < (long)p; ++i) {
        double sumA = 0.0, sumB = 0.0;
        for (int r = row0; r < row1; ++r) {
            const uint8_t *mrow = mask + (size_t)r * width;
            for (int c = 0; c < width; ++c) {
                const double k = pair_entry(&pc, idx[i], (size_t)r * width + c);
                if (mrow[c]) sumA += k; else sumB += k;
            }
        }
        D[i] = sumA + sumB;
    }
    pairs_free(&pc);
    free(mask);
    return 0;
}

/* hpc/laplacian.c:14-42 (L_A half; L_B = -alpha K_B is applied on the fly in
 * orc_nystroem). alpha = 1 / VecMean(D), hpc/utils.c:378-388. */
int orc_laplacian(const double *K_A, const double *D, unsigned p, double *L_A, double *alpha)
{
    double sum = 0.0;
    for (unsigned i = 0; i < p; ++i) sum += D[i];
    const double a = 1.0 / (sum / (double)p);
    *alpha = a;
#pragma omp parallel for schedule(static)
    for (long i = 0; i < (long)p; ++i)
        for (unsigned j = 0; j < p; ++j) {
            const double d = (i == (long)j) ? D[i] : 0.0;
            /* MatAYPX(L_A, -1, D_A): L_A = -K_A + D_A, then MatScale(alpha) */
            L_A[(size_t)i * p + j] = (-1.0 * K_A[(size_t)i * p + j] + d) * a;
        }
    return 0;
}

/* ---------------------------------------------------------------- PRNG */

static uint64_t splitmix64(uint64_t *s)
{
    uint64_t z = (*s += 0x9E3779B97F4A7C15ull);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}
static inline uint64_t rotl64(uint64_t x, int k) { return (x << k) | (x >> (64 - k)); }
static uint64_t xoshiro_next(uint64_t s[4])
{
    const uint64_t result = rotl64(s[1] * 5, 7) * 9;
    const uint64_t t = s[1] << 17;
    s[2] ^= s[0]; s[3] ^= s[1]; s[1] ^= s[2]; s[0] ^= s[3];
    s[2] ^= t; s[3] = rotl64(s[3], 45);
    return result;
}

/* Stands in for BuildRandomVectors, hpc/inverse_power_it.c:12-47 (U[0,1),
 * vector after vector). */
void orc_random_vectors(double *X, unsigned p, unsigned m, uint64_t seed)
{
    uint64_t sm = seed, s[4];
    for (int i = 0; i < 4; ++i) s[i] = splitmix64(&sm);
    for (size_t k = 0; k < (size_t)p * m; ++k)
        X[k] = (double)(xoshiro_next(s) >> 11) * (1.0 / 9007199254740992.0);
}

/* ---------------------------------------------------------------- Gram-Schmidt */

static double vdot(const double *a, const double *b, unsigned n)
{
    double s = 0.0;
    for (unsigned i = 0; i < n; ++i) s += a[i] * b[i];
    return s;
}

/* hpc/gram_schmidt.c:29-64: classical GS; every projection uses the
 * unmodified X[k] (Projection :11-21 divides by <u,u> of the already
 * normalised u). */
void orc_orthonormalise(double *X, unsigned n, unsigned m, double *norms)
{
    double *sum = (double *)malloc(sizeof(double) * n);
    double *coef = (double *)malloc(sizeof(double) * (m ? m : 1));
    for (unsigned k = 0; k < m; ++k) {
        double *xk = X + (size_t)k * n;
#pragma omp parallel for schedule(static)
        for (long j = 0; j < (long)k; ++j) {
            const double *xj = X + (size_t)j * n;
            coef[j] = vdot(xk, xj, n) / vdot(xj, xj, n);
        }
        memset(sum, 0, sizeof(double) * n);
        for (unsigned j = 0; j < k; ++j) {
            const double *xj = X + (size_t)j * n;
            const double f = coef[j];
            for (unsigned i = 0; i < n; ++i) sum[i] += f * xj[i];
        }
        for (unsigned i = 0; i < n; ++i) xk[i] = -1.0 * sum[i] + xk[i]; /* VecAXPBY :53 */
        const double nrm = sqrt(vdot(xk, xk, n));                      /* VecNormalize :59 */
        if (norms) norms[k] = nrm;
        if (nrm != 0.0) {
            const double inv = 1.0 / nrm;
            for (unsigned i = 0; i < n; ++i) xk[i] *= inv;
        }
    }
    free(coef);
    free(sum);
}

/* hpc/gram_schmidt.c:66-77 */
void orc_normalise(double *X, unsigned n, unsigned m, double *norms)
{
    for (unsigned k = 0; k < m; ++k) {
        double *xk = X + (size_t)k * n;
        const double nrm = sqrt(vdot(xk, xk, n));
        if (norms) norms[k] = nrm;
        if (nrm != 0.0) {
            const double inv = 1.0 / nrm;
            for (unsigned i = 0; i < n; ++i) xk[i] *= inv;
        }
    }
}

/* ---------------------------------------------------------------- dense helpers */

/* Y (p x m col-major) = A (p x p row-major) * X (p x m col-major), only columns
 * with active[j] != 0 (active == NULL: all). One pass over A. */
static void block_matvec(const double *A, const double *X, double *Y, unsigned p, unsigned m,
                         const uint8_t *active)
{
    /* pack the active columns row-major so the inner loop is contiguous */
    unsigned na = 0;
    unsigned *cols = (unsigned *)malloc(sizeof(unsigned) * (m ? m : 1));
    for (unsigned j = 0; j < m; ++j)
        if (!active || active[j]) cols[na++] = j;
    if (na == 0) { free(cols); return; }
    double *Xr = (double *)malloc(sizeof(double) * (size_t)p * na);
    for (unsigned a = 0; a < na; ++a) {
        const double *xc = X + (size_t)cols[a] * p;
        for (unsigned i = 0; i < p; ++i) Xr[(size_t)i * na + a] = xc[i];
    }
#pragma omp parallel
    {
        double *acc = (double *)malloc(sizeof(double) * na);
#pragma omp for schedule(static)
        for (long i = 0; i < (long)p; ++i) {
            const double *Ai = A + (size_t)i * p;
            for (unsigned a = 0; a < na; ++a) acc[a] = 0.0;
            for (unsigned k = 0; k < p; ++k) {
                const double aik = Ai[k];
                const double *xr = Xr + (size_t)k * na;
                for (unsigned a = 0; a < na; ++a) acc[a] += aik * xr[a];
            }
            for (unsigned a = 0; a < na; ++a) Y[(size_t)cols[a] * p + i] = acc[a];
        }
        free(acc);
    }
    free(Xr);
    free(cols);
}

/* hpc/inverse_power_it.c:49-80: R = (I - X X^T) A X, Frobenius norm. Formed
 * as A X - X (X^T (A X)): the same matrix without the p x p products (quirk Q5). */
double orc_residual_norm(const double *A, const double *X, unsigned p, unsigned m)
{
    double *AX = (double *)malloc(sizeof(double) * (size_t)p * m);
    double *G = (double *)malloc(sizeof(double) * (size_t)m * m);
    block_matvec(A, X, AX, p, m, NULL);
#pragma omp parallel for schedule(static)
    for (long a = 0; a < (long)m; ++a)
        for (unsigned b = 0; b < m; ++b)
            G[(size_t)a * m + b] = vdot(X + (size_t)a * p, AX + (size_t)b * p, p); /* (X^T AX)[a][b] */
    double ss = 0.0;
#pragma omp parallel for schedule(static) reduction(+ : ss)
    for (long b = 0; b < (long)m; ++b) {
        for (unsigned i = 0; i < p; ++i) {
            double r = AX[(size_t)b * p + i];
            for (unsigned a = 0; a < m; ++a) r -= X[(size_t)a * p + i] * G[(size_t)a * m + b];
            ss += r * r;
        }
    }
    free(G);
    free(AX);
    return sqrt(ss);
}

/* Jacobi-PCG on all columns at once; converged columns are frozen. Stands in
 * for the m KSPSolve calls of hpc/inverse_power_it.c:165-168 (quirk Q8). */
int orc_block_pcg(const double *A, const double *B, double *Xout, unsigned p, unsigned m,
                  double rtol, int max_it)
{
    const size_t pm = (size_t)p * m;
    double *R = (double *)malloc(sizeof(double) * pm);
    double *Z = (double *)malloc(sizeof(double) * pm);
    double *P = (double *)malloc(sizeof(double) * pm);
    double *AP = (double *)malloc(sizeof(double) * pm);
    double *X = (double *)calloc(pm, sizeof(double));
    double *dinv = (double *)malloc(sizeof(double) * p);
    double *rz = (double *)malloc(sizeof(double) * m);
    double *bn = (double *)malloc(sizeof(double) * m);
    uint8_t *active = (uint8_t *)malloc(m ? m : 1);
    for (unsigned i = 0; i < p; ++i) dinv[i] = 1.0 / A[(size_t)i * p + i];
    memcpy(R, B, sizeof(double) * pm);
    unsigned nactive = 0;
    for (unsigned j = 0; j < m; ++j) {
        double *r = R + (size_t)j * p, *z = Z + (size_t)j * p, *pp = P + (size_t)j * p;
        for (unsigned i = 0; i < p; ++i) { z[i] = dinv[i] * r[i]; pp[i] = z[i]; }
        rz[j] = vdot(r, z, p);
        bn[j] = sqrt(vdot(r, r, p));
        active[j] = (bn[j] > 0.0);
        nactive += active[j];
    }
    int it = 0;
    while (nactive > 0 && it < max_it) {
        ++it;
        block_matvec(A, P, AP, p, m, active);
#pragma omp parallel for schedule(static)
        for (long j = 0; j < (long)m; ++j) {
            if (!active[j]) continue;
            double *r = R + (size_t)j * p, *z = Z + (size_t)j * p, *pp = P + (size_t)j * p;
            double *ap = AP + (size_t)j * p, *x = X + (size_t)j * p;
            const double alpha = rz[j] / vdot(pp, ap, p);
            for (unsigned i = 0; i < p; ++i) { x[i] += alpha * pp[i]; r[i] -= alpha * ap[i]; }
            const double rn = sqrt(vdot(r, r, p));
            if (rn <= rtol * bn[j]) { active[j] = 0; continue; }
            for (unsigned i = 0; i < p; ++i) z[i] = dinv[i] * r[i];
            const double rz_new = vdot(r, z, p);
            const double beta = rz_new / rz[j];
            for (unsigned i = 0; i < p; ++i) pp[i] = z[i] + beta * pp[i];
            rz[j] = rz_new;
        }
        nactive = 0;
        for (unsigned j = 0; j < m; ++j) nactive += active[j];
    }
    memcpy(Xout, X, sizeof(double) * pm);
    free(active); free(bn); free(rz); free(dinv); free(X); free(AP); free(P); free(Z); free(R);
    return it;
}

/* hpc/inverse_power_it.c:86-252 */
int orc_inverse_power_iteration(const double *A, unsigned p, unsigned m, const double *X0,
                                int opti_gs, double epsilon, double inner_rtol, int max_outer,
                                double *eigvecs, double *eigvals, orc_eig_stats *stats)
{
    if (opti_gs < 1) opti_gs = 1; /* hpc/image_processing.c:128-140 */
    const size_t pm = (size_t)p * m;
    double *X = (double *)malloc(sizeof(double) * pm);
    double *Xb = (double *)malloc(sizeof(double) * pm);
    double *norms = (double *)malloc(sizeof(double) * (m ? m : 1));
    if (!X || !Xb || !norms) return -1;
    memcpy(X, X0, sizeof(double) * pm);
    orc_orthonormalise(X, p, m, norms);          /* :95 */
    memcpy(Xb, X, sizeof(double) * pm);          /* reference leaves X_before_orth unset if the
                                                    loop never runs (:97-101); we define it */
    double r_norm = orc_residual_norm(A, X, p, m); /* :159 */
    int it = 0, inner_total = 0;
    while (r_norm > epsilon && it < max_outer) { /* :161 */
        ++it;
        inner_total += orc_block_pcg(A, X, X, p, m, inner_rtol, 10 * (int)p + 100); /* :165-168 */
        memcpy(Xb, X, sizeof(double) * pm);      /* CopyVecs :171 */
        if (it % opti_gs == 0) orc_orthonormalise(X, p, m, norms); /* :174-177 */
        r_norm = orc_residual_norm(A, X, p, m);  /* :180 */
    }
    if (opti_gs != 1 && (it % opti_gs) != 0) orc_orthonormalise(X, p, m, norms); /* :183-186 */
    if (eigvals)
        for (unsigned j = 0; j < m; ++j) eigvals[j] = 1.0 / norms[j]; /* :204 */
    if (eigvecs) {
        orc_normalise(Xb, p, m, NULL);           /* :230 */
        memcpy(eigvecs, Xb, sizeof(double) * pm);
    }
    if (stats) { stats->outer_its = it; stats->inner_its_total = inner_total; stats->residual = r_norm; }
    free(norms); free(Xb); free(X);
    return 0;
}

/* ---------------------------------------------------------------- Nystroem */

/* pos[pixel] = row of that pixel in the sample-first ordering
 * ([samples ; remaining pixels in raster order], hpc/affinity.c:215-235). */
static unsigned *build_sample_first_pos(size_t N, unsigned p, const unsigned *idx)
{
    unsigned *pos = (unsigned *)malloc(sizeof(unsigned) * N);
    if (!pos) return NULL;
    size_t k1 = 0, k2 = p;
    for (size_t i = 0; i < N; ++i) {
        if (k1 < p && i == idx[k1]) pos[i] = (unsigned)k1++;
        else pos[i] = (unsigned)k2++;
    }
    return pos;
}

/* hpc/nystroem.c:5-69 */
int orc_nystroem(const orc_params *prm, const uint8_t *img, int width, int height,
                 unsigned p, const unsigned *idx, double alpha,
                 const double *phi_A, const double *eigvals, unsigned m, double *phi)
{
    const size_t N = (size_t)width * (size_t)height;
    unsigned *pos = build_sample_first_pos(N, p, idx);
    if (!pos) return -1;
    /* part_lower = phi_A * Pi^-1 (:41), stored row-major p x m */
    double *PL = (double *)malloc(sizeof(double) * (size_t)p * m);
    orc_pairs pc;
    if (!PL || pairs_init(&pc, prm, img, width, height) != 0) { free(PL); free(pos); return -1; }
    for (unsigned i = 0; i < p; ++i)
        for (unsigned j = 0; j < m; ++j)
            PL[(size_t)i * m + j] = phi_A[(size_t)j * p + i] * (1. / eigvals[j]); /* InverseDiagMat hpc/utils.c:559-586 */
    /* upper part (:25-34) */
    for (unsigned j = 0; j < m; ++j)
        memcpy(phi + (size_t)j * N, phi_A + (size_t)j * p, sizeof(double) * p);
    /* lower = L_B^T part_lower (:42), L_B = -alpha K_B (hpc/laplacian.c:37-38) */
#pragma omp parallel
    {
        double *acc = (double *)malloc(sizeof(double) * (m ? m : 1));
#pragma omp for schedule(dynamic, 64)
        for (long px = 0; px < (long)N; ++px) {
            const unsigned q = pos[px];
            if (q < p) continue;
            for (unsigned j = 0; j < m; ++j) acc[j] = 0.0;
            for (unsigned i = 0; i < p; ++i) {
                const double lb = -alpha * pair_entry(&pc, idx[i], (size_t)px);
                const double *pl = PL + (size_t)i * m;
                for (unsigned j = 0; j < m; ++j) acc[j] += lb * pl[j];
            }
            for (unsigned j = 0; j < m; ++j) phi[(size_t)j * N + q] = acc[j];
        }
        free(acc);
    }
    pairs_free(&pc);
    free(PL); free(pos);
    return 0;
}

/* hpc/utils.c:134-173 */
int orc_permutation(const double *in, double *out, unsigned N, unsigned m,
                    const unsigned *idx, unsigned p, int literal)
{
    if (literal) {
        for (unsigned i = 0; i < N; ++i) {
            unsigned new_pos;
            if (i < p) new_pos = idx[i];
            else {
                unsigned n = 0; /* hpc/utils.c:157-161 */
                while (n < p && idx[n] <= (i - p + n)) ++n;
                new_pos = i - p + n;
            }
            for (unsigned j = 0; j < m; ++j) out[(size_t)j * N + new_pos] = in[(size_t)j * N + i];
        }
        return 0;
    }
    unsigned *pos = build_sample_first_pos(N, p, idx);
    if (!pos) return -1;
#pragma omp parallel for schedule(static)
    for (long j = 0; j < (long)m; ++j)
        for (unsigned px = 0; px < N; ++px)
            out[(size_t)j * N + px] = in[(size_t)j * N + pos[px]];
    free(pos);
    return 0;
}

/* ---------------------------------------------------------------- filter */

static uint8_t to_png_byte(double z)
{
    /* AboveXSetY(z, 255, 255) hpc/display.c:76, then the (png_byte) cast of
     * hpc/utils.c:525. Negative values are undefined behaviour in the cast;
     * survey quirk Q4: clamp at 0, truncate toward zero. */
    if (z > 255.0) z = 255.0;
    if (!(z > 0.0)) z = 0.0;
    return (uint8_t)z;
}

/* hpc/display.c:58-83 */
int orc_result_from_laplacian(const uint8_t *img, int width, int height, const double *phi,
                              const double *f_eigvals, unsigned m, double gain,
                              double *zf, uint8_t *out)
{
    const size_t N = (size_t)width * (size_t)height;
    double *right = (double *)malloc(sizeof(double) * (m ? m : 1));
    /* right = phi^T z (:66) ; left = phi Pi (:64) */
#pragma omp parallel for schedule(static)
    for (long j = 0; j < (long)m; ++j) {
        const double *col = phi + (size_t)j * N;
        double s = 0.0;
        for (size_t i = 0; i < N; ++i) s += col[i] * (double)img[i];
        right[j] = s;
    }
#pragma omp parallel for schedule(static)
    for (long i = 0; i < (long)N; ++i) {
        double ly = 0.0;
        for (unsigned j = 0; j < m; ++j) ly += (phi[(size_t)j * N + i] * f_eigvals[j]) * right[j];
        const double z = (double)img[i] + gain * ly; /* MatAXPY(z, 3.0, Lapl_y) :73 */
        if (zf) zf[i] = z;
        out[i] = to_png_byte(z);
    }
    free(right);
    return 0;
}

/* ---------------------------------------------------------------- whole path */

/* hpc/image_processing.c:183-277 with the commented tail :240-275 as the
 * specification (survey quirk Q1). MatPow is a no-op (hpc/utils.c:721, Q3) so
 * f(Pi) = Pi. */
int orc_image_processing(const orc_params *prm, const uint8_t *img, int width, int height,
                         orc_run *run, double *eigvals_out, double *zf, uint8_t *out)
{
    const size_t N = (size_t)width * (size_t)height;
    unsigned p = run->p_requested, *idx = NULL;
    if (orc_sampling(width, height, &p, &idx) != 0) return -1;
    run->p = p;
    unsigned m = run->m;
    if (m >= p) m = p - 1; /* hpc/image_processing.c:96-108 */
    run->m = m;

    double t0 = now_s();
    double *K_A = (double *)malloc(sizeof(double) * (size_t)p * p);
    double *D = (double *)malloc(sizeof(double) * p);
    if (!K_A || !D) return -1;
    orc_affinity(prm, img, width, height, p, idx, K_A, NULL);
    orc_degree(prm, img, width, height, 0, height, p, idx, D);
    run->t_affinity = now_s() - t0;

    t0 = now_s();
    orc_laplacian(K_A, D, p, K_A, &run->alpha);
    run->t_laplacian = now_s() - t0;

    t0 = now_s();
    double *X0 = (double *)malloc(sizeof(double) * (size_t)p * m);
    double *phiA = (double *)malloc(sizeof(double) * (size_t)p * m);
    double *lam = (double *)malloc(sizeof(double) * m);
    orc_random_vectors(X0, p, m, run->seed);
    orc_inverse_power_iteration(K_A, p, m, X0, run->opti_gs, run->epsilon, run->inner_rtol,
                                run->max_outer, phiA, lam, &run->eig);
    run->t_eigen = now_s() - t0;
    if (eigvals_out) memcpy(eigvals_out, lam, sizeof(double) * m);

    t0 = now_s();
    double *phi_sf = (double *)malloc(sizeof(double) * N * m);
    double *phi = (double *)malloc(sizeof(double) * N * m);
    if (!phi_sf || !phi) return -1;
    orc_nystroem(prm, img, width, height, p, idx, run->alpha, phiA, lam, m, phi_sf);
    orc_permutation(phi_sf, phi, (unsigned)N, m, idx, p, 0);
    run->t_nystroem = now_s() - t0;

    t0 = now_s();
    orc_result_from_laplacian(img, width, height, phi, lam, m, run->gain, zf, out);
    run->t_filter = now_s() - t0;

    free(phi); free(phi_sf); free(lam); free(phiA); free(X0); free(D); free(K_A); free(idx);
    return 0;
}

/* hpc/image_processing.c:155-181: z = clamp_0^255(y - L y), L = alpha (D - K)
 * over ALL pixels (hpc/affinity.c:264-336, hpc/laplacian.c:44-65,
 * hpc/display.c:128-149). K is never stored: (L y)_i = alpha (D_i y_i - sum_j K_ij y_j). */
int orc_entire_computation(const orc_params *prm, const uint8_t *img, int width, int height,
                           double *zf, uint8_t *out)
{
    const size_t N = (size_t)width * (size_t)height;
    if (prm->kernel == ORC_NLM) return -1; /* (the full-matrix mode exists for the positional kernels only) */
    double *D = (double *)malloc(sizeof(double) * N);
    double *Ky = (double *)malloc(sizeof(double) * N);
    if (!D || !Ky) return -1;
#pragma omp parallel for schedule(dynamic, 16)
    for (long i = 0; i < (long)N; ++i) {
        const double r0 = (double)((unsigned)i / (unsigned)width), c0 = (double)((unsigned)i % (unsigned)width);
        const double v0 = (double)img[i];
        double d = 0.0, ky = 0.0;
        for (size_t j = 0; j < N; ++j) {
            const double k = orc_kernel_entry(prm, r0, c0, v0, (double)(j / (unsigned)width),
                                              (double)(j % (unsigned)width), (double)img[j]);
            d += k;
            ky += k * (double)img[j];
        }
        D[i] = d; Ky[i] = ky;
    }
    double sum = 0.0;
    for (size_t i = 0; i < N; ++i) sum += D[i];
    const double alpha = 1.0 / (sum / (double)N);
    for (size_t i = 0; i < N; ++i) {
        const double Ly = alpha * (D[i] * (double)img[i] - Ky[i]);
        double z = (double)img[i] - Ly;            /* MatAXPY(z, -1, Lapl_y) hpc/display.c:136 */
        if (zf) zf[i] = z;
        if (z > 255.0) z = 255.0;                  /* AboveXSetY :139 */
        if (z < 0.0) z = 0.0;                      /* SetNegativesToZero :141 */
        out[i] = (uint8_t)z;
    }
    free(Ky); free(D);
    return 0;
}

/* ---------------------------------------------------------------- bounded-sample helpers
 * Used by bench.py's cpu_baseline leg: the same stage arithmetic as above on a
 * slice of the workload (a band of pixel rows / a band of L_A rows) so a 4096^2
 * baseline can be timed in seconds and scaled. */

/* Rows [i0,i1) of L_A = alpha (diag(D) - K_A) straight from the samples
 * (hpc/affinity.c:181-193 + hpc/laplacian.c:31-35). out: (i1-i0) x p row-major. */
int orc_laplacian_rows(const orc_params *prm, const uint8_t *img, int width, int height, unsigned p, const unsigned *idx,
                       const double *D, double alpha, unsigned i0, unsigned i1, double *out)
{
    if (i0 > i1 || i1 > p) return -1;
    orc_pairs pc;
    if (pairs_init(&pc, prm, img, width, height) != 0) return -1;
#pragma omp parallel for schedule(static)
    for (long i = i0; i < (long)i1; ++i)
        for (unsigned j = 0; j < p; ++j) {
            const double k = pair_entry(&pc, idx[i], idx[j]);
            out[(size_t)(i - i0) * p + j] = (-1.0 * k + (((unsigned)i == j) ? D[i] : 0.0)) * alpha;
        }
    pairs_free(&pc);
    return 0;
}

/* Y (m vectors of length nrows) = Arows (nrows x p) * X (m vectors of length p). */
int orc_matvec_rows(const double *Arows, unsigned nrows, unsigned p, const double *X, unsigned m, double *Y)
{
    double *Xr = (double *)malloc(sizeof(double) * (size_t)p * m);
    if (!Xr) return -1;
    for (unsigned j = 0; j < m; ++j)
        for (unsigned i = 0; i < p; ++i) Xr[(size_t)i * m + j] = X[(size_t)j * p + i];
#pragma omp parallel
    {
        double *acc = (double *)malloc(sizeof(double) * (m ? m : 1));
#pragma omp for schedule(static)
        for (long i = 0; i < (long)nrows; ++i) {
            const double *Ai = Arows + (size_t)i * p;
            for (unsigned j = 0; j < m; ++j) acc[j] = 0.0;
            for (unsigned k = 0; k < p; ++k) {
                const double a = Ai[k];
                const double *xr = Xr + (size_t)k * m;
                for (unsigned j = 0; j < m; ++j) acc[j] += a * xr[j];
            }
            for (unsigned j = 0; j < m; ++j) Y[(size_t)j * nrows + i] = acc[j];
        }
        free(acc);
    }
    free(Xr);
    return 0;
}

/* Nystroem rows (hpc/nystroem.c:41-42) for the pixels of image rows [row0,row1), raster order,
 * sample pixels included with the extension formula (they are overwritten by phi_A in the full
 * path). out: m vectors of length (row1-row0)*width. */
int orc_nystroem_rows(const orc_params *prm, const uint8_t *img, int width, int height, int row0, int row1,
                      unsigned p, const unsigned *idx, double alpha, const double *phi_A, const double *eigvals,
                      unsigned m, double *out)
{
    if (row0 < 0 || row1 > height || row0 > row1) return -1;
    const size_t npix = (size_t)(row1 - row0) * width;
    double *PL = (double *)malloc(sizeof(double) * (size_t)p * m);
    orc_pairs pc;
    if (!PL || pairs_init(&pc, prm, img, width, height) != 0) { free(PL); return -1; }
    for (unsigned i = 0; i < p; ++i)
        for (unsigned j = 0; j < m; ++j) PL[(size_t)i * m + j] = phi_A[(size_t)j * p + i] * (1. / eigvals[j]);
#pragma omp parallel
    {
        double *acc = (double *)malloc(sizeof(double) * (m ? m : 1));
#pragma omp for schedule(dynamic, 16)
        for (long q = 0; q < (long)npix; ++q) {
            const size_t px = (size_t)row0 * width + (size_t)q;
            for (unsigned j = 0; j < m; ++j) acc[j] = 0.0;
            for (unsigned i = 0; i < p; ++i) {
                const double lb = -alpha * pair_entry(&pc, idx[i], px);
                const double *pl = PL + (size_t)i * m;
                for (unsigned j = 0; j < m; ++j) acc[j] += lb * pl[j];
            }
            for (unsigned j = 0; j < m; ++j) out[(size_t)j * npix + q] = acc[j];
        }
        free(acc);
    }
    pairs_free(&pc);
    free(PL);
    return 0;
}
