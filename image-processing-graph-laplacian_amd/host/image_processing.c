/*
 * image_processing.c -- host shell of the MI355X-native graph-Laplacian filter.
 * Same entry point, flags, stage order, log lines and output files as the
 * reference's hpc/image_processing.c, with PETSc/SLEPc/MPI replaced by the HIP
 * C-ABI (include/glf.h) through the stage mirror in stages.h.
 *
 *   image_processing -f FILE [-num_eigvals N] [-opti_gs N] [-inv_it_epsilon E]
 *                    [-num_samples P | -sample_frac F] [-sampling uniform|random] [-sampling_seed S] [-fused] [-device D] [-no_approx] [-use_slepc]
 *                    [-dump_eigvecs] [-ngpu N [-ngpu_backend rccl|loopback]] [-filter_pow K]
 *                    [-kernel bilateral|photometric|spatial|nlm] [-h_loc X] [-h_val X] [-gain X] [-dump_residual]
 *                    [-filter reference|poc|smooth|sharpen [-sharpen_beta B]] [-color]
 * -filter poc applies the Python PoC's active filter z = y - Phi diag(mu + 5) Phi^T y (python/image_processing.py:304-305) instead
 * of hpc/display.c:58-83; -filter smooth / sharpen the PoC's `smoothing` z = W y and `sharpening` z = (1 + B) W^2 y - B W^3 y
 * (python/image_processing.py:197-241, B = 1.5) with W = Phi diag(1 - mu) Phi^T from the eigenpairs this program computes.
 * -color keeps the colours of an RGB(A) input the way the PoC does (python/image_processing.py:410-432):
 * RGB -> YUV (python/utils.py:33-44), the luma plane is filtered, the chroma planes pass through, YUV -> RGB; the luma is
 * rounded to 8 bits first because every kernel here works on u8 pixel values (the PoC filters the unrounded floats).
 * -dump_residual writes results/residuals.png = |input - output| stretched to the full grey range, the PoC's residual image
 * (python/image_processing.py:378-380: plt.imsave of np.abs(y - z) with cmap 'gray' autoscales min..max).
 * -no_approx runs the full-matrix mode (hpc/image_processing.c:155-181); -use_slepc is accepted and refused.
 * -ngpu N is the reference's `mpirun -n N` (hpc/image_processing.c:30-38, 45-76): ONE process, N GPUs, one context and
 * one host thread per device, pixel rows sharded, RCCL collectives over xGMI issued by the library (glf_multi_*).
 * -ngpu_backend loopback runs the N ranks on one device with host-staged collectives (how the sharding is tested on a
 * one-GPU box). -filter_pow K: f(Pi) = Pi^K (the reference's MatPow(eigvals, 6) drops its result, hpc/utils.c:721, so the
 * default is K = 1; survey quirk Q3).
 *
 * The approximate path runs the tail the reference left commented out
 * (hpc/image_processing.c:240-275) as its specification (survey quirk Q1).
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

#include "stages.h"

static double wtime(void) /* MPI_Wtime */
{
    struct timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec;
}

/* ---- option database (PetscOptionsGet*, hpc/image_processing.c:82-154) ---------- */
static int g_argc;
static char **g_argv;

static const char *opt_value(const char *name)
{
    for (int i = 1; i + 1 < g_argc; ++i)
        if (strcmp(g_argv[i], name) == 0) return g_argv[i + 1];
    return NULL;
}
static int opt_has(const char *name)
{
    for (int i = 1; i < g_argc; ++i)
        if (strcmp(g_argv[i], name) == 0) return 1;
    return 0;
}

static void GetFilePath(char *filename, size_t len) /* :82-94 */
{
    const char *v = opt_value("-f");
    if (!v) {
        fprintf(stderr, "No filename found (option -f)\n");
        exit(1);
    }
    snprintf(filename, len, "%s", v);
}

static unsigned GetNumberEigenvalues(unsigned sample_size) /* :96-108 */
{
    const char *v = opt_value("-num_eigvals");
    long n = v ? strtol(v, NULL, 10) : -1;
    if (!v || n < 0 || n >= (long)sample_size) {
        n = (long)sample_size - 1;
        fprintf(stderr, "Invalid or invalid number of eigenvalues found (option -num_eigvals), so using %ld\n", n);
    }
    return (unsigned)n;
}

static int GetOptiGramSchmidt(void) /* :128-140 */
{
    const char *v = opt_value("-opti_gs");
    const long n = v ? strtol(v, NULL, 10) : 1;
    return n < 1 ? 1 : (int)n;
}

static double GetInverseIterationEpsilon(void) /* :142-154 */
{
    const char *v = opt_value("-inv_it_epsilon");
    return v ? strtod(v, NULL) : 0.1;
}

static unsigned GetRequestedSamples(unsigned width, unsigned height)
{
    const char *n = opt_value("-num_samples"), *f = opt_value("-sample_frac");
    if (n) return (unsigned)strtoul(n, NULL, 10);
    const double frac = f ? strtod(f, NULL) : 0.01; /* p = width*height*0.01, :187 */
    return (unsigned)(width * height * frac);
}

static void free_rows(png_bytep *rows, int height)
{
    if (!rows) return;
    for (int i = 0; i < height; ++i) free(rows[i]);
    free(rows);
}

/* ApproximationComputation, hpc/image_processing.c:183-277, stage by stage */
static png_bytep *ApproximationComputation(png_bytep *img_bytes, unsigned width, unsigned height)
{
    unsigned p = GetRequestedSamples(width, height);
    unsigned *sample_indices = NULL;
    Sampling((int)width, (int)height, &p, &sample_indices);
    if (!sample_indices || p < 2) {
        fprintf(stderr, "Sampling failed\n");
        return NULL;
    }
    printf("Sample size: %d\n", p);
    const unsigned m = GetNumberEigenvalues(p);

    double t = wtime();
    printf("Computing affinity matrices... ");
    Mat K_A = NULL, K_B = NULL;
    if (ComputeAffinityMatrices(&K_A, &K_B, img_bytes, (int)width, (int)height, p, sample_indices) != GLF_OK) goto fail;
    printf("%fs\n", wtime() - t);

    t = wtime();
    printf("Computing Laplacian matrices... ");
    Mat L_A = NULL, L_B = NULL;
    if (ComputeLaplacianMatrix(&L_A, &L_B, K_A, K_B) != GLF_OK) goto fail;
    printf("%fs\n", wtime() - t);
    MatDestroy(&K_A); /* K_B's tables are shared with L_B: destroyed at the end */

    t = wtime();
    Mat eigvals = NULL, eigvecs_A = NULL;
    printf("Computing %d smallest eigenvalues... ", m);
    if (opt_has("-use_slepc")) {
        fprintf(stderr, "-use_slepc: SLEPc is not part of this build; using the inverse subspace iteration\n");
    }
    const double epsilon = GetInverseIterationEpsilon();
    printf("(epsilon: %g) ", epsilon);
    if (InversePowerIteration(L_A, m, &eigvecs_A, &eigvals, GetOptiGramSchmidt(), epsilon) != GLF_OK) goto fail;
    printf("%fs\n", wtime() - t);
    printf("Inverse subspace iteration took %d outer iterations\n", LastEigStats()->outer_its);
    WriteDiagMat(eigvals, "results/eigenvalues_laplacian.txt");
    MatDestroy(&L_A);

    Mat eigvals_inv = InverseDiagMat(eigvals); /* :240 */
    t = wtime();
    printf("Computing Nystr\xc3\xb6m approximation... ");
    Mat eigvecs = Nystroem(L_B, eigvecs_A, eigvals_inv, width * height, p, m); /* :245 */
    if (!eigvecs) goto fail;
    printf("%fs\n", wtime() - t);
    MatDestroy(&eigvecs_A);
    MatDestroy(&eigvals_inv);
    MatDestroy(&L_B);

    Mat eigvecs_perm = Permutation(eigvecs, sample_indices, p); /* :251 */
    MatDestroy(&eigvecs);
    eigvecs = eigvecs_perm;
    if (!eigvecs) goto fail;

    if (opt_has("-dump_eigvecs")) { /* the diagnostics of hpc/image_processing.c:252-260 */
        char name[128];
        for (unsigned k = 0; k < 3 && k < m; ++k) {
            snprintf(name, sizeof(name), "results/eigenvector_%u_laplacian.txt", k);
            WriteMatCol(eigvecs, k, name);
            snprintf(name, sizeof(name), "results/eigenvector_%u_laplacian.png", k);
            WritePngMatCol(eigvecs, k, width, height, name);
        }
    }

    Mat f_eigvals = MatPow(eigvals, 6); /* :263, a no-op in the reference */
    MatDestroy(&eigvals);

    t = wtime();
    printf("Computing output image... ");
    png_bytep *output_img = ComputeResultFromLaplacian(img_bytes, eigvecs, f_eigvals, width, height); /* :269 */
    printf("%fs\n", wtime() - t);
    MatDestroy(&eigvecs);
    MatDestroy(&f_eigvals);
    MatDestroy(&K_B);
    free(sample_indices);
    return output_img;
fail:
    fprintf(stderr, "\nstage failed: %s\n", glf_ctx_last_error(glf_world()));
    free(sample_indices);
    return NULL;
}

/* EntireComputation, hpc/image_processing.c:155-181 (-no_approx): full N x N affinity and Laplacian,
 * z = clamp(y - L y). The matrices are never stored (glf_EntireComputation). */
static png_bytep *EntireComputation(png_bytep *img_bytes, unsigned width, unsigned height)
{
    glf_ctx *ctx = glf_world();
    const size_t n = (size_t)width * height;
    void *d_img = NULL, *d_out = NULL;
    uint8_t *flat = (uint8_t *)malloc(n);
    png_bytep *rows = NULL;
    if (!flat || glf_malloc(ctx, &d_img, n) != GLF_OK || glf_malloc(ctx, &d_out, n) != GLF_OK) goto out;
    for (unsigned r = 0; r < height; ++r) memcpy(flat + (size_t)r * width, img_bytes[r], width);
    if (glf_memcpy_h2d(ctx, d_img, flat, n) != GLF_OK) goto out;
    const double t = wtime();
    printf("Computing entire affinity matrix, Laplacian matrix and output image (matrices not stored)... ");
    const int rc = glf_EntireComputation(ctx, (const uint8_t *)d_img, (int)width, (int)height, stage_kernel, stage_h_loc, stage_h_val,
                                         (uint8_t *)d_out, NULL, NULL);
    if (rc != GLF_OK) {
        fprintf(stderr, "\nglf_EntireComputation: %s (%s)\n", glf_strerror(rc), glf_ctx_last_error(ctx));
        goto out;
    }
    printf("%fs\n", wtime() - t);
    if (glf_memcpy_d2h(ctx, flat, d_out, n) != GLF_OK) goto out;
    rows = (png_bytep *)malloc(sizeof(png_bytep) * height);
    for (unsigned r = 0; rows && r < height; ++r) {
        rows[r] = (png_bytep)malloc(width);
        memcpy(rows[r], flat + (size_t)r * width, width);
    }
out:
    free(flat);
    if (d_img) glf_free(ctx, d_img);
    if (d_out) glf_free(ctx, d_out);
    return rows;
}

static void fill_options(glf_options *opt, unsigned width, unsigned height)
{
    glf_options_default(opt);
    opt->num_samples = GetRequestedSamples(width, height);
    const char *v = opt_value("-num_eigvals");
    opt->num_eigvals = v ? (uint32_t)strtoul(v, NULL, 10) : 0;
    opt->opti_gs = GetOptiGramSchmidt();
    opt->epsilon = GetInverseIterationEpsilon();
    opt->kernel = stage_kernel;
    opt->sampling = stage_sampling;
    opt->sampling_seed = stage_sampling_seed;
    opt->h_loc = stage_h_loc;
    opt->h_val = stage_h_val;
    opt->gain = stage_gain;
    if ((v = opt_value("-filter_pow")) && atoi(v) > 0) opt->filter_pow = atoi(v);
    if ((v = opt_value("-filter"))) {
        if (strcmp(v, "poc") == 0) opt->filter_mode = GLF_FILTER_POC;
        else if (strcmp(v, "smooth") == 0) opt->filter_mode = GLF_FILTER_SMOOTH;
        else if (strcmp(v, "sharpen") == 0) opt->filter_mode = GLF_FILTER_SHARPEN;
    }
    if ((v = opt_value("-sharpen_beta"))) opt->filter_beta = (float)atof(v);
}

static void print_stage_times(const glf_stats *st, double epsilon)
{
    printf("Sample size: %d\n", st->p);
    printf("Computing affinity matrices... %fs\n", st->ms_affinity * 1e-3);
    printf("Computing Laplacian matrices... %fs\n", st->ms_laplacian * 1e-3);
    printf("Computing %d smallest eigenvalues... (epsilon: %g) %fs\n", st->m, epsilon, st->ms_eigen * 1e-3);
    printf("Inverse subspace iteration took %d outer iterations\n", st->eig.outer_its);
    printf("Computing Nystr\xc3\xb6m approximation... %fs\n", st->ms_nystroem * 1e-3);
    printf("Computing output image... %fs\n", st->ms_filter * 1e-3);
}

/* mpirun -n N (hpc/image_processing.c:30-76): N GPUs driven by this one process (glf_multi_*): the image is replicated on
 * every device, pixel rows are sharded, the result rows come back into one host image. */
static png_bytep *MultiComputation(png_bytep *img_bytes, unsigned width, unsigned height, int ngpu, int backend)
{
    glf_multi *world = NULL;
    int *devices = NULL;
    if (backend == GLF_MULTI_LOOPBACK) { /* every rank on the device of -device (default 0) */
        const char *dev = opt_value("-device");
        devices = (int *)malloc(sizeof(int) * (size_t)ngpu);
        for (int r = 0; devices && r < ngpu; ++r) devices[r] = dev ? atoi(dev) : 0;
    }
    int rc = glf_multi_create(&world, ngpu, devices, backend);
    free(devices);
    if (rc != GLF_OK) {
        fprintf(stderr, "glf_multi_create(%d GPUs, %s): %s\n", ngpu, backend == GLF_MULTI_RCCL ? "rccl" : "loopback", glf_strerror(rc));
        return NULL;
    }
    glf_options opt;
    fill_options(&opt, width, height);
    const size_t n = (size_t)width * height;
    uint8_t *flat = (uint8_t *)malloc(n), *flat_out = (uint8_t *)calloc(n, 1);
    glf_stats *st = (glf_stats *)calloc((size_t)ngpu, sizeof(glf_stats));
    png_bytep *rows = NULL;
    if (!flat || !flat_out || !st) goto out;
    for (unsigned r = 0; r < height; ++r) memcpy(flat + (size_t)r * width, img_bytes[r], width);
    rc = glf_multi_image_processing(world, &opt, flat, (int)width, (int)height, flat_out, NULL, NULL, st);
    if (rc != GLF_OK) {
        fprintf(stderr, "glf_multi_image_processing: %s (%s)\n", glf_strerror(rc), glf_multi_last_error(world));
        goto out;
    }
    print_stage_times(&st[0], opt.epsilon);
    for (int r = 0; r < ngpu; ++r) printf("rank %d: pixel rows [%d, %d), %.3f ms on the device\n", r, st[r].row0, st[r].row1, st[r].ms_total);
    rows = (png_bytep *)malloc(sizeof(png_bytep) * height);
    for (unsigned r = 0; rows && r < height; ++r) {
        rows[r] = (png_bytep)malloc(width);
        memcpy(rows[r], flat_out + (size_t)r * width, width);
    }
out:
    free(flat);
    free(flat_out);
    free(st);
    glf_multi_destroy(world);
    return rows;
}

/* -color (python/image_processing.py:410-432): filter the luma of an RGB image, keep its chroma. Returns rows of 3 * width
 * bytes for glf_write_png_rgb; *input_rgb receives the input as read (for results/input.png). */
static const double yuv_from_rgb[3][3] = {{0.299, 0.587, 0.114},                 /* python/utils.py:33-36 (the BT.601 YUV matrix) */
                                          {-0.14714119, -0.28886916, 0.43601035},
                                          {0.61497538, -0.51496512, -0.10001026}};
static png_bytep *ColorComputation(const char *filename, unsigned *width_out, unsigned *height_out, png_bytep **input_rgb)
{
    int w = 0, h = 0;
    png_bytep *rgb = NULL, *rows = NULL;
    if (glf_read_png_rgb(filename, &rgb, &w, &h) != 0) return NULL;
    *width_out = (unsigned)w;
    *height_out = (unsigned)h;
    *input_rgb = rgb;
    printf("Read image %s of size %dx%d => %d pixels (colour: the luma plane is filtered)\n", filename, w, h, w * h);
    const size_t n = (size_t)w * h;
    /* rgb_from_yuv = inv(yuv_from_rgb), python/utils.py:38 */
    double inv[3][3];
    {
        const double(*a)[3] = yuv_from_rgb;
        const double det = a[0][0] * (a[1][1] * a[2][2] - a[1][2] * a[2][1]) - a[0][1] * (a[1][0] * a[2][2] - a[1][2] * a[2][0]) +
                           a[0][2] * (a[1][0] * a[2][1] - a[1][1] * a[2][0]);
        for (int i = 0; i < 3; ++i)
            for (int j = 0; j < 3; ++j) {
                const int r0 = (j + 1) % 3, r1 = (j + 2) % 3, c0 = (i + 1) % 3, c1 = (i + 2) % 3;
                inv[i][j] = (a[r0][c0] * a[r1][c1] - a[r0][c1] * a[r1][c0]) / det;
            }
    }
    double *yuv = (double *)malloc(sizeof(double) * 3 * n);
    uint8_t *luma = (uint8_t *)malloc(n);
    float *zf = (float *)malloc(sizeof(float) * n);
    glf_ctx *ctx = glf_world();
    void *d_img = NULL, *d_out = NULL, *d_zf = NULL;
    if (!yuv || !luma || !zf) goto out;
    for (int r = 0; r < h; ++r)
        for (int c = 0; c < w; ++c) {
            const double R = rgb[r][3 * c], G = rgb[r][3 * c + 1], B = rgb[r][3 * c + 2];
            const size_t i = (size_t)r * w + c;
            for (int k = 0; k < 3; ++k) yuv[k * n + i] = R * yuv_from_rgb[k][0] + G * yuv_from_rgb[k][1] + B * yuv_from_rgb[k][2]; /* rgb2ycc */
            double y = yuv[i] + 0.5;
            luma[i] = (uint8_t)(y < 0.0 ? 0.0 : (y > 255.0 ? 255.0 : y));
        }
    if (glf_malloc(ctx, &d_img, n) != GLF_OK || glf_malloc(ctx, &d_out, n) != GLF_OK || glf_malloc(ctx, &d_zf, sizeof(float) * n) != GLF_OK) goto out;
    if (glf_memcpy_h2d(ctx, d_img, luma, n) != GLF_OK) goto out;
    glf_options opt;
    fill_options(&opt, (unsigned)w, (unsigned)h);
    glf_stats st;
    {
        const int rc = glf_image_processing(ctx, &opt, (const uint8_t *)d_img, w, h, (uint8_t *)d_out, (float *)d_zf, NULL, &st);
        if (rc != GLF_OK) {
            fprintf(stderr, "glf_image_processing: %s (%s)\n", glf_strerror(rc), glf_ctx_last_error(ctx));
            goto out;
        }
    }
    print_stage_times(&st, opt.epsilon);
    if (glf_memcpy_d2h(ctx, zf, d_zf, sizeof(float) * n) != GLF_OK) goto out;
    rows = (png_bytep *)malloc(sizeof(png_bytep) * (size_t)h);
    for (int r = 0; rows && r < h; ++r) {
        rows[r] = (png_bytep)malloc(3 * (size_t)w);
        for (int c = 0; rows[r] && c < w; ++c) {
            const size_t i = (size_t)r * w + c;
            const double zy = (double)zf[i], u = yuv[n + i], v = yuv[2 * n + i]; /* z[:, :, 0] = z_ycc; chroma unchanged */
            for (int k = 0; k < 3; ++k) {                                          /* ycc2rgb, then astype(uint8) made safe */
                double x = zy * inv[k][0] + u * inv[k][1] + v * inv[k][2];
                x = x < 0.0 ? 0.0 : (x > 255.0 ? 255.0 : x);
                rows[r][3 * c + k] = (png_byte)x;
            }
        }
    }
out:
    free(yuv);
    free(luma);
    free(zf);
    if (d_img) glf_free(ctx, d_img);
    if (d_out) glf_free(ctx, d_out);
    if (d_zf) glf_free(ctx, d_zf);
    return rows;
}

/* Same path through the single fused entry point (no stage materialisation:
 * Phi is written in raster order directly and K_A is never stored). */
static png_bytep *FusedComputation(png_bytep *img_bytes, unsigned width, unsigned height)
{
    glf_ctx *ctx = glf_world();
    glf_options opt;
    fill_options(&opt, width, height);
    const size_t n = (size_t)width * height;
    void *d_img = NULL, *d_out = NULL;
    uint8_t *flat = (uint8_t *)malloc(n);
    png_bytep *rows = NULL;
    if (!flat || glf_malloc(ctx, &d_img, n) != GLF_OK || glf_malloc(ctx, &d_out, n) != GLF_OK) goto out;
    for (unsigned r = 0; r < height; ++r) memcpy(flat + (size_t)r * width, img_bytes[r], width);
    if (glf_memcpy_h2d(ctx, d_img, flat, n) != GLF_OK) goto out;
    glf_stats st;
    const int rc = glf_image_processing(ctx, &opt, (const uint8_t *)d_img, (int)width, (int)height, (uint8_t *)d_out, NULL, NULL, &st);
    if (rc != GLF_OK) {
        fprintf(stderr, "glf_image_processing: %s (%s)\n", glf_strerror(rc), glf_ctx_last_error(ctx));
        goto out;
    }
    print_stage_times(&st, opt.epsilon);
    if (glf_memcpy_d2h(ctx, flat, d_out, n) != GLF_OK) goto out;
    rows = (png_bytep *)malloc(sizeof(png_bytep) * height);
    for (unsigned r = 0; rows && r < height; ++r) {
        rows[r] = (png_bytep)malloc(width);
        memcpy(rows[r], flat + (size_t)r * width, width);
    }
out:
    free(flat);
    if (d_img) glf_free(ctx, d_img);
    if (d_out) glf_free(ctx, d_out);
    return rows;
}

int main(int argc, char **argv)
{
    g_argc = argc;
    g_argv = argv;
    char filename[4096];
    const char *dev = opt_value("-device");
    if (InitProgram(dev ? atoi(dev) : 0) != GLF_OK) return 2; /* :284 */
    const double start_time = wtime();
    const char *ng = opt_value("-ngpu"), *nb = opt_value("-ngpu_backend");
    const int ngpu = ng ? atoi(ng) : 0; /* 0: the single-context paths below */
    if (ng && ngpu < 1) {
        fprintf(stderr, "-ngpu needs a positive device count\n");
        FinalizeProgram();
        return 1;
    }
    printf("Running with %d processes\n", ngpu > 0 ? ngpu : 1); /* :286: here a "process" is a GPU rank of this one process */
    { /* additions: the reference's constants as flags, same defaults */
        const char *v;
        if ((v = opt_value("-h_loc")) && atof(v) > 0.0) stage_h_loc = (float)atof(v);
        if ((v = opt_value("-h_val")) && atof(v) > 0.0) stage_h_val = (float)atof(v);
        if ((v = opt_value("-gain"))) stage_gain = (float)atof(v);
        if ((v = opt_value("-sampling"))) { /* the PoC's sampler registry, python/sampling/__init__.py:4-9 (the C reference has the grid only) */
            if (strcmp(v, "uniform") == 0 || strcmp(v, "spatially_uniform") == 0) stage_sampling = GLF_SAMPLING_UNIFORM;
            else if (strcmp(v, "random") == 0) stage_sampling = GLF_SAMPLING_RANDOM;
            else {
                fprintf(stderr, "-sampling %s: expected uniform or random\n", v);
                FinalizeProgram();
                return 1;
            }
        }
        if ((v = opt_value("-sampling_seed"))) stage_sampling_seed = strtoull(v, NULL, 10);
        if ((v = opt_value("-kernel"))) { /* bilateral (hpc/affinity.c:121) | photometric | spatial (:119-120) | nlm (python/affinity_methods/NLM.py) */
            if (strcmp(v, "bilateral") == 0) stage_kernel = GLF_KERNEL_BILATERAL;
            else if (strcmp(v, "photometric") == 0) stage_kernel = GLF_KERNEL_PHOTOMETRIC;
            else if (strcmp(v, "spatial") == 0) stage_kernel = GLF_KERNEL_SPATIAL;
            else if (strcmp(v, "nlm") == 0) {
                stage_kernel = GLF_KERNEL_NLM;
                if (!opt_value("-h_val")) stage_h_val = 3.0f; /* the PoC's h (NLM.py:12) */
            } else {
                fprintf(stderr, "-kernel %s: expected bilateral, photometric, spatial or nlm\n", v);
                FinalizeProgram();
                return 1;
            }
        }
    }
    GetFilePath(filename, sizeof(filename));

    int width = 0, height = 0;
    png_bytep *img_bytes = NULL, *output_img = NULL;
    if (opt_has("-color")) { /* python/image_processing.py:410-432 */
        unsigned cw = 0, ch = 0;
        png_bytep *in_rgb = NULL;
        png_bytep *out_rgb = ColorComputation(filename, &cw, &ch, &in_rgb);
        int cstatus = out_rgb ? 0 : 5;
        if (in_rgb && glf_write_png_rgb("results/input.png", in_rgb, cw, ch) != 0) cstatus = cstatus ? cstatus : 4;
        if (out_rgb && glf_write_png_rgb("results/output.png", out_rgb, cw, ch) != 0) cstatus = cstatus ? cstatus : 4;
        if (!in_rgb) {
            fprintf(stderr, "Could not read %s as an 8-bit gray / RGB / RGBA PNG\n", filename);
            cstatus = 1;
        }
        printf("Total computation time: %fs\n", wtime() - start_time);
        free_rows(in_rgb, (int)ch);
        free_rows(out_rgb, (int)ch);
        FinalizeProgram();
        return cstatus;
    }
    if (read_png(filename, &img_bytes, &width, &height) != 0) { /* ReadAndBcastImage :291; status checked here */
        fprintf(stderr, "Could not read %s as an 8-bit gray / RGB / RGBA PNG\n", filename);
        FinalizeProgram();
        return 1;
    }
    printf("Read image %s of size %dx%d => %d pixels\n", filename, width, height, width * height);

    int status = 0;
    if (opt_has("-no_approx")) { /* :294-297 */
        output_img = EntireComputation(img_bytes, (unsigned)width, (unsigned)height);
    } else if (ngpu > 0) {
        output_img = MultiComputation(img_bytes, (unsigned)width, (unsigned)height, ngpu,
                                      nb && strcmp(nb, "loopback") == 0 ? GLF_MULTI_LOOPBACK : GLF_MULTI_RCCL);
    } else if (opt_has("-fused")) {
        output_img = FusedComputation(img_bytes, (unsigned)width, (unsigned)height);
    } else {
        output_img = ApproximationComputation(img_bytes, (unsigned)width, (unsigned)height); /* :300 */
    }

    if (write_png("results/input.png", img_bytes, (unsigned)width, (unsigned)height) != 0) status = status ? status : 4; /* :306 */
    if (output_img) {
        if (write_png("results/output.png", output_img, (unsigned)width, (unsigned)height) != 0) status = status ? status : 4; /* :309 */
    } else if (!status) {
        status = 5;
    }
    if (output_img && opt_has("-dump_residual")) { /* residuals(y, z), python/image_processing.py:378-380 */
        int lo = 255, hi = 0;
        for (int r = 0; r < height; ++r)
            for (int c = 0; c < width; ++c) {
                const int d = abs((int)img_bytes[r][c] - (int)output_img[r][c]);
                lo = d < lo ? d : lo;
                hi = d > hi ? d : hi;
            }
        png_bytep *res = (png_bytep *)malloc(sizeof(png_bytep) * (size_t)height);
        for (int r = 0; res && r < height; ++r) {
            res[r] = (png_bytep)malloc((size_t)width);
            for (int c = 0; res[r] && c < width; ++c) {
                const int d = abs((int)img_bytes[r][c] - (int)output_img[r][c]);
                res[r][c] = (png_byte)(hi > lo ? ((d - lo) * 255 + (hi - lo) / 2) / (hi - lo) : 0);
            }
        }
        if (res && write_png("results/residuals.png", res, (unsigned)width, (unsigned)height) != 0) status = status ? status : 4;
        free_rows(res, height);
        printf("Residual |input - output|: min %d, max %d grey levels\n", lo, hi);
    }
    printf("Total computation time: %fs\n", wtime() - start_time); /* :314 */

    free_rows(img_bytes, height);
    free_rows(output_img, height);
    FinalizeProgram();
    return status;
}
