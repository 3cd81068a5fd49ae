/*
 * glf.h -- C-ABI of the MI355X-native graph-Laplacian image filter.
 *
 * Drop-in boundary for the approximate path of the reference program
 * hpc/image_processing (David-Wobrock/image-processing-graph-laplacian).
 * The reference has no FFI; its boundary is the per-stage C prototypes in
 * the hpc/ headers, all of which take PETSc Mat/Vec. Here Mat/Vec become flat HIP
 * device buffers described by the plain struct glf_mat; every entry point is
 * extern "C", takes plain pointers and sizes, and returns an int status
 * (0 = GLF_OK) instead of void. Each declaration cites the reference
 * interface it replaces (paths relative to the reference root).
 *
 * Layout conventions (all device matrices are float32, ROW-major):
 *   image        uint8  [height][width]            (x = idx / width is the row,
 *                                                   y = idx % width the column, hpc/utils.c:11-19)
 *   K_A, L_A     float  [p][lda]  lda = p rounded up to 64, zero padding (the eigen stages need it)
 *   X, Phi_A     float  [p][ld]   ld = m rounded up to 32, columns >= m are zero
 *   Phi          float  [N][ld]   row order GLF_ROWS_SAMPLE_FIRST or GLF_ROWS_RASTER
 *   K_B, L_B     never stored: a GLF_MAT_KERNEL_B descriptor (image + sample table + scale)
 *
 * Threading: one host thread drives one context; contexts are independent.
 * Multi-GPU: pixel rows are sharded over the ranks (see glf_image_processing); a rank is one GPU. The collectives
 * are RCCL calls issued by the library itself (glf_ctx_set_comm_rccl: one process per GPU; glf_multi_*: one process,
 * one thread per GPU), or callbacks the caller plugs in through glf_comm.
 */
#ifndef GLF_H
#define GLF_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define GLF_VERSION 100

/* ---- status codes ---------------------------------------------------------- */
enum {
    GLF_OK = 0,
    GLF_ERR_INVALID = -1,   /* bad argument / shape mismatch */
    GLF_ERR_NOMEM = -2,     /* hipMalloc / malloc failed */
    GLF_ERR_HIP = -3,       /* a HIP call or kernel launch failed */
    GLF_ERR_NODEVICE = -4,  /* no usable gfx950 device */
    GLF_ERR_COMM = -5,      /* a glf_comm callback failed */
    GLF_ERR_NOCONV = -6,    /* eigensolver hit max_outer / inner limit */
    GLF_ERR_IO = -7,        /* read_png / write_png */
    GLF_ERR_UNSUPPORTED = -8
};
const char *glf_strerror(int status);

/* ---- context ---------------------------------------------------------------- */
typedef struct glf_ctx glf_ctx; /* opaque: device, stream, workspace, comm */

/* Replaces InitProgram (hpc/image_processing.c:30-38: SlepcInitialize/MPI).
 * stream: a hipStream_t to launch on (e.g. torch's current stream), or NULL
 * for a stream owned by the context. */
int glf_ctx_create(glf_ctx **ctx, int device, void *hip_stream);
int glf_ctx_destroy(glf_ctx *ctx);      /* replaces SlepcFinalize, :332 */
int glf_ctx_synchronize(glf_ctx *ctx);
const char *glf_ctx_last_error(const glf_ctx *ctx);
/* Device name / CU count / memory, for reports. */
int glf_ctx_device_info(const glf_ctx *ctx, char *name, size_t name_len, int *num_cus,
                        size_t *total_mem_bytes);
/* Kernel selection. Several stages have more than one implementation of the same sums (grid-factored vs entry by entry, two
 * row-pass shapes, ...); the default picks by problem size. key / value (value NULL, "" or "auto" = default):
 *   NYS_PATH  band | rank | grid | direct     DEG_PATH  grid | direct     MV_PATH  band | rank | grid | dense
 *   ROWPASS, ROWPASS_OP  rt | v1     COLPASS  ws | v1     SWEEP_COLPASS  segments | samples
 *   NYS_NO_LUT, NO_ECR, NO_NARROW, NO_FUSED_FILTER, ZMFMA_GROUPS, VERBOSE  1 | 0     EIG_SHARD  1 | 0       GS        seq | gram          RESIDUAL  sweep | derived
 * At context creation each key is initialised from the environment variable GLF_<KEY> (read once; nothing reads the
 * environment per call). No reference counterpart (PETSc's -ksp_type / -pc_type options database is the nearest thing). */
int glf_ctx_set_tuning(glf_ctx *ctx, const char *key, const char *value);
/* Debug workspace pool (environment GLF_POOL_DEBUG=1 at context creation; no reference counterpart): every work
 * buffer is allocated at its exact size followed by a 4 KiB guard zone of canary bytes, floating-point buffers are
 * handed out filled with NaN instead of whatever an earlier call left in them, nothing is reused, and the guard is
 * verified when the buffer is released. Returns the number of guard zones found overwritten so far (0 = no kernel
 * wrote past the end of a work buffer), -1 when the context was not created in debug mode. */
int glf_ctx_debug_violations(const glf_ctx *ctx);
/* Work buffers are cached between calls (a second image of the same size allocates nothing). Bytes currently cached and not
 * in use; a cached buffer that no call has taken for 64 public calls is returned to the driver at the next allocation, so a
 * process that walks through many image sizes does not keep every size's buffers. No reference counterpart. */
size_t glf_ctx_cached_bytes(const glf_ctx *ctx);

/* Collectives supplied by the caller (replace the MPI_Allreduce / allgather
 * inside PETSc's VecDot, VecSum, MatMult: hpc/gram_schmidt.c:14-15,
 * hpc/utils.c:382, hpc/inverse_power_it.c:167). Buffers are DEVICE pointers,
 * operated in place, ordered on the context's stream. size == 1 or NULL
 * callbacks mean "single GPU". Return 0 on success. */
typedef struct glf_comm {
    int rank, size;
    int (*allreduce_sum_f32)(void *user, float *dbuf, size_t count);
    int (*allreduce_sum_f64)(void *user, double *dbuf, size_t count);
    /* In-place all-gather: dbuf holds size * count_per_rank floats, rank r's block at offset
     * r * count_per_rank (replaces the allgather inside PETSc's MPIDENSE MatMult). Optional: with
     * NULL the eigen-solve is replicated on every rank instead of row-sharded. */
    int (*allgather_f32)(void *user, float *dbuf, size_t count_per_rank);
    void *user;
} glf_comm;
int glf_ctx_set_comm(glf_ctx *ctx, const glf_comm *comm);

/* ---- native collectives: RCCL inside the library ------------------------------------------------------------------
 * The reference is an MPI program (hpc/image_processing.c:30-38 SlepcInitialize/MPI_Init, :45-76 broadcast of the image;
 * every Mat is MATMPIDENSE, hpc/affinity.c:138-142). Here a rank is a GPU and the collectives are RCCL calls the library
 * issues on the context's stream -- no callback into the host language on the data path:
 *   all-reduce (f64)  degree partial sums (p), Phi^T y (ld), every inner product / norm / Gram block of the eigen-solve
 *   all-reduce (f32)  X^T A X of the residual (ld x ld)
 *   all-gather (f32)  the operand block of each L_A application (hpc/inverse_power_it.c:167: PETSc's MPIDENSE MatMult
 *                     gathers x the same way) and the final eigenvectors
 * One process per GPU: glf_rccl_unique_id on one rank, distribute the bytes (MPI / torch.distributed / a file), then
 * glf_ctx_set_comm_rccl on every rank (ncclCommInitRank; collective over all ranks). force != 0 keeps the collectives
 * in place on a one-rank world (plumbing test on a single GPU). */
#define GLF_RCCL_ID_BYTES 128
int glf_rccl_unique_id(void *id_out, size_t bytes);
int glf_ctx_set_comm_rccl(glf_ctx *ctx, int rank, int size, const void *unique_id, size_t bytes, int force);

/* One PROCESS driving n GPUs -- the reference's `mpirun -n N image_processing` as `image_processing -ngpu N`: one
 * context and one host thread per device. GLF_MULTI_RCCL: ncclCommInitAll over `devices` (distinct GPUs of one node,
 * xGMI). GLF_MULTI_LOOPBACK: the same collectives staged through host memory between the rank threads in fixed rank
 * order; ranks may share a device (devices[i] may repeat), which RCCL refuses -- this is how the N > 1 sharding of the C
 * path is tested on a one-GPU box. devices == NULL: 0 .. n-1. */
/* info = {rank, size, backend (0 none / caller's callbacks, 1 RCCL, 2 loopback), ranks the RCCL communicator itself reports
 * (ncclCommCount; 0 when not RCCL)} */
int glf_ctx_comm_info(glf_ctx *ctx, int info[4]);
/* collectives issued through the library's own communicator since the last reset:
 * out = {all-reduce calls, all-reduce bytes, all-gather calls, all-gather bytes received} */
int glf_ctx_comm_counters(glf_ctx *ctx, unsigned long long out[4], int reset);

typedef struct glf_multi glf_multi;
enum { GLF_MULTI_RCCL = 0, GLF_MULTI_LOOPBACK = 1 };
int glf_multi_create(glf_multi **w, int n, const int *devices, int backend);
int glf_multi_destroy(glf_multi *w);
int glf_multi_size(const glf_multi *w);
glf_ctx *glf_multi_ctx(glf_multi *w, int rank);
const char *glf_multi_last_error(const glf_multi *w);
/* glf_multi_image_processing: declared below, after glf_options / glf_stats. */

/* How the Nystroem contraction L_B^T (phi_A Pi^-1) (hpc/nystroem.c:42) is evaluated:
 *  F32_MFMA   v_mfma_f32_32x32x2_f32, operands exactly f32 (shares the f32 FMA pipe with the
 *             kernel generation: about 75 % of the f32 matrix peak);
 *  F16_SPLIT  both operands split into f16 (hi, lo) pairs carrying 22 significant bits, three
 *             v_mfma_f32_32x32x16_f16 products, f32 accumulation; runs on the f16 matrix pipe so the
 *             MFMAs overlap the VALU generation (2.5x faster). Default; override per context here or
 *             with the environment variable GLF_CONTRACTION=f32|f16s at context creation. */
enum { GLF_CONTRACT_F32_MFMA = 1, GLF_CONTRACT_F16_SPLIT = 2 };
int glf_ctx_set_contraction(glf_ctx *ctx, int mode);
/* The pixel-row shard of rank `rank` of `size`: rows [*row0, *row1) = [rank*height/size, (rank+1)*height/size).
 * Host only (replaces PETSc's PETSC_DECIDE row ownership, hpc/utils.c:463-475). */
int glf_shard_rows(int height, int rank, int size, int *row0, int *row1);

/* Flat device buffers (replace MatCreate/VecCreate + MatDestroy/VecDestroy). */
int glf_malloc(glf_ctx *ctx, void **dptr, size_t bytes);
int glf_free(glf_ctx *ctx, void *dptr);
int glf_memcpy_h2d(glf_ctx *ctx, void *dst, const void *src, size_t bytes);
int glf_memcpy_d2h(glf_ctx *ctx, void *dst, const void *src, size_t bytes);
int glf_memset(glf_ctx *ctx, void *dst, int value, size_t bytes);

/* ---- matrices ----------------------------------------------------------------- */
enum { GLF_MAT_DENSE = 0, GLF_MAT_DIAG = 1, GLF_MAT_KERNEL_B = 2 };
enum { GLF_ROWS_NA = 0, GLF_ROWS_SAMPLE_FIRST = 1, GLF_ROWS_RASTER = 2 };
/* NLM: non-local means, 7 x 7 Gaussian-weighted patches of the symmetrically padded image, K = exp(-|| G o (P_i - P_j) ||^2 / h_val^2)
 * (python/affinity_methods/NLM.py:9-34, where h = 3; the C reference has bilateral / photometric / spatial only, hpc/affinity.c:8-121) */
enum { GLF_KERNEL_BILATERAL = 0, GLF_KERNEL_PHOTOMETRIC = 1, GLF_KERNEL_SPATIAL = 2, GLF_KERNEL_NLM = 3 };

/* Replaces PETSc Mat (MATMPIDENSE / MATMPIAIJ diagonal). */
typedef struct glf_mat {
    int32_t kind;        /* GLF_MAT_* */
    int32_t row_order;   /* GLF_ROWS_* for N x m eigenvector matrices */
    int64_t rows, cols;  /* logical shape */
    int64_t ld;          /* floats between consecutive rows (DENSE) */
    float *data;         /* device; DENSE: rows*ld floats; DIAG: rows floats; KERNEL_B: NULL */
    int32_t owns_data;   /* glf_mat_destroy frees data */
    /* generator descriptor (KERNEL_B): entry (i, col) = scale * K(sample i, pixel col) */
    const uint8_t *img;  /* device image */
    const float *samples;/* device float4 per sample: {row, col, value, 0}, zero-padded to 64 records */
    const uint8_t *mask; /* device uint8[N]: 1 at sample pixels */
    const uint32_t *idx; /* device sample indices, ascending */
    int32_t width, height;
    uint32_t p;
    float scale;         /* 1 for K_B, -alpha for L_B (hpc/laplacian.c:37-38) */
    float h_loc, h_val;  /* hpc/affinity.c:117-118 */
    int32_t kernel;      /* GLF_KERNEL_* (hpc/affinity.c:119-121) */
    double *degree;      /* device double[p]: row sums of [K_A K_B] cached by
                            glf_ComputeAffinityMatrices (hpc/laplacian.c:18-20) */
    int32_t owns_desc;   /* glf_mat_destroy frees samples/mask/idx/degree */
} glf_mat;

int glf_mat_create_dense(glf_ctx *ctx, glf_mat *mat, int64_t rows, int64_t cols, int64_t ld);
int glf_mat_create_diag(glf_ctx *ctx, glf_mat *mat, int64_t n);
int glf_mat_destroy(glf_ctx *ctx, glf_mat *mat); /* MatDestroy */
/* MatGetColumnVector (hpc/display.c:95): column `col` of a dense matrix into host_out[rows]. */
int glf_mat_get_column(glf_ctx *ctx, const glf_mat *mat, int64_t col, float *host_out);

/* ---- host-side stages -------------------------------------------------------- */

/* void Sampling(int, int, unsigned*, unsigned**)  hpc/sampling.h:1, hpc/sampling.c:6-33.
 * *sample_indices is malloc'd; release with glf_host_free. */
int glf_Sampling(int width, int height, unsigned *sample_size, unsigned **sample_indices);
void glf_host_free(void *ptr);

/* BuildRandomVectors, hpc/inverse_power_it.c:12-47: X0[m][p] (vector after
 * vector) = U[0,1) from xoshiro256** seeded with splitmix64(seed). */
int glf_random_vectors(double *X0, unsigned p, unsigned m, uint64_t seed);

/* Synthetic noisy test image of the benchmark configs; not part of
 * the reference. out: height*width bytes. */
int glf_synth_image(uint8_t *out, int width, int height, uint64_t seed);

/* ---- device stages (mirror the hpc/ headers, "glf_" prefixed) -------------------------- */

/* void ComputeAffinityMatrices(Mat* K_A, Mat* K_B, const png_bytep* img, int w, int h,
 *                              unsigned p, const unsigned* idx)   hpc/affinity.h:5, hpc/affinity.c:129-262
 * d_img: device image; sample_indices: HOST array (as in the reference).
 * K_A: dense p x p (allocated here). K_B: KERNEL_B descriptor whose cached
 * degree holds this rank's partial row sums of [K_A K_B] over its pixel rows,
 * already all-reduced when a comm is set. kernel = GLF_KERNEL_*. Pass K_A == NULL
 * to skip materialising K_A. */
int glf_ComputeAffinityMatrices(glf_ctx *ctx, glf_mat *K_A, glf_mat *K_B, const uint8_t *d_img,
                                int width, int height, unsigned sample_size,
                                const unsigned *sample_indices, int kernel, float h_loc, float h_val);

/* void ComputeLaplacianMatrix(Mat* L_A, Mat* L_B, Mat K_A, Mat K_B)  hpc/laplacian.h:3, hpc/laplacian.c:14-42
 * L_A = alpha (diag(D_A) - K_A) dense, L_B = KERNEL_B descriptor with scale -alpha
 * (shares K_B's tables). K_A may be NULL (entries regenerated). alpha_out optional. */
int glf_ComputeLaplacianMatrix(glf_ctx *ctx, glf_mat *L_A, glf_mat *L_B, const glf_mat *K_A,
                               const glf_mat *K_B, double *alpha_out);

typedef struct glf_eig_stats {
    int32_t outer_its, inner_its_total;
    double residual;
    /* the L_A sweeps (block mat-vecs) of the solve: launches, summed device ms (HIP events around the sweep kernel on
     * the context's stream) and the L_A bytes this rank streamed (4 p rows_of_the_rank per sweep) */
    int32_t matvecs;
    float matvec_ms;
    double matvec_bytes;
    int32_t narrow_sweeps;  /* of those sweeps, the ones applied to a packed block of the still-iterating columns only (block PCG) */
    int32_t reserved;
} glf_eig_stats;

/* void InversePowerIteration(const Mat A, unsigned m, Mat* eigvecs, Mat* eigvals,
 *                            PetscBool optiGS, PetscScalar eps)  hpc/inverse_power_it.h:3, hpc/inverse_power_it.c:86-252
 * X0: HOST double [m][p] start block (glf_random_vectors) or NULL for seed 1.
 * eigenvectors: dense p x m (ld = m rounded to 32), the normalised
 * pre-orthogonalisation iterates (:171,:230); eigenvalues: DIAG m = 1/norms (:204).
 * inner_rtol stands in for PETSc's KSP rtol default 1e-5. */
int glf_InversePowerIteration(glf_ctx *ctx, const glf_mat *A, unsigned m, glf_mat *eigenvectors,
                              glf_mat *eigenvalues, int optiGramSchmidt, double epsilon,
                              double inner_rtol, int max_outer, const double *X0,
                              glf_eig_stats *stats);

/* void OrthonormaliseVecs(Vec* X, unsigned n, unsigned p, PetscScalar* norms)  hpc/gram_schmidt.h:4, hpc/gram_schmidt.c:29-64
 * X: dense n x p (row-major, p vectors as columns); norms: HOST double[p] or NULL. */
int glf_OrthonormaliseVecs(glf_ctx *ctx, glf_mat *X, double *norms);
/* void NormaliseVecs(Vec* X, unsigned p, PetscScalar* norms)  hpc/gram_schmidt.h:5 */
int glf_NormaliseVecs(glf_ctx *ctx, glf_mat *X, double *norms);

/* Mat InverseDiagMat(Mat x)  hpc/utils.h (hpc/utils.c:559-586) */
int glf_InverseDiagMat(glf_ctx *ctx, const glf_mat *x, glf_mat *inv);

/* ---- the PoC's balancing steps of the approximated affinity (inactive experiments in the PoC) ----------
 * sinkhorn(phi, Pi)  python/image_processing.py:90-98: the alternating scalings of K = phi diag(Pi) phi^T, K never formed
 * (200 products K x = phi (Pi o (phi^T x)) for the PoC's 100 iterations). phi: dense N x m in any row order, Pi: diagonal m;
 * d_r, d_c: device double[N]. */
int glf_Sinkhorn(glf_ctx *ctx, const glf_mat *phi, const glf_mat *Pi, int iterations, double *d_r, double *d_c);
/* :99-102: rows [row0, row0 + nrows) of W_AB = diag(r) K diag(c) into d_out (device double [nrows][N]); the PoC keeps the
 * first m rows and splits them into W_A = W_AB[:, :m], W_B = W_AB[:, m:] (:103-104). */
int glf_SinkhornRows(glf_ctx *ctx, const glf_mat *phi, const glf_mat *Pi, const double *d_r, const double *d_c, int64_t row0, int nrows,
                     double *d_out);
/* orthogonalisation(A, B)  :110-127: V = [A ; B^T] A^-1/2 phi_Q Pi_Q^-1/2 with Q = A + A^-1/2 B B^T A^-1/2.
 * d_A: device double n x n (symmetric positive definite), d_B: device double n x q (row-major); d_V: device double
 * (n + q) x n; h_Pi: HOST double[n], clipped at 1. Dense f64 with a one-workgroup Jacobi eigensolver: n <= 512. */
int glf_Orthogonalisation(glf_ctx *ctx, const double *d_A, int n, const double *d_B, int q, double *d_V, double *h_Pi);

/* Mat Nystroem(Mat B, Mat phi_A, Mat Pi_A_Inv, unsigned N, unsigned n, unsigned p)  hpc/nystroem.h:3, hpc/nystroem.c:5-69
 * B: KERNEL_B descriptor (L_B). phi (allocated here): N x m, SAMPLE-FIRST rows:
 * [phi_A ; B^T (phi_A Pi_A_Inv)]. */
int glf_Nystroem(glf_ctx *ctx, const glf_mat *B, const glf_mat *phi_A, const glf_mat *Pi_A_Inv,
                 glf_mat *phi);

/* Mat Permutation(Mat m, const unsigned* idx, unsigned p)  hpc/utils.h:18, hpc/utils.c:134-173
 * sample-first rows -> raster rows. sample_indices: HOST. */
int glf_Permutation(glf_ctx *ctx, const glf_mat *in, const unsigned *sample_indices,
                    unsigned num_sample_indices, glf_mat *out);

/* png_bytep* ComputeResultFromLaplacian(const png_bytep* img, Mat phi, Mat Pi, unsigned w, unsigned h)
 * hpc/display.h:13, hpc/display.c:58-83 (+ AboveXSetY hpc/utils.c:652, OneColMat2pngbytes :492-534).
 * phi: N x m RASTER rows; Pi: DIAG m (f(eigenvalues)); gain: 3.0 (:73).
 * d_out: device uint8[N]; d_zf: optional device float[N] (z before clamp/cast). */
int glf_ComputeResultFromLaplacian(glf_ctx *ctx, const uint8_t *d_img, const glf_mat *phi,
                                   const glf_mat *Pi, unsigned width, unsigned height, float gain,
                                   uint8_t *d_out, float *d_zf);

/* ---- whole path ----------------------------------------------------------------- */

enum { GLF_FILTER_REFERENCE = 0, GLF_FILTER_POC = 1, GLF_FILTER_SMOOTH = 2, GLF_FILTER_SHARPEN = 3 };
typedef struct glf_options {
    uint32_t struct_size;   /* sizeof(glf_options) */
    uint32_t num_samples;   /* requested sample count; 0 -> width*height*sample_frac (hpc/image_processing.c:187) */
    double sample_frac;     /* 0.01 */
    uint32_t num_eigvals;   /* -num_eigvals; 0 or >= p -> p-1 (hpc/image_processing.c:96-108) */
    int32_t opti_gs;        /* -opti_gs, < 1 -> 1 (:128-140) */
    double epsilon;         /* -inv_it_epsilon, default 0.1 (:142-154) */
    double inner_rtol;      /* 1e-5 (PETSc KSP default) */
    int32_t max_outer;      /* safety cap on outer iterations (reference: none) */
    uint64_t seed;          /* X0 stream */
    float gain;             /* 3.0 hpc/display.c:73 */
    float h_loc, h_val;     /* 40, 30 hpc/affinity.c:117-118 */
    int32_t kernel;         /* GLF_KERNEL_BILATERAL */
    int32_t filter_pow;     /* 1: f(Pi) = Pi (MatPow is a no-op, hpc/utils.c:721); k: Pi^k */
    int32_t filter_mode;    /* GLF_FILTER_REFERENCE (0): z = y + gain Phi Pi^filter_pow Phi^T y, clamp, cast (hpc/display.c:58-83);
                               GLF_FILTER_POC (1): the Python PoC's active filter z = y - Phi diag(mu + 5) Phi^T y
                               (python/image_processing.py:304-305; gain and filter_pow ignored), same clamp and cast;
                               GLF_FILTER_SMOOTH (2): z = Phi diag(1 - mu) Phi^T y -- the PoC's `smoothing` (:197-219), W = I - L with
                               the eigenpairs of the renormalised Laplacian this path computes (W's are (1 - mu, the same vectors);
                               the PoC takes them by a dense eigh of W_A and the same Nystroem extension);
                               GLF_FILTER_SHARPEN (3): z = (1 + beta) W^2 y - beta W^3 y (:222-241) as Phi diag((1 + beta) s^2 -
                               beta s^3) Phi^T y, s = 1 - mu, beta = filter_beta. Neither adds y; gain and filter_pow ignored */
    int32_t skip_exact_zeros; /* 0 (default): every K_B / K_A entry is evaluated, as the reference does.
                                 1: entries that are exactly zero in the arithmetic in use (pixel-sample
                                 distance beyond the radius where exp underflows) are skipped in whole tiles;
                                 the output is bit-identical, the work is not -- see glf_stats.*_evaluated. */
    float filter_beta;      /* GLF_FILTER_SHARPEN: 1.5 (python/image_processing.py:231) */
    int32_t sampling;       /* GLF_SAMPLING_UNIFORM (0): the reference's grid (hpc/sampling.c:6-23); GLF_SAMPLING_RANDOM (1): the PoC's
                               random sampler (python/sampling/random.py:8-16: num_samples distinct pixels, ascending) -- not a tensor
                               grid, so the entry-by-entry kernels and a stored L_A run */
    uint32_t reserved_;
    uint64_t sampling_seed; /* GLF_SAMPLING_RANDOM: seed of the library's own generator (the PoC draws from an unseeded numpy stream) */
} glf_options;
enum { GLF_SAMPLING_UNIFORM = 0, GLF_SAMPLING_RANDOM = 1 };
/* python/sampling/random.py:8-16: *sample_size distinct pixel indices drawn uniformly (xoshiro256** seeded with `seed`), sorted
 * ascending; malloc'd like glf_Sampling's. */
int glf_RandomSampling(int width, int height, unsigned *sample_size, unsigned **sample_indices, uint64_t seed);
void glf_options_default(glf_options *opt);

typedef struct glf_stats {
    uint32_t p, m;
    double alpha;
    glf_eig_stats eig;
    /* device milliseconds (HIP events on the context's stream) */
    float ms_affinity, ms_laplacian, ms_eigen, ms_nystroem, ms_filter, ms_total;
    /* dominant kernel (Nystroem contraction): launches and summed device ms */
    int32_t nystroem_launches;
    float nystroem_kernel_ms;
    int32_t row0, row1;     /* this rank's pixel rows */
    int32_t contraction;    /* GLF_CONTRACT_* actually used */
    int32_t skip_exact_zeros;
    /* kernel entries covered by this rank (dense: p * pixels of the rank), evaluated one by one (direct kernels)
     * or through the factored sums of the grid forms */
    double nystroem_evaluated, degree_evaluated;
    /* f16 MFMA flops issued by the Nystroem contraction (3 products per split multiply-add) */
    double nystroem_mfma_flops;
    int32_t nystroem_path;  /* 0 direct kernel (K_B generated entry by entry), 1 grid-factored (all 256 grey levels),
                               3 grid-factored in rank form (photometric table as a rank-R expansion),
                               4 band form (entry by entry over the samples within the kernel's radius; then nystroem_evaluated
                               counts the entries the kernel evaluated and nystroem_colpass_* describe its launches, with
                               nystroem_colpass_flops = 2 ld x the (pixel, sample) pairs inside the radius) */
    int32_t matvec_path;    /* 0 stored L_A streamed per sweep, 1 L_A applied in grid-factored form (never stored), 3 the same in rank form,
                               4 L_A applied in band form (never stored) */
    /* grid-factored Nystroem: the row-pass kernel (k_grid_rowpass) alone -- launches, summed device ms (HIP events around
     * each launch) and its algorithmic flops 2 rows 256 nc nr ld (one product per multiply-add) */
    int32_t nystroem_rowpass_launches;
    float nystroem_rowpass_ms;
    double nystroem_rowpass_flops;
    /* rank form (nystroem_path 3): the fused T' + column-pass kernel (k_rank_colpass) -- launches, summed device ms (HIP
     * events around each launch), its algorithmic flops 2 ld ncs (sum over rows of (values present) R + pixels) (one product
     * per multiply-add: T' for the (row, value) pairs that occur, then the Ec contraction per pixel), and R, the terms of the
     * expansion of the photometric table (0: exact form) */
    int32_t nystroem_colpass_launches;
    float nystroem_colpass_ms;
    double nystroem_colpass_flops;
    int32_t rank_terms;
    /* 1: the filter ran in the epilogue of the band-form Nystroem kernel (Phi never written; c = Phi^T y from the degree stage's
     * value-weighted sums); ms_filter is then part of ms_nystroem. 0: Phi written, filter as its own stage */
    int32_t filter_fused;
    /* under a communicator: 1 the eigen-solve ran row-sharded (all-gather of the operand per L_A application, all-reduced inner
     * products), 0 every rank ran it on all rows (the default with the band form, whose sweep is cheaper than its all-gather) */
    int32_t eigen_sharded;
    int32_t reserved_;
} glf_stats;

/* ApproximationComputation, hpc/image_processing.c:183-277 (commented tail
 * :240-275 included). d_img: device uint8[height*width], replicated on every
 * rank. d_out: device uint8[height*width]; with a comm of size G rank g fills
 * image rows [g*height/G, (g+1)*height/G) only. d_zf optional float[N].
 * eigvals_out: HOST double[m] or NULL. */
int glf_image_processing(glf_ctx *ctx, const glf_options *opt, const uint8_t *d_img, int width,
                         int height, uint8_t *d_out, float *d_zf, double *eigvals_out,
                         glf_stats *stats);

/* By-products of one glf_image_processing call, for parity checks at sizes where the CPU oracle cannot run the whole
 * path (tests/test_gpu_large.py, bench.py's parity leg): the caller checks sampled rows of Phi / z against
 * hpc/nystroem.c:41-57 and hpc/display.c:58-83 evaluated on the CPU from these. Every pointer is optional. */
typedef struct glf_capture {
    uint32_t struct_size;  /* sizeof(glf_capture) */
    uint32_t ld;           /* out: row stride of phi_A / phi (m rounded up to 32, 64, 128 or 256) */
    float *d_phi_A;        /* device [p rounded up to 64][ld]: the eigenvectors (hpc/inverse_power_it.c:213-241) */
    size_t phi_A_floats;   /* capacity of d_phi_A in floats (GLF_ERR_INVALID when too small) */
    float *d_phi;          /* device [pixels of this rank][ld]: Phi in raster order after Nystroem + Permutation */
    size_t phi_floats;     /* capacity of d_phi in floats */
    double *h_c;           /* host [ld]: right = Phi^T y (hpc/display.c:66), summed over all ranks */
    double *h_degree;      /* host [p]: D_A = rowsum [K_A K_B] (hpc/laplacian.c:18-20), summed over all ranks */
    float *d_corr;         /* device [pixels of this rank]: the correction gain * Phi (f(Pi) Phi^T y) = z - y before it is added to y
                              (hpc/display.c:64-73); the float z resolves it to ulp(z) ~ 4e-6 grey levels only */
    size_t corr_floats;    /* capacity of d_corr in floats */
} glf_capture;
int glf_image_processing_capture(glf_ctx *ctx, const glf_options *opt, const uint8_t *d_img, int width, int height,
                                 uint8_t *d_out, float *d_zf, double *eigvals_out, glf_stats *stats, glf_capture *cap);

/* Throughput mode for a batch of equally sized tiles (BASELINE.json configs[4]: "batch of 64 x 1024x1024 noisy tiles
 * sharing one sample set"; hpc/sampling.c:6-23 gives tiles of one size the same sample grid). The reference would run its
 * main once per tile (hpc/image_processing.c:279-335); here tile t = d_imgs + t*width*height goes through
 * glf_image_processing unchanged on one of `nctx` contexts (one stream + one host thread each, tiles dealt dynamically),
 * so every output is bit-identical to the single-image call's. Replicas only: the contexts must not carry a comm.
 * d_outs: uint8[ntiles*height*width]; d_zfs (optional) float[ntiles*N]; stats (optional) HOST glf_stats[ntiles].
 * On failure returns the first failing tile's status; the message is in ctxs[0]'s last error. */
int glf_image_processing_batch(glf_ctx *const *ctxs, int nctx, const glf_options *opt, const uint8_t *d_imgs,
                               int width, int height, int ntiles, uint8_t *d_outs, float *d_zfs, glf_stats *stats);

/* ReadAndBcastImage + ApproximationComputation + the gather of the result (hpc/image_processing.c:45-76, 183-277,
 * hpc/utils.c:502-527): h_img / h_out are HOST buffers of height*width bytes; every rank receives the whole image, runs
 * glf_image_processing on its pixel rows and writes them into h_out (h_zf optional float[N]). stats: HOST glf_stats[n]
 * or NULL; eigvals_out: HOST double[m] or NULL (rank 0's, identical on every rank). */
int glf_multi_image_processing(glf_multi *w, const glf_options *opt, const uint8_t *h_img, int width, int height,
                               uint8_t *h_out, float *h_zf, double *eigvals_out, glf_stats *stats);

/* EntireComputation, hpc/image_processing.c:155-181 (-no_approx): z = clamp(y - L y) with the full N x N
 * Laplacian of ComputeEntireAffinityMatrix / ComputeEntireLaplacianMatrix / ComputeResultFromEntireLaplacian
 * (hpc/affinity.c:264-336, hpc/laplacian.c:44-65, hpc/display.c:128-149). The matrices are never stored
 * ((L y)_i = alpha (D_i y_i - (K y)_i)); O(N^2) work, limited to 4 Mpixel. d_zf, alpha_out optional. */
int glf_EntireComputation(glf_ctx *ctx, const uint8_t *d_img, int width, int height, int kernel, float h_loc, float h_val,
                          uint8_t *d_out, float *d_zf, double *alpha_out);

/* ---- image I/O (host) -------------------------------------------------------------- */
/* int read_png(const char*, png_bytep** rows, int* w, int* h)  hpc/read_img.h:3, hpc/read_img.c:9-65
 * rows: malloc'd array of `height` malloc'd rows of `width` bytes (gray 8);
 * RGB / RGBA are converted to gray with libpng's default rgb_to_gray weights
 * (hpc/read_img.c:47-50). Returns 0 / -1. */
int glf_read_png(const char *filename, uint8_t ***row_pointers, int *width, int *height);
/* int write_png(const char*, png_bytep* rows, unsigned w, unsigned h)  hpc/write_img.h:4, hpc/write_img.c:5-53 */
int glf_write_png(const char *filename, uint8_t **img_bytes, unsigned width, unsigned height);
/* Colour (python/image_processing.py:410-432: the PoC filters the luma of an RGB image and keeps the chroma; the C reference
 * converts to gray on read): the same codec with rows of 3 * width bytes, R G B interleaved. */
int glf_read_png_rgb(const char *filename, uint8_t ***row_pointers, int *width, int *height);
int glf_write_png_rgb(const char *filename, uint8_t **img_bytes, unsigned width, unsigned height);

#ifdef __cplusplus
}
#endif
#endif /* GLF_H */
