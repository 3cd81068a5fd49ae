/*
 * stages.h -- the reference's stage prototypes (hpc/ headers) re-declared over
 * the C-ABI: Mat is a pointer to a flat-device-buffer descriptor instead of a
 * PETSc object, every stage returns an int status instead of void, and the
 * communicator is the process-wide context set by InitProgram (the reference
 * uses PETSC_COMM_WORLD the same way).
 */
#ifndef GLF_HOST_STAGES_H
#define GLF_HOST_STAGES_H

#include "glf.h"

typedef unsigned char png_byte;   /* as in png.h */
typedef png_byte *png_bytep;
typedef glf_mat *Mat;             /* replaces PETSc Mat (MATMPIDENSE / diagonal MATMPIAIJ) */

glf_ctx *glf_world(void);         /* replaces PETSC_COMM_WORLD */
/* hpc/affinity.c:117-118 and hpc/display.c:73 hard-code 40, 30 and 3.0; these default to the same values */
extern float stage_h_loc, stage_h_val, stage_gain;
extern int stage_sampling;       /* GLF_SAMPLING_*: the reference's grid (hpc/sampling.c) or the PoC's random sampler (python/sampling/random.py) */
extern unsigned long long stage_sampling_seed;
extern int stage_kernel;         /* GLF_KERNEL_*: the kernel the reference selects by (un)commenting hpc/affinity.c:119-121 */

int InitProgram(int device);      /* hpc/image_processing.c:30-38 */
void FinalizeProgram(void);       /* SlepcFinalize, hpc/image_processing.c:332 */
void MatDestroy(Mat *m);

/* hpc/read_img.h:3, hpc/write_img.h:4 */
int read_png(const char *filename, png_bytep **row_pointers, int *width, int *height);
int write_png(const char *filename, png_bytep *img_bytes, unsigned int width, unsigned int height);

/* hpc/sampling.h:1 */
void Sampling(int width, int height, unsigned int *sample_size, unsigned int **sample_indices);
/* hpc/affinity.h:5 */
int ComputeAffinityMatrices(Mat *K_A, Mat *K_B, const png_bytep *img_bytes, int width, int height,
                            unsigned int sample_size, const unsigned int *sample_indices);
/* hpc/laplacian.h:3 */
int ComputeLaplacianMatrix(Mat *L_A, Mat *L_B, Mat K_A, Mat K_B);
/* hpc/inverse_power_it.h:3 (optiGramSchmidt is used as an int modulus, survey quirk Q9) */
int InversePowerIteration(const Mat A, unsigned int m, Mat *eigenvectors, Mat *eigenvalues,
                          int optiGramSchmidt, double epsilon);
/* hpc/utils.h: InverseDiagMat, MatPow (a no-op in the reference: hpc/utils.c:721), Permutation */
Mat InverseDiagMat(Mat x);
Mat MatPow(Mat A, double x);
Mat Permutation(Mat m, const unsigned int *sample_indices, unsigned int num_sample_indices);
/* hpc/nystroem.h:3 */
Mat Nystroem(Mat B, Mat phi_A, Mat Pi_A_Inv, unsigned int N, unsigned int n, unsigned int p);
/* hpc/display.h:13, :8 */
png_bytep *ComputeResultFromLaplacian(const png_bytep *img_bytes, Mat phi, Mat Pi, unsigned int width,
                                      unsigned int height);
int WriteDiagMat(Mat x, const char *filename);
/* hpc/display.h:10-11 (diagnostics of the commented tail, hpc/image_processing.c:252-260) */
int WriteMatCol(Mat x, unsigned int col_num, const char *filename);
int WritePngMatCol(Mat x, unsigned int col_num, unsigned int width, unsigned int height, const char *filename);

const glf_eig_stats *LastEigStats(void);

#endif
