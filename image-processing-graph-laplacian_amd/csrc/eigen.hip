// eigen.hip -- inverse subspace iteration on L_A (hpc/inverse_power_it.c:86-252) with
// classical Gram-Schmidt (hpc/gram_schmidt.c:29-64), on flat device buffers.
//
// Layout: A = L_A is p x p row-major with lda >= round_up(p,64), lda % 4 == 0 and ZERO padding
// columns (or, row-sharded over ranks, the column block described by MatShard). Vector blocks
// (X, P, R, AP ...) are [rows][ld] row-major, rows = p rounded up to 64 (more when sharded, see
// vec_rows), ld = m rounded up to a power of two in {32..256}; padding rows/columns are zero and
// never written. One "Vec" of the reference = one column here.
//
// Kernels:
//   k_block_matvec_f16s  Y = A X   split-f16 MFMA, A read through its symmetry, HBM bound (default)
//   k_block_matvec       Y = A X   f32-input MFMA (GLF_CONTRACT_F32_MFMA), exact f32 operands
//   k_gram               G = X^T Y f32 MFMA, split over row chunks
//   k_cg_*               Jacobi-PCG vector updates with per-column scalars
//   k_gs_*               classical Gram-Schmidt column sweeps
// All reductions go through fixed-order f64 partial buffers => reproducible.
#include "glf_internal.hpp"

#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <cstring>

namespace glf {

typedef float f32x16 __attribute__((ext_vector_type(16)));

// =====================================================================================
// Y[p32][ld] = A[p][lda] * X[p32][ld]
// =====================================================================================
// Workgroup = 4 waves x 32 rows. Each lane loads 16 consecutive floats of ITS row per
// 32-wide K tile (lanes 0-31: k 0..15, lanes 32-63: k 16..31), which are directly
// the A operands of 16 v_mfma_f32_32x32x2_f32 (lane l supplies A[i=l&31][k=l>>5]):
// MFMA t pairs k = kb+t (lower half) with k = kb+16+t (upper half); the B operand
// X[k][j] is read from an LDS copy of the X tile with the same pairing. No LDS
// round trip for A, one 128-B line per row per tile.

template <int MB> // MB = ld / 32 column blocks
__global__ __launch_bounds__(256) void k_block_matvec(const float *__restrict__ A, int64_t lda, unsigned p,
                                                       unsigned p32, const float *__restrict__ X,
                                                       float *__restrict__ Y)
{
    constexpr int LD = MB * 32;
    __shared__ __attribute__((aligned(16))) float xs[2][32 * LD];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int half = lane >> 5, l31 = lane & 31;
    const unsigned row = blockIdx.x * 128 + wave * 32 + l31;
    const unsigned rowc = row < p ? row : p - 1; // clamp loads, skip the store
    const float *arow = A + (size_t)rowc * lda + 16 * half;

    f32x16 acc[MB];
#pragma unroll
    for (int b = 0; b < MB; ++b)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[b][r] = 0.f;

    const int ntiles = p32 / 32;
    // stage X tile 0
    for (int e = threadIdx.x * 4; e < 32 * LD; e += 256 * 4)
        *reinterpret_cast<float4 *>(&xs[0][e]) = *reinterpret_cast<const float4 *>(&X[e]);
    float4 a_cur[4], a_nxt[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) a_cur[q] = *reinterpret_cast<const float4 *>(arow + 4 * q);
    __syncthreads();

    for (int t = 0; t < ntiles; ++t) {
        const int buf = t & 1;
        if (t + 1 < ntiles) {
            const float *xn = X + (size_t)(t + 1) * 32 * LD;
            for (int e = threadIdx.x * 4; e < 32 * LD; e += 256 * 4)
                *reinterpret_cast<float4 *>(&xs[buf ^ 1][e]) = *reinterpret_cast<const float4 *>(&xn[e]);
#pragma unroll
            for (int q = 0; q < 4; ++q)
                a_nxt[q] = *reinterpret_cast<const float4 *>(arow + (size_t)(t + 1) * 32 + 4 * q);
        }
        const float *xb = &xs[buf][(16 * half) * LD + l31];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const float av[4] = {a_cur[q].x, a_cur[q].y, a_cur[q].z, a_cur[q].w};
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int kk = 4 * q + e;
#pragma unroll
                for (int b = 0; b < MB; ++b)
                    acc[b] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[e], xb[kk * LD + 32 * b], acc[b], 0, 0, 0);
            }
        }
        __syncthreads(); // xs[buf] consumed by all waves, xs[buf^1] fully written
#pragma unroll
        for (int q = 0; q < 4; ++q) a_cur[q] = a_nxt[q];
    }
    // C/D layout of 32x32 MFMA: col = lane & 31, row = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5)
    const unsigned rbase = blockIdx.x * 128 + wave * 32;
#pragma unroll
    for (int b = 0; b < MB; ++b)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const unsigned orow = rbase + (r & 3) + 8 * (r >> 2) + 4 * half;
            if (orow < p) Y[(size_t)orow * LD + 32 * b + l31] = acc[b][r];
        }
}


// =====================================================================================
// Y = A X on the f16 matrix pipe with split operands (same scheme as k_nystroem_f16s):
//   A' = 2^10 A = Ahi + Alo (split on the fly from the f32 stream),  X' = T_j X = Xhi + Xlo
//   (fragment-ordered, prepared once per mat-vec), three v_mfma_f32_32x32x16_f16 products,
//   f32 accumulation, exact power-of-two rescale in the epilogue.
// The f32-input MFMA above needs 6 ms of the shared f32 FMA pipe per 29 GB sweep; this one needs
// about 1 ms of the f16 pipe, which leaves the kernel bound by the HBM stream of L_A.
// Lane (r = l & 31, h = l >> 5) loads the 32 consecutive floats A[row r][k0 + 32 h ...) of every
// 64-wide K tile (one 128-B line); MFMA step t of the tile contracts k = k0 + 32 h + 8 t + j.
// =====================================================================================

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
constexpr int MV_ASCALE_LOG2 = 10;
constexpr int MV_MAXTILES = 4096; // K tiles a workgroup can list (p <= 262 144)

__global__ void k_mv_col_absmax(const float *__restrict__ X, unsigned p, unsigned ld, float *__restrict__ out)
{
    __shared__ float sh[256];
    const unsigned c = blockIdx.x;
    float m = 0.f;
    for (unsigned i = threadIdx.x; i < p; i += 256) m = fmaxf(m, fabsf(X[(size_t)i * ld + c]));
    sh[threadIdx.x] = m;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if ((int)threadIdx.x < o) sh[threadIdx.x] = fmaxf(sh[threadIdx.x], sh[threadIdx.x + o]);
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        // power-of-two T with max * T in [2^13, 2^14); 1 for an all-zero / non-finite column
        const float mx = sh[0];
        int ex = 0;
        if (mx > 0.f && mx < 3.0e38f) ex = 13 - ilogbf(mx);
        ex = max(-100, min(100, ex));
        out[c] = ldexpf(1.0f, ex);                           // scale
        out[ld + c] = ldexpf(1.0f, -ex - MV_ASCALE_LOG2);    // inverse of (scale * 2^10)
    }
}

// X f32 [p_pad][ld] -> fragments [ktile][t (4)][jb][q (2)][lane (64)][8 halves]
__global__ __launch_bounds__(256) void k_mv_x_split(const float *__restrict__ X, unsigned p_pad, unsigned ld,
                                                     const float *__restrict__ scales, _Float16 *__restrict__ out)
{
    const size_t e = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (e >= (size_t)p_pad * ld) return;
    const unsigned s = (unsigned)(e / ld), c = (unsigned)(e % ld);
    const float v = X[e] * scales[c];
    const _Float16 hi = (_Float16)v;
    const _Float16 lo = (_Float16)(v - (float)hi);
    const unsigned mb = ld / 32;
    const unsigned kt = s / 64, h = (s % 64) / 32, t = (s % 32) / 8, j = s % 8, jb = c / 32, r = c % 32;
    const size_t frag = (((size_t)kt * 4 + t) * mb + jb) * 2;
    const size_t lane = h * 32 + r;
    out[((frag + 0) * 64 + lane) * 8 + j] = hi;
    out[((frag + 1) * 64 + lane) * 8 + j] = lo;
}

// A is SYMMETRIC (L_A; the reference requires it too: "A is symmetric (and square)",
// hpc/inverse_power_it.c:82-85), so the A-fragment element A[row r][k] is read as A[k][row r]:
// for a fixed k the 32 lanes of a wave half read 32 consecutive floats of row k -- every load
// instruction is two fully used 128-B lines, and the 4 waves of a workgroup cover 512 contiguous
// bytes of that row. (Reading A[row r][k..k+7] directly makes each lane walk its own row: 64
// different lines per instruction, 16 B used of each, the rest re-fetched through a thrashing
// L1 -- measured 2.6 TB/s.)
template <int MB>
__global__ __launch_bounds__(256) void k_block_matvec_f16s(const float *__restrict__ A, int64_t lda, unsigned p,
                                                            unsigned p_pad, unsigned row_begin, unsigned row_end,
                                                            const _Float16 *__restrict__ xfrag,
                                                            const float *__restrict__ scales, float *__restrict__ Y,
                                                            float *__restrict__ Ypart, const int4 *__restrict__ kbox,
                                                            int radius, int nslabs)
{
    constexpr int LD = MB * 32;
    constexpr int FR_F4 = 4 * MB * 2 * 64; // float4 words of one K tile's X fragments
    constexpr int PIECES = FR_F4 * 16 / 1024;
    __shared__ __attribute__((aligned(16))) float4 lds[2 * FR_F4];
    __shared__ int tlist[MV_MAXTILES];
    __shared__ int tscratch[257];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int half = lane >> 5, l31 = lane & 31;
    const unsigned rbase = row_begin + blockIdx.x * 128 + wave * 32;
    const unsigned row = rbase + l31;
    const unsigned rowc = row < row_end ? row : row_end - 1; // clamp loads, skip the store
    const float *acol = A + rowc + (size_t)(32 * half) * lda; // element (k = 32 half + i, rowc) at acol[i * lda]
    const float4 *gfr = reinterpret_cast<const float4 *>(xfrag);

    f32x16 acc[MB];
#pragma unroll
    for (int b = 0; b < MB; ++b)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[b][r] = 0.f;

    // K split: blockIdx.y sweeps tiles [kt_begin, kt_end); with gridDim.y > 1 the partial sums go to
    // slab blockIdx.y of Ypart and k_mv_sum_splits adds them in fixed order.
    const int ntiles = p_pad / 64;
    // tiles this workgroup's 128 rows actually need (all of them when radius < 0), in ascending order
    int nlist = ntiles;
    if (radius >= 0) {
        const unsigned blk0 = row_begin + blockIdx.x * 128;       // multiple of 64
        const unsigned blk1 = min(blk0 + 128, row_end);
        int4 rb = kbox[blk0 / 64];
        if (blk0 + 64 < blk1) {
            const int4 b2 = kbox[blk0 / 64 + 1];
            rb = make_int4(min(rb.x, b2.x), max(rb.y, b2.y), min(rb.z, b2.z), max(rb.w, b2.w));
        }
        const int t = threadIdx.x, per = (ntiles + 255) / 256;
        const int nvalid = (int)((p + 63) / 64); // tiles past the last sample hold nothing
        unsigned bits = 0;
        int count = 0;
        for (int i = 0; i < per; ++i) {
            const int kt = t * per + i;
            if (kt < nvalid) {
                const int4 b = kbox[kt];
                const bool rel = b.y >= rb.x - radius && b.x <= rb.y + radius && b.w >= rb.z - radius && b.z <= rb.w + radius;
                bits |= (unsigned)rel << i;
                count += rel;
            }
        }
        tscratch[t] = count;
        __syncthreads();
        int off = 0;
        for (int k = 0; k < t; ++k) off += tscratch[k];
        if (t == 255) tscratch[256] = off + count;
        for (int i = 0; i < per; ++i)
            if (bits & (1u << i)) tlist[off++] = t * per + i;
        __syncthreads();
        nlist = tscratch[256];
    }
    auto tile_at = [&](int i) { return radius < 0 ? i : tlist[i]; };
    float a_cur[32], a_nxt[32];
    auto load_a = [&](int kt, float (&dst)[32]) {
        const unsigned k0 = (unsigned)kt * 64 + 32 * half;
        if (kt + 1 < ntiles) {
#pragma unroll
            for (int i = 0; i < 32; ++i) dst[i] = acol[(size_t)kt * 64 * lda + (size_t)i * lda];
        } else { // last tile: rows k >= p do not exist (their X rows are zero): clamp the address
#pragma unroll
            for (int i = 0; i < 32; ++i) {
                const unsigned k = min(k0 + i, p - 1);
                dst[i] = A[(size_t)k * lda + rowc];
            }
        }
    };
    const float ascale = (float)(1 << MV_ASCALE_LOG2);
    // acc += sum over list entries [lb, le): prologue + double-buffered sweep
    auto sweep = [&](int lb, int le) {
        __syncthreads(); // LDS buffers of a previous sweep are free
        if (lb < le) {
            const int kt0 = tile_at(lb);
            load_a(kt0, a_cur);
            lds_dma_copy(gfr + (size_t)kt0 * FR_F4, lds, PIECES, wave, lane);
        }
        lds_dma_drain();
        __syncthreads();
        for (int li = lb; li < le; ++li) {
            const int buf = (li - lb) & 1;
            if (li + 1 < le) { // next A tile into registers, next X fragments straight into the other LDS buffer
                const int ktn = tile_at(li + 1);
                load_a(ktn, a_nxt);
                lds_dma_copy(gfr + (size_t)ktn * FR_F4, lds + (buf ^ 1) * FR_F4, PIECES, wave, lane);
            }
            const f16x8 *fr = reinterpret_cast<const f16x8 *>(lds + buf * FR_F4);
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                f16x8 ah, al;
#pragma unroll
                for (int e = 0; e < 8; e += 2) {
                    f32x2 y = {a_cur[8 * t + e] * ascale, a_cur[8 * t + e + 1] * ascale};
                    const f16x2 h2 = __builtin_convertvector(y, f16x2);
                    const f32x2 res = y - __builtin_convertvector(h2, f32x2);
                    const f16x2 l2 = __builtin_convertvector(res, f16x2);
                    ah[e] = h2[0];
                    ah[e + 1] = h2[1];
                    al[e] = l2[0];
                    al[e + 1] = l2[1];
                }
#pragma unroll
                for (int b = 0; b < MB; ++b) {
                    const f16x8 bh = fr[((t * MB + b) * 2 + 0) * 64 + lane];
                    const f16x8 bl = fr[((t * MB + b) * 2 + 1) * 64 + lane];
                    acc[b] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh, acc[b], 0, 0, 0);
                    acc[b] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bl, acc[b], 0, 0, 0);
                    acc[b] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, bh, acc[b], 0, 0, 0);
                }
            }
            if (li + 1 < le) {
#pragma unroll
                for (int i = 0; i < 32; ++i) a_cur[i] = a_nxt[i];
            }
            lds_dma_drain();
            __syncthreads();
        }
    };
    if (radius < 0) {
        // dense: blockIdx.y sweeps tiles [ntiles y / Y, ntiles (y+1) / Y); with gridDim.y > 1 the partial
        // sums go to slab blockIdx.y of Ypart and k_mv_sum_splits adds the slabs in order
        sweep((int)((int64_t)ntiles * blockIdx.y / gridDim.y), (int)((int64_t)ntiles * (blockIdx.y + 1) / gridDim.y));
    } else {
        // skipping: one workgroup walks the slabs of the dense run one after the other and adds the
        // slab sums in the same order -- the skipped tiles contribute exact zeros, so Y is bit-identical
        f32x16 run[MB];
#pragma unroll
        for (int b = 0; b < MB; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) run[b][r] = 0.f;
        int lpos = 0;
        for (int sl = 0; sl < nslabs; ++sl) {
            const int t_end = (int)((int64_t)ntiles * (sl + 1) / nslabs);
            int le = lpos;
            while (le < nlist && tlist[le] < t_end) ++le;
#pragma unroll
            for (int b = 0; b < MB; ++b)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[b][r] = 0.f;
            sweep(lpos, le);
#pragma unroll
            for (int b = 0; b < MB; ++b)
#pragma unroll
                for (int r = 0; r < 16; ++r) run[b][r] += acc[b][r];
            lpos = le;
        }
#pragma unroll
        for (int b = 0; b < MB; ++b) acc[b] = run[b];
    }
    float *dst = gridDim.y > 1 ? Ypart + (size_t)blockIdx.y * p_pad * LD : Y;
#pragma unroll
    for (int b = 0; b < MB; ++b) {
        const float inv = gridDim.y > 1 ? 1.0f : scales[LD + 32 * b + l31];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const unsigned orow = rbase + (r & 3) + 8 * (r >> 2) + 4 * half;
            if (orow < row_end) dst[(size_t)orow * LD + 32 * b + l31] = acc[b][r] * inv;
        }
    }
}

// Y[i][c] = inv[c] * sum_s Ypart[s][i][c]  (fixed order)
__global__ __launch_bounds__(256) void k_mv_sum_splits(const float *__restrict__ Ypart, int nsplit, unsigned p_pad,
                                                        unsigned row_begin, unsigned row_end, unsigned ld,
                                                        const float *__restrict__ scales, float *__restrict__ Y)
{
    const size_t e = (size_t)row_begin * ld + (size_t)blockIdx.x * 256 + threadIdx.x;
    if (e >= (size_t)row_end * ld) return;
    float s = 0.f;
    for (int k = 0; k < nsplit; ++k) s += Ypart[(size_t)k * p_pad * ld + e];
    Y[e] = s * scales[ld + (unsigned)(e % ld)];
}

static int block_matvec_f16s(glf_ctx *ctx, const float *A, int64_t lda, unsigned p, const float *X, float *Y, unsigned ld,
                             const MatShard *shard)
{
    const unsigned p_pad = (unsigned)round_up(p, VEC_PAD);
    const bool sharded = shard && shard->rows_per_rank;
    const unsigned row0 = sharded ? shard->row0 : 0u, row1 = sharded ? shard->row1 : p;
    // element (k, row) of the symmetric operator sits at A[k * lda + (row - row0)]
    const float *A_eff = A - row0;
    const int4 *kbox = shard ? shard->kbox : nullptr;
    const int radius = (kbox && shard->radius >= 0 && p_pad / 64 <= (unsigned)MV_MAXTILES) ? shard->radius : -1;
    hipStream_t st = ctx->stream;
    // scratch lives in the context (reused by every mat-vec of a solve; no allocation in the loop)
    // K split so that the grid fills the 2-workgroups-per-CU residency evenly (the kernel keeps
    // 64 VGPRs of A tile in flight: 2 waves per SIMD)
    const int nrb = (int)ceil_div(row1 > row0 ? row1 - row0 : 1u, 128);
    const int slots = 2 * ctx->prop.multiProcessorCount;
    int ksplit = 1;
    {
        double best = 0.0;
        const int ntiles = (int)(p_pad / 64);
        for (int ks = 1; ks <= 8 && ks <= ntiles; ++ks) {
            const double wgs = (double)nrb * ks;
            const double eff = wgs / (std::ceil(wgs / slots) * slots);
            if (eff > best + 0.02) { best = eff; ksplit = ks; }
        }
    }
    const size_t part_bytes = ksplit > 1 ? (size_t)ksplit * p_pad * ld * sizeof(float) : 0;
    const size_t frag_bytes = (size_t)p_pad * ld * 2 * sizeof(_Float16);
    const size_t need = 256 + 2 * ld * sizeof(float) + frag_bytes + part_bytes;
    if (ctx->mv_scratch_bytes < need) {
        if (ctx->mv_scratch) (void)hipFree(ctx->mv_scratch);
        ctx->mv_scratch = nullptr;
        ctx->mv_scratch_bytes = 0;
        GLF_HIP(ctx, hipMalloc(&ctx->mv_scratch, need));
        ctx->mv_scratch_bytes = need;
    }
    float *scales = reinterpret_cast<float *>(ctx->mv_scratch);
    char *base = reinterpret_cast<char *>(ctx->mv_scratch) + round_up(2 * ld * sizeof(float), 256);
    _Float16 *xfrag = reinterpret_cast<_Float16 *>(base);
    float *ypart = ksplit > 1 ? reinterpret_cast<float *>(base + frag_bytes) : nullptr;
    hipLaunchKernelGGL(k_mv_col_absmax, dim3(ld), dim3(256), 0, st, X, p, ld, scales);
    hipLaunchKernelGGL(k_mv_x_split, dim3((unsigned)ceil_div((int64_t)p_pad * ld, 256)), dim3(256), 0, st, X, p_pad, ld, scales,
                       xfrag);
    if (row1 > row0) {
        // with tile skipping one workgroup per row block replays the dense run's slabs itself (bit-identical Y)
        const bool skipping = radius >= 0;
        dim3 grid((unsigned)nrb, skipping ? 1u : (unsigned)ksplit), block(256);
        if (ctx->mv_pending == glf_ctx::MV_RING) GLF_TRY(mv_collect(ctx));
        GLF_HIP(ctx, hipEventRecord(ctx->mv_ev[0][ctx->mv_pending], st));
        switch (ld / 32) {
        case 1: hipLaunchKernelGGL(k_block_matvec_f16s<1>, grid, block, 0, st, A_eff, lda, p, p_pad, row0, row1, xfrag, scales, Y, ypart, kbox, radius, ksplit); break;
        case 2: hipLaunchKernelGGL(k_block_matvec_f16s<2>, grid, block, 0, st, A_eff, lda, p, p_pad, row0, row1, xfrag, scales, Y, ypart, kbox, radius, ksplit); break;
        case 4: hipLaunchKernelGGL(k_block_matvec_f16s<4>, grid, block, 0, st, A_eff, lda, p, p_pad, row0, row1, xfrag, scales, Y, ypart, kbox, radius, ksplit); break;
        case 8: hipLaunchKernelGGL(k_block_matvec_f16s<8>, grid, block, 0, st, A_eff, lda, p, p_pad, row0, row1, xfrag, scales, Y, ypart, kbox, radius, ksplit); break;
        default: return set_error(ctx, GLF_ERR_UNSUPPORTED, "ld %u", ld);
        }
        GLF_HIP(ctx, hipEventRecord(ctx->mv_ev[1][ctx->mv_pending], st));
        ++ctx->mv_pending;
        ctx->mv_bytes += 4.0 * (double)p * (double)(row1 - row0); // the L_A block this rank streams (algorithmic bytes)
        if (ksplit > 1 && !skipping)
            hipLaunchKernelGGL(k_mv_sum_splits, dim3((unsigned)ceil_div((int64_t)(row1 - row0) * ld, 256)), dim3(256), 0, st, ypart,
                               ksplit, p_pad, row0, row1, ld, scales, Y);
    }
    GLF_LAUNCH_CHECK(ctx);
    return GLF_OK; // sharded: rows [row0, row1) of Y only -- the vector kernels that follow are row-sharded too
}

int start_block_cached(glf_ctx *ctx, unsigned p, unsigned m, unsigned ld, unsigned long long seed, const float **d_block)
{
    if (!ctx->x0_block || ctx->x0_p != p || ctx->x0_m != m || ctx->x0_ld != ld || ctx->x0_seed != seed) {
        const size_t n = (size_t)round_up(p, VEC_PAD) * ld;
        std::vector<double> x0((size_t)m * p);
        glf_random_vectors(x0.data(), p, m, seed);
        std::vector<float> h(n, 0.f);
        for (unsigned j = 0; j < m; ++j)
            for (unsigned i = 0; i < p; ++i) h[(size_t)i * ld + j] = (float)x0[(size_t)j * p + i];
        if (ctx->x0_block) (void)hipFree(ctx->x0_block);
        ctx->x0_block = nullptr;
        GLF_HIP(ctx, hipMalloc(reinterpret_cast<void **>(&ctx->x0_block), n * sizeof(float)));
        GLF_HIP(ctx, hipMemcpyAsync(ctx->x0_block, h.data(), n * sizeof(float), hipMemcpyHostToDevice, ctx->stream));
        GLF_HIP(ctx, hipStreamSynchronize(ctx->stream));
        ctx->x0_p = p;
        ctx->x0_m = m;
        ctx->x0_ld = ld;
        ctx->x0_seed = seed;
    }
    *d_block = ctx->x0_block;
    return GLF_OK;
}

int mv_collect(glf_ctx *ctx)
{
    if (ctx->mv_pending == 0) return GLF_OK;
    GLF_HIP(ctx, hipEventSynchronize(ctx->mv_ev[1][ctx->mv_pending - 1]));
    for (int i = 0; i < ctx->mv_pending; ++i) {
        float ms = 0.f;
        GLF_HIP(ctx, hipEventElapsedTime(&ms, ctx->mv_ev[0][i], ctx->mv_ev[1][i]));
        ctx->mv_ms += ms;
    }
    ctx->mv_count += ctx->mv_pending;
    ctx->mv_pending = 0;
    return GLF_OK;
}

// ---- row sharding of the vector blocks (multi-GPU eigen-solve) ------------------------------------------------------
// With a sharded operator (MatShard::rows_per_rank > 0) rank g owns rows [row0, row1) of EVERY vector block: the PCG
// updates, Gram-Schmidt, the residual and the final normalisation touch those rows only, and what crosses xGMI is
//   * one all-reduce of ld (or 2 ld) f64 scalars per inner product / norm (hpc/gram_schmidt.c:14-15,59: VecDot, VecNorm),
//   * one all-reduce of the ld x ld Gram block per Gram-Schmidt / residual (hpc/inverse_power_it.c:55-72),
//   * one all-gather of the operand block before each operator application (what PETSc's MPIDENSE MatMult does,
//     hpc/inverse_power_it.c:167): the operator needs every row of its operand, nothing else does.
// The blocks keep their full size on every rank (rows_per_rank * size rows) so that the all-gather runs in place.
struct Rows {
    unsigned r0 = 0, r1 = 0; // this rank's rows
    bool dist = false;       // sums need an all-reduce, operator operands an all-gather
    unsigned n() const { return r1 - r0; }
};
static inline Rows rows_of(unsigned p, const MatShard *sh)
{
    Rows r;
    if (sh && sh->rows_per_rank) {
        r.r0 = sh->row0;
        r.r1 = sh->row1;
        r.dist = true;
    } else {
        r.r1 = p;
    }
    return r;
}
static int allreduce_d(glf_ctx *ctx, double *d, size_t n)
{
    if (!ctx->has_comm || ctx->comm.allreduce_sum_f64(ctx->comm.user, d, n) != 0)
        return set_error(ctx, GLF_ERR_COMM, "allreduce_sum_f64 callback failed");
    return GLF_OK;
}
static int allreduce_f(glf_ctx *ctx, float *d, size_t n)
{
    if (!ctx->has_comm || ctx->comm.allreduce_sum_f32(ctx->comm.user, d, n) != 0)
        return set_error(ctx, GLF_ERR_COMM, "allreduce_sum_f32 callback failed");
    return GLF_OK;
}
// every rank's row block of V -> all ranks (in place; V holds rows_per_rank * size rows)
static int allgather_rows(glf_ctx *ctx, float *V, const MatShard *sh, unsigned ld)
{
    if (!ctx->has_comm || !ctx->comm.allgather_f32) return set_error(ctx, GLF_ERR_COMM, "sharded eigen-solve without allgather_f32");
    if (ctx->comm.allgather_f32(ctx->comm.user, V, (size_t)sh->rows_per_rank * ld) != 0)
        return set_error(ctx, GLF_ERR_COMM, "allgather_f32 callback failed");
    return GLF_OK;
}

// Y[rows of this rank] = A X; X must hold all p rows (sharded callers all-gather it first)
int block_matvec(glf_ctx *ctx, const float *A, int64_t lda, unsigned p, const float *X, float *Y, unsigned ld,
                 const MatShard *shard)
{
    const unsigned p32 = (unsigned)round_up(p, VEC_PAD);
    const bool sharded = shard && shard->rows_per_rank;
    if (shard && shard->grid) { // L_A applied in grid-factored form (never stored): rows [row0, row1) are whole grid rows
        const unsigned row0 = sharded ? shard->row0 : 0u, row1 = sharded ? shard->row1 : p;
        if (ctx->mv_pending == glf_ctx::MV_RING) GLF_TRY(mv_collect(ctx));
        GLF_HIP(ctx, hipEventRecord(ctx->mv_ev[0][ctx->mv_pending], ctx->stream));
        GLF_TRY(grid_op_apply(ctx, shard->grid, X, Y, ld, shard->grid_alpha, shard->grid_degree, row0, row1, shard->grid_window));
        GLF_HIP(ctx, hipEventRecord(ctx->mv_ev[1][ctx->mv_pending], ctx->stream));
        ++ctx->mv_pending;
        return GLF_OK;
    }
    if (sharded && ctx->contraction != GLF_CONTRACT_F16_SPLIT)
        return set_error(ctx, GLF_ERR_UNSUPPORTED, "row-sharded mat-vec needs the split-f16 contraction");
    if ((!sharded && lda < (int64_t)p32) || (lda & 3) || !valid_ld(ld))
        return set_error(ctx, GLF_ERR_INVALID, "block_matvec: lda=%lld ld=%u (need lda >= round_up(p,64), lda%%4==0, ld%%32==0, ld<=256)",
                         (long long)lda, ld);
    if (reinterpret_cast<uintptr_t>(A) & 15)
        return set_error(ctx, GLF_ERR_INVALID, "block_matvec: A must be 16-byte aligned");
    if (ctx->contraction == GLF_CONTRACT_F16_SPLIT) return block_matvec_f16s(ctx, A, lda, p, X, Y, ld, shard);
    dim3 grid((unsigned)ceil_div(p, 128)), block(256);
    switch (ld / 32) {
    case 1: hipLaunchKernelGGL(k_block_matvec<1>, grid, block, 0, ctx->stream, A, lda, p, p32, X, Y); break;
    case 2: hipLaunchKernelGGL(k_block_matvec<2>, grid, block, 0, ctx->stream, A, lda, p, p32, X, Y); break;
    case 4: hipLaunchKernelGGL(k_block_matvec<4>, grid, block, 0, ctx->stream, A, lda, p, p32, X, Y); break;
    case 8: hipLaunchKernelGGL(k_block_matvec<8>, grid, block, 0, ctx->stream, A, lda, p, p32, X, Y); break;
    default: return set_error(ctx, GLF_ERR_UNSUPPORTED, "ld %u (must be 32, 64, 128 or 256)", ld);
    }
    GLF_LAUNCH_CHECK(ctx);
    return GLF_OK;
}

// =====================================================================================
// Small helpers: fixed-order sums of per-block partials
// =====================================================================================

constexpr int RED_ROWS = 128; // rows per reduction workgroup (667 workgroups at p = 85 264: every CU busy)

// out[c] = sum_blk partial[blk][c], c < ncols
// Second level of the column reductions, one 256-thread workgroup: out[v] (valid for t < ld) = sum over the nblk
// blocks of partial[(b * NV + v) * ld + t % ld]. The 256 / ld thread rows take interleaved blocks and are combined
// through LDS -- all in a fixed order.
// NTHR: threads of the workgroup (256, or 1024 where the reduction sits between two sweeps of the solver: four times the
// thread rows, a quarter of the dependent load batches).
template <int NV, int NTHR = 256>
__device__ __forceinline__ void wg_sum_partials(const double *__restrict__ partial, int nblk, unsigned ld, double (&out)[NV],
                                                double *sh /* [NV][NTHR] */)
{
    const int t = threadIdx.x, col = t % ld, part = t / ld, nparts = NTHR / ld;
    double acc[NV];
#pragma unroll
    for (int v = 0; v < NV; ++v) acc[v] = 0.0;
    // eight blocks' loads in flight at a time, added in the same order as one by one (667 dependent round trips to L2
    // made each of these single-workgroup reductions 43 us at p = 85 264)
    int b = part;
    for (; b + 7 * nparts < nblk; b += 8 * nparts) {
        double x[8][NV];
#pragma unroll
        for (int u = 0; u < 8; ++u)
#pragma unroll
            for (int v = 0; v < NV; ++v) x[u][v] = partial[((size_t)(b + u * nparts) * NV + v) * ld + col];
#pragma unroll
        for (int u = 0; u < 8; ++u)
#pragma unroll
            for (int v = 0; v < NV; ++v) acc[v] += x[u][v];
    }
    for (; b < nblk; b += nparts)
#pragma unroll
        for (int v = 0; v < NV; ++v) acc[v] += partial[((size_t)b * NV + v) * ld + col];
#pragma unroll
    for (int v = 0; v < NV; ++v) sh[v * NTHR + t] = acc[v];
    __syncthreads();
#pragma unroll
    for (int v = 0; v < NV; ++v) {
        double s = 0.0;
        if (t < (int)ld)
            for (int r = 0; r < nparts; ++r) s += sh[v * NTHR + r * ld + col];
        out[v] = s;
    }
    __syncthreads();
}

// out[c] = sum_blk partial[blk][c], c < ld (one workgroup of 256 threads)
__global__ __launch_bounds__(256) void k_sum_partials(const double *__restrict__ partial, int nblk, int ncols,
                                                       double *__restrict__ out)
{
    __shared__ double sh[256];
    double s[1];
    wg_sum_partials<1>(partial, nblk, (unsigned)ncols, s, sh);
    if ((int)threadIdx.x < ncols) out[threadIdx.x] = s[0];
}

// sums[v * ld + c] = sum_blk partial[(blk * NV + v) * ld + c]: the local sums of a row-sharded reduction, all-reduced by the
// host before the second-level kernel consumes them as ONE block (nblk = 1)
template <int NV>
__global__ __launch_bounds__(256) void k_sum_partials_nv(const double *__restrict__ partial, int nblk, unsigned ld,
                                                          double *__restrict__ sums)
{
    __shared__ double sh[NV * 256];
    double s[NV];
    wg_sum_partials<NV>(partial, nblk, ld, s, sh);
    if (threadIdx.x < ld)
#pragma unroll
        for (int v = 0; v < NV; ++v) sums[v * ld + threadIdx.x] = s[v];
}

// Block-level column reduction helper: 256 threads, column = t % ld, row lane = t / ld.
// val[v] are this thread's partial sums; result (summed over the row lanes) is written
// to partial[(blockIdx.x * NV + v) * ld + col].
template <int NV>
__device__ __forceinline__ void block_col_reduce(double (&val)[NV], unsigned ld, double *__restrict__ partial,
                                                 double *sh /* [NV][256] */)
{
    const int t = threadIdx.x;
    const int col = t % ld, nrl = 256 / ld;
#pragma unroll
    for (int v = 0; v < NV; ++v) sh[v * 256 + t] = val[v];
    __syncthreads();
    if (t < (int)ld) {
#pragma unroll
        for (int v = 0; v < NV; ++v) {
            double s = 0.0;
            for (int r = 0; r < nrl; ++r) s += sh[v * 256 + r * ld + col];
            partial[((size_t)blockIdx.x * NV + v) * ld + col] = s;
        }
    }
}

// =====================================================================================
// Jacobi-PCG on all columns at once (stands in for KSPSolve x m, hpc/inverse_power_it.c:165-168)
// =====================================================================================

struct CgScalars {    // device, per column (ld entries each)
    double *rz;       // r . z
    double *bn2;      // ||b||^2
    double *alpha;
    double *beta;
    int *active;      // 1 while the column still iterates
    int *nactive;     // single int
};

// init: R = B (B lives in XB), X = 0 (XB is overwritten at the end), P = dinv .* R
__global__ __launch_bounds__(256) void k_cg_init(const float *__restrict__ B, const float *__restrict__ dinv,
                                                  float *__restrict__ R, float *__restrict__ P,
                                                  float *__restrict__ Xs, unsigned p, unsigned ld,
                                                  double *__restrict__ partial)
{
    __shared__ double sh[2 * 256];
    const int col = threadIdx.x % ld, rl = threadIdx.x / ld, nrl = 256 / ld;
    double v[2] = {0.0, 0.0};
    const unsigned base = blockIdx.x * RED_ROWS;
    for (unsigned r = rl; r < RED_ROWS; r += nrl) {
        const unsigned i = base + r;
        if (i >= p) break;
        const size_t o = (size_t)i * ld + col;
        const float b = B[o], d = dinv[i];
        const float z = d * b;
        R[o] = b;
        P[o] = z;
        Xs[o] = 0.f;
        v[0] += (double)b * (double)z; // r.z
        v[1] += (double)b * (double)b; // ||b||^2
    }
    block_col_reduce<2>(v, ld, partial, sh);
}

__global__ __launch_bounds__(256) void k_cg_init_scalars(const double *__restrict__ partial, int nblk, unsigned ld, unsigned m,
                                                          CgScalars s)
{
    __shared__ double sh[2 * 256];
    const int c = threadIdx.x;
    double tot[2];
    wg_sum_partials<2>(partial, nblk, ld, tot, sh);
    if (c < (int)ld) {
        const double rz = tot[0], bn2 = tot[1];
        s.rz[c] = rz;
        s.bn2[c] = bn2;
        s.active[c] = (c < (int)m && bn2 > 0.0) ? 1 : 0;
        s.alpha[c] = 0.0;
        s.beta[c] = 0.0;
    }
    __syncthreads();
    if (c == 0) {
        int n = 0;
        for (unsigned k = 0; k < ld; ++k) n += s.active[k];
        *s.nactive = n;
    }
}

// partial p.Ap
__global__ __launch_bounds__(256) void k_cg_dot(const float *__restrict__ P, const float *__restrict__ AP, unsigned p,
                                                 unsigned ld, double *__restrict__ partial)
{
    __shared__ double sh[256];
    const int col = threadIdx.x % ld, rl = threadIdx.x / ld, nrl = 256 / ld;
    double v[1] = {0.0};
    const unsigned base = blockIdx.x * RED_ROWS;
    for (unsigned r = rl; r < RED_ROWS; r += nrl) {
        const unsigned i = base + r;
        if (i >= p) break;
        const size_t o = (size_t)i * ld + col;
        v[0] += (double)P[o] * (double)AP[o];
    }
    block_col_reduce<1>(v, ld, partial, sh);
}

__global__ __launch_bounds__(1024) void k_cg_alpha(const double *__restrict__ partial, int nblk, unsigned ld, CgScalars s)
{
    __shared__ double sh[1024];
    const int c = threadIdx.x;
    double pap[1];
    wg_sum_partials<1, 1024>(partial, nblk, ld, pap, sh);
    if (c < (int)ld) s.alpha[c] = s.active[c] ? s.rz[c] / pap[0] : 0.0;
}

// x += alpha p ; r -= alpha Ap ; partial ||r||^2 and r.(dinv r)
__global__ __launch_bounds__(256) void k_cg_update(float *__restrict__ Xs, float *__restrict__ R,
                                                    const float *__restrict__ P, const float *__restrict__ AP,
                                                    const float *__restrict__ dinv, unsigned p, unsigned ld,
                                                    CgScalars s, double *__restrict__ partial)
{
    __shared__ double sh[2 * 256];
    const int col = threadIdx.x % ld, rl = threadIdx.x / ld, nrl = 256 / ld;
    const bool act = s.active[col] != 0;
    const float alpha = (float)s.alpha[col];
    double v[2] = {0.0, 0.0};
    const unsigned base = blockIdx.x * RED_ROWS;
    if (act) {
        for (unsigned r = rl; r < RED_ROWS; r += nrl) {
            const unsigned i = base + r;
            if (i >= p) break;
            const size_t o = (size_t)i * ld + col;
            Xs[o] = fmaf(alpha, P[o], Xs[o]);
            const float rn = fmaf(-alpha, AP[o], R[o]);
            R[o] = rn;
            v[0] += (double)rn * (double)rn;
            v[1] += (double)rn * (double)(dinv[i] * rn);
        }
    }
    block_col_reduce<2>(v, ld, partial, sh);
}

__global__ __launch_bounds__(1024) void k_cg_beta(const double *__restrict__ partial, int nblk, unsigned ld, double rtol2,
                                                   CgScalars s)
{
    __shared__ double sh[2 * 1024];
    const int c = threadIdx.x;
    double tot[2];
    wg_sum_partials<2, 1024>(partial, nblk, ld, tot, sh);
    int still = 0; // this thread's column keeps iterating
    if (c < (int)ld && s.active[c]) {
        const double rr = tot[0], rz = tot[1];
        if (rr <= rtol2 * s.bn2[c]) { // ||r|| <= rtol ||b||
            s.active[c] = 0;
            s.beta[c] = 0.0;
        } else {
            s.beta[c] = rz / s.rz[c];
            s.rz[c] = rz;
            still = 1;
        }
    }
    const int n = __syncthreads_count(still); // (one thread re-reading the ld flags from memory took a third of the kernel)
    if (c == 0) *s.nactive = n;
}

// p = dinv r + beta p
__global__ __launch_bounds__(256) void k_cg_pupdate(float *__restrict__ P, const float *__restrict__ R,
                                                     const float *__restrict__ dinv, unsigned p, unsigned ld,
                                                     CgScalars s)
{
    const size_t e = (size_t)blockIdx.x * 256 + threadIdx.x;
    const unsigned i = (unsigned)(e / ld), col = (unsigned)(e % ld);
    if (i >= p || !s.active[col]) return;
    P[e] = fmaf((float)s.beta[col], P[e], dinv[i] * R[e]);
}

__global__ void k_diag_inv(const float *__restrict__ A, int64_t lda, unsigned p, float *__restrict__ dinv)
{
    const unsigned i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < p) dinv[i] = 1.0f / A[(size_t)i * lda + i];
}

// Narrow sweeps: once most columns of a block have converged, the operator is applied to the active ones only, packed to
// the left of a narrower block (the grid-factored operator costs in proportion to the block width, and the last step of
// nearly every solve iterates on a single straggler). Columns are independent in the operator -- scales, split and sums
// are per column -- so the packed application returns the same numbers.
__device__ __forceinline__ int cg_active_map(const int *__restrict__ active, unsigned ld, int *map /* LDS [256] */, int *count /* LDS */)
{
    const int t = threadIdx.x;
    if (t < (int)ld) {
        int rank = 0;
        for (int j = 0; j < t; ++j) rank += active[j] != 0;
        if (active[t]) map[rank] = t;
        if (t == (int)ld - 1) *count = rank + (active[t] != 0);
    }
    __syncthreads();
    return *count;
}

__global__ __launch_bounds__(256) void k_cg_pack(const float *__restrict__ P, const int *__restrict__ active, unsigned p, unsigned ld,
                                                  unsigned wn, float *__restrict__ Pc)
{
    __shared__ int map[256], count;
    const int nact = cg_active_map(active, ld, map, &count);
    const unsigned base = blockIdx.x * RED_ROWS;
    for (unsigned e = threadIdx.x; e < RED_ROWS * wn; e += 256) {
        const unsigned i = base + e / wn, r = e % wn;
        if (i >= p) break;
        Pc[(size_t)i * wn + r] = (int)r < nact ? P[(size_t)i * ld + map[r]] : 0.f;
    }
}

// rows [r0, r0 + n) of the packed result back into the active columns of AP
__global__ __launch_bounds__(256) void k_cg_unpack(const float *__restrict__ APc, const int *__restrict__ active, unsigned r0, unsigned n,
                                                    unsigned ld, unsigned wn, float *__restrict__ AP)
{
    __shared__ int map[256], count;
    const int nact = cg_active_map(active, ld, map, &count);
    const unsigned base = blockIdx.x * RED_ROWS;
    for (unsigned e = threadIdx.x; e < RED_ROWS * wn; e += 256) {
        const unsigned il = base + e / wn, r = e % wn;
        if (il >= n) break;
        if ((int)r < nact) AP[(size_t)(r0 + il) * ld + map[r]] = APc[(size_t)(r0 + il) * wn + r];
    }
}

struct CgWork {
    DevBuf<float> R, P, AP, Xs, dinv, Pc, APc;
    DevBuf<double> partial, scal, sums;
    DevBuf<int> flags;
    CgScalars s{};
    int nblk = 0;
    Rows rows;                // the rows of the vector blocks this rank updates (all of them unless the solve is sharded)
    int *h_nactive = nullptr; // in the context's pinned page
    const MatShard *shard = nullptr;
    int init(glf_ctx *ctx, unsigned p, unsigned ld, const MatShard *sh = nullptr)
    {
        shard = sh;
        rows = rows_of(p, sh);
        const size_t n = (size_t)vec_rows(p, sh, ctx->comm.size) * ld;
        GLF_TRY(R.alloc(ctx, n));
        GLF_TRY(P.alloc(ctx, n));
        GLF_TRY(AP.alloc(ctx, n));
        GLF_TRY(Xs.alloc(ctx, n));
        GLF_TRY(dinv.alloc(ctx, p));
        nblk = (int)ceil_div(rows.n(), RED_ROWS);
        GLF_TRY(partial.alloc(ctx, (size_t)std::max(1, nblk) * 2 * ld));
        GLF_TRY(sums.alloc(ctx, (size_t)2 * ld));
        GLF_TRY(scal.alloc(ctx, (size_t)4 * ld));
        GLF_TRY(flags.alloc(ctx, ld + 1));
        for (float *q : {R.p, P.p, AP.p, Xs.p}) GLF_HIP(ctx, hipMemsetAsync(q, 0, n * sizeof(float), ctx->stream));
        s.rz = scal.p;
        s.bn2 = scal.p + ld;
        s.alpha = scal.p + 2 * ld;
        s.beta = scal.p + 3 * ld;
        s.active = flags.p;
        // the convergence counter is written by the kernels straight into pinned host memory (device-visible, coherent):
        // a stream synchronise then suffices -- the 4-byte D2H copy was a copy-kernel launch of its own (~20 us per check)
        if (!ctx_pinned(ctx)) return set_error(ctx, GLF_ERR_NOMEM, "pinned host page");
        h_nactive = reinterpret_cast<int *>(ctx_pinned(ctx) + PINNED_NACTIVE);
        s.nactive = h_nactive;
        return GLF_OK;
    }
};

static int block_pcg_work(glf_ctx *ctx, CgWork &w, const float *A, int64_t lda, unsigned p, float *XB, unsigned m,
                          unsigned ld, double rtol, int max_it, int *iters)
{
    const int nblk = w.nblk;
    const Rows rows = w.rows;
    const unsigned nloc = rows.n();
    const size_t off = (size_t)rows.r0 * ld; // first element of this rank's rows in every vector block
    const unsigned nelem_blocks = (unsigned)ceil_div((int64_t)nloc * ld, 256);
    hipStream_t st = ctx->stream;
    // second-level reductions: the per-block partial sums of this rank, or (sharded) their all-reduced totals as one block
    auto reduce2 = [&](int nv, const double **src, int *src_blk) -> int {
        *src = w.partial.p;
        *src_blk = nblk;
        if (!rows.dist) return GLF_OK;
        if (nv == 1) hipLaunchKernelGGL(k_sum_partials_nv<1>, dim3(1), dim3(256), 0, st, w.partial.p, nblk, ld, w.sums.p);
        else hipLaunchKernelGGL(k_sum_partials_nv<2>, dim3(1), dim3(256), 0, st, w.partial.p, nblk, ld, w.sums.p);
        GLF_LAUNCH_CHECK(ctx);
        GLF_TRY(allreduce_d(ctx, w.sums.p, (size_t)nv * ld));
        *src = w.sums.p;
        *src_blk = 1;
        return GLF_OK;
    };
    const double *src = nullptr;
    int src_blk = 0;
    if (nblk > 0)
        hipLaunchKernelGGL(k_cg_init, dim3(nblk), dim3(256), 0, st, XB + off, w.dinv.p + rows.r0, w.R.p + off, w.P.p + off, w.Xs.p + off,
                           nloc, ld, w.partial.p);
    GLF_TRY(reduce2(2, &src, &src_blk));
    hipLaunchKernelGGL(k_cg_init_scalars, dim3(1), dim3(256), 0, st, src, src_blk, ld, m, w.s);
    GLF_LAUNCH_CHECK(ctx);
    int it = 0;
    GLF_HIP(ctx, hipStreamSynchronize(st));
    while (*(volatile int *)w.h_nactive > 0 && it < max_it) {
        ++it;
        if (rows.dist) GLF_TRY(allgather_rows(ctx, w.P.p, w.shard, ld)); // the operator needs every row of its operand
        unsigned wn = ld; // block width of this sweep
        if (w.shard && w.shard->grid && ld >= 64 && !ctx->tune.no_narrow) {
            const int na = *w.h_nactive; // (as of the synchronise that ended the step before)
            const unsigned cand = na <= 32 ? 32u : (unsigned)round_up(na, 64);
            if (cand < ld) wn = cand;
        }
        if (wn < ld) {
            if (!w.Pc.p) {
                const size_t n = (size_t)vec_rows(p, w.shard, ctx->comm.size) * ld;
                GLF_TRY(w.Pc.alloc(ctx, n));
                GLF_TRY(w.APc.alloc(ctx, n));
                GLF_HIP(ctx, hipMemsetAsync(w.Pc.p, 0, n * sizeof(float), st)); // (rows beyond p stay zero)
            }
            hipLaunchKernelGGL(k_cg_pack, dim3((unsigned)ceil_div(p, RED_ROWS)), dim3(256), 0, st, w.P.p, w.s.active, p, ld, wn, w.Pc.p);
            GLF_TRY(block_matvec(ctx, A, lda, p, w.Pc.p, w.APc.p, wn, w.shard));
            if (nblk > 0)
                hipLaunchKernelGGL(k_cg_unpack, dim3(nblk), dim3(256), 0, st, w.APc.p, w.s.active, rows.r0, nloc, ld, wn, w.AP.p);
            ++ctx->narrow_sweeps;
            if (ctx->tune.verbose) fprintf(stderr, "[glf] block PCG step %d: operator applied to %u packed columns\n", it, wn);
        } else {
            GLF_TRY(block_matvec(ctx, A, lda, p, w.P.p, w.AP.p, ld, w.shard));
        }
        if (nblk > 0) hipLaunchKernelGGL(k_cg_dot, dim3(nblk), dim3(256), 0, st, w.P.p + off, w.AP.p + off, nloc, ld, w.partial.p);
        GLF_TRY(reduce2(1, &src, &src_blk));
        hipLaunchKernelGGL(k_cg_alpha, dim3(1), dim3(1024), 0, st, src, src_blk, ld, w.s);
        if (nblk > 0)
            hipLaunchKernelGGL(k_cg_update, dim3(nblk), dim3(256), 0, st, w.Xs.p + off, w.R.p + off, w.P.p + off, w.AP.p + off,
                               w.dinv.p + rows.r0, nloc, ld, w.s, w.partial.p);
        GLF_TRY(reduce2(2, &src, &src_blk));
        hipLaunchKernelGGL(k_cg_beta, dim3(1), dim3(1024), 0, st, src, src_blk, ld, rtol * rtol, w.s);
        if (nelem_blocks > 0)
            hipLaunchKernelGGL(k_cg_pupdate, dim3(nelem_blocks), dim3(256), 0, st, w.P.p + off, w.R.p + off, w.dinv.p + rows.r0, nloc, ld, w.s);
        GLF_LAUNCH_CHECK(ctx);
        GLF_HIP(ctx, hipStreamSynchronize(st));
        if (ctx->tune.verbose) fprintf(stderr, "[glf] block PCG step %d: %d of %u columns active\n", it, *w.h_nactive, m);
    }
    if (nloc > 0) GLF_HIP(ctx, hipMemcpyAsync(XB + off, w.Xs.p + off, sizeof(float) * (size_t)nloc * ld, hipMemcpyDeviceToDevice, st));
    if (iters) *iters = it;
    return (*w.h_nactive > 0) ? set_error(ctx, GLF_ERR_NOCONV, "block PCG: %d columns unconverged after %d iterations",
                                          *w.h_nactive, it)
                              : GLF_OK;
}

// =====================================================================================
// Classical Gram-Schmidt (hpc/gram_schmidt.c:29-64), column k at a time
// =====================================================================================

// Three short launches per column. The projection proj_u(v) = <v,u>/<u,u> u (hpc/gram_schmidt.c:11-21)
// does not depend on the length of u, so the columns stay un-normalised during the sweep (q[j] = <u_j,u_j>
// is kept instead) and one final pass applies VecNormalize (:59) to all of them; norms[k] = sqrt(q[k])
// is the same pre-normalisation norm the reference returns.
constexpr int GS_ROWS = 128; // rows per Gram-Schmidt workgroup (667 workgroups at p = 85 264)

// partial[blk][j] = sum_{i in blk} X[i][k] X[i][j]   (j < k)
__global__ __launch_bounds__(256) void k_gs_dots(const float *__restrict__ X, unsigned n, unsigned ld, unsigned k,
                                                  double *__restrict__ partial)
{
    __shared__ double sh[256];
    const int col = threadIdx.x % ld, rl = threadIdx.x / ld, nrl = 256 / ld;
    double v = 0.0;
    const unsigned base = blockIdx.x * GS_ROWS;
    if (col < (int)k) {
#pragma unroll 4
        for (unsigned r = rl; r < GS_ROWS; r += nrl) {
            const unsigned i = base + r;
            if (i >= n) break;
            v += (double)X[(size_t)i * ld + k] * (double)X[(size_t)i * ld + col];
        }
    }
    sh[threadIdx.x] = v;
    __syncthreads();
    if (threadIdx.x < ld) {
        double t = 0.0;
        for (int r = 0; r < nrl; ++r) t += sh[r * ld + col];
        partial[(size_t)blockIdx.x * ld + col] = t;
    }
}

// One workgroup per column j < k: coef[j] = (sum_blk partial[blk][j]) / q[j]; the workgroup of column k-1 first
// closes q[k-1] = sum_blk normpart[k-1][blk]. With k == m (after the last column) only q[m-1] is closed.
__global__ __launch_bounds__(256) void k_gs_fin(const double *__restrict__ partial, const double *__restrict__ normpart, int nblk,
                                                 unsigned ld, unsigned k, unsigned m, double *__restrict__ q,
                                                 float *__restrict__ coef)
{
    __shared__ double sh[256];
    const int t = threadIdx.x;
    const unsigned j = blockIdx.x;
    double qj;
    if (j + 1 == k) {
        double s = 0.0;
        for (int b = t; b < nblk; b += 256) s += normpart[(size_t)j * nblk + b];
        sh[t] = s;
        __syncthreads();
        for (int o = 128; o > 0; o >>= 1) {
            if (t < o) sh[t] += sh[t + o];
            __syncthreads();
        }
        qj = sh[0];
        if (t == 0) q[j] = qj;
        __syncthreads();
    } else {
        qj = q[j];
    }
    if (k >= m && j + 1 == k) return; // closing call: no projection follows (uniform per workgroup)
    double s = 0.0;
    for (int b = t; b < nblk; b += 256) s += partial[(size_t)b * ld + j];
    sh[t] = s;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if (t < o) sh[t] += sh[t + o];
        __syncthreads();
    }
    if (t == 0) coef[j] = qj != 0.0 ? (float)(sh[0] / qj) : 0.f;
}

// x_k <- x_k - sum_{j<k} coef[j] u_j (VecAXPBY :53); normpart[k][blk] = sum_{i in blk} x_k[i]^2.
// ld / 4 lanes per row, one float4 each.
__global__ __launch_bounds__(256) void k_gs_apply(float *__restrict__ X, unsigned n, unsigned ld, unsigned k, int nblk,
                                                   const float *__restrict__ coef, double *__restrict__ normpart)
{
    __shared__ double sh[256];
    const int t = threadIdx.x;
    const int lpr = ld / 4, q4 = t % lpr, rl = t / lpr, rpp = 256 / lpr;
    float4 c4 = make_float4(0.f, 0.f, 0.f, 0.f);
    if (4 * q4 + 0 < (int)k) c4.x = coef[4 * q4 + 0];
    if (4 * q4 + 1 < (int)k) c4.y = coef[4 * q4 + 1];
    if (4 * q4 + 2 < (int)k) c4.z = coef[4 * q4 + 2];
    if (4 * q4 + 3 < (int)k) c4.w = coef[4 * q4 + 3];
    const bool owner = (int)(k / 4) == q4;
    double nrm = 0.0;
    const unsigned base = blockIdx.x * GS_ROWS;
    for (unsigned r = rl; r < GS_ROWS; r += rpp) {
        const unsigned i = base + r;
        const bool ok = i < n;
        float4 x = make_float4(0.f, 0.f, 0.f, 0.f);
        if (ok) x = *reinterpret_cast<const float4 *>(&X[(size_t)i * ld + 4 * q4]);
        float s = x.x * c4.x + x.y * c4.y + x.z * c4.z + x.w * c4.w;
        for (int o = lpr / 2; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
        if (ok && owner) {
            const float xk = (k % 4 == 0 ? x.x : k % 4 == 1 ? x.y : k % 4 == 2 ? x.z : x.w) - s;
            X[(size_t)i * ld + k] = xk;
            nrm += (double)xk * (double)xk;
        }
    }
    sh[t] = nrm;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if (t < o) sh[t] += sh[t + o];
        __syncthreads();
    }
    if (t == 0) normpart[(size_t)k * nblk + blockIdx.x] = sh[0];
}

__global__ void k_gs_norms(const double *__restrict__ q, unsigned m, double *__restrict__ norms)
{
    if (threadIdx.x < m) norms[threadIdx.x] = sqrt(q[threadIdx.x]);
}

// column norms only (NormaliseVecs, hpc/gram_schmidt.c:66-77)
__global__ __launch_bounds__(256) void k_col_sumsq(const float *__restrict__ X, unsigned n, unsigned ld,
                                                    double *__restrict__ partial)
{
    __shared__ double sh[256];
    const int col = threadIdx.x % ld, rl = threadIdx.x / ld, nrl = 256 / ld;
    double v[1] = {0.0};
    const unsigned base = blockIdx.x * RED_ROWS;
    for (unsigned r = rl; r < RED_ROWS; r += nrl) {
        const unsigned i = base + r;
        if (i >= n) break;
        const float x = X[(size_t)i * ld + col];
        v[0] += (double)x * (double)x;
    }
    block_col_reduce<1>(v, ld, partial, sh);
}

__global__ __launch_bounds__(256) void k_norms_from_partials(const double *__restrict__ partial, int nblk, unsigned ld,
                                                              double *__restrict__ norms)
{
    __shared__ double sh[256];
    double s[1];
    wg_sum_partials<1>(partial, nblk, ld, s, sh);
    if (threadIdx.x < ld) norms[threadIdx.x] = sqrt(s[0]);
}

__global__ void k_scale_all(float *__restrict__ X, unsigned n, unsigned ld, unsigned m, const double *__restrict__ norms)
{
    const size_t e = (size_t)blockIdx.x * 256 + threadIdx.x;
    const unsigned i = (unsigned)(e / ld), c = (unsigned)(e % ld);
    if (i >= n || c >= m) return;
    const double nr = norms[c];
    if (nr != 0.0) X[e] = (float)((double)X[e] / nr);
}


// ---- Gram-matrix form of the same classical Gram-Schmidt -------------------------------------------------
// The projections <v_k,u_j>/<u_j,u_j> of classical GS are determined by the Gram matrix G = V^T V alone:
//   <v_k,u_j> = G_kj - sum_{i<j} c_ij <v_k,u_i>,   c_jk = <v_k,u_j> / q_j,   q_k = <u_k,u_k> = G_kk - sum_j c_jk <v_k,u_j>
// (the LDL^T recurrence of G), and U = V C^-1 with C unit upper triangular. So one pass over the vectors builds G in
// f64 (exact products of f32 entries, f64 accumulation in fixed order), one workgroup runs the m-step recurrence and
// inverts C in f64, and one more pass applies X <- X (C^-1 diag(1/|u_k|)): 4 launches instead of 3 m, same result
// as the column-by-column sweep up to rounding (the sweep rounds every u_k to f32 before it is used; here the
// coefficients are f64 throughout). The recurrence loses digits as cond(V)^2 eps_f64: if any q_k falls below
// GSF_COND_FLOOR * G_kk (vectors dependent to ~5 digits) the device raises a flag, X is left untouched and the
// caller runs the column-by-column sweep instead.
constexpr int GSF_ROWS = 128;            // rows per Gram workgroup (667 workgroups at p = 85 264)
constexpr double GSF_COND_FLOOR = 1e-10; // q_k / G_kk below this: fall back to the sequential sweep

// Gpart[chunk][a][b] = sum_{i in chunk} X[i][a] X[i][b] for the tiles on and above the diagonal; 16 x 16 threads, each an
// E x E sub-block of a TILE x TILE tile.
// Y (optional): the second operand of G = X^T Y (the cross-panel blocks of more than 256 vectors); then every tile is computed.
template <int TILE>
__global__ __launch_bounds__(256) void k_gsf_gram(const float *__restrict__ X, unsigned n, unsigned ld,
                                                   double *__restrict__ Gpart, const float *__restrict__ Y = nullptr)
{
    constexpr int E = TILE / 16;
    const int mb = ld / TILE;
    const int ta = blockIdx.y / mb, tb = blockIdx.y % mb;
    if (!Y && tb < ta) return;
    const float *Yp = Y ? Y : X;
    const int tx = threadIdx.x % 16, ty = threadIdx.x / 16;
    double acc[E][E];
#pragma unroll
    for (int u = 0; u < E; ++u)
#pragma unroll
        for (int v = 0; v < E; ++v) acc[u][v] = 0.0;
    const unsigned r0 = blockIdx.x * GSF_ROWS, r1 = min(r0 + GSF_ROWS, n);
    const int ca = ta * TILE + E * ty, cb = tb * TILE + E * tx;
#pragma unroll 4
    for (unsigned i = r0; i < r1; ++i) {
        const float *row = X + (size_t)i * ld, *rowb = Yp + (size_t)i * ld;
        float a[E], b[E];
#pragma unroll
        for (int u = 0; u < E; ++u) {
            a[u] = row[ca + u];
            b[u] = rowb[cb + u];
        }
#pragma unroll
        for (int u = 0; u < E; ++u)
#pragma unroll
            for (int v = 0; v < E; ++v) acc[u][v] = fma((double)a[u], (double)b[v], acc[u][v]);
    }
    double *out = Gpart + (size_t)blockIdx.x * ld * ld;
#pragma unroll
    for (int u = 0; u < E; ++u)
#pragma unroll
        for (int v = 0; v < E; ++v) out[(size_t)(ca + u) * ld + cb + v] = acc[u][v];
}

// The same chunk sums on the f64 matrix pipe (v_mfma_f64_16x16x4_f64: exact products of the f32 entries, f64 accumulation):
// one wave per (chunk, tile pair); M = 16 columns a of X, N = 16 columns b of Y, K = 4 rows per step -- lane l holds
// X[row 4 s + (l >> 4)][a0 + (l & 15)] as the A operand and the like from Y as B; the result registers of a lane are
// G[a0 + (l >> 4) + 4 reg][b0 + (l & 15)]. The loads of 8 steps (32 rows) are issued together. The vector form reached
// ~15 TFLOP/s of f64 (49 us for 0.7 GFLOP at cfg4).
typedef double f64x4 __attribute__((ext_vector_type(4)));
template <int TILE>
__global__ __launch_bounds__(64) void k_gsf_gram_mfma(const float *__restrict__ X, unsigned n, unsigned ld,
                                                       double *__restrict__ Gpart, const float *__restrict__ Y = nullptr)
{
    constexpr int S = TILE / 16; // 16 x 16 sub-tiles per side
    const int mb = ld / TILE;
    const int ta = blockIdx.y / mb, tb = blockIdx.y % mb;
    if (!Y && tb < ta) return;
    const float *Yp = Y ? Y : X;
    const int lane = threadIdx.x, l15 = lane & 15, lq = lane >> 4;
    f64x4 acc[S][S];
#pragma unroll
    for (int u = 0; u < S; ++u)
#pragma unroll
        for (int v = 0; v < S; ++v)
#pragma unroll
            for (int r = 0; r < 4; ++r) acc[u][v][r] = 0.0;
    const unsigned r0 = blockIdx.x * GSF_ROWS, r1 = min(r0 + GSF_ROWS, n);
    const float *xa = X + ta * TILE + l15, *yb = Yp + tb * TILE + l15;
    const bool sym = !Y && ta == tb; // a diagonal tile of X^T X: the sub-tiles below its diagonal are mirror images
    for (unsigned ib = r0; ib < r1; ib += 32) {
        float a[8][S], b[8][S];
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            const unsigned i = min(ib + 4 * q + lq, r1 - 1);
#pragma unroll
            for (int u = 0; u < S; ++u) {
                a[q][u] = xa[(size_t)i * ld + 16 * u];
                b[q][u] = yb[(size_t)i * ld + 16 * u];
            }
        }
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            const bool ok = ib + 4 * q + lq < r1; // rows past the end of the chunk contribute zeros
            double da[S], db[S];
#pragma unroll
            for (int u = 0; u < S; ++u) {
                da[u] = ok ? (double)a[q][u] : 0.0;
                db[u] = ok ? (double)b[q][u] : 0.0;
            }
#pragma unroll
            for (int u = 0; u < S; ++u)
#pragma unroll
                for (int v = 0; v < S; ++v)
                    if (!sym || v >= u) acc[u][v] = __builtin_amdgcn_mfma_f64_16x16x4f64(da[u], db[v], acc[u][v], 0, 0, 0);
        }
    }
    double *out = Gpart + (size_t)blockIdx.x * ld * ld;
#pragma unroll
    for (int u = 0; u < S; ++u)
#pragma unroll
        for (int v = 0; v < S; ++v) {
            if (sym && v < u) continue;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const size_t ga = (size_t)(ta * TILE + 16 * u + lq + 4 * r), gb = (size_t)(tb * TILE + 16 * v + l15);
                out[ga * ld + gb] = acc[u][v][r];
                if (sym && v > u) out[gb * ld + ga] = acc[u][v][r]; // the mirror image inside a diagonal tile
            }
        }
}

// G = sum over chunks (fixed order: four interleaved partial sums per entry, combined through LDS), mirrored into the
// tiles below the diagonal; 64 entries per workgroup
__global__ __launch_bounds__(256) void k_gsf_sum(const double *__restrict__ Gpart, int nchunks, unsigned ld, int tile,
                                                  double *__restrict__ G, int full = 0)
{
    __shared__ double sh[256];
    const unsigned e = blockIdx.x * 64 + (threadIdx.x & 63);
    const int part = threadIdx.x >> 6;
    const unsigned a = e / ld, b = e % ld;
    const bool live = e < ld * ld && (full || b / tile >= a / tile);
    double s = 0.0;
    if (live) { // eight chunks' loads in flight at a time, added in the same order
        int c = part;
        for (; c + 28 < nchunks; c += 32) {
            double x[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) x[u] = Gpart[(size_t)(c + 4 * u) * ld * ld + e];
#pragma unroll
            for (int u = 0; u < 8; ++u) s += x[u];
        }
        for (; c < nchunks; c += 4) s += Gpart[(size_t)c * ld * ld + e];
    }
    sh[threadIdx.x] = s;
    __syncthreads();
    if (threadIdx.x < 64 && live) {
        const double tot = (sh[threadIdx.x] + sh[64 + threadIdx.x]) + (sh[128 + threadIdx.x] + sh[192 + threadIdx.x]);
        G[e] = tot;
        if (!full && b / tile > a / tile) G[(size_t)b * ld + a] = tot;
    }
}

// One workgroup. S (ld x ld, f64, LDS for ld <= 64, else global) starts as G and is reduced in place:
// upper part S[j][k] <- c_jk, diagonal <- q_j, then the strictly lower part S[k][j] <- T[j][k], T = C^-1.
// Outputs: Tn[j][k] = T[j][k] / |u_k| (upper triangular, zero elsewhere), norms[k] = |u_k| = sqrt(q_k), flag.
// REGS (ld <= 64): the Schur complement lives in registers, element (k, l) = (t / 64 + 4 r, t % 64) in a[r] of thread t;
// step j broadcasts its column j through a double-buffered LDS vector, so a step is one barrier and 18 LDS reads per
// thread instead of two barriers and a read-modify-write sweep of S in LDS (the 64 steps took ~60 of the kernel's 106 us).
template <bool REGS>
__device__ void gsf_recur_body(double *S, unsigned ld, unsigned m, const double *__restrict__ G, double *__restrict__ Tn,
                               double *__restrict__ norms, int *__restrict__ flag, int *__restrict__ flag_host)
{
    const unsigned t = threadIdx.x;
    __shared__ int bad;
    if (t == 0) bad = 0;
    if (REGS) {
        // Thread t holds the elements (row ty + 4 r, column tx) of the Schur complement (a) and of T = C^-1 (tr). T needs no
        // separate triangular inversion: u_k = v_k - sum_{j<k} c_jk u_j means T[:, k] = e_k - sum_{j<k} c_jk T[:, j], and
        // column j of T is final when step j begins -- the same rank-one update as the Schur complement's, with column j of
        // T broadcast through LDS beside column j of S (the inversion by back substitution was 40 of the kernel's 90 us).
        __shared__ __attribute__((aligned(16))) double colj[2][64];
        __shared__ double gdiag[64];
        const unsigned tx = t & 63, ty = t >> 6;
        if (t < 64) gdiag[t] = t < m ? G[(size_t)t * ld + t] : 0.0; // (a global load per step for the check below cost 1 us each)
        double a[16], tr[16], myq = 0.0;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const unsigned k = ty + 4 * r;
            a[r] = (k < ld && tx < ld) ? G[(size_t)k * ld + tx] : 0.0;
            tr[r] = k == tx ? 1.0 : 0.0;
        }
        __syncthreads();
        // Column j travels as ONE vector: element k is S[k][j] for k > j (what the Schur update reads), T[k][j] for k < j and
        // q_j at k = j -- a thread needs exactly one of the two per row -- stored wave by wave (k = ty + 4 r at ty * 16 + r),
        // so that the four lanes owning the column write their 16 values as 8 ds_write_b128 instead of 32 single stores
        // executed by whole waves (a third of a step).
        for (unsigned j = 0; j < m; ++j) {
            double *cj = colj[j & 1];
            if (tx == j) {
                double v[16];
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const unsigned k = ty + 4 * r;
                    v[r] = k >= j ? a[r] : tr[r];
                }
#pragma unroll
                for (int r = 0; r < 16; r += 2) *reinterpret_cast<double2 *>(&cj[ty * 16 + r]) = make_double2(v[r], v[r + 1]);
            }
            __syncthreads(); // (the buffer written two steps ago is free: every thread passed the barrier of step j - 1 since)
            const double qj = cj[(j & 3) * 16 + (j >> 2)];
            if (tx == j) myq = qj;
            if (t == 0 && !(qj > GSF_COND_FLOOR * gdiag[j]) && gdiag[j] > 0.0) bad = 1;
            // c_j,tx (0 for a zero vector, as the sweep does); T[j][j] = 1 is not in the vector (the slot holds q_j)
            const double cl = (qj != 0.0 && tx > j && tx < m) ? cj[(tx & 3) * 16 + (tx >> 2)] / qj : 0.0;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const unsigned k = ty + 4 * r;
                const double cv = k == j ? 1.0 : cj[ty * 16 + r];
                if (k > j && k < m) a[r] -= cv * cl; // S[k][l] -= <v_k,u_j> c_jl, k, l > j
                if (k <= j) tr[r] -= cv * cl;        // T[:, l] -= c_jl T[:, j] (T[k][j] = 0 below the diagonal, T[j][j] = 1)
            }
        }
        // Tn[j][k] = T[j][k] / |u_k| (upper triangular, zero elsewhere; a zero norm leaves the column unscaled), norms, flag
        const double nk = myq > 0.0 ? sqrt(myq) : 0.0, sc = nk != 0.0 ? 1.0 / nk : 1.0;
        if (tx < ld) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const unsigned i = ty + 4 * r;
                if (i < ld) Tn[(size_t)i * ld + tx] = (tx < m && i <= tx) ? tr[r] * sc : 0.0;
            }
            if (ty == 0 && tx < m) norms[tx] = nk;
        }
        __syncthreads();
        if (t == 0) {
            *flag = bad;
            if (flag_host) *flag_host = bad; // pinned host memory: read after a stream synchronise, no copy launch
        }
        return;
    } else {
    for (unsigned e = t; e < ld * ld; e += 256) S[e] = G[e];
    __syncthreads();
    for (unsigned j = 0; j < m; ++j) {
        const double qj = S[(size_t)j * ld + j];
        if (t == 0 && !(qj > GSF_COND_FLOOR * G[(size_t)j * ld + j]) && G[(size_t)j * ld + j] > 0.0) bad = 1;
        // c_jk for k > j (0 for a zero vector, as the sweep does)
        for (unsigned k = j + 1 + t; k < m; k += 256) S[(size_t)j * ld + k] = qj != 0.0 ? S[(size_t)k * ld + j] / qj : 0.0;
        __syncthreads();
        // Schur complement: S[k][l] -= c_jk <v_l,u_j>, k, l > j  (<v_l,u_j> = S[l][j], still untouched)
        const unsigned w = m - j - 1;
        for (unsigned e = t; e < w * w; e += 256) {
            const unsigned k = j + 1 + e / w, l = j + 1 + e % w;
            S[(size_t)k * ld + l] -= S[(size_t)j * ld + k] * S[(size_t)l * ld + j];
        }
        __syncthreads(); // (column j below the diagonal is read above and only overwritten by the inversion below)
    }
    }
    // T = C^-1, column k by thread k: T[k][k] = 1, T[j][k] = -sum_{i=j+1..k} c_ji T[i][k]; stored at S[k][j] (j < k)
    // Four adjacent lanes share a column (the dot product over i is dealt to them mod 4 and summed by two shuffles), and
    // the operands of four terms are read before they are used: the single-lane loop was a chain of ~2000 dependent LDS
    // round trips for the last column (100 of the kernel's 170 us). A lane reads T[i][k] entries its quad's lane 0 wrote
    // in earlier iterations: same wave, and a wave's LDS / global accesses complete in order.
    for (unsigned k = t >> 2; k < m; k += 64) {
        const unsigned s4 = t & 3;
        for (int j = (int)k - 1; j >= 0; --j) {
            double acc = 0.0;
            unsigned i = (unsigned)j + 1 + s4;
            for (; i + 12 < k; i += 16) {
                const double a0 = S[(size_t)j * ld + i], a1 = S[(size_t)j * ld + i + 4], a2 = S[(size_t)j * ld + i + 8],
                             a3 = S[(size_t)j * ld + i + 12];
                const double b0 = S[(size_t)k * ld + i], b1 = S[(size_t)k * ld + i + 4], b2 = S[(size_t)k * ld + i + 8],
                             b3 = S[(size_t)k * ld + i + 12];
                acc = fma(a0, b0, acc);
                acc = fma(a1, b1, acc);
                acc = fma(a2, b2, acc);
                acc = fma(a3, b3, acc);
            }
            for (; i < k; i += 4) acc = fma(S[(size_t)j * ld + i], S[(size_t)k * ld + i], acc);
            acc += __shfl_xor(acc, 1);
            acc += __shfl_xor(acc, 2);
            if (s4 == 0) S[(size_t)k * ld + j] = -(acc + S[(size_t)j * ld + k]); // i = k term: c_jk * 1
        }
    }
    __syncthreads();
    for (unsigned e = t; e < ld * ld; e += 256) {
        const unsigned j = e / ld, k = e % ld;
        double v = 0.0;
        if (k < m && j <= k) {
            const double qk = S[(size_t)k * ld + k];
            const double nk = qk > 0.0 ? sqrt(qk) : 0.0;
            v = (j == k ? 1.0 : S[(size_t)k * ld + j]) * (nk != 0.0 ? 1.0 / nk : 1.0); // zero norm: left unscaled
        }
        Tn[e] = v;
    }
    if (t < m) {
        const double qk = S[(size_t)t * ld + t];
        norms[t] = qk > 0.0 ? sqrt(qk) : 0.0;
    }
    if (t == 0) {
        *flag = bad;
        if (flag_host) *flag_host = bad; // pinned host memory: read after a stream synchronise, no copy launch
    }
}

__global__ __launch_bounds__(256) void k_gsf_recur_lds(unsigned ld, unsigned m, const double *__restrict__ G,
                                                        double *__restrict__ Tn, double *__restrict__ norms,
                                                        int *__restrict__ flag, int *__restrict__ flag_host)
{
    __shared__ double S[64 * 64];
    gsf_recur_body<true>(S, ld, m, G, Tn, norms, flag, flag_host);
}

__global__ __launch_bounds__(256) void k_gsf_recur_global(unsigned ld, unsigned m, const double *__restrict__ G,
                                                           double *S, double *__restrict__ Tn, double *__restrict__ norms,
                                                           int *__restrict__ flag, int *__restrict__ flag_host)
{
    gsf_recur_body<false>(S, ld, m, G, Tn, norms, flag, flag_host);
}

// X[i][k] <- sum_{j<=k} X[i][j] Tn[j][k] (f64 accumulation), in place; GSF_APPLY_TILE / ld rows per workgroup (32 at ld = 64:
// with 128 the 42 workgroups of a 5329-sample block left most of the chip idle), thread = column.
constexpr unsigned GSF_APPLY_TILE = 2048;
__global__ __launch_bounds__(256) void k_gsf_apply(float *__restrict__ X, unsigned n, unsigned ld, unsigned m,
                                                    const double *__restrict__ Tn, const int *__restrict__ flag)
{
    if (*flag) return; // ill-conditioned: the caller runs the sequential sweep on the untouched X
    __shared__ float xs[GSF_APPLY_TILE];
    __shared__ double ts[4096];
    const unsigned rows = GSF_APPLY_TILE / ld, jb = min(4096u / ld, ld); // rows per workgroup; T rows per staged block (Tn has ld rows)
    const unsigned r0 = blockIdx.x * rows;
    const unsigned col = threadIdx.x % ld, rl = threadIdx.x / ld, nrl = 256 / ld, rpt = rows / nrl; // rpt = 8
    for (unsigned e = threadIdx.x; e < rows * ld; e += 256) {
        const unsigned i = r0 + e / ld;
        xs[e] = i < n ? X[(size_t)i * ld + e % ld] : 0.f;
    }
    double acc[8];
#pragma unroll
    for (int q = 0; q < 8; ++q) acc[q] = 0.0;
    for (unsigned j0 = 0; j0 < m; j0 += jb) {
        __syncthreads();
        for (unsigned e = threadIdx.x; e < jb * ld; e += 256) ts[e] = Tn[(size_t)j0 * ld + e];
        __syncthreads();
        const unsigned jn = min(jb, m - j0);
        for (unsigned j = 0; j < jn; ++j) {
            const double tv = ts[j * ld + col];
#pragma unroll
            for (int q = 0; q < 8; ++q) acc[q] = fma((double)xs[(rl + q * nrl) * ld + j0 + j], tv, acc[q]);
        }
    }
    (void)rpt;
    if (col < m)
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            const unsigned i = r0 + rl + q * nrl;
            if (i < n) X[(size_t)i * ld + col] = (float)acc[q];
        }
}

// The same product for ld <= 64 on the f64 matrix pipe: M = 16 rows of X, N = 16 columns k, K = the columns j in blocks of 16
// (within a block lane group q = lane / 16 takes j = 16 jb + 4 q + s at step s, so a lane reads one float4 of its row per
// block). Tn is upper triangular: column tile t needs the blocks jb <= t only (40 of the 64 MFMAs of a row tile). Tn sits in
// LDS with rows of LD + 4 doubles (the two j rows a half-wave reads are 4 apart: 32 banks apart with that stride).
// `tiles` row tiles per wave (1 .. 4: enough workgroups for every CU before a workgroup's copy of Tn is shared by more rows).
template <int LD>
__global__ __launch_bounds__(256) void k_gsf_apply_mfma(float *__restrict__ X, unsigned n, unsigned m, const double *__restrict__ Tn,
                                                         const int *__restrict__ flag, int tiles)
{
    if (*flag) return; // ill-conditioned: the caller runs the sequential sweep on the untouched X
    constexpr int NTL = LD / 16, TS = LD + 4;
    __shared__ double ts[LD * TS];
    for (unsigned e = threadIdx.x; e < LD * LD; e += 256) ts[(e / LD) * TS + e % LD] = Tn[e];
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, l15 = lane & 15, lq = lane >> 4;
#pragma unroll 1
    for (int tile = 0; tile < tiles; ++tile) {
        const unsigned row0 = (blockIdx.x * 4 * tiles + wave * tiles + tile) * 16;
        if (row0 >= n) break; // (wave-uniform; no barrier below)
        const unsigned ra = min(row0 + l15, n - 1); // this lane's row as the A operand (a row past the end repeats the last: not stored)
        float4 xa[NTL];
#pragma unroll
        for (int jb = 0; jb < NTL; ++jb) xa[jb] = *reinterpret_cast<const float4 *>(X + (size_t)ra * LD + 16 * jb + 4 * lq);
        f64x4 acc[NTL];
#pragma unroll
        for (int t = 0; t < NTL; ++t)
#pragma unroll
            for (int r = 0; r < 4; ++r) acc[t][r] = 0.0;
#pragma unroll
        for (int jb = 0; jb < NTL; ++jb) {
            const double a4[4] = {(double)xa[jb].x, (double)xa[jb].y, (double)xa[jb].z, (double)xa[jb].w};
#pragma unroll
            for (int sk = 0; sk < 4; ++sk)
#pragma unroll
                for (int t = jb; t < NTL; ++t) // T[j][k] = 0 for j > k: the column tiles left of the block's diagonal get nothing
                    acc[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(a4[sk], ts[(16 * jb + 4 * lq + sk) * TS + 16 * t + l15], acc[t], 0, 0, 0);
        }
        // result register r of a lane: row lq + 4 r of the tile, column 16 t + l15
#pragma unroll
        for (int t = 0; t < NTL; ++t)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const unsigned i = row0 + lq + 4 * r;
                if (i < n && 16 * t + l15 < (int)m) X[(size_t)i * LD + 16 * t + l15] = (float)acc[t][r];
            }
    }
}

// Y[i][c] -= sum_a X[i][a] C[a][c] (f64 accumulation), a < ma: the projection of one block of vectors onto an earlier,
// already orthonormal block (the cross-panel terms of classical Gram-Schmidt for more than 256 vectors, and X (X^T A X) of the
// residual). Same tiling as k_gsf_apply.
__global__ __launch_bounds__(256) void k_gsf_sub(float *__restrict__ Y, const float *__restrict__ X, unsigned n, unsigned ld, unsigned ma,
                                                  const double *__restrict__ Cm)
{
    __shared__ float xs[GSF_APPLY_TILE];
    __shared__ double ts[4096];
    const unsigned rows = GSF_APPLY_TILE / ld, jb = min(4096u / ld, ld);
    const unsigned r0 = blockIdx.x * rows;
    const unsigned col = threadIdx.x % ld, rl = threadIdx.x / ld, nrl = 256 / ld;
    for (unsigned e = threadIdx.x; e < rows * ld; e += 256) {
        const unsigned i = r0 + e / ld;
        xs[e] = i < n ? X[(size_t)i * ld + e % ld] : 0.f;
    }
    double acc[8];
#pragma unroll
    for (int q = 0; q < 8; ++q) acc[q] = 0.0;
    for (unsigned j0 = 0; j0 < ma; j0 += jb) {
        __syncthreads();
        for (unsigned e = threadIdx.x; e < jb * ld; e += 256) ts[e] = Cm[(size_t)j0 * ld + e];
        __syncthreads();
        const unsigned jn = min(jb, ma - j0);
        for (unsigned j = 0; j < jn; ++j) {
            const double tv = ts[j * ld + col];
#pragma unroll
            for (int q = 0; q < 8; ++q) acc[q] = fma((double)xs[(rl + q * nrl) * ld + j0 + j], tv, acc[q]);
        }
    }
#pragma unroll
    for (int q = 0; q < 8; ++q) {
        const unsigned i = r0 + rl + q * nrl;
        if (i < n) Y[(size_t)i * ld + col] = (float)((double)Y[(size_t)i * ld + col] - acc[q]);
    }
}

static void launch_gsf_apply(hipStream_t st, float *X, unsigned n, unsigned ld, unsigned m, const double *Tn, const int *flag)
{
    const int tiles = (int)std::max<int64_t>(1, std::min<int64_t>(4, ceil_div(n, 16 * 4 * 256)));
    const unsigned rows_wg = 16 * 4 * (unsigned)tiles;
    if (ld == 64)
        hipLaunchKernelGGL(k_gsf_apply_mfma<64>, dim3((unsigned)ceil_div(n, rows_wg)), dim3(256), 0, st, X, n, m, Tn, flag, tiles);
    else if (ld == 32)
        hipLaunchKernelGGL(k_gsf_apply_mfma<32>, dim3((unsigned)ceil_div(n, rows_wg)), dim3(256), 0, st, X, n, m, Tn, flag, tiles);
    else
        hipLaunchKernelGGL(k_gsf_apply, dim3((unsigned)ceil_div(n, GSF_APPLY_TILE / ld)), dim3(256), 0, st, X, n, ld, m, Tn, flag);
}

struct GsFusedWork {
    DevBuf<double> Gpart, G, S, Tn;
    DevBuf<int> flag;
    int *h_flag = nullptr; // in the context's pinned page
    int nchunks = 0;
    unsigned n = 0, ld = 0;
    int init(glf_ctx *ctx, unsigned n_, unsigned ld_)
    {
        if (n == n_ && ld == ld_) return GLF_OK;
        n = n_;
        ld = ld_;
        nchunks = (int)ceil_div(n, GSF_ROWS);
        GLF_TRY(Gpart.alloc(ctx, (size_t)std::max(1, nchunks) * ld * ld));
        GLF_TRY(G.alloc(ctx, (size_t)ld * ld));
        GLF_TRY(Tn.alloc(ctx, (size_t)ld * ld));
        if (ld > 64) GLF_TRY(S.alloc(ctx, (size_t)ld * ld));
        GLF_TRY(flag.alloc(ctx, 1));
        if (!ctx_pinned(ctx)) return set_error(ctx, GLF_ERR_NOMEM, "pinned host page");
        h_flag = reinterpret_cast<int *>(ctx_pinned(ctx) + PINNED_GSFLAG);
        return GLF_OK;
    }
};

struct GsWork {
    DevBuf<double> partial, norms, q, normpart;
    DevBuf<float> coef;
    GsFusedWork fused;
    bool last_fused = false; // the last orthonormalise_dev took the Gram-matrix form (fused.Tn holds X_new = X_old Tn)
    int nblk = 0;
    int init(glf_ctx *ctx, unsigned n, unsigned ld)
    {
        nblk = (int)ceil_div(n, GS_ROWS);
        // `partial` is also the scratch of normalise_dev (RED_ROWS blocks, fewer)
        GLF_TRY(partial.alloc(ctx, (size_t)nblk * 2 * ld));
        GLF_TRY(norms.alloc(ctx, ld));
        GLF_TRY(q.alloc(ctx, ld));
        GLF_TRY(coef.alloc(ctx, ld));
        GLF_TRY(normpart.alloc(ctx, (size_t)nblk * ld));
        GLF_HIP(ctx, hipMemsetAsync(norms.p, 0, sizeof(double) * ld, ctx->stream));
        GLF_HIP(ctx, hipMemsetAsync(q.p, 0, sizeof(double) * ld, ctx->stream));
        GLF_HIP(ctx, hipMemsetAsync(coef.p, 0, sizeof(float) * ld, ctx->stream));
        return GLF_OK;
    }
};

// returns GLF_OK with *fell_back = 1 when the device found V too ill-conditioned (X untouched)
// rows: this rank's rows of X (X points at row 0 of the whole block); dist: the Gram block is all-reduced
static int orthonormalise_fused_dev(glf_ctx *ctx, GsFusedWork &f, float *X, Rows rows, unsigned m, unsigned ld,
                                    double *d_norms, int *fell_back)
{
    hipStream_t st = ctx->stream;
    const unsigned n = rows.n();
    float *Xl = X + (size_t)rows.r0 * ld;
    GLF_TRY(f.init(ctx, n, ld));
    const int tile = ld >= 64 ? 64 : 32, mb = (int)ld / tile;
    if (f.nchunks > 0) {
        if (tile == 64)
            hipLaunchKernelGGL((k_gsf_gram_mfma<64>), dim3(f.nchunks, mb * mb), dim3(64), 0, st, Xl, n, ld, f.Gpart.p);
        else
            hipLaunchKernelGGL((k_gsf_gram_mfma<32>), dim3(f.nchunks, mb * mb), dim3(64), 0, st, Xl, n, ld, f.Gpart.p);
    }
    hipLaunchKernelGGL(k_gsf_sum, dim3((ld * ld + 63) / 64), dim3(256), 0, st, f.Gpart.p, f.nchunks, ld, tile, f.G.p);
    GLF_LAUNCH_CHECK(ctx);
    if (rows.dist) GLF_TRY(allreduce_d(ctx, f.G.p, (size_t)ld * ld)); // G = V^T V over all ranks' rows (hpc/gram_schmidt.c:14-15)
    if (ld <= 64)
        hipLaunchKernelGGL(k_gsf_recur_lds, dim3(1), dim3(256), 0, st, ld, m, f.G.p, f.Tn.p, d_norms, f.flag.p, f.h_flag);
    else
        hipLaunchKernelGGL(k_gsf_recur_global, dim3(1), dim3(256), 0, st, ld, m, f.G.p, f.S.p, f.Tn.p, d_norms, f.flag.p, f.h_flag);
    if (n > 0)
        launch_gsf_apply(st, Xl, n, ld, m, f.Tn.p, f.flag.p);
    GLF_LAUNCH_CHECK(ctx);
    GLF_HIP(ctx, hipStreamSynchronize(st));
    *fell_back = *(volatile int *)f.h_flag;
    return GLF_OK;
}

static int orthonormalise_seq_dev(glf_ctx *ctx, GsWork &w, float *X, unsigned n, unsigned m, unsigned ld)
{
    hipStream_t st = ctx->stream;
    for (unsigned k = 0; k < m; ++k) {
        if (k > 0) {
            hipLaunchKernelGGL(k_gs_dots, dim3(w.nblk), dim3(256), 0, st, X, n, ld, k, w.partial.p);
            hipLaunchKernelGGL(k_gs_fin, dim3(k), dim3(256), 0, st, w.partial.p, w.normpart.p, w.nblk, ld, k, m, w.q.p, w.coef.p);
        }
        hipLaunchKernelGGL(k_gs_apply, dim3(w.nblk), dim3(256), 0, st, X, n, ld, k, w.nblk, w.coef.p, w.normpart.p);
    }
    // close q[m-1], then norms and the deferred VecNormalize of every column
    hipLaunchKernelGGL(k_gs_fin, dim3(m), dim3(256), 0, st, w.partial.p, w.normpart.p, w.nblk, ld, m, m, w.q.p, w.coef.p);
    hipLaunchKernelGGL(k_gs_norms, dim3(1), dim3(256), 0, st, w.q.p, m, w.norms.p);
    hipLaunchKernelGGL(k_scale_all, dim3((unsigned)ceil_div((int64_t)n * ld, 256)), dim3(256), 0, st, X, n, ld, m, w.norms.p);
    GLF_LAUNCH_CHECK(ctx);
    return GLF_OK;
}

// OrthonormaliseVecs (hpc/gram_schmidt.c:29-64): the Gram-matrix form unless GLF_GS=seq or the vectors are too
// ill-conditioned for it. Either way w.norms holds the post-GS norms |u_k|.
// n = rows of the whole block; rows = the rows this rank updates. A sharded call that has to fall back to the column
// sweep first all-gathers X (shard != nullptr) and then sweeps all n rows on every rank (identical results everywhere):
// *replicated tells the caller that X is whole and valid on every rank afterwards.
static int orthonormalise_dev(glf_ctx *ctx, GsWork &w, float *X, unsigned n, unsigned m, unsigned ld, Rows rows = Rows{},
                              const MatShard *shard = nullptr, bool *replicated = nullptr)
{
    if (!rows.dist) rows.r1 = n, rows.r0 = 0;
    if (replicated) *replicated = false;
    w.last_fused = false;
    if (!ctx->tune.gs_seq) {
        int fell_back = 0;
        GLF_TRY(orthonormalise_fused_dev(ctx, w.fused, X, rows, m, ld, w.norms.p, &fell_back));
        if (!fell_back) {
            w.last_fused = true;
            return GLF_OK;
        }
        if (ctx->tune.verbose) fprintf(stderr, "[glf] Gram-Schmidt: ill-conditioned block, column-by-column sweep\n");
    }
    if (rows.dist) {
        GLF_TRY(allgather_rows(ctx, X, shard, ld));
        if (replicated) *replicated = true;
    }
    return orthonormalise_seq_dev(ctx, w, X, n, m, ld);
}

int orthonormalise(glf_ctx *ctx, float *X, unsigned n, unsigned m, unsigned ld, double *h_norms)
{
    if (!valid_ld(ld) || m > ld) return set_error(ctx, GLF_ERR_INVALID, "orthonormalise: ld=%u m=%u", ld, m);
    GsWork w;
    GLF_TRY(w.init(ctx, n, ld));
    GLF_TRY(orthonormalise_dev(ctx, w, X, n, m, ld));
    if (h_norms) GLF_HIP(ctx, hipMemcpyAsync(h_norms, w.norms.p, sizeof(double) * m, hipMemcpyDeviceToHost, ctx->stream));
    GLF_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return GLF_OK;
}

// NormaliseVecs (hpc/gram_schmidt.c:66-77) on this rank's rows; sharded: the squared column norms are all-reduced
static int normalise_dev(glf_ctx *ctx, float *X, unsigned n, unsigned m, unsigned ld, double *d_partial, double *d_norms,
                         Rows rows = Rows{}, double *d_sums = nullptr)
{
    if (!rows.dist) rows.r1 = n, rows.r0 = 0;
    const unsigned nloc = rows.n();
    float *Xl = X + (size_t)rows.r0 * ld;
    const int nblk = (int)ceil_div(nloc, RED_ROWS);
    hipStream_t st = ctx->stream;
    if (nblk > 0) hipLaunchKernelGGL(k_col_sumsq, dim3(nblk), dim3(256), 0, st, Xl, nloc, ld, d_partial);
    if (rows.dist) {
        hipLaunchKernelGGL(k_sum_partials_nv<1>, dim3(1), dim3(256), 0, st, d_partial, nblk, ld, d_sums);
        GLF_LAUNCH_CHECK(ctx);
        GLF_TRY(allreduce_d(ctx, d_sums, ld));
        hipLaunchKernelGGL(k_norms_from_partials, dim3(1), dim3(256), 0, st, d_sums, 1, ld, d_norms);
    } else {
        hipLaunchKernelGGL(k_norms_from_partials, dim3(1), dim3(256), 0, st, d_partial, nblk, ld, d_norms);
    }
    if (nloc > 0)
        hipLaunchKernelGGL(k_scale_all, dim3((unsigned)ceil_div((int64_t)nloc * ld, 256)), dim3(256), 0, st, Xl, nloc, ld, m, d_norms);
    GLF_LAUNCH_CHECK(ctx);
    return GLF_OK;
}

int normalise(glf_ctx *ctx, float *X, unsigned n, unsigned m, unsigned ld, double *h_norms)
{
    if (!valid_ld(ld) || m > ld) return set_error(ctx, GLF_ERR_INVALID, "normalise: ld=%u m=%u", ld, m);
    GsWork w;
    GLF_TRY(w.init(ctx, n, ld));
    GLF_TRY(normalise_dev(ctx, X, n, m, ld, w.partial.p, w.norms.p));
    if (h_norms) GLF_HIP(ctx, hipMemcpyAsync(h_norms, w.norms.p, sizeof(double) * m, hipMemcpyDeviceToHost, ctx->stream));
    GLF_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return GLF_OK;
}

// =====================================================================================
// Residual || (I - X X^T) A X ||_F  (hpc/inverse_power_it.c:49-80) as || AX - X (X^T AX) ||_F
// =====================================================================================

constexpr int GRAM_ROWS = 256; // rows per Gram workgroup chunk

// Gpart[chunk][a][b] = sum_{i in chunk} X[i][a] * Y[i][b]; one wave per 32x32 (a,b) tile.
__global__ __launch_bounds__(64) void k_gram(const float *__restrict__ X, const float *__restrict__ Y, unsigned n,
                                              unsigned ld, float *__restrict__ Gpart)
{
    const int lane = threadIdx.x, half = lane >> 5, l31 = lane & 31;
    const int mb = ld / 32;
    const int ta = blockIdx.x / mb, tb = blockIdx.x % mb;
    const unsigned r0 = blockIdx.y * GRAM_ROWS;
    const unsigned r1 = min(r0 + GRAM_ROWS, n);
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    // The two halves of the wave supply rows i0 and i0 + 1 (zeros past the last row: with an odd count the next row belongs
    // to another rank's block). The loads of 8 MFMAs are issued together, unconditionally (row index clamped, value zeroed
    // afterwards): one conditional pair of loads per MFMA made the kernel a chain of 128 exposed memory latencies (45 us
    // for 3 us of matrix work).
    const float *xa = X + 32 * ta + l31, *yb = Y + 32 * tb + l31;
    for (unsigned ib = r0; ib < r1; ib += 16) {
        float a[8], b[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const unsigned i = min(ib + 2 * u + half, r1 - 1);
            a[u] = xa[(size_t)i * ld];
            b[u] = yb[(size_t)i * ld];
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const bool ok = ib + 2 * u + half < r1;
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(ok ? a[u] : 0.f, ok ? b[u] : 0.f, acc, 0, 0, 0);
        }
    }
    float *g = Gpart + (size_t)blockIdx.y * ld * ld;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int a = 32 * ta + (r & 3) + 8 * (r >> 2) + 4 * half;
        g[(size_t)a * ld + 32 * tb + l31] = acc[r];
    }
}

// 64 entries per workgroup; the 4 thread rows take interleaved chunks (fixed order)
__global__ __launch_bounds__(256) void k_gram_sum(const float *__restrict__ Gpart, int nchunks, unsigned ld, float *__restrict__ G)
{
    __shared__ double sh[256];
    const unsigned e = blockIdx.x * 64 + (threadIdx.x & 63);
    const int part = threadIdx.x >> 6;
    double s = 0.0;
    if (e < ld * ld) { // eight chunks' loads in flight at a time, added in the same order
        int c = part;
        for (; c + 28 < nchunks; c += 32) {
            float x[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) x[u] = Gpart[(size_t)(c + 4 * u) * ld * ld + e];
#pragma unroll
            for (int u = 0; u < 8; ++u) s += (double)x[u];
        }
        for (; c < nchunks; c += 4) s += (double)Gpart[(size_t)c * ld * ld + e];
    }
    sh[threadIdx.x] = s;
    __syncthreads();
    if (threadIdx.x < 64 && e < ld * ld) G[e] = (float)((sh[threadIdx.x] + sh[64 + threadIdx.x]) + (sh[128 + threadIdx.x] + sh[192 + threadIdx.x]));
}

// partial[blk][b] = sum_i (Y[i][b] - sum_a X[i][a] G[a][b])^2
__global__ __launch_bounds__(256) void k_resid(const float *__restrict__ X, const float *__restrict__ Y,
                                                const float *__restrict__ G, unsigned p, unsigned ld, unsigned m,
                                                double *__restrict__ partial)
{
    __shared__ double sh[256];
    __shared__ float xrow[8][256];
    __shared__ float Gs[64 * 64]; // G for ld <= 64: the inner loop read it from global memory, one dependent load per term
    const int col = threadIdx.x % ld, rl = threadIdx.x / ld, nrl = 256 / ld;
    const bool g_lds = ld <= 64;
    if (g_lds)
        for (unsigned e = threadIdx.x; e < ld * ld; e += 256) Gs[e] = G[e];
    const float *Gc = g_lds ? Gs : G;
    double v[1] = {0.0};
    const unsigned base = blockIdx.x * RED_ROWS;
    for (unsigned r0 = 0; r0 < RED_ROWS; r0 += nrl) {
        const unsigned i = base + r0 + rl;
        const bool ok = i < p;
        __syncthreads();
        xrow[rl][col] = ok ? X[(size_t)i * ld + col] : 0.f;
        __syncthreads();
        if (ok && col < (int)m) {
            float s = Y[(size_t)i * ld + col];
            for (unsigned a = 0; a < m; ++a) s = fmaf(-xrow[rl][a], Gc[(size_t)a * ld + col], s);
            v[0] += (double)s * (double)s;
        }
        if (base + r0 + nrl > p && base + r0 >= p) break; // uniform: whole group past the end
    }
    block_col_reduce<1>(v, ld, partial, sh);
}

// The same partial sums for ld <= 64 on the f32 matrix pipe (exact f32 products, f32 accumulation like the vector form's FMA
// chain): a workgroup takes RED_ROWS = 128 rows, wave w the 32 rows 32 w .. 32 w + 31 as the M tile of v_mfma_f32_32x32x2_f32,
// K = the ld columns of X (lane half h takes a = (ld / 2) h + s at step s, so that a lane reads ld / 2 consecutive floats of its
// row), N = 32-column tiles of G read from LDS. The vector form spent 67 us on 0.7 GFLOP at cfg4 (64 dependent FMAs per row
// and thread, two barriers per four rows).
template <int LD>
__global__ __launch_bounds__(256) void k_resid_mfma(const float *__restrict__ X, const float *__restrict__ Y,
                                                     const float *__restrict__ G, unsigned p, unsigned m, double *__restrict__ partial)
{
    constexpr int NTL = LD / 32, KH = LD / 2; // n-tiles; k-steps (two columns of X each)
    __shared__ float Gs[LD * LD];
    __shared__ double colsum[4][LD];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, half = lane >> 5, l31 = lane & 31;
    for (unsigned e = threadIdx.x; e < LD * LD; e += 256) Gs[e] = G[e];
    __syncthreads();
    const unsigned row0 = blockIdx.x * RED_ROWS + 32 * wave, rowa = row0 + l31; // this lane's row as the A operand
    float xa[KH];
    {
        const float *xr = X + (size_t)min(rowa, p - 1) * LD + KH * half;
#pragma unroll
        for (int q = 0; q < KH / 4; ++q) {
            const float4 v4 = reinterpret_cast<const float4 *>(xr)[q];
            xa[4 * q] = v4.x, xa[4 * q + 1] = v4.y, xa[4 * q + 2] = v4.z, xa[4 * q + 3] = v4.w;
        }
        if (rowa >= p)
#pragma unroll
            for (int q = 0; q < KH; ++q) xa[q] = 0.f;
    }
    f32x16 acc[NTL];
#pragma unroll
    for (int nt = 0; nt < NTL; ++nt)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[nt][r] = 0.f;
#pragma unroll
    for (int sk = 0; sk < KH; ++sk)
#pragma unroll
        for (int nt = 0; nt < NTL; ++nt)
            acc[nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(xa[sk], Gs[(KH * half + sk) * LD + 32 * nt + l31], acc[nt], 0, 0, 0);
    // accumulator register r of a lane: row (r & 3) + 8 (r >> 2) + 4 half of the tile, column 32 nt + l31
#pragma unroll
    for (int nt = 0; nt < NTL; ++nt) {
        double v = 0.0;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const unsigned i = row0 + (r & 3) + 8 * (r >> 2) + 4 * half;
            if (i < p) {
                const float sres = Y[(size_t)i * LD + 32 * nt + l31] - acc[nt][r];
                v += (double)sres * (double)sres;
            }
        }
        v += __shfl_xor(v, 32, 64); // the two halves hold different rows of the same column
        if (half == 0) colsum[wave][32 * nt + l31] = 32 * nt + l31 < (int)m ? v : 0.0;
    }
    __syncthreads();
    if (threadIdx.x < LD)
        partial[(size_t)blockIdx.x * LD + threadIdx.x] =
            (colsum[0][threadIdx.x] + colsum[1][threadIdx.x]) + (colsum[2][threadIdx.x] + colsum[3][threadIdx.x]);
}

struct ResWork {
    DevBuf<float> AX, Gpart, G;
    DevBuf<double> partial, sums;
    double *h_sums = nullptr; // in the context's pinned page: the last reduction kernel writes into it (no D2H copy launch)
    int nchunks = 0;
    Rows rows;
    const MatShard *shard = nullptr;
    int init(glf_ctx *ctx, unsigned p, unsigned ld, const MatShard *sh = nullptr)
    {
        shard = sh;
        rows = rows_of(p, sh);
        const size_t vr = vec_rows(p, sh, ctx->comm.size);
        nchunks = (int)ceil_div(rows.n(), GRAM_ROWS);
        GLF_TRY(AX.alloc(ctx, vr * ld));
        GLF_HIP(ctx, hipMemsetAsync(AX.p, 0, sizeof(float) * vr * ld, ctx->stream));
        GLF_TRY(Gpart.alloc(ctx, (size_t)std::max(1, nchunks) * ld * ld));
        GLF_TRY(G.alloc(ctx, (size_t)ld * ld));
        GLF_TRY(partial.alloc(ctx, (size_t)std::max<int64_t>(1, ceil_div(rows.n(), RED_ROWS)) * ld));
        GLF_TRY(sums.alloc(ctx, ld));
        if (!ctx_pinned(ctx) || ld > 256) return set_error(ctx, GLF_ERR_NOMEM, "pinned host page");
        h_sums = reinterpret_cast<double *>(ctx_pinned(ctx) + PINNED_SUMS);
        return GLF_OK;
    }
};

// ax_ready: w.AX already holds (this rank's rows of) A X, derived from the PCG state (see inverse_power_iteration) -- no
// sweep over L_A. Sharded: X^T A X (ld x ld) and the ld squared column norms are all-reduced (hpc/inverse_power_it.c:55-72).
static int residual_dev(glf_ctx *ctx, ResWork &w, const float *A, int64_t lda, unsigned p, float *X, unsigned m,
                        unsigned ld, double *h_out, bool ax_ready = false)
{
    const Rows rows = w.rows;
    const unsigned nloc = rows.n();
    const size_t off = (size_t)rows.r0 * ld;
    const int nblk = (int)ceil_div(nloc, RED_ROWS);
    hipStream_t st = ctx->stream;
    if (!ax_ready) {
        if (rows.dist) GLF_TRY(allgather_rows(ctx, X, w.shard, ld)); // the operator needs every row of X
        GLF_TRY(block_matvec(ctx, A, lda, p, X, w.AX.p, ld, w.shard));
    }
    const int mb = ld / 32;
    if (w.nchunks > 0) hipLaunchKernelGGL(k_gram, dim3(mb * mb, w.nchunks), dim3(64), 0, st, X + off, w.AX.p + off, nloc, ld, w.Gpart.p);
    hipLaunchKernelGGL(k_gram_sum, dim3((ld * ld + 63) / 64), dim3(256), 0, st, w.Gpart.p, w.nchunks, ld, w.G.p);
    GLF_LAUNCH_CHECK(ctx);
    if (rows.dist) GLF_TRY(allreduce_f(ctx, w.G.p, (size_t)ld * ld));
    if (nblk > 0 && ld == 64)
        hipLaunchKernelGGL(k_resid_mfma<64>, dim3(nblk), dim3(256), 0, st, X + off, w.AX.p + off, w.G.p, nloc, m, w.partial.p);
    else if (nblk > 0 && ld == 32)
        hipLaunchKernelGGL(k_resid_mfma<32>, dim3(nblk), dim3(256), 0, st, X + off, w.AX.p + off, w.G.p, nloc, m, w.partial.p);
    else if (nblk > 0)
        hipLaunchKernelGGL(k_resid, dim3(nblk), dim3(256), 0, st, X + off, w.AX.p + off, w.G.p, nloc, ld, m, w.partial.p);
    if (rows.dist) {
        hipLaunchKernelGGL(k_sum_partials_nv<1>, dim3(1), dim3(256), 0, st, w.partial.p, nblk, ld, w.sums.p);
        GLF_LAUNCH_CHECK(ctx);
        GLF_TRY(allreduce_d(ctx, w.sums.p, ld));
        GLF_HIP(ctx, hipMemcpyAsync(w.h_sums, w.sums.p, sizeof(double) * ld, hipMemcpyDeviceToHost, st));
    } else {
        hipLaunchKernelGGL(k_sum_partials, dim3(1), dim3(256), 0, st, w.partial.p, nblk, (int)ld, w.h_sums);
    }
    GLF_LAUNCH_CHECK(ctx);
    GLF_HIP(ctx, hipStreamSynchronize(st));
    double ss = 0.0;
    for (unsigned c = 0; c < m; ++c) ss += ((volatile double *)w.h_sums)[c];
    *h_out = std::sqrt(ss);
    return GLF_OK;
}

// out[i] -= r[i]
__global__ void k_sub_inplace(float *__restrict__ out, const float *__restrict__ r, size_t n)
{
    const size_t e = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (e < n) out[e] -= r[e];
}

// =====================================================================================
// InversePowerIteration, hpc/inverse_power_it.c:86-252
// =====================================================================================
//
// The residual of every outer iteration needs A X_new (hpc/inverse_power_it.c:49-80, :180). X_new = Y Tn with
// Y the PCG solution of A Y = B (B = the previous X) and Tn the triangular map of the Gram-Schmidt, and the PCG
// carries R = B - A Y, so A X_new = (B - R) Tn without another 4 p^2-byte sweep over L_A. Used whenever the
// Gram-Schmidt took the Gram-matrix form (or was skipped by opti_gs); GLF_RESIDUAL=sweep forces the explicit sweep.

int inverse_power_iteration(glf_ctx *ctx, const float *A, int64_t lda, unsigned p, unsigned m, unsigned ld,
                            const double *h_X0, int opti_gs, double epsilon, double inner_rtol, int max_outer,
                            float *d_eigvecs, double *h_eigvals, glf_eig_stats *stats, const MatShard *shard,
                            const float *d_dinv, const float *d_X0_block)
{
    if (m == 0 || m > p || !valid_ld(ld) || m > ld)
        return set_error(ctx, GLF_ERR_INVALID, "inverse_power_iteration: m=%u ld=%u p=%u (m <= 256 supported)", m, ld, p);
    if (opti_gs < 1) opti_gs = 1; // hpc/image_processing.c:128-140
    const unsigned p32 = (unsigned)round_up(p, VEC_PAD);
    const size_t n = (size_t)vec_rows(p, shard, ctx->comm.size) * ld; // >= p32 * ld
    const size_t n_out = (size_t)p32 * ld;                           // what the caller's d_eigvecs holds
    hipStream_t st = ctx->stream;

    DevBuf<float> X, Xb;
    GLF_TRY(X.alloc(ctx, n));
    GLF_TRY(Xb.alloc(ctx, n));
    // X0: a ready device block [p32][ld], or host double [m][p] (vector after vector) -> device float [p32][ld]
    if (d_X0_block) {
        GLF_HIP(ctx, hipMemsetAsync(X.p, 0, sizeof(float) * n, st));
        GLF_HIP(ctx, hipMemcpyAsync(X.p, d_X0_block, sizeof(float) * n_out, hipMemcpyDeviceToDevice, st));
    } else {
        std::vector<double> own;
        if (!h_X0) {
            own.resize((size_t)m * p);
            glf_random_vectors(own.data(), p, m, 1);
            h_X0 = own.data();
        }
        std::vector<float> h(n, 0.f);
        for (unsigned j = 0; j < m; ++j)
            for (unsigned i = 0; i < p; ++i) h[(size_t)i * ld + j] = (float)h_X0[(size_t)j * p + i];
        GLF_HIP(ctx, hipMemcpyAsync(X.p, h.data(), sizeof(float) * n, hipMemcpyHostToDevice, st));
        GLF_HIP(ctx, hipStreamSynchronize(st));
    }
    GsWork gs;
    GLF_TRY(gs.init(ctx, p, ld));
    CgWork cg;
    GLF_TRY(cg.init(ctx, p, ld, shard));
    ResWork rs;
    GLF_TRY(rs.init(ctx, p, ld, shard));
    // Sharded solve (see Rows above): every vector kernel below touches this rank's rows [rows.r0, rows.r1) only.
    const Rows rows = rows_of(p, shard);
    const unsigned nloc = rows.n();
    const size_t off = (size_t)rows.r0 * ld, nloc_ld = (size_t)nloc * ld;
    if (d_dinv) // Jacobi preconditioner supplied by the caller (a sharded A does not hold the whole diagonal)
        GLF_HIP(ctx, hipMemcpyAsync(cg.dinv.p, d_dinv, sizeof(float) * p, hipMemcpyDeviceToDevice, st));
    else if (shard && shard->rows_per_rank)
        return set_error(ctx, GLF_ERR_INVALID, "sharded inverse iteration needs the diagonal (d_dinv)");
    else
        hipLaunchKernelGGL(k_diag_inv, dim3((p + 255) / 256), dim3(256), 0, st, A, lda, p, cg.dinv.p);
    GLF_LAUNCH_CHECK(ctx);

    GLF_TRY(mv_collect(ctx));
    const int mv_count0 = ctx->mv_count, narrow0 = ctx->narrow_sweeps;
    const double mv_ms0 = ctx->mv_ms, mv_bytes0 = ctx->mv_bytes;
    bool replicated = false;
    GLF_TRY(orthonormalise_dev(ctx, gs, X.p, p, m, ld, rows, shard, &replicated)); // :95
    // The reference leaves X_k_before_orth unset when the loop never runs (:97-101); define it.
    GLF_HIP(ctx, hipMemcpyAsync(Xb.p, X.p, sizeof(float) * n, hipMemcpyDeviceToDevice, st));
    double r_norm = 0.0;
    GLF_TRY(residual_dev(ctx, rs, A, lda, p, X.p, m, ld, &r_norm)); // :159
    const bool verbose = ctx->tune.verbose; // the reference logs every outer iteration (:164-181)
    if (verbose) fprintf(stderr, "[glf rank %d] initial residual %.9g\n", ctx->comm.rank, r_norm);
    int it = 0, inner_total = 0, rc = GLF_OK;
    const bool derive_ax = !ctx->tune.residual_sweep;
    while (r_norm > epsilon) { // :161
        if (it >= max_outer) {
            rc = set_error(ctx, GLF_ERR_NOCONV, "inverse iteration: residual %g > %g after %d outer iterations", r_norm,
                           epsilon, it);
            break;
        }
        ++it;
        int inner = 0;
        if (derive_ax && nloc) GLF_HIP(ctx, hipMemcpyAsync(rs.AX.p + off, X.p + off, sizeof(float) * nloc_ld, hipMemcpyDeviceToDevice, st)); // B
        GLF_TRY(block_pcg_work(ctx, cg, A, lda, p, X.p, m, ld, inner_rtol, 10 * (int)p + 100, &inner)); // :165-168
        inner_total += inner;
        if (nloc) GLF_HIP(ctx, hipMemcpyAsync(Xb.p + off, X.p + off, sizeof(float) * nloc_ld, hipMemcpyDeviceToDevice, st)); // CopyVecs :171
        bool ax_ready = derive_ax;
        if (derive_ax && nloc) // A Y = B - R
            hipLaunchKernelGGL(k_sub_inplace, dim3((unsigned)ceil_div((int64_t)nloc_ld, 256)), dim3(256), 0, st, rs.AX.p + off, cg.R.p + off, nloc_ld);
        if (it % opti_gs == 0) {
            GLF_TRY(orthonormalise_dev(ctx, gs, X.p, p, m, ld, rows, shard, &replicated)); // :174-177
            if (ax_ready && gs.last_fused) { // A X_new = (A Y) Tn
                if (nloc) launch_gsf_apply(st, rs.AX.p + off, nloc, ld, m, gs.fused.Tn.p, gs.fused.flag.p);
            } else {
                ax_ready = false; // column-by-column sweep: no triangular map at hand
            }
        }
        GLF_LAUNCH_CHECK(ctx);
        GLF_TRY(residual_dev(ctx, rs, A, lda, p, X.p, m, ld, &r_norm, ax_ready)); // :180
        if (verbose)
            fprintf(stderr, "[glf rank %d] outer iteration %d: %d block-CG steps, residual %.9g\n", ctx->comm.rank, it, inner, r_norm);
    }
    if (opti_gs != 1 && (it % opti_gs) != 0)
        GLF_TRY(orthonormalise_dev(ctx, gs, X.p, p, m, ld, rows, shard, &replicated)); // :183-186

    if (h_eigvals) { // eigenvalues = 1 / norms, :204
        std::vector<double> nr(m);
        GLF_HIP(ctx, hipMemcpyAsync(nr.data(), gs.norms.p, sizeof(double) * m, hipMemcpyDeviceToHost, st));
        GLF_HIP(ctx, hipStreamSynchronize(st));
        for (unsigned j = 0; j < m; ++j) h_eigvals[j] = 1.0 / nr[j];
    }
    if (d_eigvecs) { // NormaliseVecs(X_k_before_orth), :230
        GLF_TRY(normalise_dev(ctx, Xb.p, p, m, ld, gs.partial.p, gs.norms.p, rows, cg.sums.p));
        if (rows.dist) GLF_TRY(allgather_rows(ctx, Xb.p, shard, ld)); // every rank extends with the whole Phi_A (hpc/nystroem.c:41)
        GLF_HIP(ctx, hipMemcpyAsync(d_eigvecs, Xb.p, sizeof(float) * n_out, hipMemcpyDeviceToDevice, st));
    }
    GLF_HIP(ctx, hipStreamSynchronize(st));
    GLF_TRY(mv_collect(ctx));
    if (stats) {
        stats->outer_its = it;
        stats->inner_its_total = inner_total;
        stats->residual = r_norm;
        stats->matvecs = ctx->mv_count - mv_count0;
        stats->matvec_ms = (float)(ctx->mv_ms - mv_ms0);
        stats->matvec_bytes = ctx->mv_bytes - mv_bytes0;
        stats->narrow_sweeps = ctx->narrow_sweeps - narrow0;
    }
    return rc;
}

// =====================================================================================
// More than 256 eigenpairs (the reference's default is m = p - 1, hpc/image_processing.c:96-108)
// =====================================================================================
// The m vectors are held as ceil(m / 256) PANELS of 256 columns, panel q a [rows][256] block of its own (the last one
// zero-padded), so that every kernel above runs unchanged on one panel. Columns are independent in the inner solves, the
// operator applications and the final normalisation; the two places where vectors meet are
//   * classical Gram-Schmidt: for panel q the coefficients <v_k, u_j> against every EARLIER panel come from the original
//     v_k (C_jq = X_j^T V_q for all j < q first, then V_q -= sum_j X_j C_jq -- exactly the classical sweep's terms,
//     hpc/gram_schmidt.c:47-57, as small f64 GEMMs), then the panel is orthonormalised within itself (Gram-matrix form);
//   * the residual || A X - X (X^T A X) ||_F: R_j = A X_j - sum_i X_i (X_i^T A X_j), panel pair by panel pair.
// This path is about coverage, not speed (O(p m^2) f64 work on the vector pipe); it is single-rank (or replicated).
constexpr unsigned PANEL = 256;

struct PanelWork {
    DevBuf<double> Gpart, C;   // Gram chunks; the coefficient blocks C_jq of one panel q (npan x 256 x 256)
    DevBuf<float> R;           // one residual panel
    DevBuf<double> partial, sums;
    int nchunks = 0;
    int init(glf_ctx *ctx, unsigned p, unsigned npan)
    {
        nchunks = (int)ceil_div(p, GSF_ROWS);
        GLF_TRY(Gpart.alloc(ctx, (size_t)nchunks * PANEL * PANEL));
        GLF_TRY(C.alloc(ctx, (size_t)npan * PANEL * PANEL));
        GLF_TRY(R.alloc(ctx, (size_t)round_up(p, VEC_PAD) * PANEL));
        GLF_TRY(partial.alloc(ctx, (size_t)ceil_div(p, RED_ROWS) * PANEL));
        GLF_TRY(sums.alloc(ctx, PANEL));
        return GLF_OK;
    }
};

// C = X^T Y in f64 ([256][256], all entries), X and Y two panels of n rows
static int panel_gram(glf_ctx *ctx, PanelWork &w, const float *X, const float *Y, unsigned n, double *C)
{
    const int mb = PANEL / 64;
    hipLaunchKernelGGL((k_gsf_gram<64>), dim3(w.nchunks, mb * mb), dim3(256), 0, ctx->stream, X, n, PANEL, w.Gpart.p, Y);
    hipLaunchKernelGGL(k_gsf_sum, dim3((PANEL * PANEL + 63) / 64), dim3(256), 0, ctx->stream, w.Gpart.p, w.nchunks, PANEL, 64, C, 1);
    GLF_LAUNCH_CHECK(ctx);
    return GLF_OK;
}

static inline unsigned panel_cols(unsigned m, unsigned q) { return std::min(PANEL, m - q * PANEL); }

// OrthonormaliseVecs over all panels; norms: device double [npan * 256]
static int panels_orthonormalise(glf_ctx *ctx, GsWork &gs, PanelWork &w, float *X, size_t pstride, unsigned p, unsigned m,
                                 double *d_norms)
{
    const unsigned npan = (unsigned)ceil_div(m, PANEL);
    hipStream_t st = ctx->stream;
    for (unsigned q = 0; q < npan; ++q) {
        float *Vq = X + q * pstride;
        for (unsigned j = 0; j < q; ++j) // <v_k, u_j> from the ORIGINAL v_k (classical Gram-Schmidt), all earlier panels first
            GLF_TRY(panel_gram(ctx, w, X + j * pstride, Vq, p, w.C.p + (size_t)j * PANEL * PANEL));
        for (unsigned j = 0; j < q; ++j)
            hipLaunchKernelGGL(k_gsf_sub, dim3((unsigned)ceil_div(p, GSF_APPLY_TILE / PANEL)), dim3(256), 0, st, Vq, X + j * pstride, p, PANEL,
                               PANEL, w.C.p + (size_t)j * PANEL * PANEL);
        GLF_LAUNCH_CHECK(ctx);
        GLF_TRY(orthonormalise_dev(ctx, gs, Vq, p, panel_cols(m, q), PANEL));
        GLF_HIP(ctx, hipMemcpyAsync(d_norms + (size_t)q * PANEL, gs.norms.p, sizeof(double) * PANEL, hipMemcpyDeviceToDevice, st));
    }
    return GLF_OK;
}

// || A X - X (X^T A X) ||_F over all panels (AX: npan panels of scratch)
static int panels_residual(glf_ctx *ctx, PanelWork &w, const float *A, int64_t lda, unsigned p, float *X, float *AX, size_t pstride,
                           unsigned m, const MatShard *shard, double *h_out)
{
    const unsigned npan = (unsigned)ceil_div(m, PANEL);
    hipStream_t st = ctx->stream;
    const int nblk = (int)ceil_div(p, RED_ROWS);
    for (unsigned j = 0; j < npan; ++j) GLF_TRY(block_matvec(ctx, A, lda, p, X + j * pstride, AX + j * pstride, PANEL, shard));
    double ss = 0.0;
    std::vector<double> h(PANEL);
    for (unsigned j = 0; j < npan; ++j) {
        GLF_HIP(ctx, hipMemcpyAsync(w.R.p, AX + j * pstride, sizeof(float) * (size_t)p * PANEL, hipMemcpyDeviceToDevice, st));
        for (unsigned i = 0; i < npan; ++i) {
            GLF_TRY(panel_gram(ctx, w, X + i * pstride, AX + j * pstride, p, w.C.p));
            hipLaunchKernelGGL(k_gsf_sub, dim3((unsigned)ceil_div(p, GSF_APPLY_TILE / PANEL)), dim3(256), 0, st, w.R.p, X + i * pstride, p,
                               PANEL, panel_cols(m, i), w.C.p);
        }
        hipLaunchKernelGGL(k_col_sumsq, dim3(nblk), dim3(256), 0, st, w.R.p, p, PANEL, w.partial.p);
        hipLaunchKernelGGL(k_sum_partials_nv<1>, dim3(1), dim3(256), 0, st, w.partial.p, nblk, PANEL, w.sums.p);
        GLF_LAUNCH_CHECK(ctx);
        GLF_HIP(ctx, hipMemcpyAsync(h.data(), w.sums.p, sizeof(double) * PANEL, hipMemcpyDeviceToHost, st));
        GLF_HIP(ctx, hipStreamSynchronize(st));
        for (unsigned c = 0; c < panel_cols(m, j); ++c) ss += h[c];
    }
    *h_out = std::sqrt(ss);
    return GLF_OK;
}

// InversePowerIteration for m > 256. d_eigvecs: npan panels of [round_up(p, 64)][256] (panel-major); X0 = the seeded start
// block (hpc/inverse_power_it.c:12-47), host double [m][p] or nullptr for glf_random_vectors(seed).
int inverse_power_iteration_panels(glf_ctx *ctx, const float *A, int64_t lda, unsigned p, unsigned m, const double *h_X0,
                                   unsigned long long seed, int opti_gs, double epsilon, double inner_rtol, int max_outer,
                                   float *d_eigvecs, double *h_eigvals, glf_eig_stats *stats, const MatShard *shard,
                                   const float *d_dinv)
{
    if (m <= PANEL || m > p) return set_error(ctx, GLF_ERR_INVALID, "inverse_power_iteration_panels: m=%u p=%u", m, p);
    if (shard && shard->rows_per_rank) return set_error(ctx, GLF_ERR_UNSUPPORTED, "more than 256 eigenpairs: the eigen-solve is not row-sharded");
    if (opti_gs < 1) opti_gs = 1;
    const unsigned npan = (unsigned)ceil_div(m, PANEL), p32 = (unsigned)round_up(p, VEC_PAD);
    const size_t pstride = (size_t)p32 * PANEL, total = pstride * npan;
    hipStream_t st = ctx->stream;
    DevBuf<float> X, Xb, AX;
    DevBuf<double> norms;
    GLF_TRY(X.alloc(ctx, total));
    GLF_TRY(Xb.alloc(ctx, total));
    GLF_TRY(AX.alloc(ctx, total));
    GLF_TRY(norms.alloc(ctx, (size_t)npan * PANEL));
    {
        std::vector<double> own;
        if (!h_X0) {
            own.resize((size_t)m * p);
            glf_random_vectors(own.data(), p, m, seed);
            h_X0 = own.data();
        }
        std::vector<float> h(total, 0.f);
        for (unsigned j = 0; j < m; ++j)
            for (unsigned i = 0; i < p; ++i) h[(j / PANEL) * pstride + (size_t)i * PANEL + j % PANEL] = (float)h_X0[(size_t)j * p + i];
        GLF_HIP(ctx, hipMemcpyAsync(X.p, h.data(), sizeof(float) * total, hipMemcpyHostToDevice, st));
        GLF_HIP(ctx, hipStreamSynchronize(st));
    }
    GLF_HIP(ctx, hipMemsetAsync(AX.p, 0, sizeof(float) * total, st));
    GsWork gs;
    GLF_TRY(gs.init(ctx, p, PANEL));
    CgWork cg;
    GLF_TRY(cg.init(ctx, p, PANEL, shard));
    PanelWork pw;
    GLF_TRY(pw.init(ctx, p, npan));
    if (d_dinv) GLF_HIP(ctx, hipMemcpyAsync(cg.dinv.p, d_dinv, sizeof(float) * p, hipMemcpyDeviceToDevice, st));
    else hipLaunchKernelGGL(k_diag_inv, dim3((p + 255) / 256), dim3(256), 0, st, A, lda, p, cg.dinv.p);
    GLF_LAUNCH_CHECK(ctx);
    GLF_TRY(mv_collect(ctx));
    const int mv_count0 = ctx->mv_count, narrow0 = ctx->narrow_sweeps;
    const double mv_ms0 = ctx->mv_ms, mv_bytes0 = ctx->mv_bytes;

    GLF_TRY(panels_orthonormalise(ctx, gs, pw, X.p, pstride, p, m, norms.p)); // :95
    GLF_HIP(ctx, hipMemcpyAsync(Xb.p, X.p, sizeof(float) * total, hipMemcpyDeviceToDevice, st));
    double r_norm = 0.0;
    GLF_TRY(panels_residual(ctx, pw, A, lda, p, X.p, AX.p, pstride, m, shard, &r_norm)); // :159
    const bool verbose = ctx->tune.verbose;
    if (verbose) fprintf(stderr, "[glf] %u panels; initial residual %.9g\n", npan, r_norm);
    int it = 0, inner_total = 0, rc = GLF_OK;
    while (r_norm > epsilon) { // :161
        if (it >= max_outer) {
            rc = set_error(ctx, GLF_ERR_NOCONV, "inverse iteration: residual %g > %g after %d outer iterations", r_norm, epsilon, it);
            break;
        }
        ++it;
        int inner_max = 0;
        for (unsigned q = 0; q < npan; ++q) { // :165-168: the solves are independent column by column
            int inner = 0;
            GLF_TRY(block_pcg_work(ctx, cg, A, lda, p, X.p + q * pstride, panel_cols(m, q), PANEL, inner_rtol, 10 * (int)p + 100, &inner));
            inner_max = std::max(inner_max, inner);
        }
        inner_total += inner_max; // block-Krylov steps of the slowest panel (what the single-panel count means)
        GLF_HIP(ctx, hipMemcpyAsync(Xb.p, X.p, sizeof(float) * total, hipMemcpyDeviceToDevice, st)); // CopyVecs :171
        if (it % opti_gs == 0) GLF_TRY(panels_orthonormalise(ctx, gs, pw, X.p, pstride, p, m, norms.p)); // :174-177
        GLF_TRY(panels_residual(ctx, pw, A, lda, p, X.p, AX.p, pstride, m, shard, &r_norm)); // :180
        if (verbose) fprintf(stderr, "[glf] outer iteration %d: %d block-CG steps, residual %.9g\n", it, inner_max, r_norm);
    }
    if (opti_gs != 1 && (it % opti_gs) != 0) GLF_TRY(panels_orthonormalise(ctx, gs, pw, X.p, pstride, p, m, norms.p)); // :183-186
    if (h_eigvals) { // eigenvalues = 1 / norms, :204
        std::vector<double> nr((size_t)npan * PANEL);
        GLF_HIP(ctx, hipMemcpyAsync(nr.data(), norms.p, sizeof(double) * nr.size(), hipMemcpyDeviceToHost, st));
        GLF_HIP(ctx, hipStreamSynchronize(st));
        for (unsigned j = 0; j < m; ++j) h_eigvals[j] = 1.0 / nr[j];
    }
    if (d_eigvecs) { // NormaliseVecs(X_k_before_orth), :230
        for (unsigned q = 0; q < npan; ++q)
            GLF_TRY(normalise_dev(ctx, Xb.p + q * pstride, p, panel_cols(m, q), PANEL, gs.partial.p, gs.norms.p));
        GLF_HIP(ctx, hipMemcpyAsync(d_eigvecs, Xb.p, sizeof(float) * total, hipMemcpyDeviceToDevice, st));
    }
    GLF_HIP(ctx, hipStreamSynchronize(st));
    GLF_TRY(mv_collect(ctx));
    if (stats) {
        stats->outer_its = it;
        stats->inner_its_total = inner_total;
        stats->residual = r_norm;
        stats->matvecs = ctx->mv_count - mv_count0;
        stats->matvec_ms = (float)(ctx->mv_ms - mv_ms0);
        stats->matvec_bytes = ctx->mv_bytes - mv_bytes0;
        stats->narrow_sweeps = ctx->narrow_sweeps - narrow0;
    }
    return rc;
}

// OrthonormaliseVecs / NormaliseVecs on a panel-major block of more than 256 vectors (stage API)
int orthonormalise_panels(glf_ctx *ctx, float *X, unsigned n, unsigned m, double *h_norms, bool normalise_only)
{
    const unsigned npan = (unsigned)ceil_div(m, PANEL), n32 = (unsigned)round_up(n, VEC_PAD);
    const size_t pstride = (size_t)n32 * PANEL;
    GsWork gs;
    GLF_TRY(gs.init(ctx, n, PANEL));
    DevBuf<double> norms;
    GLF_TRY(norms.alloc(ctx, (size_t)npan * PANEL));
    if (normalise_only) {
        for (unsigned q = 0; q < npan; ++q) {
            GLF_TRY(normalise_dev(ctx, X + q * pstride, n, panel_cols(m, q), PANEL, gs.partial.p, gs.norms.p));
            GLF_HIP(ctx, hipMemcpyAsync(norms.p + (size_t)q * PANEL, gs.norms.p, sizeof(double) * PANEL, hipMemcpyDeviceToDevice, ctx->stream));
        }
    } else {
        PanelWork pw;
        GLF_TRY(pw.init(ctx, n, npan));
        GLF_TRY(panels_orthonormalise(ctx, gs, pw, X, pstride, n, m, norms.p));
        GLF_HIP(ctx, hipStreamSynchronize(ctx->stream));
    }
    if (h_norms) {
        std::vector<double> h((size_t)npan * PANEL);
        GLF_HIP(ctx, hipMemcpyAsync(h.data(), norms.p, sizeof(double) * h.size(), hipMemcpyDeviceToHost, ctx->stream));
        GLF_HIP(ctx, hipStreamSynchronize(ctx->stream));
        for (unsigned j = 0; j < m; ++j) h_norms[j] = h[j];
    }
    GLF_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return GLF_OK;
}

} // namespace glf
