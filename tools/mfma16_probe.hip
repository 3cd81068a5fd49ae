// tools/mfma16_probe.hip -- sustained rate of v_mfma_f32_32x32x16_f16 on gfx950 in the shape the row pass uses
// (8 accumulators per wave, 24 MFMAs per k-step), alone and with the k-step's 8 ds_read_b128 in front of them.
// Build: hipcc -O3 --offload-arch=gfx950 tools/mfma16_probe.hip -o tools/mfma16_probe
#include <hip/hip_runtime.h>
#include <cstdio>
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int MODE, int NW> // 0: registers only, 1: B fragments from LDS every k-step, 2: plus 4 A fragments from global (L2)
__global__ __launch_bounds__(NW * 64) void k_probe(const f16x8 *__restrict__ gsrc, float *__restrict__ out, int iters)
{
    extern __shared__ f16x8 lds[];
    const int lane = threadIdx.x & 63;
    for (int i = threadIdx.x; i < 19 * 8 * 64; i += NW * 64) lds[i] = gsrc[i & 4095];
    __syncthreads();
    f32x16 acc[8];
    for (int j = 0; j < 8; ++j)
        for (int i = 0; i < 16; ++i) acc[j][i] = 0.f;
    f16x8 a[4], b[8];
    for (int j = 0; j < 4; ++j) a[j] = gsrc[j * 64 + lane];
    for (int j = 0; j < 8; ++j) b[j] = lds[j * 64 + lane];
    for (int it = 0; it < iters; ++it) {
        const int ks = it % 19;
        if (MODE >= 1)
            for (int j = 0; j < 8; ++j) b[j] = lds[(ks * 8 + j) * 64 + lane];
        if (MODE >= 2)
            for (int j = 0; j < 4; ++j) a[j] = gsrc[((size_t)(blockIdx.x & 15) * 76 + ks * 4 + j) * 64 + lane];
#pragma unroll
        for (int s = 0; s < 3; ++s)
#pragma unroll
            for (int m = 0; m < 2; ++m)
#pragma unroll
                for (int n = 0; n < 4; ++n)
                    acc[m * 4 + n] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[2 * m + (s == 2)], b[2 * n + (s == 1)], acc[m * 4 + n], 0, 0, 0);
    }
    float s = 0.f;
    for (int j = 0; j < 8; ++j)
        for (int i = 0; i < 16; ++i) s += acc[j][i];
    if (s == 12345.f) out[threadIdx.x] = s;
}

template <int MODE, int NW>
static void run(const char *name, const f16x8 *g, float *out, int wgs)
{
    const int iters = 19 * 64;
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    const size_t lds = 19 * 8 * 1024;
    hipFuncSetAttribute(reinterpret_cast<const void *>(k_probe<MODE, NW>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    for (int rep = 0; rep < 3; ++rep) {
        hipEventRecord(e0);
        hipLaunchKernelGGL((k_probe<MODE, NW>), dim3(wgs), dim3(NW * 64), lds, 0, g, out, iters);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
    }
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    const double fl = (double)wgs * NW * iters * 24.0 * 32 * 32 * 16 * 2;
    printf("%-44s wgs %5d  %8.3f ms  %8.1f TFLOP/s\n", name, wgs, ms, fl / ms / 1e9);
}

int main()
{
    f16x8 *g;
    float *out;
    hipMalloc(&g, 16 * 76 * 64 * 16 + (1 << 20));
    { // pseudo-random finite f16 operands (all-zero operands toggle nothing and flatter the power-limited rate)
        const size_t nh = (16 * 76 * 64 * 16 + (1 << 20)) / 2;
        unsigned short *h = (unsigned short *)malloc(nh * 2);
        unsigned x = 12345u;
        for (size_t i = 0; i < nh; ++i) {
            x = x * 1664525u + 1013904223u;
            h[i] = (unsigned short)(((x >> 16) & 0x83FFu) | 0x3400u); // +-[0.25, 0.5)
        }
        hipMemcpy(g, h, nh * 2, hipMemcpyHostToDevice);
        free(h);
    }
    hipMalloc(&out, 4096);
    { // sustained rate: ~150 ms of back-to-back launches (power management reacts within milliseconds)
        const int iters = 19 * 64, wgs = 2048, reps = 24;
        const size_t lds = 19 * 8 * 1024;
        hipEvent_t e0, e1;
        hipEventCreate(&e0);
        hipEventCreate(&e1);
        for (int mode = 0; mode < 2; ++mode) {
            hipEventRecord(e0);
            for (int r = 0; r < reps; ++r) {
                if (mode == 0) hipLaunchKernelGGL((k_probe<0, 8>), dim3(wgs), dim3(512), lds, 0, g, out, iters);
                else hipLaunchKernelGGL((k_probe<1, 8>), dim3(wgs), dim3(512), lds, 0, g, out, iters);
            }
            hipEventRecord(e1);
            hipEventSynchronize(e1);
            float ms;
            hipEventElapsedTime(&ms, e0, e1);
            const double fl = (double)reps * wgs * 8 * iters * 24.0 * 32 * 32 * 16 * 2;
            printf("sustained, %s: %d launches %8.2f ms  %8.1f TFLOP/s\n", mode ? "B from LDS per k-step" : "registers only", reps, ms, fl / ms / 1e9);
        }
    }
    for (int wgs : {256, 2048}) {
        run<0, 4>("registers only, 4 waves/CU", g, out, wgs);
        run<0, 8>("registers only, 8 waves/CU", g, out, wgs);
        run<1, 8>("B from LDS per k-step, 8 waves/CU", g, out, wgs);
        run<2, 8>("B from LDS + A from L2 per k-step, 8 waves/CU", g, out, wgs);
        run<2, 4>("B from LDS + A from L2 per k-step, 4 waves/CU", g, out, wgs);
    }
    return 0;
}
