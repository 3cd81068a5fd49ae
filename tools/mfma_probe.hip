// mfma_probe.hip -- micro-benchmarks behind the Nystroem kernel design (f32 MFMA 32x32x2 on gfx950).
// Build: hipcc -O3 --offload-arch=gfx950 tools/mfma_probe.hip -o /tmp/mfma_probe ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));

// V0: pure MFMA, NACC accumulators, operands fixed in registers
template <int NACC>
__global__ __launch_bounds__(256) void k_pure(float *out, int iters, float a0, float b0)
{
    f32x16 acc[NACC];
    for (int i = 0; i < NACC; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    float a = a0 + threadIdx.x, b = b0;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[i], 0, 0, 0);
    }
    float s = 0.f;
    for (int i = 0; i < NACC; ++i) for (int r = 0; r < 16; ++r) s += acc[i][r];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

// V1: MFMA + kernel generation in VALU (9 ops + exp per A value), sample/psi values from registers (no LDS)
template <int PB, int MB>
__global__ __launch_bounds__(256) void k_gen(float *out, int iters, float s_loc, float s_val)
{
    f32x16 acc[PB][MB];
    for (int b = 0; b < PB; ++b) for (int j = 0; j < MB; ++j) for (int r = 0; r < 16; ++r) acc[b][j][r] = 0.f;
    float pr[PB], pc[PB], pv[PB];
    for (int b = 0; b < PB; ++b) { pr[b] = threadIdx.x + b; pc[b] = threadIdx.x * 3 + b; pv[b] = (threadIdx.x * 7 + b) & 255; }
    float sx = 1.f, sy = 2.f, sz = 3.f, bf = 0.5f;
    for (int it = 0; it < iters; ++it) {
        float a[PB];
#pragma unroll
        for (int b = 0; b < PB; ++b) {
            const float dr = pr[b] - sx, dc = pc[b] - sy, dv = pv[b] - sz;
            const float q = fmaf(dc, dc, dr * dr);
            a[b] = __builtin_amdgcn_exp2f(-fmaf(dv * dv, s_val, q * s_loc));
        }
#pragma unroll
        for (int j = 0; j < MB; ++j)
#pragma unroll
            for (int b = 0; b < PB; ++b) acc[b][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[b], bf + j, acc[b][j], 0, 0, 0);
        sx += 1.f; sy += 0.5f; sz += 0.25f;
    }
    float s = 0.f;
    for (int b = 0; b < PB; ++b) for (int j = 0; j < MB; ++j) for (int r = 0; r < 16; ++r) s += acc[b][j][r];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

// V1p: as V1 with the two pixel blocks' generation packed into v_pk_* (float2) ops
typedef float f32x2 __attribute__((ext_vector_type(2)));
template <int MB>
__global__ __launch_bounds__(256) void k_gen_pk(float *out, int iters, float s_loc, float s_val)
{
    f32x16 acc[2][MB];
    for (int b = 0; b < 2; ++b) for (int j = 0; j < MB; ++j) for (int r = 0; r < 16; ++r) acc[b][j][r] = 0.f;
    f32x2 pr, pc, pv;
    for (int b = 0; b < 2; ++b) { pr[b] = threadIdx.x + b; pc[b] = threadIdx.x * 3 + b; pv[b] = (threadIdx.x * 7 + b) & 255; }
    float sx = 1.f, sy = 2.f, sz = 3.f, bf = 0.5f;
    for (int it = 0; it < iters; ++it) {
        const f32x2 dr = pr - sx, dc = pc - sy, dv = pv - sz;
        const f32x2 q = dc * dc + dr * dr;
        const f32x2 t = (dv * dv) * s_val + q * s_loc;
        float a[2] = {__builtin_amdgcn_exp2f(-t[0]), __builtin_amdgcn_exp2f(-t[1])};
#pragma unroll
        for (int j = 0; j < MB; ++j)
#pragma unroll
            for (int b = 0; b < 2; ++b) acc[b][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[b], bf + j, acc[b][j], 0, 0, 0);
        sx += 1.f; sy += 0.5f; sz += 0.25f;
    }
    float s = 0.f;
    for (int b = 0; b < 2; ++b) for (int j = 0; j < MB; ++j) for (int r = 0; r < 16; ++r) s += acc[b][j][r];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

// V1v: pure VALU generation, no MFMA (how many cycles does one A value cost?)
template <int PK>
__global__ __launch_bounds__(256) void k_valu_only(float *out, int iters, float s_loc, float s_val)
{
    f32x2 pr, pc, pv, accv = {0.f, 0.f};
    for (int b = 0; b < 2; ++b) { pr[b] = threadIdx.x + b; pc[b] = threadIdx.x * 3 + b; pv[b] = (threadIdx.x * 7 + b) & 255; }
    float sx = 1.f, sy = 2.f, sz = 3.f;
    for (int it = 0; it < iters; ++it) {
        if (PK) {
            const f32x2 dr = pr - sx, dc = pc - sy, dv = pv - sz;
            const f32x2 q = dc * dc + dr * dr;
            const f32x2 t = (dv * dv) * s_val + q * s_loc;
            accv[0] += __builtin_amdgcn_exp2f(-t[0]);
            accv[1] += __builtin_amdgcn_exp2f(-t[1]);
        } else {
            for (int b = 0; b < 2; ++b) {
                const float dr = pr[b] - sx, dc = pc[b] - sy, dv = pv[b] - sz;
                const float q = fmaf(dc, dc, dr * dr);
                accv[b] += __builtin_amdgcn_exp2f(-fmaf(dv * dv, s_val, q * s_loc));
            }
        }
        sx += 1.f; sy += 0.5f; sz += 0.25f;
    }
    out[blockIdx.x * 256 + threadIdx.x] = accv[0] + accv[1];
}

// V3: generation in VALU, contraction on the f16 matrix pipe with split operands
// (K = hi + lo in f16, Psi = hi + lo in f16; products hi*hi + hi*lo + lo*hi, f32 accumulate).
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef __fp16 f16x2 __attribute__((ext_vector_type(2)));
template <int PB, int MB, int NPROD>
__global__ __launch_bounds__(256) void k_gen_f16(float *out, int iters, float s_loc, float s_val)
{
    f32x16 acc[PB][MB];
    for (int b = 0; b < PB; ++b) for (int j = 0; j < MB; ++j) for (int r = 0; r < 16; ++r) acc[b][j][r] = 0.f;
    float pr[PB], pc[PB], pv[PB];
    for (int b = 0; b < PB; ++b) { pr[b] = threadIdx.x + b; pc[b] = threadIdx.x * 3 + b; pv[b] = (threadIdx.x * 7 + b) & 255; }
    float sx = 1.f, sy = 2.f, sz = 3.f;
    f16x8 bh[MB], bl[MB];
    for (int j = 0; j < MB; ++j) for (int e = 0; e < 8; ++e) { bh[j][e] = (_Float16)(0.5f + j + e); bl[j][e] = (_Float16)(0.001f * e); }
    for (int it = 0; it < iters; ++it) {
        f16x8 ah[PB], al[PB];
#pragma unroll
        for (int b = 0; b < PB; ++b) {
#pragma unroll
            for (int e = 0; e < 8; e += 2) {
                float y[2];
#pragma unroll
                for (int u = 0; u < 2; ++u) {
                    const float dr = pr[b] - (sx + e + u), dc = pc[b] - (sy + e + u), dv = pv[b] - (sz + e + u);
                    const float q = fmaf(dc, dc, dr * dr);
                    y[u] = __builtin_amdgcn_exp2f(15.f - fmaf(dv * dv, s_val, q * s_loc));
                }
                const f16x2 h = __builtin_amdgcn_cvt_pkrtz(y[0], y[1]);
                const float r0 = y[0] - (float)h[0], r1 = y[1] - (float)h[1];
                const f16x2 l = __builtin_amdgcn_cvt_pkrtz(r0, r1);
                ah[b][e] = (_Float16)h[0]; ah[b][e + 1] = (_Float16)h[1];
                al[b][e] = (_Float16)l[0]; al[b][e + 1] = (_Float16)l[1];
            }
        }
#pragma unroll
        for (int j = 0; j < MB; ++j)
#pragma unroll
            for (int b = 0; b < PB; ++b) {
                acc[b][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[b], bh[j], acc[b][j], 0, 0, 0);
                if (NPROD >= 2) acc[b][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[b], bl[j], acc[b][j], 0, 0, 0);
                if (NPROD >= 3) acc[b][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[b], bh[j], acc[b][j], 0, 0, 0);
            }
        sx += 16.f; sy += 0.5f; sz += 0.25f;
    }
    float s = 0.f;
    for (int b = 0; b < PB; ++b) for (int j = 0; j < MB; ++j) for (int r = 0; r < 16; ++r) s += acc[b][j][r];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

// V2: as V1 but sample records and psi come from LDS like the real kernel
template <int PB, int MB>
__global__ __launch_bounds__(256) void k_gen_lds(float *out, int iters, float s_loc, float s_val)
{
    constexpr int LD = MB * 32, KC = 64;
    __shared__ __attribute__((aligned(16))) float lds[KC * 4 + KC * LD];
    for (int e = threadIdx.x; e < KC * 4 + KC * LD; e += 256) lds[e] = (e % 97) * 0.01f;
    __syncthreads();
    const int lane = threadIdx.x & 63, half = lane >> 5, l31 = lane & 31;
    f32x16 acc[PB][MB];
    for (int b = 0; b < PB; ++b) for (int j = 0; j < MB; ++j) for (int r = 0; r < 16; ++r) acc[b][j][r] = 0.f;
    float pr[PB], pc[PB], pv[PB];
    for (int b = 0; b < PB; ++b) { pr[b] = threadIdx.x + b; pc[b] = threadIdx.x * 3 + b; pv[b] = (threadIdx.x * 7 + b) & 255; }
    const float4 *stb = reinterpret_cast<const float4 *>(lds) + half * (KC / 2);
    const float *psb = lds + KC * 4 + (half * (KC / 2)) * LD + l31;
    for (int it = 0; it < iters; ++it) {
#pragma unroll 4
        for (int kk = 0; kk < KC / 2; ++kk) {
            const float4 s = stb[kk];
            float a[PB];
#pragma unroll
            for (int b = 0; b < PB; ++b) {
                const float dr = pr[b] - s.x, dc = pc[b] - s.y, dv = pv[b] - s.z;
                const float q = fmaf(dc, dc, dr * dr);
                a[b] = __builtin_amdgcn_exp2f(-fmaf(dv * dv, s_val, q * s_loc));
            }
#pragma unroll
            for (int j = 0; j < MB; ++j) {
                const float bf = psb[kk * LD + 32 * j];
#pragma unroll
                for (int b = 0; b < PB; ++b) acc[b][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[b], bf, acc[b][j], 0, 0, 0);
            }
        }
    }
    float s = 0.f;
    for (int b = 0; b < PB; ++b) for (int j = 0; j < MB; ++j) for (int r = 0; r < 16; ++r) s += acc[b][j][r];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <typename F>
static double time_ms(F launch, int reps = 3)
{
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    launch();
    hipDeviceSynchronize();
    hipEventRecord(e0);
    for (int r = 0; r < reps; ++r) launch();
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0; hipEventElapsedTime(&ms, e0, e1);
    return ms / reps;
}

int main()
{
    float *out; hipMalloc(&out, 4096 * 256 * sizeof(float));
    const int WG = 2048; // 8 per CU
    const double fl_per_mfma = 4096.0;
    auto report = [&](const char *name, double ms, double mfmas_per_wave) {
        const double flops = mfmas_per_wave * fl_per_mfma * WG * 4;
        printf("%-28s %8.3f ms  %7.2f TFLOP/s  (%.1f%% of 157.3)\n", name, ms, flops / ms / 1e9, flops / ms / 1e9 / 157.3 * 100);
    };
    const int it = 20000;
    report("pure NACC=1", time_ms([&] { hipLaunchKernelGGL(k_pure<1>, dim3(WG), dim3(256), 0, 0, out, it * 4, 1.f, 2.f); }), it * 4.0);
    report("pure NACC=4", time_ms([&] { hipLaunchKernelGGL(k_pure<4>, dim3(WG), dim3(256), 0, 0, out, it, 1.f, 2.f); }), it * 4.0);
    report("pure NACC=8", time_ms([&] { hipLaunchKernelGGL(k_pure<8>, dim3(WG), dim3(256), 0, 0, out, it / 2, 1.f, 2.f); }), it * 4.0);
    report("gen PB2 MB2 (regs)", time_ms([&] { hipLaunchKernelGGL((k_gen<2, 2>), dim3(WG), dim3(256), 0, 0, out, it, 1e-3f, 2e-3f); }), it * 4.0);
    report("gen PB2 MB4 (regs)", time_ms([&] { hipLaunchKernelGGL((k_gen<2, 4>), dim3(WG), dim3(256), 0, 0, out, it / 2, 1e-3f, 2e-3f); }), it * 4.0);
    report("gen PB1 MB2 (regs)", time_ms([&] { hipLaunchKernelGGL((k_gen<1, 2>), dim3(WG), dim3(256), 0, 0, out, it * 2, 1e-3f, 2e-3f); }), it * 4.0);
    report("gen PK  MB2 (regs)", time_ms([&] { hipLaunchKernelGGL((k_gen_pk<2>), dim3(WG), dim3(256), 0, 0, out, it, 1e-3f, 2e-3f); }), it * 4.0);
    report("gen PK  MB4 (regs)", time_ms([&] { hipLaunchKernelGGL((k_gen_pk<4>), dim3(WG), dim3(256), 0, 0, out, it / 2, 1e-3f, 2e-3f); }), it * 4.0);
    {
        double ms0 = time_ms([&] { hipLaunchKernelGGL((k_valu_only<0>), dim3(WG), dim3(256), 0, 0, out, it, 1e-3f, 2e-3f); });
        double ms1 = time_ms([&] { hipLaunchKernelGGL((k_valu_only<1>), dim3(WG), dim3(256), 0, 0, out, it, 1e-3f, 2e-3f); });
        // cycles per generated A VGPR per SIMD: 8 waves per SIMD, 2 values per iteration
        const double cyc = 2.4e6 / (8.0 * it * 2.0);
        printf("valu-only scalar: %.3f ms = %.1f cyc per A value;  packed: %.3f ms = %.1f cyc per A value\n", ms0, ms0 * cyc, ms1, ms1 * cyc);
    }
    {
        // entries (K evaluations) per second: f32 path = PB values per iteration per lane, f16 path = 8*PB
        auto eps = [&](double ms, double vals_per_lane) { return vals_per_lane * 256.0 * WG / (ms * 1e-3); };
        double ms = time_ms([&] { hipLaunchKernelGGL((k_gen<2, 2>), dim3(WG), dim3(256), 0, 0, out, it, 1e-3f, 2e-3f); });
        printf("entries/s  f32 MFMA PB2 MB2        : %.3e\n", eps(ms, 2.0 * it));
        ms = time_ms([&] { hipLaunchKernelGGL((k_gen_f16<2, 2, 3>), dim3(WG), dim3(256), 0, 0, out, it / 8, 1e-3f, 2e-3f); });
        printf("entries/s  f16x2 split 3 products   : %.3e  (%.3f ms)\n", eps(ms, 16.0 * (it / 8)), ms);
        ms = time_ms([&] { hipLaunchKernelGGL((k_gen_f16<2, 2, 1>), dim3(WG), dim3(256), 0, 0, out, it / 8, 1e-3f, 2e-3f); });
        printf("entries/s  f16 single product       : %.3e  (%.3f ms)\n", eps(ms, 16.0 * (it / 8)), ms);
        ms = time_ms([&] { hipLaunchKernelGGL((k_gen_f16<1, 2, 3>), dim3(WG), dim3(256), 0, 0, out, it / 4, 1e-3f, 2e-3f); });
        printf("entries/s  f16x2 split PB1          : %.3e  (%.3f ms)\n", eps(ms, 8.0 * (it / 4)), ms);
        ms = time_ms([&] { hipLaunchKernelGGL((k_gen_f16<2, 4, 3>), dim3(WG), dim3(256), 0, 0, out, it / 8, 1e-3f, 2e-3f); });
        printf("entries/s  f16x2 split MB4 (m=128)  : %.3e  (%.3f ms)\n", eps(ms, 16.0 * (it / 8)), ms);
        ms = time_ms([&] { hipLaunchKernelGGL((k_valu_only<0>), dim3(WG), dim3(256), 0, 0, out, it, 1e-3f, 2e-3f); });
        printf("entries/s  VALU generation only     : %.3e\n", eps(ms, 2.0 * it));
    }
    report("gen+lds PB2 MB2", time_ms([&] { hipLaunchKernelGGL((k_gen_lds<2, 2>), dim3(WG), dim3(256), 0, 0, out, it / 32, 1e-3f, 2e-3f); }), (it / 32) * 32 * 4.0);
    report("gen+lds PB1 MB2", time_ms([&] { hipLaunchKernelGGL((k_gen_lds<1, 2>), dim3(WG), dim3(256), 0, 0, out, it / 16, 1e-3f, 2e-3f); }), (it / 16) * 32 * 2.0);
    report("gen+lds PB4 MB2", time_ms([&] { hipLaunchKernelGGL((k_gen_lds<4, 2>), dim3(WG), dim3(256), 0, 0, out, it / 64, 1e-3f, 2e-3f); }), (it / 64) * 32 * 8.0);
    report("gen+lds PB2 MB4", time_ms([&] { hipLaunchKernelGGL((k_gen_lds<2, 4>), dim3(WG), dim3(256), 0, 0, out, it / 64, 1e-3f, 2e-3f); }), (it / 64) * 32 * 8.0);
    // occupancy sweep for the pure loop: fewer workgroups per CU
    for (int wg : {256, 512, 1024}) {
        const double ms = time_ms([&] { hipLaunchKernelGGL(k_pure<4>, dim3(wg), dim3(256), 0, 0, out, it, 1.f, 2.f); });
        printf("pure NACC=4 grid %4d: %8.3f ms  %7.2f TFLOP/s\n", wg, ms, it * 4.0 * fl_per_mfma * wg * 4 / ms / 1e9);
    }
    hipFree(out);
    return 0;
}
