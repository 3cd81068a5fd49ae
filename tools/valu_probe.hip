// tools/valu_probe.hip -- issue cost (cycles per wave-instruction) of the vector instructions the f16 operand split can be
// built from on gfx950, alone on a SIMD and beside two more waves of the same stream: v_mul_f32, v_cvt_pk_f16_f32,
// v_fma_mix_f32 (f16 addend), v_fma_mixlo_f16 / v_fma_mixhi_f16 (f16 result halves), v_pk_mul_f32.
// Build: hipcc -O3 --offload-arch=gfx950 tools/valu_probe.hip -o tools/valu_probe
#include <hip/hip_runtime.h>
#include <cstdio>

typedef float f2 __attribute__((ext_vector_type(2)));
#define REP16(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7) X(8) X(9) X(10) X(11) X(12) X(13) X(14) X(15)

template <int OP>
__global__ void k_probe(float *out, long long *cycles, int iters)
{
    float r[16], a = threadIdx.x * 1e-3f + 1.f, b = 0.999f;
    for (int i = 0; i < 16; ++i) r[i] = i;
    f2 q[16], qa = {a, b}, qb = {b, a};
    double d[16], da = a, db = b;
    for (int i = 0; i < 16; ++i) d[i] = i;
    for (int i = 0; i < 16; ++i) q[i] = qa;
    const long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
#define MUL(i) asm volatile("v_mul_f32 %0, %1, %2" : "=v"(r[i]) : "v"(a), "v"(b));
#define CVT(i) asm volatile("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(r[i]) : "v"(a), "v"(b));
#define MIX(i) asm volatile("v_fma_mix_f32 %0, %1, %2, -%3 op_sel_hi:[0,0,1]" : "=v"(r[i]) : "v"(a), "v"(b), "v"(a));
#define MLO(i) asm volatile("v_fma_mixlo_f16 %0, %1, %2, 0" : "+v"(r[i]) : "v"(a), "v"(b));
#define MHI(i) asm volatile("v_fma_mixhi_f16 %0, %1, %2, 0" : "+v"(r[i]) : "v"(a), "v"(b));
#define MLH(i) asm volatile("v_fma_mixlo_f16 %0, %1, %2, 0\n\tv_fma_mixhi_f16 %0, %2, %1, 0" : "+v"(r[i]) : "v"(a), "v"(b));
#define PKM(i) asm volatile("v_pk_mul_f32 %0, %1, %2" : "=v"(q[i]) : "v"(qa), "v"(qb));
#define FMA(i) asm volatile("v_fma_f32 %0, %1, %2, %3" : "=v"(r[i]) : "v"(a), "v"(b), "v"(a));
#define F64(i) asm volatile("v_fma_f64 %0, %1, %2, %3" : "=v"(d[i]) : "v"(da), "v"(db), "v"(da));
#define C64(i) asm volatile("v_cvt_f64_f32 %0, %1" : "=v"(d[i]) : "v"(a));
#define A64(i) asm volatile("v_fma_f64 %0, %1, %2, %0" : "+v"(d[i]) : "v"(da), "v"(db));
#define A32(i) asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(r[i]) : "v"(a), "v"(b));
        if (OP == 0) { REP16(MUL) }
        if (OP == 1) { REP16(CVT) }
        if (OP == 2) { REP16(MIX) }
        if (OP == 3) { REP16(MLO) }
        if (OP == 4) { REP16(MHI) }
        if (OP == 5) { REP16(MLH) }
        if (OP == 6) { REP16(PKM) }
        if (OP == 7) { REP16(FMA) }
        if (OP == 8) { REP16(F64) }
        if (OP == 9) { REP16(C64) }
        if (OP == 10) { REP16(A64) }
        if (OP == 11) { REP16(A32) }
    }
    const long long t1 = __builtin_readcyclecounter();
    float s = 0;
    for (int i = 0; i < 16; ++i) s += r[i] + q[i][0] + q[i][1] + (float)d[i];
    if (s == 123.456f) out[threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) *cycles = t1 - t0;
}

template <int OP>
static void run(const char *name, int per_iter, float *out, long long *cyc)
{
    const int iters = 4096;
    for (int waves = 4; waves <= 12; waves += 8) { // one / three waves per SIMD
        long long c = 0;
        for (int rep = 0; rep < 2; ++rep) {
            hipLaunchKernelGGL(k_probe<OP>, dim3(256), dim3(waves * 64), 0, 0, out, cyc, iters);
            hipDeviceSynchronize();
        }
        hipMemcpy(&c, cyc, sizeof c, hipMemcpyDeviceToHost);
        printf("%-34s %2d waves/CU  %6.2f cycles per instruction per wave, %6.2f per SIMD slot\n", name, waves,
               (double)c / ((double)iters * per_iter), (double)c / ((double)iters * per_iter) / (waves / 4));
    }
}

int main()
{
    float *out;
    long long *cyc;
    hipMalloc(&out, 4096 * sizeof(float));
    hipMalloc(&cyc, sizeof(long long));
    run<0>("v_mul_f32", 16, out, cyc);
    run<7>("v_fma_f32", 16, out, cyc);
    run<1>("v_cvt_pk_f16_f32", 16, out, cyc);
    run<2>("v_fma_mix_f32 (f16 addend)", 16, out, cyc);
    run<3>("v_fma_mixlo_f16", 16, out, cyc);
    run<4>("v_fma_mixhi_f16", 16, out, cyc);
    run<5>("v_fma_mixlo_f16 + mixhi same reg", 32, out, cyc);
    run<6>("v_pk_mul_f32", 16, out, cyc);
    run<8>("v_fma_f64", 16, out, cyc);
    run<9>("v_cvt_f64_f32", 16, out, cyc);
    run<10>("v_fma_f64, 16 accumulation chains", 16, out, cyc);
    run<11>("v_fma_f32, 16 accumulation chains", 16, out, cyc);
    return 0;
}
