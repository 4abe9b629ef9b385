// valubench.hip -- issue rates of the VALU instructions the chain kernel is made of, on gfx950.
// Diagnostic only.  Each test runs R unrolled copies of one instruction on 8 independent register
// sets per lane, with W waves per SIMD, and reports cycles per wave-instruction per SIMD.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

typedef float f32x2 __attribute__((ext_vector_type(2)));

#define REP8(S) S(0) S(1) S(2) S(3) S(4) S(5) S(6) S(7)

template <int OP>
__global__ void k(float *out, int iters, float seed) {
    float a[8]; f32x2 p[8];
    for (int i = 0; i < 8; i++) { a[i] = seed + threadIdx.x * 1e-3f + i; p[i] = f32x2{ a[i], a[i] + 0.5f }; }
    unsigned long long smask = 0x5555555555555555ull ^ (unsigned long long)iters;
    float vmask = __uint_as_float(0xFFFFFFFFu ^ (threadIdx.x == 12345 ? 1u : 0u));
    long long t0 = clock64();
    for (int it = 0; it < iters; it++) {
#define MUL(i) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(a[i]) : "v"(seed));
#define FMA(i) asm volatile("v_fma_f32 %0, %0, %1, %0" : "+v"(a[i]) : "v"(seed));
#define PKMUL(i) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(p[i]) : "v"(p[(i + 1) & 7]));
#define PKFMA(i) asm volatile("v_pk_fma_f32 %0, %0, %1, %0" : "+v"(p[i]) : "v"(p[(i + 1) & 7]));
#define CVTF(i) asm volatile("v_cvt_f32_f16 %0, %0" : "+v"(a[i]));
#define CVTSDWA(i) asm volatile("v_cvt_f32_f16_sdwa %0, %0 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1" : "+v"(a[i]));
#define PKRTZ(i) asm volatile("v_cvt_pkrtz_f16_f32 %0, %0, %1" : "+v"(a[i]) : "v"(seed));
#define RCP(i) asm volatile("v_rcp_f32 %0, %0" : "+v"(a[i]));
#define MAX3(i) asm volatile("v_max3_f32 %0, %0, %1, %1" : "+v"(a[i]) : "v"(seed));
#define DIVSC(i) asm volatile("v_div_scale_f32 %0, vcc, %0, %1, %0" : "+v"(a[i]) : "v"(seed) : "vcc");
#define DIVFX(i) asm volatile("v_div_fixup_f32 %0, %0, %1, %0" : "+v"(a[i]) : "v"(seed));
#define CNDM(i) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(a[i]) : "v"(seed));
#define CNDS(i) asm volatile("v_cndmask_b32_e64 %0, %0, %1, %2" : "+v"(a[i]) : "v"(seed), "s"(smask));
#define CNDCMP(i) asm volatile("v_cmp_lt_f32 vcc, %0, %1\n\tv_cndmask_b32 %0, %0, %1, vcc" : "+v"(a[i]) : "v"(seed) : "vcc");
#define ANDM(i) asm volatile("v_and_b32 %0, %0, %1" : "+v"(a[i]) : "v"(vmask));
#define PKADD(i) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(p[i]) : "v"(p[(i + 1) & 7]));
#define PKADDDEP(i) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(p[0]) : "v"(p[(i + 1) & 7]));
#define MOV(i) asm volatile("v_mov_b32 %0, %1" : "+v"(a[i]) : "v"(seed));
#define LSHLSDWA(i) asm volatile("v_lshlrev_b32_sdwa %0, %1, %0 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_1" : "+v"(a[i]) : "v"(1));
        if (OP == 0) { REP8(MUL) REP8(MUL) }
        if (OP == 1) { REP8(FMA) REP8(FMA) }
        if (OP == 2) { REP8(PKMUL) REP8(PKMUL) }
        if (OP == 3) { REP8(PKFMA) REP8(PKFMA) }
        if (OP == 4) { REP8(CVTF) REP8(CVTF) }
        if (OP == 5) { REP8(CVTSDWA) REP8(CVTSDWA) }
        if (OP == 6) { REP8(PKRTZ) REP8(PKRTZ) }
        if (OP == 7) { REP8(RCP) REP8(RCP) }
        if (OP == 8) { REP8(MAX3) REP8(MAX3) }
        if (OP == 9) { REP8(DIVSC) REP8(DIVSC) }
        if (OP == 10) { REP8(DIVFX) REP8(DIVFX) }
        if (OP == 11) { REP8(CNDM) REP8(CNDM) }
        if (OP == 12) { REP8(LSHLSDWA) REP8(LSHLSDWA) }
        if (OP == 13) { REP8(CNDS) REP8(CNDS) }
        if (OP == 14) { REP8(CNDCMP) REP8(CNDCMP) }
        if (OP == 15) { REP8(ANDM) REP8(ANDM) }
        if (OP == 16) { REP8(PKADD) REP8(PKADD) }
        if (OP == 17) { REP8(PKADDDEP) REP8(PKADDDEP) }
        if (OP == 18) { REP8(MOV) REP8(MOV) }
    }
    long long t1 = clock64();
    float s = 0;
    for (int i = 0; i < 8; i++) s += a[i] + p[i].x + p[i].y;
    out[blockIdx.x * blockDim.x + threadIdx.x] = s + (float)(t1 - t0) * 0.0f;
    if (threadIdx.x == 0 && blockIdx.x == 0) ((long long *)out)[1 << 20] = t1 - t0;
}

template <int OP>
static void run(const char *name, float *d, int waves_per_simd) {
    const int iters = 4000;
    int block = 256 * waves_per_simd;           // 4 SIMDs per CU
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    hipLaunchKernelGGL(k<OP>, dim3(256), dim3(block), 0, 0, d, 10, 1.0001f);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL(k<OP>, dim3(256), dim3(block), 0, 0, d, iters, 1.0001f);
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    long long cyc; CK(hipMemcpy(&cyc, (char *)d + sizeof(long long) * (1 << 20), sizeof cyc, hipMemcpyDeviceToHost));
    double insts_per_simd = (double)iters * 16 * waves_per_simd;
    // clock64 is a constant-rate counter (100 MHz) on gfx9; use wall time and an assumed shader clock readout too
    printf("%-16s waves/SIMD %d: %.3f ms  -> %.2f ns per wave-instruction per SIMD (%.2f cycles @2.4GHz)\n",
           name, waves_per_simd, ms, ms * 1e6 / insts_per_simd, ms * 1e6 / insts_per_simd * 2.4);
}

int main() {
    float *d; CK(hipMalloc((void **)&d, (8 << 20) + 64));
    for (int w : { 2, 3, 4 }) {
        run<0>("v_mul_f32", d, w); run<1>("v_fma_f32", d, w); run<2>("v_pk_mul_f32", d, w); run<3>("v_pk_fma_f32", d, w);
        run<4>("v_cvt_f32_f16", d, w); run<5>("v_cvt_f32_f16_sdwa", d, w); run<6>("v_cvt_pkrtz", d, w); run<7>("v_rcp_f32", d, w);
        run<8>("v_max3_f32", d, w); run<9>("v_div_scale_f32", d, w); run<10>("v_div_fixup_f32", d, w); run<11>("v_cndmask_b32", d, w);
        run<12>("v_lshlrev_sdwa", d, w); run<13>("v_cndmask sgpr mask", d, w); run<14>("v_cmp+v_cndmask (2 instr)", d, w); run<15>("v_and_b32", d, w);
        run<16>("v_pk_add_f32", d, w); run<17>("v_pk_add_f32 dependent chain", d, w); run<18>("v_mov_b32", d, w);
    }
    return 0;
}
