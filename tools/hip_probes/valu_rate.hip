// VALU issue-rate probe for gfx950: wave-instructions per cycle per SIMD for a few instruction kinds.
// build: hipcc --offload-arch=gfx950 -O3 valu_rate.hip -o valu_rate
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f2 __attribute__((ext_vector_type(2)));

template <int KIND>
__global__ __launch_bounds__(256) void k(float* out, int iters, float seed) {
    float a[8];
    f2 p[8];
    double d[8];
    for (int i = 0; i < 8; ++i) { a[i] = seed + threadIdx.x * 1e-3f + i; p[i] = f2{a[i], a[i] + 1.f}; d[i] = a[i]; }
    const double dm = seed * 0.999;
    const float m = seed * 0.999f, c = seed * 1e-3f;
    const f2 pm = {m, m}, pc = {c, c};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                if (KIND == 0) a[i] = __builtin_fmaf(a[i], m, c);                                   // v_fma_f32
                if (KIND == 1) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p[i]) : "v"(pm), "v"(pc));
                if (KIND == 2) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(a[i]) : "v"(m));
                if (KIND == 3) asm volatile("v_cvt_f32_u32 %0, %0" : "+v"(a[i]));
                if (KIND == 4) asm volatile("v_max_f32 %0, %0, %1" : "+v"(a[i]) : "v"(m));
                if (KIND == 5) asm volatile("v_rcp_f32 %0, %0" : "+v"(a[i]));
                if (KIND == 6) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(p[i]) : "v"(pm));
                if (KIND == 7) asm volatile("v_add_u32 %0, %0, %1" : "+v"(a[i]) : "v"(m));
                if (KIND == 8) asm volatile("v_cvt_f32_u32_sdwa %0, %0 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1" : "+v"(a[i]));
                if (KIND == 10) asm volatile("v_add_f64 %0, %0, %1" : "+v"(d[i]) : "v"(dm));
                if (KIND == 11) asm volatile("v_mul_f64 %0, %0, %1" : "+v"(d[i]) : "v"(dm));
                if (KIND == 12) asm volatile("v_cvt_f64_f32 %0, %1" : "+v"(d[i]) : "v"(a[i]));
                if (KIND == 13) asm volatile("v_cvt_f32_f64 %0, %1" : "+v"(a[i]) : "v"(d[i]));
                if (KIND == 14) asm volatile("v_fma_f64 %0, %0, %1, %1" : "+v"(d[i]) : "v"(dm));
                if (KIND == 15) asm volatile("v_cndmask_b32 %0, %0, %1, s[10:11]" : "+v"(a[i]) : "v"(m) : "s10", "s11");
                if (KIND == 16) asm volatile("v_min_f32 %0, %0, %1" : "+v"(a[i]) : "v"(m));
                if (KIND == 17) asm volatile("v_med3_f32 %0, %0, %1, %1" : "+v"(a[i]) : "v"(m));
                if (KIND == 18) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(a[i]) : "v"(m));
                if (KIND == 19) asm volatile("v_add_f32 %0, %0, %1" : "+v"(a[i]) : "v"(m));
                if (KIND == 20) asm volatile("v_mov_b32 %0, %1" : "+v"(a[i]) : "v"(m));
                if (KIND == 21) asm volatile("v_rndne_f32 %0, %0" : "+v"(a[i]));
                if (KIND == 22) asm volatile("v_cvt_i32_f32 %0, %0" : "+v"(a[i]));
                if (KIND == 23) asm volatile("v_lshl_or_b32 %0, %0, 16, %1" : "+v"(a[i]) : "v"(m));
                if (KIND == 9) asm volatile("v_cmp_lt_f32 vcc, %0, %1" : : "v"(a[i]), "v"(m) : "vcc");
            }
    }
    float s = 0;
    for (int i = 0; i < 8; ++i) s += a[i] + p[i].x + p[i].y + (float)d[i];
    if (s == 12345.678f) out[0] = s;
}

template <int KIND>
void run(const char* name, float* d, int waves_per_simd) {
    const int iters = 4096;
    const int blocks = 256 * waves_per_simd;     // 4 waves per block = one per SIMD
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k<KIND>, dim3(blocks), dim3(256), 0, 0, d, 16, 1.0f);
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<KIND>, dim3(blocks), dim3(256), 0, 0, d, iters, 1.0f);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    const double winstr = (double)blocks * 4 * iters * 32;          // wave-instructions
    const double per_simd_per_s = winstr / 1024 / (ms * 1e-3);
    printf("%-22s waves/SIMD %d: %8.3f ms  %.3f G wave-instr/s/SIMD (= %.2f cycles per wave-instr at 2.4 GHz)\n", name, waves_per_simd, ms,
           per_simd_per_s / 1e9, 2.4e9 / per_simd_per_s);
}

int main() {
    float* d;
    hipMalloc(&d, 1024);
    for (int w : {4, 8}) {
        run<0>("v_fma_f32", d, w);
        run<1>("v_pk_fma_f32", d, w);
        run<6>("v_pk_mul_f32", d, w);
        run<2>("v_cndmask_b32", d, w);
        run<3>("v_cvt_f32_u32", d, w);
        run<8>("v_cvt_f32_u32_sdwa", d, w);
        run<4>("v_max_f32", d, w);
        run<7>("v_add_u32", d, w);
        run<9>("v_cmp_lt_f32", d, w);
        run<5>("v_rcp_f32", d, w);
        run<10>("v_add_f64", d, w);
        run<11>("v_mul_f64", d, w);
        run<14>("v_fma_f64", d, w);
        run<12>("v_cvt_f64_f32", d, w);
        run<13>("v_cvt_f32_f64", d, w);
        run<15>("v_cndmask_b32 sgpr", d, w);
        run<16>("v_min_f32", d, w);
        run<17>("v_med3_f32", d, w);
        run<18>("v_mul_f32", d, w);
        run<19>("v_add_f32", d, w);
        run<20>("v_mov_b32", d, w);
        run<21>("v_rndne_f32", d, w);
        run<22>("v_cvt_i32_f32", d, w);
        run<23>("v_lshl_or_b32", d, w);
    }
    return 0;
}
