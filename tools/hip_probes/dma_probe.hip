// dma_probe.hip -- memory-system ceilings of the access shapes the fuse kernels can choose from (MI355X).
//   read variants  (LDS-DMA `buffer_load_dwordx4 ... lds`, data dropped):
//     R0 aligned 1 KiB rows (64 lanes x 16 B contiguous, 16-byte aligned)
//     R1 the same rows shifted by 2 bytes (every lane's 16 bytes straddle two 16-byte slots)
//     R2 pieces: 8 rows x 128 B per instruction (row pitch 1 KiB), aligned
//     R3 pieces shifted by 2 bytes
//     R4 plain global_load_dwordx4 into VGPRs, aligned rows (reference)
//   write variants (global_store_dwordx4 of a constant):
//     W0 aligned 1 KiB rows, W1 rows shifted by 2 bytes, W2 pieces 8 x 128 B (pitch 3494 B, unaligned)
//   copy variants: C0 DMA aligned rows -> LDS -> aligned stores; C1 unaligned source + unaligned pitch-3494 destination
// build: hipcc --offload-arch=gfx950 -O3 tools/dma_probe.hip -o tools/dma_probe ; run: tools/dma_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef unsigned int u32x4_t __attribute__((ext_vector_type(4)));
typedef unsigned int u32x4_a2 __attribute__((ext_vector_type(4), aligned(2)));
typedef __attribute__((address_space(3))) void* lds_ptr_t;

template <int MODE>
__global__ __launch_bounds__(256) void rd(const char* __restrict__ src, long long nbytes, int iters, u32x4_t* sink) {
    extern __shared__ u32x4_t lds[];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const long long wid = (long long)blockIdx.x * 4 + wave, nw = (long long)gridDim.x * 4;
    u32x4_t* my = lds + wave * 64 * 8;
    u32x4_t acc = {0, 0, 0, 0};
    const long long per = 8192;                                   // bytes per wave step: 8 instructions x 1 KiB
    const long long nsteps = nbytes / per;
    for (long long st = wid; st < nsteps; st += nw) {
        const char* p = src + st * per;
        const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc((void*)p, 0, (int)per + 64, 0x00020000);
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            int off;
            if (MODE == 0 || MODE == 4) off = k * 1024 + lane * 16;
            else if (MODE == 1) off = k * 1024 + lane * 16 + 2;
            else if (MODE == 2) off = (lane >> 3) * 1024 + k * 128 + (lane & 7) * 16;       // 8 rows x 128 B, the 8 instructions walk along the rows
            else off = (lane >> 3) * 1024 + k * 128 + (lane & 7) * 16 + 2;
            if (MODE == 4) { u32x4_t v = __builtin_amdgcn_raw_buffer_load_b128(r, off, 0, 0); acc.x ^= v.x; acc.y ^= v.y; acc.z ^= v.z; acc.w ^= v.w; }
            else __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (lds_ptr_t)(my + k * 64), 16, off, 0, 0, 0);
        }
        if (MODE != 4) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    if (iters < 0) sink[threadIdx.x] = (MODE == 4) ? acc : my[lane];
}

template <int MODE>
__global__ __launch_bounds__(256) void wr(char* __restrict__ dst, long long nbytes) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const long long wid = (long long)blockIdx.x * 4 + wave, nw = (long long)gridDim.x * 4;
    const long long per = (MODE == 2) ? 8LL * 3494 : 8192;
    const long long nsteps = nbytes / per - 1;
    const u32x4_t v = {1u, 2u, 3u, (unsigned)lane};
    for (long long st = wid; st < nsteps; st += nw) {
        char* p = dst + st * per;
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            long long off;
            if (MODE == 0) off = k * 1024 + lane * 16;
            else if (MODE == 1) off = k * 1024 + lane * 16 + 2;
            else off = (long long)(lane >> 3) * 3494 + k * 128 + (lane & 7) * 16;          // 8 output rows (pitch 3494 B) x 128 B, unaligned
            u32x4_a2 o; o.x = v.x; o.y = v.y; o.z = v.z; o.w = v.w;
            if (MODE == 2 && k * 128 + (lane & 7) * 16 + 16 > 1024) continue;
            *reinterpret_cast<u32x4_a2*>(p + off) = o;
        }
    }
}

template <int MODE>
__global__ __launch_bounds__(256) void cp(const char* __restrict__ src, char* __restrict__ dst, long long nbytes) {
    extern __shared__ u32x4_t lds[];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const long long wid = (long long)blockIdx.x * 4 + wave, nw = (long long)gridDim.x * 4;
    u32x4_t* my = lds + wave * 64 * 8 * 2;
    const long long per = 8192, nsteps = nbytes / per - 2;
    int buf = 0;
    long long st = wid;
    auto issue = [&](long long s, u32x4_t* b) __attribute__((always_inline)) {
        const char* p = src + s * per;
        const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc((void*)p, 0, (int)per + 64, 0x00020000);
#pragma unroll
        for (int k = 0; k < 8; ++k)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (lds_ptr_t)(b + k * 64), 16, k * 1024 + lane * 16 + (MODE ? 2 : 0), 0, 0, 0);
    };
    if (st < nsteps) issue(st, my);
    for (; st < nsteps; st += nw) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (st + nw < nsteps) issue(st + nw, my + (buf ^ 1) * 512);
        char* q = dst + (MODE ? (st * 8) * 3494LL : st * per);
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const u32x4_t v = my[buf * 512 + k * 64 + lane];
            u32x4_a2 o; o.x = v.x; o.y = v.y; o.z = v.z; o.w = v.w;
            *reinterpret_cast<u32x4_a2*>(q + (MODE ? (long long)k * 3494 + lane * 16 : k * 1024 + lane * 16)) = o;
        }
        buf ^= 1;
    }
}

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
int main() {
    const long long N = 8LL << 30;
    char *a, *b; u32x4_t* sink;
    CK(hipMalloc(&a, N + 4096)); CK(hipMalloc(&b, N * 4 + 4096)); CK(hipMalloc(&sink, 4096));
    CK(hipMemset(a, 1, N + 4096)); CK(hipMemset(b, 0, N));
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    auto timeit = [&](const char* name, auto launch, double bytes) {
        launch(); hipDeviceSynchronize();
        float best = 1e9f;
        for (int r = 0; r < 3; ++r) { hipEventRecord(e0); launch(); hipEventRecord(e1); hipEventSynchronize(e1); float ms; hipEventElapsedTime(&ms, e0, e1); best = ms < best ? ms : best; }
        printf("%-44s %7.3f ms  %6.2f TB/s\n", name, best, bytes / best / 1e9);
    };
    for (int blocks : {2048, 4096}) {
        printf("-- grid %d x 256 threads\n", blocks);
        timeit("R0 DMA aligned 1 KiB rows", [&] { hipLaunchKernelGGL(rd<0>, dim3(blocks), dim3(256), 4 * 8 * 1024, 0, a, N, 1, sink); }, N);
        timeit("R1 DMA rows shifted 2 B", [&] { hipLaunchKernelGGL(rd<1>, dim3(blocks), dim3(256), 4 * 8 * 1024, 0, a, N, 1, sink); }, N);
        timeit("R2 DMA pieces 8 x 128 B aligned", [&] { hipLaunchKernelGGL(rd<2>, dim3(blocks), dim3(256), 4 * 8 * 1024, 0, a, N, 1, sink); }, N);
        timeit("R3 DMA pieces 8 x 128 B shifted 2 B", [&] { hipLaunchKernelGGL(rd<3>, dim3(blocks), dim3(256), 4 * 8 * 1024, 0, a, N, 1, sink); }, N);
        timeit("R4 VGPR loads aligned rows", [&] { hipLaunchKernelGGL(rd<4>, dim3(blocks), dim3(256), 4 * 8 * 1024, 0, a, N, 1, sink); }, N);
        timeit("W0 stores aligned rows", [&] { hipLaunchKernelGGL(wr<0>, dim3(blocks), dim3(256), 0, 0, b, N); }, N);
        timeit("W1 stores rows shifted 2 B", [&] { hipLaunchKernelGGL(wr<1>, dim3(blocks), dim3(256), 0, 0, b, N); }, N);
        timeit("W2 stores pieces 8 x 128 B pitch 3494", [&] { hipLaunchKernelGGL(wr<2>, dim3(blocks), dim3(256), 0, 0, b, N); }, N / 8192.0 * 8 * 1024);
        timeit("C0 copy aligned (DMA -> LDS -> store)", [&] { hipLaunchKernelGGL(cp<0>, dim3(blocks), dim3(256), 4 * 16 * 1024, 0, a, b, N / 2); }, (double)N);
        timeit("C1 copy src +2 B, dst pitch 3494", [&] { hipLaunchKernelGGL(cp<1>, dim3(blocks), dim3(256), 4 * 16 * 1024, 0, a, b, N / 2); }, (double)N);
    }
    return 0;
}
