// Standalone probe: what limits a row-structured tile copy?  hipcc --offload-arch=gfx950 -O3 -o copy_probe copy_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef unsigned int u32x4_t __attribute__((ext_vector_type(4)));
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

__global__ __launch_bounds__(256) void v0_flat(const uint4* __restrict__ src, uint4* __restrict__ dst, size_t n16) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += (size_t)gridDim.x * blockDim.x) dst[i] = src[i];
}
// brick = 4 planes x 32 rows x 512 px (u16): one wave per brick, lane = 16 B of a row; G loads in flight
template <int G, bool BUF, bool CVT>
__global__ __launch_bounds__(256) void v1_rows(const unsigned short* __restrict__ src, unsigned short* __restrict__ dst, int nz, int ny, int nx,
                                               int oy_stride, int nbricks_y) {
    const int lane = threadIdx.x & 63;
    const int brick = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int by = brick % nbricks_y, bz = brick / nbricks_y;
    if (bz * 4 >= nz) return;
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)src, 0, (int)((size_t)nz * ny * nx * 2), 0x00020000);
    for (int p = 0; p < 4; ++p) {
        const int z = bz * 4 + p;
        for (int gb = 0; gb < 32; gb += G) {
            u32x4_t raw[G];
#pragma unroll
            for (int g = 0; g < G; ++g) {
                const int y = by * 32 + gb + g;
                const size_t o = ((size_t)z * ny + y) * nx + lane * 8;
                if (BUF) raw[g] = __builtin_amdgcn_raw_buffer_load_b128(rs, (int)(o * 2), 0, 0);
                else raw[g] = *reinterpret_cast<const u32x4_t*>(src + o);
            }
#pragma unroll
            for (int g = 0; g < G; ++g) {
                const int y = by * 32 + gb + g;
                u32x4_t v = raw[g];
                if (CVT) {
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        const float a = (float)(v[k] & 0xffffu), b = (float)(v[k] >> 16);
                        v[k] = (unsigned)(int)a | ((unsigned)(int)b << 16);
                    }
                }
                *reinterpret_cast<u32x4_t*>(dst + ((size_t)z * ny + y) * oy_stride + lane * 8) = v;
            }
        }
    }
}
template <int G>
__global__ __launch_bounds__(256) void v2_sub(const unsigned short* __restrict__ src, unsigned short* __restrict__ dst, int nz, int ny, int nx,
                                              int x0, int w, int ostride, int ox0, int nbricks_y) {
    const int lane = threadIdx.x & 63;
    const int brick = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int by = brick % nbricks_y, bz = brick / nbricks_y;
    if (bz * 4 >= nz) return;
    const bool act = lane * 8 < w;
    const int nvalid = min(8, w - lane * 8);
    for (int p = 0; p < 4; ++p) {
        const int z = bz * 4 + p;
        for (int gb = 0; gb < 32; gb += G) {
            u32x4_t raw[G];
#pragma unroll
            for (int g = 0; g < G; ++g) {
                const int y = by * 32 + gb + g;
                const size_t o = ((size_t)z * ny + y) * nx + x0 + (act ? lane * 8 : 0);
                typedef unsigned int v4 __attribute__((ext_vector_type(4), aligned(2)));
                const v4 t = *reinterpret_cast<const v4*>(src + o);
                raw[g] = t;
            }
#pragma unroll
            for (int g = 0; g < G; ++g) {
                const int y = by * 32 + gb + g;
                unsigned short* q = dst + ((size_t)z * ny + y) * ostride + ox0 + lane * 8;
                if (act) {
                    if (nvalid == 8) { typedef unsigned int v4 __attribute__((ext_vector_type(4), aligned(2))); v4 t = raw[g]; *reinterpret_cast<v4*>(q) = t; }
                    else for (int j = 0; j < nvalid; ++j) q[j] = (unsigned short)((raw[g][j >> 1] >> (16 * (j & 1))) & 0xffff);
                }
            }
        }
    }
}
// one wavefront writes BOTH pieces of its rows (A = px [0, wa), B = px [wa, 512)), piece after piece per row group: the lines
// shared by the two pieces are completed by the same wavefront within a few hundred cycles (row-owning design)
template <int G>
__global__ __launch_bounds__(256) void v3_both(const unsigned short* __restrict__ src, unsigned short* __restrict__ dst, int nz, int ny, int nx,
                                               int wa, int ostride, int ox0, int nbricks_y) {
    const int lane = threadIdx.x & 63;
    const int brick = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int by = brick % nbricks_y, bz = brick / nbricks_y;
    if (bz * 4 >= nz) return;
    typedef unsigned int v4 __attribute__((ext_vector_type(4), aligned(2)));
    for (int p = 0; p < 4; ++p) {
        const int z = bz * 4 + p;
        for (int gb = 0; gb < 32; gb += G) {
            for (int piece = 0; piece < 2; ++piece) {
                const int x0 = piece ? wa : 0, w = piece ? 512 - wa : wa;
                const bool act = lane * 8 < w;
                const int nvalid = min(8, w - lane * 8);
                v4 raw[G];
#pragma unroll
                for (int g = 0; g < G; ++g) {
                    const int y = by * 32 + gb + g;
                    raw[g] = *reinterpret_cast<const v4*>(src + ((size_t)z * ny + y) * nx + x0 + (act ? lane * 8 : 0));
                }
#pragma unroll
                for (int g = 0; g < G; ++g) {
                    const int y = by * 32 + gb + g;
                    unsigned short* q = dst + ((size_t)z * ny + y) * ostride + ox0 + x0 + lane * 8;
                    if (act) {
                        if (nvalid == 8) { v4 t = raw[g]; *reinterpret_cast<v4*>(q) = t; }
                        else for (int j = 0; j < nvalid; ++j) q[j] = (unsigned short)((raw[g][j >> 1] >> (16 * (j & 1))) & 0xffff);
                    }
                }
            }
        }
    }
}
template <typename F> float time_ms(F f) {
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    f(); f();
    hipEventRecord(a); for (int i = 0; i < 5; ++i) f(); hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b); return ms / 5;
}
int main() {
    const int nz = 1024, ny = 1024, nx = 512;
    const size_t n = (size_t)nz * ny * nx, bytes = n * 2;
    unsigned short *src, *dst;
    CHECK(hipMalloc(&src, bytes + (1 << 20))); CHECK(hipMalloc(&dst, bytes + (1 << 20)));
    CHECK(hipMemset(src, 1, bytes));
    const int nby = ny / 32, nbricks = (nz / 4) * nby;
    auto rep = [&](const char* name, float ms) { printf("%-34s %.3f ms  %.0f GB/s\n", name, ms, 2.0 * bytes / ms / 1e6); };
    rep("v0 flat uint4 grid-stride", time_ms([&] { hipLaunchKernelGGL(v0_flat, dim3(256 * 16), dim3(256), 0, 0, (const uint4*)src, (uint4*)dst, bytes / 16); }));
    rep("v1 rows G=8 global", time_ms([&] { hipLaunchKernelGGL((v1_rows<8, false, false>), dim3(nbricks / 4), dim3(256), 0, 0, src, dst, nz, ny, nx, nx, nby); }));
    rep("v1 rows G=8 buffer", time_ms([&] { hipLaunchKernelGGL((v1_rows<8, true, false>), dim3(nbricks / 4), dim3(256), 0, 0, src, dst, nz, ny, nx, nx, nby); }));
    rep("v1 rows G=8 buffer cvt", time_ms([&] { hipLaunchKernelGGL((v1_rows<8, true, true>), dim3(nbricks / 4), dim3(256), 0, 0, src, dst, nz, ny, nx, nx, nby); }));
    rep("v1 rows G=4 buffer cvt", time_ms([&] { hipLaunchKernelGGL((v1_rows<4, true, true>), dim3(nbricks / 4), dim3(256), 0, 0, src, dst, nz, ny, nx, nx, nby); }));
    rep("v1 rows G=16 buffer cvt", time_ms([&] { hipLaunchKernelGGL((v1_rows<16, true, true>), dim3(nbricks / 4), dim3(256), 0, 0, src, dst, nz, ny, nx, nx, nby); }));
    rep("v1 rows G=32 buffer", time_ms([&] { hipLaunchKernelGGL((v1_rows<32, true, false>), dim3(nbricks / 4), dim3(256), 0, 0, src, dst, nz, ny, nx, nx, nby); }));
    // misaligned output rows (stride nx + 6 px, like the 3484-byte rows of the mosaic)
    rep("v1 rows G=8 buffer cvt, odd stride", time_ms([&] { hipLaunchKernelGGL((v1_rows<8, true, true>), dim3(nbricks / 4), dim3(256), 0, 0, src, dst + 3, nz, ny, nx, nx, nby); }));
    // sub-range copies: 410 of 512 px per source row into rows of a wider destination (the mosaic's pattern)
    {
        const int w = 410, x0 = 102, onx = 1742;
        unsigned short* big;
        const size_t orows = (size_t)nz * ny;   // as many destination rows
        CHECK(hipMalloc(&big, orows * 1792 * 2 + (1 << 20)));
        auto run = [&](const char* name, int ostride, int ox0) {
            rep(name, time_ms([&] { hipLaunchKernelGGL((v2_sub<8>), dim3(nbricks / 4), dim3(256), 0, 0, src, big, nz, ny, nx, x0, w, ostride, ox0, nby); }) * (512.0 / w));
        };
        run("v2 410/512 -> stride 410 (scaled)", 410, 0);
        run("v2 410/512 -> stride 512 (scaled)", 512, 0);
        run("v2 410/512 -> stride 1742 (scaled)", onx, 512);
        // which part of the loss is alignment?  (w, x0, destination pitch and start varied; time scaled to full rows)
        auto run2 = [&](const char* name, int w2, int x02, int ostride, int ox0) {
            rep(name, time_ms([&] { hipLaunchKernelGGL((v2_sub<8>), dim3(nbricks / 4), dim3(256), 0, 0, src, big, nz, ny, nx, x02, w2, ostride, ox0, nby); }) * (512.0 / w2));
        };
        run2("v2 410/512 -> pitch 1792 @512", 410, 102, 1792, 512);
        run2("v2 384/512 (x0=128) -> pitch 1742 @512", 384, 128, 1742, 512);
        run2("v2 384/512 (x0=128) -> pitch 1792 @512", 384, 128, 1792, 512);
        run2("v2 384/512 (x0=128) -> pitch 384", 384, 128, 384, 0);
        run2("v2 512/512 -> pitch 1742 @512", 512, 0, 1742, 512);
        run2("v2 512/512 -> pitch 1792 @512", 512, 0, 1792, 512);
        // two writers of the boundary lines: piece A = px [0,410) and piece B = px [410,512) of every row, written by
        // two launches one after the other, by two launches on two streams, and the alignment-friendly split 384 | 128
        hipStream_t s1, s2; CHECK(hipStreamCreate(&s1)); CHECK(hipStreamCreate(&s2));
        auto two = [&](const char* name, int wa, int ostride, hipStream_t sa, hipStream_t sb) {
            rep(name, time_ms([&] {
                hipLaunchKernelGGL((v2_sub<8>), dim3(nbricks / 4), dim3(256), 0, sa, src, big, nz, ny, nx, 0, wa, ostride, 512, nby);
                hipLaunchKernelGGL((v2_sub<8>), dim3(nbricks / 4), dim3(256), 0, sb, src, big, nz, ny, nx, wa, 512 - wa, ostride, 512 + wa, nby);
                if (sa != sb) { hipStreamSynchronize(sa); hipStreamSynchronize(sb); }
            }));
        };
        two("410|102 pitch 1742, sequential", 410, 1742, 0, 0);
        two("410|102 pitch 1742, two streams", 410, 1742, s1, s2);
        two("384|128 pitch 1792, sequential", 384, 1792, 0, 0);
        two("384|128 pitch 1792, two streams", 384, 1792, s1, s2);
        two("410|102 pitch 1792, sequential", 410, 1792, 0, 0);
        rep("410|102 pitch 1742, same wavefront", time_ms([&] { hipLaunchKernelGGL((v3_both<8>), dim3(nbricks / 4), dim3(256), 0, 0, src, big, nz, ny, nx, 410, 1742, 512, nby); }));
        rep("384|128 pitch 1792, same wavefront", time_ms([&] { hipLaunchKernelGGL((v3_both<8>), dim3(nbricks / 4), dim3(256), 0, 0, src, big, nz, ny, nx, 384, 1792, 512, nby); }));
    }
    return 0;
}
