// mvs_fft_slab.hip -- the phase correlation's transforms in three passes over HBM instead of six (gfx950).
//
// Replaces, for crops with one short axis (a whole-line DFT length S, 17..64) and two power-of-two axes (64 / 128 / 256), the
// sequence fftn -> cross power -> ifftn -> argmax of skimage.registration.phase_cross_correlation as the reference calls it
// (registration.py:422-431):
//   1. slab_kernel<S>   forward along the short axis and one long axis by one workgroup per slab (mvs_fft_slab.inc), reading the two
//                       real crops;
//   2. long_xp_kernel   the remaining axis: forward transform of a line AND of its partner line (-k), the cross power of both
//                       normalisations (mvs_xpower_value: the same arithmetic as the single-axis path), the inverse transform of
//                       both lines -- the 3D spectrum itself never reaches memory; the plain cross power does, for the upsampled
//                       refinement (updft_yx2_kernel);
//   3. slab_kernel<S>   inverse along the two slab axes, reduced to the per-workgroup peaks of both channels (no store).
// 5 x 8 n bytes (n = voxels of a crop) instead of 12-13 x 8 n.  The DC term of the packed spectrum, which sets the power-of-two
// channel scales, is the sum of the slabs' DC terms; every workgroup of pass 2 adds them in the same order.
#include "mvs_fft.h"
#include "mvs_fft_dev.h"
#include "mvs_fft_reg.h"

#include <algorithm>

namespace {

template <int R1, int R2>
__global__ __launch_bounds__(256) void long_xp_kernel(LongArgs A) {
    constexpr int M = R1 * R2, TPL = R1 > R2 ? R1 : R2, LPB = 256 / TPL;      // threads per line, lines per workgroup
    constexpr int PS = TPL + 1, LS = (TPL * PS > M ? TPL * PS : M) + 1;        // padded strides (float2) that fit every exchange
    __shared__ float2 ex[LPB * LS];
    __shared__ float2 tw[M / 2];
    __shared__ float2 red[256];
    const int tid = (int)threadIdx.x;
    for (int t = tid; t < M / 2; t += 256) tw[t] = A.tw[t];
    // DC term of the packed spectrum: the slabs' DC terms, added in a fixed order (the same value in every workgroup)
    red[tid] = tid < A.ndc ? A.dc[tid] : make_float2(0.f, 0.f);
    __syncthreads();
    for (int off = 128; off > 0; off >>= 1) {
        if (tid < off) red[tid] = make_float2(red[tid].x + red[tid + off].x, red[tid].y + red[tid + off].y);
        __syncthreads();
    }
    const float2 z0 = red[0];
    float scale_phase = 1.f, scale_plain = 1.f;
    if (A.sel_b >= 0) mvs_xpower_scales(z0, A.ntotal, &scale_phase, &scale_plain);
    if (blockIdx.x == 0 && tid == 0 && A.z0_out) *A.z0_out = z0;
    // adjacent lines run along adjacent threads (coalesced: the lines of a pass lie side by side); slots 2 p / 2 p + 1 hold a line and
    // its partner (-a, -b) -- numbered over the canonical lines like fft_reg2_kernel's pairs: row a = 0 and, for even PA, row PA / 2
    // are their own partner rows (canonical: b <= PB / 2), rows 1 .. (PA - 1) / 2 pair with rows PA - a whole
    const int line = tid % LPB, idx = tid / LPB;
    int a = -1, b = 0;
    {
        const int h = A.PB / 2 + 1, F = (A.PA - 1) / 2;
        long long q = (long long)blockIdx.x * (LPB / 2) + (line >> 1);
        if (q < h) { a = 0; b = (int)q; }
        else {
            q -= h;
            if (q < (long long)F * A.PB) { a = 1 + (int)(q / A.PB); b = (int)(q % A.PB); }
            else {
                q -= (long long)F * A.PB;
                if (!(A.PA & 1) && q < h) { a = A.PA / 2; b = (int)q; }
            }
        }
    }
    bool live = a >= 0;
    bool selfp = false;
    if (live) {
        const int ma = a ? A.PA - a : 0, mb = b ? A.PB - b : 0;
        selfp = ma == a && mb == b;
        if (line & 1) {
            if (selfp) live = false;          // its own partner: the even slot has it
            a = ma; b = mb;
        }
    }
    const long long base = live ? (long long)a * A.pa_stride + b : 0;
    float2* row = ex + line * LS;
    const float2* prow = ex + (selfp ? line : (line ^ 1)) * LS;
    // ---- forward: pass 1 (j' = idx), exchange, pass 2 (p = idx): u[k] = X[idx + R1 k] ----
    if (idx < R2) {
        float2 v[R1];
#pragma unroll
        for (int q = 0; q < R1; ++q) v[q] = live ? A.Z[base + (long long)(q * R2 + idx) * A.stride] : make_float2(0.f, 0.f);
        dft_reg<R1>(v);
#pragma unroll
        for (int p = 0; p < R1; ++p) row[p * PS + idx] = p == 0 ? v[0] : cmul(v[p], tw_at(tw, idx * p, M / 2));
    }
    __syncthreads();
    float2 u[R2];
    if (idx < R1) {
#pragma unroll
        for (int j = 0; j < R2; ++j) u[j] = row[idx * PS + j];
        dft_reg<R2>(u);
    }
    __syncthreads();                                       // everybody has read the exchange
    if (idx < R1) {
#pragma unroll
        for (int k = 0; k < R2; ++k) row[idx + R1 * k] = u[k];      // the line's spectrum in natural order, for the partner
    }
    __syncthreads();
    // ---- cross power with the partner line's Z(-k); its inverse transform starts in place: this thread holds the samples
    // idx + R1 k, k < R2, which is what pass 1 of an (R2, R1) transform wants (cf. bluestein_reg_kernel)
    if (idx < R1) {
#pragma unroll
        for (int k = 0; k < R2; ++k) {
            const int m = idx + R1 * k;
            const float2 zp = prow[(M - m) & (M - 1)];
            float2 p, p1;
            const float2 w = mvs_xpower_value(u[k], zp, A.sel_a, A.sel_b, scale_phase, scale_plain, &p, &p1);
            if (live) A.P2[base + (long long)m * A.stride] = p;
            u[k] = make_float2(w.y, w.x);                  // IDFT(x) = swap(DFT(swap(x)))
        }
    }
    __syncthreads();                                       // everybody has read the partner's spectrum
    if (idx < R1) {
        dft_reg<R2>(u);
#pragma unroll
        for (int p = 0; p < R2; ++p) row[p * PS + idx] = p == 0 ? u[0] : cmul(u[p], tw_at(tw, idx * p, M / 2));
    }
    __syncthreads();
    if (idx < R2) {
        float2 w[R1];
#pragma unroll
        for (int j = 0; j < R1; ++j) w[j] = row[idx * PS + j];
        dft_reg<R1>(w);
        if (live) {
#pragma unroll
            for (int k = 0; k < R1; ++k) A.CC[base + (long long)(idx + R2 * k) * A.stride] = make_float2(w[k].y, w[k].x);
        }
    }
}

bool slab_long_length(int64_t n) { return n == 64 || n == 128 || n == 256; }

}  // namespace

// which axis is the short one (-1: the crop is not of the slab kind)
int mvs_phasecorr_slab_axis(const MvsContext* c, const int64_t shape[3]) {
    if (c->fft_no_slab || c->fft_no_line || c->reg_unfused) return -1;
    int as = -1;
    for (int k = 0; k < 3; ++k) {
        if (slab_long_length(shape[k])) continue;
        if (shape[k] > 64 || !mvs_dft_line_length((int)shape[k]) || as >= 0) return -1;
        as = k;
    }
    if (as >= 0 && !((c->fft_slab_axes >> as) & 1)) return -1;
    return as;
}
// workgroups of the last pass (= entries of the peak arrays the caller must provide)
int mvs_phasecorr_slab_peaks(const int64_t shape[3], int short_axis) { return (int)(short_axis == 0 ? shape[1] : shape[0]); }

// a, b: the two real crops; Z, CC, P2: complex work volumes of the crop's shape; dc: >= 256 complex values of scratch.
// On return: P2 = plain cross power (natural layout), peak_val / peak_idx [2][n_peak] the per-workgroup peaks of the two packed
// correlations (flat C-order indices), *z0_out the DC term of the packed spectrum (written by the device: host-visible memory).
int mvs_phasecorr_slab(MvsContext* c, const float* a, const float* b, float2* Z, float2* CC, float2* P2, float2* dc, const int64_t shape[3],
                       int short_axis, int sel_a, int sel_b, float* const peak_val[2], long long* const peak_idx[2], float2* z0_out) {
    const long long nz = shape[0], ny = shape[1], nx = shape[2];
    const int S = (int)shape[short_axis];
    SlabArgs F;
    LongArgs G;
    int L, N2;      // the slab's long axis, the remaining axis
    if (short_axis == 2) {             // (nz, ny = L, nx = S): slabs over z, each one contiguous block; lines of pass 2 along z
        L = (int)ny; N2 = (int)nz;
        F.short_contig = 1; F.slab_stride = ny * nx; F.row_stride = nx;
        G.stride = ny * nx; G.PA = (int)ny; G.PB = (int)nx; G.pa_stride = nx;
    } else if (short_axis == 1) {      // (nz, ny = S, nx = L): slabs over z, rows = y; lines of pass 2 along z
        L = (int)nx; N2 = (int)nz;
        F.short_contig = 0; F.slab_stride = ny * nx; F.row_stride = nx;
        G.stride = ny * nx; G.PA = (int)ny; G.PB = (int)nx; G.pa_stride = nx;
    } else {                           // (nz = S, ny, nx = L): slabs over y, rows = z; lines of pass 2 along y
        L = (int)nx; N2 = (int)ny;
        F.short_contig = 0; F.slab_stride = nx; F.row_stride = ny * nx;
        G.stride = nx; G.PA = (int)nz; G.PB = (int)nx; G.pa_stride = ny * nx;
    }
    const int nslab = N2;
    if (nslab > 256) return mvs_fail(c, MVS_ERR_UNSUPPORTED, "phase correlation (slab path): %d slabs", nslab);
    const int R1 = L == 64 ? 8 : 16, R2 = L == 256 ? 16 : 8;
    F.S = S; F.L = L;
    // LDS of slab_kernel: a batch of lines of the long axis (slab_long_axis: LPB lines of pitch LS) or, for rows of S contiguous
    // samples, the staging area of 64 rows
    const int lpb = 256 / std::max(R1, R2), ls = (std::max(L, R1 * (R2 + 1)) + 1) | 1;
    const size_t lds = (size_t)std::max(lpb * ls, 64 * (S | 1)) * sizeof(float2);
    int rc = mvs_fft_twiddles(c, L, &F.tw);
    if (rc) return rc;
    rc = mvs_fft_twiddles(c, N2, &G.tw);
    if (rc) return rc;
    // 1. forward slabs
    F.data = Z; F.re_src = a; F.im_src = b; F.inverse = 0; F.dc = dc;
    hipError_t err = hipSuccess;
    if (!(S <= 44 ? mvs_launch_slab_lo(c, F, (unsigned)nslab, lds, &err) : mvs_launch_slab_hi(c, F, (unsigned)nslab, lds, &err)))
        return mvs_fail(c, MVS_ERR_UNSUPPORTED, "phase correlation (slab path): no slab kernel for S = %d", S);
    MVS_HIP_TRY(c, err);
    // 2. the remaining axis: forward, cross power, inverse
    G.Z = Z; G.CC = CC; G.P2 = P2; G.sel_a = sel_a; G.sel_b = sel_b; G.dc = dc; G.ndc = nslab; G.ntotal = nz * ny * nx; G.z0_out = z0_out;
    {
        const long long h = G.PB / 2 + 1, ncanon = h + (long long)((G.PA - 1) / 2) * G.PB + ((G.PA % 2 == 0) ? h : 0);
        const int lpw = N2 == 64 ? 32 : 16;
        const unsigned grid = (unsigned)((ncanon + lpw / 2 - 1) / (lpw / 2));
        if (N2 == 256) MVS_DUP("long_xp", hipLaunchKernelGGL((long_xp_kernel<16, 16>), dim3(grid), dim3(256), 0, c->stream, G));
        else if (N2 == 128) hipLaunchKernelGGL((long_xp_kernel<16, 8>), dim3(grid), dim3(256), 0, c->stream, G);
        else hipLaunchKernelGGL((long_xp_kernel<8, 8>), dim3(grid), dim3(256), 0, c->stream, G);
        MVS_HIP_TRY(c, hipGetLastError());
    }
    // 3. inverse slabs, reduced to the peaks
    F.data = CC; F.re_src = nullptr; F.im_src = nullptr; F.inverse = 1; F.dc = nullptr;
    for (int k = 0; k < 2; ++k) { F.peak_val[k] = peak_val[k]; F.peak_idx[k] = peak_idx[k]; }
    if (!(S <= 44 ? mvs_launch_slab_lo(c, F, (unsigned)nslab, lds, &err) : mvs_launch_slab_hi(c, F, (unsigned)nslab, lds, &err)))
        return mvs_fail(c, MVS_ERR_UNSUPPORTED, "phase correlation (slab path): no slab kernel for S = %d", S);
    MVS_HIP_TRY(c, err);
    return MVS_OK;
}
