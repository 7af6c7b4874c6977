// mvs_score.hip -- candidate scoring of the pairwise registration (gfx950).
//
// mvs_score_candidates == the loop of the reference's registration.phase_correlation_registration
// (src/multiview_stitcher/registration.py:493-556), for n translation candidates t:
//   im1t = scipy.ndimage.affine_transform(im1, translate(t), order=1, mode="constant", cval=NaN)
//   mask = ~isnan(im1t) & ~isnan(im0);  skip (-1,-1) if empty or < 10 % of im1's valid voxels
//   region = union / intersection of the valid bounding boxes of im0 and im1t
//   `continue` if nanmax(im1t[region]) <= im1_min                                   (Q3)
//   SSIM(nan_to_num(im0[region]), nan_to_num(im1t[region]), data_range, win)        (skimage, float32)
//   quality = spearmanr(im0[mask], im1t[mask] - 1)                                  (scipy.stats)
// Everything voxel-sized runs on the device; the host only sees counters, bounding boxes and sums.
#include "mvs_internal.h"

#include <cstring>

#include <rocprim/rocprim.hpp>

#include <cmath>
#include <vector>

int mvs_stage_float_volume(MvsContext* c, const float* src, int32_t mem, long long n, int slot, float** dptr);   // mvs_reg.hip
int mvs_device_nanminmax(MvsContext* c, const float* d_in, long long n, float* mn, float* mx, long long* nvalid);   // mvs_reg.hip

namespace {

inline int grid_for(long long n) { return (int)std::min<long long>((n + 255) / 256, 256 * 8); }

struct Shape3 { int nz, ny, nx; };

// ---- shifted copy of im1 with scipy's affine_transform semantics (order 1, cval NaN) ------------
__device__ __forceinline__ int tap2(int i0, int n) {
    int i1 = i0 + 1;
    if (i1 >= n) i1 = (n > 1) ? n - 2 : 0;   // mirrored edge offset, weight 0 there
    return i1;
}

// Stats gathered while shifting: [0] #(valid im1t & valid im0), [1..6] bbox of valid im1t (min z,y,x, max z,y,x)
__global__ __launch_bounds__(256) void shift_kernel(const float* __restrict__ im1, const float* __restrict__ im0,
                                                    float* __restrict__ out, Shape3 S, double tz, double ty, double tx,
                                                    unsigned long long* __restrict__ count, int* __restrict__ bbox) {
    const long long n = (long long)S.nz * S.ny * S.nx;
    unsigned long long cnt = 0;
    int mnz = 0x7fffffff, mny = 0x7fffffff, mnx = 0x7fffffff, mxz = -1, mxy = -1, mxx = -1;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        const int x = (int)(i % S.nx);
        const long long t = i / S.nx;
        const int y = (int)(t % S.ny);
        const int z = (int)(t / S.ny);
        // identity matrix rows: ((z*1 + y*0) + x*0) + t  -- same rounding as scipy's loop
        const double cz = (double)z + tz, cy = (double)y + ty, cx = (double)x + tx;
        float r = NAN;
        if (!(cz < 0.0 || cz > (double)(S.nz - 1) || cy < 0.0 || cy > (double)(S.ny - 1) || cx < 0.0 || cx > (double)(S.nx - 1))) {
            const double fz = floor(cz), fy = floor(cy), fx = floor(cx);
            const int iz = (int)fz, iy = (int)fy, ix = (int)fx;
            const double wz = cz - fz, wy = cy - fy, wx = cx - fx;
            const int iz1 = tap2(iz, S.nz), iy1 = tap2(iy, S.ny), ix1 = tap2(ix, S.nx);
            const long long sy = S.nx, sz = (long long)S.ny * S.nx;
            double acc = 0.0;
            // scipy accumulates coeff * wz * wy * wx over the taps in z-major order
            acc += (double)im1[iz * sz + iy * sy + ix] * (1.0 - wz) * (1.0 - wy) * (1.0 - wx);
            acc += (double)im1[iz * sz + iy * sy + ix1] * (1.0 - wz) * (1.0 - wy) * wx;
            acc += (double)im1[iz * sz + iy1 * sy + ix] * (1.0 - wz) * wy * (1.0 - wx);
            acc += (double)im1[iz * sz + iy1 * sy + ix1] * (1.0 - wz) * wy * wx;
            acc += (double)im1[iz1 * sz + iy * sy + ix] * wz * (1.0 - wy) * (1.0 - wx);
            acc += (double)im1[iz1 * sz + iy * sy + ix1] * wz * (1.0 - wy) * wx;
            acc += (double)im1[iz1 * sz + iy1 * sy + ix] * wz * wy * (1.0 - wx);
            acc += (double)im1[iz1 * sz + iy1 * sy + ix1] * wz * wy * wx;
            r = (float)acc;
        }
        out[i] = r;
        if (r == r) {
            mnz = min(mnz, z); mny = min(mny, y); mnx = min(mnx, x);
            mxz = max(mxz, z); mxy = max(mxy, y); mxx = max(mxx, x);
            const float a = im0[i];
            if (a == a) ++cnt;
        }
    }
    for (int off = 32; off > 0; off >>= 1) {
        cnt += __shfl_down(cnt, off);
        mnz = min(mnz, __shfl_down(mnz, off)); mny = min(mny, __shfl_down(mny, off)); mnx = min(mnx, __shfl_down(mnx, off));
        mxz = max(mxz, __shfl_down(mxz, off)); mxy = max(mxy, __shfl_down(mxy, off)); mxx = max(mxx, __shfl_down(mxx, off));
    }
    // one set of atomics per workgroup (the per-wavefront version spent most of the kernel in atomic contention)
    __shared__ unsigned long long s_cnt[4];
    __shared__ int s_bb[4][6];
    const int wave = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) {
        s_cnt[wave] = cnt;
        s_bb[wave][0] = mnz; s_bb[wave][1] = mny; s_bb[wave][2] = mnx; s_bb[wave][3] = mxz; s_bb[wave][4] = mxy; s_bb[wave][5] = mxx;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int w = 1; w < 4; ++w) {
            cnt += s_cnt[w];
            mnz = min(mnz, s_bb[w][0]); mny = min(mny, s_bb[w][1]); mnx = min(mnx, s_bb[w][2]);
            mxz = max(mxz, s_bb[w][3]); mxy = max(mxy, s_bb[w][4]); mxx = max(mxx, s_bb[w][5]);
        }
        if (cnt) atomicAdd(count, cnt);
        if (mxz >= 0) {
            atomicMin(&bbox[0], mnz); atomicMin(&bbox[1], mny); atomicMin(&bbox[2], mnx);
            atomicMax(&bbox[3], mxz); atomicMax(&bbox[4], mxy); atomicMax(&bbox[5], mxx);
        }
    }
}

// bbox of the non-NaN voxels of one image (get_bb_from_nanmask, registration.py:482-489)
__global__ __launch_bounds__(256) void bbox_kernel(const float* __restrict__ im, Shape3 S, int* __restrict__ bbox) {
    const long long n = (long long)S.nz * S.ny * S.nx;
    int mnz = 0x7fffffff, mny = 0x7fffffff, mnx = 0x7fffffff, mxz = -1, mxy = -1, mxx = -1;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        const float v = im[i];
        if (v == v) {
            const int x = (int)(i % S.nx);
            const long long t = i / S.nx;
            const int y = (int)(t % S.ny), z = (int)(t / S.ny);
            mnz = min(mnz, z); mny = min(mny, y); mnx = min(mnx, x);
            mxz = max(mxz, z); mxy = max(mxy, y); mxx = max(mxx, x);
        }
    }
    for (int off = 32; off > 0; off >>= 1) {
        mnz = min(mnz, __shfl_down(mnz, off)); mny = min(mny, __shfl_down(mny, off)); mnx = min(mnx, __shfl_down(mnx, off));
        mxz = max(mxz, __shfl_down(mxz, off)); mxy = max(mxy, __shfl_down(mxy, off)); mxx = max(mxx, __shfl_down(mxx, off));
    }
    __shared__ int s_bb[4][6];
    const int wave = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) {
        s_bb[wave][0] = mnz; s_bb[wave][1] = mny; s_bb[wave][2] = mnx; s_bb[wave][3] = mxz; s_bb[wave][4] = mxy; s_bb[wave][5] = mxx;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int w = 1; w < 4; ++w) {
            mnz = min(mnz, s_bb[w][0]); mny = min(mny, s_bb[w][1]); mnx = min(mnx, s_bb[w][2]);
            mxz = max(mxz, s_bb[w][3]); mxy = max(mxy, s_bb[w][4]); mxx = max(mxx, s_bb[w][5]);
        }
        if (mxz >= 0) {
            atomicMin(&bbox[0], mnz); atomicMin(&bbox[1], mny); atomicMin(&bbox[2], mnx);
            atomicMax(&bbox[3], mxz); atomicMax(&bbox[4], mxy); atomicMax(&bbox[5], mxx);
        }
    }
}

// Extract the region [lo, lo+R) of im0 / im1t: x = nan_to_num(im0), y = nan_to_num(im1t), products, and
// the region's nanmax(im1t) / "has NaN" flags (for registration.py:530, 539).
__global__ __launch_bounds__(256) void region_kernel(const float* __restrict__ im0, const float* __restrict__ im1t, Shape3 S,
                                                     int lz, int ly, int lx, Shape3 R, float* __restrict__ X,
                                                     float* __restrict__ Y, float* __restrict__ XX, float* __restrict__ YY,
                                                     float* __restrict__ XY, unsigned int* __restrict__ maxbits,
                                                     int* __restrict__ hasnan) {
    const long long n = (long long)R.nz * R.ny * R.nx;
    float mx = -INFINITY;
    int hn = 0;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        const int x = (int)(i % R.nx);
        const long long t = i / R.nx;
        const int y = (int)(t % R.ny), z = (int)(t / R.ny);
        const long long src = ((long long)(z + lz) * S.ny + (y + ly)) * S.nx + (x + lx);
        float a = im0[src], b = im1t[src];
        if (b == b) mx = fmaxf(mx, b); else hn = 1;
        if (a != a) a = 0.f;
        if (b != b) b = 0.f;
        X[i] = a; Y[i] = b; XX[i] = a * a; YY[i] = b * b; XY[i] = a * b;
    }
    for (int off = 32; off > 0; off >>= 1) {
        mx = fmaxf(mx, __shfl_down(mx, off));
        hn |= __shfl_down(hn, off);
    }
    __shared__ float s_mx[4];
    __shared__ int s_hn[4];
    if ((threadIdx.x & 63) == 0) { s_mx[threadIdx.x >> 6] = mx; s_hn[threadIdx.x >> 6] = hn; }
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int w = 1; w < 4; ++w) { mx = fmaxf(mx, s_mx[w]); hn |= s_hn[w]; }
        // order-preserving float -> uint mapping so atomicMax works for negative values too
        unsigned int u = __float_as_uint(mx);
        u = (u & 0x80000000u) ? ~u : (u | 0x80000000u);
        atomicMax(maxbits, u);
        if (hn) atomicOr(hasnan, 1);
    }
}

// scipy.ndimage.uniform_filter1d(size=win, mode="reflect") along one axis for the five SSIM inputs at once:
// double accumulation, float32 output
struct Five { const float* src[5]; float* dst[5]; };
__global__ __launch_bounds__(256) void box1d_kernel(Five P, Shape3 R, int axis, int win) {
    const long long n = (long long)R.nz * R.ny * R.nx;
    const int dims[3] = {R.nz, R.ny, R.nx};
    const long long strides[3] = {(long long)R.ny * R.nx, R.nx, 1};
    const int len = dims[axis];
    const long long st = strides[axis];
    const int h = win / 2;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        const int x = (int)(i % R.nx);
        const long long t = i / R.nx;
        const int y = (int)(t % R.ny), z = (int)(t / R.ny);
        const int pos = (axis == 0) ? z : (axis == 1) ? y : x;
        const long long base = i - (long long)pos * st;
        double acc[5] = {0.0, 0.0, 0.0, 0.0, 0.0};
        for (int k = -h; k <= h; ++k) {
            int p = pos + k;
            // reflect: d c b a | a b c d | d c b a  (period 2*len)
            if (len == 1) p = 0;
            else {
                const int period = 2 * len;
                p %= period;
                if (p < 0) p += period;
                if (p >= len) p = period - 1 - p;
            }
            const long long o = base + (long long)p * st;
#pragma unroll
            for (int a = 0; a < 5; ++a) acc[a] += (double)P.src[a][o];
        }
#pragma unroll
        for (int a = 0; a < 5; ++a) P.dst[a][i] = (float)(acc[a] / (double)win);
    }
}

// SSIM map and the sum over the cropped interior (skimage structural_similarity, float32 map, float64 mean)
__global__ __launch_bounds__(256) void ssim_kernel(const float* __restrict__ ux, const float* __restrict__ uy,
                                                   const float* __restrict__ uxx, const float* __restrict__ uyy,
                                                   const float* __restrict__ uxy, Shape3 R, int pad, int ndim, float cov_norm,
                                                   float C1, float C2, double* __restrict__ partial) {
    const long long n = (long long)R.nz * R.ny * R.nx;
    double acc = 0.0;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        const int x = (int)(i % R.nx);
        const long long t = i / R.nx;
        const int y = (int)(t % R.ny), z = (int)(t / R.ny);
        const bool in = (x >= pad && x < R.nx - pad) && (y >= pad && y < R.ny - pad) && (ndim == 2 || (z >= pad && z < R.nz - pad));
        if (!in) continue;
        const float a = ux[i], b = uy[i];
        const float vx = cov_norm * (uxx[i] - a * a);
        const float vy = cov_norm * (uyy[i] - b * b);
        const float vxy = cov_norm * (uxy[i] - a * b);
        const float A1 = 2.f * a * b + C1, A2 = 2.f * vxy + C2;
        const float B1 = a * a + b * b + C1, B2 = vx + vy + C2;
        const float Sv = (A1 * A2) / (B1 * B2);
        acc += (double)Sv;
    }
    for (int off = 32; off > 0; off >>= 1) acc += __shfl_down(acc, off);
    __shared__ double s[4];
    if ((threadIdx.x & 63) == 0) s[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) partial[blockIdx.x] = s[0] + s[1] + s[2] + s[3];
}

// compaction of the jointly valid voxels: kx = im0[mask], ky = im1t[mask] - 1 (float32, like the reference).
// Each thread takes 8 consecutive voxels; a workgroup reserves its output range with ONE atomic (the order of
// the compacted pairs is irrelevant to a rank correlation).
__global__ __launch_bounds__(256) void compact_kernel(const float* __restrict__ im0, const float* __restrict__ im1t, long long n,
                                                      float* __restrict__ kx, float* __restrict__ ky,
                                                      unsigned int* __restrict__ counter) {
    constexpr int K = 8;
    __shared__ unsigned int s_wave[4];
    __shared__ unsigned int s_base;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const long long tile = 256LL * K;
    for (long long t0 = (long long)blockIdx.x * tile; t0 < n; t0 += (long long)gridDim.x * tile) {
        const long long i0 = t0 + (long long)threadIdx.x * K;
        float a[K], b[K];
        unsigned int cnt = 0;
#pragma unroll
        for (int k = 0; k < K; ++k) {
            const long long i = i0 + k;
            a[k] = (i < n) ? im0[i] : NAN;
            b[k] = (i < n) ? im1t[i] : NAN;
            cnt += (a[k] == a[k] && b[k] == b[k]) ? 1u : 0u;
        }
        // exclusive scan of cnt over the workgroup
        unsigned int incl = cnt;
        for (int off = 1; off < 64; off <<= 1) {
            const unsigned int o = __shfl_up(incl, off);
            if (lane >= off) incl += o;
        }
        if (lane == 63) s_wave[wave] = incl;
        __syncthreads();
        unsigned int wbase = 0;
        for (int w = 0; w < wave; ++w) wbase += s_wave[w];
        if (threadIdx.x == 255) s_base = atomicAdd(counter, wbase + incl);
        __syncthreads();
        unsigned int p = s_base + wbase + incl - cnt;
#pragma unroll
        for (int k = 0; k < K; ++k)
            if (a[k] == a[k] && b[k] == b[k]) {
                kx[p] = a[k];
                ky[p] = b[k] - 1.0f;
                ++p;
            }
        __syncthreads();
    }
}

__global__ void iota_kernel(unsigned int* __restrict__ v, unsigned int n) {
    for (unsigned int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) v[i] = i;
}

// average ranks (scipy.stats.rankdata method="average") from sorted keys: rank = (lo + hi + 1) / 2 where
// [lo, hi) is the run of equal keys around sorted position i; scattered back to the original order
__global__ __launch_bounds__(256) void ranks_kernel(const float* __restrict__ sorted, const unsigned int* __restrict__ idx,
                                                    unsigned int n, float* __restrict__ rank_out) {
    for (unsigned int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        const float k = sorted[i];
        const bool tie_l = (i > 0) && (sorted[i - 1] == k), tie_r = (i + 1 < n) && (sorted[i + 1] == k);
        if (!tie_l && !tie_r) {               // no tie (the common case for float images): rank = position + 1
            rank_out[idx[i]] = (float)((double)i + 1.0);
            continue;
        }
        // galloping search outwards from i: runs of equal keys are short compared with n, and the probes
        // stay in the cache lines around i instead of bouncing over the whole array
        unsigned int first = i, last = i + 1, lo, hi;
        if (tie_l) {
            unsigned int pos = i, step = 1;
            while (pos >= step && sorted[pos - step] == k) { pos -= step; step <<= 1; }
            lo = pos >= step ? pos - step + 1 : 0; hi = pos;          // first index with key == k in [lo, hi]
            while (lo < hi) { const unsigned int m = (lo + hi) >> 1; if (sorted[m] < k) lo = m + 1; else hi = m; }
            first = lo;
        }
        if (tie_r) {
            unsigned int pos = i, step = 1;
            while (pos + step < n && sorted[pos + step] == k) { pos += step; step <<= 1; }
            lo = pos + 1; hi = min(pos + step, n);                    // first index with key > k in [lo, hi]
            while (lo < hi) { const unsigned int m = (lo + hi) >> 1; if (sorted[m] <= k) lo = m + 1; else hi = m; }
            last = lo;
        }
        // ranks up to 2^32: keep them exact by storing (first + last + 1) / 2 as float pairs would lose bits,
        // so store as float the doubled rank split: exact for n < 2^24, else rounded (documented)
        rank_out[idx[i]] = (float)(0.5 * ((double)first + (double)last + 1.0));
    }
}

// sums for the Pearson correlation of the two rank vectors (both have mean (n+1)/2)
__global__ __launch_bounds__(256) void rankcorr_kernel(const float* __restrict__ rx, const float* __restrict__ ry, unsigned int n,
                                                       double mean, double* __restrict__ partial) {
    double sxy = 0.0, sxx = 0.0, syy = 0.0;
    for (unsigned int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        const double a = (double)rx[i] - mean, b = (double)ry[i] - mean;
        sxy += a * b; sxx += a * a; syy += b * b;
    }
    for (int off = 32; off > 0; off >>= 1) {
        sxy += __shfl_down(sxy, off); sxx += __shfl_down(sxx, off); syy += __shfl_down(syy, off);
    }
    __shared__ double s[3][4];
    if ((threadIdx.x & 63) == 0) { s[0][threadIdx.x >> 6] = sxy; s[1][threadIdx.x >> 6] = sxx; s[2][threadIdx.x >> 6] = syy; }
    __syncthreads();
    if (threadIdx.x == 0)
        for (int k = 0; k < 3; ++k) partial[blockIdx.x * 3 + k] = s[k][0] + s[k][1] + s[k][2] + s[k][3];
}

struct DeviceBump {   // bump allocator over one scratch slot
    char* base; size_t cap, used;
    template <typename T> T* take(size_t count) {
        const size_t bytes = (count * sizeof(T) + 255) / 256 * 256;
        if (used + bytes > cap) return nullptr;
        T* p = (T*)(base + used);
        used += bytes;
        return p;
    }
};

int rank_vector(MvsContext* c, float* keys, unsigned int n, float* keys_sorted, unsigned int* idx_in, unsigned int* idx_out,
                void* temp, size_t temp_bytes, float* ranks) {
    const int gb = grid_for(n);
    hipLaunchKernelGGL(iota_kernel, dim3(gb), dim3(256), 0, c->stream, idx_in, n);
    MVS_HIP_TRY(c, rocprim::radix_sort_pairs(temp, temp_bytes, keys, keys_sorted, idx_in, idx_out, (size_t)n, 0, 32, c->stream));
    hipLaunchKernelGGL(ranks_kernel, dim3(gb), dim3(256), 0, c->stream, keys_sorted, idx_out, n, ranks);
    MVS_HIP_TRY(c, hipGetLastError());
    return MVS_OK;
}

}  // namespace

extern "C" int mvs_score_candidates(int device, const float* fixed, const float* moving, int32_t mem, int32_t ndim,
                                    const int64_t shape[3], const double* t_candidates, int32_t n_candidates,
                                    int32_t region_mode, double data_range, double im1_min, int32_t quality_for_all,
                                    double* ssim_out, double* spearman_out, int32_t* code_out) {
    MvsContext* c;
    int rc = mvs_check_ready(device, &c);
    if (rc) return rc;
    std::lock_guard<std::mutex> lock(c->mu);
    if (!fixed || !moving || !shape || !t_candidates || !ssim_out || !spearman_out || !code_out || n_candidates < 0)
        return mvs_fail(c, MVS_ERR_INVALID_ARG, "mvs_score_candidates: bad argument");
    if (ndim != 2 && ndim != 3) return mvs_fail(c, MVS_ERR_INVALID_ARG, "mvs_score_candidates: ndim must be 2 or 3");
    if (ndim == 2 && shape[0] != 1) return mvs_fail(c, MVS_ERR_INVALID_ARG, "mvs_score_candidates: 2D needs shape[0]==1");
    MVS_HIP_TRY(c, hipSetDevice(device));
    const Shape3 S = {(int)shape[0], (int)shape[1], (int)shape[2]};
    const long long n = (long long)S.nz * S.ny * S.nx;
    if (n >= (1ll << 31)) return mvs_fail(c, MVS_ERR_UNSUPPORTED, "mvs_score_candidates: volume too large");
    const int k0 = 3 - ndim;

    float *im0, *im1;
    rc = mvs_stage_float_volume(c, fixed, mem, n, 4, &im0);
    if (rc) return rc;
    rc = mvs_stage_float_volume(c, moving, mem, n, 5, &im1);
    if (rc) return rc;

    size_t sort_temp_bytes = 0;
    MVS_HIP_TRY(c, rocprim::radix_sort_pairs(nullptr, sort_temp_bytes, (float*)nullptr, (float*)nullptr, (unsigned int*)nullptr,
                                            (unsigned int*)nullptr, (size_t)n, 0, 32, c->stream));
    const int gb = grid_for(n);
    const size_t need = (size_t)n * 4 * 14 + sort_temp_bytes + (size_t)gb * 64 + 64 * 1024;
    char* base = (char*)mvs_scratch(c, 6, need);
    if (!base) return MVS_ERR_HIP;
    DeviceBump B{base, need, 0};
    float* im1t = B.take<float>(n);
    float* X = B.take<float>(n);  float* Y = B.take<float>(n);
    float* XX = B.take<float>(n); float* YY = B.take<float>(n); float* XY = B.take<float>(n);
    float* T0 = B.take<float>(n); float* T1 = B.take<float>(n);   // filter ping-pong; reused as sort outputs
    float* UX = B.take<float>(n); float* UY = B.take<float>(n);
    float* UXX = B.take<float>(n); float* UYY = B.take<float>(n); float* UXY = B.take<float>(n);
    unsigned int* IDX = (unsigned int*)B.take<float>(n);
    void* sort_temp = B.take<char>(sort_temp_bytes);
    double* partial = B.take<double>((size_t)gb * 4);
    char* small = B.take<char>(4096);
    if (!small) return mvs_fail(c, MVS_ERR_HIP, "mvs_score_candidates: scratch layout");
    unsigned long long* d_count = (unsigned long long*)small;
    int* d_bbox = (int*)(small + 64);
    int* d_bbox0 = (int*)(small + 128);
    unsigned int* d_maxbits = (unsigned int*)(small + 192);
    int* d_hasnan = (int*)(small + 196);
    unsigned int* d_counter = (unsigned int*)(small + 200);

    // valid voxels of im1 and bbox of im0 (registration.py:400, 491)
    const int bb_init[6] = {0x7fffffff, 0x7fffffff, 0x7fffffff, -1, -1, -1};
    MVS_HIP_TRY(c, hipMemcpyAsync(d_bbox0, bb_init, sizeof(bb_init), hipMemcpyHostToDevice, c->stream));
    hipLaunchKernelGGL(bbox_kernel, dim3(std::min(gb, 512)), dim3(256), 0, c->stream, im0, S, d_bbox0);
    MVS_HIP_TRY(c, hipMemcpyAsync(d_bbox, bb_init, sizeof(bb_init), hipMemcpyHostToDevice, c->stream));
    hipLaunchKernelGGL(bbox_kernel, dim3(std::min(gb, 512)), dim3(256), 0, c->stream, im1, S, d_bbox);
    int bb0[6], bbm[6];
    MVS_HIP_TRY(c, hipMemcpyAsync(bb0, d_bbox0, sizeof(bb0), hipMemcpyDeviceToHost, c->stream));
    MVS_HIP_TRY(c, hipMemcpyAsync(bbm, d_bbox, sizeof(bbm), hipMemcpyDeviceToHost, c->stream));
    float mn1, mx1;
    long long nv1 = 0;
    rc = mvs_device_nanminmax(c, im1, n, &mn1, &mx1, &nv1);   // synchronises the stream
    if (rc) return rc;
    const unsigned int valid1 = (unsigned int)nv1;

    // ---- Spearman over the jointly valid voxels of the candidate whose shifted image is in im1t ----
    std::vector<unsigned long long> cnts((size_t)std::max(n_candidates, 1), 0ull);
    auto spearman_of_current = [&](int ic) -> int {
        MVS_HIP_TRY(c, hipMemsetAsync(d_counter, 0, 4, c->stream));
        hipLaunchKernelGGL(compact_kernel, dim3(gb), dim3(256), 0, c->stream, im0, im1t, n, X, Y, d_counter);
        const unsigned int m = (unsigned int)cnts[ic];
        int r = rank_vector(c, X, m, XX, IDX, (unsigned int*)UXY, sort_temp, sort_temp_bytes, UX);
        if (r) return r;
        r = rank_vector(c, Y, m, YY, IDX, (unsigned int*)UXY, sort_temp, sort_temp_bytes, UY);
        if (r) return r;
        const int mgb = grid_for(m);
        hipLaunchKernelGGL(rankcorr_kernel, dim3(mgb), dim3(256), 0, c->stream, UX, UY, m, 0.5 * ((double)m + 1.0), partial);
        std::vector<double> hp((size_t)mgb * 3);
        MVS_HIP_TRY(c, hipMemcpyAsync(hp.data(), partial, sizeof(double) * hp.size(), hipMemcpyDeviceToHost, c->stream));
        MVS_HIP_TRY(c, hipStreamSynchronize(c->stream));
        double sxy = 0, sxx = 0, syy = 0;
        for (int i = 0; i < mgb; ++i) { sxy += hp[i * 3]; sxx += hp[i * 3 + 1]; syy += hp[i * 3 + 2]; }
        spearman_out[ic] = sxy / std::sqrt(sxx * syy);
        return MVS_OK;
    };

    MVS_HIP_TRY(c, hipEventRecord(c->ev_start, c->stream));
    for (int ic = 0; ic < n_candidates; ++ic) {
        double t[3] = {0.0, 0.0, 0.0};
        for (int k = 0; k < ndim; ++k) t[k0 + k] = t_candidates[ic * ndim + k];
        ssim_out[ic] = -1.0;
        spearman_out[ic] = -1.0;
        code_out[ic] = 0;

        // Upper bound of the mask count from the valid bounding boxes: im1t can only be valid where x + t lies in
        // im1's valid box.  If even the bound fails the 10 % test the candidate is rejected exactly as the
        // reference rejects it (registration.py:503-505) without touching the volume.
        {
            double bound = 1.0;
            const int dimv[3] = {S.nz, S.ny, S.nx};
            for (int k = 0; k < 3; ++k) {
                const double lo1 = std::ceil((double)bbm[k] - t[k] - 1.0), hi1 = std::floor((double)bbm[3 + k] - t[k] + 1.0);
                const double lo = std::max(std::max(lo1, (double)bb0[k]), 0.0);
                const double hi = std::min(std::min(hi1, (double)bb0[3 + k]), (double)(dimv[k] - 1));
                bound *= std::max(hi - lo + 1.0, 0.0);
            }
            if (valid1 == 0 || bound == 0.0 || bound / (double)valid1 < 0.1) {
                code_out[ic] = 1;
                continue;
            }
        }
        MVS_HIP_TRY(c, hipMemsetAsync(d_count, 0, 8, c->stream));
        MVS_HIP_TRY(c, hipMemcpyAsync(d_bbox, bb_init, sizeof(bb_init), hipMemcpyHostToDevice, c->stream));
        hipLaunchKernelGGL(shift_kernel, dim3(std::min(gb, 1024)), dim3(256), 0, c->stream, im1, im0, im1t, S, t[0], t[1], t[2], d_count, d_bbox);
        unsigned long long cnt = 0;
        int bb1[6];
        MVS_HIP_TRY(c, hipMemcpyAsync(&cnt, d_count, 8, hipMemcpyDeviceToHost, c->stream));
        MVS_HIP_TRY(c, hipMemcpyAsync(bb1, d_bbox, sizeof(bb1), hipMemcpyDeviceToHost, c->stream));
        MVS_HIP_TRY(c, hipStreamSynchronize(c->stream));
        if (cnt == 0 || (double)cnt / (double)valid1 < 0.1) {   // registration.py:503-505
            code_out[ic] = 1;
            continue;
        }
        // region slices (registration.py:509-528)
        int lo[3], hi[3];
        for (int k = 0; k < 3; ++k) {
            if (region_mode == 0) { lo[k] = std::min(bb0[k], bb1[k]); hi[k] = std::max(bb0[3 + k], bb1[3 + k]) + 1; }
            else { lo[k] = std::max(bb0[k], bb1[k]); hi[k] = std::min(bb0[3 + k], bb1[3 + k]) + 1; }
        }
        Shape3 R = {std::max(hi[0] - lo[0], 0), std::max(hi[1] - lo[1], 0), std::max(hi[2] - lo[2], 0)};
        const long long rn = (long long)R.nz * R.ny * R.nx;
        float region_nanmax = NAN;
        int region_hasnan = 0;
        if (rn > 0) {
            MVS_HIP_TRY(c, hipMemsetAsync(d_maxbits, 0, 8, c->stream));   // maxbits + hasnan
            hipLaunchKernelGGL(region_kernel, dim3(grid_for(rn)), dim3(256), 0, c->stream, im0, im1t, S, lo[0], lo[1], lo[2], R,
                               X, Y, XX, YY, XY, d_maxbits, d_hasnan);
            unsigned int mb[2];
            MVS_HIP_TRY(c, hipMemcpyAsync(mb, d_maxbits, 8, hipMemcpyDeviceToHost, c->stream));
            MVS_HIP_TRY(c, hipStreamSynchronize(c->stream));
            region_hasnan = (int)mb[1];
            if (mb[0] != 0) {
                unsigned int u = (mb[0] & 0x80000000u) ? (mb[0] & 0x7fffffffu) : ~mb[0];
                memcpy(&region_nanmax, &u, 4);
                if (region_nanmax == -INFINITY) region_nanmax = NAN;   // all-NaN region
            }
        }
        // `if np.nanmax(im1t[mask_slices]) <= im1_min: continue` (Q3: nothing is appended)
        if (region_nanmax <= (float)im1_min) {
            code_out[ic] = 2;
            continue;
        }
        // ---- SSIM ----
        int min_shape = 0x7fffffff;
        for (int k = k0; k < 3; ++k) min_shape = std::min(min_shape, (&R.nz)[k]);
        int win = std::min(7, min_shape - ((min_shape - 1) % 2));
        const float region_max = region_hasnan ? NAN : region_nanmax;   // np.max propagates NaN
        if (win < 3 || region_max <= (float)im1_min) {
            ssim_out[ic] = -1.0;
        } else {
            // ping-pong between the input set {X,Y,XX,YY,XY} and the output set {UX,..}: 2 or 3 passes
            float* setA[5] = {X, Y, XX, YY, XY};
            float* setB[5] = {UX, UY, UXX, UYY, UXY};
            const int rgb = grid_for(rn);
            float** cur = setA;
            float** nxt = setB;
            for (int axis = k0; axis < 3; ++axis) {
                Five P5;
                for (int a = 0; a < 5; ++a) { P5.src[a] = cur[a]; P5.dst[a] = nxt[a]; }
                hipLaunchKernelGGL(box1d_kernel, dim3(rgb), dim3(256), 0, c->stream, P5, R, axis, win);
                float** tmp = cur; cur = nxt; nxt = tmp;
            }
            float** fin = cur;   // holds the filtered arrays
            double NP = 1.0;
            for (int k = 0; k < ndim; ++k) NP *= (double)win;
            const float cov_norm = (float)(NP / (NP - 1.0));
            // (K1 * R) ** 2 with R a float32 scalar: numpy keeps this in float32
            const float Rf = (float)data_range;
            const float C1 = (0.01f * Rf) * (0.01f * Rf);
            const float C2 = (0.03f * Rf) * (0.03f * Rf);
            const int pad = (win - 1) / 2;
            hipLaunchKernelGGL(ssim_kernel, dim3(rgb), dim3(256), 0, c->stream, fin[0], fin[1], fin[2], fin[3], fin[4], R, pad, ndim, cov_norm, C1, C2, partial);
            std::vector<double> hp(rgb);
            MVS_HIP_TRY(c, hipMemcpyAsync(hp.data(), partial, sizeof(double) * rgb, hipMemcpyDeviceToHost, c->stream));
            MVS_HIP_TRY(c, hipStreamSynchronize(c->stream));
            double sum = 0.0;
            for (double v : hp) sum += v;
            double cropn = 1.0;
            for (int k = k0; k < 3; ++k) cropn *= (double)((&R.nz)[k] - 2 * pad);
            ssim_out[ic] = sum / cropn;
        }
        cnts[ic] = cnt;
        if (!quality_for_all) continue;
        rc = spearman_of_current(ic);
        if (rc) return rc;
    }
    if (!quality_for_all) {
        // The reference reports the Spearman coefficient of the SSIM-argmax candidate only (registration.py:
        // 543-556), so the rank correlation is evaluated for the candidates that hold the best SSIM (all of
        // them when several tie) and is NaN for the others.
        double best = -INFINITY;
        for (int ic = 0; ic < n_candidates; ++ic)
            if (code_out[ic] != 2 && ssim_out[ic] > best) best = ssim_out[ic];
        for (int ic = 0; ic < n_candidates; ++ic) {
            if (code_out[ic] != 0) continue;
            if (!(ssim_out[ic] == best)) { spearman_out[ic] = NAN; continue; }
            double t[3] = {0.0, 0.0, 0.0};
            for (int k = 0; k < ndim; ++k) t[k0 + k] = t_candidates[ic * ndim + k];
            MVS_HIP_TRY(c, hipMemsetAsync(d_count, 0, 8, c->stream));
            MVS_HIP_TRY(c, hipMemcpyAsync(d_bbox, bb_init, sizeof(bb_init), hipMemcpyHostToDevice, c->stream));
            hipLaunchKernelGGL(shift_kernel, dim3(std::min(gb, 1024)), dim3(256), 0, c->stream, im1, im0, im1t, S, t[0], t[1], t[2], d_count, d_bbox);
            rc = spearman_of_current(ic);
            if (rc) return rc;
        }
    }
    MVS_HIP_TRY(c, hipEventRecord(c->ev_stop, c->stream));
    c->timing_valid = true;
    MVS_HIP_TRY(c, hipStreamSynchronize(c->stream));
    return MVS_OK;
}
