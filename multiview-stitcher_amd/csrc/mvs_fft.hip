// mvs_fft.hip -- batched complex64 line FFTs (LDS-staged Stockham autosort, gfx950).
//
// Replaces scipy.fft.fftn / ifftn inside skimage.registration.phase_cross_correlation as called from
// the reference's registration.py:422-431.  The transform sizes are the overlap-crop shapes themselves
// (registration.py:314-316: 104, 411, 27, 53, ...) and circular correlation depends on N, so there is no
// padding to friendlier sizes: powers of two run as a radix-2 Stockham autosort FFT, every other N as
// a Bluestein chirp-z transform built on the same Stockham core of size M = pow2 >= 2N-1.
//
// A 3D transform is three passes over HBM (one per axis).  Each workgroup stages `lpb` lines in LDS
// (two ping-pong buffers), runs log2(M) butterfly stages there and writes the lines back, so every
// pass reads and writes each complex element exactly once.  Lines along y/z are taken `lpb` adjacent
// x positions at a time so that global accesses stay coalesced along x.
#include "mvs_fft.h"

#include <cmath>
#include <map>
#include <vector>

namespace {

__device__ __forceinline__ float2 cmul(float2 a, float2 b) {
    return make_float2(a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x);
}

struct FftArgs {
    float2* data;
    long long n_lines;        // number of lines of this pass
    int n;                    // transform length
    int M, log2M;             // Stockham size (== n for powers of two)
    long long stride;         // element stride along the transform axis
    long long inner;          // lines are enumerated l = outer * inner + i: base = outer * outer_stride + i
    long long outer_stride;
    int lpb;                  // lines per workgroup
    int inverse;              // 1: conjugate transform (unnormalised)
    const float2* tw;         // exp(-2 pi i m / M), m < M/2
    const float2* chirp;      // Bluestein: w_n = exp(-i pi n^2 / N), n < N
    const float2* bfft;       // Bluestein: FFT_M of the wrapped conj chirp, scaled by 1/M
};

// radix-2 Stockham stages on `lpb` lines of length M held in LDS; returns the buffer holding the result
__device__ __forceinline__ float2* stockham(float2* a, float2* b, int lpb, int M, int log2M, const float2* tw, bool inv) {
    const int half = M >> 1;
    for (int s = 0; s < log2M; ++s) {
        const int p = 1 << s;
        for (int t = threadIdx.x; t < lpb * half; t += blockDim.x) {
            const int line = t / half, i = t - line * half;
            const int k = i & (p - 1);
            const int j = ((i - k) << 1) + k;
            float2 w = tw[k * (half >> s)];
            if (inv) w.y = -w.y;
            const float2 u0 = a[line * M + i];
            const float2 u1 = cmul(w, a[line * M + i + half]);
            b[line * M + j] = make_float2(u0.x + u1.x, u0.y + u1.y);
            b[line * M + j + p] = make_float2(u0.x - u1.x, u0.y - u1.y);
        }
        __syncthreads();
        float2* tmp = a; a = b; b = tmp;
    }
    return a;
}

template <bool BLUESTEIN>
__global__ __launch_bounds__(256) void fft_lines_kernel(FftArgs A) {
    extern __shared__ float2 lds[];
    float2* buf0 = lds;
    float2* buf1 = lds + (size_t)A.lpb * A.M;
    const long long l0 = (long long)blockIdx.x * A.lpb;
    const int nl = (int)min((long long)A.lpb, A.n_lines - l0);
    const bool inv = A.inverse != 0;

    // ---- load: thread -> (k, line) with line fastest so adjacent x positions coalesce ----
    for (int t = threadIdx.x; t < A.lpb * A.M; t += blockDim.x) {
        int line, k;
        if (A.stride == 1) { line = t / A.M; k = t - line * A.M; }
        else { k = t / A.lpb; line = t - k * A.lpb; }
        float2 v = make_float2(0.f, 0.f);
        if (line < nl && k < A.n) {
            const long long l = l0 + line;
            const long long base = (l / A.inner) * A.outer_stride + (l % A.inner);
            v = A.data[base + (long long)k * A.stride];
            if (BLUESTEIN) {
                if (inv) v.y = -v.y;              // IDFT(x) = conj(DFT(conj x))
                v = cmul(v, A.chirp[k]);
            }
        }
        buf0[line * A.M + k] = v;
    }
    __syncthreads();

    float2* r;
    if (!BLUESTEIN) {
        r = stockham(buf0, buf1, A.lpb, A.M, A.log2M, A.tw, inv);
    } else {
        r = stockham(buf0, buf1, A.lpb, A.M, A.log2M, A.tw, false);
        float2* o = (r == buf0) ? buf1 : buf0;
        for (int t = threadIdx.x; t < A.lpb * A.M; t += blockDim.x) {
            const int k = t % A.M;
            r[t] = cmul(r[t], A.bfft[k]);
        }
        __syncthreads();
        r = stockham(r, o, A.lpb, A.M, A.log2M, A.tw, true);
    }

    // ---- store ----
    for (int t = threadIdx.x; t < A.lpb * A.n; t += blockDim.x) {
        int line, k;
        if (A.stride == 1) { line = t / A.n; k = t - line * A.n; }
        else { k = t / A.lpb; line = t - k * A.lpb; }
        if (line < nl) {
            float2 v = r[line * A.M + k];
            if (BLUESTEIN) {
                v = cmul(v, A.chirp[k]);
                if (inv) v.y = -v.y;
            }
            const long long l = l0 + line;
            const long long base = (l / A.inner) * A.outer_stride + (l % A.inner);
            A.data[base + (long long)k * A.stride] = v;
        }
    }
}

struct FftPlan {
    int n = 0, M = 0, log2M = 0;
    bool bluestein = false;
    float2* tw = nullptr;
    float2* chirp = nullptr;
    float2* bfft = nullptr;
};

std::map<long long, FftPlan> g_plans;   // key: device * 2^32 + n
std::mutex g_plan_mu;

void host_fft_pow2(std::vector<double>& re, std::vector<double>& im) {
    const size_t n = re.size();
    for (size_t i = 1, j = 0; i < n; ++i) {
        size_t bit = n >> 1;
        for (; j & bit; bit >>= 1) j ^= bit;
        j ^= bit;
        if (i < j) { std::swap(re[i], re[j]); std::swap(im[i], im[j]); }
    }
    for (size_t len = 2; len <= n; len <<= 1) {
        const double ang = -2.0 * M_PI / (double)len;
        for (size_t i = 0; i < n; i += len)
            for (size_t k = 0; k < len / 2; ++k) {
                const double wr = cos(ang * (double)k), wi = sin(ang * (double)k);
                const double xr = re[i + k + len / 2] * wr - im[i + k + len / 2] * wi;
                const double xi = re[i + k + len / 2] * wi + im[i + k + len / 2] * wr;
                re[i + k + len / 2] = re[i + k] - xr; im[i + k + len / 2] = im[i + k] - xi;
                re[i + k] += xr; im[i + k] += xi;
            }
    }
}

int get_plan(MvsContext* c, int n, FftPlan* out) {
    std::lock_guard<std::mutex> lock(g_plan_mu);
    const long long key = ((long long)mvs_hip_device(c->device) << 32) | (unsigned)n;   // plans live on the GPU, whichever lane made them
    auto it = g_plans.find(key);
    if (it != g_plans.end()) { *out = it->second; return MVS_OK; }
    FftPlan p;
    p.n = n;
    const bool pow2 = (n & (n - 1)) == 0;
    p.bluestein = !pow2;
    int M = 1;
    if (pow2) M = n;
    else while (M < 2 * n - 1) M <<= 1;
    p.M = M;
    p.log2M = 0;
    while ((1 << p.log2M) < M) ++p.log2M;
    std::vector<float2> tw(std::max(M / 2, 1));
    for (int m = 0; m < M / 2; ++m) {
        const double a = -2.0 * M_PI * (double)m / (double)M;
        tw[m] = make_float2((float)cos(a), (float)sin(a));
    }
    MVS_HIP_TRY(c, hipMalloc(&p.tw, tw.size() * sizeof(float2)));
    MVS_HIP_TRY(c, hipMemcpy(p.tw, tw.data(), tw.size() * sizeof(float2), hipMemcpyHostToDevice));
    if (p.bluestein) {
        std::vector<double> wr(n), wi(n);
        std::vector<float2> chirp(n);
        for (int k = 0; k < n; ++k) {
            const long long k2 = ((long long)k * k) % (2LL * n);   // exp(-i pi k^2 / n) has period 2n in k^2
            const double a = -M_PI * (double)k2 / (double)n;
            wr[k] = cos(a); wi[k] = sin(a);
            chirp[k] = make_float2((float)wr[k], (float)wi[k]);
        }
        std::vector<double> br(M, 0.0), bi(M, 0.0);
        br[0] = wr[0]; bi[0] = -wi[0];
        for (int k = 1; k < n; ++k) {
            br[k] = br[M - k] = wr[k];
            bi[k] = bi[M - k] = -wi[k];
        }
        host_fft_pow2(br, bi);
        std::vector<float2> bf(M);
        for (int k = 0; k < M; ++k) bf[k] = make_float2((float)(br[k] / M), (float)(bi[k] / M));
        MVS_HIP_TRY(c, hipMalloc(&p.chirp, n * sizeof(float2)));
        MVS_HIP_TRY(c, hipMemcpy(p.chirp, chirp.data(), n * sizeof(float2), hipMemcpyHostToDevice));
        MVS_HIP_TRY(c, hipMalloc(&p.bfft, M * sizeof(float2)));
        MVS_HIP_TRY(c, hipMemcpy(p.bfft, bf.data(), M * sizeof(float2), hipMemcpyHostToDevice));
    }
    g_plans[key] = p;
    *out = p;
    return MVS_OK;
}

}  // namespace

// In-place 3D (or 2D when shape[0]==1) complex64 FFT of a C-contiguous (nz,ny,nx) array on c->stream.
// inverse: unnormalised conjugate transform (the caller applies 1/N where it matters).
int mvs_fft3_c2c(MvsContext* c, float2* data, const int64_t shape[3], bool inverse) {
    const long long nz = shape[0], ny = shape[1], nx = shape[2];
    for (int axis = 2; axis >= 0; --axis) {
        const int n = (int)shape[axis];
        if (n == 1) continue;
        if (n > 4096) return mvs_fail(c, MVS_ERR_UNSUPPORTED, "FFT length %d > 4096 not supported", n);
        FftPlan p;
        int rc = get_plan(c, n, &p);
        if (rc) return rc;
        FftArgs A;
        A.data = data;
        A.n = n; A.M = p.M; A.log2M = p.log2M;
        A.inverse = inverse ? 1 : 0;
        A.tw = p.tw; A.chirp = p.chirp; A.bfft = p.bfft;
        if (axis == 2) { A.stride = 1; A.n_lines = nz * ny; A.inner = 1; A.outer_stride = nx; }
        else if (axis == 1) { A.stride = nx; A.n_lines = nz * nx; A.inner = nx; A.outer_stride = ny * nx; }
        else { A.stride = ny * nx; A.n_lines = ny * nx; A.inner = ny * nx; A.outer_stride = 0; }
        int lpb = std::max(1, std::min(16, 4096 / p.M));
        if (axis == 2) lpb = std::max(1, std::min(4, 2048 / p.M));
        A.lpb = lpb;
        const size_t lds = 2ull * lpb * p.M * sizeof(float2);
        const long long nblocks = (A.n_lines + lpb - 1) / lpb;
        if (p.bluestein) {
            MVS_HIP_TRY(c, hipFuncSetAttribute((const void*)fft_lines_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
            hipLaunchKernelGGL(fft_lines_kernel<true>, dim3((unsigned)nblocks), dim3(256), lds, c->stream, A);
        } else {
            MVS_HIP_TRY(c, hipFuncSetAttribute((const void*)fft_lines_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
            hipLaunchKernelGGL(fft_lines_kernel<false>, dim3((unsigned)nblocks), dim3(256), lds, c->stream, A);
        }
        MVS_HIP_TRY(c, hipGetLastError());
    }
    return MVS_OK;
}
