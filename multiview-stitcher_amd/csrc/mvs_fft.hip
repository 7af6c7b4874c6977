// mvs_fft.hip -- batched complex64 line FFTs (LDS-staged Stockham autosort, gfx950).
//
// Replaces scipy.fft.fftn / ifftn inside skimage.registration.phase_cross_correlation as called from
// the reference's registration.py:422-431.  The transform sizes are the overlap-crop shapes themselves
// (registration.py:314-316: 104, 411, 27, 53, ...) and circular correlation depends on N, so there is no
// padding to friendlier sizes: powers of two run as a radix-2 Stockham autosort FFT, every other N as
// a Bluestein chirp-z transform built on the same Stockham core of size M = pow2 >= 2N-1.  Short lines -- powers of two up to
// 256 and Bluestein sizes M <= 256 (N <= 128: the ~50-voxel axis of a binned overlap crop) -- run in registers around one or two
// LDS exchanges instead (fft_reg2_kernel, bluestein_reg_kernel).
//
// A 3D transform is three passes over HBM (one per axis).  Each workgroup stages `lpb` lines in LDS
// (two ping-pong buffers), runs log2(M) butterfly stages there and writes the lines back, so every
// pass reads and writes each complex element exactly once.  Lines along y/z are taken `lpb` adjacent
// x positions at a time so that global accesses stay coalesced along x.
#include "mvs_fft.h"
#include "mvs_fft_dev.h"
#include "mvs_fft_reg.h"

#include <cmath>
#include <cstdlib>
#include <map>
#include <vector>

namespace {



// Stockham autosort stages on `lpb` lines of length M held in LDS: radix-4 passes (half the LDS round trips and
// barriers of radix 2) plus one radix-2 pass when log2 M is odd.  `tw` = exp(-2 pi i m / M), m < M/2, also in LDS.
// Returns the buffer holding the result.

__device__ __forceinline__ float2* stockham(float2* a, float2* b, int lpb, int M, int log2M, const float2* tw, bool inv) {
    const int half = M >> 1, quarter = M >> 2;
    const int log2q = log2M - 2;
    int s = 0;           // log2 of the current sub-transform length p
    for (; s + 2 <= log2M; s += 2) {
        const int p = 1 << s;
        const int tstep = M >> (s + 2);                    // twiddle index step: w = exp(-2 pi i k / (4 p))
        for (int t = threadIdx.x; t < (lpb << log2q); t += blockDim.x) {
            const int line = t >> log2q, i = t & (quarter - 1);
            const int k = i & (p - 1);
            const int j = ((i - k) << 2) + k;
            const float2* al = a + line * M;
            float2 w1 = tw_at(tw, k * tstep, half), w2 = tw_at(tw, 2 * k * tstep, half), w3 = tw_at(tw, 3 * k * tstep, half);
            if (inv) { w1.y = -w1.y; w2.y = -w2.y; w3.y = -w3.y; }
            const float2 a0 = al[i];
            const float2 u1 = cmul(w1, al[i + quarter]), u2 = cmul(w2, al[i + 2 * quarter]), u3 = cmul(w3, al[i + 3 * quarter]);
            const float2 s02 = make_float2(a0.x + u2.x, a0.y + u2.y), d02 = make_float2(a0.x - u2.x, a0.y - u2.y);
            const float2 s13 = make_float2(u1.x + u3.x, u1.y + u3.y), d13 = make_float2(u1.x - u3.x, u1.y - u3.y);
            // forward: -i * d13 = (d13.y, -d13.x); inverse: +i * d13 = (-d13.y, d13.x)
            const float2 r13 = inv ? make_float2(-d13.y, d13.x) : make_float2(d13.y, -d13.x);
            float2* bl = b + line * M;
            bl[j] = make_float2(s02.x + s13.x, s02.y + s13.y);
            bl[j + p] = make_float2(d02.x + r13.x, d02.y + r13.y);
            bl[j + 2 * p] = make_float2(s02.x - s13.x, s02.y - s13.y);
            bl[j + 3 * p] = make_float2(d02.x - r13.x, d02.y - r13.y);
        }
        __syncthreads();
        float2* tmp = a; a = b; b = tmp;
    }
    if (s < log2M) {     // one radix-2 pass left
        const int p = 1 << s;
        const int log2h = log2M - 1;
        for (int t = threadIdx.x; t < (lpb << log2h); t += blockDim.x) {
            const int line = t >> log2h, i = t & (half - 1);
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
    float2* tw = buf1 + (size_t)A.lpb * A.M;          // M/2 twiddles
    const int l0 = blockIdx.x * A.lpb;
    const int nl = min(A.lpb, (int)A.n_lines - l0);
    const bool inv = A.inverse != 0;
    const int M = A.M, log2M = A.log2M, n = A.n, lpb = A.lpb;
    for (int t = threadIdx.x; t < (M >> 1); t += blockDim.x) tw[t] = A.tw[t];

    // ---- load.  Along x (stride 1) a line is contiguous: threads run along k.  Along y / z the `lpb` lines of a
    // workgroup are adjacent x positions: threads run along the lines first so that global accesses coalesce; a
    // thread's line (and its base offset) is then the same in every iteration (lpb is a power of two <= 16). ----
    if (A.stride == 1) {
        for (int line = 0; line < lpb; ++line) {
            const long long base = (line < nl) ? line_base(A, l0 + line) : 0;
            for (int k = threadIdx.x; k < M; k += blockDim.x) {
                float2 v = make_float2(0.f, 0.f);
                if (line < nl && k < n) {
                    v = A.data[base + k];
                    if (BLUESTEIN) {
                        if (inv) v.y = -v.y;              // IDFT(x) = conj(DFT(conj x))
                        v = cmul(v, A.chirp[k]);
                    }
                }
                buf0[line * M + k] = v;
            }
        }
    } else {
        const int line = threadIdx.x & (lpb - 1), kstep = blockDim.x / lpb;
        const long long base = (line < nl) ? line_base(A, l0 + line) : 0;
        for (int k = threadIdx.x / lpb; k < M; k += kstep) {
            float2 v = make_float2(0.f, 0.f);
            if (line < nl && k < n) {
                v = A.data[base + (long long)k * A.stride];
                if (BLUESTEIN) {
                    if (inv) v.y = -v.y;
                    v = cmul(v, A.chirp[k]);
                }
            }
            buf0[line * M + k] = v;
        }
    }
    __syncthreads();

    float2* r;
    if (!BLUESTEIN) {
        r = stockham(buf0, buf1, lpb, M, log2M, tw, inv);
    } else {
        r = stockham(buf0, buf1, lpb, M, log2M, tw, false);
        float2* o = (r == buf0) ? buf1 : buf0;
        for (int t = threadIdx.x; t < (lpb << log2M); t += blockDim.x) {
            const int k = t & (M - 1);
            r[t] = cmul(r[t], A.bfft[k]);
        }
        __syncthreads();
        r = stockham(r, o, lpb, M, log2M, tw, true);
    }

    // ---- store ----
    if (A.stride == 1) {
        for (int line = 0; line < nl; ++line) {
            const long long base = line_base(A, l0 + line);
            for (int k = threadIdx.x; k < n; k += blockDim.x) {
                float2 v = r[line * M + k];
                if (BLUESTEIN) {
                    v = cmul(v, A.chirp[k]);
                    if (inv) v.y = -v.y;
                }
                A.data[base + k] = v;
            }
        }
    } else {
        const int line = threadIdx.x & (lpb - 1), kstep = blockDim.x / lpb;
        if (line < nl) {
            const long long base = line_base(A, l0 + line);
            for (int k = threadIdx.x / lpb; k < n; k += kstep) {
                float2 v = r[line * M + k];
                if (BLUESTEIN) {
                    v = cmul(v, A.chirp[k]);
                    if (inv) v.y = -v.y;
                }
                A.data[base + (long long)k * A.stride] = v;
            }
        }
    }
}

// ---- short power-of-two lines (M = 64, 128, 256) in registers -----------------------------------------------------------------
// M = R1 x R2: a thread gathers R1 samples of its line at stride R2, transforms them in registers (radix 4 x 4 / 4 x 2 / 4), applies
// the twiddles exp(-2 pi i j' p / M) and leaves them in LDS; after ONE barrier a thread picks up the R2 values of one residue p and
// transforms those: X[p + R1 k'].  Two register transforms, one LDS round trip and one barrier replace the log4(M) butterfly
// stages of the Stockham kernel with their LDS round trip, barrier and index arithmetic each (28.5 -> 13.7 us for a
// 256-point pass over a 256 x 256 x 51 crop).  The inverse is the forward transform of the input with real and imaginary parts
// swapped, swapped back on output.
// workgroup reduction (256 threads) and the write of the workgroup's partials
__device__ __forceinline__ void peak_flush(Peak2 p, const FftArgs& A) {
    __shared__ float sv[2][4];
    __shared__ long long si[2][4];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int k = 0; k < 2; ++k) {
        for (int off = 32; off > 0; off >>= 1) {
            const float ob = __shfl_down(p.v[k], off);
            const long long oi = __shfl_down(p.i[k], off);
            if (ob > p.v[k] || (ob == p.v[k] && oi < p.i[k])) { p.v[k] = ob; p.i[k] = oi; }
        }
        if (lane == 0) { sv[k][wave] = p.v[k]; si[k][wave] = p.i[k]; }
    }
    __syncthreads();
    if (threadIdx.x == 0) {
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            for (int w = 1; w < 4; ++w)
                if (sv[k][w] > p.v[k] || (sv[k][w] == p.v[k] && si[k][w] < p.i[k])) { p.v[k] = sv[k][w]; p.i[k] = si[k][w]; }
            A.peak_val[k][blockIdx.x] = p.v[k];
            A.peak_idx[k][blockIdx.x] = p.i[k];
        }
    }
}

template <int R1, int R2>
__global__ __launch_bounds__(256) void fft_reg2_kernel(FftArgs A) {
    constexpr int M = R1 * R2, TPL = R1 > R2 ? R1 : R2, LPB = 256 / TPL;      // threads per line, lines per workgroup
    constexpr int PS = R2 + 1, LS = R1 * PS + 1;                                // padded strides (float2): residue p, line
    __shared__ float2 ex[LPB * LS];
    __shared__ float2 tw[M / 2];
    for (int t = threadIdx.x; t < M / 2; t += 256) tw[t] = A.tw[t];
    const bool inv = A.inverse != 0;
    const bool contig = A.stride == 1;
    // along x threads run along the line first, along y / z along the lines (adjacent x positions) first: coalesced either way
    const int line = contig ? (int)threadIdx.x / TPL : (int)threadIdx.x % LPB;
    const int idx = contig ? (int)threadIdx.x % TPL : (int)threadIdx.x / LPB;
    long long l = (long long)blockIdx.x * LPB + line;
    bool live = l < A.n_lines;
    if (A.xp_pair) {
        // The cross power of line (kz, ky) needs the spectrum of its own line AND of the partner line (-kz, -ky), whose cross power
        // needs the same two lines: line slots 2 p and 2 p + 1 of a workgroup take such a pair, so every element of the packed
        // spectrum is fetched from HBM once instead of twice (53 -> 27 MB per 51 x 256 x 256 pass; the partner of a line used to be
        // handled by a workgroup half a launch away).  Pairs are numbered over the canonical lines: row kz = 0 and, for even nz,
        // row nz / 2 are their own partner rows (canonical: ky <= ny / 2), rows 1 .. (nz - 1) / 2 pair with rows nz - kz whole.
        const int nyl = A.xp_ny, nzl = A.xp_nz, h = nyl / 2 + 1, F = (nzl - 1) / 2;
        long long q = (long long)blockIdx.x * (LPB / 2) + (line >> 1);
        int kz = -1, ky = 0;
        if (q < h) { kz = 0; ky = (int)q; }
        else {
            q -= h;
            if (q < (long long)F * nyl) { kz = 1 + (int)(q / nyl); ky = (int)(q % nyl); }
            else {
                q -= (long long)F * nyl;
                if (!(nzl & 1) && q < h) { kz = nzl / 2; ky = (int)q; }
            }
        }
        live = kz >= 0;
        if (live && (line & 1)) {
            const int my = ky ? nyl - ky : 0, mz = kz ? nzl - kz : 0;
            if (my == ky && mz == kz) live = false;          // its own partner: the even slot has it
            kz = mz; ky = my;
        }
        l = live ? (long long)kz * nyl + ky : 0;
    }
    const long long base = live ? line_base(A, (int)l) : 0;
    float2* row = ex + line * LS;
    Peak2 pk;
    peak_init(pk);
    XpLine xl;
    if (A.xp_src) xl = xp_line(A, live ? l : 0);
    __syncthreads();                                                             // (twiddles)
    if (idx < R2) {                                                              // pass 1: j' = idx
        float2 v[R1];
#pragma unroll
        for (int q = 0; q < R1; ++q) {
            v[q] = make_float2(0.f, 0.f);
            if (live) v[q] = A.xp_src ? xp_load(A, xl, base, q * R2 + idx) : fft_load(A, base + (long long)(q * R2 + idx) * A.stride);
            if (inv) v[q] = make_float2(v[q].y, v[q].x);
        }
        dft_reg<R1>(v);
#pragma unroll
        for (int p = 0; p < R1; ++p) {
            const float2 w = p == 0 ? make_float2(1.f, 0.f) : tw_at(tw, idx * p, M / 2);
            row[p * PS + idx] = p == 0 ? v[0] : cmul(v[p], w);
        }
    }
    __syncthreads();
    if (idx < R1) {                                                              // pass 2: p = idx
        float2 u[R2];
#pragma unroll
        for (int j = 0; j < R2; ++j) u[j] = row[idx * PS + j];
        dft_reg<R2>(u);
        if (live) {
#pragma unroll
            for (int k = 0; k < R2; ++k) {
                const float2 o = inv ? make_float2(u[k].y, u[k].x) : u[k];
                const long long at = base + (long long)(idx + R1 * k) * A.stride;
                if (A.peak_val[0]) peak_add(pk, o, at); else A.data[at] = o;
            }
        }
    }
    if (A.peak_val[0]) peak_flush(pk, A);
}

// Bluestein's chirp-z for short lines on the same register transforms (n <= 128: M = 64, 128 or 256 points): sample x chirp ->
// forward transform (R1 then R2) -> x spectrum of the chirp filter -> inverse transform with the factors the other way round (R2
// then R1: a thread ends the forward transform holding the residue class p mod R1, which is exactly what the first pass of an
// (R2, R1) transform wants, so nothing moves in between) -> x chirp.  Two LDS exchanges per line instead of 2 log4(M) + 3.
template <int R1, int R2>
__global__ __launch_bounds__(256) void bluestein_reg_kernel(FftArgs A) {
    constexpr int M = R1 * R2, TPL = R1 > R2 ? R1 : R2, LPB = 256 / TPL;
    constexpr int PS = TPL + 1, LS = TPL * PS + 1;                               // padded strides (float2) that fit both exchanges
    __shared__ float2 ex[LPB * LS];
    __shared__ float2 tw[M / 2];
    for (int t = threadIdx.x; t < M / 2; t += 256) tw[t] = A.tw[t];
    const int n = A.n;
    const bool inv = A.inverse != 0;
    const bool contig = A.stride == 1;
    const int line = contig ? (int)threadIdx.x / TPL : (int)threadIdx.x % LPB;
    const int idx = contig ? (int)threadIdx.x % TPL : (int)threadIdx.x / LPB;
    const long long l = (long long)blockIdx.x * LPB + line;
    const bool live = l < A.n_lines;
    const long long base = live ? line_base(A, (int)l) : 0;
    float2* row = ex + line * LS;
    Peak2 pk;
    peak_init(pk);
    XpLine xl;
    if (A.xp_src) xl = xp_line(A, live ? l : 0);
    __syncthreads();                                                             // (twiddles)
    if (idx < R2) {                                                              // forward, pass 1: j' = idx
        float2 v[R1];
#pragma unroll
        for (int q = 0; q < R1; ++q) {
            const int j = q * R2 + idx;
            v[q] = make_float2(0.f, 0.f);
            if (live && j < n) {
                float2 x = A.xp_src ? xp_load(A, xl, base, j) : fft_load(A, base + (long long)j * A.stride);
                if (inv) x.y = -x.y;                                             // IDFT(x) = conj(DFT(conj x))
                v[q] = cmul(x, A.chirp[j]);
            }
        }
        dft_reg<R1>(v);
#pragma unroll
        for (int p = 0; p < R1; ++p) row[p * PS + idx] = p == 0 ? v[0] : cmul(v[p], tw_at(tw, idx * p, M / 2));
    }
    __syncthreads();
    float2 u[R2];
    if (idx < R1) {                                                              // forward, pass 2: p = idx holds X[p + R1 k]
#pragma unroll
        for (int j = 0; j < R2; ++j) u[j] = row[idx * PS + j];
        dft_reg<R2>(u);
        // x filter spectrum (scaled by 1 / M on the host); swap re / im: the inverse transform is a forward one of the swapped input
#pragma unroll
        for (int k = 0; k < R2; ++k) {
            const float2 y = cmul(u[k], A.bfft[idx + R1 * k]);
            u[k] = make_float2(y.y, y.x);
        }
        // inverse, pass 1 with the factors swapped (R2 then R1): this thread is j'' = idx and holds the samples q R1 + idx, q < R2
        dft_reg<R2>(u);
    }
    __syncthreads();                                                             // everybody has read the first exchange
    if (idx < R1) {
#pragma unroll
        for (int p = 0; p < R2; ++p) row[p * PS + idx] = p == 0 ? u[0] : cmul(u[p], tw_at(tw, idx * p, M / 2));
    }
    __syncthreads();
    if (idx < R2) {                                                              // inverse, pass 2: p'' = idx -> Y[p'' + R2 k], k < R1
        float2 w[R1];
#pragma unroll
        for (int j = 0; j < R1; ++j) w[j] = row[idx * PS + j];
        dft_reg<R1>(w);
        if (live) {
#pragma unroll
            for (int k = 0; k < R1; ++k) {
                const int m = idx + R2 * k;
                if (m < n) {
                    float2 y = cmul(make_float2(w[k].y, w[k].x), A.chirp[m]);
                    if (inv) y.y = -y.y;
                    const long long at = base + (long long)m * A.stride;
                    if (A.peak_val[0]) peak_add(pk, y, at); else A.data[at] = y;
                }
            }
        }
    }
    if (A.peak_val[0]) peak_flush(pk, A);
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

// ---- lengths beyond the LDS core (powers of two > 4096, other lengths > 2048): four-step transform in device memory ----
// A line of length M = M1 x M2 (both powers of two <= 4096) is copied into a contiguous scratch line and seen as a matrix
// [j1][j2], j = j1 M2 + j2: (A) M2 transforms of length M1 along j1, (B) twiddle exp(-+2 pi i k1 j2 / M), (C) M1 transforms of
// length M2 along j2 -- all by the LDS kernel above, reading and writing the scratch array once per step; X[k2 M1 + k1] ends up
// at [k1][k2].  A power-of-two length is scattered back from that transposed order.  Any other length runs as Bluestein's
// chirp-z on this core: M >= 2 n - 1, the spectrum of the chirp filter is stored in the SAME transposed order, and the inverse
// transform runs the three steps backwards (C', B', A'), which returns natural order -- no transpose at all.
struct BigPlan {
    int n = 0, M = 0, M1 = 0, M2 = 0;
    bool pow2 = false;
    float2* chirp = nullptr;     // n (Bluestein)
    float2* bfft_t = nullptr;    // M, transposed order, scaled by 1 / M (Bluestein)
};
std::map<long long, BigPlan> g_big_plans;

__global__ __launch_bounds__(256) void big_gather_kernel(const float2* __restrict__ data, float2* __restrict__ scr, long long l0, int nl, int n, int M,
                                                         long long stride, long long inner, long long outer_stride,
                                                         const float2* __restrict__ chirp, int inv) {
    const long long total = (long long)nl * M;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        // lines fastest when they are adjacent in memory (stride != 1), positions fastest otherwise: coalesced either way
        int line, k;
        if (stride == 1) { line = (int)(i / M); k = (int)(i - (long long)line * M); }
        else { k = (int)(i / nl); line = (int)(i - (long long)k * nl); }
        float2 v = make_float2(0.f, 0.f);
        if (k < n) {
            const long long l = l0 + line;
            v = data[(l / inner) * outer_stride + (l % inner) + (long long)k * stride];
            if (chirp) {
                if (inv) v.y = -v.y;              // IDFT(x) = conj(DFT(conj x))
                v = cmul(v, chirp[k]);
            }
        }
        scr[(long long)line * M + k] = v;
    }
}

__global__ __launch_bounds__(256) void big_twiddle_kernel(float2* __restrict__ scr, long long total, int M, int M2, int log2M2, int conj_tw) {
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const int pos = (int)(i & (M - 1));
        const int k1 = pos >> log2M2, j2 = pos & (M2 - 1);
        const int t = (int)(((long long)k1 * j2) & (M - 1));          // exp(-2 pi i t / M), the product taken modulo M exactly
        float sn, cs;
        sincospif(-2.f * (float)t / (float)M, &sn, &cs);
        float2 w = make_float2(cs, conj_tw ? -sn : sn);
        scr[i] = cmul(scr[i], w);
    }
}

__global__ __launch_bounds__(256) void big_mul_kernel(float2* __restrict__ scr, const float2* __restrict__ b, long long total, int M) {
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x)
        scr[i] = cmul(scr[i], b[i & (M - 1)]);
}

__global__ __launch_bounds__(256) void big_scatter_kernel(float2* __restrict__ data, const float2* __restrict__ scr, long long l0, int nl, int n, int M,
                                                          int M1, int M2, long long stride, long long inner, long long outer_stride,
                                                          const float2* __restrict__ chirp, int inv, int transposed) {
    const long long total = (long long)nl * n;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        int line, k;
        if (stride == 1) { line = (int)(i / n); k = (int)(i - (long long)line * n); }
        else { k = (int)(i / nl); line = (int)(i - (long long)k * nl); }
        const int pos = transposed ? (k % M1) * M2 + (k / M1) : k;      // X[k2 M1 + k1] sits at [k1][k2]
        float2 v = scr[(long long)line * M + pos];
        if (chirp) {
            v = cmul(v, chirp[k]);
            if (inv) v.y = -v.y;
        }
        const long long l = l0 + line;
        data[(l / inner) * outer_stride + (l % inner) + (long long)k * stride] = v;
    }
}

int get_big_plan(MvsContext* c, int n, BigPlan* out) {
    std::lock_guard<std::mutex> lock(g_plan_mu);
    const long long key = ((long long)mvs_hip_device(c->device) << 32) | (unsigned)n;
    auto it = g_big_plans.find(key);
    if (it != g_big_plans.end()) { *out = it->second; return MVS_OK; }
    BigPlan p;
    p.n = n;
    p.pow2 = (n & (n - 1)) == 0;
    int M = 1;
    if (p.pow2) M = n;
    else while (M < 2 * n - 1) M <<= 1;
    int log2M = 0;
    while ((1 << log2M) < M) ++log2M;
    p.M = M;
    p.M1 = 1 << ((log2M + 1) / 2);
    p.M2 = M / p.M1;
    if (!p.pow2) {
        std::vector<double> wr(n), wi(n);
        std::vector<float2> chirp(n);
        for (int k = 0; k < n; ++k) {
            const long long k2 = ((long long)k * k) % (2LL * n);
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
        std::vector<float2> bt(M);
        for (int k1 = 0; k1 < p.M1; ++k1)
            for (int k2 = 0; k2 < p.M2; ++k2) {
                const int k = k2 * p.M1 + k1;
                bt[(size_t)k1 * p.M2 + k2] = make_float2((float)(br[k] / M), (float)(bi[k] / M));
            }
        MVS_HIP_TRY(c, hipMalloc(&p.chirp, (size_t)n * sizeof(float2)));
        MVS_HIP_TRY(c, hipMemcpy(p.chirp, chirp.data(), (size_t)n * sizeof(float2), hipMemcpyHostToDevice));
        MVS_HIP_TRY(c, hipMalloc(&p.bfft_t, (size_t)M * sizeof(float2)));
        MVS_HIP_TRY(c, hipMemcpy(p.bfft_t, bt.data(), (size_t)M * sizeof(float2), hipMemcpyHostToDevice));
    }
    g_big_plans[key] = p;
    *out = p;
    return MVS_OK;
}

// batched power-of-two line transforms (length <= 4096) of a scratch array through the LDS kernel
int launch_pow2_lines(MvsContext* c, float2* data, int len, long long n_lines, long long stride, long long inner, long long outer_stride, bool inverse) {
    FftPlan p;
    int rc = get_plan(c, len, &p);
    if (rc) return rc;
    FftArgs A;
    A.data = data;
    A.n = len; A.M = p.M; A.log2M = p.log2M;
    A.inverse = inverse ? 1 : 0;
    A.tw = p.tw; A.chirp = nullptr; A.bfft = nullptr;
    A.stride = stride; A.n_lines = n_lines; A.inner = inner; A.outer_stride = outer_stride;
    int lpb = std::max(1, std::min(8, (stride == 1 ? 2048 : 4096) / p.M));
    A.lpb = lpb;
    const size_t lds = (2ull * lpb * p.M + p.M / 2 + 1) * sizeof(float2);
    const long long nblocks = (n_lines + lpb - 1) / lpb;
    if (nblocks > 0x7fffffffLL) return mvs_fail(c, MVS_ERR_UNSUPPORTED, "FFT: too many lines");
    MVS_HIP_TRY(c, hipFuncSetAttribute((const void*)fft_lines_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL(fft_lines_kernel<false>, dim3((unsigned)nblocks), dim3(256), lds, c->stream, A);
    MVS_HIP_TRY(c, hipGetLastError());
    return MVS_OK;
}

int fft_axis_big(MvsContext* c, float2* data, int n, long long n_lines, long long stride, long long inner, long long outer_stride, bool inverse) {
    if (n > (1 << 22)) return mvs_fail(c, MVS_ERR_UNSUPPORTED, "FFT length %d > 4194304 not supported", n);
    BigPlan p;
    int rc = get_big_plan(c, n, &p);
    if (rc) return rc;
    int log2M2 = 0;
    while ((1 << log2M2) < p.M2) ++log2M2;
    const long long batch = std::max<long long>(1, std::min<long long>(n_lines, ((long long)256 << 20) / ((long long)p.M * 8)));
    void* scr_v = nullptr;
    rc = mvs_malloc(c->device, (uint64_t)batch * p.M * sizeof(float2), &scr_v);
    if (rc) return rc;
    float2* scr = (float2*)scr_v;
    auto grid = [](long long total) { return (unsigned)std::min<long long>((total + 255) / 256, 256 * 32); };
    for (long long l0 = 0; l0 < n_lines && !rc; l0 += batch) {
        const int nl = (int)std::min<long long>(batch, n_lines - l0);
        const long long total = (long long)nl * p.M;
        const bool blu = !p.pow2;
        // (when the lines are adjacent in memory, the batch must not straddle an outer block: the gather / scatter kernels
        // address every line on its own, so any batch is fine)
        hipLaunchKernelGGL(big_gather_kernel, dim3(grid(total)), dim3(256), 0, c->stream, data, scr, l0, nl, n, p.M, stride, inner, outer_stride,
                           blu ? p.chirp : nullptr, inverse ? 1 : 0);
        const bool inv1 = blu ? false : inverse;             // Bluestein: forward core transform first, inverse second
        rc = launch_pow2_lines(c, scr, p.M1, (long long)nl * p.M2, p.M2, p.M2, p.M, inv1);                  // (A)
        if (rc) break;
        hipLaunchKernelGGL(big_twiddle_kernel, dim3(grid(total)), dim3(256), 0, c->stream, scr, total, p.M, p.M2, log2M2, inv1 ? 1 : 0);   // (B)
        rc = launch_pow2_lines(c, scr, p.M2, (long long)nl * p.M1, 1, 1, p.M2, inv1);                       // (C)
        if (rc) break;
        if (blu) {
            hipLaunchKernelGGL(big_mul_kernel, dim3(grid(total)), dim3(256), 0, c->stream, scr, p.bfft_t, total, p.M);
            rc = launch_pow2_lines(c, scr, p.M2, (long long)nl * p.M1, 1, 1, p.M2, true);                   // (C')
            if (rc) break;
            hipLaunchKernelGGL(big_twiddle_kernel, dim3(grid(total)), dim3(256), 0, c->stream, scr, total, p.M, p.M2, log2M2, 1);          // (B')
            rc = launch_pow2_lines(c, scr, p.M1, (long long)nl * p.M2, p.M2, p.M2, p.M, true);              // (A')
            if (rc) break;
        }
        hipLaunchKernelGGL(big_scatter_kernel, dim3(grid((long long)nl * n)), dim3(256), 0, c->stream, data, scr, l0, nl, n, p.M, p.M1, p.M2, stride,
                           inner, outer_stride, blu ? p.chirp : nullptr, inverse ? 1 : 0, blu ? 0 : 1);
    }
    hipError_t e = hipGetLastError();
    mvs_free(c->device, scr_v);          // (stream-ordered pool: the block is only handed out again to work queued after this)
    if (rc) return rc;
    if (e != hipSuccess) return mvs_fail(c, MVS_ERR_HIP, "FFT four-step launch failed: %s", hipGetErrorString(e));
    return MVS_OK;
}

}  // namespace

// device table exp(-2 pi i m / n), m < n / 2, of a power-of-two length (the plan cache of this file), for kernels of other units
int mvs_fft_twiddles(MvsContext* c, int n, const float2** tw) {
    if (n < 2 || (n & (n - 1)) != 0) return mvs_fail(c, MVS_ERR_INVALID_ARG, "mvs_fft_twiddles: n = %d is not a power of two", n);
    FftPlan p;
    const int rc = get_plan(c, n, &p);
    if (rc) return rc;
    *tw = p.tw;
    return MVS_OK;
}

// In-place 3D (or 2D when shape[0]==1) complex64 FFT of a C-contiguous (nz,ny,nx) array on c->stream.
// inverse: unnormalised conjugate transform (the caller applies 1/N where it matters).
// does a line of n samples run on the register kernels (which carry the MvsFftFuse options)?
bool mvs_fft_reg_length(int n) {
    if (n < 2) return false;
    if (mvs_dft_line_length(n)) return true;
    if ((n & (n - 1)) == 0) return n == 64 || n == 128 || n == 256;
    int M = 1;
    while (M < 2 * n - 1) M <<= 1;
    return M >= 64 && M <= 256;
}

int mvs_fft3_c2c(MvsContext* c, float2* data, const int64_t shape[3], bool inverse, MvsFftFuse* fuse) {
    const long long nz = shape[0], ny = shape[1], nx = shape[2];
    int first_axis = -1, last_axis = -1;
    for (int axis = 2; axis >= 0; --axis)
        if (shape[axis] > 1) { if (first_axis < 0) first_axis = axis; last_axis = axis; }
    if (fuse) { fuse->n_peak = 0; fuse->src_used = false; fuse->xp_used = false; }
    for (int axis = 2; axis >= 0; --axis) {
        const int n = (int)shape[axis];
        if (n == 1) continue;
        const bool pow2n = (n & (n - 1)) == 0;
        if (n > 4096 || (!pow2n && n > 2048)) {      // beyond the LDS core (Bluestein of n > 2048 needs M = 8192 points per line)
            int rcb;
            if (axis == 2) rcb = fft_axis_big(c, data, n, nz * ny, 1, 1, nx, inverse);
            else if (axis == 1) rcb = fft_axis_big(c, data, n, nz * nx, nx, nx, ny * nx, inverse);
            else rcb = fft_axis_big(c, data, n, ny * nx, ny * nx, ny * nx, 0, inverse);
            if (rcb) return rcb;
            continue;
        }
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
        // lines per workgroup (measured on 51 x 256 x 256: 8 beats 4 and 16 for both kinds of pass; LDS 33 KB -> 4 workgroups / CU)
        int lpb = std::max(1, std::min(8, 4096 / p.M));
        if (axis == 2) lpb = std::max(1, std::min(8, 2048 / p.M));
        A.lpb = lpb;
        const size_t lds = (2ull * lpb * p.M + p.M / 2 + 1) * sizeof(float2);   // two line buffers + the twiddles
        const long long nblocks = (A.n_lines + lpb - 1) / lpb;
        const bool reg_pow2 = !p.bluestein && (n == 64 || n == 128 || n == 256);
        const bool reg_blue = p.bluestein && p.M <= 256 && p.M >= 64;
        const bool reg_line = mvs_dft_line_length(n) && !c->fft_no_line;
        if (reg_pow2 || reg_blue || reg_line) {
            const int lpw = reg_line ? 64 : ((reg_pow2 && n == 64) || (reg_blue && p.M == 64)) ? 32 : 16;      // lines per workgroup
            unsigned grid = (unsigned)((A.n_lines + lpw - 1) / lpw);
            // the fusions at the two ends of the transform (MvsFftFuse)
            if (fuse && axis == first_axis && fuse->re_src && fuse->im_src) {
                A.re_src = fuse->re_src;
                A.im_src = fuse->im_src;
                fuse->src_used = true;
            }
            if (fuse && axis == first_axis && axis == 2 && inverse && fuse->xp_src && fuse->xp_p2) {
                A.xp_src = fuse->xp_src; A.xp_p2 = fuse->xp_p2;
                A.xp_ny = (int)ny; A.xp_nz = (int)nz;
                A.xp_sel_a = fuse->xp_sel_a; A.xp_sel_b = fuse->xp_sel_b;
                fuse->xp_used = true;
                if (reg_pow2 && lpw % 2 == 0 && !c->fft_no_pair) {      // partner lines side by side (fft_reg2_kernel)
                    const long long h = ny / 2 + 1, ncanon = h + ((nz - 1) / 2) * ny + ((nz % 2 == 0) ? h : 0);
                    A.xp_pair = 1;
                    grid = (unsigned)((ncanon + lpw / 2 - 1) / (lpw / 2));
                }
            }
            if (fuse && axis == last_axis && fuse->peak_val[0] && (long long)grid <= fuse->peak_cap) {
                for (int k = 0; k < 2; ++k) { A.peak_val[k] = fuse->peak_val[k]; A.peak_idx[k] = fuse->peak_idx[k]; }
                fuse->n_peak = (int)grid;
            }
            if (reg_line) {
                // short composite lines: the whole line in the registers of one thread (mvs_dft_small.h)
                if (!(n <= 44 ? mvs_launch_dft_line_lo(c, A, grid) : mvs_launch_dft_line_hi(c, A, grid))) return mvs_fail(c, MVS_ERR_UNSUPPORTED, "FFT: no whole-line kernel for n = %d", n);
            } else if (reg_pow2) {
                // short power-of-two lines: two register transforms around one LDS exchange
                if (n == 256) hipLaunchKernelGGL((fft_reg2_kernel<16, 16>), dim3(grid), dim3(256), 0, c->stream, A);
                else if (n == 128) hipLaunchKernelGGL((fft_reg2_kernel<16, 8>), dim3(grid), dim3(256), 0, c->stream, A);
                else hipLaunchKernelGGL((fft_reg2_kernel<8, 8>), dim3(grid), dim3(256), 0, c->stream, A);
            } else {
                // ... and Bluestein lines of up to 128 samples
                if (p.M == 256) hipLaunchKernelGGL((bluestein_reg_kernel<16, 16>), dim3(grid), dim3(256), 0, c->stream, A);
                else if (p.M == 128) hipLaunchKernelGGL((bluestein_reg_kernel<16, 8>), dim3(grid), dim3(256), 0, c->stream, A);
                else hipLaunchKernelGGL((bluestein_reg_kernel<8, 8>), dim3(grid), dim3(256), 0, c->stream, A);
            }
        } else if (p.bluestein) {
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
