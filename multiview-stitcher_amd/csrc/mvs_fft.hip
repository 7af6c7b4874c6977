// mvs_fft.hip -- batched complex64 line FFTs (LDS-staged Stockham autosort, gfx950).
//
// Replaces scipy.fft.fftn / ifftn inside skimage.registration.phase_cross_correlation as called from
// the reference's registration.py:422-431.  The transform sizes are the overlap-crop shapes themselves
// (registration.py:314-316: 104, 411, 27, 53, ...) and circular correlation depends on N, so there is no
// padding to friendlier sizes: powers of two run as a radix-2 Stockham autosort FFT, every other N as
// a Bluestein chirp-z transform built on the same Stockham core of size M = pow2 >= 2N-1 -- except short lines (N <= 64: the
// ~50-voxel axis of a binned overlap crop), which run as a direct DFT with the whole line in registers (dft_direct_kernel).
//
// A 3D transform is three passes over HBM (one per axis).  Each workgroup stages `lpb` lines in LDS
// (two ping-pong buffers), runs log2(M) butterfly stages there and writes the lines back, so every
// pass reads and writes each complex element exactly once.  Lines along y/z are taken `lpb` adjacent
// x positions at a time so that global accesses stay coalesced along x.
#include "mvs_fft.h"

#include <cmath>
#include <cstdlib>
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

// Stockham autosort stages on `lpb` lines of length M held in LDS: radix-4 passes (half the LDS round trips and
// barriers of radix 2) plus one radix-2 pass when log2 M is odd.  `tw` = exp(-2 pi i m / M), m < M/2, also in LDS.
// Returns the buffer holding the result.
__device__ __forceinline__ float2 tw_at(const float2* tw, int m, int half) {   // exp(-2 pi i m / M) for m < M
    const float2 w = tw[m & (half - 1)];
    return (m & half) ? make_float2(-w.x, -w.y) : w;
}

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

// element offset of line l of this pass
__device__ __forceinline__ long long line_base(const FftArgs& A, int l) {
    const int inner = (int)A.inner;
    return (long long)(l / inner) * A.outer_stride + (l % inner);
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

struct FftPlan {
    int n = 0, M = 0, log2M = 0;
    bool bluestein = false;
    float2* tw = nullptr;
    float2* chirp = nullptr;
    float2* bfft = nullptr;
};

std::map<long long, FftPlan> g_plans;   // key: device * 2^32 + n
std::mutex g_plan_mu;

// ---- direct DFT of short lines (n <= NMAX, not a power of two) --------------------------------------------------------------
// Bluestein pays two length-M transforms (M >= 2 n - 1: 128 points for n = 51) plus three chirp multiplications per line, all
// through LDS with a barrier per butterfly stage; for a short line the n^2 multiply-adds of the definition are cheaper.  A thread
// owns one line: its n samples sit in registers (the j loop is unrolled to NMAX and leaves at n), the outputs are produced four
// at a time, and the four factors exp(-2 pi i (j k mod n) / n) of a (k block, j) step are uniform across the wavefront: one
// 32-byte scalar load from a table laid out in exactly that order, used as scalar operands of the multiply-adds.  In place: all
// inputs are read before the first output is written.  Lines along y / z: the lines of a wavefront are adjacent x positions
// (coalesced as in the LDS kernel); lines along x are staged through LDS so that global accesses stay contiguous.
struct DftArgs {
    float2* data;
    long long n_lines;
    int n;
    long long stride, inner, outer_stride;
    int inverse;
    const float2* wtab;       // [ceil(n / 4)][round_up(n, 8)][4]: exp(-2 pi i (j k mod n) / n) for k = 4 kb + q (0 for j, k beyond n)
};

// KS wavefronts share a workgroup's 64 lines (and its table): wavefront w produces the output blocks w, w + KS, ... -- with one
// wavefront per line set a 256 x 256 x 51 crop gives every SIMD exactly one wavefront and nothing hides the LDS latency.
template <int NMAX, bool CONTIG, int KS>
__global__ __launch_bounds__(64 * KS) void dft_direct_kernel(DftArgs A) {
    // LDS: the factor table (uniform-address reads: broadcasts) and, for lines along x (CONTIG), a staging area [k][line] that turns
    // contiguous global accesses into one line per thread
    extern __shared__ float4 dft_lds[];
    const int n = A.n;
    const long long l0 = (long long)blockIdx.x * 64;
    const int nl = (int)min((long long)64, A.n_lines - l0);
    const int t = threadIdx.x & 63, ks = threadIdx.x >> 6;
    const bool inv = A.inverse != 0;
    const bool mine = t < nl;
    const int nkb = (n + 3) >> 2, nj = (n + 7) & ~7;
    float2* stage = reinterpret_cast<float2*>(dft_lds + (size_t)nkb * nj * 2);      // behind the table (lines along x only)
    float2 x[NMAX];
    long long base = 0;
    if (CONTIG) {
        // lines along x of a C-contiguous array: the nl lines of the workgroup are ONE contiguous run of nl * n samples
        const int total = nl * n;
        const float2* run = A.data + l0 * (long long)n;
        for (int i = threadIdx.x; i < total; i += 64 * KS) {
            const int line = i / n, k = i - line * n;
            stage[k * 65 + line] = run[i];
        }
        __syncthreads();
#pragma unroll
        for (int j = 0; j < NMAX; ++j) x[j] = (j < n) ? stage[j * 65 + t] : make_float2(0.f, 0.f);
        __syncthreads();
    } else {
        const long long l = l0 + (mine ? t : 0);
        base = (l / A.inner) * A.outer_stride + (l % A.inner);
#pragma unroll
        for (int j = 0; j < NMAX; ++j) {
            x[j] = make_float2(0.f, 0.f);
            if (j < n) x[j] = A.data[base + (long long)j * A.stride];
        }
    }
    {   // the factor table: nkb * nj entries of 4 complex factors (32 bytes)
        const float4* src = reinterpret_cast<const float4*>(A.wtab);
        const int n16 = nkb * nj * 2;
        for (int i = threadIdx.x; i < n16; i += 64 * KS) dft_lds[i] = src[i];
    }
    __syncthreads();
    if (inv) {                                           // IDFT(x) = conj(DFT(conj x)), unnormalised
#pragma unroll
        for (int j = 0; j < NMAX; ++j) x[j].y = -x[j].y;
    }
    for (int kb = __builtin_amdgcn_readfirstlane(ks); kb < nkb; kb += KS) {      // (ks is uniform per wavefront)
        // X = sum_j (x.re + i x.im) W = P + i Q with P = sum x.re W, Q = sum x.im W: both are plain packed multiply-adds of a
        // broadcast real with the (re, im) pair of the factor -- no sign flips, no swaps inside the loop
        typedef float f2 __attribute__((ext_vector_type(2)));
        f2 P[4], Q[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) { P[q] = (f2){0.f, 0.f}; Q[q] = (f2){0.f, 0.f}; }
        const float4* wrow = dft_lds + (size_t)kb * nj * 2;
#pragma unroll
        for (int jb = 0; jb < NMAX; jb += 8) {
            // 8 samples per uniform branch (samples and factors beyond n are zero)
            if (jb < n) {
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const float4 wa = wrow[(jb + u) * 2], wb = wrow[(jb + u) * 2 + 1];
                    const f2 w[4] = {(f2){wa.x, wa.y}, (f2){wa.z, wa.w}, (f2){wb.x, wb.y}, (f2){wb.z, wb.w}};
                    // v_pk_fma_f32 with the real (imaginary) part of the sample broadcast to both halves by the operand selects:
                    // the sample stays ONE register pair (written as intrinsics the compiler keeps (re, re) and (im, im) copies
                    // of all 64 samples and runs out of registers for a second wavefront per SIMD)
                    const f2 xp = (f2){x[jb + u].x, x[jb + u].y};
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel_hi:[0,1,1]" : "+v"(P[q]) : "v"(xp), "v"(w[q]));
                        asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[1,0,0] op_sel_hi:[1,1,1]" : "+v"(Q[q]) : "v"(xp), "v"(w[q]));
                    }
                }
            }
        }
        float2 acc[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) acc[q] = make_float2(P[q].x - Q[q].y, P[q].y + Q[q].x);
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int k = 4 * kb + q;
            if (k < n) {
                float2 v = acc[q];
                if (inv) v.y = -v.y;
                if (CONTIG) stage[k * 65 + t] = v;       // (all samples were read out of the staging area before the first barrier)
                else if (mine) A.data[base + (long long)k * A.stride] = v;
            }
        }
    }
    if (CONTIG) {
        __syncthreads();
        const int total = nl * n;
        float2* run = A.data + l0 * (long long)n;
        for (int i = threadIdx.x; i < total; i += 64 * KS) {
            const int line = i / n, k = i - line * n;
            run[i] = stage[k * 65 + line];
        }
    }
}

constexpr int kDirectMax = 64;
std::map<long long, float2*> g_direct_tabs;      // key: device * 2^32 + n

int get_direct_table(MvsContext* c, int n, const float2** out) {
    std::lock_guard<std::mutex> lock(g_plan_mu);
    const long long key = ((long long)mvs_hip_device(c->device) << 32) | (unsigned)n;
    auto it = g_direct_tabs.find(key);
    if (it != g_direct_tabs.end()) { *out = it->second; return MVS_OK; }
    const int nkb = (n + 3) / 4, nj = (n + 7) / 8 * 8;
    std::vector<float2> tab((size_t)nkb * nj * 4, make_float2(0.f, 0.f));
    for (int kb = 0; kb < nkb; ++kb)
        for (int j = 0; j < n; ++j)
            for (int q = 0; q < 4; ++q) {
                const int k = 4 * kb + q;
                if (k >= n) continue;
                const double a = -2.0 * M_PI * (double)(((long long)j * k) % n) / (double)n;
                tab[((size_t)kb * nj + j) * 4 + q] = make_float2((float)cos(a), (float)sin(a));
            }
    float2* d = nullptr;
    MVS_HIP_TRY(c, hipMalloc(&d, tab.size() * sizeof(float2)));
    MVS_HIP_TRY(c, hipMemcpy(d, tab.data(), tab.size() * sizeof(float2), hipMemcpyHostToDevice));
    g_direct_tabs[key] = d;
    *out = d;
    return MVS_OK;
}


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

// In-place 3D (or 2D when shape[0]==1) complex64 FFT of a C-contiguous (nz,ny,nx) array on c->stream.
// inverse: unnormalised conjugate transform (the caller applies 1/N where it matters).
int mvs_fft3_c2c(MvsContext* c, float2* data, const int64_t shape[3], bool inverse) {
    const long long nz = shape[0], ny = shape[1], nx = shape[2];
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
        if (!pow2n && n <= kDirectMax) {               // short lines: direct DFT in registers
            DftArgs D;
            D.data = data;
            D.n = n;
            D.inverse = inverse ? 1 : 0;
            int rcd = get_direct_table(c, n, &D.wtab);
            if (rcd) return rcd;
            if (axis == 2) { D.stride = 1; D.n_lines = nz * ny; D.inner = 1; D.outer_stride = nx; }
            else if (axis == 1) { D.stride = nx; D.n_lines = nz * nx; D.inner = nx; D.outer_stride = ny * nx; }
            else { D.stride = ny * nx; D.n_lines = ny * nx; D.inner = ny * nx; D.outer_stride = 0; }
            const long long nblocks = (D.n_lines + 63) / 64;
            const int nmax = n <= 32 ? 32 : 64;
            const size_t tab_bytes = (size_t)((n + 3) / 4) * ((n + 7) / 8 * 8) * 32, stage_bytes = (size_t)nmax * 65 * sizeof(float2);   // rows padded to 65: the transposing accesses spread over the banks
            const size_t lds = axis == 2 ? tab_bytes + stage_bytes : tab_bytes;
#define MVS_DFT(NM, CT, KS) do { MVS_HIP_TRY(c, hipFuncSetAttribute((const void*)dft_direct_kernel<NM, CT, KS>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)); \
                                 hipLaunchKernelGGL((dft_direct_kernel<NM, CT, KS>), dim3((unsigned)nblocks), dim3(64 * KS), lds, c->stream, D); } while (0)
            // 4 wavefronts per line set: measured 35.6 / 53.5 us (y or z / x lines of 51 samples, 256 x 256 of them) against 37.6 / 90 with 2
            if (axis == 2) { if (nmax == 32) MVS_DFT(32, true, 4); else MVS_DFT(64, true, 4); }
            else { if (nmax == 32) MVS_DFT(32, false, 4); else MVS_DFT(64, false, 4); }
#undef MVS_DFT
            MVS_HIP_TRY(c, hipGetLastError());
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
