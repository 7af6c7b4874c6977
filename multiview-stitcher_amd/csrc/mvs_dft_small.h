// mvs_dft_small.h -- internal: whole-line DFTs of short composite lengths in the registers of ONE thread.
//
// The transform lengths of the phase correlation are the overlap-crop shapes themselves (registration.py:314-316), e.g. the 51
// samples across the binned overlap of two 512^3 tiles.  Bluestein on the power-of-two register transforms pads such a line to
// 128 points and runs two transforms plus three chirp products (bluestein_reg_kernel: 43 us per pass of a 51 x 256 x 256 crop at
// 0.15 of the HBM roofline).  A thread can instead hold the whole line -- N <= 64 complex samples are 2 N registers -- and run a
// mixed-radix transform on it: the length is split into two factors (prime-factor map when they are coprime: no twiddles at all;
// Cooley-Tukey otherwise), recursively, down to leaves that are the radix-4 / 8 / 16 butterflies of mvs_fft.hip or DENSE DFTs of
// a prime length P <= 19 in their symmetric form (x_j +- x_{P-j}: (P-1)^2 real multiply-adds instead of 4 P^2).  Every index and
// every twiddle factor is a compile-time constant (static_for over integral_constant: the roots of unity are literals in the
// instruction stream, nothing is loaded from a table), the arrays live in registers, nothing crosses lanes, and for lines along
// y / z the loads and stores of a wavefront are 512 contiguous bytes per sample index.
#pragma once
#include <hip/hip_runtime.h>

#include <type_traits>

namespace sdft {

template <int I, int E, class F>
__host__ __device__ __forceinline__ void static_for(F&& f) {
    if constexpr (I < E) {
        f(std::integral_constant<int, I>{});
        static_for<I + 1, E>(f);
    }
}

// ---- compile-time roots of unity -------------------------------------------------------------------------------------------
constexpr double kPi = 3.14159265358979323846264338327950288;
constexpr double cx_sin(double x) {      // |x| <= pi: Taylor series, 26 terms (largest term ~5: 1e-15 absolute)
    double x2 = x * x, term = x, sum = x;
    for (int i = 1; i < 26; ++i) { term *= -x2 / (double)((2 * i) * (2 * i + 1)); sum += term; }
    return sum;
}
constexpr double cx_cos(double x) {
    double x2 = x * x, term = 1.0, sum = 1.0;
    for (int i = 1; i < 26; ++i) { term *= -x2 / (double)((2 * i - 1) * (2 * i)); sum += term; }
    return sum;
}
constexpr double cx_angle(int m, int M) {      // 2 pi (m mod M) / M folded into (-pi, pi]
    m %= M;
    if (m < 0) m += M;
    double x = 2.0 * kPi * (double)m / (double)M;
    return 2 * m > M ? x - 2.0 * kPi : x;
}
// exact values where the angle is a multiple of pi / 2 (the series would leave 1e-17 instead of 0)
constexpr float root_cos(int m, int M) {
    m = ((m % M) + M) % M;
    if (m == 0) return 1.f;
    if (2 * m == M) return -1.f;
    if (4 * m == M || 4 * m == 3 * M) return 0.f;
    return (float)cx_cos(cx_angle(m, M));
}
constexpr float root_sin(int m, int M) {
    m = ((m % M) + M) % M;
    if (m == 0 || 2 * m == M) return 0.f;
    if (4 * m == M) return 1.f;
    if (4 * m == 3 * M) return -1.f;
    return (float)cx_sin(cx_angle(m, M));
}

constexpr int gcd(int a, int b) { return b == 0 ? a : gcd(b, a % b); }
constexpr int inv_mod(int a, int m) {      // a^-1 mod m (gcd(a, m) == 1)
    a %= m;
    for (int x = 1; x < m; ++x)
        if ((a * x) % m == 1) return x;
    return 1;
}
constexpr bool is_prime(int n) {
    if (n < 2) return false;
    for (int d = 2; d * d <= n; ++d)
        if (n % d == 0) return false;
    return true;
}
constexpr bool is_leaf(int n) { return n == 1 || n == 2 || n == 4 || n == 8 || n == 16 || (is_prime(n) && n <= 19); }
// first factor of a composite length: a leaf, preferably coprime to the rest (prime-factor map: no twiddles)
constexpr int pick_factor(int n) {
    if (n == 16) return 4;
    constexpr int cand[] = {16, 8, 4, 19, 17, 13, 11, 7, 5, 3, 2};
    for (int a : cand)
        if (n % a == 0 && a < n && gcd(a, n / a) == 1) return a;
    for (int a : cand)
        if (n % a == 0 && a < n) return a;
    return 1;
}
constexpr bool supported(int n) {      // every prime factor <= 19
    if (n < 2) return false;
    for (int p = 2; p <= 19; ++p)
        while (n % p == 0) n /= p;
    return n == 1;
}

// ---- complex helpers ---------------------------------------------------------------------------------------------------------
__host__ __device__ __forceinline__ float2 add(float2 a, float2 b) { return make_float2(a.x + b.x, a.y + b.y); }
__host__ __device__ __forceinline__ float2 sub(float2 a, float2 b) { return make_float2(a.x - b.x, a.y - b.y); }
// v * W_M^m = v * (cos - i sin)(2 pi m / M), factor known at compile time
template <int M, int m>
__host__ __device__ __forceinline__ float2 mul_root(float2 v) {
    constexpr float c = root_cos(m, M), s = root_sin(m, M);
    if constexpr (s == 0.f && c == 1.f) return v;
    else if constexpr (s == 0.f && c == -1.f) return make_float2(-v.x, -v.y);
    else if constexpr (c == 0.f && s == 1.f) return make_float2(v.y, -v.x);
    else if constexpr (c == 0.f && s == -1.f) return make_float2(-v.y, v.x);
    else return make_float2(fmaf(v.x, c, v.y * s), fmaf(v.y, c, -(v.x * s)));
}

template <int N>
__host__ __device__ __forceinline__ void dft_nat(float2 (&v)[N]);

// dense DFT of an odd prime length in its symmetric form: with s_j = x_j + x_{P-j}, d_j = x_j - x_{P-j} (j <= h = (P-1)/2)
//   X_k, X_{P-k} = (x_0 + sum_j s_j cos(2 pi j k / P)) -+ i (sum_j d_j sin(2 pi j k / P))
template <int P>
__host__ __device__ __forceinline__ void dft_prime(float2 (&v)[P]) {
    constexpr int H = (P - 1) / 2;
    float2 s[H + 1], d[H + 1];
    const float2 x0 = v[0];
    float2 tot = x0;
    static_for<1, H + 1>([&](auto j_) {
        constexpr int j = decltype(j_)::value;
        s[j] = add(v[j], v[P - j]);
        d[j] = sub(v[j], v[P - j]);
        tot = add(tot, s[j]);
    });
    v[0] = tot;
    static_for<1, H + 1>([&](auto k_) {
        constexpr int k = decltype(k_)::value;
        float2 a = x0, b = make_float2(0.f, 0.f);
        static_for<1, H + 1>([&](auto j_) {
            constexpr int j = decltype(j_)::value;
            constexpr float c = root_cos(j * k, P), sn = root_sin(j * k, P);
            a.x = fmaf(s[j].x, c, a.x);
            a.y = fmaf(s[j].y, c, a.y);
            b.x = fmaf(d[j].x, sn, b.x);
            b.y = fmaf(d[j].y, sn, b.y);
        });
        v[k] = make_float2(a.x + b.y, a.y - b.x);           // a - i b
        v[P - k] = make_float2(a.x - b.y, a.y + b.x);       // a + i b
    });
}

__host__ __device__ __forceinline__ void dft4_(float2& a0, float2& a1, float2& a2, float2& a3) {
    const float2 s02 = add(a0, a2), d02 = sub(a0, a2), s13 = add(a1, a3), t = sub(a1, a3);
    const float2 d13 = make_float2(t.y, -t.x);                // * (-i)
    a0 = add(s02, s13); a1 = add(d02, d13); a2 = sub(s02, s13); a3 = sub(d02, d13);
}

// composite lengths: N = A x B.  gcd(A, B) = 1: Good-Thomas (input n = (B a + A b) mod N, output k = (B B^-1 k1 + A A^-1 k2) mod N,
// no twiddles); otherwise Cooley-Tukey (n = a B + b, twiddle W_N^(b k1), k = k1 + A k2).  All indices are compile-time constants,
// so the "copies" between the arrays are register renamings.
template <int N>
__host__ __device__ __forceinline__ void dft_composite(float2 (&v)[N]) {
    constexpr int A = pick_factor(N), B = N / A;
    static_assert(A > 1 && B > 1, "unsupported length");
    constexpr bool pfa = gcd(A, B) == 1;
    float2 y[N];
    static_for<0, B>([&](auto b_) {
        constexpr int b = decltype(b_)::value;
        float2 col[A];
        static_for<0, A>([&](auto a_) {
            constexpr int a = decltype(a_)::value;
            col[a] = v[pfa ? (B * a + A * b) % N : a * B + b];
        });
        dft_nat<A>(col);
        static_for<0, A>([&](auto k_) {
            constexpr int k1 = decltype(k_)::value;
            if constexpr (pfa) y[k1 * B + b] = col[k1];
            else y[k1 * B + b] = mul_root<N, b * k1>(col[k1]);
        });
    });
    static_for<0, A>([&](auto k_) {
        constexpr int k1 = decltype(k_)::value;
        float2 row[B];
        static_for<0, B>([&](auto b_) { constexpr int b = decltype(b_)::value; row[b] = y[k1 * B + b]; });
        dft_nat<B>(row);
        static_for<0, B>([&](auto k2_) {
            constexpr int k2 = decltype(k2_)::value;
            constexpr int k = pfa ? (B * inv_mod(B, A) * k1 + A * inv_mod(A, B) * k2) % N : k1 + A * k2;
            v[k] = row[k2];
        });
    });
}

// forward DFT, natural order in and out
template <int N>
__host__ __device__ __forceinline__ void dft_nat(float2 (&v)[N]) {
    static_assert(supported(N) || N == 1, "length with a prime factor above 19");
    if constexpr (N == 1) {
    } else if constexpr (N == 2) {
        const float2 a = v[0], b = v[1];
        v[0] = add(a, b);
        v[1] = sub(a, b);
    } else if constexpr (N == 4) {
        dft4_(v[0], v[1], v[2], v[3]);
    } else if constexpr (is_prime(N)) {
        dft_prime<N>(v);
    } else {
        dft_composite<N>(v);      // (8 = 4 x 2 and 16 = 4 x 4 by Cooley-Tukey with their trivial / sqrt(1/2) twiddles as literals)
    }
}

}  // namespace sdft
