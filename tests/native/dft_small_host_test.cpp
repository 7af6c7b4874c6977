// Host-side check of csrc/mvs_dft_small.h (the functions are __host__ __device__): every supported length against a direct
// double-precision DFT.  Built and run by tests/test_dft_small_host.py with hipcc (no GPU needed).
#include <cmath>
#include <cstdio>
#include <cstdlib>

#include "mvs_dft_small.h"

template <int N>
static double check() {
    float2 v[N];
    double xr[N], xi[N];
    unsigned s = 12345u + N;
    for (int j = 0; j < N; ++j) {
        s = s * 1664525u + 1013904223u; xr[j] = (double)(s >> 8) / (1 << 24) - 0.5;
        s = s * 1664525u + 1013904223u; xi[j] = (double)(s >> 8) / (1 << 24) - 0.5;
        v[j] = make_float2((float)xr[j], (float)xi[j]);
        xr[j] = v[j].x; xi[j] = v[j].y;
    }
    sdft::dft_nat<N>(v);
    double worst = 0.0, scale = 0.0;
    for (int k = 0; k < N; ++k) {
        double ar = 0, ai = 0;
        for (int j = 0; j < N; ++j) {
            const double a = -2.0 * M_PI * (double)((j * k) % N) / N;
            ar += xr[j] * cos(a) - xi[j] * sin(a);
            ai += xr[j] * sin(a) + xi[j] * cos(a);
        }
        worst = fmax(worst, hypot(v[k].x - ar, v[k].y - ai));
        scale = fmax(scale, hypot(ar, ai));
    }
    printf("%d %.3e\n", N, worst / scale);
    return worst / scale;
}

template <int N>
static void run_all(double& worst) {
    if constexpr (N >= 2) {
        if constexpr (sdft::supported(N)) worst = fmax(worst, check<N>());
        run_all<N - 1>(worst);
    }
}

int main() {
    double worst = 0.0;
    run_all<64>(worst);
    printf("worst %.3e\n", worst);
    return worst < 2e-6 ? 0 : 1;
}
