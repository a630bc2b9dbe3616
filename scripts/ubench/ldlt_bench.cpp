// Host LDL^T of solveSystemF's (8W+4)^2 system: the round-2 left-looking version (two dependent accumulator chains) against the blocked one (eight independent
// chains, threshold pivoting). Build and run on the GPU box's host: g++ -O3 -std=c++17 scripts/ubench/ldlt_bench.cpp -o /tmp/ldlt_bench && /tmp/ldlt_bench
#include <cstdio>
#include <chrono>
#include <random>
#define __host__
#define __device__
#include "../../nalo-slam_amd/csrc/host_math.h"
using namespace nalo;
int main(int argc, char** argv) {
    const double slack = argc > 1 ? atof(argv[1]) : 64.0;
    for (int n : {68, 100}) {
        const int lda = (n + 7) / 8 * 8;
        std::mt19937 g(1); std::normal_distribution<double> N(0, 1);
        std::vector<double> M(n * n), H(n * n, 0.0), b(n), x0(n), x1(n);
        for (auto& v : M) v = N(g);
        for (int i = 0; i < n; ++i) for (int j = 0; j < n; ++j) { double s = 0; for (int k = 0; k < n; ++k) s += M[i * n + k] * M[j * n + k]; H[i * n + j] = s; }
        std::vector<double> sv(n);
        for (int i = 0; i < n; ++i) sv[i] = 1.0 / std::sqrt(H[i * n + i] + 10);                  // Jacobi scaling, as solveSystemF
        for (int i = 0; i < n; ++i) for (int j = 0; j < n; ++j) H[i * n + j] *= sv[i] * sv[j];
        for (auto& v : b) v = N(g);
        std::vector<double> A0(n * n), A1(lda * lda), w0(2 * n), w1(3 * lda); std::vector<int> p0(n), p1(n);
        double t0 = 0, t1 = 0; const int reps = 5000;
        for (int r = 0; r < reps; ++r) {
            A0 = H; for (int i = 0; i < lda * lda; ++i) A1[i] = 0; for (int i = 0; i < n; ++i) for (int j = 0; j < n; ++j) A1[i * lda + j] = H[i * n + j];
            auto a = std::chrono::steady_clock::now();
            ldlt_solve_inplace(n, A0.data(), b.data(), x0.data(), w0.data(), p0.data());
            auto c = std::chrono::steady_clock::now();
            ldlt_solve_blocked(n, lda, A1.data(), b.data(), x1.data(), w1.data(), p1.data(), slack);
            auto d = std::chrono::steady_clock::now();
            t0 += std::chrono::duration<double, std::micro>(c - a).count(); t1 += std::chrono::duration<double, std::micro>(d - c).count();
        }
        double e = 0, s = 0; for (int i = 0; i < n; ++i) { e = std::max(e, std::fabs(x0[i] - x1[i])); s = std::max(s, std::fabs(x0[i])); }
        int swaps = 0; for (int i = 0; i < n; ++i) swaps += p1[i] != i;
        printf("n=%d slack=%g: round-2 LDLT %.2f us, blocked %.2f us, |x_old - x_new| / |x| = %.2e, permuted entries %d\n", n, slack, t0 / reps, t1 / reps, e / s, swaps);
    }
}
