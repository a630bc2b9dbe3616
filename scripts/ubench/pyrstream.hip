// What can a makeImages-shaped stream reach on the device? (round 4, VERDICT r3 #6: pyr_one_pass at 0.37 of HBM peak.) Per level-0 pixel the pass reads 4 B and
// writes a 16-B texel + a 4-B absSquaredGrad value; the upper levels add a third. This times, at 1920x1072:
//   fill16      one 16-B nontemporal store per lane (the write side alone, 33 MB)
//   px1         one pixel per lane: 4-B load, 16-B + 4-B stores (41 MB)
//   px4row      four consecutive pixels per lane: one 16-B load, four 16-B stores, one 16-B store
//   tile        a 64x16 tile per 256-lane workgroup, rows of 64 lanes (the store pattern of pyr_one_pass: 1 KB + 256 B contiguous per wave instruction)
// build: hipcc --offload-arch=gfx950 -O3 -o scripts/ubench/pyrstream scripts/ubench/pyrstream.hip
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f4 __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(256) void fill16(f4* __restrict__ d, size_t n) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n) { f4 v; v.x = (float)i; v.y = 1.f; v.z = 2.f; v.w = 0.f; __builtin_nontemporal_store(v, d + i); }
}
template <int NT>
__global__ __launch_bounds__(256) void px1(const float* __restrict__ I, f4* __restrict__ d, float* __restrict__ g, size_t n) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const float c = I[i];
    f4 v; v.x = c; v.y = c * 0.5f; v.z = c * 0.25f; v.w = 0.f;
    if (NT) { __builtin_nontemporal_store(v, d + i); __builtin_nontemporal_store(c * c, g + i); } else { d[i] = v; g[i] = c * c; }
}
__global__ __launch_bounds__(256) void px4row(const f4* __restrict__ I, f4* __restrict__ d, f4* __restrict__ g, size_t n4) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n4) return;
    const f4 c = I[i];
    f4 gg;
#pragma unroll
    for (int k = 0; k < 4; ++k) { f4 v; v.x = c[k]; v.y = c[k] * 0.5f; v.z = c[k] * 0.25f; v.w = 0.f; __builtin_nontemporal_store(v, d + 4 * i + k); gg[k] = c[k] * c[k]; }
    __builtin_nontemporal_store(gg, g + i);
}
template <int TH>
__global__ __launch_bounds__(256) void tile(const float* __restrict__ I, f4* __restrict__ d, float* __restrict__ g, int w, int h) {
    const int tiles_x = w / 64, x0 = (blockIdx.x % tiles_x) * 64, y0 = (blockIdx.x / tiles_x) * TH;
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    float c[TH / 4];
#pragma unroll
    for (int k = 0; k < TH / 4; ++k) c[k] = I[(size_t)(y0 + ty + 4 * k) * w + x0 + tx];
#pragma unroll
    for (int k = 0; k < TH / 4; ++k) {
        const size_t i = (size_t)(y0 + ty + 4 * k) * w + x0 + tx;
        f4 v; v.x = c[k]; v.y = c[k] * 0.5f; v.z = c[k] * 0.25f; v.w = 0.f;
        __builtin_nontemporal_store(v, d + i); __builtin_nontemporal_store(c[k] * c[k], g + i);
    }
}
template <class F> double timeit(F f, int iters) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    f(); hipDeviceSynchronize();
    hipEventRecord(e0); for (int i = 0; i < iters; ++i) f(); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); return ms * 1e3 / iters;
}
int main() {
    const int w = 1920, h = 1072; const size_t n = (size_t)w * h;
    float *I, *g; f4* d; hipMalloc(&I, n * 4 * 4); hipMalloc(&g, n * 4 * 4); hipMalloc(&d, n * 16 * 4); hipMemset(I, 0, n * 4);
    const int it = 200;
    double t;
    t = timeit([&] { fill16<<<(n + 255) / 256, 256>>>(d, n); }, it); printf("fill16  %.2f us  %.2f TB/s (write only, %.1f MB)\n", t, n * 16 / t / 1e6, n * 16 / 1e6);
    t = timeit([&] { px1<1><<<(n + 255) / 256, 256>>>(I, d, g, n); }, it); printf("px1 nt  %.2f us  %.2f TB/s (%.1f MB)\n", t, n * 24 / t / 1e6, n * 24 / 1e6);
    t = timeit([&] { px1<0><<<(n + 255) / 256, 256>>>(I, d, g, n); }, it); printf("px1     %.2f us  %.2f TB/s\n", t, n * 24 / t / 1e6);
    t = timeit([&] { px4row<<<(n / 4 + 255) / 256, 256>>>((const f4*)I, d, (f4*)g, n / 4); }, it); printf("px4row  %.2f us  %.2f TB/s\n", t, n * 24 / t / 1e6);
    t = timeit([&] { tile<16><<<(w / 64) * (h / 16), 256>>>(I, d, g, w, h); }, it); printf("tile16  %.2f us  %.2f TB/s\n", t, n * 24 / t / 1e6);
    t = timeit([&] { tile<8><<<(w / 64) * (h / 8), 256>>>(I, d, g, w, h); }, it); printf("tile8   %.2f us  %.2f TB/s\n", t, n * 24 / t / 1e6);
    t = timeit([&] { tile<32><<<(w / 64) * (h / 32), 256>>>(I, d, g, w, h); }, it); printf("tile32  %.2f us  %.2f TB/s\n", t, n * 24 / t / 1e6);
    // the whole pyramid's bytes as four px1 launches back to back (levels 0-3): the launch-per-level floor
    t = timeit([&] { size_t m = n; for (int l = 0; l < 4; ++l) { px1<1><<<(m + 255) / 256, 256>>>(I, d, g, m); m /= 4; } }, it); printf("px1 x 4 levels  %.2f us\n", t);
    return 0;
}
