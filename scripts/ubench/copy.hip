// Streaming copy / triad bandwidth of the device: which shape of a plain float4 kernel reaches the guide's 6.3 TB/s (MI355X_MICROARCH.md: "6.29 TB/s measured, float4 copy")?
// Picks the shape nalo_hbm_calibrate uses. build: hipcc --offload-arch=gfx950 -O3 -o scripts/ubench/copy scripts/ubench/copy.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <algorithm>
typedef float f4 __attribute__((ext_vector_type(4)));
template <int U, int NT>
__global__ __launch_bounds__(256) void copy_k(const float4* __restrict__ a_, float4* __restrict__ d_, size_t n) {
    const f4* __restrict__ a = reinterpret_cast<const f4*>(a_); f4* __restrict__ d = reinterpret_cast<f4*>(d_);
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    for (; i + (U - 1) * stride < n; i += U * stride) {
        f4 v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) v[u] = NT ? __builtin_nontemporal_load(a + i + u * stride) : a[i + u * stride];
#pragma unroll
        for (int u = 0; u < U; ++u) { if (NT) __builtin_nontemporal_store(v[u], d + i + u * stride); else d[i + u * stride] = v[u]; }
    }
    for (; i < n; i += stride) d[i] = a[i];
}
template <int U, int NT> double run(const float4* a, float4* d, size_t n, int wg_per_cu, int iters) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int grid = (int)std::min<size_t>((n + 255) / 256, (size_t)256 * wg_per_cu);
    copy_k<U, NT><<<grid, 256>>>(a, d, n);
    hipEventRecord(e0);
    for (int i = 0; i < iters; ++i) copy_k<U, NT><<<grid, 256>>>(a, d, n);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    return 2.0 * n * 16 * iters / (ms * 1e-3) / 1e12;
}
int main() {
    const size_t bytes = (size_t)1 << 30, n = bytes / 16;
    float4 *a, *d; hipMalloc(&a, bytes); hipMalloc(&d, bytes); hipMemset(a, 0, bytes); hipMemset(d, 0, bytes);
    for (int wpc : {4, 8, 16, 32, 1 << 20}) {
        printf("wg/CU %7d:  U1 %.2f  U2 %.2f  U4 %.2f  U8 %.2f | nt: U1 %.2f  U4 %.2f  U8 %.2f TB/s\n", wpc, run<1, 0>(a, d, n, wpc, 10), run<2, 0>(a, d, n, wpc, 10), run<4, 0>(a, d, n, wpc, 10),
               run<8, 0>(a, d, n, wpc, 10), run<1, 1>(a, d, n, wpc, 10), run<4, 1>(a, d, n, wpc, 10), run<8, 1>(a, d, n, wpc, 10));
    }
    double t;
    { hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1); hipMemcpyAsync(d, a, bytes, hipMemcpyDeviceToDevice, 0); hipEventRecord(e0);
      for (int i = 0; i < 10; ++i) hipMemcpyAsync(d, a, bytes, hipMemcpyDeviceToDevice, 0);
      hipEventRecord(e1); hipEventSynchronize(e1); float ms; hipEventElapsedTime(&ms, e0, e1); t = 2.0 * bytes * 10 / (ms * 1e-3) / 1e12; }
    printf("hipMemcpyAsync D2D: %.2f TB/s\n", t);
    return 0;
}
