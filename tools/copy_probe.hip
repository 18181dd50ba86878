// What can a plain kernel move on this box?  float4 copy / write-only / read-only streams over 1 GiB, a few launch shapes, plain and
// non-temporal accesses — the context for "at the copy rate" statements (the guide quotes 6.29 TB/s for a float4 copy; torch's
// copy_ of a byte tensor, which bench.py reports as measured_copy_peak, reaches 4.7-5.2 on the same boxes).
//   hipcc --offload-arch=gfx950 -O3 tools/copy_probe.hip -o /tmp/copy_probe && /tmp/copy_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float v4f __attribute__((ext_vector_type(4)));
__device__ __forceinline__ float4 nt_ld(const float4* p) { v4f v = __builtin_nontemporal_load(reinterpret_cast<const v4f*>(p)); return make_float4(v.x, v.y, v.z, v.w); }
__device__ __forceinline__ void nt_st(float4* p, float4 f) { v4f v = { f.x, f.y, f.z, f.w }; __builtin_nontemporal_store(v, reinterpret_cast<v4f*>(p)); }
template <int NT> __global__ void __launch_bounds__(256) k_copy(const float4* __restrict__ a, float4* __restrict__ b, size_t n) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        float4 v = NT ? nt_ld(&a[i]) : a[i];
        if (NT) nt_st(&b[i], v); else b[i] = v;
    }
}
template <int NT> __global__ void __launch_bounds__(256) k_fill(float4* __restrict__ b, size_t n) {
    const float4 v = make_float4(1, 2, 3, 4);
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) { if (NT) nt_st(&b[i], v); else b[i] = v; }
}
__global__ void __launch_bounds__(256) k_read(const float4* __restrict__ a, size_t n, float* sink) {
    float s = 0;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) { float4 v = a[i]; s += v.x + v.y + v.z + v.w; }
    if (s == 12345.678f) *sink = s;
}
template <class F> static double run(F f, double bytes) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int i = 0; i < 3; ++i) f();
    hipEventRecord(e0); for (int i = 0; i < 20; ++i) f(); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms = 0; hipEventElapsedTime(&ms, e0, e1);
    return bytes * 20 / (ms * 1e-3) / 1e9;
}
int main() {
    const size_t bytes = 1ull << 30, n = bytes / 16;
    float4 *a, *b; float* sink;
    hipMalloc(&a, bytes); hipMalloc(&b, bytes); hipMalloc(&sink, 4);
    hipMemset(a, 1, bytes); hipMemset(b, 0, bytes);
    for (int grid : {2048, 8192, 32768, (int)(n / 256)}) {
        printf("{\"grid\": %d, \"copy_GBps\": %.0f, \"copy_nt_GBps\": %.0f, \"fill_GBps\": %.0f, \"fill_nt_GBps\": %.0f, \"read_GBps\": %.0f}\n", grid,
               run([&] { hipLaunchKernelGGL(k_copy<0>, dim3(grid), dim3(256), 0, 0, a, b, n); }, 2.0 * bytes),
               run([&] { hipLaunchKernelGGL(k_copy<1>, dim3(grid), dim3(256), 0, 0, a, b, n); }, 2.0 * bytes),
               run([&] { hipLaunchKernelGGL(k_fill<0>, dim3(grid), dim3(256), 0, 0, b, n); }, 1.0 * bytes),
               run([&] { hipLaunchKernelGGL(k_fill<1>, dim3(grid), dim3(256), 0, 0, b, n); }, 1.0 * bytes),
               run([&] { hipLaunchKernelGGL(k_read, dim3(grid), dim3(256), 0, 0, a, n, sink); }, 1.0 * bytes));
    }
    printf("{\"hipMemcpyDtoD_GBps\": %.0f}\n", run([&] { hipMemcpyAsync(b, a, bytes, hipMemcpyDeviceToDevice, 0); }, 2.0 * bytes));
    return 0;
}
