// Which access patterns reach the box's streaming rate?  1 GiB float4 fill / copy as
//   oneshot      one thread per element, workgroups in address order (many short workgroups)
//   permuted     the same, workgroup index bit-reversed (concurrent workgroups touch scattered addresses)
//   stride       G resident workgroups, grid-stride loop (concurrent workgroups touch one moving window)
//   blocked      G resident workgroups, each walks ITS OWN contiguous 1/G of the buffer (G sequential streams far apart — what a kernel
//                whose workgroups own consecutive pieces of the output does while they are resident together)
//   hipcc --offload-arch=gfx950 -O3 tools/stream_patterns.hip -o /tmp/sp && /tmp/sp
#include <hip/hip_runtime.h>
#include <cstdio>
enum { ONESHOT, PERMUTED, STRIDE, BLOCKED };
template <int MODE, bool COPY> __global__ void __launch_bounds__(256) k(const float4* __restrict__ a, float4* __restrict__ b, size_t n, int logwg, size_t chunk) {
    const float4 c = make_float4(1, 2, 3, 4);
    if (MODE == ONESHOT || MODE == PERMUTED) {
        size_t wg = blockIdx.x;
        if (MODE == PERMUTED) wg = __brev((unsigned)wg) >> (32 - logwg);
        const size_t i = wg * 256 + threadIdx.x;
        if (i < n) b[i] = COPY ? a[i] : c;
    } else if (MODE == STRIDE) {
        for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) b[i] = COPY ? a[i] : c;
    } else {
        const size_t per = n / gridDim.x, i0 = blockIdx.x * per;          // chunk = elements written before moving on (256: one row of the workgroup)
        for (size_t i = threadIdx.x; i < per; i += 256) b[i0 + i] = COPY ? a[i0 + i] : c;
    }
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
    float4 *a, *b;
    hipMalloc(&a, bytes); hipMalloc(&b, bytes);
    hipMemset(a, 1, bytes); hipMemset(b, 0, bytes);
    const int nwg = (int)(n / 256), logwg = 18;   // 262144 workgroups
#define R(MODE, COPY, G) run([&] { hipLaunchKernelGGL((k<MODE, COPY>), dim3(G), dim3(256), 0, 0, a, b, n, logwg, (size_t)256); }, (COPY ? 2.0 : 1.0) * bytes)
    printf("{\"pattern\": \"oneshot\", \"fill_GBps\": %.0f, \"copy_GBps\": %.0f}\n", R(ONESHOT, false, nwg), R(ONESHOT, true, nwg));
    printf("{\"pattern\": \"permuted\", \"fill_GBps\": %.0f, \"copy_GBps\": %.0f}\n", R(PERMUTED, false, nwg), R(PERMUTED, true, nwg));
    for (int G : {1024, 2048, 4096, 8192, 16384}) {
        printf("{\"pattern\": \"stride\", \"G\": %d, \"fill_GBps\": %.0f, \"copy_GBps\": %.0f}\n", G, R(STRIDE, false, G), R(STRIDE, true, G));
        printf("{\"pattern\": \"blocked\", \"G\": %d, \"fill_GBps\": %.0f, \"copy_GBps\": %.0f}\n", G, R(BLOCKED, false, G), R(BLOCKED, true, G));
    }
    return 0;
}
