// tools/stream_probe.hip — what this box's memory system gives a kernel with k_fused2's traffic and NOTHING else to do.
//   hipcc --offload-arch=gfx950 -O3 -o mesh2splat_amd/_build/stream_probe tools/stream_probe.hip && mesh2splat_amd/_build/stream_probe
// Three kernels, HIP events over 40 launches each, inputs and outputs larger than the Infinity Cache where it matters:
//   copy     float4 in -> float4 out, 1 GiB each way (the guide's "float4 copy": 6.29 TB/s)
//   write    non-temporal 16 B / lane stores only, 1 GiB
//   mimic    config 3's shape: 3916 workgroups of 256 threads; each reads its 256 triangles from eleven planes (144 B / triangle,
//            the widths of TriPlanes) and writes 700 records of 96 B as the conversion kernels do — every wave stages nothing,
//            computes nothing, and stores 32 records (3 KB) with three non-temporal 16 B / lane instructions at a time —
//            into one contiguous range per workgroup.  407 MB per launch, the algorithmic bytes of config 3.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); std::exit(1); } } while (0)

__device__ __forceinline__ void nt_store4(float4* p, float4 v) {
    __builtin_nontemporal_store(v.x, &p->x); __builtin_nontemporal_store(v.y, &p->y);
    __builtin_nontemporal_store(v.z, &p->z); __builtin_nontemporal_store(v.w, &p->w);
}

__global__ void __launch_bounds__(256) k_copy(const float4* __restrict__ in, float4* __restrict__ out, size_t n) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) out[i] = in[i];
}
__global__ void __launch_bounds__(256) k_write(float4* __restrict__ out, size_t n) {
    const float4 v = make_float4(1.0f, 2.0f, 3.0f, (float)threadIdx.x);
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) nt_store4(&out[i], v);
}

struct Planes { const float4 *A0, *A1; const float* A2; const float4* B0; const float2* B1; const float4 *C0, *C1; const float* C2; const float4 *D0, *D1, *D2; };

// rec_per_wg records per workgroup (all four waves take turns of 32 records)
__global__ void __launch_bounds__(256, 3) k_mimic(Planes p, uint32_t n_tri, float4* __restrict__ out, uint32_t rec_per_wg) {
    const uint32_t t = blockIdx.x * 256 + threadIdx.x;
    float acc = 0.0f;
    if (t < n_tri) {
        const float4 a0 = p.A0[t], a1 = p.A1[t], b0 = p.B0[t], c0 = p.C0[t], c1 = p.C1[t], d0 = p.D0[t], d1 = p.D1[t], d2 = p.D2[t];
        const float2 b1 = p.B1[t];
        acc = a0.x + a1.y + p.A2[t] + b0.z + b1.x + c0.w + c1.x + p.C2[t] + d0.y + d1.z + d2.w;
    }
    const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float4* base = out + (size_t)blockIdx.x * rec_per_wg * 6;
    const float4 v = make_float4(acc, 1.0f, 2.0f, 3.0f);
    for (uint32_t r0 = wave * 32; r0 < rec_per_wg; r0 += 4 * 32) {
        const uint32_t nrec = min(32u, rec_per_wg - r0);
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            const uint32_t q = lane + 64u * j;
            if (q / 6u < nrec) nt_store4(&base[(size_t)r0 * 6 + q], v);
        }
    }
}

static float time_ms(hipStream_t st, int reps, void (*launch)(void*), void* ctx) {
    hipEvent_t a, b;
    CHECK(hipEventCreate(&a)); CHECK(hipEventCreate(&b));
    for (int i = 0; i < 3; ++i) launch(ctx);
    CHECK(hipStreamSynchronize(st));
    CHECK(hipEventRecord(a, st));
    for (int i = 0; i < reps; ++i) launch(ctx);
    CHECK(hipEventRecord(b, st));
    CHECK(hipEventSynchronize(b));
    float ms = 0; CHECK(hipEventElapsedTime(&ms, a, b));
    return ms / reps;
}

struct Ctx { hipStream_t st; float4 *in, *out; size_t n; Planes p; uint32_t n_tri, rec; float4* rout; };
static void l_copy(void* c_) { Ctx* c = (Ctx*)c_; hipLaunchKernelGGL(k_copy, dim3(256 * 32), dim3(256), 0, c->st, c->in, c->out, c->n); }
static void l_write(void* c_) { Ctx* c = (Ctx*)c_; hipLaunchKernelGGL(k_write, dim3(256 * 32), dim3(256), 0, c->st, c->out, c->n); }
static void l_mimic(void* c_) { Ctx* c = (Ctx*)c_; hipLaunchKernelGGL(k_mimic, dim3((c->n_tri + 255) / 256), dim3(256), 0, c->st, c->p, c->n_tri, c->rout, c->rec); }

int main() {
    Ctx c{};
    CHECK(hipStreamCreate(&c.st));
    const size_t bytes = (size_t)1 << 30;
    c.n = bytes / 16;
    CHECK(hipMalloc(&c.in, bytes)); CHECK(hipMalloc(&c.out, bytes));
    CHECK(hipMemset(c.in, 1, bytes)); CHECK(hipMemset(c.out, 0, bytes));
    const float copy_ms = time_ms(c.st, 20, l_copy, &c);
    const float write_ms = time_ms(c.st, 20, l_write, &c);
    // config 3: 1 002 252 triangles, 2 738 368 records -> 699.3 per workgroup of 256 triangles
    c.n_tri = 1002252; c.rec = 700;
    const size_t widths[11] = { 16, 16, 4, 16, 8, 16, 16, 4, 16, 16, 16 };
    char* planes = nullptr; size_t off[11], cur = 0;
    for (int k = 0; k < 11; ++k) { off[k] = cur; cur = (cur + c.n_tri * widths[k] + 255) / 256 * 256; }
    CHECK(hipMalloc(&planes, cur)); CHECK(hipMemset(planes, 0, cur));
    c.p = Planes{ (const float4*)(planes + off[0]), (const float4*)(planes + off[1]), (const float*)(planes + off[2]), (const float4*)(planes + off[3]),
                  (const float2*)(planes + off[4]), (const float4*)(planes + off[5]), (const float4*)(planes + off[6]), (const float*)(planes + off[7]),
                  (const float4*)(planes + off[8]), (const float4*)(planes + off[9]), (const float4*)(planes + off[10]) };
    const size_t n_wg = (c.n_tri + 255) / 256;
    CHECK(hipMalloc(&c.rout, n_wg * c.rec * 96));
    const float mimic_ms = time_ms(c.st, 40, l_mimic, &c);
    const double mimic_bytes = 144.0 * c.n_tri + 96.0 * (double)n_wg * c.rec;
    std::printf("{\"copy_GBps\": %.1f, \"write_only_GBps\": %.1f, \"mimic_ms\": %.4f, \"mimic_bytes\": %.0f, \"mimic_GBps\": %.1f, \"mimic_frac_of_8TBps\": %.3f}\n",
                2.0 * bytes / (copy_ms * 1e-3) / 1e9, bytes / (write_ms * 1e-3) / 1e9, mimic_ms, mimic_bytes, mimic_bytes / (mimic_ms * 1e-3) / 1e9,
                mimic_bytes / (mimic_ms * 1e-3) / 8e12);
    return 0;
}
