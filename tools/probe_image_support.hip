// Probe (not product code): does the target have image-sampling instructions?  For gfx950 the compiler answers
// "The image/texture API not supported on the device" -- MI355X has no texture units, hence the software sampler.
//   hipcc --offload-arch=gfx950 --offload-device-only -S tools/probe_image_support.hip -o /dev/null
#include <hip/hip_runtime.h>
__global__ void k(hipTextureObject_t t, float4* out, float lod) {
    float u = threadIdx.x * 0.01f, v = blockIdx.x * 0.02f;
    out[blockIdx.x * blockDim.x + threadIdx.x] = tex2DLod<float4>(t, u, v, lod);
}
