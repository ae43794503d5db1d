// Store-bandwidth probe for the level-3 tensor layout ([12][64][128][416] fp32, 164 MB): what does a kernel that writes all 12
// channel planes of its voxels get, compared with a contiguous fill?
//   hipcc -O3 --offload-arch=gfx950 tools/probe_store.hip -o tools/probe_store && tools/probe_store
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

constexpr int C = 12;
constexpr long DHW = 64L * 128 * 416;

__global__ void k_fill(float4* out, long n4) {
  for (long i = blockIdx.x * 256L + threadIdx.x; i < n4; i += (long)gridDim.x * 256) out[i] = make_float4(1.f, 2.f, 3.f, 4.f);
}
// thread per quad, 12 channel planes, 16-byte stores
__global__ void k_quad12(float* out, long nq) {
  const long q = blockIdx.x * 256L + threadIdx.x;
  if (q >= nq) return;
#pragma unroll
  for (int c = 0; c < C; ++c) *reinterpret_cast<float4*>(out + c * DHW + 4 * q) = make_float4((float)c, 2.f, 3.f, (float)q);
}
// thread per voxel, 12 channel planes, 4-byte stores
__global__ void k_vox12(float* out, long n) {
  const long v = blockIdx.x * 256L + threadIdx.x;
  if (v >= n) return;
#pragma unroll
  for (int c = 0; c < C; ++c) out[c * DHW + v] = (float)c + (float)v;
}
// thread per quad, grid-stride persistent (2048 workgroups), 12 planes
__global__ void k_quad12_persist(float* out, long nq) {
  for (long q = blockIdx.x * 256L + threadIdx.x; q < nq; q += (long)gridDim.x * 256) {
#pragma unroll
    for (int c = 0; c < C; ++c) *reinterpret_cast<float4*>(out + c * DHW + 4 * q) = make_float4((float)c, 2.f, 3.f, (float)q);
  }
}
// channel-major: blockIdx.y = channel, each workgroup streams one plane region
__global__ void k_plane_major(float* out, long nq) {
  const int c = blockIdx.y;
  for (long q = blockIdx.x * 256L + threadIdx.x; q < nq; q += (long)gridDim.x * 256)
    *reinterpret_cast<float4*>(out + c * DHW + 4 * q) = make_float4((float)c, 2.f, 3.f, (float)q);
}

template <class F>
static float time_us(F f, int n = 20) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  for (int i = 0; i < 3; ++i) f();
  hipDeviceSynchronize();
  hipEventRecord(e0);
  for (int i = 0; i < n; ++i) f();
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms = 0;
  hipEventElapsedTime(&ms, e0, e1);
  return ms * 1e3f / n;
}

int main() {
  float* out;
  const long n = C * DHW;
  hipMalloc(&out, n * sizeof(float));
  const double mb = n * 4 / 1e6;
  const long nq = DHW / 4;
  float t;
  t = time_us([&] { hipLaunchKernelGGL(k_fill, dim3(4096), dim3(256), 0, 0, (float4*)out, n / 4); });
  printf("fill (contiguous, float4, 4096 WGs grid-stride) : %7.1f us  %5.2f TB/s\n", t, mb / t);
  t = time_us([&] { hipLaunchKernelGGL(k_quad12, dim3((unsigned)((nq + 255) / 256)), dim3(256), 0, 0, out, nq); });
  printf("quad x 12 planes (float4 stores)                : %7.1f us  %5.2f TB/s\n", t, mb / t);
  t = time_us([&] { hipLaunchKernelGGL(k_vox12, dim3((unsigned)((DHW + 255) / 256)), dim3(256), 0, 0, out, DHW); });
  printf("voxel x 12 planes (dword stores)                : %7.1f us  %5.2f TB/s\n", t, mb / t);
  t = time_us([&] { hipLaunchKernelGGL(k_quad12_persist, dim3(2048), dim3(256), 0, 0, out, nq); });
  printf("quad x 12 planes, 2048 persistent WGs           : %7.1f us  %5.2f TB/s\n", t, mb / t);
  t = time_us([&] { hipLaunchKernelGGL(k_plane_major, dim3(256, C), dim3(256), 0, 0, out, nq); });
  printf("plane-major (blockIdx.y = channel)              : %7.1f us  %5.2f TB/s\n", t, mb / t);
  // the same patterns over EIGHT buffers in rotation (1.3 GB: nothing written is still in the 256 MB infinity cache when it is
  // written again) — what a layer's output write looks like inside a forward pass
  float* bufs[8];
  for (int i = 0; i < 8; ++i) hipMalloc(&bufs[i], n * sizeof(float));
  int r = 0;
  t = time_us([&] { hipLaunchKernelGGL(k_fill, dim3(4096), dim3(256), 0, 0, (float4*)bufs[r++ & 7], n / 4); }, 24);
  printf("cold: fill                                      : %7.1f us  %5.2f TB/s\n", t, mb / t);
  t = time_us([&] { hipLaunchKernelGGL(k_quad12, dim3((unsigned)((nq + 255) / 256)), dim3(256), 0, 0, bufs[r++ & 7], nq); }, 24);
  printf("cold: quad x 12 planes (float4 stores)          : %7.1f us  %5.2f TB/s\n", t, mb / t);
  t = time_us([&] { hipLaunchKernelGGL(k_vox12, dim3((unsigned)((DHW + 255) / 256)), dim3(256), 0, 0, bufs[r++ & 7], DHW); }, 24);
  printf("cold: voxel x 12 planes (dword stores)          : %7.1f us  %5.2f TB/s\n", t, mb / t);
  t = time_us([&] { hipLaunchKernelGGL(k_plane_major, dim3(256, C), dim3(256), 0, 0, bufs[r++ & 7], nq); }, 24);
  printf("cold: plane-major                               : %7.1f us  %5.2f TB/s\n", t, mb / t);
  for (int i = 0; i < 8; ++i) hipFree(bufs[i]);
  hipFree(out);
  return 0;
}
