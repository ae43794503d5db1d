// Hardware probe (not product code): cycles per bf16 MFMA (16x16x32 vs the legacy 16x16x16) and how many plain VALU / LDS-read
// instructions fit beside them — from the same wave and from a partner wave on the same SIMD.
//   hipcc -O3 --offload-arch=gfx950 tools/probe_mfma_bf16.hip -o tools/probe_mfma_bf16
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1;} } while (0)

// MODE 0: x32 only; 1: x16 only; 2: x32 + NV VALU per MFMA (same wave); 3: waves 0-3 MFMA x32, waves 4-7 VALU only (NV per "slot");
// 4: x32 + NV ds_read_b64 per MFMA (same wave); 5: waves 4-7 issue ds_read_b64 only
template <int MODE, int NV>
__global__ __launch_bounds__(512) void k(float* out, int iters, float av, long long* cyc) {
  __shared__ float lds[8192];
  for (int i = threadIdx.x; i < 8192; i += blockDim.x) lds[i] = i * 0.001f;
  __syncthreads();
  f32x4 acc[8];
  for (int i = 0; i < 8; ++i) acc[i] = (f32x4){0, 0, 0, 0};
  bf16x8 a, b;
  for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(av + i * 0.01f + threadIdx.x * 0.001f); b[i] = (__bf16)(av * 0.5f - i * 0.02f); }
  const s16x4 a4 = __builtin_bit_cast(s16x4, __builtin_shufflevector(a, a, 0, 1, 2, 3)), b4 = __builtin_bit_cast(s16x4, __builtin_shufflevector(b, b, 0, 1, 2, 3));
  float v[8];
  for (int i = 0; i < 8; ++i) v[i] = av + i;
  const bool second = threadIdx.x >= 256;
  const float2* lp = reinterpret_cast<const float2*>(lds) + (threadIdx.x & 63);
  float2 lacc = {0, 0};
  long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const bool do_mfma = (MODE == 3 || MODE == 5) ? !second : true;
      const bool do_valu = (MODE == 2) || (MODE == 3 && second);
      const bool do_lds = (MODE == 4) || (MODE == 5 && second);
      if (do_mfma) {
        if (MODE == 1) acc[i] = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(a4, b4, acc[i], 0, 0, 0);
        else acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc[i], 0, 0, 0);
      }
      if (do_valu) {
#pragma unroll
        for (int j = 0; j < NV; ++j) v[(i + j) & 7] = __builtin_fmaf(v[(i + j) & 7], 1.0001f, 0.5f);
      }
      if (do_lds) {
#pragma unroll
        for (int j = 0; j < NV; ++j) { const float2 t = lp[((it + i * NV + j) & 15) * 64]; lacc.x += t.x; lacc.y += t.y; }
      }
    }
  }
  long long t1 = __builtin_amdgcn_s_memtime();
  float s = lacc.x + lacc.y;
  for (int i = 0; i < 8; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3] + v[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
  if (threadIdx.x == 256 && blockIdx.x == 0) cyc[1] = t1 - t0;
}

template <int MODE, int NV>
static int run(const char* what, float* d, long long* c, int threads) {
  const int iters = 4000;
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int rep = 0; rep < 2; ++rep) {
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL((k<MODE, NV>), dim3(256), dim3(threads), 0, 0, d, iters, 1.f, c);
    CK(hipEventRecord(e1)); CK(hipDeviceSynchronize());
  }
  float ms; CK(hipEventElapsedTime(&ms, e0, e1));
  long long h[2]; CK(hipMemcpy(h, c, 16, hipMemcpyDeviceToHost));
  printf("%-58s threads %3d: %.3f ms; wave0 %.2f cycles per MFMA slot, wave4 %.2f\n", what, threads, ms, (double)h[0] / iters / 8, (double)h[1] / iters / 8);
  return 0;
}

int main() {
  float* d; long long* c;
  CK(hipMalloc(&d, 256 * 512 * 4)); CK(hipMalloc(&c, 16)); CK(hipMemset(c, 0, 16));
  run<0, 0>("16x16x32 bf16, one wave per SIMD", d, c, 256);
  run<0, 0>("16x16x32 bf16, two waves per SIMD", d, c, 512);
  run<1, 0>("16x16x16 bf16 (legacy), one wave per SIMD", d, c, 256);
  run<1, 0>("16x16x16 bf16 (legacy), two waves per SIMD", d, c, 512);
  run<2, 1>("x32 + 1 v_fma per MFMA, same wave", d, c, 256);
  run<2, 2>("x32 + 2 v_fma per MFMA, same wave", d, c, 256);
  run<2, 3>("x32 + 3 v_fma per MFMA, same wave", d, c, 256);
  run<2, 4>("x32 + 4 v_fma per MFMA, same wave", d, c, 256);
  run<2, 2>("x32 + 2 v_fma per MFMA, two such waves per SIMD", d, c, 512);
  run<2, 4>("x32 + 4 v_fma per MFMA, two such waves per SIMD", d, c, 512);
  run<3, 1>("waves 0-3 x32 only | waves 4-7 1 v_fma per slot", d, c, 512);
  run<3, 2>("waves 0-3 x32 only | waves 4-7 2 v_fma per slot", d, c, 512);
  run<3, 3>("waves 0-3 x32 only | waves 4-7 3 v_fma per slot", d, c, 512);
  run<3, 4>("waves 0-3 x32 only | waves 4-7 4 v_fma per slot", d, c, 512);
  run<3, 6>("waves 0-3 x32 only | waves 4-7 6 v_fma per slot", d, c, 512);
  run<4, 1>("x32 + 1 ds_read_b64 per MFMA, same wave", d, c, 256);
  run<4, 2>("x32 + 2 ds_read_b64 per MFMA, same wave", d, c, 256);
  run<5, 1>("waves 0-3 x32 only | waves 4-7 1 ds_read_b64 per slot", d, c, 512);
  run<5, 2>("waves 0-3 x32 only | waves 4-7 2 ds_read_b64 per slot", d, c, 512);
  run<5, 4>("waves 0-3 x32 only | waves 4-7 4 ds_read_b64 per slot", d, c, 512);
  return 0;
}
