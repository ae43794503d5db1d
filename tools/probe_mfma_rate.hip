// Hardware probe (not product code): issue interval of v_mfma_f32_4x4x1_16b_f32 with / without the CBSZ/ABID
// A-broadcast, with distinct operand registers, at 1 / 2 / 4 waves per SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1;} } while (0)

template <int BCAST>
__global__ void k(float* out, int iters, float av) {
  f32x4 acc[12];
  for (int i = 0; i < 12; ++i) acc[i] = (f32x4){0, 0, 0, 0};
  float a[4], b[6];
  for (int i = 0; i < 4; ++i) a[i] = av + threadIdx.x * 0.001f + i;
  for (int i = 0; i < 6; ++i) b[i] = av * 0.5f + threadIdx.x * 0.002f + i;
  long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; ++it) {
#define MM(i, ai, bi, id) acc[i] = BCAST ? __builtin_amdgcn_mfma_f32_4x4x1f32(a[ai], b[bi], acc[i], 4, id, 0) \
                                         : __builtin_amdgcn_mfma_f32_4x4x1f32(a[ai], b[bi], acc[i], 0, 0, 0);
    MM(0, 0, 0, 0) MM(1, 1, 1, 1) MM(2, 2, 2, 2) MM(3, 3, 3, 3) MM(4, 0, 4, 4) MM(5, 1, 5, 5)
    MM(6, 2, 0, 6) MM(7, 3, 1, 7) MM(8, 0, 2, 8) MM(9, 1, 3, 9) MM(10, 2, 4, 10) MM(11, 3, 5, 11)
  }
  long long t1 = __builtin_amdgcn_s_memtime();
  float s = 0;
  for (int i = 0; i < 12; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (threadIdx.x == 0 && blockIdx.x == 0) out[1 << 22] = (float)(t1 - t0);
}

int main() {
  float* d;
  CK(hipMalloc(&d, ((1 << 22) + 16) * 4));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  const int iters = 20000;
  for (int bc = 0; bc < 2; ++bc)
    for (int wg_per_cu = 1; wg_per_cu <= 4; wg_per_cu *= 2) {
      const int blocks = 256 * wg_per_cu;   // 256-thread blocks: 1 wave per SIMD per block
      for (int rep = 0; rep < 2; ++rep) {
        CK(hipEventRecord(e0));
        if (bc) hipLaunchKernelGGL(k<1>, dim3(blocks), dim3(256), 0, 0, d, iters, 1.f);
        else hipLaunchKernelGGL(k<0>, dim3(blocks), dim3(256), 0, 0, d, iters, 1.f);
        CK(hipEventRecord(e1)); CK(hipDeviceSynchronize());
      }
      float ms; CK(hipEventElapsedTime(&ms, e0, e1));
      float cyc; CK(hipMemcpy(&cyc, d + (1 << 22), 4, hipMemcpyDeviceToHost));
      const double mfmas_per_simd = (double)iters * 12 * wg_per_cu;
      printf("bcast=%d waves/SIMD=%d: %.3f ms  %.1f TFLOP/s  wall cycles/MFMA/SIMD @2.4GHz = %.2f  (wave0 cycles/MFMA %.2f)\n", bc, wg_per_cu, ms,
             blocks * 4.0 * iters * 12 * 512 / ms * 1e-9, ms * 1e-3 * 2.4e9 / mfmas_per_simd, cyc / iters / 12);
    }
  return 0;
}
