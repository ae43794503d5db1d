// Hardware probe (not product code): semantics + rate of v_mfma_f32_4x4x1_16b_f32
// with CBSZ/ABID A-broadcast, and gfx9 wave_shr/wave_shl DPP.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <cmath>
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int ABID>
__global__ void k_bcast(const float* a, const float* b, float* d) {
  int l = threadIdx.x;
  f32x4 c = {0, 0, 0, 0};
  c = __builtin_amdgcn_mfma_f32_4x4x1f32(a[l], b[l], c, 4, ABID, 0);
  for (int j = 0; j < 4; ++j) d[l * 4 + j] = c[j];
}
__global__ void k_plain(const float* a, const float* b, float* d) {
  int l = threadIdx.x;
  f32x4 c = {0, 0, 0, 0};
  c = __builtin_amdgcn_mfma_f32_4x4x1f32(a[l], b[l], c, 0, 0, 0);
  for (int j = 0; j < 4; ++j) d[l * 4 + j] = c[j];
}
__global__ void k_dpp(const float* a, float* d) {
  int l = threadIdx.x;
  float v = a[l];
  int vi = __float_as_int(v);
  int shr = __builtin_amdgcn_update_dpp(0, vi, 0x138, 0xf, 0xf, true);  // wave_shr:1
  int shl = __builtin_amdgcn_update_dpp(0, vi, 0x130, 0xf, 0xf, true);  // wave_shl:1
  d[l] = __int_as_float(shr);
  d[64 + l] = __int_as_float(shl);
}
// rate: NACC independent accumulators, ITER iterations
template <int NACC>
__global__ void k_rate(float* out, int iters, float av, float bv) {
  f32x4 acc[NACC];
  for (int i = 0; i < NACC; ++i) acc[i] = (f32x4){0, 0, 0, 0};
  float a = av + threadIdx.x, b = bv;
  long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; ++it) {
#define MM(i) acc[i] = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, acc[i], 4, i, 0);
    MM(0) MM(1) MM(2) MM(3) MM(4) MM(5) MM(6) MM(7) MM(8) MM(9) MM(10) MM(11)
  }
  long long t1 = __builtin_amdgcn_s_memtime();
  float s = 0;
  for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (threadIdx.x == 0 && blockIdx.x == 0) out[1 << 20] = (float)(t1 - t0);
}
template <int NACC>
__global__ void k_rate16(float* out, int iters, float av, float bv) {
  f32x4 acc[NACC];
  for (int i = 0; i < NACC; ++i) acc[i] = (f32x4){0, 0, 0, 0};
  float a = av + threadIdx.x, b = bv;
  long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < NACC; ++i)
      acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[i], 0, 0, 0);
  }
  long long t1 = __builtin_amdgcn_s_memtime();
  float s = 0;
  for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (threadIdx.x == 0 && blockIdx.x == 0) out[1 << 20] = (float)(t1 - t0);
}
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

int main() {
  float *a, *b, *d;
  CK(hipMalloc(&a, 256 * 4)); CK(hipMalloc(&b, 256 * 4)); CK(hipMalloc(&d, ((1 << 20) + 16) * 4));
  std::vector<float> ha(64), hb(64), hd(256);
  for (int i = 0; i < 64; ++i) { ha[i] = 1 + i; hb[i] = 100.f + 3 * i + (i % 5) * 0.5f; }
  CK(hipMemcpy(a, ha.data(), 256, hipMemcpyHostToDevice));
  CK(hipMemcpy(b, hb.data(), 256, hipMemcpyHostToDevice));
  // plain: expect D[lane=4blk+n][reg m] = A[lane 4blk+m] * B[lane 4blk+n]
  k_plain<<<1, 64>>>(a, b, d); CK(hipDeviceSynchronize());
  CK(hipMemcpy(hd.data(), d, 1024, hipMemcpyDeviceToHost));
  int bad = 0;
  for (int l = 0; l < 64; ++l) for (int m = 0; m < 4; ++m) {
    float e = ha[(l & ~3) + m] * hb[l];
    if (hd[l * 4 + m] != e) { if (bad < 4) printf("plain mismatch l=%d m=%d got %g exp %g\n", l, m, hd[l*4+m], e); ++bad; }
  }
  printf("PLAIN layout D[4b+n][m]=A[4b+m]*B[4b+n]: %s (%d bad)\n", bad ? "FAIL" : "OK", bad);
  // bcast abid=5: expect A from lanes 20..23
  k_bcast<5><<<1, 64>>>(a, b, d); CK(hipDeviceSynchronize());
  CK(hipMemcpy(hd.data(), d, 1024, hipMemcpyDeviceToHost));
  bad = 0;
  for (int l = 0; l < 64; ++l) for (int m = 0; m < 4; ++m) {
    float e = ha[20 + m] * hb[l];
    if (hd[l * 4 + m] != e) { if (bad < 4) printf("bcast mismatch l=%d m=%d got %g exp %g\n", l, m, hd[l*4+m], e); ++bad; }
  }
  printf("BCAST cbsz=4 abid=5 D[l][m]=A[20+m]*B[l]: %s (%d bad)\n", bad ? "FAIL" : "OK", bad);
  k_bcast<15><<<1, 64>>>(a, b, d); CK(hipDeviceSynchronize());
  CK(hipMemcpy(hd.data(), d, 1024, hipMemcpyDeviceToHost));
  bad = 0;
  for (int l = 0; l < 64; ++l) for (int m = 0; m < 4; ++m) if (hd[l*4+m] != ha[60+m]*hb[l]) ++bad;
  printf("BCAST cbsz=4 abid=15: %s (%d bad)\n", bad ? "FAIL" : "OK", bad);
  // dpp
  k_dpp<<<1, 64>>>(a, d); CK(hipDeviceSynchronize());
  CK(hipMemcpy(hd.data(), d, 512, hipMemcpyDeviceToHost));
  printf("DPP wave_shr1: lane0=%g lane1=%g lane16=%g lane32=%g lane63=%g (expect 0,1,16,32,63 if lane l gets lane l-1 of a=1+i)\n", hd[0], hd[1], hd[16], hd[32], hd[63]);
  printf("DPP wave_shl1: lane0=%g lane15=%g lane31=%g lane62=%g lane63=%g (expect 2,17,33,64,0)\n", hd[64], hd[64+15], hd[64+31], hd[64+62], hd[64+63]);
  // rate
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  const int iters = 20000;
  {
    const int NACC = 12;
    k_rate<NACC><<<1024, 256>>>(d, 100, 1.f, 1.f); CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0)); k_rate<NACC><<<1024, 256>>>(d, iters, 1.f, 1.f); CK(hipEventRecord(e1)); CK(hipDeviceSynchronize());
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    float cyc; CK(hipMemcpy(&cyc, d + (1 << 20), 4, hipMemcpyDeviceToHost));
    double fl = 1024.0 * 4 * iters * NACC * 512.0;
    printf("4x4x1 rate: %.3f ms, %.1f TFLOP/s, %.2f cycles/MFMA/wave (1 wave/SIMD view: cyc/iters/NACC)\n", ms, fl / ms * 1e-9, cyc / iters / NACC);
  }
  {
    const int NACC = 4;
    k_rate16<NACC><<<1024, 256>>>(d, 100, 1.f, 1.f); CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0)); k_rate16<NACC><<<1024, 256>>>(d, iters, 1.f, 1.f); CK(hipEventRecord(e1)); CK(hipDeviceSynchronize());
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    float cyc; CK(hipMemcpy(&cyc, d + (1 << 20), 4, hipMemcpyDeviceToHost));
    double fl = 1024.0 * 4 * iters * NACC * 2048.0;
    printf("16x16x4 rate: %.3f ms, %.1f TFLOP/s, %.2f cycles/MFMA/wave\n", ms, fl / ms * 1e-9, cyc / iters / NACC);
  }
  return 0;
}
