// Feature-Net stem: 3x3 / pad 1 / stride S 2-D ConvBR (reference: stem2d1 = ConvBR_2d(6, 12, 3, stride=3, padding=1),
// src/models/rag_model.py:201; ConvBR_2d src/automl/operations_2d.py:31-47).  SURVEY.md §8(f) row N1.
//
// HBM-bound (reads the full-resolution 6-channel image once, writes 1/9 of the pixels): a thread owns one OUTPUT
// pixel and all NCO output channels; the 3x3xCin weights and the folded BN live in LDS (global reads of them inside
// the loop could alias the stores and would not be scalar).
#include "common.h"

namespace ragmi {

struct C2SArgs {
  const void* x;
  void* y;
  const float* w;      // [Cout][Cin][3][3]
  const float* scale;
  const float* shift;
  int Cin, Cout, H, W, Ho, Wo, stride, relu;
};

template <class T, int NCO>
__global__ __launch_bounds__(256) void conv2d_k3_strided_kernel(C2SArgs a) {
  extern __shared__ float wl[];   // [Cout][Cin*9] then scale[Cout], shift[Cout]
  const int nw = a.Cout * a.Cin * 9;
  for (int i = threadIdx.x; i < nw; i += 256) wl[i] = a.w[i];
  for (int i = threadIdx.x; i < a.Cout; i += 256) {
    wl[nw + i] = a.scale ? a.scale[i] : 1.f;
    wl[nw + a.Cout + i] = a.scale ? a.shift[i] : 0.f;
  }
  __syncthreads();
  const int64_t opix = (int64_t)a.Ho * a.Wo;
  const int64_t o = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (o >= opix) return;
  const int b = blockIdx.y, co0 = blockIdx.z * NCO;
  const int ox = (int)(o % a.Wo), oy = (int)(o / a.Wo);
  const int ix0 = ox * a.stride - 1, iy0 = oy * a.stride - 1;
  const T* xb = static_cast<const T*>(a.x) + (int64_t)b * a.Cin * a.H * a.W;
  float acc[NCO];
#pragma unroll
  for (int j = 0; j < NCO; ++j) acc[j] = 0.f;
  for (int ci = 0; ci < a.Cin; ++ci) {
    float v[9];
#pragma unroll
    for (int t = 0; t < 9; ++t) {
      const int iy = iy0 + t / 3, ix = ix0 + t % 3;
      const bool ok = (unsigned)iy < (unsigned)a.H && (unsigned)ix < (unsigned)a.W;
      v[t] = ok ? ld(xb + ((int64_t)ci * a.H + iy) * a.W + ix) : 0.f;
    }
#pragma unroll
    for (int j = 0; j < NCO; ++j) {
      const int co = co0 + j;
      if (co >= a.Cout) break;
      const float* wr = wl + (co * a.Cin + ci) * 9;
#pragma unroll
      for (int t = 0; t < 9; ++t) acc[j] = fmaf(wr[t], v[t], acc[j]);
    }
  }
  T* yb = static_cast<T*>(a.y) + (int64_t)b * a.Cout * opix + o;
#pragma unroll
  for (int j = 0; j < NCO; ++j) {
    const int co = co0 + j;
    if (co >= a.Cout) break;
    float r = fmaf(acc[j], wl[nw + co], wl[nw + a.Cout + co]);
    st(yb + (int64_t)co * opix, a.relu ? fmaxf(r, 0.f) : r);
  }
}

}  // namespace ragmi

extern "C" int ragmi_conv2d_k3_strided_fwd(const void* x, const void* weight, const void* scale, const void* shift, int relu,
                                           void* y, int B, int Cin, int Cout, int H, int W, int stride, int dtype, void* stream) {
  using namespace ragmi;
  RAGMI_REQUIRE(x && weight && y, RAGMI_EINVAL, "conv2d_k3_strided: null pointer");
  RAGMI_REQUIRE((scale == nullptr) == (shift == nullptr), RAGMI_EINVAL, "conv2d_k3_strided: scale/shift must both be given or both NULL");
  RAGMI_REQUIRE(B > 0 && Cin > 0 && Cout > 0 && H > 0 && W > 0 && stride >= 1, RAGMI_EINVAL, "conv2d_k3_strided: bad size");
  RAGMI_REQUIRE(dtype_ok(dtype), RAGMI_EUNSUPPORTED, "conv2d_k3_strided: dtype %d not built", dtype);
  const size_t lds = ((size_t)Cout * Cin * 9 + 2 * Cout) * sizeof(float);
  RAGMI_REQUIRE(B <= 65535 && lds <= 48 * 1024, RAGMI_EUNSUPPORTED, "conv2d_k3_strided: B or Cout*Cin too large");
  const int Ho = (H + 2 - 3) / stride + 1, Wo = (W + 2 - 3) / stride + 1;
  C2SArgs a{x, y, (const float*)weight, (const float*)scale, (const float*)shift, Cin, Cout, H, W, Ho, Wo, stride, relu};
  constexpr int NCO = 12;
  dim3 grid((unsigned)ceil_div((int64_t)Ho * Wo, 256), B, (unsigned)ceil_div(Cout, NCO));
  hipStream_t s = static_cast<hipStream_t>(stream);
  if (dtype == RAGMI_BF16) hipLaunchKernelGGL((conv2d_k3_strided_kernel<bf16_t, NCO>), grid, dim3(256), lds, s, a);
  else hipLaunchKernelGGL((conv2d_k3_strided_kernel<float, NCO>), grid, dim3(256), lds, s, a);
  return check_launch("conv2d_k3_strided");
}
