// K1 — concat-and-shift cost volume (reference: src/models/rag_model.py:375-383).
//
// HBM-bound pure write: 2C*d planes of h*w per pair (327 MB at the headline
// config) from two [C,h,w] feature maps (5 MB, L2-resident).  For a fixed
// (b, c, i) the output plane has the SAME flat layout as the input plane shifted
// by i elements: out[p] = (x >= i) ? R[p - i] : 0, x = p % w.  A thread therefore
// owns one 16-byte column p..p+3 of the plane for ALL d disparities: it computes
// x once, keeps a 4-wide sliding window of the right feature in registers (one
// new scalar per disparity) and issues d coalesced 16-B stores (1 KiB per
// wave-instruction), with the x<i zero fill done in registers — no memset pass.
#include "common.h"

namespace ragmi {

template <class T, bool VEC>
__global__ __launch_bounds__(256) void costvol_kernel(const T* __restrict__ L, const T* __restrict__ R,
                                                      T* __restrict__ cost, int C, int d, int hw, int w) {
  const int c2 = blockIdx.y;  // 0..2C-1
  const int b = blockIdx.z;
  const int p = (blockIdx.x * 256 + threadIdx.x) * 4;
  if (p >= hw) return;
  const bool right = c2 >= C;
  const int c = right ? c2 - C : c2;
  const T* src = (right ? R : L) + ((int64_t)b * C + c) * hw;
  T* dst = cost + (((int64_t)b * 2 * C + c2) * d) * (int64_t)hw + p;

  int x[4];
  float v[4];
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    x[k] = (p + k) % w;
    v[k] = (p + k < hw) ? ld(src + p + k) : 0.f;
  }
  for (int i = 0; i < d; ++i) {
    float o[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) o[k] = (x[k] >= i) ? v[k] : 0.f;
    if (VEC) {
      st4(dst, o);
    } else {
#pragma unroll
      for (int k = 0; k < 4; ++k)
        if (p + k < hw) st(dst + k, o[k]);
    }
    dst += hw;
    if (right) {  // slide the window one element to the left: window(i+1)[k] = R[p + k - (i+1)]
      v[3] = v[2];
      v[2] = v[1];
      v[1] = v[0];
      const int q = p - i - 1;
      v[0] = q >= 0 ? ld(src + q) : 0.f;
    }
  }
}

}  // namespace ragmi

extern "C" int ragmi_costvol_fwd(const void* left_fea, const void* right_fea, void* cost, int B, int C, int d,
                                 int h, int w, int dtype, void* stream) {
  using namespace ragmi;
  RAGMI_REQUIRE(left_fea && right_fea && cost, RAGMI_EINVAL, "costvol: null pointer");
  RAGMI_REQUIRE(B > 0 && C > 0 && d > 0 && h > 0 && w > 0, RAGMI_EINVAL, "costvol: non-positive size");
  RAGMI_REQUIRE(dtype_ok(dtype), RAGMI_EUNSUPPORTED, "costvol: dtype %d not built", dtype);
  RAGMI_REQUIRE(2 * C <= 65535 && B <= 65535, RAGMI_EUNSUPPORTED, "costvol: B or C too large for the grid");
  const int hw = h * w;
  dim3 grid((unsigned)ceil_div(ceil_div(hw, 4), 256), 2 * C, B);
  const bool vec = (hw % 4 == 0) && aligned4(cost, dtype);
  hipStream_t s = static_cast<hipStream_t>(stream);
  auto go = [&](auto tag) {
    using T = decltype(tag);
    if (vec)
      hipLaunchKernelGGL((costvol_kernel<T, true>), grid, dim3(256), 0, s, (const T*)left_fea, (const T*)right_fea, (T*)cost, C, d, hw, w);
    else
      hipLaunchKernelGGL((costvol_kernel<T, false>), grid, dim3(256), 0, s, (const T*)left_fea, (const T*)right_fea, (T*)cost, C, d, hw, w);
  };
  if (dtype == RAGMI_BF16) go(bf16_t{}); else go(float{});
  return check_launch("costvol");
}
