// K2 — fused Disp.forward: trilinear x3 upsample (align_corners=False) -> Softmin over the
// disparity axis -> DisparityRegression.  Reference: src/models/rag_model.py:18-44.
//
// The reference materialises three [B,192,H,W] tensors (3.7 GB of traffic at the headline
// config); here each output pixel streams the d coarse planes once (4 bilinear taps per
// plane, L1/L2-served: neighbouring pixels share taps), lerps along the disparity axis in
// registers and keeps an online softmax (running max / sum / disparity-weighted sum).
// Algorithmic traffic: read d*h*w*4 B, write Ho*Wo*4 B per pair (15.5 MB at headline).
#include "common.h"

namespace ragmi {

struct DispArgs {
  const void* cost;   // [B, d, h, w], float or bf16_t
  float* out;         // [B, Ho, Wo]
  int d, h, w, maxdisp, Ho, Wo;
  float sd, sh, sw;
};

template <class T>
__global__ __launch_bounds__(256) void disp_softargmin_kernel(DispArgs a) {
  extern __shared__ float4 ztab[];                 // [maxdisp]
  for (int dd = threadIdx.x; dd < a.maxdisp; dd += 256) {
    const LinIdx lz = lin_index(dd, a.d, a.maxdisp, a.sd, 0);
    const bool same = lz.i1 == lz.i0;
    ztab[dd] = make_float4((float)lz.i0, lz.w0, same ? 0.f : lz.w1, same ? lz.w1 : 0.f);
  }
  __syncthreads();
  const int64_t npix = (int64_t)a.Ho * a.Wo;
  const int64_t o = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (o >= npix) return;
  const int b = blockIdx.y;
  const int ox = (int)(o % a.Wo), oy = (int)(o / a.Wo);
  const LinIdx ly = lin_index(oy, a.h, a.Ho, a.sh, 0);
  const LinIdx lx = lin_index(ox, a.w, a.Wo, a.sw, 0);
  const int hw = a.h * a.w;
  const T* base = static_cast<const T*>(a.cost) + (int64_t)b * a.d * hw;
  const int o00 = ly.i0 * a.w + lx.i0, o01 = ly.i0 * a.w + lx.i1;
  const int o10 = ly.i1 * a.w + lx.i0, o11 = ly.i1 * a.w + lx.i1;
  const float w00 = ly.w0 * lx.w0, w01 = ly.w0 * lx.w1, w10 = ly.w1 * lx.w0, w11 = ly.w1 * lx.w1;
  (void)w00; (void)w01; (void)w10; (void)w11;

  auto plane = [&](int z) -> float {  // bilinear sample of coarse plane z at (oy, ox); x innermost like ATen
    const T* p = base + (int64_t)z * hw;
    return ly.w0 * (lx.w0 * ld(p + o00) + lx.w1 * ld(p + o01)) + ly.w1 * (lx.w0 * ld(p + o10) + lx.w1 * ld(p + o11));
  };

  // Walk the fine disparities in order; the coarse pair (cz, cz+1) only ever moves forward, so each
  // plane is sampled once.  Online softmax of -cost in base 2, branch-free:
  //   m' = max(m, t); s = s*2^((m-m')k) + 2^((t-m')k); differences are formed BEFORE scaling by k=log2(e)
  //   so large |cost| does not lose the bits that matter near the maximum.
  constexpr float K = 1.4426950408889634f;
  // The fine-disparity taps (i0, i1 == i0, w0, w1) are the same for every pixel: one table per workgroup in LDS instead
  // of ~12 VALU instructions of index arithmetic per fine sample per pixel (the kernel is VALU-bound).
  int cz = 0;
  float b0 = plane(0), b1 = plane(a.d > 1 ? 1 : 0);
  float m = -INFINITY, s = 0.f, ws = 0.f;
  for (int dd = 0; dd < a.maxdisp; ++dd) {
    const float4 tb = ztab[dd];                    // (i0, w0, w1 if i1 != i0 else 0, w1 if i1 == i0 else 0)
    const int i0 = (int)tb.x;
    while (i0 > cz) {   // rarely more than one step (only when down-sampling the disparity axis)
      ++cz;
      b0 = b1;
      b1 = plane(cz + 1 < a.d ? cz + 1 : a.d - 1);
    }
    const float t = -fmaf(tb.y + tb.w, b0, tb.z * b1);
    // one of (m - m'), (t - m') is exactly 0 and 2^0 is exactly 1: ONE transcendental per sample, the same bits as two
    const bool up = t > m;                                   // the running maximum moves
    const float x = __builtin_amdgcn_exp2f((up ? m - t : t - m) * K);   // 2^(-inf) = 0 on the first sample
    const float r = up ? x : 1.f, e = up ? 1.f : x;
    s = fmaf(s, r, e);
    ws = fmaf(ws, r, e * (float)dd);
    m = up ? t : m;
  }
  a.out[(int64_t)b * npix + o] = ws / s;
}

// standalone DisparityRegression: out = sum_d prob[:, d] * d
template <class T>
__global__ __launch_bounds__(256) void disparity_regression_kernel(const T* __restrict__ prob, float* __restrict__ out,
                                                                  int D, int64_t hw) {
  const int64_t o = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (o >= hw) return;
  const int b = blockIdx.y;
  const T* p = prob + (int64_t)b * D * hw + o;
  float acc = 0.f;
#pragma unroll 8
  for (int dd = 0; dd < D; ++dd) acc = fmaf(ld(p + (int64_t)dd * hw), (float)dd, acc);
  out[(int64_t)b * hw + o] = acc;
}

}  // namespace ragmi

extern "C" int ragmi_disp_softargmin_fwd(const void* cost, void* out, int B, int d, int h, int w, int maxdisp, int Ho,
                                         int Wo, int dtype, void* stream) {
  using namespace ragmi;
  RAGMI_REQUIRE(cost && out, RAGMI_EINVAL, "disp_softargmin: null pointer");
  RAGMI_REQUIRE(B > 0 && d > 0 && h > 0 && w > 0 && maxdisp > 0 && Ho > 0 && Wo > 0, RAGMI_EINVAL,
                "disp_softargmin: non-positive size");
  RAGMI_REQUIRE(dtype_ok(dtype), RAGMI_EUNSUPPORTED, "disp_softargmin: dtype %d not built", dtype);
  RAGMI_REQUIRE(B <= 65535 && (int64_t)h * w < (1ll << 30), RAGMI_EUNSUPPORTED, "disp_softargmin: size too large");
  DispArgs a{cost, (float*)out, d, h, w, maxdisp, Ho, Wo,
             lin_scale(d, maxdisp, 0), lin_scale(h, Ho, 0), lin_scale(w, Wo, 0)};
  dim3 grid((unsigned)ceil_div((int64_t)Ho * Wo, 256), B);
  const size_t lds = (size_t)maxdisp * sizeof(float4);
  RAGMI_REQUIRE(lds <= 64 * 1024, RAGMI_EUNSUPPORTED, "disp_softargmin: maxdisp %d exceeds the tap table (4096)", maxdisp);
  if (dtype == RAGMI_BF16) hipLaunchKernelGGL(disp_softargmin_kernel<bf16_t>, grid, dim3(256), lds, static_cast<hipStream_t>(stream), a);
  else hipLaunchKernelGGL(disp_softargmin_kernel<float>, grid, dim3(256), lds, static_cast<hipStream_t>(stream), a);
  return check_launch("disp_softargmin");
}

extern "C" int ragmi_disparity_regression_fwd(const void* prob, void* out, int B, int D, int H, int W, int dtype,
                                              void* stream) {
  using namespace ragmi;
  RAGMI_REQUIRE(prob && out, RAGMI_EINVAL, "disparity_regression: null pointer");
  RAGMI_REQUIRE(B > 0 && D > 0 && H > 0 && W > 0, RAGMI_EINVAL, "disparity_regression: non-positive size");
  RAGMI_REQUIRE(dtype_ok(dtype), RAGMI_EUNSUPPORTED, "disparity_regression: dtype %d not built", dtype);
  RAGMI_REQUIRE(B <= 65535, RAGMI_EUNSUPPORTED, "disparity_regression: B too large");
  const int64_t hw = (int64_t)H * W;
  dim3 grid((unsigned)ceil_div(hw, 256), B);
  if (dtype == RAGMI_BF16)
    hipLaunchKernelGGL(disparity_regression_kernel<bf16_t>, grid, dim3(256), 0, static_cast<hipStream_t>(stream), (const bf16_t*)prob, (float*)out, D, hw);
  else
    hipLaunchKernelGGL(disparity_regression_kernel<float>, grid, dim3(256), 0, static_cast<hipStream_t>(stream), (const float*)prob, (float*)out, D, hw);
  return check_launch("disparity_regression");
}
