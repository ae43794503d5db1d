// K5 — trilinear resample with ATen's source-index rule (F.interpolate mode='trilinear').
// Reference call sites: Cell_3d down/up-sampling, align_corners=True
// (src/models/rag_model.py:146-153) and the matching head's nn.Upsample (:357-358).
//
// HBM-bound gather.  A thread owns one output voxel (x fastest -> coalesced stores),
// computes its 3 (index, weight) pairs once and reuses them for every channel.
#include "common.h"

namespace ragmi {

struct ResampleArgs {
  const void* x;
  void* y;
  int C, Di, Hi, Wi, Do, Ho, Wo;
  float sd, sh, sw;
  int align;
  int64_t x_bstride, y_bstride;   // elements between batch items (channel slices of wider buffers)
  int y_ch0, relu;
};

template <class T>
__global__ __launch_bounds__(256) void trilinear_kernel(ResampleArgs a) {
  const int64_t ovol = (int64_t)a.Do * a.Ho * a.Wo;
  const int64_t o = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (o >= ovol) return;
  const int b = blockIdx.y;
  const int ox = (int)(o % a.Wo);
  const int64_t t = o / a.Wo;
  const int oy = (int)(t % a.Ho);
  const int oz = (int)(t / a.Ho);
  const LinIdx lz = lin_index(oz, a.Di, a.Do, a.sd, a.align);
  const LinIdx ly = lin_index(oy, a.Hi, a.Ho, a.sh, a.align);
  const LinIdx lx = lin_index(ox, a.Wi, a.Wo, a.sw, a.align);
  const int64_t ivol = (int64_t)a.Di * a.Hi * a.Wi;
  const int64_t r00 = ((int64_t)lz.i0 * a.Hi + ly.i0) * a.Wi, r01 = ((int64_t)lz.i0 * a.Hi + ly.i1) * a.Wi;
  const int64_t r10 = ((int64_t)lz.i1 * a.Hi + ly.i0) * a.Wi, r11 = ((int64_t)lz.i1 * a.Hi + ly.i1) * a.Wi;
  const T* xp = static_cast<const T*>(a.x) + (int64_t)b * a.x_bstride;
  T* yp = static_cast<T*>(a.y) + (int64_t)b * a.y_bstride + (int64_t)a.y_ch0 * ovol + o;
#pragma unroll 1   // measured on the 12-channel head upsample: 1 -> 53 us, 2 -> 57 us, 4 -> 79 us (store-bound: issue the stores early)
  for (int c = 0; c < a.C; ++c) {
    const T* pc = xp + c * ivol;
    const float v000 = ld(pc + r00 + lx.i0), v001 = ld(pc + r00 + lx.i1);
    const float v010 = ld(pc + r01 + lx.i0), v011 = ld(pc + r01 + lx.i1);
    const float v100 = ld(pc + r10 + lx.i0), v101 = ld(pc + r10 + lx.i1);
    const float v110 = ld(pc + r11 + lx.i0), v111 = ld(pc + r11 + lx.i1);
    // x innermost, then y, then z (ATen's cpu_upsample_linear nesting)
    const float a0 = ly.w0 * (lx.w0 * v000 + lx.w1 * v001) + ly.w1 * (lx.w0 * v010 + lx.w1 * v011);
    const float a1 = ly.w0 * (lx.w0 * v100 + lx.w1 * v101) + ly.w1 * (lx.w0 * v110 + lx.w1 * v111);
    const float v = lz.w0 * a0 + lz.w1 * a1;
    st(yp + c * ovol, a.relu ? fmaxf(v, 0.f) : v);
  }
}

}  // namespace ragmi

static int trilinear_launch(const void* x, int64_t x_bstride, void* y, int64_t y_bstride, int y_ch0, int relu, int B, int C, int Di,
                            int Hi, int Wi, int Do, int Ho, int Wo, int align_corners, int dtype, void* stream, const char* what) {
  using namespace ragmi;
  RAGMI_REQUIRE(x && y, RAGMI_EINVAL, "%s: null pointer", what);
  RAGMI_REQUIRE(B > 0 && C > 0 && Di > 0 && Hi > 0 && Wi > 0 && Do > 0 && Ho > 0 && Wo > 0 && y_ch0 >= 0, RAGMI_EINVAL,
                "%s: non-positive size", what);
  RAGMI_REQUIRE(dtype_ok(dtype), RAGMI_EUNSUPPORTED, "%s: dtype %d not built", what, dtype);
  RAGMI_REQUIRE(B <= 65535, RAGMI_EUNSUPPORTED, "%s: B too large", what);
  ResampleArgs a{x, y, C, Di, Hi, Wi, Do, Ho, Wo,
                 lin_scale(Di, Do, align_corners), lin_scale(Hi, Ho, align_corners), lin_scale(Wi, Wo, align_corners),
                 align_corners ? 1 : 0, x_bstride, y_bstride, y_ch0, relu ? 1 : 0};
  const int64_t ovol = (int64_t)Do * Ho * Wo;
  dim3 grid((unsigned)ceil_div(ovol, 256), B);
  if (dtype == RAGMI_BF16) hipLaunchKernelGGL(trilinear_kernel<bf16_t>, grid, dim3(256), 0, static_cast<hipStream_t>(stream), a);
  else hipLaunchKernelGGL(trilinear_kernel<float>, grid, dim3(256), 0, static_cast<hipStream_t>(stream), a);
  return check_launch(what);
}

extern "C" int ragmi_trilinear3d_fwd(const void* x, void* y, int B, int C, int Di, int Hi, int Wi, int Do, int Ho,
                                     int Wo, int align_corners, int dtype, void* stream) {
  return trilinear_launch(x, (int64_t)C * Di * Hi * Wi, y, (int64_t)C * Do * Ho * Wo, 0, 0, B, C, Di, Hi, Wi, Do, Ho, Wo, align_corners,
                          dtype, stream, "trilinear3d");
}

extern "C" int ragmi_trilinear3d_act_fwd(const void* x, int64_t x_bstride, void* y, int64_t y_bstride, int y_ch0, int relu, int B, int C,
                                         int Di, int Hi, int Wi, int Do, int Ho, int Wo, int align_corners, int dtype, void* stream) {
  return trilinear_launch(x, x_bstride, y, y_bstride, y_ch0, relu, B, C, Di, Hi, Wi, Do, Ho, Wo, align_corners, dtype, stream,
                          "trilinear3d_act");
}
