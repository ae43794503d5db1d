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
    const float a0 = lerp2(ly.w0, lerp2(lx.w0, v000, lx.w1, v001), ly.w1, lerp2(lx.w0, v010, lx.w1, v011));
    const float a1 = lerp2(ly.w0, lerp2(lx.w0, v100, lx.w1, v101), ly.w1, lerp2(lx.w0, v110, lx.w1, v111));
    const float v = lerp2(lz.w0, a0, lz.w1, a1);
    st(yp + c * ovol, a.relu ? fmaxf(v, 0.f) : v);
  }
}

// Upsampling by two or more (every scale <= 0.5: the head's nn.Upsample and the cells' up-resampling, rag_model.py:146-153, 357-365).
// The generic kernel gathers 8 taps per output element straight from global memory — 8 vector loads per stored dword, and it is the
// texture-address path that saturates (163 MB of output at the headline shape: 54 us for a 27 us write).  Here a workgroup owns
// 4 x 8 x 32 output voxels; the input block they interpolate from (<= 4 x 6 x 18 voxels per channel at scale <= 0.5) is staged
// through LDS (four channels at a time, double-buffered), a thread takes the four output planes of its (y, x): per channel it
// samples the (at most four) input planes
// bilinearly — 4 LDS reads each, shared by the output planes that use them — and blends along z.  Same arithmetic and nesting
// (x, then y, then z) as the generic kernel: results are bit-identical.
constexpr int TU_TZ = 4, TU_TY = 8, TU_TX = 32, TU_PZ = 4, TU_PY = 6, TU_PX = 18, TU_PV = TU_PZ * TU_PY * TU_PX;

constexpr int TU_CG = 4, TU_NPF = (TU_CG * TU_PV + 255) / 256;      // channels per pipeline stage; staging registers per thread

template <class T>
__global__ __launch_bounds__(256) void trilinear_up_kernel(ResampleArgs a) {
  __shared__ float tu_tile[2][TU_CG * TU_PV];            // two stages of [TU_CG][TU_PZ][TU_PY][TU_PX]
  const int tid = threadIdx.x, tx = tid % TU_TX, ty = tid / TU_TX;
  const int nbx = (a.Wo + TU_TX - 1) / TU_TX, nby = (a.Ho + TU_TY - 1) / TU_TY;
  const int ox0 = ((int)blockIdx.x % nbx) * TU_TX, oy0 = ((int)blockIdx.x / nbx % nby) * TU_TY, oz0 = ((int)blockIdx.x / (nbx * nby)) * TU_TZ;
  const int b = blockIdx.y;
  const int pz0 = lin_index(oz0, a.Di, a.Do, a.sd, a.align).i0, py0 = lin_index(oy0, a.Hi, a.Ho, a.sh, a.align).i0,
            px0 = lin_index(ox0, a.Wi, a.Wo, a.sw, a.align).i0;
  const int64_t ivol = (int64_t)a.Di * a.Hi * a.Wi, ovol = (int64_t)a.Do * a.Ho * a.Wo;
  const T* xp = static_cast<const T*>(a.x) + (int64_t)b * a.x_bstride;
  // this thread's staging elements: the same (channel-in-group, voxel) slots for every channel group — offsets computed once
  int soff[TU_NPF];
#pragma unroll
  for (int j = 0; j < TU_NPF; ++j) {
    const int e = min(tid + j * 256, TU_CG * TU_PV - 1), c = min(e / TU_PV, a.C - 1), r = e % TU_PV;   // C < TU_CG: copies of the last channel
    const int gz = min(pz0 + r / (TU_PY * TU_PX), a.Di - 1), gy = min(py0 + r / TU_PX % TU_PY, a.Hi - 1), gx = min(px0 + r % TU_PX, a.Wi - 1);
    soff[j] = (int)(c * ivol + ((int64_t)gz * a.Hi + gy) * a.Wi + gx);      // < 2^31: checked on the host
  }
  float pf[TU_NPF];
  auto prefetch = [&](int c0) {              // channels c0 .. c0 + TU_CG - 1 (clamped: the copies past C are never read)
    const T* src = xp + (int64_t)min(c0, max(a.C - TU_CG, 0)) * ivol;
#pragma unroll
    for (int j = 0; j < TU_NPF; ++j) pf[j] = ld(src + soff[j]);
  };
  auto commit = [&](float* buf) {
#pragma unroll
    for (int j = 0; j < TU_NPF; ++j)
      if (tid + j * 256 < TU_CG * TU_PV) buf[tid + j * 256] = pf[j];
  };
  const int ox = min(ox0 + tx, a.Wo - 1), oy = min(oy0 + ty, a.Ho - 1);      // out-of-range threads shadow the last voxel, store nothing
  const LinIdx ly = lin_index(oy, a.Hi, a.Ho, a.sh, a.align);
  const LinIdx lx = lin_index(ox, a.Wi, a.Wo, a.sw, a.align);
  const int o00 = (ly.i0 - py0) * TU_PX + (lx.i0 - px0), o01 = o00 + (lx.i1 - lx.i0);
  const int o10 = o00 + (ly.i1 - ly.i0) * TU_PX, o11 = o10 + (lx.i1 - lx.i0);
  LinIdx lz[TU_TZ];
#pragma unroll
  for (int k = 0; k < TU_TZ; ++k) lz[k] = lin_index(min(oz0 + k, a.Do - 1), a.Di, a.Do, a.sd, a.align);     // wave-uniform
  const bool inside = ox0 + tx < a.Wo && oy0 + ty < a.Ho;
  T* yp = static_cast<T*>(a.y) + (int64_t)b * a.y_bstride + (int64_t)a.y_ch0 * ovol + ((int64_t)oz0 * a.Ho + oy) * a.Wo + ox;
  const int64_t zs = (int64_t)a.Ho * a.Wo;
  // channel groups flow through two LDS stages: group g+1 travels global -> registers while group g is interpolated
  // (staged all at once, the 27 us of load latency sat in front of every workgroup's first output)
  const int ng = (a.C + TU_CG - 1) / TU_CG;
  // a partial last group re-reads channels C-TU_CG.. (clamped above): its channel index inside the stage shifts accordingly
  prefetch(0);
  commit(tu_tile[0]);
  __syncthreads();
  for (int g = 0; g < ng; ++g) {
    if (g + 1 < ng) prefetch((g + 1) * TU_CG);
    const float* buf = tu_tile[g & 1];
    const int cbase = min(g * TU_CG, max(a.C - TU_CG, 0));                   // first channel held by this stage
#pragma unroll
    for (int cc = 0; cc < TU_CG; ++cc) {
      const int c = cbase + cc;
      if (c < g * TU_CG || c >= a.C) continue;                               // already written by the previous group / past the end
      const float* t = buf + cc * TU_PV;
      float pl[TU_PZ];
#pragma unroll
      for (int p = 0; p < TU_PZ; ++p) {
        const float* q = t + p * (TU_PY * TU_PX);
        pl[p] = lerp2(ly.w0, lerp2(lx.w0, q[o00], lx.w1, q[o01]), ly.w1, lerp2(lx.w0, q[o10], lx.w1, q[o11]));     // x innermost, then y (ATen's nesting)
      }
#pragma unroll
      for (int k = 0; k < TU_TZ; ++k) {
        const int i0 = lz[k].i0 - pz0, i1 = lz[k].i1 - pz0;      // wave-uniform plane picks
        const float a0 = i0 == 0 ? pl[0] : i0 == 1 ? pl[1] : i0 == 2 ? pl[2] : pl[3];
        const float a1 = i1 == 0 ? pl[0] : i1 == 1 ? pl[1] : i1 == 2 ? pl[2] : pl[3];
        const float v = lerp2(lz[k].w0, a0, lz[k].w1, a1);
        if (inside && oz0 + k < a.Do) st(yp + c * ovol + k * zs, a.relu ? fmaxf(v, 0.f) : v);
      }
    }
    if (g + 1 < ng) commit(tu_tile[(g + 1) & 1]);
    __syncthreads();
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
  // scale <= 0.5 on every axis bounds the input block of a 4 x 8 x 32 output tile by 4 x 6 x 18 (i0 of the last output is at most
  // floor(0.5 * (T - 1)) + 1 past i0 of the first, and i1 one further)
  const int64_t nblk = ceil_div(Wo, TU_TX) * ceil_div(Ho, TU_TY) * ceil_div(Do, TU_TZ);
  if (a.sd <= 0.5f && a.sh <= 0.5f && a.sw <= 0.5f && nblk < (1ll << 31) && (int64_t)C * Di * Hi * Wi < (1ll << 31)) {
    dim3 ugrid((unsigned)nblk, B);
    if (dtype == RAGMI_BF16) hipLaunchKernelGGL(trilinear_up_kernel<bf16_t>, ugrid, dim3(256), 0, static_cast<hipStream_t>(stream), a);
    else hipLaunchKernelGGL(trilinear_up_kernel<float>, ugrid, dim3(256), 0, static_cast<hipStream_t>(stream), a);
    return check_launch(what);
  }
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
