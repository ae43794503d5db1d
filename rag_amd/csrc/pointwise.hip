// K4 — 1x1x1 ConvBR_3d (channel mix + folded BN + ReLU) and the identity-branch add.
// Reference: Cell_3d.pre_preprocess/preprocess (src/models/rag_model.py:125-126, 154-155),
// last_6_3d / last_12_3d (:270-271), ConvBR_3d (src/automl/operations_3d.py:31-47).
//
// HBM-bound (AI = 2*Cin*Cout / 4(Cin+Cout) <= 8 FLOP/B): a thread owns 4 consecutive
// voxels (one 16-B column of every channel plane), streams the Cin planes once with
// coalesced dwordx4 loads, keeps all NCO outputs in registers, and writes NCO coalesced
// dwordx4 stores.  Weights/scale/shift are wave-uniform -> scalar (SGPR) loads.
#include "common.h"

namespace ragmi {

struct K1Args {
  const float* x;
  int64_t x_bstride;
  const float* w;  // [Cout][Cin]
  const float* scale;
  const float* shift;
  float* y;
  int64_t y_bstride;
  int y_ch0;
  int Cin, Cout, co0;
  int64_t dhw;
  int relu;
};

template <int NCO, bool VEC>
__global__ __launch_bounds__(256) void conv_k1_kernel(K1Args a) {
  constexpr int V = VEC ? 4 : 1;
  const int64_t p = ((int64_t)blockIdx.x * 256 + threadIdx.x) * V;
  if (p >= a.dhw) return;
  const int b = blockIdx.y;
  a.co0 = blockIdx.z * NCO;   // output-channel slab of this block
  const float* xp = a.x + b * a.x_bstride + p;
  float acc[NCO][V];
#pragma unroll
  for (int j = 0; j < NCO; ++j)
#pragma unroll
    for (int k = 0; k < V; ++k) acc[j][k] = 0.f;

#pragma unroll 4
  for (int ci = 0; ci < a.Cin; ++ci) {
    float xv[V];
    if (VEC) {
      const float4 t = *reinterpret_cast<const float4*>(xp + (int64_t)ci * a.dhw);
      xv[0] = t.x;
      if (V > 1) { xv[1 % V] = t.y; xv[2 % V] = t.z; xv[3 % V] = t.w; }
    } else {
      xv[0] = xp[(int64_t)ci * a.dhw];
    }
#pragma unroll
    for (int j = 0; j < NCO; ++j) {
      const int co = a.co0 + j;
      const float wv = co < a.Cout ? a.w[(int64_t)co * a.Cin + ci] : 0.f;
#pragma unroll
      for (int k = 0; k < V; ++k) acc[j][k] = fmaf(wv, xv[k], acc[j][k]);
    }
  }
  float* yp = a.y + b * a.y_bstride + p;
#pragma unroll
  for (int j = 0; j < NCO; ++j) {
    const int co = a.co0 + j;
    if (co >= a.Cout) break;
    const float sc = a.scale ? a.scale[co] : 1.f;
    const float sh = a.scale ? a.shift[co] : 0.f;
    float o[V];
#pragma unroll
    for (int k = 0; k < V; ++k) {
      float v = a.scale ? fmaf(acc[j][k], sc, sh) : acc[j][k];
      o[k] = a.relu ? fmaxf(v, 0.f) : v;
    }
    float* dst = yp + (int64_t)(a.y_ch0 + co) * a.dhw;
    if (VEC)
      *reinterpret_cast<float4*>(dst) = make_float4(o[0], o[1 % V], o[2 % V], o[3 % V]);
    else
      dst[0] = o[0];
  }
}

template <bool VEC>
__global__ __launch_bounds__(256) void add_kernel(const float* __restrict__ a, int64_t a_bs, const float* __restrict__ b,
                                                  int64_t b_bs, float* __restrict__ y, int64_t y_bs, int64_t n) {
  constexpr int V = VEC ? 4 : 1;
  const int64_t p = ((int64_t)blockIdx.x * 256 + threadIdx.x) * V;
  if (p >= n) return;
  const int bi = blockIdx.y;
  if (VEC) {
    const float4 u = *reinterpret_cast<const float4*>(a + bi * a_bs + p);
    const float4 v = *reinterpret_cast<const float4*>(b + bi * b_bs + p);
    *reinterpret_cast<float4*>(y + bi * y_bs + p) = make_float4(u.x + v.x, u.y + v.y, u.z + v.z, u.w + v.w);
  } else {
    y[bi * y_bs + p] = a[bi * a_bs + p] + b[bi * b_bs + p];
  }
}

template <int NCO, bool VEC>
static void launch_k1_nco(const K1Args& a, int B, hipStream_t s) {
  constexpr int V = VEC ? 4 : 1;
  dim3 grid((unsigned)ceil_div(ceil_div(a.dhw, V), 256), B, (unsigned)ceil_div(a.Cout, NCO));
  hipLaunchKernelGGL((conv_k1_kernel<NCO, VEC>), grid, dim3(256), 0, s, a);
}

// Pick the widest output slab per thread that still leaves enough threads to fill the chip:
// large volumes read the input once (NCO = Cout); tiny ones (level-12: 53k voxels) are latency-
// bound, so they trade L2 re-reads of the input for parallelism (slabs of 4 on blockIdx.z).
template <bool VEC>
static void launch_k1(const K1Args& a, int B, hipStream_t s) {
  constexpr int V = VEC ? 4 : 1;
  const int64_t threads = (int64_t)B * ceil_div(a.dhw, V);
  const int64_t want = 256 * 256 * 2;   // ~2 workgroups per CU
  static const int widths[5] = {24, 16, 12, 8, 4};
  int cover = 24;   // smallest slab covering Cout (Cout > 24 runs in slabs of 24)
  for (int w : widths)
    if (w >= a.Cout) cover = w;
  int nco = 4;
  if (threads >= want) {
    nco = cover;    // plenty of threads: read the input once
  } else {
    for (int w : widths)   // widest slab that still yields enough blocks
      if (w <= cover && threads * ceil_div(a.Cout, w) >= want) { nco = w; break; }
  }
  switch (nco) {
    case 24: launch_k1_nco<24, VEC>(a, B, s); break;
    case 16: launch_k1_nco<16, VEC>(a, B, s); break;
    case 12: launch_k1_nco<12, VEC>(a, B, s); break;
    case 8: launch_k1_nco<8, VEC>(a, B, s); break;
    default: launch_k1_nco<4, VEC>(a, B, s); break;
  }
}

}  // namespace ragmi

extern "C" int ragmi_conv3d_k1_fwd(const void* x, int64_t x_bstride, const void* weight, const void* scale,
                                   const void* shift, int relu, void* y, int64_t y_bstride, int y_ch0, int B, int Cin,
                                   int Cout, int64_t DHW, int dtype, void* stream) {
  using namespace ragmi;
  RAGMI_REQUIRE(x && weight && y, RAGMI_EINVAL, "conv3d_k1: null pointer");
  RAGMI_REQUIRE((scale == nullptr) == (shift == nullptr), RAGMI_EINVAL, "conv3d_k1: scale/shift must both be given or both NULL");
  RAGMI_REQUIRE(B > 0 && Cin > 0 && Cout > 0 && DHW > 0 && y_ch0 >= 0, RAGMI_EINVAL, "conv3d_k1: bad size");
  RAGMI_REQUIRE(dtype == RAGMI_F32, RAGMI_EUNSUPPORTED, "conv3d_k1: dtype %d not built", dtype);
  RAGMI_REQUIRE(B <= 65535 && Cout <= 4 * 65535, RAGMI_EUNSUPPORTED, "conv3d_k1: B or Cout too large");
  K1Args a{(const float*)x, x_bstride, (const float*)weight, (const float*)scale, (const float*)shift,
           (float*)y, y_bstride, y_ch0, Cin, Cout, 0, DHW, relu};
  // 16-B columns need alignment; small volumes use one voxel per thread for 4x the parallelism
  const bool vec = (DHW % 4 == 0) && (x_bstride % 4 == 0) && (y_bstride % 4 == 0) &&
                   ((reinterpret_cast<uintptr_t>(x) & 15) == 0) && ((reinterpret_cast<uintptr_t>(y) & 15) == 0) &&
                   (int64_t)B * DHW >= (1 << 19);
  hipStream_t s = static_cast<hipStream_t>(stream);
  if (vec) launch_k1<true>(a, B, s); else launch_k1<false>(a, B, s);
  return check_launch("conv3d_k1");
}

extern "C" int ragmi_add_fwd(const void* a, int64_t a_bstride, int a_ch0, const void* b, int64_t b_bstride, int b_ch0,
                             void* y, int64_t y_bstride, int y_ch0, int B, int C, int64_t DHW, int dtype, void* stream) {
  using namespace ragmi;
  RAGMI_REQUIRE(a && b && y, RAGMI_EINVAL, "add: null pointer");
  RAGMI_REQUIRE(B > 0 && C > 0 && DHW > 0 && a_ch0 >= 0 && b_ch0 >= 0 && y_ch0 >= 0, RAGMI_EINVAL, "add: bad size");
  RAGMI_REQUIRE(dtype == RAGMI_F32, RAGMI_EUNSUPPORTED, "add: dtype %d not built", dtype);
  RAGMI_REQUIRE(B <= 65535, RAGMI_EUNSUPPORTED, "add: B too large");
  const float* ap = (const float*)a + (int64_t)a_ch0 * DHW;
  const float* bp = (const float*)b + (int64_t)b_ch0 * DHW;
  float* yp = (float*)y + (int64_t)y_ch0 * DHW;
  const int64_t n = (int64_t)C * DHW;
  const bool vec = (n % 4 == 0) && (DHW % 4 == 0) && (a_bstride % 4 == 0) && (b_bstride % 4 == 0) && (y_bstride % 4 == 0) &&
                   (((reinterpret_cast<uintptr_t>(ap) | reinterpret_cast<uintptr_t>(bp) | reinterpret_cast<uintptr_t>(yp)) & 15) == 0);
  hipStream_t s = static_cast<hipStream_t>(stream);
  if (vec) {
    dim3 grid((unsigned)ceil_div(n / 4, 256), B);
    hipLaunchKernelGGL(add_kernel<true>, grid, dim3(256), 0, s, ap, a_bstride, bp, b_bstride, yp, y_bstride, n);
  } else {
    dim3 grid((unsigned)ceil_div(n, 256), B);
    hipLaunchKernelGGL(add_kernel<false>, grid, dim3(256), 0, s, ap, a_bstride, bp, b_bstride, yp, y_bstride, n);
  }
  return check_launch("add");
}
