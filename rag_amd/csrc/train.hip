// Training-step kernels of the Matching-Net path (BASELINE config 5; reference: approaches/rag.py:155-219 drives
// autograd through ConvBR_3d = Conv3d -> BatchNorm3d(train) -> ReLU, operations_3d.py:40-47, F.interpolate, the cost
// loop and Disp).  Data-gradients of the convolutions reuse the forward kernels with transposed / flipped weights;
// this file holds what has no forward twin: batch statistics, the BN+ReLU affine pass and its backward, the weight
// gradients, and the adjoints of resampling, cost volume and soft-argmin.  fp32 only (training runs in fp32).
#include <cstdlib>

#include "common.h"

namespace ragmi {

// ---------------------------------------------------------------------------------------------------------------
// Per-channel reductions are two-level and deterministic: every workgroup writes its partial pair to
// part[(c * nparts + b * gridDim.x + blockIdx.x) * 2 + {0,1}], a one-workgroup-per-channel kernel then sums the
// partials in double and does the per-channel arithmetic (no same-address float atomics, no zero-filled buffers,
// no host-side vector math).
__device__ __forceinline__ void block_pair_store(float s, float q, float* __restrict__ part, int64_t slot) {
  __shared__ float rs[256], rq[256];
  rs[threadIdx.x] = s; rq[threadIdx.x] = q;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if ((int)threadIdx.x < o) { rs[threadIdx.x] += rs[threadIdx.x + o]; rq[threadIdx.x] += rq[threadIdx.x + o]; }
    __syncthreads();
  }
  if (threadIdx.x == 0) { part[slot * 2] = rs[0]; part[slot * 2 + 1] = rq[0]; }
}
__device__ __forceinline__ void block_pair_sum(const float* __restrict__ part, int nparts, int c, double& s, double& q) {
  __shared__ double ds[256], dq[256];
  double ls = 0.0, lq = 0.0;
  for (int i = threadIdx.x; i < nparts; i += 256) { ls += part[((int64_t)c * nparts + i) * 2]; lq += part[((int64_t)c * nparts + i) * 2 + 1]; }
  ds[threadIdx.x] = ls; dq[threadIdx.x] = lq;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if ((int)threadIdx.x < o) { ds[threadIdx.x] += ds[threadIdx.x + o]; dq[threadIdx.x] += dq[threadIdx.x + o]; }
    __syncthreads();
  }
  s = ds[0]; q = dq[0];
}

// partial (sum, sum of squares) of x[B, C, DHW] (batch stride in elements; channel planes dense)
__global__ __launch_bounds__(256) void bn_stats_kernel(const float* __restrict__ x, int64_t x_bstride, int64_t dhw,
                                                       float* __restrict__ part, int nparts) {
  const int c = blockIdx.y, b = blockIdx.z;
  const float* p = x + b * x_bstride + (int64_t)c * dhw;
  const bool vec = (dhw & 3) == 0 && ((reinterpret_cast<uintptr_t>(p) & 15) == 0);
  float s = 0.f, q = 0.f;
  for (int64_t i = ((int64_t)blockIdx.x * 256 + threadIdx.x) * 4; i < dhw; i += (int64_t)gridDim.x * 1024) {
    if (vec) {
      const float4 v = *reinterpret_cast<const float4*>(p + i);
      s += (v.x + v.y) + (v.z + v.w);
      q += (v.x * v.x + v.y * v.y) + (v.z * v.z + v.w * v.w);
    } else {
      for (int k = 0; k < 4 && i + k < dhw; ++k) { const float v = p[i + k]; s += v; q += v * v; }
    }
  }
  block_pair_store(s, q, part, (int64_t)c * nparts + (int64_t)b * gridDim.x + blockIdx.x);
}

// train-mode BatchNorm bookkeeping of one channel per workgroup: batch mean / biased variance -> (mean, invstd, scale,
// shift) and the momentum update of the running statistics (unbiased variance), nn.BatchNorm semantics
struct BnFinalizeArgs {
  const float* part;
  const float* gamma;
  const float* beta;
  float* running_mean;
  float* running_var;
  long long* num_batches_tracked;
  float* mean;
  float* invstd;
  float* scale;
  float* shift;
  int nparts;
  double n;
  float eps, momentum;
};
__global__ __launch_bounds__(256) void bn_finalize_kernel(BnFinalizeArgs a) {
  const int c = blockIdx.x;
  double s, q;
  block_pair_sum(a.part, a.nparts, c, s, q);
  if (threadIdx.x != 0) return;
  const double mean = s / a.n;
  double var = q / a.n - mean * mean;
  var = var > 0.0 ? var : 0.0;
  const float invstd = (float)(1.0 / sqrt(var + (double)a.eps));
  const float sc = a.gamma[c] * invstd;
  a.mean[c] = (float)mean;
  a.invstd[c] = invstd;
  a.scale[c] = sc;
  a.shift[c] = a.beta[c] - (float)mean * sc;
  if (a.running_mean) {
    const float m = a.momentum;
    a.running_mean[c] = (1.f - m) * a.running_mean[c] + m * (float)mean;
    a.running_var[c] = (1.f - m) * a.running_var[c] + m * (float)(var * (a.n / (a.n > 1.0 ? a.n - 1.0 : 1.0)));
    if (c == 0 && a.num_batches_tracked) *a.num_batches_tracked += 1;
  }
}

// y[b, y_ch0 + c] = act(x[b, c] * scale[c] + shift[c]) (+ res[b, res_ch0 + c])
__global__ __launch_bounds__(256) void bn_act_kernel(const float* __restrict__ x, int64_t x_bstride, const float* __restrict__ scale,
                                                     const float* __restrict__ shift, int relu, const float* __restrict__ res,
                                                     int64_t res_bstride, int res_ch0, float* __restrict__ y, int64_t y_bstride,
                                                     int y_ch0, int64_t dhw) {
  const int c = blockIdx.y, b = blockIdx.z;
  const int64_t i0 = ((int64_t)blockIdx.x * 256 + threadIdx.x) * 4;
  if (i0 >= dhw) return;
  const float* px = x + b * x_bstride + (int64_t)c * dhw;
  const float* pr = res ? res + b * res_bstride + (int64_t)(res_ch0 + c) * dhw : nullptr;
  float* py = y + b * y_bstride + (int64_t)(y_ch0 + c) * dhw;
  const float sc = scale[c], sh = shift[c];
  auto one = [&](float xv, float rv) {
    float v = fmaf(xv, sc, sh);
    if (relu) v = fmaxf(v, 0.f);
    return v + rv;
  };
  const uintptr_t al = reinterpret_cast<uintptr_t>(px) | reinterpret_cast<uintptr_t>(py) | reinterpret_cast<uintptr_t>(pr);
  if ((dhw & 3) == 0 && (al & 15) == 0) {
    const float4 xv = *reinterpret_cast<const float4*>(px + i0);
    const float4 rv = pr ? *reinterpret_cast<const float4*>(pr + i0) : make_float4(0.f, 0.f, 0.f, 0.f);
    *reinterpret_cast<float4*>(py + i0) = make_float4(one(xv.x, rv.x), one(xv.y, rv.y), one(xv.z, rv.z), one(xv.w, rv.w));
  } else {
    for (int k = 0; k < 4 && i0 + k < dhw; ++k) py[i0 + k] = one(px[i0 + k], pr ? pr[i0 + k] : 0.f);
  }
}

// g = dy * (x*scale+shift > 0 if relu);  partial (sum g, sum g*x) per workgroup
__global__ __launch_bounds__(256) void bn_act_bwd_reduce_kernel(const float* __restrict__ dy, int64_t dy_bstride, int dy_ch0,
                                                                const float* __restrict__ x, int64_t x_bstride,
                                                                const float* __restrict__ scale, const float* __restrict__ shift,
                                                                int relu, int64_t dhw, float* __restrict__ part, int nparts) {
  const int c = blockIdx.y, b = blockIdx.z;
  const float* pd = dy + b * dy_bstride + (int64_t)(dy_ch0 + c) * dhw;
  const float* px = x + b * x_bstride + (int64_t)c * dhw;
  const float sc = scale[c], sh = shift[c];
  const bool vec = (dhw & 3) == 0 && (((reinterpret_cast<uintptr_t>(pd) | reinterpret_cast<uintptr_t>(px)) & 15) == 0);
  float s = 0.f, q = 0.f;
  auto one = [&](float g, float xv) {
    if (relu && fmaf(xv, sc, sh) <= 0.f) g = 0.f;
    s += g;
    q = fmaf(g, xv, q);
  };
  for (int64_t i = ((int64_t)blockIdx.x * 256 + threadIdx.x) * 4; i < dhw; i += (int64_t)gridDim.x * 1024) {
    if (vec) {
      const float4 g4 = *reinterpret_cast<const float4*>(pd + i), x4 = *reinterpret_cast<const float4*>(px + i);
      one(g4.x, x4.x); one(g4.y, x4.y); one(g4.z, x4.z); one(g4.w, x4.w);
    } else {
      for (int k = 0; k < 4 && i + k < dhw; ++k) one(pd[i + k], px[i + k]);
    }
  }
  block_pair_store(s, q, part, (int64_t)c * nparts + (int64_t)b * gridDim.x + blockIdx.x);
}

// per-channel coefficients of the ReLU+BN adjoint dx = g*c1 + x*c2 + c3, plus dgamma / dbeta
struct BnCoeffArgs {
  const float* part;
  const float* mean;
  const float* invstd;
  const float* scale;
  float* c1;
  float* c2;
  float* c3;
  float* dgamma;
  float* dbeta;
  int nparts, training, accumulate;
  double n;
};
__global__ __launch_bounds__(256) void bn_bwd_coeffs_kernel(BnCoeffArgs a) {
  const int c = blockIdx.x;
  double sg, sgx;
  block_pair_sum(a.part, a.nparts, c, sg, sgx);
  if (threadIdx.x != 0) return;
  const double mean = a.mean[c], invstd = a.invstd[c], sc = a.scale[c];
  const double sgxh = invstd * (sgx - mean * sg);          // sum of g * xhat
  if (a.dgamma) a.dgamma[c] = (a.accumulate ? a.dgamma[c] : 0.f) + (float)sgxh;
  if (a.dbeta) a.dbeta[c] = (a.accumulate ? a.dbeta[c] : 0.f) + (float)sg;
  a.c1[c] = (float)sc;
  if (a.training) {
    // dx = a (g - mean(g) - xhat mean(g xhat)), a = gamma * invstd: linear in g and x per channel
    const double mg = sg / a.n, mgxh = sgxh / a.n;
    a.c2[c] = (float)(-sc * invstd * mgxh);
    a.c3[c] = (float)(sc * (invstd * mean * mgxh - mg));
  } else {
    a.c2[c] = 0.f;
    a.c3[c] = 0.f;
  }
}

// ---------------------------------------------------------------------------------------------------------------
// Fused second halves: every workgroup re-derives its channel's coefficients from the (<= 256) partials — a couple of
// microseconds of redundant work per workgroup — instead of a separate one-workgroup-per-channel launch in between:
// train-mode BatchNorm + ReLU forward = bn_stats + bn_finalize_act, its adjoint = bn_act_bwd_reduce + bn_coeffs_apply.
struct BnFinActArgs {
  BnFinalizeArgs f;
  const float* x;
  float* y;
  int64_t x_bstride, y_bstride, dhw;
  int y_ch0, relu;
  const float* res;        // optional: y = act(..) + res[b, res_ch0 + c] (a cell's running sum of branches)
  int64_t res_bstride;
  int res_ch0;
};
__global__ __launch_bounds__(256) void bn_finalize_act_kernel(BnFinActArgs a) {
  const int c = blockIdx.y, b = blockIdx.z;
  double s, q;
  block_pair_sum(a.f.part, a.f.nparts, c, s, q);
  const double mean = s / a.f.n;
  double var = q / a.f.n - mean * mean;
  var = var > 0.0 ? var : 0.0;
  const float invstd = (float)(1.0 / sqrt(var + (double)a.f.eps));
  const float sc = a.f.gamma[c] * invstd, sh = a.f.beta[c] - (float)mean * sc;
  if (blockIdx.x == 0 && b == 0 && threadIdx.x == 0) {
    a.f.mean[c] = (float)mean;
    a.f.invstd[c] = invstd;
    a.f.scale[c] = sc;
    a.f.shift[c] = sh;
    if (a.f.running_mean) {
      const float m = a.f.momentum;
      a.f.running_mean[c] = (1.f - m) * a.f.running_mean[c] + m * (float)mean;
      a.f.running_var[c] = (1.f - m) * a.f.running_var[c] + m * (float)(var * (a.f.n / (a.f.n > 1.0 ? a.f.n - 1.0 : 1.0)));
      if (c == 0 && a.f.num_batches_tracked) *a.f.num_batches_tracked += 1;
    }
  }
  const int64_t i0 = ((int64_t)blockIdx.x * 256 + threadIdx.x) * 4;
  if (i0 >= a.dhw) return;
  const float* px = a.x + b * a.x_bstride + (int64_t)c * a.dhw;
  float* py = a.y + b * a.y_bstride + (int64_t)(a.y_ch0 + c) * a.dhw;
  const float* pr = a.res ? a.res + b * a.res_bstride + (int64_t)(a.res_ch0 + c) * a.dhw : nullptr;
  auto one = [&](float xv) {
    const float v = fmaf(xv, sc, sh);
    return a.relu ? fmaxf(v, 0.f) : v;
  };
  const uintptr_t al = reinterpret_cast<uintptr_t>(px) | reinterpret_cast<uintptr_t>(py) | reinterpret_cast<uintptr_t>(pr);
  if ((a.dhw & 3) == 0 && (al & 15) == 0) {
    const float4 xv = *reinterpret_cast<const float4*>(px + i0);
    float4 o = make_float4(one(xv.x), one(xv.y), one(xv.z), one(xv.w));
    if (pr) {
      const float4 rv = *reinterpret_cast<const float4*>(pr + i0);
      o.x += rv.x; o.y += rv.y; o.z += rv.z; o.w += rv.w;
    }
    *reinterpret_cast<float4*>(py + i0) = o;
  } else {
    for (int k = 0; k < 4 && i0 + k < a.dhw; ++k) py[i0 + k] = one(px[i0 + k]) + (pr ? pr[i0 + k] : 0.f);
  }
}

struct BnCoeffApplyArgs {
  const float* part;
  const float* mean;
  const float* invstd;
  const float* scale;
  const float* shift;
  float* dgamma;
  float* dbeta;
  const float* dy;
  const float* x;
  float* dx;
  int64_t dy_bstride, x_bstride, dx_bstride, dhw;
  int nparts, training, accumulate, relu, dy_ch0;
  double n;
};
__global__ __launch_bounds__(256) void bn_coeffs_apply_kernel(BnCoeffApplyArgs a) {
  const int c = blockIdx.y, b = blockIdx.z;
  double sg, sgx;
  block_pair_sum(a.part, a.nparts, c, sg, sgx);
  const double mean = a.mean[c], invstd = a.invstd[c], scd = a.scale[c];
  const double sgxh = invstd * (sgx - mean * sg);          // sum of g * xhat
  if (blockIdx.x == 0 && b == 0 && threadIdx.x == 0) {
    if (a.dgamma) a.dgamma[c] = (a.accumulate ? a.dgamma[c] : 0.f) + (float)sgxh;
    if (a.dbeta) a.dbeta[c] = (a.accumulate ? a.dbeta[c] : 0.f) + (float)sg;
  }
  float k1 = (float)scd, k2 = 0.f, k3 = 0.f;
  if (a.training) {
    const double mg = sg / a.n, mgxh = sgxh / a.n;
    k2 = (float)(-scd * invstd * mgxh);
    k3 = (float)(scd * (invstd * mean * mgxh - mg));
  }
  const int64_t i0 = ((int64_t)blockIdx.x * 256 + threadIdx.x) * 4;
  if (i0 >= a.dhw) return;
  const float* px = a.x + b * a.x_bstride + (int64_t)c * a.dhw;
  const float* pg = a.dy + b * a.dy_bstride + (int64_t)(a.dy_ch0 + c) * a.dhw;
  float* po = a.dx + b * a.dx_bstride + (int64_t)c * a.dhw;
  const float sc = a.scale[c], sh = a.shift[c];
  auto one = [&](float g, float xv) {
    if (a.relu && fmaf(xv, sc, sh) <= 0.f) g = 0.f;
    return fmaf(g, k1, fmaf(xv, k2, k3));
  };
  const uintptr_t al = reinterpret_cast<uintptr_t>(px) | reinterpret_cast<uintptr_t>(pg) | reinterpret_cast<uintptr_t>(po);
  if ((a.dhw & 3) == 0 && (al & 15) == 0) {
    const float4 g4 = *reinterpret_cast<const float4*>(pg + i0), x4 = *reinterpret_cast<const float4*>(px + i0);
    *reinterpret_cast<float4*>(po + i0) = make_float4(one(g4.x, x4.x), one(g4.y, x4.y), one(g4.z, x4.z), one(g4.w, x4.w));
  } else {
    for (int k = 0; k < 4 && i0 + k < a.dhw; ++k) po[i0 + k] = one(pg[i0 + k], px[i0 + k]);
  }
}

// dx[b, c] = g * c1[c] + x * c2[c] + c3[c]   (train-mode BN backward is linear in g and x per channel; eval: c2 = c3 = 0)
__global__ __launch_bounds__(256) void bn_act_bwd_apply_kernel(const float* __restrict__ dy, int64_t dy_bstride, int dy_ch0,
                                                               const float* __restrict__ x, int64_t x_bstride,
                                                               const float* __restrict__ scale, const float* __restrict__ shift,
                                                               int relu, const float* __restrict__ c1, const float* __restrict__ c2,
                                                               const float* __restrict__ c3, float* __restrict__ dx,
                                                               int64_t dx_bstride, int64_t dhw) {
  const int c = blockIdx.y, b = blockIdx.z;
  const int64_t i0 = ((int64_t)blockIdx.x * 256 + threadIdx.x) * 4;
  if (i0 >= dhw) return;
  const float* px = x + b * x_bstride + (int64_t)c * dhw;
  const float* pg = dy + b * dy_bstride + (int64_t)(dy_ch0 + c) * dhw;
  float* po = dx + b * dx_bstride + (int64_t)c * dhw;
  const float sc = scale[c], sh = shift[c], k1 = c1[c], k2 = c2[c], k3 = c3[c];
  auto one = [&](float g, float xv) {
    if (relu && fmaf(xv, sc, sh) <= 0.f) g = 0.f;
    return fmaf(g, k1, fmaf(xv, k2, k3));
  };
  const uintptr_t al = reinterpret_cast<uintptr_t>(px) | reinterpret_cast<uintptr_t>(pg) | reinterpret_cast<uintptr_t>(po);
  if ((dhw & 3) == 0 && (al & 15) == 0) {
    const float4 g4 = *reinterpret_cast<const float4*>(pg + i0), x4 = *reinterpret_cast<const float4*>(px + i0);
    *reinterpret_cast<float4*>(po + i0) = make_float4(one(g4.x, x4.x), one(g4.y, x4.y), one(g4.z, x4.z), one(g4.w, x4.w));
  } else {
    for (int k = 0; k < 4 && i0 + k < dhw; ++k) po[i0 + k] = one(pg[i0 + k], px[i0 + k]);
  }
}

// ---------------------------------------------------------------------------------------------------------------
// Weight gradient of the 3x3x3 convolution on the MFMA: dw[co][ci][tap] += sum_v g[co][v] * x[ci][v + tap].
// The contraction index is the voxel.  v_mfma_f32_4x4x1 does 16 independent 4x4 outer products per instruction
// (D[lane 4b+n][reg m] += A[lane 4b+m] * B[lane 4b+n]); block b takes voxel v_b, its A lanes hold g[co0+m][v_b], its
// B lanes x[ci0+n][v_b + tap], so one instruction adds 16 voxels to a 4(co) x 4(ci) tile of one tap (256 MACs, the
// forward kernel's rate).  A workgroup is 3 waves (wave = dz plane of the taps) over a 2 x 8 x 32 voxel tile with the
// x halo of 4 input channels and the g tile of CG*4 output channels in LDS; each wave keeps 9 (dy,dx) x CG
// accumulator tiles.  Workgroups are persistent (grid-stride over tiles), so the 16 per-block partial sums are
// reduced across lanes and flushed with float atomics once per workgroup, not per tile.
// LDS plane strides are = 16 mod 32 banks: the 4 channel lanes of a block land 2 per bank, the minimum for 64 lanes.
constexpr int WG_TZ = 2, WG_TY = 8, WG_TX = 32, WG_XS = WG_TX + 2, WG_PS = (WG_TZ + 2) * (WG_TY + 2) * WG_XS;   // 1360
constexpr int WG_NV = WG_TZ * WG_TY * WG_TX, WG_GP = WG_NV + 16;                                                 // 512, 528
static_assert(WG_PS % 32 == 16 && WG_GP % 32 == 16, "LDS plane strides must be 16 mod 32");
struct WgradArgs {
  const float* x;
  const float* g;
  float* part;           // [gridDim.x][CoutP][CinP][27] partial sums, CoutP / CinP = channel counts padded to the grid
  int64_t x_bstride, g_bstride;
  int g_ch0, Cin, Cout, D, H, W, tiles_x, tiles_y, tiles_z, ntiles;
};
template <int CG, bool VEC>
__global__ __launch_bounds__(192) void conv3d_k3_wgrad_kernel(WgradArgs a) {
  constexpr int XROWS = 4 * (WG_TZ + 2) * (WG_TY + 2);            // 160 halo rows of 34
  constexpr int GROWS = CG * 4 * WG_TZ * WG_TY;                    // rows of 32 of the g tile
  constexpr int NXQ = (XROWS * 8 + 191) / 192, NXH = (XROWS * 2 + 191) / 192, NGQ = (GROWS * 8 + 191) / 192;
  static_assert(NXQ + NXH + NGQ <= 32, "validity mask is one 32-bit word");
  __shared__ float xs[4 * WG_PS];
  __shared__ __attribute__((aligned(16))) float gs[CG * 4 * WG_GP];
  const int tid = threadIdx.x, dz = tid >> 6, lane = tid & 63, blk = lane >> 2, n = lane & 3;
  const int ci0 = blockIdx.y * 4, co0 = blockIdx.z * (CG * 4);
  const int HW = a.H * a.W;
  const int64_t DHW = (int64_t)HW * a.D;
  f32x4 acc[9][CG];
#pragma unroll
  for (int t = 0; t < 9; ++t)
#pragma unroll
    for (int c = 0; c < CG; ++c) acc[t][c] = f32x4{0.f, 0.f, 0.f, 0.f};

  // software pipeline: the next tile's x halo / g tile travel HBM -> registers while the MFMAs of this tile run.
  // The loads are unconditional (addresses clamped into the volume, like the forward kernel's staging) so that they
  // stay straight-line code ahead of the MFMA loop; what lies outside the volume is zeroed at commit through `valid`.
  float4 xq[NXQ], gq[NGQ];
  float xh[NXH];
  unsigned valid = 0;
  auto row4 = [&](const float* row, int gx) -> float4 {
    if constexpr (VEC) return *reinterpret_cast<const float4*>(row + gx);
    else return make_float4(row[gx], row[min(gx + 1, a.W - 1)], row[min(gx + 2, a.W - 1)], row[min(gx + 3, a.W - 1)]);
  };
  auto prefetch = [&](int tile) {
    int t = tile;
    const int x0 = (t % a.tiles_x) * WG_TX; t /= a.tiles_x;
    const int y0 = (t % a.tiles_y) * WG_TY; t /= a.tiles_y;
    const int z0 = (t % a.tiles_z) * WG_TZ;
    const int b = t / a.tiles_z;
    const float* xb = a.x + b * a.x_bstride;
    const float* gb = a.g + b * a.g_bstride + (int64_t)a.g_ch0 * DHW;
    valid = 0;
#pragma unroll
    for (int p = 0; p < NXQ; ++p) {
      const int f = p * 192 + tid, row = f >> 3, q = f & 7;
      const int c = row / ((WG_TZ + 2) * (WG_TY + 2)), zz = (row / (WG_TY + 2)) % (WG_TZ + 2), yy = row % (WG_TY + 2);
      const int gz = z0 - 1 + zz, gy = y0 - 1 + yy, gx = x0 + 4 * q;
      const bool ok = row < XROWS && (unsigned)gz < (unsigned)a.D && (unsigned)gy < (unsigned)a.H && gx < a.W && ci0 + c < a.Cin;
      valid |= (ok ? 1u : 0u) << p;
      const int cc = min(ci0 + c, a.Cin - 1), cz = min(max(gz, 0), a.D - 1), cy = min(max(gy, 0), a.H - 1);
      xq[p] = row4(xb + cc * DHW + (unsigned)(cz * HW + cy * a.W), VEC ? min(gx, a.W - 4) : min(gx, a.W - 1));
    }
#pragma unroll
    for (int p = 0; p < NXH; ++p) {
      const int f = p * 192 + tid, row = f >> 1, side = f & 1;
      const int c = row / ((WG_TZ + 2) * (WG_TY + 2)), zz = (row / (WG_TY + 2)) % (WG_TZ + 2), yy = row % (WG_TY + 2);
      const int gz = z0 - 1 + zz, gy = y0 - 1 + yy, gx = side ? x0 + WG_TX : x0 - 1;
      const bool ok = row < XROWS && (unsigned)gz < (unsigned)a.D && (unsigned)gy < (unsigned)a.H && (unsigned)gx < (unsigned)a.W &&
                      ci0 + c < a.Cin;
      valid |= (ok ? 1u : 0u) << (NXQ + p);
      const int cc = min(ci0 + c, a.Cin - 1), cz = min(max(gz, 0), a.D - 1), cy = min(max(gy, 0), a.H - 1);
      xh[p] = xb[cc * DHW + (unsigned)(cz * HW + cy * a.W + min(max(gx, 0), a.W - 1))];
    }
#pragma unroll
    for (int p = 0; p < NGQ; ++p) {
      const int f = p * 192 + tid, row = f >> 3, q = f & 7;
      const int c = row / (WG_TZ * WG_TY), zz = (row / WG_TY) % WG_TZ, yy = row % WG_TY;
      const int gz = z0 + zz, gy = y0 + yy, gx = x0 + 4 * q;
      const bool ok = row < GROWS && gz < a.D && gy < a.H && gx < a.W && co0 + c < a.Cout;
      valid |= (ok ? 1u : 0u) << (NXQ + NXH + p);
      const int cc = min(co0 + c, a.Cout - 1), cz = min(gz, a.D - 1), cy = min(gy, a.H - 1);
      gq[p] = row4(gb + cc * DHW + (unsigned)(cz * HW + cy * a.W), VEC ? min(gx, a.W - 4) : min(gx, a.W - 1));
    }
  };
  auto commit = [&](int x0) {                         // x0: the tile's first x (scalar path: per-element validity along x)
    auto masked = [&](float4 v, bool ok, int gx) -> float4 {
      if (!ok) return make_float4(0.f, 0.f, 0.f, 0.f);
      if constexpr (!VEC) {
        if (gx + 1 >= a.W) v.y = 0.f;
        if (gx + 2 >= a.W) v.z = 0.f;
        if (gx + 3 >= a.W) v.w = 0.f;
      }
      return v;
    };
#pragma unroll
    for (int p = 0; p < NXQ; ++p) {
      const int f = p * 192 + tid, row = f >> 3, q = f & 7;
      if (row < XROWS) {
        const int c = row / ((WG_TZ + 2) * (WG_TY + 2)), r = row % ((WG_TZ + 2) * (WG_TY + 2));
        const float4 v = masked(xq[p], (valid >> p) & 1u, x0 + 4 * q);
        float* d = xs + c * WG_PS + r * WG_XS + 1 + 4 * q;
        d[0] = v.x; d[1] = v.y; d[2] = v.z; d[3] = v.w;
      }
    }
#pragma unroll
    for (int p = 0; p < NXH; ++p) {
      const int f = p * 192 + tid, row = f >> 1, side = f & 1;
      if (row < XROWS) {
        const int c = row / ((WG_TZ + 2) * (WG_TY + 2)), r = row % ((WG_TZ + 2) * (WG_TY + 2));
        xs[c * WG_PS + r * WG_XS + (side ? WG_TX + 1 : 0)] = ((valid >> (NXQ + p)) & 1u) ? xh[p] : 0.f;
      }
    }
#pragma unroll
    for (int p = 0; p < NGQ; ++p) {
      const int f = p * 192 + tid, row = f >> 3, q = f & 7;
      if (row < GROWS) {
        const int c = row / (WG_TZ * WG_TY), r = row % (WG_TZ * WG_TY);
        *reinterpret_cast<float4*>(gs + c * WG_GP + r * WG_TX + 4 * q) = masked(gq[p], (valid >> (NXQ + NXH + p)) & 1u, x0 + 4 * q);
      }
    }
  };

  int tile = blockIdx.x;
  if (tile < a.ntiles) prefetch(tile);
  for (; tile < a.ntiles; tile += gridDim.x) {
    __syncthreads();                                   // the previous tile's LDS reads are done
    commit((tile % a.tiles_x) * WG_TX);
    __syncthreads();
    // unconditional (past the end it re-reads the last tile; never committed) and pinned: a branch around the prefetch makes the
    // compiler drain it before the first LDS read, and without the barrier the scheduler sinks it below the MFMA loop
    prefetch(min(tile + (int)gridDim.x, a.ntiles - 1));
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll 2
    for (int s = 0; s < WG_NV / 16; ++s) {
      const int v = s * 16 + blk;                      // this block's voxel of the step: 16 consecutive x
      const int xx = v % WG_TX, yy = (v / WG_TX) % WG_TY, zz = v / (WG_TX * WG_TY);
      float av[CG];
#pragma unroll
      for (int c = 0; c < CG; ++c) av[c] = gs[(c * 4 + n) * WG_GP + v];
      const float* xr = xs + n * WG_PS + ((zz + dz) * (WG_TY + 2) + yy) * WG_XS + xx;
#pragma unroll
      for (int t9 = 0; t9 < 9; ++t9) {
        const float bv = xr[(t9 / 3) * WG_XS + (t9 % 3)];
#pragma unroll
        for (int c = 0; c < CG; ++c) acc[t9][c] = __builtin_amdgcn_mfma_f32_4x4x1f32(av[c], bv, acc[t9][c], 0, 0, 0);
      }
    }
  }
  // sum the 16 blocks (lanes 4b+n, fixed n); lanes 0..3 store this workgroup's partial: reg m of lane n is
  // dw[co0+4c+m][ci0+n][dz*9+t9].  Every workgroup writes its whole slice (zeros included): no initialisation needed.
  const int CinP = gridDim.y * 4, CoutP = gridDim.z * CG * 4;
  float* part = a.part + (int64_t)blockIdx.x * CoutP * CinP * 27;
#pragma unroll
  for (int t9 = 0; t9 < 9; ++t9)
#pragma unroll
    for (int c = 0; c < CG; ++c)
#pragma unroll
      for (int m = 0; m < 4; ++m) {
        float v = acc[t9][c][m];
        v += __shfl_xor(v, 4);
        v += __shfl_xor(v, 8);
        v += __shfl_xor(v, 16);
        v += __shfl_xor(v, 32);
        if (blk == 0) part[((int64_t)(co0 + c * 4 + m) * CinP + ci0 + n) * 27 + dz * 9 + t9] = v;
      }
}

// dw[co][ci][tap] (+)= sum over the workgroups' partials: one wave per output element, lanes stride over the partials.
// The Cout rows may be split over up to 8 destination tensors (stacked sibling convolutions write each unit's weight
// gradient in place); planar: the destination is a 2-D [.,.,3,3] weight, only the dz = 1 plane (taps 9..17) is kept.
struct WgradDst {
  float* p[8];
  int n, rows_per_dst, accumulate, planar;
};
__global__ __launch_bounds__(256) void conv3d_k3_wgrad_reduce_kernel(const float* __restrict__ part, WgradDst dst, int nparts, int Cin,
                                                                     int Cout, int CinP, int CoutP) {
  const int i = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (i >= Cout * Cin * 27) return;
  const int tap = i % 27, ci = (i / 27) % Cin, co = i / (27 * Cin);
  if (dst.planar && (tap < 9 || tap >= 18)) return;
  const int64_t stride = (int64_t)CoutP * CinP * 27, off = ((int64_t)co * CinP + ci) * 27 + tap;
  float s = 0.f;
  for (int p = lane; p < nparts; p += 64) s += part[p * stride + off];
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
  if (lane == 0) {
    const int k = co / dst.rows_per_dst, r = co % dst.rows_per_dst;
    float* d = dst.p[k] + (dst.planar ? ((int64_t)r * Cin + ci) * 9 + (tap - 9) : ((int64_t)r * Cin + ci) * 27 + tap);
    *d = dst.accumulate ? *d + s : s;
  }
}

// dw[co][ci] += sum_v g[co][v] * x[ci][v]   (1x1x1 conv).  A workgroup owns a 4 (co) x 12 (ci) block of dw and a slab of
// voxels: each thread keeps the 48 partial sums of its voxels in registers (x and g are each read once per block row /
// column), the workgroup reduces them (wave shuffles, then LDS across the 4 waves) and flushes 48 float atomics.
constexpr int K1W_CO = 4, K1W_CI = 12;
__global__ __launch_bounds__(256) void conv3d_k1_wgrad_kernel(const float* __restrict__ x, int64_t x_bstride, const float* __restrict__ g,
                                                              int64_t g_bstride, int g_ch0, float* __restrict__ dw, int Cin, int Cout,
                                                              int64_t dhw, int nci_blocks) {
  const int co0 = (blockIdx.y / nci_blocks) * K1W_CO, ci0 = (blockIdx.y % nci_blocks) * K1W_CI, b = blockIdx.z;
  const float* pg = g + b * g_bstride + (int64_t)(g_ch0 + co0) * dhw;
  const float* px = x + b * x_bstride + (int64_t)ci0 * dhw;
  const int nco = min(K1W_CO, Cout - co0), nci = min(K1W_CI, Cin - ci0);
  float acc[K1W_CO][K1W_CI];
#pragma unroll
  for (int i = 0; i < K1W_CO; ++i)
#pragma unroll
    for (int j = 0; j < K1W_CI; ++j) acc[i][j] = 0.f;
  for (int64_t v = (int64_t)blockIdx.x * 256 + threadIdx.x; v < dhw; v += (int64_t)gridDim.x * 256) {
    float gv[K1W_CO], xv[K1W_CI];
#pragma unroll
    for (int i = 0; i < K1W_CO; ++i) gv[i] = i < nco ? pg[i * dhw + v] : 0.f;
#pragma unroll
    for (int j = 0; j < K1W_CI; ++j) xv[j] = j < nci ? px[j * dhw + v] : 0.f;
#pragma unroll
    for (int i = 0; i < K1W_CO; ++i)
#pragma unroll
      for (int j = 0; j < K1W_CI; ++j) acc[i][j] = fmaf(gv[i], xv[j], acc[i][j]);
  }
  __shared__ float red[4][K1W_CO * K1W_CI];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
#pragma unroll
  for (int i = 0; i < K1W_CO; ++i)
#pragma unroll
    for (int j = 0; j < K1W_CI; ++j) {
      float v = acc[i][j];
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
      if (lane == 0) red[wave][i * K1W_CI + j] = v;
    }
  __syncthreads();
  if (threadIdx.x < K1W_CO * K1W_CI) {
    const int i = threadIdx.x / K1W_CI, j = threadIdx.x % K1W_CI;
    if (i < nco && j < nci)
      atomicAdd(dw + (int64_t)(co0 + i) * Cin + ci0 + j, (red[0][threadIdx.x] + red[1][threadIdx.x]) + (red[2][threadIdx.x] + red[3][threadIdx.x]));
  }
}

// ---------------------------------------------------------------------------------------------------------------
// adjoint of the trilinear resample as a GATHER: one thread per input voxel sums, over the output voxels whose taps
// touch it, dy * (product of the per-axis tap weights) — no atomics, no zero-filled destination, deterministic.
// Per axis the touching outputs are a contiguous range found from the inverse of ATen's source-index map and then
// trimmed with the exact forward rule (lin_index), so the weights are the forward's bit for bit.
struct TriBwdArgs {
  const float* dy;
  float* dx;
  int C, Di, Hi, Wi, Do, Ho, Wo;
  float sd, sh, sw;
  int align, cpt;
};
__device__ __forceinline__ float tap_weight(int o, int i, int in_size, int out_size, float scale, int align) {
  const LinIdx l = lin_index(o, in_size, out_size, scale, align);
  return (l.i0 == i ? l.w0 : 0.f) + (l.i1 == i ? l.w1 : 0.f);     // i0 == i1 at the clamped end: both taps land here
}
__device__ __forceinline__ void tap_range(int i, int in_size, int out_size, float scale, int align, int& lo, int& hi) {
  if (in_size == out_size) { lo = hi = i; return; }
  float a = 0.f, b = (float)(out_size - 1);
  if (scale > 0.f) {
    if (align) { a = ((float)i - 1.f) / scale; b = ((float)i + 1.f) / scale; }
    else { a = ((float)i - 0.5f) / scale - 0.5f; b = ((float)i + 1.5f) / scale - 0.5f; }
  }
  lo = max(0, (int)floorf(a) - 1);
  hi = min(out_size - 1, (int)ceilf(b) + 1);
  while (lo <= hi && tap_weight(lo, i, in_size, out_size, scale, align) == 0.f) ++lo;
  while (hi >= lo && tap_weight(hi, i, in_size, out_size, scale, align) == 0.f) --hi;
}
__global__ __launch_bounds__(256) void trilinear_bwd_kernel(TriBwdArgs a) {
  const int64_t ivol = (int64_t)a.Di * a.Hi * a.Wi, ovol = (int64_t)a.Do * a.Ho * a.Wo;
  const int64_t p = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (p >= ivol) return;
  const int b = blockIdx.y;
  const int ix = (int)(p % a.Wi);
  const int64_t t = p / a.Wi;
  const int iy = (int)(t % a.Hi), iz = (int)(t / a.Hi);
  int z0, z1, y0, y1, x0, x1;
  tap_range(iz, a.Di, a.Do, a.sd, a.align, z0, z1);
  tap_range(iy, a.Hi, a.Ho, a.sh, a.align, y0, y1);
  tap_range(ix, a.Wi, a.Wo, a.sw, a.align, x0, x1);
  // channels per thread: the tap ranges / weights depend on the voxel only, so a thread reuses them over `cpt` channels.  With at
  // most four taps per axis (every x2 / x0.5 / x3 resample of the networks) the weights are evaluated ONCE per voxel into registers
  // (round 4: they were re-derived — a lin_index each — inside the tap loops of every channel: most of the kernel's instructions);
  // same values, same order of products and sums.
  constexpr int MT = 4;
  const int nz = z1 - z0 + 1, ny = y1 - y0 + 1, nx = x1 - x0 + 1;
  if (nz <= MT && ny <= MT && nx <= MT) {
    float wz[MT], wy[MT], wx[MT];
#pragma unroll
    for (int k = 0; k < MT; ++k) {
      wz[k] = k < nz ? tap_weight(z0 + k, iz, a.Di, a.Do, a.sd, a.align) : 0.f;
      wy[k] = k < ny ? tap_weight(y0 + k, iy, a.Hi, a.Ho, a.sh, a.align) : 0.f;
      wx[k] = k < nx ? tap_weight(x0 + k, ix, a.Wi, a.Wo, a.sw, a.align) : 0.f;
    }
    for (int c = blockIdx.z * a.cpt; c < min(a.C, (int)(blockIdx.z + 1) * a.cpt); ++c) {
      const float* pc = a.dy + ((int64_t)b * a.C + c) * ovol;
      float acc = 0.f;
#pragma unroll
      for (int kz = 0; kz < MT; ++kz) {
        if (kz >= nz) break;
#pragma unroll
        for (int ky = 0; ky < MT; ++ky) {
          if (ky >= ny) break;
          const float wzy = wz[kz] * wy[ky];
          const float* pr = pc + ((int64_t)(z0 + kz) * a.Ho + (y0 + ky)) * a.Wo + x0;
#pragma unroll
          for (int kx = 0; kx < MT; ++kx) {
            if (kx >= nx) break;
            acc = fmaf(pr[kx], wzy * wx[kx], acc);
          }
        }
      }
      a.dx[((int64_t)b * a.C + c) * ivol + p] = acc;
    }
    return;
  }
  for (int c = blockIdx.z * a.cpt; c < min(a.C, (int)(blockIdx.z + 1) * a.cpt); ++c) {
    const float* pc = a.dy + ((int64_t)b * a.C + c) * ovol;
    float acc = 0.f;
    for (int oz = z0; oz <= z1; ++oz) {
      const float wz = tap_weight(oz, iz, a.Di, a.Do, a.sd, a.align);
      for (int oy = y0; oy <= y1; ++oy) {
        const float wzy = wz * tap_weight(oy, iy, a.Hi, a.Ho, a.sh, a.align);
        const float* pr = pc + ((int64_t)oz * a.Ho + oy) * a.Wo;
        for (int ox = x0; ox <= x1; ++ox) acc = fmaf(pr[ox], wzy * tap_weight(ox, ix, a.Wi, a.Wo, a.sw, a.align), acc);
      }
    }
    a.dx[((int64_t)b * a.C + c) * ivol + p] = acc;
  }
}

// adjoint of the cost volume: dL[c,y,x] = sum_{i<=x} dcost[c,i,y,x];  dR[c,y,x] = sum_{i, x+i<w} dcost[C+c,i,y,x+i]
__global__ __launch_bounds__(256) void costvol_bwd_kernel(const float* __restrict__ dcost, float* __restrict__ dL, float* __restrict__ dR,
                                                          int C, int d, int h, int w) {
  const int64_t hw = (int64_t)h * w;
  const int64_t p = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (p >= hw) return;
  const int c = blockIdx.y, b = blockIdx.z;
  const int x = (int)(p % w);
  const float* pl = dcost + (((int64_t)b * 2 * C + c) * d) * hw + p;
  const float* pr = dcost + (((int64_t)b * 2 * C + C + c) * d) * hw + p;
  float sl = 0.f, sr = 0.f;
  for (int i = 0; i < d; ++i) {
    if (i <= x) sl += pl[(int64_t)i * hw];
    if (x + i < w) sr += pr[(int64_t)i * hw + i];
  }
  dL[((int64_t)b * C + c) * hw + p] = sl;
  dR[((int64_t)b * C + c) * hw + p] = sr;
}

// adjoint of the fused Disp: out = sum_d p_d * d with p = softmin over the upsampled cost.
// d out / d v_d = -p_d (d - out); v_d = trilinear taps of the coarse cost -> scatter (dcost pre-zeroed).
struct DispBwdArgs {
  const float* cost;
  const float* dout;
  float* dcost;
  int d, h, w, maxdisp, Ho, Wo;
  float sd, sh, sw;
};
// A workgroup owns a 16 x 16 tile of fine pixels; the coarse cells its pixels touch form a window of at most 8 x 8.  All pixels
// reach a coarse plane's flush point at the same fine disparity, so a plane's contributions are reduced over the tile WITHOUT
// atomics: every thread parks its value in LDS, a separable gather (columns, then rows: W_x[16][8] and W_y[16][8] hold the
// bilinear weights of each fine column / row onto the window's columns / rows) gives the 64 cell sums, and each cell is added to
// global memory once per plane.  (The first version summed into an LDS window with ds_add_f32: ~9 lanes per address, 447 of the
// kernel's 565 us.)
constexpr int DB_T = 16, DB_WIN = 8;      // fine tile edge; coarse window edge (16/3 -> 6 cells + 1 each side for the taps)
__global__ __launch_bounds__(256) void disp_softargmin_bwd_kernel(DispBwdArgs a) {
  extern __shared__ float4 ztab[];         // [maxdisp]: the fine-disparity taps, as in the forward kernel
  __shared__ float pix[DB_T * DB_T], colsum[DB_T][DB_WIN], wxw[DB_T][DB_WIN], wyw[DB_T][DB_WIN];
  const int b = blockIdx.z, tid = threadIdx.x;
  const int tx = tid & (DB_T - 1), ty = tid >> 4;
  const int ox = blockIdx.x * DB_T + tx, oy = blockIdx.y * DB_T + ty;
  const bool live = ox < a.Wo && oy < a.Ho;
  // window origin: the first coarse cell touched by the tile's first pixel (uniform over the workgroup)
  const int wy0 = lin_index(min(blockIdx.y * DB_T, a.Ho - 1), a.h, a.Ho, a.sh, 0).i0;
  const int wx0 = lin_index(min(blockIdx.x * DB_T, a.Wo - 1), a.w, a.Wo, a.sw, 0).i0;
  for (int dd = tid; dd < a.maxdisp; dd += 256) {
    const LinIdx lz = lin_index(dd, a.d, a.maxdisp, a.sd, 0);
    const bool same = lz.i1 == lz.i0;
    ztab[dd] = make_float4((float)lz.i0, lz.w0, same ? 0.f : lz.w1, same ? lz.w1 : 0.f);
  }
  if (tid < 2 * DB_T) {                    // bilinear weights of the tile's columns (tid < 16) / rows onto the window
    const bool col = tid < DB_T;
    const int k = tid & (DB_T - 1);
    const int o = (col ? blockIdx.x : blockIdx.y) * DB_T + k, n_out = col ? a.Wo : a.Ho;
    const LinIdx l = lin_index(min(o, n_out - 1), col ? a.w : a.h, n_out, col ? a.sw : a.sh, 0);
    const int j0 = l.i0 - (col ? wx0 : wy0), j1 = l.i1 - (col ? wx0 : wy0);
    for (int j = 0; j < DB_WIN; ++j) {
      const float wv = o < n_out ? (j == j0 ? l.w0 : 0.f) + (j == j1 ? l.w1 : 0.f) : 0.f;
      if (col) wxw[k][j] = wv; else wyw[k][j] = wv;
    }
  }
  __syncthreads();
  const int hw = a.h * a.w;
  const float* base = a.cost + (int64_t)b * a.d * hw;
  float* gbase = a.dcost + (int64_t)b * a.d * hw;
  const LinIdx ly = lin_index(min(oy, a.Ho - 1), a.h, a.Ho, a.sh, 0);
  const LinIdx lx = lin_index(min(ox, a.Wo - 1), a.w, a.Wo, a.sw, 0);
  const int o00 = ly.i0 * a.w + lx.i0, o01 = ly.i0 * a.w + lx.i1, o10 = ly.i1 * a.w + lx.i0, o11 = ly.i1 * a.w + lx.i1;
  auto plane = [&](int z) -> float {
    const float* p = base + (int64_t)z * hw;
    return lerp2(ly.w0, lerp2(lx.w0, p[o00], lx.w1, p[o01]), ly.w1, lerp2(lx.w0, p[o10], lx.w1, p[o11]));
  };
  // Both passes walk the fine disparities in order like the forward kernel (disp.hip): the coarse pair (cz, cz+1) only moves
  // forward, so each coarse plane is sampled once per pass, the fine taps come from the LDS table and the softmax runs in base 2
  // with one exponential per sample.
  constexpr float K = 1.4426950408889634f;
  // pass 1: softmax statistics (max, sum, expectation), exactly as the forward
  float m = -INFINITY, s = 0.f, ws = 0.f;
  {
    int cz = 0;
    float b0 = plane(0), b1 = plane(a.d > 1 ? 1 : 0);
    for (int dd = 0; dd < a.maxdisp; ++dd) {
      const float4 tb = ztab[dd];
      const int i0 = (int)tb.x;
      while (i0 > cz) { ++cz; b0 = b1; b1 = plane(cz + 1 < a.d ? cz + 1 : a.d - 1); }
      const float t = -fmaf(tb.y + tb.w, b0, tb.z * b1);
      const bool up = t > m;
      const float x = __builtin_amdgcn_exp2f((up ? m - t : t - m) * K);
      const float r = up ? x : 1.f, e = up ? 1.f : x;
      s = fmaf(s, r, e);
      ws = fmaf(ws, r, e * (float)dd);
      m = up ? t : m;
    }
  }
  const float outv = ws / s;
  const float ginv = live ? -a.dout[((int64_t)b * a.Ho + oy) * a.Wo + ox] / s : 0.f;   // pixels outside the image contribute nothing
  // one coarse plane's gradient: tile-wide reduction of the per-pixel values onto the window cells, then one global add per cell
  auto flush = [&](int z, float gv) {      // called by EVERY thread at the same point
    pix[tid] = gv;
    __syncthreads();
    if (tid < DB_T * DB_WIN) {             // (row r, window column j): sum over the row's 16 columns
      const int r = tid >> 3, j = tid & (DB_WIN - 1);
      float acc = 0.f;
#pragma unroll
      for (int c = 0; c < DB_T; ++c) acc = fmaf(wxw[c][j], pix[r * DB_T + c], acc);
      colsum[r][j] = acc;
    }
    __syncthreads();
    if (tid < DB_WIN * DB_WIN) {           // (window row jy, window column jx): sum over the 16 rows
      const int jy = tid >> 3, jx = tid & (DB_WIN - 1);
      float acc = 0.f;
#pragma unroll
      for (int r = 0; r < DB_T; ++r) acc = fmaf(wyw[r][jy], colsum[r][jx], acc);
      const int cx = wx0 + jx, cy = wy0 + jy;
      if (acc != 0.f && cx < a.w && cy < a.h) atomicAdd(gbase + (int64_t)z * hw + cy * a.w + cx, acc);
    }
    // (the next flush writes pix only after its own barrier-separated readers are done: colsum readers finish before any thread
    // can pass the first barrier of the next flush, because they must arrive at it themselves)
  };
  // pass 2: walk the fine samples again; a0 / a1 collect the gradient of coarse planes cz / cz+1
  int cz = 0;
  float b0 = plane(0), b1 = plane(a.d > 1 ? 1 : 0);
  float a0 = 0.f, a1 = 0.f;
  for (int dd = 0; dd < a.maxdisp; ++dd) {
    const float4 tb = ztab[dd];
    const int i0 = (int)tb.x;
    while (i0 > cz) {                      // wave- and workgroup-uniform: the table is the same for every pixel
      flush(cz, a0); a0 = a1; a1 = 0.f;
      ++cz; b0 = b1; b1 = plane(cz + 1 < a.d ? cz + 1 : a.d - 1);
    }
    const float t = -fmaf(tb.y + tb.w, b0, tb.z * b1);
    const float gv = ginv * __builtin_amdgcn_exp2f((t - m) * K) * ((float)dd - outv);   // d out / d v_fine, v = +cost (softMIN)
    a0 = fmaf(gv, tb.y + tb.w, a0);
    a1 = fmaf(gv, tb.z, a1);
  }
  flush(cz, a0);
  if (cz + 1 < a.d) flush(cz + 1, a1);
}

// ---------------------------------------------------------------------------------------------------------------
// Feature-Net stem (2-D 3x3, pad 1, stride s): data gradient as a gather, weight gradient as a blocked reduction.
// dx[b,ci,y,x] = sum_{co,ky,kx : (y+1-ky) % s == 0, (x+1-kx) % s == 0} g[b,co,(y+1-ky)/s,(x+1-kx)/s] * w[co,ci,ky,kx]
__global__ __launch_bounds__(256) void conv2d_strided_dgrad_kernel(const float* __restrict__ g, const float* __restrict__ w,
                                                                   float* __restrict__ dx, int Cin, int Cout, int H, int W, int Ho,
                                                                   int Wo, int s) {
  const int64_t pix = (int64_t)H * W, opix = (int64_t)Ho * Wo;
  const int64_t p = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (p >= pix) return;
  const int ci = blockIdx.y, b = blockIdx.z;
  const int y = (int)(p / W), x = (int)(p % W);
  float acc = 0.f;
  for (int ky = 0; ky < 3; ++ky) {
    const int ty = y + 1 - ky;
    if (ty < 0 || ty % s != 0 || ty / s >= Ho) continue;
    for (int kx = 0; kx < 3; ++kx) {
      const int tx = x + 1 - kx;
      if (tx < 0 || tx % s != 0 || tx / s >= Wo) continue;
      const float* pg = g + (int64_t)b * Cout * opix + (int64_t)(ty / s) * Wo + tx / s;
      for (int co = 0; co < Cout; ++co) acc = fmaf(pg[co * opix], w[((co * Cin + ci) * 3 + ky) * 3 + kx], acc);
    }
  }
  dx[((int64_t)b * Cin + ci) * pix + p] = acc;
}

// dw[co,ci,ky,kx] += sum_{b,oy,ox} g[b,co,oy,ox] * x[b,ci,oy*s-1+ky,ox*s-1+kx];  blockIdx.y = co*Cin+ci, blockIdx.x = pixel slab
__global__ __launch_bounds__(256) void conv2d_strided_wgrad_kernel(const float* __restrict__ x, const float* __restrict__ g,
                                                                   float* __restrict__ dw, int B, int Cin, int Cout, int H, int W,
                                                                   int Ho, int Wo, int s) {
  const int co = blockIdx.y / Cin, ci = blockIdx.y % Cin;
  const int64_t opix = (int64_t)Ho * Wo, total = opix * B;
  float acc[9];
#pragma unroll
  for (int t = 0; t < 9; ++t) acc[t] = 0.f;
  for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (int64_t)gridDim.x * 256) {
    const int b = (int)(e / opix);
    const int64_t o = e % opix;
    const int oy = (int)(o / Wo), ox = (int)(o % Wo);
    const float gv = g[((int64_t)b * Cout + co) * opix + o];
    const float* px = x + ((int64_t)b * Cin + ci) * H * W;
#pragma unroll
    for (int t = 0; t < 9; ++t) {
      const int iy = oy * s - 1 + t / 3, ix = ox * s - 1 + t % 3;
      if ((unsigned)iy < (unsigned)H && (unsigned)ix < (unsigned)W) acc[t] = fmaf(gv, px[(int64_t)iy * W + ix], acc[t]);
    }
  }
  __shared__ float red[9][256];
#pragma unroll
  for (int t = 0; t < 9; ++t) red[t][threadIdx.x] = acc[t];
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if ((int)threadIdx.x < o) {
#pragma unroll
      for (int t = 0; t < 9; ++t) red[t][threadIdx.x] += red[t][threadIdx.x + o];
    }
    __syncthreads();
  }
  if (threadIdx.x < 9) atomicAdd(dw + (int64_t)blockIdx.y * 9 + threadIdx.x, red[threadIdx.x][0]);
}

// adjoint of DisparityRegression: dprob[b,d,y,x] = dout[b,y,x] * d
__global__ __launch_bounds__(256) void disparity_regression_bwd_kernel(const float* __restrict__ dout, float* __restrict__ dprob, int D,
                                                                       int64_t hw) {
  const int64_t p = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (p >= hw) return;
  const int b = blockIdx.y;
  const float gv = dout[(int64_t)b * hw + p];
  for (int d = 0; d < D; ++d) dprob[((int64_t)b * D + d) * hw + p] = gv * (float)d;
}

}  // namespace ragmi

// ---------------------------------------------------------------------------------------------------------------
// workgroups along the voxel axis of the two-level reductions: enough to fill the chip, few enough that the second level is short
namespace ragmi {
static unsigned reduce_blocks(int B, int C, int64_t DHW) {
  const int64_t want = ceil_div(4096, (int64_t)B * C);
  // <= 256 partials per channel (gx * B): the fused second halves re-reduce them in every workgroup
  return (unsigned)std::max<int64_t>(1, std::min<int64_t>(std::min<int64_t>(want, ceil_div(DHW, 1024)), std::max(1, 256 / B)));
}
}  // namespace ragmi

extern "C" int64_t ragmi_bn_workspace_elems(int B, int C, int64_t DHW) {
  if (B <= 0 || C <= 0 || DHW <= 0) return 0;
  return (int64_t)2 * C * B * ragmi::reduce_blocks(B, C, DHW);
}

extern "C" int ragmi_bn_train_stats_fwd(const void* x, int64_t x_bstride, int B, int C, int64_t DHW, const void* gamma, const void* beta,
                                        void* running_mean, void* running_var, void* num_batches_tracked, float momentum, float eps,
                                        void* workspace, void* mean, void* invstd, void* scale, void* shift, void* stream) {
  using namespace ragmi;
  RAGMI_REQUIRE(x && gamma && beta && workspace && mean && invstd && scale && shift, RAGMI_EINVAL, "bn_train_stats: null pointer");
  RAGMI_REQUIRE((running_mean == nullptr) == (running_var == nullptr), RAGMI_EINVAL, "bn_train_stats: running_mean/var go together");
  RAGMI_REQUIRE(B > 0 && C > 0 && DHW > 0 && B <= 65535 && C <= 65535, RAGMI_EINVAL, "bn_train_stats: bad size");
  RAGMI_REQUIRE(momentum >= 0.f && momentum <= 1.f, RAGMI_EUNSUPPORTED, "bn_train_stats: momentum must be in [0,1] (cumulative average not built)");
  const unsigned gx = reduce_blocks(B, C, DHW);
  const int nparts = (int)(gx * B);
  hipStream_t st = static_cast<hipStream_t>(stream);
  hipLaunchKernelGGL(bn_stats_kernel, dim3(gx, C, B), dim3(256), 0, st, (const float*)x, x_bstride, DHW, (float*)workspace, nparts);
  BnFinalizeArgs a{(const float*)workspace, (const float*)gamma, (const float*)beta, (float*)running_mean, (float*)running_var,
                   (long long*)num_batches_tracked, (float*)mean, (float*)invstd, (float*)scale, (float*)shift, nparts,
                   (double)B * (double)DHW, eps, momentum};
  hipLaunchKernelGGL(bn_finalize_kernel, dim3(C), dim3(256), 0, st, a);
  return check_launch("bn_train_stats");
}

extern "C" int ragmi_bn_train_act_fwd(const void* x, int64_t x_bstride, int B, int C, int64_t DHW, const void* gamma, const void* beta,
                                      void* running_mean, void* running_var, void* num_batches_tracked, float momentum, float eps,
                                      int relu, void* workspace, void* mean, void* invstd, void* scale, void* shift, void* y,
                                      int64_t y_bstride, int y_ch0, const void* res, int64_t res_bstride, int res_ch0, void* stream) {
  using namespace ragmi;
  RAGMI_REQUIRE(x && gamma && beta && workspace && mean && invstd && scale && shift && y, RAGMI_EINVAL, "bn_train_act: null pointer");
  RAGMI_REQUIRE((running_mean == nullptr) == (running_var == nullptr), RAGMI_EINVAL, "bn_train_act: running_mean/var go together");
  RAGMI_REQUIRE(B > 0 && C > 0 && DHW > 0 && B <= 65535 && C <= 65535, RAGMI_EINVAL, "bn_train_act: bad size");
  RAGMI_REQUIRE(momentum >= 0.f && momentum <= 1.f, RAGMI_EUNSUPPORTED, "bn_train_act: momentum must be in [0,1] (cumulative average not built)");
  const unsigned gx = reduce_blocks(B, C, DHW);
  const int nparts = (int)(gx * B);
  hipStream_t st = static_cast<hipStream_t>(stream);
  hipLaunchKernelGGL(bn_stats_kernel, dim3(gx, C, B), dim3(256), 0, st, (const float*)x, x_bstride, DHW, (float*)workspace, nparts);
  BnFinActArgs a{};
  a.f = BnFinalizeArgs{(const float*)workspace, (const float*)gamma, (const float*)beta, (float*)running_mean, (float*)running_var,
                       (long long*)num_batches_tracked, (float*)mean, (float*)invstd, (float*)scale, (float*)shift, nparts,
                       (double)B * (double)DHW, eps, momentum};
  a.x = (const float*)x; a.y = (float*)y; a.x_bstride = x_bstride; a.y_bstride = y_bstride; a.dhw = DHW; a.y_ch0 = y_ch0; a.relu = relu;
  a.res = (const float*)res; a.res_bstride = res_bstride; a.res_ch0 = res_ch0;
  hipLaunchKernelGGL(bn_finalize_act_kernel, dim3((unsigned)ceil_div(DHW, 1024), C, B), dim3(256), 0, st, a);
  return check_launch("bn_train_act");
}

extern "C" int ragmi_bn_act_bwd(const void* dy, int64_t dy_bstride, int dy_ch0, const void* x, int64_t x_bstride, const void* scale,
                                const void* shift, int relu, const void* mean, const void* invstd, int training, int B, int C,
                                int64_t DHW, void* workspace, void* dx, int64_t dx_bstride, void* dgamma, void* dbeta, int accumulate,
                                void* stream) {
  using namespace ragmi;
  RAGMI_REQUIRE(dy && x && scale && shift && mean && invstd && workspace && dx, RAGMI_EINVAL, "bn_act_bwd: null pointer");
  RAGMI_REQUIRE(B > 0 && C > 0 && DHW > 0 && B <= 65535 && C <= 65535, RAGMI_EINVAL, "bn_act_bwd: bad size");
  const unsigned gx = reduce_blocks(B, C, DHW);
  const int nparts = (int)(gx * B);
  hipStream_t st = static_cast<hipStream_t>(stream);
  hipLaunchKernelGGL(bn_act_bwd_reduce_kernel, dim3(gx, C, B), dim3(256), 0, st, (const float*)dy, dy_bstride, dy_ch0, (const float*)x,
                     x_bstride, (const float*)scale, (const float*)shift, relu, DHW, (float*)workspace, nparts);
  BnCoeffApplyArgs a{};
  a.part = (const float*)workspace; a.mean = (const float*)mean; a.invstd = (const float*)invstd; a.scale = (const float*)scale;
  a.shift = (const float*)shift; a.dgamma = (float*)dgamma; a.dbeta = (float*)dbeta; a.dy = (const float*)dy; a.x = (const float*)x;
  a.dx = (float*)dx; a.dy_bstride = dy_bstride; a.x_bstride = x_bstride; a.dx_bstride = dx_bstride; a.dhw = DHW; a.nparts = nparts;
  a.training = training ? 1 : 0; a.accumulate = accumulate ? 1 : 0; a.relu = relu; a.dy_ch0 = dy_ch0; a.n = (double)B * (double)DHW;
  hipLaunchKernelGGL(bn_coeffs_apply_kernel, dim3((unsigned)ceil_div(DHW, 1024), C, B), dim3(256), 0, st, a);
  return check_launch("bn_act_bwd");
}

extern "C" int ragmi_bn_act_fwd(const void* x, int64_t x_bstride, const void* scale, const void* shift, int relu, const void* res,
                                int64_t res_bstride, int res_ch0, void* y, int64_t y_bstride, int y_ch0, int B, int C, int64_t DHW,
                                void* stream) {
  using namespace ragmi;
  RAGMI_REQUIRE(x && scale && shift && y, RAGMI_EINVAL, "bn_act: null pointer");
  RAGMI_REQUIRE(B > 0 && C > 0 && DHW > 0 && B <= 65535 && C <= 65535, RAGMI_EINVAL, "bn_act: bad size");
  hipLaunchKernelGGL(bn_act_kernel, dim3((unsigned)ceil_div(DHW, 1024), C, B), dim3(256), 0, static_cast<hipStream_t>(stream),
                     (const float*)x, x_bstride, (const float*)scale, (const float*)shift, relu, (const float*)res, res_bstride, res_ch0,
                     (float*)y, y_bstride, y_ch0, DHW);
  return check_launch("bn_act");
}

extern "C" int ragmi_bn_act_bwd_coeffs(const void* dy, int64_t dy_bstride, int dy_ch0, const void* x, int64_t x_bstride,
                                      const void* scale, const void* shift, int relu, const void* mean, const void* invstd, int training,
                                      int B, int C, int64_t DHW, void* workspace, void* c1, void* c2, void* c3, void* dgamma, void* dbeta,
                                      int accumulate, void* stream) {
  using namespace ragmi;
  RAGMI_REQUIRE(dy && x && scale && shift && mean && invstd && workspace && c1 && c2 && c3, RAGMI_EINVAL,
                "bn_act_bwd_coeffs: null pointer");
  RAGMI_REQUIRE(B > 0 && C > 0 && DHW > 0 && B <= 65535 && C <= 65535, RAGMI_EINVAL, "bn_act_bwd_coeffs: bad size");
  const unsigned gx = reduce_blocks(B, C, DHW);
  const int nparts = (int)(gx * B);
  hipStream_t st = static_cast<hipStream_t>(stream);
  hipLaunchKernelGGL(bn_act_bwd_reduce_kernel, dim3(gx, C, B), dim3(256), 0, st, (const float*)dy, dy_bstride, dy_ch0, (const float*)x,
                     x_bstride, (const float*)scale, (const float*)shift, relu, DHW, (float*)workspace, nparts);
  BnCoeffArgs a{(const float*)workspace, (const float*)mean, (const float*)invstd, (const float*)scale, (float*)c1, (float*)c2, (float*)c3,
                (float*)dgamma, (float*)dbeta, nparts, training ? 1 : 0, accumulate ? 1 : 0, (double)B * (double)DHW};
  hipLaunchKernelGGL(bn_bwd_coeffs_kernel, dim3(C), dim3(256), 0, st, a);
  return check_launch("bn_act_bwd_coeffs");
}

extern "C" int ragmi_bn_act_bwd_apply(const void* dy, int64_t dy_bstride, int dy_ch0, const void* x, int64_t x_bstride,
                                      const void* scale, const void* shift, int relu, const void* c1, const void* c2, const void* c3,
                                      void* dx, int64_t dx_bstride, int B, int C, int64_t DHW, void* stream) {
  using namespace ragmi;
  RAGMI_REQUIRE(dy && x && scale && shift && c1 && c2 && c3 && dx, RAGMI_EINVAL, "bn_act_bwd_apply: null pointer");
  RAGMI_REQUIRE(B > 0 && C > 0 && DHW > 0 && B <= 65535 && C <= 65535, RAGMI_EINVAL, "bn_act_bwd_apply: bad size");
  hipLaunchKernelGGL(bn_act_bwd_apply_kernel, dim3((unsigned)ceil_div(DHW, 1024), C, B), dim3(256), 0, static_cast<hipStream_t>(stream),
                     (const float*)dy, dy_bstride, dy_ch0, (const float*)x, x_bstride, (const float*)scale, (const float*)shift, relu,
                     (const float*)c1, (const float*)c2, (const float*)c3, (float*)dx, dx_bstride, DHW);
  return check_launch("bn_act_bwd_apply");
}

namespace ragmi {
struct WgradPlan {
  int cg, gx, gy, gz, tx, ty, tz;
  int64_t ntiles;
};
// launch geometry shared by the workspace query and the launch
static int wgrad_plan(int B, int Cin, int Cout, int D, int H, int W, WgradPlan& p) {
  p.tx = (int)ceil_div(W, WG_TX); p.ty = (int)ceil_div(H, WG_TY); p.tz = (int)ceil_div(D, WG_TZ);
  p.ntiles = (int64_t)p.tx * p.ty * p.tz * B;
  // output-channel groups per workgroup: the largest of 4,3,2,1 that divides the group count (12 -> 3, 16 -> 4, 24 -> 3 x 2)
  const int ngroups = (int)ceil_div(Cout, 4);
  p.cg = 1;
  for (int c = 4; c >= 1; --c)
    if (ngroups % c == 0) { p.cg = c; break; }
  p.gy = (int)ceil_div(Cin, 4);
  p.gz = ngroups / p.cg;
  RAGMI_REQUIRE(p.ntiles < (1ll << 31) && p.gy <= 65535 && p.gz <= 65535, RAGMI_EUNSUPPORTED, "conv3d_k3_wgrad: grid too large");
  RAGMI_REQUIRE((int64_t)std::max(Cin, Cout) * D * H * W < (1ll << 31), RAGMI_EUNSUPPORTED, "conv3d_k3_wgrad: volume too large for 32-bit offsets");
  // persistent workgroups: as many as are resident at once (occupancy x CUs), per device
  static LaunchState state[5];
  const void* fn = p.cg == 4 ? (const void*)conv3d_k3_wgrad_kernel<4, true> : p.cg == 3 ? (const void*)conv3d_k3_wgrad_kernel<3, true>
                 : p.cg == 2 ? (const void*)conv3d_k3_wgrad_kernel<2, true> : (const void*)conv3d_k3_wgrad_kernel<1, true>;
  const int resident = state[p.cg].slots(fn, 192, 0, 64 * 1024);
  if (resident <= 0) return fail(RAGMI_ELAUNCH, "conv3d_k3_wgrad: occupancy query failed");
  p.gx = (int)std::max<int64_t>(1, std::min<int64_t>(p.ntiles, resident / ((int64_t)p.gy * p.gz)));
  return RAGMI_OK;
}
}  // namespace ragmi

extern "C" int64_t ragmi_conv3d_k3_wgrad_workspace_elems(int B, int Cin, int Cout, int D, int H, int W) {
  using namespace ragmi;
  if (B <= 0 || Cin <= 0 || Cout <= 0 || D <= 0 || H <= 0 || W <= 0) return 0;
  WgradPlan p;
  if (wgrad_plan(B, Cin, Cout, D, H, W, p) != RAGMI_OK) return -1;
  return (int64_t)p.gx * (p.gz * p.cg * 4) * (p.gy * 4) * 27;
}

extern "C" int ragmi_conv3d_k3_wgrad(const void* x, int64_t x_bstride, const void* g, int64_t g_bstride, int g_ch0, void* const* dw_list,
                                     int n_dw, int accumulate, int planar2d, void* workspace, int B, int Cin, int Cout, int D, int H,
                                     int W, void* stream) {
  using namespace ragmi;
  RAGMI_REQUIRE(x && g && dw_list && workspace, RAGMI_EINVAL, "conv3d_k3_wgrad: null pointer");
  RAGMI_REQUIRE(B > 0 && Cin > 0 && Cout > 0 && D > 0 && H > 0 && W > 0, RAGMI_EINVAL, "conv3d_k3_wgrad: bad size");
  RAGMI_REQUIRE(n_dw >= 1 && n_dw <= 8 && Cout % n_dw == 0, RAGMI_EINVAL, "conv3d_k3_wgrad: 1..8 destinations that split Cout evenly");
  WgradDst dst{};
  for (int k = 0; k < n_dw; ++k) {
    RAGMI_REQUIRE(dw_list[k], RAGMI_EINVAL, "conv3d_k3_wgrad: null destination");
    dst.p[k] = static_cast<float*>(dw_list[k]);
  }
  dst.n = n_dw; dst.rows_per_dst = Cout / n_dw; dst.accumulate = accumulate ? 1 : 0; dst.planar = planar2d ? 1 : 0;
  WgradPlan p;
  const int rc = wgrad_plan(B, Cin, Cout, D, H, W, p);
  if (rc != RAGMI_OK) return rc;
  const bool vec = W % 4 == 0 && x_bstride % 4 == 0 && g_bstride % 4 == 0 && (reinterpret_cast<uintptr_t>(x) & 15) == 0 &&
                   (reinterpret_cast<uintptr_t>(g) & 15) == 0;
  WgradArgs a{(const float*)x, (const float*)g, (float*)workspace, x_bstride, g_bstride, g_ch0, Cin, Cout, D, H, W, p.tx, p.ty, p.tz,
              (int)p.ntiles};
  const dim3 grid(p.gx, p.gy, p.gz), block(192);
  hipStream_t st = static_cast<hipStream_t>(stream);
#define RAGMI_WGRAD_LAUNCH(CGV)                                                                   \
  if (vec) hipLaunchKernelGGL((conv3d_k3_wgrad_kernel<CGV, true>), grid, block, 0, st, a);        \
  else hipLaunchKernelGGL((conv3d_k3_wgrad_kernel<CGV, false>), grid, block, 0, st, a)
  switch (p.cg) {
    case 4: RAGMI_WGRAD_LAUNCH(4); break;
    case 3: RAGMI_WGRAD_LAUNCH(3); break;
    case 2: RAGMI_WGRAD_LAUNCH(2); break;
    default: RAGMI_WGRAD_LAUNCH(1); break;
  }
#undef RAGMI_WGRAD_LAUNCH
  const int total = Cout * Cin * 27;
  hipLaunchKernelGGL(conv3d_k3_wgrad_reduce_kernel, dim3((unsigned)ceil_div(total, 4)), dim3(256), 0, st, (const float*)workspace, dst,
                     p.gx, Cin, Cout, p.gy * 4, p.gz * p.cg * 4);
  return check_launch("conv3d_k3_wgrad");
}

extern "C" int ragmi_conv3d_k1_wgrad(const void* x, int64_t x_bstride, const void* g, int64_t g_bstride, int g_ch0, void* dw, int B,
                                     int Cin, int Cout, int64_t DHW, void* stream) {
  using namespace ragmi;
  RAGMI_REQUIRE(x && g && dw, RAGMI_EINVAL, "conv3d_k1_wgrad: null pointer");
  RAGMI_REQUIRE(B > 0 && Cin > 0 && Cout > 0 && DHW > 0 && B <= 65535 && Cout <= 65535, RAGMI_EINVAL, "conv3d_k1_wgrad: bad size");
  const int nci = (int)ceil_div(Cin, K1W_CI), nco = (int)ceil_div(Cout, K1W_CO);
  RAGMI_REQUIRE((int64_t)nci * nco <= 65535, RAGMI_EUNSUPPORTED, "conv3d_k1_wgrad: too many channel blocks");
  // about two workgroups per CU over the launch; each flushes 48 same-address atomics (measured on the training step:
  // 256 -> 282.8, 512 -> 283.5, 1024 -> 281.1, 2048 -> 277.2 pairs/s)
  const unsigned gx = (unsigned)std::max<int64_t>(1, std::min<int64_t>(ceil_div(DHW, 256), ceil_div(512, (int64_t)nci * nco * B)));
  hipLaunchKernelGGL(conv3d_k1_wgrad_kernel, dim3(gx, nci * nco, B), dim3(256), 0, static_cast<hipStream_t>(stream), (const float*)x,
                     x_bstride, (const float*)g, g_bstride, g_ch0, (float*)dw, Cin, Cout, DHW, nci);
  return check_launch("conv3d_k1_wgrad");
}

extern "C" int ragmi_trilinear3d_bwd(const void* dy, void* dx, int B, int C, int Di, int Hi, int Wi, int Do, int Ho, int Wo,
                                     int align_corners, void* stream) {
  using namespace ragmi;
  RAGMI_REQUIRE(dy && dx, RAGMI_EINVAL, "trilinear3d_bwd: null pointer");
  RAGMI_REQUIRE(B > 0 && C > 0 && Di > 0 && Hi > 0 && Wi > 0 && Do > 0 && Ho > 0 && Wo > 0 && B <= 65535, RAGMI_EINVAL,
                "trilinear3d_bwd: bad size");
  TriBwdArgs a{(const float*)dy, (float*)dx, C, Di, Hi, Wi, Do, Ho, Wo, lin_scale(Di, Do, align_corners),
               lin_scale(Hi, Ho, align_corners), lin_scale(Wi, Wo, align_corners), align_corners ? 1 : 0, 1};
  // enough threads to fill the chip (>= ~256K), otherwise as many channels per thread as possible
  const int64_t vox_threads = (int64_t)Di * Hi * Wi * B;
  a.cpt = (int)std::max<int64_t>(1, std::min<int64_t>(C, vox_threads * C / (256 * 1024)));
  RAGMI_REQUIRE(ceil_div(C, a.cpt) <= 65535, RAGMI_EUNSUPPORTED, "trilinear3d_bwd: too many channels");
  hipLaunchKernelGGL(trilinear_bwd_kernel, dim3((unsigned)ceil_div((int64_t)Di * Hi * Wi, 256), B, (unsigned)ceil_div(C, a.cpt)), dim3(256), 0,
                     static_cast<hipStream_t>(stream), a);
  return check_launch("trilinear3d_bwd");
}

extern "C" int ragmi_costvol_bwd(const void* dcost, void* dleft, void* dright, int B, int C, int d, int h, int w, void* stream) {
  using namespace ragmi;
  RAGMI_REQUIRE(dcost && dleft && dright, RAGMI_EINVAL, "costvol_bwd: null pointer");
  RAGMI_REQUIRE(B > 0 && C > 0 && d > 0 && h > 0 && w > 0 && B <= 65535 && C <= 65535, RAGMI_EINVAL, "costvol_bwd: bad size");
  hipLaunchKernelGGL(costvol_bwd_kernel, dim3((unsigned)ceil_div((int64_t)h * w, 256), C, B), dim3(256), 0, static_cast<hipStream_t>(stream),
                     (const float*)dcost, (float*)dleft, (float*)dright, C, d, h, w);
  return check_launch("costvol_bwd");
}

extern "C" int ragmi_disp_softargmin_bwd(const void* cost, const void* dout, void* dcost, int B, int d, int h, int w, int maxdisp,
                                         int Ho, int Wo, void* stream) {
  using namespace ragmi;
  RAGMI_REQUIRE(cost && dout && dcost, RAGMI_EINVAL, "disp_softargmin_bwd: null pointer");
  RAGMI_REQUIRE(B > 0 && d > 0 && h > 0 && w > 0 && maxdisp > 0 && Ho > 0 && Wo > 0 && B <= 65535, RAGMI_EINVAL,
                "disp_softargmin_bwd: bad size");
  DispBwdArgs a{(const float*)cost, (const float*)dout, (float*)dcost, d, h, w, maxdisp, Ho, Wo,
                lin_scale(d, maxdisp, 0), lin_scale(h, Ho, 0), lin_scale(w, Wo, 0)};
  // the coarse window of a 16 x 16 fine tile must fit DB_WIN cells per axis: holds for the x3 upsample of Disp (16/3 + 2 taps <= 8)
  RAGMI_REQUIRE(Ho == 3 * h && Wo == 3 * w, RAGMI_EUNSUPPORTED, "disp_softargmin_bwd: built for the x3 upsample of Disp (Ho = 3h, Wo = 3w)");
  const size_t lds = (size_t)maxdisp * sizeof(float4);
  RAGMI_REQUIRE(lds <= 48 * 1024, RAGMI_EUNSUPPORTED, "disp_softargmin_bwd: maxdisp = %d exceeds the tap table (3072)", maxdisp);
  hipLaunchKernelGGL(disp_softargmin_bwd_kernel, dim3((unsigned)ceil_div(Wo, DB_T), (unsigned)ceil_div(Ho, DB_T), B), dim3(256), lds,
                     static_cast<hipStream_t>(stream), a);
  return check_launch("disp_softargmin_bwd");
}

extern "C" int ragmi_conv2d_k3_strided_dgrad(const void* g, const void* weight, void* dx, int B, int Cin, int Cout, int H, int W,
                                             int stride, void* stream) {
  using namespace ragmi;
  RAGMI_REQUIRE(g && weight && dx, RAGMI_EINVAL, "conv2d_k3_strided_dgrad: null pointer");
  RAGMI_REQUIRE(B > 0 && Cin > 0 && Cout > 0 && H > 0 && W > 0 && stride >= 1 && B <= 65535 && Cin <= 65535, RAGMI_EINVAL,
                "conv2d_k3_strided_dgrad: bad size");
  const int Ho = (H - 1) / stride + 1, Wo = (W - 1) / stride + 1;
  hipLaunchKernelGGL(conv2d_strided_dgrad_kernel, dim3((unsigned)ceil_div((int64_t)H * W, 256), Cin, B), dim3(256), 0,
                     static_cast<hipStream_t>(stream), (const float*)g, (const float*)weight, (float*)dx, Cin, Cout, H, W, Ho, Wo, stride);
  return check_launch("conv2d_k3_strided_dgrad");
}

extern "C" int ragmi_conv2d_k3_strided_wgrad(const void* x, const void* g, void* dw, int B, int Cin, int Cout, int H, int W, int stride,
                                             void* stream) {
  using namespace ragmi;
  RAGMI_REQUIRE(x && g && dw, RAGMI_EINVAL, "conv2d_k3_strided_wgrad: null pointer");
  RAGMI_REQUIRE(B > 0 && Cin > 0 && Cout > 0 && H > 0 && W > 0 && stride >= 1 && (int64_t)Cin * Cout <= 65535, RAGMI_EINVAL,
                "conv2d_k3_strided_wgrad: bad size");
  const int Ho = (H - 1) / stride + 1, Wo = (W - 1) / stride + 1;
  const unsigned gx = (unsigned)std::min<int64_t>(ceil_div((int64_t)B * Ho * Wo, 1024), 64);
  hipLaunchKernelGGL(conv2d_strided_wgrad_kernel, dim3(gx, Cin * Cout), dim3(256), 0, static_cast<hipStream_t>(stream), (const float*)x,
                     (const float*)g, (float*)dw, B, Cin, Cout, H, W, Ho, Wo, stride);
  return check_launch("conv2d_k3_strided_wgrad");
}

extern "C" int ragmi_disparity_regression_bwd(const void* dout, void* dprob, int B, int D, int H, int W, void* stream) {
  using namespace ragmi;
  RAGMI_REQUIRE(dout && dprob, RAGMI_EINVAL, "disparity_regression_bwd: null pointer");
  RAGMI_REQUIRE(B > 0 && D > 0 && H > 0 && W > 0 && B <= 65535, RAGMI_EINVAL, "disparity_regression_bwd: bad size");
  hipLaunchKernelGGL(disparity_regression_bwd_kernel, dim3((unsigned)ceil_div((int64_t)H * W, 256), B), dim3(256), 0,
                     static_cast<hipStream_t>(stream), (const float*)dout, (float*)dprob, D, (int64_t)H * W);
  return check_launch("disparity_regression_bwd");
}
