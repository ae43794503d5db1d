// K4 — 1x1x1 ConvBR_3d (channel mix + folded BN + ReLU) and the identity-branch add.
// Reference: Cell_3d.pre_preprocess/preprocess (src/models/rag_model.py:125-126, 154-155),
// last_6_3d / last_12_3d (:270-271), ConvBR_3d (src/automl/operations_3d.py:31-47).
//
// HBM-bound (AI = 2*Cin*Cout / 4(Cin+Cout) <= 8 FLOP/B): a thread owns 4 consecutive
// voxels (one 16-B column of every channel plane), streams the Cin planes once with
// coalesced dwordx4 loads, keeps all NCO outputs in registers, and writes NCO coalesced
// dwordx4 stores.  Weights/scale/shift are wave-uniform -> scalar (SGPR) loads.
#include <algorithm>

#include "common.h"

namespace ragmi {

struct K1Args {
  const void* x;   // activations: float or bf16_t, per the kernel's storage type
  int64_t x_bstride;
  const float* w;  // element (co, ci) at w[co * w_sco + ci * w_sci]: [Cout][Cin] row-major, or its transpose read in place
  const float* scale;
  const float* shift;
  void* y;
  int64_t y_bstride;
  int y_ch0;
  int Cin, Cout, co0;
  int64_t dhw;
  int relu;
  int w_sco, w_sci;
  int y_f32;       // bf16 storage kernels only: y is fp32 (RAGMI_BF16 | RAGMI_OUT_F32: the resample launch that crosses into the fp32 levels)
};

template <class T, int NCO, bool VEC>
__global__ __launch_bounds__(256) void conv_k1_kernel(K1Args a) {
  constexpr int V = VEC ? 4 : 1;
  // this slab's weights [ci][NCO] through LDS (broadcast 16-byte reads) instead of NCO scalar loads per input channel
  extern __shared__ __attribute__((aligned(16))) float k1_w[];
  a.co0 = blockIdx.z * NCO;   // output-channel slab of this block
  for (int i = threadIdx.x; i < a.Cin * NCO; i += 256) {
    const int ci = i / NCO, co = a.co0 + i % NCO;
    k1_w[i] = co < a.Cout ? a.w[co * a.w_sco + ci * a.w_sci] : 0.f;
  }
  // the slab's folded BatchNorm beside the weights: read per output channel in the epilogue they were two dependent global loads and
  // a vmcnt(0) per channel, one after the other — up to NCO serial memory round trips behind the last multiply-add of kernels that
  // take 5-20 us (round 5, found in the listing)
  float* const k1_bn = k1_w + a.Cin * NCO;                 // scale[NCO] | shift[NCO]
  if (threadIdx.x < NCO) {
    const int co = min(a.co0 + (int)threadIdx.x, a.Cout - 1);
    k1_bn[threadIdx.x] = a.scale ? a.scale[co] : 1.f;
    k1_bn[NCO + threadIdx.x] = a.scale ? a.shift[co] : 0.f;
  }
  __syncthreads();
  const int64_t p = ((int64_t)blockIdx.x * 256 + threadIdx.x) * V;
  if (p >= a.dhw) return;
  const int b = blockIdx.y;
  const T* xp = static_cast<const T*>(a.x) + b * a.x_bstride + p;
  float acc[NCO][V];
#pragma unroll
  for (int j = 0; j < NCO; ++j)
#pragma unroll
    for (int k = 0; k < V; ++k) acc[j][k] = 0.f;

#pragma unroll 4
  for (int ci = 0; ci < a.Cin; ++ci) {
    float xv[V];
    if constexpr (VEC) {
      ld4(xp + (int64_t)ci * a.dhw, xv);
    } else {
      xv[0] = ld(xp + (int64_t)ci * a.dhw);
    }
    float wv[NCO];
#pragma unroll
    for (int j = 0; j < NCO; j += 4) {
      const float4 q = *reinterpret_cast<const float4*>(k1_w + ci * NCO + j);
      wv[j] = q.x; wv[j + 1] = q.y; wv[j + 2] = q.z; wv[j + 3] = q.w;
    }
#pragma unroll
    for (int j = 0; j < NCO; ++j)
#pragma unroll
      for (int k = 0; k < V; ++k) acc[j][k] = fmaf(wv[j], xv[k], acc[j][k]);
  }
  T* yp = static_cast<T*>(a.y) + b * a.y_bstride + p;
#pragma unroll
  for (int j = 0; j < NCO; ++j) {
    const int co = a.co0 + j;
    if (co >= a.Cout) break;
    const float sc = k1_bn[j], sh = k1_bn[NCO + j];
    float o[V];
#pragma unroll
    for (int k = 0; k < V; ++k) {
      float v = a.scale ? fmaf(acc[j][k], sc, sh) : acc[j][k];
      o[k] = a.relu ? fmaxf(v, 0.f) : v;
    }
    T* dst = yp + (int64_t)(a.y_ch0 + co) * a.dhw;
    if constexpr (VEC) st4(dst, o); else st(dst, o[0]);
  }
}

// Two 1x1x1 ConvBR_3d in a row as ONE launch (the head's last_12_3d -> the channel mix of last_6_3d, rag_model.py:358-365: the second
// one runs conv-first, in front of its upsample): y = act2(bn2(W2 . act1(bn1(W1 . x)))) per voxel, the intermediate in registers.  The
// same fmaf chains, in the same order, as two conv_k1 launches (under bf16 storage the intermediate is rounded as its store would),
// so the results are theirs bit for bit.  One thread per voxel: these volumes are tiny (level 12: 53 k voxels) and latency-bound.
template <class T, int CMID, int NCO>
__global__ __launch_bounds__(256) void conv_k1_chain_kernel(K1Args a, const float* __restrict__ w2, const float* __restrict__ scale2,
                                                            const float* __restrict__ shift2, int relu2) {
  extern __shared__ __attribute__((aligned(16))) float kc_w[];      // W1 as [ci][CMID] | W2 as [cm][NCO] | scale1, shift1 [CMID] | scale2, shift2 [NCO]
  float* const w2l = kc_w + a.Cin * CMID;
  float* const bn1 = w2l + CMID * NCO;
  float* const bn2 = bn1 + 2 * CMID;
  for (int i = threadIdx.x; i < a.Cin * CMID; i += 256) kc_w[i] = a.w[(i % CMID) * a.Cin + i / CMID];
  for (int i = threadIdx.x; i < CMID * NCO; i += 256) w2l[i] = w2[(i % NCO) * CMID + i / NCO];
  if (threadIdx.x < CMID) { bn1[threadIdx.x] = a.scale ? a.scale[threadIdx.x] : 1.f; bn1[CMID + threadIdx.x] = a.scale ? a.shift[threadIdx.x] : 0.f; }
  if (threadIdx.x < NCO) { bn2[threadIdx.x] = scale2 ? scale2[threadIdx.x] : 1.f; bn2[NCO + threadIdx.x] = scale2 ? shift2[threadIdx.x] : 0.f; }
  __syncthreads();
  const int64_t p = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (p >= a.dhw) return;
  const int b = blockIdx.y;
  const T* xp = static_cast<const T*>(a.x) + b * a.x_bstride + p;
  float h[CMID];
#pragma unroll
  for (int j = 0; j < CMID; ++j) h[j] = 0.f;
#pragma unroll 4
  for (int ci = 0; ci < a.Cin; ++ci) {
    const float xv = ld(xp + (int64_t)ci * a.dhw);
#pragma unroll
    for (int j = 0; j < CMID; j += 4) {
      const float4 q = *reinterpret_cast<const float4*>(kc_w + ci * CMID + j);
      h[j] = fmaf(q.x, xv, h[j]); h[j + 1] = fmaf(q.y, xv, h[j + 1]); h[j + 2] = fmaf(q.z, xv, h[j + 2]); h[j + 3] = fmaf(q.w, xv, h[j + 3]);
    }
  }
#pragma unroll
  for (int j = 0; j < CMID; ++j) {
    float v = a.scale ? fmaf(h[j], bn1[j], bn1[CMID + j]) : h[j];
    v = a.relu ? fmaxf(v, 0.f) : v;
    if constexpr (!std::is_same<T, float>::value) { const bf16_t r = to_bf16(v); v = ld(&r); }      // (the store's rounding)
    h[j] = v;
  }
  float acc[NCO];
#pragma unroll
  for (int k = 0; k < NCO; ++k) acc[k] = 0.f;
#pragma unroll
  for (int j = 0; j < CMID; ++j) {
#pragma unroll
    for (int k = 0; k < NCO; k += 4) {
      const float4 q = *reinterpret_cast<const float4*>(w2l + j * NCO + k);
      acc[k] = fmaf(q.x, h[j], acc[k]); acc[k + 1] = fmaf(q.y, h[j], acc[k + 1]); acc[k + 2] = fmaf(q.z, h[j], acc[k + 2]); acc[k + 3] = fmaf(q.w, h[j], acc[k + 3]);
    }
  }
  T* yp = static_cast<T*>(a.y) + b * a.y_bstride + p;
#pragma unroll
  for (int k = 0; k < NCO; ++k) {
    float v = scale2 ? fmaf(acc[k], bn2[k], bn2[NCO + k]) : acc[k];
    st(yp + (int64_t)(a.y_ch0 + k) * a.dhw, relu2 ? fmaxf(v, 0.f) : v);
  }
}

// Trilinear resample fused into the 1x1x1 ConvBR_3d that consumes it (Cell_3d: s1 = interpolate(prev) ->
// preprocess, rag_model.py:146-155; head: last_6_3d(upsample_12(.)), :358-365).  A thread owns one OUTPUT voxel:
// it computes its 3 (index, weight) pairs once, gathers the 8 taps of every input channel (for the x0.5 case each
// input voxel is read exactly once overall), interpolates in ATen's nesting order and feeds the channel mix.
// The interpolated tensor is never written to HBM.
struct __attribute__((packed, aligned(4))) K1RF2 { float a, b; };   // two neighbouring voxels, dword-aligned
struct K1RArgs {
  K1Args k;
  int Di, Hi, Wi, Do, Ho, Wo;
  float sd, sh, sw;
  int align;
};

// up to three such convs at the same output size (a cell's pre_preprocess and preprocess; round 4: also the NEXT cell's
// pre_preprocess when it resamples the same tensor to the same size — the second gather of that tensor then hits the L2 the first
// one filled) run as ONE launch: blockIdx.z = conv's first slab + output-channel slab
struct K1RPair {
  K1RArgs c[3];
  int splits[3];
};

template <class T, int NCO>
__global__ __launch_bounds__(256) void conv_k1_resample_kernel(K1RPair pr) {
  // this slab's weights, [ci][NCO], staged once per workgroup and read back as broadcast 16-byte LDS reads.  Read straight from
  // global memory they are NCO scalar loads per input channel, each behind its own branch and wait: on the small (level-12)
  // launches that scalar traffic, not the gathers, was the kernel
  extern __shared__ __attribute__((aligned(16))) float k1r_w[];
  const int which = (int)blockIdx.z >= pr.splits[0] + pr.splits[1] ? 2 : ((int)blockIdx.z >= pr.splits[0] ? 1 : 0);
  // (read-only views of the kernel arguments: a modified copy indexed by `which` would live in scratch memory)
  const K1RArgs& r = pr.c[which];
  const K1Args& a = r.k;
  const int64_t ovol = (int64_t)r.Do * r.Ho * r.Wo;
  const int co0 = ((int)blockIdx.z - (which == 2 ? pr.splits[0] + pr.splits[1] : (which ? pr.splits[0] : 0))) * NCO;
  for (int i = threadIdx.x; i < a.Cin * NCO; i += 256) {
    const int ci = i / NCO, co = co0 + i % NCO;
    k1r_w[i] = co < a.Cout ? a.w[co * a.w_sco + ci * a.w_sci] : 0.f;
  }
  // (the slab's folded BatchNorm beside the weights, as conv_k1_kernel: no per-channel global loads in the epilogue)
  float* const k1r_bn = k1r_w + a.Cin * NCO;               // scale[NCO] | shift[NCO]
  if (threadIdx.x < NCO) {
    const int co = min(co0 + (int)threadIdx.x, a.Cout - 1);
    k1r_bn[threadIdx.x] = a.scale ? a.scale[co] : 1.f;
    k1r_bn[NCO + threadIdx.x] = a.scale ? a.shift[co] : 0.f;
  }
  __syncthreads();
  const int64_t o = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (o >= ovol) return;
  const int b = blockIdx.y;
  const int ox = (int)(o % r.Wo);
  const int64_t t = o / r.Wo;
  const int oy = (int)(t % r.Ho), oz = (int)(t / r.Ho);
  const LinIdx lz = lin_index(oz, r.Di, r.Do, r.sd, r.align);
  const LinIdx ly = lin_index(oy, r.Hi, r.Ho, r.sh, r.align);
  const LinIdx lx = lin_index(ox, r.Wi, r.Wo, r.sw, r.align);
  const int64_t ivol = (int64_t)r.Di * r.Hi * r.Wi;
  const int r00 = (lz.i0 * r.Hi + ly.i0) * r.Wi, r01 = (lz.i0 * r.Hi + ly.i1) * r.Wi;
  const int r10 = (lz.i1 * r.Hi + ly.i0) * r.Wi, r11 = (lz.i1 * r.Hi + ly.i1) * r.Wi;
  const T* xp = static_cast<const T*>(a.x) + b * a.x_bstride;
  float acc[NCO];
#pragma unroll
  for (int j = 0; j < NCO; ++j) acc[j] = 0.f;
  // U channels per trip: 8U independent gathers in flight before the first use.  4 (32 loads, ~70 VGPRs) beats 8 (64 loads, 105
  // VGPRs) on the big HBM-bound launches — cell 3 at the headline shape: 78 us vs 91 — and 2 is no better; the 4-wide slab keeps 8
  constexpr int U = NCO <= 4 ? 8 : 4;
  if (r.Di == r.Do && r.Hi == r.Ho && r.Wi == r.Wo) {
    // an input already at the output size (the partner of a resampled one in a paired launch): index o, weight 1 — one load
    // per channel instead of eight, 16 channels in flight per trip; the same value the interpolation formula gives (1*v + 0*v')
    constexpr int UI = 16;
    for (int c0 = 0; c0 < a.Cin; c0 += UI) {
      float xv[UI];
#pragma unroll
      for (int u = 0; u < UI; ++u) xv[u] = ld(xp + (int64_t)min(c0 + u, a.Cin - 1) * ivol + o);
#pragma unroll
      for (int u = 0; u < UI; ++u) {
        if (c0 + u >= a.Cin) break;
        const int ci = c0 + u;
        float wv[NCO];
#pragma unroll
        for (int j = 0; j < NCO; j += 4) {
          const float4 q = *reinterpret_cast<const float4*>(k1r_w + ci * NCO + j);
          wv[j] = q.x; wv[j + 1] = q.y; wv[j + 2] = q.z; wv[j + 3] = q.w;
        }
#pragma unroll
        for (int j = 0; j < NCO; ++j) acc[j] = fmaf(wv[j], xv[u], acc[j]);
      }
    }
  } else {
    const bool pair = lx.i1 == lx.i0 + 1;
    for (int c0 = 0; c0 < a.Cin; c0 += U) {
      float tap[U][8];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const T* pc = xp + (int64_t)min(c0 + u, a.Cin - 1) * ivol;
        if constexpr (std::is_same<T, float>::value) {
          // the two x taps are neighbours (i1 = i0 + 1) except at the clamped right border: ONE 8-byte load per row instead of two
          // 4-byte loads with a stride of two voxels across the wave (x0.5: every cache line was requested twice, half used each
          // time).  dword alignment is all a multi-dword global load needs.  `pair` is wave-uniform almost everywhere.
          if (pair) {
            const K1RF2 q0 = *reinterpret_cast<const K1RF2*>(pc + r00 + lx.i0), q1 = *reinterpret_cast<const K1RF2*>(pc + r01 + lx.i0);
            const K1RF2 q2 = *reinterpret_cast<const K1RF2*>(pc + r10 + lx.i0), q3 = *reinterpret_cast<const K1RF2*>(pc + r11 + lx.i0);
            tap[u][0] = q0.a; tap[u][1] = q0.b; tap[u][2] = q1.a; tap[u][3] = q1.b;
            tap[u][4] = q2.a; tap[u][5] = q2.b; tap[u][6] = q3.a; tap[u][7] = q3.b;
            continue;
          }
        }
        tap[u][0] = ld(pc + r00 + lx.i0); tap[u][1] = ld(pc + r00 + lx.i1); tap[u][2] = ld(pc + r01 + lx.i0); tap[u][3] = ld(pc + r01 + lx.i1);
        tap[u][4] = ld(pc + r10 + lx.i0); tap[u][5] = ld(pc + r10 + lx.i1); tap[u][6] = ld(pc + r11 + lx.i0); tap[u][7] = ld(pc + r11 + lx.i1);
      }
#pragma unroll
      for (int u = 0; u < U; ++u) {
        if (c0 + u >= a.Cin) break;
        const int ci = c0 + u;
        const float a0 = lerp2(ly.w0, lerp2(lx.w0, tap[u][0], lx.w1, tap[u][1]), ly.w1, lerp2(lx.w0, tap[u][2], lx.w1, tap[u][3]));
        const float a1 = lerp2(ly.w0, lerp2(lx.w0, tap[u][4], lx.w1, tap[u][5]), ly.w1, lerp2(lx.w0, tap[u][6], lx.w1, tap[u][7]));
        const float xv = lerp2(lz.w0, a0, lz.w1, a1);
        float wv[NCO];
#pragma unroll
        for (int j = 0; j < NCO; j += 4) {
          const float4 q = *reinterpret_cast<const float4*>(k1r_w + ci * NCO + j);
          wv[j] = q.x; wv[j + 1] = q.y; wv[j + 2] = q.z; wv[j + 3] = q.w;
        }
#pragma unroll
        for (int j = 0; j < NCO; ++j) acc[j] = fmaf(wv[j], xv, acc[j]);
      }
    }
  }
  const int64_t yo = b * a.y_bstride + o;
  const bool yf32 = !std::is_same<T, float>::value && a.y_f32;
#pragma unroll
  for (int j = 0; j < NCO; ++j) {
    const int co = co0 + j;
    if (co >= a.Cout) break;
    float v = a.scale ? fmaf(acc[j], k1r_bn[j], k1r_bn[NCO + j]) : acc[j];
    v = a.relu ? fmaxf(v, 0.f) : v;
    const int64_t off = yo + (int64_t)(a.y_ch0 + co) * ovol;
    if (yf32) static_cast<float*>(a.y)[off] = v; else st(static_cast<T*>(a.y) + off, v);
  }
}

template <class T, int NCO>
static void launch_k1r_nco(const K1RArgs* r, int n, int B, hipStream_t s) {
  const int64_t ovol = (int64_t)r[0].Do * r[0].Ho * r[0].Wo;
  K1RPair pr{};
  int nz = 0, cin_max = 0;
  for (int i = 0; i < 3; ++i) {
    pr.c[i] = r[i < n ? i : n - 1];
    pr.splits[i] = i < n ? (int)ceil_div(r[i].k.Cout, NCO) : 0;
    nz += pr.splits[i];
    cin_max = std::max(cin_max, pr.c[i].k.Cin);
  }
  dim3 grid((unsigned)ceil_div(ovol, 256), B, (unsigned)nz);
  const size_t wlds = (size_t)(cin_max + 2) * NCO * sizeof(float);
  hipLaunchKernelGGL((conv_k1_resample_kernel<T, NCO>), grid, dim3(256), wlds, s, pr);
}

static int fill_k1r(K1RArgs& r, const void* x, int64_t x_bstride, int Di, int Hi, int Wi, const void* weight, const void* scale,
                    const void* shift, int relu, void* y, int64_t y_bstride, int y_ch0, int B, int Cin, int Cout, int Do, int Ho,
                    int Wo, int align_corners) {
  RAGMI_REQUIRE(x && weight && y, RAGMI_EINVAL, "conv3d_k1_resample: null pointer");
  RAGMI_REQUIRE((scale == nullptr) == (shift == nullptr), RAGMI_EINVAL, "conv3d_k1_resample: scale/shift must both be given or both NULL");
  RAGMI_REQUIRE(B > 0 && Cin > 0 && Cout > 0 && Di > 0 && Hi > 0 && Wi > 0 && Do > 0 && Ho > 0 && Wo > 0 && y_ch0 >= 0, RAGMI_EINVAL,
                "conv3d_k1_resample: bad size");
  RAGMI_REQUIRE(B <= 65535 && (int64_t)Di * Hi * Wi < (1ll << 31) && Cin <= 512, RAGMI_EUNSUPPORTED, "conv3d_k1_resample: size too large");
  r.k = K1Args{x, x_bstride, (const float*)weight, (const float*)scale, (const float*)shift,
               y, y_bstride, y_ch0, Cin, Cout, 0, (int64_t)Do * Ho * Wo, relu, Cin, 1};
  r.Di = Di; r.Hi = Hi; r.Wi = Wi; r.Do = Do; r.Ho = Ho; r.Wo = Wo;
  r.sd = lin_scale(Di, Do, align_corners); r.sh = lin_scale(Hi, Ho, align_corners); r.sw = lin_scale(Wi, Wo, align_corners);
  r.align = align_corners ? 1 : 0;
  return RAGMI_OK;
}

template <class T>
static int launch_k1r(const K1RArgs* r, int n, int B, hipStream_t s) {
  // slab width (output channels per thread; every slab gathers the input again).  One slab covering all outputs when the
  // launch has threads to spare or is HBM-bound (>= 128 MB of input: re-reading it costs more than the parallelism buys —
  // cell 4 at the headline shape: 25.6 us with 16 vs 30.2 with 12); otherwise the widest EVEN split that yields enough threads
  // (cell 6: 26.0 us with 8 vs 29.9 with 12, whose second slab carries 4 of 16 channels at the full gather cost)
  int cmax = 0;
  for (int i = 0; i < n; ++i) cmax = std::max(cmax, r[i].k.Cout);
  const int64_t threads = (int64_t)B * r[0].Do * r[0].Ho * r[0].Wo * n, want = 256 * 256 * 2;
  int64_t in_bytes = 0;
  for (int i = 0; i < n; ++i) in_bytes += (int64_t)B * r[i].k.Cin * r[i].Di * r[i].Hi * r[i].Wi * (int64_t)sizeof(T);
  static const int widths[5] = {24, 16, 12, 8, 4};
  int cover = 24;
  for (int w : widths)
    if (w >= cmax) cover = w;
  int nco = 4;
  if (threads >= want || in_bytes >= (128ll << 20)) {
    nco = cover;
  } else {
    for (int w : widths)
      if (w <= cover && (w == cover || cmax % w == 0) && threads * ceil_div(cmax, w) >= want) { nco = w; break; }
  }
  switch (nco) {
    case 24: launch_k1r_nco<T, 24>(r, n, B, s); break;
    case 16: launch_k1r_nco<T, 16>(r, n, B, s); break;
    case 12: launch_k1r_nco<T, 12>(r, n, B, s); break;
    case 8: launch_k1r_nco<T, 8>(r, n, B, s); break;
    default: launch_k1r_nco<T, 4>(r, n, B, s); break;
  }
  return check_launch("conv3d_k1_resample");
}

template <class T, bool VEC>
__global__ __launch_bounds__(256) void add_kernel(const T* __restrict__ a, int64_t a_bs, const T* __restrict__ b,
                                                  int64_t b_bs, T* __restrict__ y, int64_t y_bs, int64_t n) {
  constexpr int V = VEC ? 4 : 1;
  const int64_t p = ((int64_t)blockIdx.x * 256 + threadIdx.x) * V;
  if (p >= n) return;
  const int bi = blockIdx.y;
  if constexpr (VEC) {
    float u[4], v[4], o[4];
    ld4(a + bi * a_bs + p, u);
    ld4(b + bi * b_bs + p, v);
#pragma unroll
    for (int k = 0; k < 4; ++k) o[k] = u[k] + v[k];
    st4(y + bi * y_bs + p, o);
  } else {
    st(y + bi * y_bs + p, ld(a + bi * a_bs + p) + ld(b + bi * b_bs + p));
  }
}

template <class T, int NCO, bool VEC>
static void launch_k1_nco(const K1Args& a, int B, hipStream_t s) {
  constexpr int V = VEC ? 4 : 1;
  dim3 grid((unsigned)ceil_div(ceil_div(a.dhw, V), 256), B, (unsigned)ceil_div(a.Cout, NCO));
  hipLaunchKernelGGL((conv_k1_kernel<T, NCO, VEC>), grid, dim3(256), (size_t)(a.Cin + 2) * NCO * sizeof(float), s, a);
}

// Pick the widest output slab per thread that still leaves enough threads to fill the chip:
// large volumes read the input once (NCO = Cout); tiny ones (level-12: 53k voxels) are latency-
// bound, so they trade L2 re-reads of the input for parallelism (slabs of 4 on blockIdx.z).
template <class T, bool VEC>
static void launch_k1(const K1Args& a, int B, hipStream_t s) {
  constexpr int V = VEC ? 4 : 1;
  const int64_t threads = (int64_t)B * ceil_div(a.dhw, V);
  const int64_t want = 256 * 256 * 4;   // ~4 workgroups per CU (round 4 sweep at the headline shapes: 1.5 per CU +6 us over the small launches, 2 -> 4: -1 us)
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
    case 24: launch_k1_nco<T, 24, VEC>(a, B, s); break;
    case 16: launch_k1_nco<T, 16, VEC>(a, B, s); break;
    case 12: launch_k1_nco<T, 12, VEC>(a, B, s); break;
    case 8: launch_k1_nco<T, 8, VEC>(a, B, s); break;
    default: launch_k1_nco<T, 4, VEC>(a, B, s); break;
  }
}

}  // namespace ragmi

extern "C" int ragmi_conv3d_k1_fwd(const void* x, int64_t x_bstride, const void* weight, const void* scale,
                                   const void* shift, int relu, void* y, int64_t y_bstride, int y_ch0, int B, int Cin,
                                   int Cout, int64_t DHW, int dtype, void* stream) {
  return ragmi_conv3d_k1_fwd_ex(x, x_bstride, weight, 0, scale, shift, relu, y, y_bstride, y_ch0, B, Cin, Cout, DHW, dtype, stream);
}

extern "C" int ragmi_conv3d_k1_fwd_ex(const void* x, int64_t x_bstride, const void* weight, int w_transposed, const void* scale,
                                      const void* shift, int relu, void* y, int64_t y_bstride, int y_ch0, int B, int Cin,
                                      int Cout, int64_t DHW, int dtype, void* stream) {
  using namespace ragmi;
  RAGMI_REQUIRE(x && weight && y, RAGMI_EINVAL, "conv3d_k1: null pointer");
  RAGMI_REQUIRE((scale == nullptr) == (shift == nullptr), RAGMI_EINVAL, "conv3d_k1: scale/shift must both be given or both NULL");
  RAGMI_REQUIRE(B > 0 && Cin > 0 && Cout > 0 && DHW > 0 && y_ch0 >= 0, RAGMI_EINVAL, "conv3d_k1: bad size");
  RAGMI_REQUIRE(dtype_ok(dtype), RAGMI_EUNSUPPORTED, "conv3d_k1: dtype %d not built", dtype);
  RAGMI_REQUIRE(B <= 65535 && Cout <= 4 * 65535 && Cin <= 512, RAGMI_EUNSUPPORTED, "conv3d_k1: B, Cin or Cout too large");
  K1Args a{x, x_bstride, (const float*)weight, (const float*)scale, (const float*)shift,
           y, y_bstride, y_ch0, Cin, Cout, 0, DHW, relu, w_transposed ? 1 : Cin, w_transposed ? Cout : 1};
  // 16-B columns need alignment; small volumes use one voxel per thread for 4x the parallelism
  const bool vec = (DHW % 4 == 0) && (x_bstride % 4 == 0) && (y_bstride % 4 == 0) && aligned4(x, dtype) && aligned4(y, dtype) &&
                   (int64_t)B * DHW >= (1 << 18);
  hipStream_t s = static_cast<hipStream_t>(stream);
  if (dtype == RAGMI_BF16) {
    if (vec) launch_k1<bf16_t, true>(a, B, s); else launch_k1<bf16_t, false>(a, B, s);
  } else {
    if (vec) launch_k1<float, true>(a, B, s); else launch_k1<float, false>(a, B, s);
  }
  return check_launch("conv3d_k1");
}

extern "C" int ragmi_conv3d_k1_chain_supported(int Cin, int Cmid, int Cout) { return (Cin >= 1 && Cin <= 64 && Cmid == 24 && Cout == 12) ? 1 : 0; }

extern "C" int ragmi_conv3d_k1_chain_fwd(const void* x, int64_t x_bstride, const void* weight1, const void* scale1, const void* shift1, int relu1,
                                         int Cmid, const void* weight2, const void* scale2, const void* shift2, int relu2, void* y, int64_t y_bstride,
                                         int y_ch0, int B, int Cin, int Cout, int64_t DHW, int dtype, void* stream) {
  using namespace ragmi;
  RAGMI_REQUIRE(x && weight1 && weight2 && y, RAGMI_EINVAL, "conv3d_k1_chain: null pointer");
  RAGMI_REQUIRE((scale1 == nullptr) == (shift1 == nullptr) && (scale2 == nullptr) == (shift2 == nullptr), RAGMI_EINVAL,
                "conv3d_k1_chain: scale/shift must both be given or both NULL");
  RAGMI_REQUIRE(B > 0 && B <= 65535 && DHW > 0 && y_ch0 >= 0, RAGMI_EINVAL, "conv3d_k1_chain: bad size");
  RAGMI_REQUIRE(dtype_ok(dtype), RAGMI_EUNSUPPORTED, "conv3d_k1_chain: dtype %d not built", dtype);
  RAGMI_REQUIRE(ragmi_conv3d_k1_chain_supported(Cin, Cmid, Cout), RAGMI_EUNSUPPORTED,
                "conv3d_k1_chain: built for Cin <= 64 -> 24 -> 12 channels (the head's last_12_3d -> last_6_3d), got %d -> %d -> %d", Cin, Cmid, Cout);
  K1Args a{x, x_bstride, (const float*)weight1, (const float*)scale1, (const float*)shift1, y, y_bstride, y_ch0, Cin, Cmid, 0, DHW, relu1, Cin, 1};
  const dim3 grid((unsigned)ceil_div(DHW, 256), B);
  const size_t lds = (size_t)(Cin * 24 + 24 * 12 + 2 * 24 + 2 * 12) * sizeof(float);
  hipStream_t s = static_cast<hipStream_t>(stream);
  if (dtype == RAGMI_BF16)
    hipLaunchKernelGGL((conv_k1_chain_kernel<bf16_t, 24, 12>), grid, dim3(256), lds, s, a, (const float*)weight2, (const float*)scale2, (const float*)shift2, relu2);
  else
    hipLaunchKernelGGL((conv_k1_chain_kernel<float, 24, 12>), grid, dim3(256), lds, s, a, (const float*)weight2, (const float*)scale2, (const float*)shift2, relu2);
  return check_launch("conv3d_k1_chain");
}

extern "C" int ragmi_conv3d_k1_resample_fwd(const void* x, int64_t x_bstride, int Di, int Hi, int Wi, const void* weight,
                                            const void* scale, const void* shift, int relu, void* y, int64_t y_bstride,
                                            int y_ch0, int B, int Cin, int Cout, int Do, int Ho, int Wo, int align_corners,
                                            int dtype, void* stream) {
  using namespace ragmi;
  const bool out_f32 = dtype == (RAGMI_BF16 | RAGMI_OUT_F32);      // mixed storage: bf16 in, fp32 out (include/rag_amd.h)
  if (out_f32) dtype = RAGMI_BF16;
  RAGMI_REQUIRE(dtype_ok(dtype), RAGMI_EUNSUPPORTED, "conv3d_k1_resample: dtype %d not built", dtype);
  K1RArgs r{};
  const int rc = fill_k1r(r, x, x_bstride, Di, Hi, Wi, weight, scale, shift, relu, y, y_bstride, y_ch0, B, Cin, Cout, Do, Ho, Wo, align_corners);
  if (rc != RAGMI_OK) return rc;
  r.k.y_f32 = out_f32 ? 1 : 0;
  return dtype == RAGMI_BF16 ? launch_k1r<bf16_t>(&r, 1, B, static_cast<hipStream_t>(stream))
                             : launch_k1r<float>(&r, 1, B, static_cast<hipStream_t>(stream));
}

extern "C" int ragmi_conv3d_k1_resample_pair_fwd(const ragmi_k1r_t* a, const ragmi_k1r_t* b, void* y, int64_t y_bstride, int B,
                                                 int Do, int Ho, int Wo, int align_corners, int dtype, void* stream) {
  using namespace ragmi;
  RAGMI_REQUIRE(a && b, RAGMI_EINVAL, "conv3d_k1_resample_pair: null descriptor");
  RAGMI_REQUIRE(dtype_ok(dtype), RAGMI_EUNSUPPORTED, "conv3d_k1_resample_pair: dtype %d not built", dtype);
  K1RArgs r[2]{};
  const ragmi_k1r_t* d[2] = {a, b};
  for (int i = 0; i < 2; ++i) {
    const int rc = fill_k1r(r[i], d[i]->x, d[i]->x_bstride, d[i]->Di, d[i]->Hi, d[i]->Wi, d[i]->weight, d[i]->scale, d[i]->shift,
                            d[i]->relu, y, y_bstride, d[i]->y_ch0, B, d[i]->Cin, d[i]->Cout, Do, Ho, Wo, align_corners);
    if (rc != RAGMI_OK) return rc;
  }
  return dtype == RAGMI_BF16 ? launch_k1r<bf16_t>(r, 2, B, static_cast<hipStream_t>(stream))
                             : launch_k1r<float>(r, 2, B, static_cast<hipStream_t>(stream));
}

extern "C" int ragmi_conv3d_k1_resample_multi_fwd(const ragmi_k1r_t* const* specs, void* const* ys, const int64_t* y_bstrides, int n, int B,
                                                  int Do, int Ho, int Wo, int align_corners, int dtype, void* stream) {
  using namespace ragmi;
  RAGMI_REQUIRE(specs && ys && y_bstrides && n >= 1 && n <= 3, RAGMI_EINVAL, "conv3d_k1_resample_multi: 1..3 descriptors");
  RAGMI_REQUIRE(dtype_ok(dtype), RAGMI_EUNSUPPORTED, "conv3d_k1_resample_multi: dtype %d not built", dtype);
  K1RArgs r[3]{};
  for (int i = 0; i < n; ++i) {
    RAGMI_REQUIRE(specs[i], RAGMI_EINVAL, "conv3d_k1_resample_multi: null descriptor");
    const ragmi_k1r_t* d = specs[i];
    const int rc = fill_k1r(r[i], d->x, d->x_bstride, d->Di, d->Hi, d->Wi, d->weight, d->scale, d->shift, d->relu, ys[i], y_bstrides[i],
                            d->y_ch0, B, d->Cin, d->Cout, Do, Ho, Wo, align_corners);
    if (rc != RAGMI_OK) return rc;
  }
  return dtype == RAGMI_BF16 ? launch_k1r<bf16_t>(r, n, B, static_cast<hipStream_t>(stream))
                             : launch_k1r<float>(r, n, B, static_cast<hipStream_t>(stream));
}

extern "C" int ragmi_add_fwd(const void* a, int64_t a_bstride, int a_ch0, const void* b, int64_t b_bstride, int b_ch0,
                             void* y, int64_t y_bstride, int y_ch0, int B, int C, int64_t DHW, int dtype, void* stream) {
  using namespace ragmi;
  RAGMI_REQUIRE(a && b && y, RAGMI_EINVAL, "add: null pointer");
  RAGMI_REQUIRE(B > 0 && C > 0 && DHW > 0 && a_ch0 >= 0 && b_ch0 >= 0 && y_ch0 >= 0, RAGMI_EINVAL, "add: bad size");
  RAGMI_REQUIRE(dtype_ok(dtype), RAGMI_EUNSUPPORTED, "add: dtype %d not built", dtype);
  RAGMI_REQUIRE(B <= 65535, RAGMI_EUNSUPPORTED, "add: B too large");
  const size_t es = dtype_size(dtype);
  const char* ap = (const char*)a + (int64_t)a_ch0 * DHW * es;
  const char* bp = (const char*)b + (int64_t)b_ch0 * DHW * es;
  char* yp = (char*)y + (int64_t)y_ch0 * DHW * es;
  const int64_t n = (int64_t)C * DHW;
  const bool vec = (n % 4 == 0) && (DHW % 4 == 0) && (a_bstride % 4 == 0) && (b_bstride % 4 == 0) && (y_bstride % 4 == 0) &&
                   aligned4(ap, dtype) && aligned4(bp, dtype) && aligned4(yp, dtype);
  hipStream_t s = static_cast<hipStream_t>(stream);
  auto go = [&](auto tag) {
    using T = decltype(tag);
    if (vec)
      hipLaunchKernelGGL((add_kernel<T, true>), dim3((unsigned)ceil_div(n / 4, 256), B), dim3(256), 0, s, (const T*)ap, a_bstride,
                         (const T*)bp, b_bstride, (T*)yp, y_bstride, n);
    else
      hipLaunchKernelGGL((add_kernel<T, false>), dim3((unsigned)ceil_div(n, 256), B), dim3(256), 0, s, (const T*)ap, a_bstride,
                         (const T*)bp, b_bstride, (T*)yp, y_bstride, n);
  };
  if (dtype == RAGMI_BF16) go(bf16_t{}); else go(float{});
  return check_launch("add");
}
