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
    return lerp2(ly.w0, lerp2(lx.w0, ld(p + o00), lx.w1, ld(p + o01)), ly.w1, lerp2(lx.w0, ld(p + o10), lx.w1, ld(p + o11)));
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

// The reference's own shape — maxdisp = 3 d and a 3x spatial upsampling (rag_model.py:40, 272-273) — as a tiled, two-pass kernel.
//  * A workgroup owns 8 x 32 output pixels; the coarse cost block they interpolate from (d planes x 5 x 13 values) is staged through
//    LDS once: per pixel that replaces 4 d scattered global loads (every plane is a different cache line, the L1 misses them all)
//    by 4 d LDS reads.
//  * Pass 1 takes the minimum over the d bilinear plane samples: every fine sample is a convex combination of two neighbouring
//    planes, so -min is the softmin's largest exponent (up to rounding), which is all the max-subtraction needs.  Pass 2 walks the
//    planes again with a window (previous, current, next) and evaluates the three fine samples of coarse index k from it: lerp,
//    difference to the minimum, one exp2, two accumulations — 8 VALU instructions per sample instead of the ~14 of the online form
//    (compare / select / rescale), and no serial dependence on a running maximum.
//  * Tap table in LDS, built with the SAME lin_index arithmetic as the generic kernel: entry dd holds the weights of the three
//    window planes (one of them zero), so fp32 source indices that land a hair below an integer (i0 = k - 1, lambda ~ 1) are
//    reproduced exactly.  fmaf(wp, vp, fmaf(wc, vc, wn * vn)) with one zero weight is bit-identical to the generic kernel's
//    fmaf(w0, b0, w1 * b1).
constexpr int DX3_TY = 8, DX3_TX = 32, DX3_CR = 5, DX3_CC = 13, DX3_CP = DX3_CR * DX3_CC;

template <class T>
__global__ __launch_bounds__(256) void disp_softargmin_x3_kernel(DispArgs a) {
  extern __shared__ __attribute__((aligned(16))) float dx3_lds[];
  float4* const ztab = reinterpret_cast<float4*>(dx3_lds);            // [maxdisp] (wp, wc, wn, dd)
  float* const tile = dx3_lds + 4 * a.maxdisp;                        // [d][5][13]
  const int tid = threadIdx.x, tx = tid % DX3_TX, ty = tid / DX3_TX;
  // XCD-aware tile order: workgroup j runs on XCD j % 8 (round-robin dispatch); every XCD walks one contiguous chunk of the
  // x-fastest tile list, so tiles that share coarse rows / columns of the cost meet in ONE L2.  With blockIdx.x = tile x the eight
  // neighbours of a tile sat on eight different XCDs and every L2 fetched its own copy of the shared halo: 77 MB from the memory
  // side for a 13.6 MB tensor (profiles/r04y_pmc_summary.txt).
  const int ntx = (a.Wo + DX3_TX - 1) / DX3_TX, nty = (a.Ho + DX3_TY - 1) / DX3_TY, ntile = ntx * nty;
  const int chunk = (ntile + 7) / 8, jb = blockIdx.x;
  const int tidx = (jb & 7) * chunk + (jb >> 3);
  if ((jb >> 3) >= chunk || tidx >= ntile) return;
  const int ox0 = (tidx % ntx) * DX3_TX, oy0 = (tidx / ntx) * DX3_TY, b = blockIdx.z;
  const int ox = min(ox0 + tx, a.Wo - 1), oy = min(oy0 + ty, a.Ho - 1);       // out-of-range threads shadow the last pixel, store nothing
  const int cy0 = lin_index(oy0, a.h, a.Ho, a.sh, 0).i0, cx0 = lin_index(ox0, a.w, a.Wo, a.sw, 0).i0;
  const int hw = a.h * a.w, D = a.d;
  const T* base = static_cast<const T*>(a.cost) + (int64_t)b * D * hw;
  for (int i = tid; i < D * DX3_CP; i += 256) {
    const int z = i / DX3_CP, r = i % DX3_CP;
    const int gy = min(cy0 + r / DX3_CC, a.h - 1), gx = min(cx0 + r % DX3_CC, a.w - 1);
    tile[i] = ld(base + (int64_t)z * hw + gy * a.w + gx);
  }
  for (int dd = tid; dd < a.maxdisp; dd += 256) {
    const LinIdx lz = lin_index(dd, D, a.maxdisp, a.sd, 0);
    const int k = dd / 3;
    float4 e = make_float4(0.f, 0.f, 0.f, (float)dd);
    if (lz.i0 < k) { e.x = lz.w0; e.y = lz.w1; }                      // planes (k-1, k)
    else if (lz.i1 != lz.i0) { e.y = lz.w0; e.z = lz.w1; }            // planes (k, k+1)
    else e.y = lz.w0 + lz.w1;                                         // clamped at the last plane
    ztab[dd] = e;
  }
  const LinIdx ly = lin_index(oy, a.h, a.Ho, a.sh, 0);
  const LinIdx lx = lin_index(ox, a.w, a.Wo, a.sw, 0);
  // the four corners' LDS addresses: a plane is then four reads at a common (compile-time, once unrolled) offset
  const float* const t00 = tile + (ly.i0 - cy0) * DX3_CC + (lx.i0 - cx0);
  const float* const t01 = t00 + (lx.i1 - lx.i0);
  const float* const t10 = t00 + (ly.i1 - ly.i0) * DX3_CC;
  const float* const t11 = t10 + (lx.i1 - lx.i0);
  __syncthreads();
  auto plane = [&](int z) {                  // bilinear sample of coarse plane z at (oy, ox); x innermost like ATen
    const int o = z * DX3_CP;
    return lerp2(ly.w0, lerp2(lx.w0, t00[o], lx.w1, t01[o]), ly.w1, lerp2(lx.w0, t10[o], lx.w1, t11[o]));
  };
  constexpr float K = 1.4426950408889634f;
  float s = 0.f, ws = 0.f;
  // d = 64 (maxdisp 192, the reference's only configuration, rag_model.py:274): the 64 plane samples of pass 1 stay in REGISTERS for
  // pass 2 — no second round of 4 d LDS reads and bilinear arithmetic (same operations on the same values: identical bits).  Chunks
  // of 8 planes fenced by sched_barrier: unfenced, the scheduler hoisted all 256 operand reads and the kernel spilled or ran at
  // 87 us (round 2).  Same box, both builds: 49.9 -> 40.3 us (90 VGPRs, 5 waves per SIMD).
  if (D == 64) {
    float v[64];
    float lo = INFINITY;
#pragma unroll
    for (int z = 0; z < 64; ++z) {
      v[z] = plane(z);
      lo = fminf(lo, v[z]);
      if ((z & 7) == 7) __builtin_amdgcn_sched_barrier(0);
    }
#pragma unroll
    for (int k = 0; k < 64; ++k) {
      const float vp_ = v[k > 0 ? k - 1 : 0], vc_ = v[k], vn_ = v[k < 63 ? k + 1 : 63];
#pragma unroll
      for (int j = 0; j < 3; ++j) {
        const float4 tb = ztab[3 * k + j];
        const float c = fmaf(tb.x, vp_, fmaf(tb.y, vc_, tb.z * vn_));
        const float e = __builtin_amdgcn_exp2f((lo - c) * K);
        s += e;
        ws = fmaf(e, tb.w, ws);
      }
      if ((k & 7) == 7) __builtin_amdgcn_sched_barrier(0);
    }
  } else {
  float lo = INFINITY;
#pragma clang loop unroll_count(4) vectorize(disable)
  for (int z = 0; z < D; ++z) lo = fminf(lo, plane(z));
  // exponent of a sample: (min - cost) * K: the difference is formed BEFORE the scaling by K, like the generic kernel, so large
  // |cost| does not lose the bits that matter near the minimum
  float vc = plane(0), vp = vc, vn = plane(min(1, D - 1));
  for (int k = 0; k < D; ++k) {
#pragma unroll
    for (int j = 0; j < 3; ++j) {
      const float4 tb = ztab[3 * k + j];     // wave-uniform address: one broadcast read
      const float c = fmaf(tb.x, vp, fmaf(tb.y, vc, tb.z * vn));
      const float e = __builtin_amdgcn_exp2f((lo - c) * K);
      s += e;
      ws = fmaf(e, tb.w, ws);
    }
    vp = vc; vc = vn; vn = plane(min(k + 2, D - 1));
  }
  }
  if (ox0 + tx < a.Wo && oy0 + ty < a.Ho) a.out[(int64_t)b * a.Ho * a.Wo + (int64_t)oy * a.Wo + ox] = ws / s;
}

// The wavefront-reduction form north_star names, for the same configuration (maxdisp = 3 d, 3x upsampling): 16 lanes share one
// output pixel, lane l owns the coarse planes 4l .. 4l+3 (d <= 64) and their 12 fine samples; a wave step is 4 pixels.
//  * per step a lane takes 4 bilinear plane samples (16 LDS reads — the thread-per-pixel form makes 2 x 4 d = 512 per pixel, plus a
//    broadcast tap-table read per fine sample) and gets the two neighbouring planes' samples from the lanes beside it (DPP row shifts);
//  * its 12 (wp, wc, wn, dd) tap entries are per-lane constants of the whole launch (registers), built with the same lin_index
//    arithmetic as the tap table of the tiled kernel;
//  * minimum, sum of exponentials and weighted sum are reduced over the 16 lanes of a DPP row (4 steps each, no LDS).
// Same reference exponent (the minimum over the coarse plane samples) and the same per-sample arithmetic as the tiled kernel; only
// the order of the two final sums differs (per lane, then across lanes).
__device__ __forceinline__ float dpp_row_sum(float v) {
  v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0xB1, 0xF, 0xF, false));     // quad_perm [1,0,3,2]
  v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x4E, 0xF, 0xF, false));     // quad_perm [2,3,0,1]
  v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x141, 0xF, 0xF, false));    // row_half_mirror
  v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x140, 0xF, 0xF, false));    // row_mirror
  return v;
}
__device__ __forceinline__ float dpp_row_min(float v) {
  v = fminf(v, __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0xB1, 0xF, 0xF, false)));
  v = fminf(v, __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x4E, 0xF, 0xF, false)));
  v = fminf(v, __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x141, 0xF, 0xF, false)));
  v = fminf(v, __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x140, 0xF, 0xF, false)));
  return v;
}

template <class T>
__global__ __launch_bounds__(256) void disp_softargmin_wave_kernel(DispArgs a) {
  extern __shared__ __attribute__((aligned(16))) float dxw_lds[];
  float* const tile = dxw_lds;                                         // [d][5][13]: plane stride 65 = 1 mod 32
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l = lane & 15, pq = lane >> 4;
  const int ox0 = blockIdx.x * DX3_TX, oy0 = blockIdx.y * DX3_TY, b = blockIdx.z;
  const int cy0 = lin_index(oy0, a.h, a.Ho, a.sh, 0).i0, cx0 = lin_index(ox0, a.w, a.Wo, a.sw, 0).i0;
  const int hw = a.h * a.w, D = a.d;
  const T* base = static_cast<const T*>(a.cost) + (int64_t)b * D * hw;
  for (int i = tid; i < D * DX3_CP; i += 256) {
    const int z = i / DX3_CP, r = i % DX3_CP;
    const int gy = min(cy0 + r / DX3_CC, a.h - 1), gx = min(cx0 + r % DX3_CC, a.w - 1);
    tile[i] = ld(base + (int64_t)z * hw + gy * a.w + gx);
  }
  // lane l owns the coarse planes k = l + 16 q (q = 0..3): the 16 lanes of a pixel read 16 CONSECUTIVE planes at a time, one bank
  // each.  Its 12 fine samples dd = 3 k + j: weights of the window (plane k-1, k, k+1) — per-lane constants of the launch
  float wp[12], wc[12], wn[12], fd[12];
#pragma unroll
  for (int i = 0; i < 12; ++i) {
    const int k = l + 16 * (i / 3), dd = 3 * k + i % 3;
    const LinIdx lz = lin_index(min(dd, a.maxdisp - 1), D, a.maxdisp, a.sd, 0);
    wp[i] = wc[i] = wn[i] = 0.f;
    fd[i] = (float)dd;
    if (k < D && dd < a.maxdisp) {
      if (lz.i0 < k) { wp[i] = lz.w0; wc[i] = lz.w1; }
      else if (lz.i1 != lz.i0) { wc[i] = lz.w0; wn[i] = lz.w1; }
      else wc[i] = lz.w0 + lz.w1;
    }
  }
  __syncthreads();
  constexpr float K = 1.4426950408889634f;
  for (int step = 0; step < 16; ++step) {
    const int idx = step * 4 + pq, ty = wave * 2 + (idx >> 5), tx = idx & 31;
    const int ox = min(ox0 + tx, a.Wo - 1), oy = min(oy0 + ty, a.Ho - 1);
    const LinIdx ly = lin_index(oy, a.h, a.Ho, a.sh, 0);
    const LinIdx lx = lin_index(ox, a.w, a.Wo, a.sw, 0);
    const float* const t00 = tile + (ly.i0 - cy0) * DX3_CC + (lx.i0 - cx0);
    const int o01 = lx.i1 - lx.i0, o10 = (ly.i1 - ly.i0) * DX3_CC;
    float v[4], vm[4], vx[4];
    float lo = INFINITY;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int k = l + 16 * q;
      const float* const t = t00 + min(k, D - 1) * DX3_CP;
      v[q] = lerp2(ly.w0, lerp2(lx.w0, t[0], lx.w1, t[o01]), ly.w1, lerp2(lx.w0, t[o10], lx.w1, t[o10 + o01]));
      lo = k < D ? fminf(lo, v[q]) : lo;
    }
    // plane k-1 / k+1: the lane before / after in the pixel's row, same q — across the row's ends it is q -/+ 1 of the lane at the other end
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      vm[q] = __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v[q]), 0x121, 0xF, 0xF, false));   // row_ror:1: lane l <- lane l-1 (0 <- 15)
      vx[q] = __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v[q]), 0x12F, 0xF, 0xF, false));   // row_ror:15: lane l <- lane l+1 (15 <- 0)
    }
    lo = dpp_row_min(lo);
    float s = 0.f, ws = 0.f;
#pragma unroll
    for (int i = 0; i < 12; ++i) {
      const int q = i / 3;
      const float vp_ = l == 0 ? (q == 0 ? v[0] : vm[q - 1]) : vm[q];         // (plane -1 does not exist: its weight is 0)
      const float vn_ = l == 15 ? (q == 3 ? v[3] : vx[q + 1]) : vx[q];
      const float c = fmaf(wp[i], vp_, fmaf(wc[i], v[q], wn[i] * vn_));
      float e = __builtin_amdgcn_exp2f((lo - c) * K);
      e = (wp[i] + wc[i] + wn[i]) > 0.f ? e : 0.f;                     // samples past maxdisp / the volume (d < 64)
      s += e;
      ws = fmaf(e, fd[i], ws);
    }
    s = dpp_row_sum(s);
    ws = dpp_row_sum(ws);
    if (l == 0 && ox0 + tx < a.Wo && oy0 + ty < a.Ho) a.out[(int64_t)b * a.Ho * a.Wo + (int64_t)oy * a.Wo + ox] = ws / s;
  }
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
  const size_t lds3 = lds + (size_t)d * DX3_CP * sizeof(float);
  if (maxdisp == 3 * d && Ho == 3 * h && Wo == 3 * w && lds3 <= 64 * 1024) {   // the reference's configuration (rag_model.py:40, 272-273): the tiled form
    const dim3 g3((unsigned)ceil_div(Wo, DX3_TX), (unsigned)ceil_div(Ho, DX3_TY), (unsigned)B);
    const dim3 g1((unsigned)(ceil_div((int64_t)g3.x * g3.y, 8) * 8), 1, (unsigned)B);      // the tiled kernel decodes its tile itself (XCD-aware)
#ifdef RAGMI_DISP_WAVE      // A/B build (tools/build_variant.sh): the wavefront-reduction form, d <= 64
    if (d <= 64) {
      const size_t ldsw = (size_t)d * DX3_CP * sizeof(float);
      if (dtype == RAGMI_BF16) hipLaunchKernelGGL(disp_softargmin_wave_kernel<bf16_t>, g3, dim3(256), ldsw, static_cast<hipStream_t>(stream), a);
      else hipLaunchKernelGGL(disp_softargmin_wave_kernel<float>, g3, dim3(256), ldsw, static_cast<hipStream_t>(stream), a);
      return check_launch("disp_softargmin");
    }
#endif
    if (dtype == RAGMI_BF16) hipLaunchKernelGGL(disp_softargmin_x3_kernel<bf16_t>, g1, dim3(256), lds3, static_cast<hipStream_t>(stream), a);
    else hipLaunchKernelGGL(disp_softargmin_x3_kernel<float>, g1, dim3(256), lds3, static_cast<hipStream_t>(stream), a);
    return check_launch("disp_softargmin");
  }
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
