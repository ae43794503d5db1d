// Depth-1 volumes on the split-operand matrix-core form (RAGMI_F32X3, fp32 storage): the Feature Net's 3x3 ConvBR_2d and the dual
// launches of its Cell_2d (rag_model.py:285-323; operations_2d.py), which reach the library as 3x3x3 convolutions of [B, C, 1, H, W]
// volumes with the 2-D weight in the middle z-slice.  On a depth-1 volume the taps dz != 1 only ever meet zero padding, so the
// convolution IS its middle slice: of the 27-tap packed fragments (tap-major pairs, conv3d_x3_common.h) only the K-slices that
// hold taps 9..17 are issued, and the pairs of those slices that belong to taps 8 / 18 read a record of zeros.
// The fp32 matrix-core kernel spends 16-23 us per such launch at the headline shape (2 x 128 x 416 pixels, 0.3 GFLOP): a handful
// of workgroups walking 27 taps; here a launch is a few hundred small workgroups with one barrier pair each.
//   rows = 16 output channels, columns = 16 consecutive pixels, K = 8 pairs of (tap, 4-channel group) x 4 channels;
//   workgroup = 8 x 32 pixels of one sample and one block of 16 output channels: halo 10 x 34 of every input channel (both sets of
//   a dual launch) -> registers -> per-set largest |x| -> operand scale 2^-e -> FP16 hi / lo records [group][row][x] in LDS;
//   wave w owns rows 2w, 2w+1 (4 column tiles); the weight fragments of the issued slices sit in registers.
#include "conv3d_x3_common.h"

namespace ragmi {

constexpr int C2_TX = 32, C2_TY = 8, C2_HX = C2_TX + 2, C2_HY = C2_TY + 2, C2_THREADS = 256;
constexpr int C2_RS = C2_HX + 1;                       // records per halo row
constexpr int C2_GS = C2_HY * C2_RS + 2;               // records per channel group; ONE record of zeros sits behind the last group

// the slices of the packed fragments that hold taps 9..17 (the middle z-slice) of a set with NCGS channel groups
template <int NCGS> struct C2Slices {
  static constexpr int NSLS = (NCGS * 27 + 7) / 8;                                  // K-slices per set as packed
  static constexpr int S0 = (9 * NCGS) / 8, S1 = (18 * NCGS - 1) / 8, NSU = S1 - S0 + 1;
};

// weight fragments of the issued slices of this lane -> registers (issued before the staging so that they travel under it)
template <int NCGS, int NSET>
__device__ __forceinline__ void c2_load_weights(const X3Extra& e, int cog, int lane, uint4 (&ah)[NSET][C2Slices<NCGS>::NSU],
                                                uint4 (&al)[NSET][C2Slices<NCGS>::NSU]) {
  using S = C2Slices<NCGS>;
#pragma unroll
  for (int st = 0; st < NSET; ++st) {
    const uint4* const wf = e.wf[st] + (int64_t)cog * S::NSLS * 2 * 64;
#pragma unroll
    for (int s = 0; s < S::NSU; ++s) { ah[st][s] = wf[((S::S0 + s) * 2 + 0) * 64 + lane]; al[st][s] = wf[((S::S0 + s) * 2 + 1) * 64 + lane]; }
  }
}

// the products and the epilogue of one workgroup tile: the halo records of all sets are in LDS (scaled by mul[set]); wave w owns rows
// 2w, 2w+1 (4 column tiles of 16 pixels); out = sum over sets of act(bn_set(conv_set))
template <int NCGS, int NSET>
__device__ __forceinline__ void c2_products(const K3Args& a, const X3Extra& e, const uint2* lhi, const uint2* llo, const float (&mul)[NSET],
                                            const uint4 (&ah)[NSET][C2Slices<NCGS>::NSU], const uint4 (&al)[NSET][C2Slices<NCGS>::NSU],
                                            int cog, int b, int xb, int yb) {
  using S = C2Slices<NCGS>;
  constexpr int NCG = NCGS * NSET, NSU = S::NSU;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, n = lane & 15, kb = lane >> 4;
  const int64_t HW = (int64_t)a.H * a.W;
  // operand record offsets of this lane quarter within a set: slice, pair -> (tap, group) -> group * GS + dy * RS + dx;
  // pairs of the taps 8 / 18: the record of zeros
  int poff[NSU][2];
  bool pzero[NSU][2];
#pragma unroll
  for (int s = 0; s < NSU; ++s)
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int P = 8 * (S::S0 + s) + 2 * kb + j, tap = P / NCGS, cg = P % NCGS;
      pzero[s][j] = tap < 9 || tap > 17;
      poff[s][j] = pzero[s][j] ? 0 : cg * C2_GS + ((tap - 9) / 3) * C2_RS + (tap - 9) % 3;
    }
  // epilogue constants of this lane's four output channels (rows 4 kb + r of the block)
  const int g = cog * 4 + kb, ngroups = (a.Cout + 3) >> 2;
  float osc[NSET][4], osh[NSET][4];
#pragma unroll
  for (int st = 0; st < NSET; ++st)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int co = cog * 16 + 4 * kb + r;
      const bool okc = co < a.Cout;
      osc[st][r] = ((okc && a.scale[st]) ? a.scale[st][co] : 1.f) * (okc ? e.wmul[st][co] : 1.f) * (1.f / mul[st]);
      osh[st][r] = (okc && a.shift[st]) ? a.shift[st][co] : 0.f;
    }
  float act_floor = (a.relu & 1) ? 0.f : __builtin_nanf("");   // max(u, NaN) = u
  asm volatile("" : "+v"(act_floor));
  const int ych = g < ngroups ? a.y_ch[g < RAGMI_MAX_GROUPS ? g : 0] : 0;
  float* const yb_ = static_cast<float*>(a.y) + b * a.y_bstride + (int64_t)ych * HW;
#pragma unroll
  for (int t = 0; t < 4; ++t) {
    const int row = 2 * wave + (t >> 1), col = (t & 1) * 16 + n;      // tile t of this wave
    const int base = row * C2_RS + col;
    float v[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int st = 0; st < NSET; ++st) {
      f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int s = 0; s < NSU; ++s) {
        const int o0 = pzero[s][0] ? NCG * C2_GS : st * NCGS * C2_GS + base + poff[s][0];
        const int o1 = pzero[s][1] ? NCG * C2_GS : st * NCGS * C2_GS + base + poff[s][1];
        const uint2 h0 = lhi[o0], h1 = lhi[o1], l0 = llo[o0], l1 = llo[o1];
        const uint4 bh = make_uint4(h0.x, h0.y, h1.x, h1.y), bl = make_uint4(l0.x, l0.y, l1.x, l1.y);
        acc = x3_mma<false>(ah[st][s], bh, acc);
        acc = x3_mma<false>(ah[st][s], bl, acc);
        acc = x3_mma<false>(al[st][s], bh, acc);
      }
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float u = fmaxf(fmaf(acc[r], osc[st][r], osh[st][r]), act_floor);
        v[r] = st == 0 ? u : v[r] + u;
      }
    }
    const int y = yb + row, x = xb + col;
    if (y < a.H && x < a.W && g < ngroups) {
#pragma unroll
      for (int r = 0; r < 4; ++r) yb_[r * HW + (int64_t)y * a.W + x] = v[r];
    }
  }
}

template <int NCGS, int NSET>
__global__ __launch_bounds__(C2_THREADS) void conv2d_x3_kernel(K3Args a, X3Extra e) {
  constexpr int NCG = NCGS * NSET;
  constexpr int NREC = NCG * C2_HY * C2_HX, NPF = (NREC + C2_THREADS - 1) / C2_THREADS;
  __shared__ __attribute__((aligned(16))) uint2 lhi[NCG * C2_GS + 1], llo[NCG * C2_GS + 1];
  __shared__ unsigned lmax[2];
  const int tid = threadIdx.x, lane = tid & 63;
  const int ncog = (a.Cout + 15) >> 4;
  const int cog = blockIdx.z % ncog, b = blockIdx.z / ncog;
  const int xb = blockIdx.x * C2_TX, yb = blockIdx.y * C2_TY;
  const int64_t HW = (int64_t)a.H * a.W;
  const float* const src = static_cast<const float*>(a.x) + b * a.x_bstride;
  if (tid < 2) lmax[tid] = 0u;
  if (tid == 0) { lhi[NCG * C2_GS] = make_uint2(0u, 0u); llo[NCG * C2_GS] = make_uint2(0u, 0u); }
  // halo records of this thread: 4 channels of one (group, row, column); unconditional clamped loads, zeros substituted at the commit
  float pf[NPF][4];
  unsigned valid = 0;
#pragma unroll
  for (int p = 0; p < NPF; ++p) {
    const int el = p * C2_THREADS + tid, cg = el / (C2_HY * C2_HX), r = el % (C2_HY * C2_HX);
    const int gy = yb + r / C2_HX - 1, gx = xb + r % C2_HX - 1;
    const bool ok = el < NREC && (unsigned)gy < (unsigned)a.H && (unsigned)gx < (unsigned)a.W;
    valid |= (ok ? 1u : 0u) << p;
    const float* const pc = src + (int64_t)min(cg, NCG - 1) * 4 * HW + (int64_t)min(max(gy, 0), a.H - 1) * a.W + min(max(gx, 0), a.W - 1);
#pragma unroll
    for (int c = 0; c < 4; ++c) pf[p][c] = pc[c * HW];
  }
  uint4 ah[NSET][C2Slices<NCGS>::NSU], al[NSET][C2Slices<NCGS>::NSU];
  c2_load_weights<NCGS, NSET>(e, cog, lane, ah, al);
  // per-set operand scale from the exact maximum of the set's halo tile
  float m[NSET];
#pragma unroll
  for (int st = 0; st < NSET; ++st) m[st] = 0.f;
#pragma unroll
  for (int p = 0; p < NPF; ++p) {
    const int el = p * C2_THREADS + tid, cg = el / (C2_HY * C2_HX);
    const float mp = ((valid >> p) & 1u) ? x3_scalable_max4(pf[p][0], pf[p][1], pf[p][2], pf[p][3]) : 0.f;
#pragma unroll
    for (int st = 0; st < NSET; ++st) m[st] = fmaxf(m[st], (cg / NCGS == st) ? mp : 0.f);
  }
  __syncthreads();                                                   // lmax and the zero record are written
#pragma unroll
  for (int st = 0; st < NSET; ++st) {
    const float wm = x3_wave_max(m[st]);
    if (lane == 0) atomicMax(&lmax[st], __float_as_uint(wm));
  }
  __syncthreads();
  float mul[NSET];
#pragma unroll
  for (int st = 0; st < NSET; ++st) mul[st] = x3_pow2_scale(__uint_as_float(lmax[st]), 16384.f);
#pragma unroll
  for (int p = 0; p < NPF; ++p) {
    const int el = p * C2_THREADS + tid;
    if (el >= NREC) continue;
    const int cg = el / (C2_HY * C2_HX), r = el % (C2_HY * C2_HX);
    const bool ok = (valid >> p) & 1u;
    const float ml = (NSET == 2 && cg / NCGS == 1) ? mul[NSET - 1] : mul[0];
    unsigned l01, l23;
    const unsigned h01 = x3_split2h(ok ? pf[p][0] : 0.f, ok ? pf[p][1] : 0.f, ml, l01), h23 = x3_split2h(ok ? pf[p][2] : 0.f, ok ? pf[p][3] : 0.f, ml, l23);
    const int dst = cg * C2_GS + (r / C2_HX) * C2_RS + r % C2_HX;
    lhi[dst] = make_uint2(h01, h23);
    llo[dst] = make_uint2(l01, l23);
  }
  __syncthreads();
  c2_products<NCGS, NSET>(a, e, lhi, llo, mul, ah, al, cog, b, xb, yb);
}

// ---------------------------------------------------------------------------------------------------------------
// One launch per Cell_2d (rag_model.py:143-177 with the 2-D operations; Feature Net cells, all new states fed by a conv from each
// input): the two 1x1 ConvBR_2d in front of the cell — pre_preprocess / preprocess, each on the bilinear (align_corners=True)
// resample of its input to the cell's size — are computed in the STAGING of the dual 3x3 launch instead of as launches of their
// own (one to three per cell, 5-12 us each next to a 10 us convolution).  Per halo pixel of the tile and per set: the resampled
// input vector (ATen's index rule and nesting, common.h), the channel mix as an fmaf chain over the input channels in order, the
// folded BatchNorm and the ReLU — the arithmetic of conv_k1_resample_kernel — then the operand scale and split as above.
// The interpolated tensors and s0 / s1 are never written.
struct C2In {
  const float* x;          // [B, Cin, Hi, Wi]
  int64_t bstride;
  int Cin, Hi, Wi;
  const float* w;          // [C][Cin]
  const float* scale;      // folded BatchNorm, [C] (may be null: identity)
  const float* shift;
  int relu;
  float sh, sw;            // lin_scale(Hi, H, 1), lin_scale(Wi, W, 1)
};
constexpr int C2_MAX_CIN = 48;

template <int NCGS>
__global__ __launch_bounds__(C2_THREADS) void cell2d_x3_kernel(K3Args a, X3Extra e, C2In in0, C2In in1) {
  constexpr int NSET = 2, NCG = NCGS * NSET, C = 4 * NCGS;
  constexpr int NI = 3;                                             // item rounds per thread (below)
  __shared__ __attribute__((aligned(16))) uint2 lhi[NCG * C2_GS + 1], llo[NCG * C2_GS + 1];
  __shared__ __attribute__((aligned(16))) float lw[NSET][C2_MAX_CIN * C];      // [set][ci][C]: broadcast 16-byte reads
  __shared__ float lsc[NSET][2][C];
  __shared__ unsigned lmax[2];
  const int tid = threadIdx.x, lane = tid & 63;
  const int ncog = (a.Cout + 15) >> 4;
  const int cog = blockIdx.z % ncog, b = blockIdx.z / ncog;
  const int xb = blockIdx.x * C2_TX, yb = blockIdx.y * C2_TY;
  if (tid < 2) lmax[tid] = 0u;
  if (tid == 0) { lhi[NCG * C2_GS] = make_uint2(0u, 0u); llo[NCG * C2_GS] = make_uint2(0u, 0u); }
  // (compile-time set: `st ? in1 : in0` with a lane-dependent st is an ADDRESS select — every field then comes from the argument
  // segment by a vector load, the weights by a second, dependent one: six serial memory round trips in front of the staging)
#pragma unroll
  for (int st = 0; st < NSET; ++st) {
    const C2In& in = st ? in1 : in0;
    for (int i = tid; i < C2_MAX_CIN * C; i += C2_THREADS) {
      const int ci = i / C, j = i % C;
      lw[st][i] = ci < in.Cin ? in.w[j * in.Cin + ci] : 0.f;
    }
    if (tid < C) {
      lsc[st][0][tid] = in.scale ? in.scale[tid] : 1.f;
      lsc[st][1][tid] = in.scale ? in.shift[tid] : 0.f;
    }
  }
  uint4 ah[NSET][C2Slices<NCGS>::NSU], al[NSET][C2Slices<NCGS>::NSU];
  c2_load_weights<NCGS, NSET>(e, cog, lane, ah, al);
  __syncthreads();                                                   // the 1x1 weights, lmax and the zero record are written
  // items = (set, halo pixel): rounds 0 and 1 take pixels 0..255 of set 0 and of set 1 (the set — hence the input descriptor — is
  // compile time there), round 2 the remaining 84 pixels of both sets (threads 0..83 set 0, 84..167 set 1)
  static_assert(NI == 3 && C2_HY * C2_HX - C2_THREADS <= C2_THREADS / 2, "item rounds below assume 256 < halo pixels <= 384");
  constexpr int NREST = C2_HY * C2_HX - C2_THREADS;
  float sv[NI][C];
  float m[NSET] = {0.f, 0.f};
  const int item_set[NI] = {0, 1, tid < NREST ? 0 : 1};
  const int item_px[NI] = {tid, tid, C2_THREADS + (tid < NREST ? tid : tid - NREST)};
  const bool item_on[NI] = {true, true, tid < 2 * NREST};
  // All three items of a thread advance TOGETHER through the input channels, U channels x 4 taps of each in flight per trip: a thread's
  // staging is a chain of memory round trips and nothing else (first version, one item after the other, U = 4: 18 trips for 24
  // channels, 27-30 us per cell — slower than the separate launches; this form: 3 trips).  An input already at the cell's size is
  // read with ONE load per channel in the two full rounds (a workgroup-uniform test); the mixed last round takes the four-tap code
  // for both sets (lin_index gives such an input the pair (i, i) with weights (1, 0): the loads repeat an address).
  constexpr int U = 8;
  const float* xp[NI];
  int o4[NI][4], cin[NI], ivol[NI];
  float wy[NI][2], wx[NI][2];
  bool on[NI];
#pragma unroll
  for (int p = 0; p < NI; ++p) {
    const bool second = item_set[p] != 0;
    const int Hi = second ? in1.Hi : in0.Hi, Wi = second ? in1.Wi : in0.Wi;
    const int r = item_px[p], gy = yb + r / C2_HX - 1, gx = xb + r % C2_HX - 1;
    on[p] = item_on[p] && (unsigned)gy < (unsigned)a.H && (unsigned)gx < (unsigned)a.W;
    const LinIdx ly = lin_index(min(max(gy, 0), a.H - 1), Hi, a.H, second ? in1.sh : in0.sh, 1);
    const LinIdx lx = lin_index(min(max(gx, 0), a.W - 1), Wi, a.W, second ? in1.sw : in0.sw, 1);
    xp[p] = (second ? in1.x : in0.x) + b * (second ? in1.bstride : in0.bstride);
    cin[p] = second ? in1.Cin : in0.Cin;
    ivol[p] = Hi * Wi;
    o4[p][0] = ly.i0 * Wi + lx.i0; o4[p][1] = ly.i0 * Wi + lx.i1; o4[p][2] = ly.i1 * Wi + lx.i0; o4[p][3] = ly.i1 * Wi + lx.i1;
    wy[p][0] = ly.w0; wy[p][1] = ly.w1; wx[p][0] = lx.w0; wx[p][1] = lx.w1;
#pragma unroll
    for (int j = 0; j < C; ++j) sv[p][j] = 0.f;
  }
  const int cmax = max(in0.Cin, in1.Cin);
  const bool ident[NI] = {in0.Hi == a.H && in0.Wi == a.W, in1.Hi == a.H && in1.Wi == a.W, false};      // workgroup-uniform
  for (int c0 = 0; c0 < cmax; c0 += U) {
    float t[NI][U][4];
#pragma unroll
    for (int p = 0; p < NI; ++p)
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const float* const pc = xp[p] + (int64_t)min(c0 + u, cin[p] - 1) * ivol[p];
        t[p][u][0] = pc[o4[p][0]];
        if (!ident[p]) {
#pragma unroll
          for (int k = 1; k < 4; ++k) t[p][u][k] = pc[o4[p][k]];
        }
      }
#pragma unroll
    for (int p = 0; p < NI; ++p) {
      const float* const wv = lw[item_set[p]];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        if (c0 + u < cin[p]) {
          // x innermost, then y (ATen's nesting; the depth axis of these depth-1 volumes interpolates with weights (1, 0))
          const float xv = ident[p] ? t[p][u][0]
                                    : lerp2(wy[p][0], lerp2(wx[p][0], t[p][u][0], wx[p][1], t[p][u][1]), wy[p][1], lerp2(wx[p][0], t[p][u][2], wx[p][1], t[p][u][3]));
#pragma unroll
          for (int j = 0; j < C; j += 4) {
            const float4 w4 = *reinterpret_cast<const float4*>(wv + (c0 + u) * C + j);
            sv[p][j] = fmaf(w4.x, xv, sv[p][j]); sv[p][j + 1] = fmaf(w4.y, xv, sv[p][j + 1]);
            sv[p][j + 2] = fmaf(w4.z, xv, sv[p][j + 2]); sv[p][j + 3] = fmaf(w4.w, xv, sv[p][j + 3]);
          }
        }
      }
    }
  }
#pragma unroll
  for (int p = 0; p < NI; ++p) {
    const bool second = item_set[p] != 0;
    const int rl = second ? in1.relu : in0.relu;
    float mp = 0.f;
#pragma unroll
    for (int j = 0; j < C; ++j) {
      const float u = fmaf(sv[p][j], lsc[item_set[p]][0][j], lsc[item_set[p]][1][j]);
      sv[p][j] = on[p] ? (rl ? fmaxf(u, 0.f) : u) : 0.f;              // off: the 3x3 convolution's zero padding / no item
    }
#pragma unroll
    for (int j = 0; j < C; j += 4) mp = fmaxf(mp, x3_scalable_max4(sv[p][j], sv[p][j + 1], sv[p][j + 2], sv[p][j + 3]));
    if (second) m[1] = fmaxf(m[1], mp); else m[0] = fmaxf(m[0], mp);
  }
#pragma unroll
  for (int st = 0; st < NSET; ++st) {
    const float wm = x3_wave_max(m[st]);
    if (lane == 0) atomicMax(&lmax[st], __float_as_uint(wm));
  }
  __syncthreads();
  float mul[NSET];
#pragma unroll
  for (int st = 0; st < NSET; ++st) mul[st] = x3_pow2_scale(__uint_as_float(lmax[st]), 16384.f);
#pragma unroll
  for (int p = 0; p < NI; ++p) {
    if (!item_on[p]) continue;
    const int st = item_set[p], r = item_px[p];
    const float ml = st ? mul[1] : mul[0];
#pragma unroll
    for (int q = 0; q < NCGS; ++q) {
      unsigned l01, l23;
      const unsigned h01 = x3_split2h(sv[p][4 * q], sv[p][4 * q + 1], ml, l01), h23 = x3_split2h(sv[p][4 * q + 2], sv[p][4 * q + 3], ml, l23);
      const int dst = (st * NCGS + q) * C2_GS + (r / C2_HX) * C2_RS + r % C2_HX;
      lhi[dst] = make_uint2(h01, h23);
      llo[dst] = make_uint2(l01, l23);
    }
  }
  __syncthreads();
  c2_products<NCGS, NSET>(a, e, lhi, llo, mul, ah, al, cog, b, xb, yb);
}

bool x2d_eligible(const K3Args& a, int nset, int dtype) {
  if (dtype != RAGMI_F32X3 || a.D != 1 || a.res != nullptr || a.ntail > 0 || a.ndown > 0 || !a.store_main) return false;
  const int nc = a.nchunks[0];
  if (nc < 1 || nc > 4 || (nset == 2 && (a.nchunks[1] != nc || nc > 2)) || a.Cin != nset * nc * 4 || a.Cout % 4 != 0) return false;
  if (a.W < 16 || a.H < 2 || (int64_t)a.B * ((a.Cout + 15) / 16) > 65535 || (int64_t)a.Cin * a.H * a.W >= (1ll << 31)) return false;
  return true;
}

int x2d_launch(K3Args a, int nset, int dtype, hipStream_t st) {
  X3Extra e{};
  x3_weight_sections(e, a, nset, dtype);
  const dim3 grid((unsigned)ceil_div(a.W, C2_TX), (unsigned)ceil_div(a.H, C2_TY), (unsigned)(a.B * ((a.Cout + 15) / 16)));
  const int nc = a.nchunks[0];
#define RAGMI_C2(NCGS_, NSET_) hipLaunchKernelGGL((conv2d_x3_kernel<NCGS_, NSET_>), grid, dim3(C2_THREADS), 0, st, a, e)
  if (nset == 2) { if (nc == 1) RAGMI_C2(1, 2); else RAGMI_C2(2, 2); }
  else switch (nc) {
    case 1: RAGMI_C2(1, 1); break;
    case 2: RAGMI_C2(2, 1); break;
    case 3: RAGMI_C2(3, 1); break;
    default: RAGMI_C2(4, 1); break;
  }
#undef RAGMI_C2
  return check_launch("conv2d_x3");
}

bool cell2d_supported(int C, int Cin0, int Cin1, int Cout, int H, int W, int dtype) {
  return dtype == RAGMI_F32X3 && (C == 4 || C == 8) && Cin0 >= 1 && Cin1 >= 1 && Cin0 <= C2_MAX_CIN && Cin1 <= C2_MAX_CIN && Cout >= 4 &&
         Cout % 4 == 0 && (Cout + 3) / 4 <= RAGMI_MAX_GROUPS && W >= 16 && H >= 2;
}

int cell2d_launch(K3Args a, const C2In& in0, const C2In& in1, int dtype, hipStream_t st) {
  X3Extra e{};
  x3_weight_sections(e, a, 2, dtype);
  RAGMI_REQUIRE((int64_t)a.B * ((a.Cout + 15) / 16) <= 65535, RAGMI_EUNSUPPORTED, "cell2d: batch x output blocks too large");
  const dim3 grid((unsigned)ceil_div(a.W, C2_TX), (unsigned)ceil_div(a.H, C2_TY), (unsigned)(a.B * ((a.Cout + 15) / 16)));
  if (a.nchunks[0] == 1) hipLaunchKernelGGL((cell2d_x3_kernel<1>), grid, dim3(C2_THREADS), 0, st, a, e, in0, in1);
  else hipLaunchKernelGGL((cell2d_x3_kernel<2>), grid, dim3(C2_THREADS), 0, st, a, e, in0, in1);
  return check_launch("cell2d_x3");
}

}  // namespace ragmi

extern "C" int ragmi_cell2d_supported(int C, int Cin0, int Cin1, int Cout, int H, int W, int dtype) {
  return ragmi::cell2d_supported(C, Cin0, Cin1, Cout, H, W, dtype) ? 1 : 0;
}

extern "C" int ragmi_cell2d_fwd(const ragmi_cell2d_in_t* s0, const ragmi_cell2d_in_t* s1, int C, const void* packedA, const void* scaleA,
                                const void* shiftA, const void* packedB, const void* scaleB, const void* shiftB, int relu, void* y,
                                int64_t y_bstride, const int32_t* y_group_ch, int B, int Cout, int H, int W, int dtype, void* stream) {
  using namespace ragmi;
  RAGMI_REQUIRE(s0 && s1 && s0->x && s1->x && s0->weight && s1->weight && packedA && packedB && y, RAGMI_EINVAL, "cell2d: null pointer");
  RAGMI_REQUIRE((scaleA == nullptr) == (shiftA == nullptr) && (scaleB == nullptr) == (shiftB == nullptr) &&
                (s0->scale == nullptr) == (s0->shift == nullptr) && (s1->scale == nullptr) == (s1->shift == nullptr), RAGMI_EINVAL,
                "cell2d: scale/shift must both be given or both NULL");
  RAGMI_REQUIRE(B > 0 && H > 0 && W > 0 && s0->Hi > 0 && s0->Wi > 0 && s1->Hi > 0 && s1->Wi > 0, RAGMI_EINVAL, "cell2d: bad size");
  RAGMI_REQUIRE(cell2d_supported(C, s0->Cin, s1->Cin, Cout, H, W, dtype), RAGMI_EUNSUPPORTED,
                "cell2d: shape / dtype not built (ragmi_cell2d_supported)");
  RAGMI_REQUIRE((int64_t)s0->Cin * s0->Hi * s0->Wi < (1ll << 31) && (int64_t)s1->Cin * s1->Hi * s1->Wi < (1ll << 31) && (int64_t)Cout * H * W < (1ll << 31),
                RAGMI_EUNSUPPORTED, "cell2d: plane too large");
  K3Args a{};
  a.x = nullptr; a.y = y; a.y_bstride = y_bstride;
  a.B = B; a.Cin = 2 * C; a.Cout = Cout; a.D = 1; a.H = H; a.W = W; a.relu = relu ? 1 : 0;
  a.nchunks[0] = a.nchunks[1] = C / 4;
  a.wp[0] = (const float*)packedA; a.scale[0] = (const float*)scaleA; a.shift[0] = (const float*)shiftA;
  a.wp[1] = (const float*)packedB; a.scale[1] = (const float*)scaleB; a.shift[1] = (const float*)shiftB;
  a.store_main = 1;
  const int ng = (Cout + 3) / 4;
  for (int g = 0; g < ng; ++g) {
    a.y_ch[g] = y_group_ch ? y_group_ch[g] : 4 * g;
    RAGMI_REQUIRE(a.y_ch[g] >= 0, RAGMI_EINVAL, "cell2d: negative destination channel");
  }
  auto fill = [&](const ragmi_cell2d_in_t* s) {
    C2In in{};
    in.x = (const float*)s->x; in.bstride = s->x_bstride; in.Cin = s->Cin; in.Hi = s->Hi; in.Wi = s->Wi;
    in.w = (const float*)s->weight; in.scale = (const float*)s->scale; in.shift = (const float*)s->shift; in.relu = s->relu ? 1 : 0;
    in.sh = lin_scale(s->Hi, H, 1); in.sw = lin_scale(s->Wi, W, 1);
    return in;
  };
  return cell2d_launch(a, fill(s0), fill(s1), dtype, static_cast<hipStream_t>(stream));
}
