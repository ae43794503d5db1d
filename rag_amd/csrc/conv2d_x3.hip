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

template <int NCGS, int NSET>
__global__ __launch_bounds__(C2_THREADS) void conv2d_x3_kernel(K3Args a, X3Extra e) {
  constexpr int NCG = NCGS * NSET;
  constexpr int NSLS = (NCGS * 27 + 7) / 8;                             // K-slices per set as packed
  constexpr int S0 = (9 * NCGS) / 8, S1 = (18 * NCGS - 1) / 8, NSU = S1 - S0 + 1;     // the slices that hold taps 9..17
  constexpr int NREC = NCG * C2_HY * C2_HX, NPF = (NREC + C2_THREADS - 1) / C2_THREADS;
  __shared__ __attribute__((aligned(16))) uint2 lhi[NCG * C2_GS + 1], llo[NCG * C2_GS + 1];
  __shared__ unsigned lmax[2];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, n = lane & 15, kb = lane >> 4;
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
  // weight fragments of the issued slices -> registers
  uint4 ah[NSET][NSU], al[NSET][NSU];
#pragma unroll
  for (int st = 0; st < NSET; ++st) {
    const uint4* const wf = e.wf[st] + (int64_t)cog * NSLS * 2 * 64;
#pragma unroll
    for (int s = 0; s < NSU; ++s) { ah[st][s] = wf[((S0 + s) * 2 + 0) * 64 + lane]; al[st][s] = wf[((S0 + s) * 2 + 1) * 64 + lane]; }
  }
  // operand record offsets of this lane quarter within a set: slice, pair -> (tap, group) -> group * GS + dy * RS + dx;
  // pairs of the taps 8 / 18: the record of zeros (the set's first group is added per set below)
  int poff[NSU][2];
  bool pzero[NSU][2];
#pragma unroll
  for (int s = 0; s < NSU; ++s)
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int P = 8 * (S0 + s) + 2 * kb + j, tap = P / NCGS, cg = P % NCGS;
      pzero[s][j] = tap < 9 || tap > 17;
      poff[s][j] = pzero[s][j] ? 0 : cg * C2_GS + ((tap - 9) / 3) * C2_RS + (tap - 9) % 3;
    }
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

}  // namespace ragmi
