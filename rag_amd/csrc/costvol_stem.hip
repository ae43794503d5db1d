// Cost volume + first 3x3x3 ConvBR (stem3d0) without the cost volume.
//
// Reference: Network.forward builds cost[b, c, i, y, x] = L[b,c,y,x] and cost[b, C+c, i, y, x] = R[b,c,y,x-i] for
// x >= i, zero elsewhere (src/models/rag_model.py:375-383) and feeds it to stem3d0 = ConvBR_3d(2C, Cout, 3, 1, 1)
// (rag_model.py:234, 341; operations_3d.py:40-47).  The left half does not depend on the disparity plane i and the right
// half depends on x - i only, so the convolution over (i, y, x) collapses:
//
//   pre[co, i, y, x] = sum_{c,dy,dx} L[c, y+dy, x+dx] * (sum over the admissible dz of wL[co,c,dz,dy,dx])
//                    + sum_{c,dy,e } R[c, y+dy, (x-i)+e] * (sum over the admissible (dz,dx), dx-dz = e, of wR[co,c,dz,dy,dx])
//
// "admissible" = the tap lies inside the volume (0 <= i+dz < D, x+dx < W) and on the non-zero side of the cost volume
// (x+dx >= i+dz).  Which taps are admissible depends only on a handful of classes of the output position:
//   cls = [i == 0] + 2 [i == D-1]      (z border)          tc = clamp(x - i, -3, 2)   (distance to the x = i diagonal)
//   xr  = [x == W-1]                   (right border, matters for the right half only)
// so pre = A[cls, tc][co, y, x] + B[cls, xr][co, y, x - i] with A a 3x3 and B a 3x5 two-dimensional convolution of the
// feature maps with pre-summed weights — EXACT (same products, summed in a different order), 15 + 6 small planes instead
// of a 53 GFLOP 3-D convolution over a 327 MB tensor that is never materialised.  Three kernels:
//   costvol_stem_weights : the pre-summed weight variants (once per weight version)
//   costvol_stem_planes  : the variant planes (VALU, input tile + variant weights in LDS)
//   costvol_stem_combine : out = act(scale * (A + B) + shift) (+ fused consumer 1x1x1 tails), HBM-write-bound
#include "conv3d_x3_common.h"

namespace ragmi {

constexpr int CS_MAXC = 16;                 // feature channels C and output channels Cout supported by the register tiles
constexpr int CS_NCLS = 4, CS_NTC = 5;      // z-border classes; tc = -2..2 (tc = -3 contributes nothing)

// WA[cls][tcidx][c][dy][dx][co], WB[cls][xr][c][dy][e][co]  (co fastest: one broadcast LDS read feeds all outputs)
__global__ __launch_bounds__(256) void costvol_stem_weights_kernel(const float* __restrict__ w, float* __restrict__ wa,
                                                                   float* __restrict__ wb, int C, int Cout) {
  const int na = CS_NCLS * CS_NTC * C * 9 * Cout, nb = CS_NCLS * 2 * C * 15 * Cout;
  const int idx = blockIdx.x * 256 + threadIdx.x;
  if (idx < na) {
    int t = idx;
    const int co = t % Cout; t /= Cout;
    const int dx = t % 3 - 1; t /= 3;
    const int dy = t % 3; t /= 3;
    const int c = t % C; t /= C;
    const int tc = t % CS_NTC - 2, cls = t / CS_NTC;
    float s = 0.f;
    for (int dz = -1; dz <= 1; ++dz) {
      const bool ok = (dz >= 0 || !(cls & 1)) && (dz <= 0 || !(cls & 2)) && dz <= tc + dx;
      if (ok) s += w[(((int64_t)co * 2 * C + c) * 3 + (dz + 1)) * 9 + dy * 3 + (dx + 1)];
    }
    wa[idx] = s;
  } else if (idx < na + nb) {
    int t = idx - na;
    const int co = t % Cout; t /= Cout;
    const int e = t % 5 - 2; t /= 5;
    const int dy = t % 3; t /= 3;
    const int c = t % C; t /= C;
    const int xr = t % 2, cls = t / 2;
    float s = 0.f;
    for (int dz = -1; dz <= 1; ++dz) {
      const int dx = e + dz;
      const bool ok = dx >= -1 && dx <= 1 && (dz >= 0 || !(cls & 1)) && (dz <= 0 || !(cls & 2)) && (dx < 1 || !xr);
      if (ok) s += w[(((int64_t)co * 2 * C + C + c) * 3 + (dz + 1)) * 9 + dy * 3 + (dx + 1)];
    }
    wb[idx - na] = s;
  }
}

// one set of planes: out[co][y][xi] = sum_{c,dy,k} src[c][y+dy-1][x0 + xi + k - kh] * w[c][dy][k][co]
struct PlaneDesc {
  int w_off;      // float offset of the variant's weights in the weight buffer
  int out_off;    // float offset of the plane set in the workspace (per batch item: + b * ws_bstride)
  int width;      // plane width (xi = 0..width-1)
  int x0;         // source column of xi = 0
  int right;      // 0: left features, 3 taps (kh = 1);  1: right features, 5 taps (kh = 2)
  int variant;    // index of the variant's weight fragments / multipliers (matrix-core form)
};
constexpr int CS_MAXDESC = 32;
struct PlanesArgs {
  const void* left;
  const void* right;
  const float* wts;
  float* ws;
  int64_t ws_bstride;
  int C, Cout, H, W, ndesc;
  PlaneDesc d[CS_MAXDESC];
};
constexpr int CS_TX = 64, CS_TY = 8, CS_PX = 2;     // workgroup tile; pixels per thread along x (4 px: 87 us, 2 px: 53 us, 1 px: 57 us — the
                                                    // grid is small, so threads count for more than weight reuse per LDS read)
constexpr int CS_RS = CS_TX + 5;                     // LDS row stride = 5 mod 32: the 16 x 4 lanes of a wave spread 2 per bank
constexpr int CS_NT = (CS_TX / CS_PX) * CS_TY;       // threads per workgroup
constexpr int CS_CC = 4;                             // input channels staged in LDS at a time
static_assert(CS_RS % 32 == 5, "row stride must be 5 mod 32");
// COUT4 = Cout / 4 when Cout is a multiple of 4 (weights read as float4 broadcasts), 0 = any Cout (scalar reads)
template <class T, int COUT4>
__global__ __launch_bounds__(CS_NT) void costvol_stem_planes_kernel(PlanesArgs a) {
  constexpr int NCO = COUT4 ? COUT4 * 4 : CS_MAXC;
  extern __shared__ __attribute__((aligned(16))) float cs_lds[];   // weights [C][3][ntap][Cout] | tile [CS_CC][TY+2][RS]
  float* wl = cs_lds;
  float* tile = cs_lds + ((a.C * 15 * a.Cout + 3) & ~3);
  const PlaneDesc d = a.d[blockIdx.z % a.ndesc];
  const int b = blockIdx.z / a.ndesc;
  const int xb = blockIdx.x * CS_TX, yb = blockIdx.y * CS_TY;
  if (xb >= d.width) return;                                      // uniform: descriptors have different widths
  const int ntap = d.right ? 5 : 3, kh = d.right ? 2 : 1;
  const T* src = static_cast<const T*>(d.right ? a.right : a.left) + (int64_t)b * a.C * a.H * a.W;
  const int nw = a.C * 3 * ntap * a.Cout;
  for (int e = threadIdx.x; e < nw; e += CS_NT) wl[e] = a.wts[d.w_off + e];
  const int xx = (threadIdx.x % (CS_TX / CS_PX)) * CS_PX, yy = threadIdx.x / (CS_TX / CS_PX);
  float acc[CS_PX][NCO];
#pragma unroll
  for (int p = 0; p < CS_PX; ++p)
#pragma unroll
    for (int co = 0; co < NCO; ++co) acc[p][co] = 0.f;
  // input channels go through LDS CS_CC at a time: a small tile keeps many workgroups resident (the loop body is a chain
  // of LDS reads feeding FMAs, so occupancy is what hides its latency)
  auto body = [&](auto ntap_) {
    constexpr int NT = decltype(ntap_)::value;
    for (int c0 = 0; c0 < a.C; c0 += CS_CC) {
      __syncthreads();
      for (int e = threadIdx.x; e < CS_CC * (CS_TY + 2) * (CS_TX + 4); e += CS_NT) {
        const int tx = e % (CS_TX + 4), ty = (e / (CS_TX + 4)) % (CS_TY + 2), cl = e / ((CS_TX + 4) * (CS_TY + 2));
        const int gy = yb + ty - 1, gx = d.x0 + xb + tx - kh, c = c0 + cl;
        const bool ok = c < a.C && (unsigned)gy < (unsigned)a.H && (unsigned)gx < (unsigned)a.W;
        tile[(cl * (CS_TY + 2) + ty) * CS_RS + tx] = ok ? ld(src + ((int64_t)c * a.H + gy) * a.W + gx) : 0.f;
      }
      __syncthreads();
      const int nc = min(CS_CC, a.C - c0);
      for (int cl = 0; cl < nc; ++cl) {
        const int c = c0 + cl;
#pragma unroll
        for (int dy = 0; dy < 3; ++dy) {
          float v[CS_PX + NT - 1];
#pragma unroll
          for (int k = 0; k < CS_PX + NT - 1; ++k) v[k] = tile[(cl * (CS_TY + 2) + yy + dy) * CS_RS + xx + k];
#pragma unroll
          for (int k = 0; k < NT; ++k) {
            const float* wr = wl + ((c * 3 + dy) * NT + k) * a.Cout;
            if constexpr (COUT4 > 0) {
#pragma unroll
              for (int q = 0; q < COUT4; ++q) {
                const float4 w4 = *reinterpret_cast<const float4*>(wr + 4 * q);
#pragma unroll
                for (int p = 0; p < CS_PX; ++p) {
                  acc[p][4 * q + 0] = fmaf(w4.x, v[p + k], acc[p][4 * q + 0]);
                  acc[p][4 * q + 1] = fmaf(w4.y, v[p + k], acc[p][4 * q + 1]);
                  acc[p][4 * q + 2] = fmaf(w4.z, v[p + k], acc[p][4 * q + 2]);
                  acc[p][4 * q + 3] = fmaf(w4.w, v[p + k], acc[p][4 * q + 3]);
                }
              }
            } else {
#pragma unroll
              for (int co = 0; co < NCO; ++co)
                if (co < a.Cout) {
                  const float wv = wr[co];
#pragma unroll
                  for (int p = 0; p < CS_PX; ++p) acc[p][co] = fmaf(wv, v[p + k], acc[p][co]);
                }
            }
          }
        }
      }
    }
  };
  if (d.right) body(std::integral_constant<int, 5>{}); else body(std::integral_constant<int, 3>{});
  const int y = yb + yy;
  if (y >= a.H) return;
  // workspace layout (round 5): a plane set is [y][xi][Cout] — the channels of a pixel contiguous — so that the consumers (the combine
  // kernel, and stem3d1's staging when stem3d0 is expanded there: conv3d_x3.hip) fetch a pixel's channels with 16-byte loads
  float* out = a.ws + b * a.ws_bstride + d.out_off + ((int64_t)y * d.width + xb + xx) * a.Cout;
#pragma unroll
  for (int p = 0; p < CS_PX; ++p) {
    if (xb + xx + p >= d.width) continue;
    if constexpr (COUT4 > 0) {
#pragma unroll
      for (int q = 0; q < COUT4; ++q)
        *reinterpret_cast<float4*>(out + p * a.Cout + 4 * q) = make_float4(acc[p][4 * q], acc[p][4 * q + 1], acc[p][4 * q + 2], acc[p][4 * q + 3]);
    } else {
#pragma unroll
      for (int co = 0; co < NCO; ++co)
        if (co < a.Cout) out[p * a.Cout + co] = acc[p][co];
    }
  }
}

// ---------------------------------------------------------------------------------------------------------------
// The variant planes on the 16-bit matrix cores (RAGMI_F32X3: the split-operand contract of conv3d_x3.hip — fp32 operands as two
// power-of-two-scaled FP16 halves, hi*hi + hi*lo + lo*hi, fp32 accumulation).  The VALU kernel above spends 53 us on 1.5 GFLOP
// (one broadcast LDS read per 8 FMAs); as 16x16x32 products the same planes are ~0.4 M MFMAs, a few microseconds of matrix time.
//   rows = 16 output channels, columns = 16 consecutive plane columns xi, K = 8 pairs of (tap, 4-channel group) x 4 channels,
//   pairs tap-major: P = tap * (C / 4) + cg, tap = dy * ntap + k  (A planes: 9 taps, B planes: 15)
// Weight fragments per variant (packed once per weight version by costvol_stem_pack_kernel, after the fp32 variants):
//   wf[((v * NS + s) * 2 + hl) * 64 + lane] (uint4 = 8 halves of w * 2^k[co]),  wmul[v * 16 + co] = 2^-k[co]
// with v = cls * CS_NTC + tcidx (A, NS = CS_NSA slices) or CS_NCLS * CS_NTC + cls * 2 + xr (B, CS_NSB slices: slots of CS_NSB each).
constexpr int CS_NVA = CS_NCLS * CS_NTC, CS_NVB = CS_NCLS * 2;
__host__ __device__ constexpr int cs_nslices(int C, int ntap) { return ((C / 4) * 3 * ntap + 7) / 8; }
__host__ __device__ inline int64_t cs_frag_words(int C) {   // floats of all fragment slots + multipliers
  return ((int64_t)CS_NVA * cs_nslices(C, 3) + (int64_t)CS_NVB * cs_nslices(C, 5)) * 2 * 64 * 4 + (int64_t)(CS_NVA + CS_NVB) * 16;
}

__global__ __launch_bounds__(256) void costvol_stem_pack_kernel(const float* __restrict__ variants, float* __restrict__ frag, int C, int Cout) {
  const int v = blockIdx.x, isb = v >= CS_NVA ? 1 : 0, ntap = isb ? 5 : 3, ncg = C / 4;
  const int ns = cs_nslices(C, ntap);
  const int na = CS_NCLS * CS_NTC * C * 9 * Cout;
  const float* const w = isb ? variants + na + (int64_t)(v - CS_NVA) * C * 15 * Cout : variants + (int64_t)v * C * 9 * Cout;   // [c][dy][k][co]
  __shared__ unsigned rowmax[16];
  if (threadIdx.x < 16) rowmax[threadIdx.x] = 0u;
  __syncthreads();
  const int nw = C * 3 * ntap * Cout;
  for (int i = threadIdx.x; i < nw; i += 256) atomicMax(&rowmax[i % Cout], __float_as_uint(fabsf(w[i])));
  __syncthreads();
  const int nsa = cs_nslices(C, 3), nsb = cs_nslices(C, 5);
  uint4* const wf = reinterpret_cast<uint4*>(frag) + (isb ? ((int64_t)CS_NVA * nsa + (int64_t)(v - CS_NVA) * nsb) : (int64_t)v * nsa) * 2 * 64;
  float* const wmul = frag + ((int64_t)CS_NVA * nsa + (int64_t)CS_NVB * nsb) * 2 * 64 * 4 + v * 16;
  if (threadIdx.x < 16) wmul[threadIdx.x] = threadIdx.x < Cout ? 1.f / x3_row_mul(rowmax[threadIdx.x]) : 1.f;
  for (int idx = threadIdx.x; idx < ns * 64; idx += 256) {
    const int lane = idx & 63, sl = idx >> 6, co = lane & 15, kb = lane >> 4;
    const float mul = co < Cout ? x3_row_mul(rowmax[co]) : 1.f;
    unsigned short hi[8], lo[8];
    for (int j = 0; j < 8; ++j) {
      const int P = 8 * sl + 2 * kb + (j >> 2), tap = P / ncg, cg = P % ncg, c = 4 * cg + (j & 3);
      const float val = (tap < 3 * ntap && co < Cout) ? w[((c * 3 + tap / ntap) * ntap + tap % ntap) * Cout + co] * mul : 0.f;
      hi[j] = x3_f16_bits(val);
      lo[j] = x3_f16_bits(val - (float)__builtin_bit_cast(_Float16, hi[j]));
    }
    auto pk = [](const unsigned short* h) {
      return make_uint4(h[0] | ((unsigned)h[1] << 16), h[2] | ((unsigned)h[3] << 16), h[4] | ((unsigned)h[5] << 16), h[6] | ((unsigned)h[7] << 16));
    };
    wf[(sl * 2 + 0) * 64 + lane] = pk(hi);
    wf[(sl * 2 + 1) * 64 + lane] = pk(lo);
  }
}

// One workgroup (256 threads) = an 8 x 64 tile of one variant plane set: the (10 x 68) halo of all C feature channels goes
// HBM -> registers -> (largest |x| of the tile -> operand scale 2^-e) -> FP16 hi / lo records [cg][row][x][4 ch] in LDS; wave w then
// owns rows 2w, 2w+1 (8 column tiles of 16), its weight fragments (NS slices, hi + lo) sit in registers.
constexpr int CSM_TX = 64, CSM_TY = 8, CSM_HX = CSM_TX + 4, CSM_HY = CSM_TY + 2, CSM_THREADS = 256;
constexpr int CSM_RS = CSM_HX + 1;                         // record stride of a halo row (8-byte records)
template <class TF, int NCG, int NTAP>
__device__ __forceinline__ void costvol_stem_planes_mfma_body(const PlanesArgs& a, const PlaneDesc& d, const uint4* __restrict__ frag_a,
                                                              const uint4* __restrict__ frag_b, const float* __restrict__ wmul_all,
                                                              uint2* lhi, uint2* llo, unsigned& lmax) {
  constexpr int NS = (NCG * 3 * NTAP + 7) / 8, KH = NTAP == 5 ? 2 : 1;
  constexpr int NREC = NCG * CSM_HY * CSM_HX, NPF = (NREC + CSM_THREADS - 1) / CSM_THREADS;
  const int b = blockIdx.z / a.ndesc;
  const int xb = blockIdx.x * CSM_TX, yb = blockIdx.y * CSM_TY;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, n = lane & 15, kb = lane >> 4;
  const TF* const src = static_cast<const TF*>(d.right ? a.right : a.left) + (int64_t)b * a.C * a.H * a.W;      // TF: the features' storage type
  if (tid == 0) lmax = 0u;
  // halo records of this thread: 4 channels of one (row, column)
  float pf[NPF][4];
  unsigned valid = 0;
  const int64_t HW = (int64_t)a.H * a.W;
#pragma unroll
  for (int p = 0; p < NPF; ++p) {
    const int el = p * CSM_THREADS + tid, cg = el / (CSM_HY * CSM_HX), r = el % (CSM_HY * CSM_HX);
    const int ty = r / CSM_HX, tx = r % CSM_HX, gy = yb + ty - 1, gx = d.x0 + xb + tx - KH;
    const bool ok = el < NREC && (unsigned)gy < (unsigned)a.H && (unsigned)gx < (unsigned)a.W;
    valid |= (ok ? 1u : 0u) << p;
    const TF* const pc = src + (int64_t)min(cg, NCG - 1) * 4 * HW + (int64_t)min(max(gy, 0), a.H - 1) * a.W + min(max(gx, 0), a.W - 1);
#pragma unroll
    for (int c = 0; c < 4; ++c) pf[p][c] = ld(pc + c * HW);         // unconditional (clamped) loads; zeros substituted at the commit
  }
  // weight fragments of this variant -> registers (the same for every tile of the plane set)
  const int v = d.variant;
  const uint4* const wf = (NTAP == 5 ? frag_b + (int64_t)(v - CS_NVA) * NS * 2 * 64 : frag_a + (int64_t)v * NS * 2 * 64);
  uint4 ah[NS], al[NS];
#pragma unroll
  for (int s = 0; s < NS; ++s) { ah[s] = wf[(s * 2 + 0) * 64 + lane]; al[s] = wf[(s * 2 + 1) * 64 + lane]; }
  // operand record offsets of this lane quarter: slice s, pair j -> (tap, cg) -> (cg * HY + dy) * RS + k   (padding pairs: pair 0's)
  int poff[NS][2];
#pragma unroll
  for (int s = 0; s < NS; ++s)
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int P = 8 * s + 2 * kb + j, tap = P / NCG, cg = P % NCG;
      poff[s][j] = tap < 3 * NTAP ? (cg * CSM_HY + tap / NTAP) * CSM_RS + tap % NTAP : 0;
    }
  float m = 0.f;
#pragma unroll
  for (int p = 0; p < NPF; ++p) {
    const float mp = x3_scalable_max4(pf[p][0], pf[p][1], pf[p][2], pf[p][3]);
    m = fmaxf(m, ((valid >> p) & 1u) ? mp : 0.f);
  }
  m = x3_wave_max(m);
  __syncthreads();                                                   // lmax is zero
  if (lane == 0) atomicMax(&lmax, __float_as_uint(m));
  __syncthreads();
  const float mul = x3_pow2_scale(__uint_as_float(lmax), 16384.f);   // exact maximum: no headroom needed
#pragma unroll
  for (int p = 0; p < NPF; ++p) {
    const int el = p * CSM_THREADS + tid;
    if (el >= NREC) continue;
    const int cg = el / (CSM_HY * CSM_HX), r = el % (CSM_HY * CSM_HX);
    const bool ok = (valid >> p) & 1u;
    unsigned l01, l23;
    const unsigned h01 = x3_split2h(ok ? pf[p][0] : 0.f, ok ? pf[p][1] : 0.f, mul, l01), h23 = x3_split2h(ok ? pf[p][2] : 0.f, ok ? pf[p][3] : 0.f, mul, l23);
    const int dst = (cg * CSM_HY + r / CSM_HX) * CSM_RS + r % CSM_HX;
    lhi[dst] = make_uint2(h01, h23);
    llo[dst] = make_uint2(l01, l23);
  }
  __syncthreads();
  const float inv = 1.f / mul;
  float osc[4];
#pragma unroll
  for (int r = 0; r < 4; ++r) osc[r] = wmul_all[v * 16 + 4 * kb + r] * inv;
  float* const out = a.ws + b * a.ws_bstride + d.out_off;
#pragma unroll
  for (int t = 0; t < 8; ++t) {
    const int row = 2 * wave + (t >> 2), col = (t & 3) * 16 + n;      // tile t of this wave: row, 16 columns
    const int base = row * CSM_RS + col;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int s = 0; s < NS; ++s) {
      const uint2 h0 = lhi[base + poff[s][0]], h1 = lhi[base + poff[s][1]], l0 = llo[base + poff[s][0]], l1 = llo[base + poff[s][1]];
      const uint4 bh = make_uint4(h0.x, h0.y, h1.x, h1.y), bl = make_uint4(l0.x, l0.y, l1.x, l1.y);
      acc = x3_mma<false>(ah[s], bh, acc);
      acc = x3_mma<false>(ah[s], bl, acc);
      acc = x3_mma<false>(al[s], bh, acc);
    }
    const int y = yb + row, xi = xb + col;
    if (y < a.H && xi < d.width && 4 * kb < a.Cout) {      // [y][xi][Cout] (Cout a multiple of 4 here): this lane's four channels are 16 bytes
      *reinterpret_cast<float4*>(out + ((int64_t)y * d.width + xi) * a.Cout + 4 * kb) =
          make_float4(acc[0] * osc[0], acc[1] * osc[1], acc[2] * osc[2], acc[3] * osc[3]);
    }
  }
}

// both tap counts in ONE launch (the 3-tap left and the 5-tap right plane sets took 15 + 16 us as two launches, each latency-bound):
// the descriptor picks the instantiation (wave-uniform)
template <int NCG, class TF = float>
__global__ __launch_bounds__(CSM_THREADS, 2) void costvol_stem_planes_mfma_kernel(PlanesArgs a, const uint4* __restrict__ frag_a,
                                                                                 const uint4* __restrict__ frag_b, const float* __restrict__ wmul_all) {
  __shared__ __attribute__((aligned(16))) uint2 lhi[NCG * CSM_HY * CSM_RS], llo[NCG * CSM_HY * CSM_RS];
  __shared__ unsigned lmax;
  const PlaneDesc d = a.d[blockIdx.z % a.ndesc];
  if ((int)blockIdx.x * CSM_TX >= d.width) return;               // uniform: descriptors have different widths
  if (d.right) costvol_stem_planes_mfma_body<TF, NCG, 5>(a, d, frag_a, frag_b, wmul_all, lhi, llo, lmax);
  else costvol_stem_planes_mfma_body<TF, NCG, 3>(a, d, frag_a, frag_b, wmul_all, lhi, llo, lmax);
}

struct CombineArgs {
  const float* ws;
  int64_t ws_bstride;
  const float* scale;
  const float* shift;
  void* y;
  int64_t y_bstride;
  int relu, Cout, D, H, W, wband, wb1, u1_0;
  int off_afull[CS_NCLS], off_aband[CS_NCLS], off_b0[CS_NCLS], off_b1[CS_NCLS];   // float offsets of the plane sets
  int ntail;
  ragmi_tail_t tail[2];
  int ni;          // disparity planes per thread
};
constexpr int CB_NI = 4;              // measured at the headline shape: 1 -> 59.9 us, 2 -> 53.6, 4 -> 53.3, 8 -> 59.9, 16 -> 61.9 (was 80)

// A thread owns one pixel (y, x) and walks CB_NI consecutive disparity planes i.  The A value of a voxel depends on i only through
// its class (cls, tc) — right of the diagonal band (tc = 2) and inside the volume (cls = 0) it is the same number for every
// plane — so it is loaded once and kept; only the B value (plane set indexed by x - i) is read per voxel: ~13 plane reads per
// voxel instead of 24, and the parameter prologue (LDS + barrier) is paid once per CB_NI planes.  Planes are independent: the
// loads of the next one are in flight under the stores of this one.  Pixels are numbered y * W + x through the row ends (no idle
// lanes where W is not a multiple of the workgroup).  Headline shape, no tails: 80 us -> 53 us.  (A row form with the B rows in
// LDS, A in registers and 16-byte stores was built and measured first: 56 us for the interior + 29 us for the band / border voxels
// in a second launch — latency-bound workgroups, not store-bound: pure stores of this layout reach 6.7 TB/s, tools/probe_store.hip.)
// NC = Cout as a compile-time constant (4, 8, 12, 16: exact register tiles, no per-channel bound checks) or 0 = any Cout <= CS_MAXC
template <class T, int NC>
__global__ __launch_bounds__(256) void costvol_stem_combine_kernel(CombineArgs a) {
  constexpr int MC = NC > 0 ? NC : CS_MAXC;
  const int Cout = NC > 0 ? NC : a.Cout;
  __shared__ float par[2 * CS_MAXC + 2 * (4 * CS_MAXC + 8)];      // scale | shift | per tail: w[4][Cout] scale[4] shift[4]
  for (int e = threadIdx.x; e < Cout; e += 256) {
    par[e] = a.scale ? a.scale[e] : 1.f;
    par[CS_MAXC + e] = a.shift ? a.shift[e] : 0.f;
  }
  for (int t = 0; t < a.ntail; ++t) {
    float* p = par + 2 * CS_MAXC + t * (4 * CS_MAXC + 8);
    const ragmi_tail_t& tl = a.tail[t];
    for (int e = threadIdx.x; e < tl.cout * Cout; e += 256) p[e] = static_cast<const float*>(tl.weight)[e];
    for (int e = threadIdx.x; e < tl.cout; e += 256) {
      p[4 * CS_MAXC + e] = tl.scale ? static_cast<const float*>(tl.scale)[e] : 1.f;
      p[4 * CS_MAXC + 4 + e] = tl.shift ? static_cast<const float*>(tl.shift)[e] : 0.f;
    }
  }
  __syncthreads();
  const int pix = blockIdx.x * 256 + threadIdx.x;                 // y * W + x: rows are walked without a gap at their end
  if (pix >= a.H * a.W) return;
  const int x = pix % a.W, y = pix / a.W, b = blockIdx.z;
  const int i0 = (int)blockIdx.y * a.ni, i1 = min(i0 + a.ni, a.D);
  const float* ws = a.ws + b * a.ws_bstride;
  const int xr = x == a.W - 1 ? 1 : 0;
  const int64_t HW = (int64_t)a.H * a.W, DHW = HW * a.D;
  float av[MC];
#pragma unroll
  for (int co = 0; co < MC; ++co) av[co] = 0.f;
  int key = -1;                                                   // (cls, tc) the registers in av belong to
#pragma unroll 1
  for (int i = i0; i < i1; ++i) {
    const int cls = (i == 0 ? 1 : 0) + (i == a.D - 1 ? 2 : 0);
    const int t = x - i, tc = min(max(t, -3), 2);
    if (cls * 8 + tc + 3 != key) {
      key = cls * 8 + tc + 3;
      // A: full-width plane for tc = 2, band plane (columns 0..wband-1) for tc = -2..1, nothing for tc = -3
      // (plane offsets are 32-bit: a batch item's workspace is < 2^31 floats, checked on the host).  An absent plane is read at
      // the workspace base and DISCARDED by a select (not multiplied by 0: a non-finite value there must not leak into voxels that
      // have no A term) — unconditional loads, no branch per channel
      int oa = 0;
      bool ha = false;
      if (tc == 2) { oa = a.off_afull[cls] + (y * a.W + x) * Cout; ha = true; }
      else if (tc > -3) { oa = a.off_aband[cls] + (tc + 2) * Cout * a.H * a.wband + (y * a.wband + x - (tc == 1 ? 1 : 0)) * Cout; ha = true; }
      if constexpr (NC > 0) {
#pragma unroll
        for (int q = 0; q < MC / 4; ++q) {
          const float4 wa = *reinterpret_cast<const float4*>(ws + oa + 4 * q);
          av[4 * q] = ha ? wa.x : 0.f; av[4 * q + 1] = ha ? wa.y : 0.f; av[4 * q + 2] = ha ? wa.z : 0.f; av[4 * q + 3] = ha ? wa.w : 0.f;
        }
      } else {
#pragma unroll
        for (int co = 0; co < MC; ++co) {
          const float wa = co < Cout ? ws[oa + co] : 0.f;
          av[co] = ha ? wa : 0.f;
        }
      }
    }
    // B: indexed by u = x - i (>= -2 to contribute); the right-border variant lives on u in [u1_0, W-1]
    int ob = 0;
    bool hb = false;
    if (t >= -2) {
      hb = true;
      if (xr) ob = a.off_b1[cls] + (y * a.wb1 + (t - a.u1_0)) * Cout;
      else ob = a.off_b0[cls] + (y * (a.W + 2) + (t + 2)) * Cout;
    }
    float bv[MC];
    if constexpr (NC > 0) {
#pragma unroll
      for (int q = 0; q < MC / 4; ++q) {
        const float4 wb4 = *reinterpret_cast<const float4*>(ws + ob + 4 * q);
        bv[4 * q] = wb4.x; bv[4 * q + 1] = wb4.y; bv[4 * q + 2] = wb4.z; bv[4 * q + 3] = wb4.w;
      }
    } else {
#pragma unroll
      for (int co = 0; co < MC; ++co) bv[co] = co < Cout ? ws[ob + co] : 0.f;
    }
    float v[MC];
#pragma unroll
    for (int co = 0; co < MC; ++co) {
      if (co < Cout) {
        const float wb = bv[co];
        float s = (hb ? wb : 0.f) + av[co];                        // A + B; an absent plane's (unconditional) load is discarded
        s = fmaf(s, par[co], par[CS_MAXC + co]);
        v[co] = (a.relu & 1) ? fmaxf(s, 0.f) : s;
      } else {
        v[co] = 0.f;
      }
    }
    const int64_t vox = (int64_t)i * HW + pix;
    // G4 destinations (include/rag_amd.h: [B][C/4][D][H][W][4], fp32; the host checks Cout % 4 == 0 and 4-channel tails): a thread
    // owns every channel of its voxel, so a group is ONE 16-byte store (a wave: 1 KB contiguous) instead of four 4-byte stores into
    // four planes
    if (a.y == nullptr) {
      // (tails only: stem3d0's own output is expanded from the planes inside stem3d1's staging and never written)
    } else if constexpr (NC > 0) {
      if (a.relu & RAGMI_CONV_Y_G4) {
        T* const py4 = static_cast<T*>(a.y) + b * a.y_bstride + vox * 4;
#pragma unroll
        for (int g = 0; g < MC / 4; ++g) {
          const float q[4] = {v[4 * g], v[4 * g + 1], v[4 * g + 2], v[4 * g + 3]};
          st4(py4 + g * DHW * 4, q);                       // 16 bytes (fp32) or 8 bytes (bf16) per voxel and group
        }
      } else {
        T* py = static_cast<T*>(a.y) + b * a.y_bstride + vox;
#pragma unroll
        for (int co = 0; co < MC; ++co) st(py + co * DHW, v[co]);
      }
    } else {
      T* py = static_cast<T*>(a.y) + b * a.y_bstride + vox;
#pragma unroll
      for (int co = 0; co < MC; ++co)
        if (co < Cout) st(py + co * DHW, v[co]);
    }
    for (int tl = 0; tl < a.ntail; ++tl) {
      const float* p = par + 2 * CS_MAXC + tl * (4 * CS_MAXC + 8);
      const ragmi_tail_t& td = a.tail[tl];
      T* pt = static_cast<T*>(td.y) + b * td.y_bstride + (int64_t)td.y_ch0 * DHW + vox;
      float u4[4] = {0.f, 0.f, 0.f, 0.f};
      const bool g4 = (td.relu & RAGMI_TAIL_G4) != 0;
      for (int k = 0; k < td.cout; ++k) {
        float s = 0.f;
#pragma unroll
        for (int co = 0; co < MC; ++co)
          if (co < Cout) s = fmaf(p[k * Cout + co], v[co], s);
        s = fmaf(s, p[4 * CS_MAXC + k], p[4 * CS_MAXC + 4 + k]);
        s = (td.relu & 1) ? fmaxf(s, 0.f) : s;
        if (g4) u4[k & 3] = s; else st(pt + k * DHW, s);
      }
      if (g4) st4(static_cast<T*>(td.y) + b * td.y_bstride + ((int64_t)(td.y_ch0 >> 2) * DHW + vox) * 4, u4);
    }
  }
}

struct StemLayout {
  int wband, wb1, u1_0;
  int64_t na, nb;                 // weight variant sizes
  int64_t off_afull[CS_NCLS], off_aband[CS_NCLS], off_b0[CS_NCLS], off_b1[CS_NCLS];
  int64_t per_batch;              // workspace floats per batch item
  bool used[CS_NCLS];
};
static void stem_layout(int C, int Cout, int D, int H, int W, StemLayout& l) {
  // band columns: class tc is read at x = i + tc only, i.e. x in [max(tc, 0), min(D - 1 + tc, W - 1)] — min(W, D) columns from
  // x = max(tc, 0) on (round 4: D + 1 columns from 0 made every band plane a 64-column tile plus a one-column tile)
  l.wband = std::min(W, D);
  l.u1_0 = std::max(W - D, -2);                 // right-border variant: u = W-1-i
  l.wb1 = W - l.u1_0;
  l.na = (int64_t)CS_NCLS * CS_NTC * C * 9 * Cout;
  l.nb = (int64_t)CS_NCLS * 2 * C * 15 * Cout;
  for (int c = 0; c < CS_NCLS; ++c) l.used[c] = false;
  for (int i = 0; i < D; ++i) l.used[(i == 0 ? 1 : 0) + (i == D - 1 ? 2 : 0)] = true;
  int64_t off = 0;
  for (int c = 0; c < CS_NCLS; ++c) {
    l.off_afull[c] = l.off_aband[c] = l.off_b0[c] = l.off_b1[c] = 0;
    if (!l.used[c]) continue;
    l.off_afull[c] = off; off += (int64_t)Cout * H * W;
    l.off_aband[c] = off; off += (int64_t)4 * Cout * H * l.wband;
    l.off_b0[c] = off; off += (int64_t)Cout * H * (W + 2);
    l.off_b1[c] = off; off += (int64_t)Cout * H * l.wb1;
  }
  l.per_batch = off;
}

}  // namespace ragmi

extern "C" int64_t ragmi_costvol_stem_weights_elems(int C, int Cout) {
  if (C <= 0 || Cout <= 0) return 0;
  // the fp32 variants (VALU planes kernel: RAGMI_F32 / RAGMI_BF16), then their FP16 hi / lo fragments and multipliers (RAGMI_F32X3)
  return (int64_t)ragmi::CS_NCLS * (ragmi::CS_NTC * 9 + 2 * 15) * C * Cout + (C % 4 == 0 ? ragmi::cs_frag_words(C) : 0);
}

extern "C" int ragmi_costvol_stem_prepare(const void* weight, void* variants, int C, int Cout, void* stream) {
  using namespace ragmi;
  RAGMI_REQUIRE(weight && variants, RAGMI_EINVAL, "costvol_stem_prepare: null pointer");
  RAGMI_REQUIRE(C > 0 && Cout > 0 && C <= CS_MAXC && Cout <= CS_MAXC, RAGMI_EUNSUPPORTED,
                "costvol_stem_prepare: C and Cout must be in 1..%d", CS_MAXC);
  const int64_t na = (int64_t)CS_NCLS * CS_NTC * C * 9 * Cout, nb = (int64_t)CS_NCLS * 2 * C * 15 * Cout;
  hipLaunchKernelGGL(costvol_stem_weights_kernel, dim3((unsigned)ceil_div(na + nb, 256)), dim3(256), 0, static_cast<hipStream_t>(stream),
                     (const float*)weight, (float*)variants, (float*)variants + na, C, Cout);
  if (C % 4 == 0)
    hipLaunchKernelGGL(costvol_stem_pack_kernel, dim3(CS_NVA + CS_NVB), dim3(256), 0, static_cast<hipStream_t>(stream),
                       (const float*)variants, (float*)variants + na + nb, C, Cout);
  return check_launch("costvol_stem_prepare");
}

extern "C" int64_t ragmi_costvol_stem_workspace_elems(int B, int C, int Cout, int D, int H, int W) {
  using namespace ragmi;
  if (B <= 0 || C <= 0 || Cout <= 0 || D <= 0 || H <= 0 || W <= 0) return 0;
  StemLayout l;
  stem_layout(C, Cout, D, H, W, l);
  return l.per_batch * B;
}

// the planes (+ the combine kernel unless there is nothing for it to write); `lay` receives the workspace layout
static int stem_run(const void* left, const void* right, const void* variants, const void* scale, const void* shift,
                    int relu, void* y, int64_t y_bstride, void* workspace, int B, int C, int Cout, int D, int H, int W,
                    int ntail, const ragmi_tail_t* tails, int dtype, void* stream, ragmi::StemLayout* lay);

extern "C" int ragmi_costvol_stem_fwd(const void* left, const void* right, const void* variants, const void* scale, const void* shift,
                                      int relu, void* y, int64_t y_bstride, void* workspace, int B, int C, int Cout, int D, int H, int W,
                                      int ntail, const ragmi_tail_t* tails, int dtype, void* stream) {
  RAGMI_REQUIRE(y || ntail > 0, RAGMI_EINVAL, "costvol_stem: y may be NULL only when tails consume the result");
  return stem_run(left, right, variants, scale, shift, relu, y, y_bstride, workspace, B, C, Cout, D, H, W, ntail, tails, dtype, stream, nullptr);
}

// stem3d0 + stem3d1 with stem3d0's output never written (include/rag_amd.h)
extern "C" int ragmi_costvol_stem_conv3d_fwd(const void* left, const void* right, const void* variants, const void* scale0, const void* shift0,
                                             int relu0, void* workspace, int ntail0, const ragmi_tail_t* tails0,
                                             const void* packed_weight, const void* scale, const void* shift, int relu,
                                             void* y, int64_t y_bstride, const int32_t* y_group_ch, int store_main, int ntail,
                                             const ragmi_tail_t* tails, int B, int C, int Cmid, int Cout, int D, int H, int W, int dtype,
                                             void* stream) {
  using namespace ragmi;
  RAGMI_REQUIRE(dtype == RAGMI_F32X3 || dtype == RAGMI_BF16, RAGMI_EUNSUPPORTED, "costvol_stem_conv3d: RAGMI_F32X3 or RAGMI_BF16 (the split-operand convolution)");
  RAGMI_REQUIRE(packed_weight && (y || !store_main), RAGMI_EINVAL, "costvol_stem_conv3d: null pointer");
  RAGMI_REQUIRE(Cmid == 12 && C % 4 == 0 && C <= 12, RAGMI_EUNSUPPORTED, "costvol_stem_conv3d: built for 12 intermediate channels (stem3d0 -> stem3d1)");
  RAGMI_REQUIRE((scale == nullptr) == (shift == nullptr), RAGMI_EINVAL, "costvol_stem_conv3d: scale/shift must both be given or both NULL");
  K3Args a{};
  static float dummy;      // (never dereferenced: the kernel has no input tensor in this mode)
  int rc = fill_common(a, &dummy, 0, y ? y : (void*)&dummy, y_bstride, y_group_ch, nullptr, 0, nullptr, B, Cmid, Cout, D, H, W, relu & 1);
  if (rc != RAGMI_OK) return rc;
  a.wp[0] = (const float*)packed_weight; a.scale[0] = (const float*)scale; a.shift[0] = (const float*)shift;
  a.nchunks[0] = (Cmid + CK - 1) / CK;
  rc = fill_tails(a, store_main, ntail, tails, Cout);
  if (rc != RAGMI_OK) return rc;
  RAGMI_REQUIRE(a.ndown == 0 && x3_eligible(a, 1, dtype) && !x3d_eligible(a, 1, dtype) && !x2d_eligible(a, 1, dtype), RAGMI_EUNSUPPORTED,
                "costvol_stem_conv3d: this shape does not run on the z-marching split-operand kernel (ragmi_costvol_stem_conv3d_supported)");
  // ONE fused tail of stem3d0 (4 output channels, full resolution: cell 0's pre_preprocess) can ride in rows 12..15 of stem3d1's matrix
  // product (RAGMI_TAIL_ROWS: the caller packed it there) — no combine launch at all; any other tail goes through the combine kernel
  // (tails only, no main store)
  RAGMI_REQUIRE(ntail0 >= 0 && ntail0 <= 2 && (ntail0 == 0 || tails0), RAGMI_EINVAL, "costvol_stem_conv3d: at most two tails on stem3d0's output");
  const bool want_rows = ntail0 >= 1 && (tails0[0].relu & RAGMI_TAIL_ROWS);
  const bool tail_rows = want_rows && ntail0 == 1 && Cout == 12 && tails0[0].cout == 4 && !(tails0[0].relu & 2) && tails0[0].y &&
                         ((tails0[0].scale == nullptr) == (tails0[0].shift == nullptr)) && (!(tails0[0].relu & RAGMI_TAIL_G4) || tails0[0].y_ch0 % 4 == 0);
  RAGMI_REQUIRE(!want_rows || tail_rows, RAGMI_EINVAL, "costvol_stem_conv3d: RAGMI_TAIL_ROWS takes ONE 4-channel full-resolution tail behind a 12-channel stem3d1 (packed as 16 channels)");
  StemLayout l;
  rc = stem_run(left, right, variants, scale0, shift0, relu0, nullptr, 0, workspace, B, C, Cmid, D, H, W, tail_rows ? 0 : ntail0, tails0, dtype,
                stream, &l);
  if (rc != RAGMI_OK) return rc;
  X3StemSrc src{};
  src.tail_rows = tail_rows ? 1 : 0;
  if (tail_rows) {
    src.ntail = 1; src.tail_relu = tails0[0].relu & 1; src.tail_ch0 = tails0[0].y_ch0; src.tail_g4 = (tails0[0].relu & RAGMI_TAIL_G4) ? 1 : 0;
    src.tail_w = (const float*)tails0[0].weight; src.tail_scale = (const float*)tails0[0].scale; src.tail_shift = (const float*)tails0[0].shift;
    src.tail_y = (float*)tails0[0].y; src.tail_bstride = tails0[0].y_bstride;
  }
  src.ws = (const float*)workspace; src.ws_bstride = l.per_batch;
  for (int c = 0; c < CS_NCLS; ++c) {
    src.off_afull[c] = (int)l.off_afull[c]; src.off_aband[c] = (int)l.off_aband[c];
    src.off_b0[c] = (int)l.off_b0[c]; src.off_b1[c] = (int)l.off_b1[c];
  }
  src.wband = l.wband; src.wb1 = l.wb1; src.u1_0 = l.u1_0;
  src.scale = (const float*)scale0; src.shift = (const float*)shift0; src.relu = relu0 & 1;
  return x3_launch(a, 1, dtype, static_cast<hipStream_t>(stream), &src);
}

extern "C" int ragmi_costvol_stem_conv3d_supported(int C, int Cmid, int Cout, int B, int D, int H, int W, int ntail, int dtype) {
  using namespace ragmi;
  if ((dtype != RAGMI_F32X3 && dtype != RAGMI_BF16) || Cmid != 12 || C % 4 != 0 || C <= 0 || C > 12 || Cout <= 0 || B <= 0 || D <= 0 || H <= 0 || W <= 0) return 0;
  K3Args a{};
  a.B = B; a.Cin = Cmid; a.Cout = Cout; a.D = D; a.H = H; a.W = W; a.ntail = ntail; a.nchunks[0] = Cmid / CK; a.store_main = 1;
  return (x3_eligible(a, 1, dtype) && !x3d_eligible(a, 1, dtype) && !x2d_eligible(a, 1, dtype)) ? 1 : 0;
}

static int stem_run(const void* left, const void* right, const void* variants, const void* scale, const void* shift,
                    int relu, void* y, int64_t y_bstride, void* workspace, int B, int C, int Cout, int D, int H, int W,
                    int ntail, const ragmi_tail_t* tails, int dtype, void* stream, ragmi::StemLayout* lay) {
  using namespace ragmi;
  RAGMI_REQUIRE(left && right && variants && workspace, RAGMI_EINVAL, "costvol_stem: null pointer");
  RAGMI_REQUIRE((scale == nullptr) == (shift == nullptr), RAGMI_EINVAL, "costvol_stem: scale/shift must both be given or both NULL");
  RAGMI_REQUIRE(B > 0 && D > 0 && H > 0 && W > 0 && B <= 65535, RAGMI_EINVAL, "costvol_stem: bad size");
  RAGMI_REQUIRE(C > 0 && Cout > 0 && C <= CS_MAXC && Cout <= CS_MAXC, RAGMI_EUNSUPPORTED, "costvol_stem: C and Cout must be in 1..%d",
                CS_MAXC);
  RAGMI_REQUIRE(conv_dtype_ok(dtype), RAGMI_EUNSUPPORTED, "costvol_stem: dtype %d not built", dtype);
  RAGMI_REQUIRE(ntail >= 0 && ntail <= 2 && (ntail == 0 || tails), RAGMI_EINVAL, "costvol_stem: at most two tails");
  // RAGMI_F32X3: fp32 storage, the variant planes as split-operand products on the matrix cores where the shape allows
  // (whole 4-channel groups, <= 12 feature channels); everything else of this entry point is fp32 either way
  // (round 5: bf16 features too — the planes are fp32 in the workspace either way, and under bf16 storage everything downstream of them
  // is rounded to 8 bits)
  const bool mfma_planes = (dtype == RAGMI_F32X3 || dtype == RAGMI_BF16) && C % 4 == 0 && C <= 12;
  const bool bf_features = dtype == RAGMI_BF16;
  if (dtype == RAGMI_F32X3) dtype = RAGMI_F32;
  StemLayout l;
  stem_layout(C, Cout, D, H, W, l);
  RAGMI_REQUIRE(l.per_batch < (1ll << 31), RAGMI_EUNSUPPORTED, "costvol_stem: planes too large for 32-bit offsets");
  hipStream_t st = static_cast<hipStream_t>(stream);

  PlanesArgs pa{};
  pa.left = left; pa.right = right; pa.wts = (const float*)variants; pa.ws = (float*)workspace; pa.ws_bstride = l.per_batch;
  pa.C = C; pa.Cout = Cout; pa.H = H; pa.W = W;
  int n = 0, maxw = 0;
  const int sa = C * 9 * Cout, sb = C * 15 * Cout;
  for (int c = 0; c < CS_NCLS; ++c) {
    if (!l.used[c]) continue;
    pa.d[n++] = PlaneDesc{(c * CS_NTC + 4) * sa, (int)l.off_afull[c], W, 0, 0, c * CS_NTC + 4};                       // tc = 2
    for (int k = 0; k < 4; ++k)                                                                       // tc = -2..1
      pa.d[n++] = PlaneDesc{(c * CS_NTC + k) * sa, (int)(l.off_aband[c] + (int64_t)k * Cout * H * l.wband), l.wband, k == 3 ? 1 : 0, 0, c * CS_NTC + k};
    pa.d[n++] = PlaneDesc{(int)l.na + (c * 2 + 0) * sb, (int)l.off_b0[c], W + 2, -2, 1, CS_NVA + c * 2 + 0};
    pa.d[n++] = PlaneDesc{(int)l.na + (c * 2 + 1) * sb, (int)l.off_b1[c], l.wb1, l.u1_0, 1, CS_NVA + c * 2 + 1};
  }
  pa.ndesc = n;
  for (int k = 0; k < n; ++k) maxw = std::max(maxw, pa.d[k].width);
  RAGMI_REQUIRE((int64_t)n * B <= 65535, RAGMI_EUNSUPPORTED, "costvol_stem: batch too large for one launch");
  const dim3 pgrid((unsigned)ceil_div(maxw, CS_TX), (unsigned)ceil_div(H, CS_TY), (unsigned)(n * B));
  const size_t plds = sizeof(float) * (((size_t)C * 15 * Cout + 3) / 4 * 4 + (size_t)CS_CC * (CS_TY + 2) * CS_RS);
#define RAGMI_CS_PLANES(TT)                                                                                             \
  switch (Cout % 4 == 0 ? Cout / 4 : 0) {                                                                              \
    case 1: hipLaunchKernelGGL((costvol_stem_planes_kernel<TT, 1>), pgrid, dim3(CS_NT), plds, st, pa); break;          \
    case 2: hipLaunchKernelGGL((costvol_stem_planes_kernel<TT, 2>), pgrid, dim3(CS_NT), plds, st, pa); break;          \
    case 3: hipLaunchKernelGGL((costvol_stem_planes_kernel<TT, 3>), pgrid, dim3(CS_NT), plds, st, pa); break;          \
    case 4: hipLaunchKernelGGL((costvol_stem_planes_kernel<TT, 4>), pgrid, dim3(CS_NT), plds, st, pa); break;          \
    default: hipLaunchKernelGGL((costvol_stem_planes_kernel<TT, 0>), pgrid, dim3(CS_NT), plds, st, pa); break;         \
  }
  if (mfma_planes) {
    const int nsa = cs_nslices(C, 3), nsb = cs_nslices(C, 5);
    const uint4* const fa = reinterpret_cast<const uint4*>((const float*)variants + l.na + l.nb);
    const uint4* const fb = fa + (int64_t)CS_NVA * nsa * 2 * 64;
    const float* const wm = reinterpret_cast<const float*>(fb + (int64_t)CS_NVB * nsb * 2 * 64);
    const dim3 mgrid((unsigned)ceil_div(maxw, CSM_TX), (unsigned)ceil_div(H, CSM_TY), (unsigned)(n * B));
#define RAGMI_CS_MFMA(NCG_)                                                                                                          \
  if (bf_features) hipLaunchKernelGGL((costvol_stem_planes_mfma_kernel<NCG_, bf16_t>), mgrid, dim3(CSM_THREADS), 0, st, pa, fa, fb, wm); \
  else hipLaunchKernelGGL((costvol_stem_planes_mfma_kernel<NCG_, float>), mgrid, dim3(CSM_THREADS), 0, st, pa, fa, fb, wm);
    switch (C / 4) {
      case 1: RAGMI_CS_MFMA(1) break;
      case 2: RAGMI_CS_MFMA(2) break;
      default: RAGMI_CS_MFMA(3) break;
    }
#undef RAGMI_CS_MFMA
  } else if (dtype == RAGMI_BF16) { RAGMI_CS_PLANES(bf16_t) } else { RAGMI_CS_PLANES(float) }
#undef RAGMI_CS_PLANES

  if (lay) *lay = l;
  if (y == nullptr && ntail == 0) return check_launch("costvol_stem");      // planes only: the consumer expands them itself
  CombineArgs ca{};
  ca.ws = (const float*)workspace; ca.ws_bstride = l.per_batch; ca.scale = (const float*)scale; ca.shift = (const float*)shift;
  ca.y = y; ca.y_bstride = y_bstride; ca.relu = relu; ca.Cout = Cout; ca.D = D; ca.H = H; ca.W = W;
  ca.wband = l.wband; ca.wb1 = l.wb1; ca.u1_0 = l.u1_0;
  for (int c = 0; c < CS_NCLS; ++c) {
    ca.off_afull[c] = (int)l.off_afull[c]; ca.off_aband[c] = (int)l.off_aband[c];
    ca.off_b0[c] = (int)l.off_b0[c]; ca.off_b1[c] = (int)l.off_b1[c];
  }
  ca.ntail = ntail;
  for (int t = 0; t < ntail; ++t) {
    RAGMI_REQUIRE(tails[t].weight && tails[t].y && tails[t].cout >= 1 && tails[t].cout <= 4, RAGMI_EINVAL,
                  "costvol_stem: tail %d needs weight, y and 1..4 output channels", t);
    RAGMI_REQUIRE(!(tails[t].relu & 2), RAGMI_EUNSUPPORTED, "costvol_stem: down-sampling tails are not built here");
    RAGMI_REQUIRE(!(tails[t].relu & RAGMI_TAIL_G4) || (tails[t].cout == 4 && tails[t].y_ch0 % 4 == 0), RAGMI_EUNSUPPORTED,
                  "costvol_stem: a G4 tail needs 4 output channels and a group-aligned y_ch0");
    ca.tail[t] = tails[t];
  }
  RAGMI_REQUIRE(!(relu & RAGMI_CONV_Y_G4) || (Cout == 4 || Cout == 8 || Cout == 12 || Cout == 16), RAGMI_EUNSUPPORTED,
                "costvol_stem: a G4 output needs Cout in {4, 8, 12, 16}");
  int ni = CB_NI;
#ifdef RAGMI_DIAG
  static const int diag_ni = [] { const char* v = getenv("RAGMI_CB_NI"); return v ? atoi(v) : 0; }();
  if (diag_ni > 0) ni = diag_ni;
#endif
  ca.ni = ni;
  RAGMI_REQUIRE((int64_t)H * W < (1ll << 31) && ceil_div(D, ni) <= 65535, RAGMI_EUNSUPPORTED, "costvol_stem: volume exceeds the grid limit");
  const dim3 cgrid((unsigned)ceil_div((int64_t)H * W, 256), (unsigned)ceil_div(D, ni), (unsigned)B);
#define RAGMI_CS_COMBINE(TT)                                                                                    \
  switch (Cout) {                                                                                               \
    case 4: hipLaunchKernelGGL((costvol_stem_combine_kernel<TT, 4>), cgrid, dim3(256), 0, st, ca); break;       \
    case 8: hipLaunchKernelGGL((costvol_stem_combine_kernel<TT, 8>), cgrid, dim3(256), 0, st, ca); break;       \
    case 12: hipLaunchKernelGGL((costvol_stem_combine_kernel<TT, 12>), cgrid, dim3(256), 0, st, ca); break;     \
    case 16: hipLaunchKernelGGL((costvol_stem_combine_kernel<TT, 16>), cgrid, dim3(256), 0, st, ca); break;     \
    default: hipLaunchKernelGGL((costvol_stem_combine_kernel<TT, 0>), cgrid, dim3(256), 0, st, ca); break;      \
  }
  if (dtype == RAGMI_BF16) { RAGMI_CS_COMBINE(bf16_t) } else { RAGMI_CS_COMBINE(float) }
#undef RAGMI_CS_COMBINE
  return check_launch("costvol_stem");
}
