// Shared pieces of the split-operand 3x3x3 convolution kernels (conv3d_x3.hip: z-marching and deep forms): operand splits, the MFMA wrapper, operand-scale helpers, packed-fragment layout and launch extras.
#pragma once
#include <cstdlib>

#include "conv3d_k3.h"

namespace ragmi {

typedef __bf16 x3_bf16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 x3_f16x8 __attribute__((ext_vector_type(8)));

// z-marching: a workgroup owns a (y, x) tile of 8 x 32 voxels and walks a segment of the depth axis keeping a ring of three
// input planes (10 x 34 halo) in LDS: every input plane is fetched once per column (halo overhead 1.33x instead of 2.66x
// for a 2-deep box tile), the next plane travels HBM -> registers while the current one is multiplied.
constexpr int X3_TY = 8, X3_TX = 32;
constexpr int64_t X3_MIN_VOXELS = 1 << 18;      // below this the z-marching columns do not fill the chip
constexpr int X3_HY = X3_TY + 2, X3_HX = X3_TX + 2, X3_PL = X3_HY * X3_HX;   // one halo plane: 340 voxels
// LDS geometry of the halo ring, in 8-byte records: [channel group][ring slot][y][x].  A ds_read_b64 is served 32 lanes at a time,
// i.e. TWO lane quarters (16 lanes = 128 contiguous bytes each) share the 64 banks: they pass in one go only if their addresses are
// 128 bytes apart mod 256, or overlap with IDENTICAL addresses (two taps of one halo row).  Which two operand pairs meet in a pass
// is a free choice per K-slice — the pairs of a slice can sit in any lane quarter as long as the weight fragments sit there too — so
// the strides below are padded and the pairs permuted (x3_pair_perm) for the fewest conflicted passes over the three ring phases
// (search: tools/x3_bank_search.py):
//  * one channel group per set (the level-3 dual cells): pairs as packed — quarters two taps apart; rows of 49 records put the row
//    wraps 128 bytes apart, 9 records behind every plane do the same for the plane wraps: 3 of 48 passes conflicted (round 3: 9);
//  * more groups: rows of 34, 20 records (160 B) behind every group (group stride = 128 mod 256) and quarters (0,1) / (2,3) reading
//    the SAME tap of neighbouring groups wherever the slice allows: stem3d1 (3 groups) 6 of 132 passes conflicted — as packed it was
//    120 of 132, every operand read of that launch at half rate; even group counts 0.
constexpr int x3_row_stride(int ncg, int nset) { return ncg == nset ? 49 : X3_HX; }
constexpr int x3_plane_stride(int ncg, int nset) { return X3_HY * x3_row_stride(ncg, nset) + (ncg == nset ? 9 : 0); }
constexpr int x3_group_stride(int ncg, int nset) { return 3 * x3_plane_stride(ncg, nset) + (ncg == nset ? 0 : 20); }
// pair (0..7, position in the packed K-slice `s` of a set with `ncgs` channel groups) that lane quarter i >> 1 holds as its operand i & 1
__host__ __device__ inline int x3_pair_perm(int ncgs, int s, int i) {
  static constexpr unsigned char even[8] = {0, 2, 1, 3, 4, 6, 5, 7};
  static constexpr unsigned char three[11][8] = {{0,2,1,5,3,6,4,7}, {0,1,2,7,3,4,6,5}, {0,2,1,3,4,5,7,6}, {0,2,1,3,4,6,5,7}, {0,1,3,2,4,5,7,6}, {0,2,1,3,4,6,5,7},
                                                 {0,2,1,5,3,6,4,7}, {0,1,6,4,2,5,3,7}, {0,2,1,3,4,5,7,6}, {0,2,1,5,3,6,4,7}, {0,2,1,3,4,6,5,7}};
  static constexpr unsigned char five[17][8] = {{0,2,1,7,3,5,4,6}, {0,2,1,3,4,6,5,7}, {0,2,1,3,4,6,5,7}, {0,1,5,2,3,6,4,7}, {0,2,1,7,3,5,4,6}, {0,2,1,3,4,6,5,7},
                                                {0,2,1,7,3,5,4,6}, {0,2,1,3,4,6,5,7}, {0,1,5,2,3,6,4,7}, {0,2,1,3,4,6,5,7}, {0,2,1,7,3,5,4,6}, {0,2,1,7,3,5,4,6},
                                                {0,2,1,3,4,6,5,7}, {0,2,1,3,4,6,5,7}, {0,2,1,7,3,5,4,6}, {0,2,1,7,3,5,4,6}, {0,2,1,3,4,6,5,7}};
  if (ncgs == 1) return i;
  if (ncgs == 3) return three[s][i];
  if (ncgs == 5) return five[s][i];
  return even[i];
}
constexpr int X3_THREADS = 512, X3_WAVES = X3_THREADS / 64;
// 8 waves per workgroup, two column tiles each: at <= 128 VGPRs two workgroups (4 waves per SIMD) share a CU, which hides the
// LDS-read latency in front of every MFMA group far better than 4 waves x 4 tiles at 248 VGPRs did (557 -> 601 maps/s)
constexpr int X3_NT = X3_TY * X3_TX / 16 / X3_WAVES;                  // 16-voxel column tiles per wave per plane (2)

__device__ __forceinline__ unsigned short x3_bf16_rn(float v) {
  unsigned u = __float_as_uint(v);
  u += 0x7fffu + ((u >> 16) & 1u);
  return (unsigned short)(u >> 16);
}
__device__ __forceinline__ void x3_split(float v, unsigned short& hi, unsigned short& lo) {
  hi = x3_bf16_rn(v);
  lo = x3_bf16_rn(v - __uint_as_float((unsigned)hi << 16));
}
// two values at once through the packed converter (v_cvt_pk_bf16_f32, round to nearest even): returns the packed hi pair,
// writes the packed lo pair
typedef __bf16 x3_bf16x2 __attribute__((ext_vector_type(2)));
typedef _Float16 x3_f16x2 __attribute__((ext_vector_type(2)));
typedef float x3_f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ unsigned x3_split2(float v0, float v1, unsigned& lo) {
  const x3_bf16x2 h = __builtin_convertvector(x3_f32x2{v0, v1}, x3_bf16x2);
  const unsigned hb = __builtin_bit_cast(unsigned, h);
  const float r0 = v0 - __uint_as_float(hb << 16), r1 = v1 - __uint_as_float(hb & 0xffff0000u);
  lo = __builtin_bit_cast(unsigned, __builtin_convertvector(x3_f32x2{r0, r1}, x3_bf16x2));
  return hb;
}
// fp16 halves of two values scaled by the power of two `mul`: hi = fp16(v * mul), lo = fp16(v * mul - hi) (the difference is exact)
__device__ __forceinline__ unsigned x3_split2h(float v0, float v1, float mul, unsigned& lo) {
  const float s0 = v0 * mul, s1 = v1 * mul;
  const x3_f16x2 h = __builtin_convertvector(x3_f32x2{s0, s1}, x3_f16x2);
  const float r0 = s0 - (float)h.x, r1 = s1 - (float)h.y;
  lo = __builtin_bit_cast(unsigned, __builtin_convertvector(x3_f32x2{r0, r1}, x3_f16x2));
  return __builtin_bit_cast(unsigned, h);
}
__device__ __forceinline__ unsigned short x3_f16_bits(float v) { return __builtin_bit_cast(unsigned short, (_Float16)v); }
// one 16x16x32 product on the matrix cores: bf16 operands (bf16 activation storage) or fp16 operands (fp32 storage)
template <bool BF>
__device__ __forceinline__ f32x4 x3_mma(const uint4& a, const uint4& b, const f32x4& c) {
  if constexpr (BF) return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(x3_bf16x8, a), __builtin_bit_cast(x3_bf16x8, b), c, 0, 0, 0);
  else return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(x3_f16x8, a), __builtin_bit_cast(x3_f16x8, b), c, 0, 0, 0);
}
constexpr unsigned X3_SCALE_FLOOR_BITS = 0x0d000000u;      // 2^-101: the smallest operand scale (Inf and |x| > ~2^110 cannot be scaled into fp16)
// largest power of two p with p * m <= target (m > 0 finite), clamped to [2^-101, 2^99]: the operand scale
__device__ __forceinline__ float x3_pow2_scale(float m, float target) {
  const float q = target / fmaxf(m, 1e-30f);
  return __uint_as_float(min(max(__float_as_uint(q) & 0x7f800000u, X3_SCALE_FLOOR_BITS), 0x71000000u));
}
// largest |v| of four values among those a power-of-two scale can bring into fp16's range: bit patterns from 2^115 up (finite
// outliers the floor scale 2^-101 cannot fit, Inf, NaN) count as 0.  Bit patterns of non-negative floats order like the values.
constexpr unsigned X3_UNSCALABLE_BITS = 0x79000000u;      // 2^115
__device__ __forceinline__ float x3_scalable_max4(float v0, float v1, float v2, float v3) {
  auto f = [](float v) { const unsigned b = __float_as_uint(v) & 0x7fffffffu; return b < X3_UNSCALABLE_BITS ? b : 0u; };
  return __uint_as_float(max(max(f(v0), f(v1)), max(f(v2), f(v3))));
}
constexpr float X3_F16_CAP = 60000.f;         // |x| * 2^-e must stay below fp16's 65504
constexpr float X3_ACT_TARGET = 2048.f;       // 2^11: the largest scaled |x| when a column's scale is chosen (16x headroom)
constexpr float X3_W_TARGET = 1024.f;         // 2^10: the largest scaled |w| of an output channel
// largest value over the wave's 64 lanes (values >= 0), returned in every lane: four DPP steps reduce each row of 16 lanes
// (xor 1, xor 2, half-row mirror, row mirror), four v_readlane + scalar max join the rows — no LDS traffic and no waits
// (__shfl_xor is six ds_bpermute round trips; in the deep-level kernel, once per 54 MFMAs, that was +10 us per launch)
__device__ __forceinline__ float x3_wave_max(float m) {
  int v = __float_as_int(m);                     // non-negative floats order like their bit patterns
  v = max(v, __builtin_amdgcn_update_dpp(0, v, 0xB1, 0xF, 0xF, false));     // quad_perm [1,0,3,2]
  v = max(v, __builtin_amdgcn_update_dpp(0, v, 0x4E, 0xF, 0xF, false));     // quad_perm [2,3,0,1]
  v = max(v, __builtin_amdgcn_update_dpp(0, v, 0x141, 0xF, 0xF, false));    // row_half_mirror
  v = max(v, __builtin_amdgcn_update_dpp(0, v, 0x140, 0xF, 0xF, false));    // row_mirror
  const int r = max(max(__builtin_amdgcn_readlane(v, 0), __builtin_amdgcn_readlane(v, 16)),
                    max(__builtin_amdgcn_readlane(v, 32), __builtin_amdgcn_readlane(v, 48)));
  return __int_as_float(r);
}

// packed weight fragments of ONE accumulator set (a conv with Cout outputs and Cin = 4 * ncgs inputs):
// wf[((cog * nsls + s) * 2 + hl) * 64 + lane] (uint4 = 8 halves): A[row = lane & 15][k = 8 (lane>>4) + j],
// k -> pair P = 8 s + 2 (lane>>4) + (j>>2) = tap * ncgs + cg (TAP-MAJOR since round 4), channel 4 cg + (j&3); pairs past 27 * ncgs
// are zeros.  Tap-major puts the 8 (16) channels of ONE tap into one (two) lane quarter(s) of a slice, which is exactly the operand
// record of the deep-level kernel: both kernels read the same fragments, 27 taps in 28 K slots for 8 / 16 channels.
// The source is indexed like the fp32
// pack (transpose / planar options of ragmi_conv3d_k3_pack_ex).  Two sections: bf16 halves of w (bf16 activation storage),
// then fp16 halves of w * 2^k[co] followed by the per-output-channel multipliers 2^-k[co] (fp32 storage, RAGMI_F32X3).
__device__ __forceinline__ float x3_w_at(const float* __restrict__ w, int Cout, int Cin, int co, int ci, int tap, int transpose, int planar) {
  if (co >= Cout || ci >= Cin) return 0.f;
  const int taps = planar ? 9 : 27;
  int t = planar ? tap - 9 : tap;
  if (t < 0 || t >= taps) return 0.f;
  if (transpose) t = taps - 1 - t;
  return transpose ? w[((int64_t)ci * Cout + co) * taps + t] : w[((int64_t)co * Cin + ci) * taps + t];
}
// power of two that brings the largest |w| of an output channel (given as the bit pattern of that maximum) to [2^9, 2^10]
// (1 for an all-zero or absent channel)
__device__ __forceinline__ float x3_row_mul(unsigned maxbits) {
  const float m = __uint_as_float(maxbits);
  return m > 0.f ? x3_pow2_scale(m, X3_W_TARGET) : 1.f;
}
// largest |w| of every output channel, as float bit patterns in rowmax[Cout <= 64] (LDS): the workgroup's threads stride the
// weight tensor once (a scan per thread cost ~10 us per pack, 150 packs per training step)
__device__ __forceinline__ void x3_row_max(const float* __restrict__ w, unsigned* rowmax, int Cout, int Cin, int transpose, int planar,
                                           int tid, int nthreads) {
  for (int i = tid; i < 64; i += nthreads) rowmax[i] = 0u;
  __syncthreads();
  const int taps = planar ? 9 : 27, n = Cout * Cin * taps;
  for (int i = tid; i < n; i += nthreads) {
    const int co = transpose ? (i / taps) % Cout : i / (Cin * taps);
    atomicMax(&rowmax[co], __float_as_uint(fabsf(w[i])));
  }
  __syncthreads();
}
// half == 0: bf16 fragments; half == 1: scaled fp16 fragments + multipliers (wmul[cog * 16 + row] = 2^-k)
__device__ __forceinline__ void x3_pack_one(const float* __restrict__ w, uint4* __restrict__ wf, float* __restrict__ wmul, const unsigned* rowmax,
                                            int Cout, int Cin, int nsls, int ncog, int transpose, int planar, int half, int idx) {
  if (idx >= ncog * nsls * 64) return;
  const int lane = idx & 63, s = (idx >> 6) % nsls, cog = idx / (64 * nsls);
  const int co = cog * 16 + (lane & 15), kb = lane >> 4;
  const float mul = (half && co < Cout) ? x3_row_mul(rowmax[co]) : 1.f;
  if (half && s == 0 && kb == 0) wmul[cog * 16 + (lane & 15)] = 1.f / mul;
  unsigned short hi[8], lo[8];
  for (int j = 0; j < 8; ++j) {
    const int ncgs = (Cin + 3) / 4, P = 8 * s + 2 * kb + (j >> 2), cg = P % ncgs, tap = P / ncgs, ci = 4 * cg + (j & 3);
    const float v = tap < 27 ? x3_w_at(w, Cout, Cin, co, ci, tap, transpose, planar) * mul : 0.f;
    if (half) {
      hi[j] = x3_f16_bits(v);
      lo[j] = x3_f16_bits(v - (float)__builtin_bit_cast(_Float16, hi[j]));
    } else {
      x3_split(v, hi[j], lo[j]);
    }
  }
  auto pk = [](const unsigned short* h) {
    return make_uint4(h[0] | ((unsigned)h[1] << 16), h[2] | ((unsigned)h[3] << 16), h[4] | ((unsigned)h[5] << 16), h[6] | ((unsigned)h[7] << 16));
  };
  wf[((int64_t)(cog * nsls + s) * 2 + 0) * 64 + lane] = pk(hi);
  wf[((int64_t)(cog * nsls + s) * 2 + 1) * 64 + lane] = pk(lo);
}

// stem3d0 expanded in the consumer's staging (XSRC == 2 of conv3d_x3_kernel; costvol_stem.hip): the variant planes of
// ragmi_costvol_stem_fwd ([y][xi][Cm] per plane set, fp32) and what the combine kernel would apply to them
struct X3StemSrc {
  const float* ws;           // workspace of the planes (per sample: + b * ws_bstride)
  int64_t ws_bstride;
  int off_afull[4], off_aband[4], off_b0[4], off_b1[4];      // float offsets of the plane sets per z-border class
  int wband, wb1, u1_0;
  const float* scale;        // stem3d0's folded BatchNorm (may be null: identity)
  const float* shift;
  int relu;
  // one consumer 1x1x1 conv of stem3d0's output (4 output channels: cell 0's pre_preprocess), computed by the staging thread that
  // owns the voxel (ntail == 0: none)
  int ntail, tail_relu, tail_ch0, tail_g4;
  const float* tail_w;       // [4][Cm]
  const float* tail_scale;   // [4] (may be null)
  const float* tail_shift;
  float* tail_y;
  int64_t tail_bstride;
  int tail_rows;              // the fused tail of stem3d0 is computed by rows 12..15 of the matrix product (RAGMI_TAIL_ROWS), not by the staging thread
};

struct X3Extra {
  X3StemSrc src;             // XSRC == 2 only
  const uint4* wf[2];        // packed fragments per accumulator set (the section of the storage type: bf16 or scaled fp16)
  const float* wmul[2];      // fp16 section: per-output-channel multiplier 2^-k that undoes the weight scale (null for bf16 storage)
  int nseg, seg_len, nwork, bf16;   // bf16 != 0: bf16 activation storage (kernel instantiation selector)
  int ngrp, grp, nsplit;     // z-marching form: groups per sample; (column, segment) pairs per group of a sample, of which the last nsplit are two half items
  float dsd, dsh, dsw;       // down-sampling tails: (in - 1) / (in / 2 - 1) per axis (lin_scale, align_corners=True)
};
// fragment words (floats) of ONE section for a conv with these channel counts
inline int64_t x3_frag_words(int Cout, int Cin) {
  const int ncgs = (Cin + 3) / 4, nsls = (ncgs * 27 + 7) / 8, ncog = (Cout + 15) / 16;
  return (int64_t)ncog * nsls * 2 * 64 * 4;
}
// fills e.wf / e.wmul from the packed buffers of the call (after the fp32-MFMA section of each)
inline void x3_weight_sections(X3Extra& e, const K3Args& a, int nset, int dtype) {
  const int ngroups = (a.Cout + 3) / 4;
  for (int s = 0; s < nset; ++s) {
    const float* base = a.wp[s] + (int64_t)ngroups * a.nchunks[s] * PACK_PER_GC;
    const int64_t fw = x3_frag_words(a.Cout, a.nchunks[s] * 4);
    e.wf[s] = reinterpret_cast<const uint4*>(dtype == RAGMI_BF16 ? base : base + fw);
    e.wmul[s] = dtype == RAGMI_BF16 ? nullptr : base + 2 * fw;
  }
}


// conv3d_x3q.hip: the level-3 dual-cell form (four-slot ring, one barrier per plane step, immediate operand addresses)
bool xq_takes(const K3Args& a, int nset, int dtype);
int xq_launch(const K3Args& a, const X3Extra& e, int nset, dim3 grid, hipStream_t st);

// (the plane-stationary form measured in rounds 2-3 and not shipped is in the history: git show e045bcd:tools/experiments/conv3d_x3p.hip;
// its numbers are profiles/r03_x3p_investigation.md)

}  // namespace ragmi
