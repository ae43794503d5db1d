// K3 (kernel template) — fused 3x3x3 ConvBR_3d (+ Cell_3d running sum / channel concat) on the CDNA4 matrix cores.
// Reference: ConvBR_3d src/automl/operations_3d.py:31-47; call sites stem3d0/1
// (src/models/rag_model.py:234-235, 341-343), Cell_3d._ops (:134-137, 160-176), last_3_3d (:269).
//
// Design (gfx950, fp32 exact):
//  * The contraction runs on v_mfma_f32_4x4x1_16b_f32: 16 independent 4x4 outer products
//    per instruction = "64 voxels x 4 output channels += w[4] * x[64]" for one (cin, tap).
//    B operand: lane l holds the input value of ITS voxel (thread-per-voxel, NCDHW-natural,
//    coalesced).  D: lane l holds the 4 output channels of its voxel.  Output-channel counts
//    of 4/8/12/16 map with zero padding waste (a 16x16x4 tile would idle 25-75% of its N).
//  * A operand via CBSZ=4/ABID broadcast: all 16 blocks take A from block ABID, so ONE VGPR
//    holds the weight fragments of 16 different (cin, tap) pairs and the 108 pairs of a
//    4-channel chunk live in 7 VGPRs per output group — weights are register-resident for
//    the whole tile and cost no LDS traffic (pre-packed by ragmi_conv3d_k3_pack).
//  * Input halo tile (4 ch x 6 x (TY+2) x (TX+2)) staged through LDS once per tile; each lane
//    then reads (R+2) rows x 3 dx per (cin, dz) and reuses them for its R output rows.
//  * Epilogue fuses folded BatchNorm (scale/shift), ReLU, the Cell_3d running sum (res, may
//    alias y) and torch.cat (per-group destination channel), so none of them is a pass.
//  * Measured ceiling of the 4x4x1 form: 134 TFLOP/s (tools/probe_mfma.hip) vs 157 spec.
#pragma once
#include <algorithm>
#include <cstdlib>

#include "common.h"

namespace ragmi {

constexpr int CK = 4;                        // input channels per LDS chunk
constexpr int NPAIR = CK * 27;               // (cin, tap) pairs per chunk = 108
constexpr int NVG = (NPAIR + 15) / 16;       // VGPRs per output group per chunk = 7
constexpr int PACK_PER_GC = NVG * 64;        // packed floats per (group, chunk) = 448

// fp32-MFMA section of the packed weights: element idx of packed[groups][nchunks][NVG][64]; lane 4a+m of VGPR v holds
// w[co = 4g + m][pair q = 16v + a] (q = c_local * 27 + tap).  transpose: the data-gradient weight (Cout <-> Cin swapped,
// taps flipped); planar: the source is a 2-D 3x3 weight embedded in the middle z-slice.
__device__ __forceinline__ float k3_pack_value(const float* __restrict__ w, int Cout, int Cin, int nchunks, int64_t idx, int transpose,
                                               int planar) {
  const int lane = (int)(idx & 63);
  int64_t t = idx >> 6;
  const int v = (int)(t % NVG);
  t /= NVG;
  const int ch = (int)(t % nchunks);
  const int g = (int)(t / nchunks);
  const int a = lane >> 2, m = lane & 3;
  const int q = 16 * v + a;  // pair index: c_local * 27 + tap
  float val = 0.f;
  if (q < NPAIR) {
    const int ci = ch * CK + q / 27, tap = q % 27, co = g * 4 + m;
    if (ci < Cin && co < Cout) {
      const int taps = planar ? 9 : 27;
      int tt = planar ? tap - 9 : tap;                   // planar: only dz == 1 (taps 9..17) is non-zero
      if (tt >= 0 && tt < taps) {
        if (transpose) tt = taps - 1 - tt;
        val = transpose ? w[((int64_t)ci * Cout + co) * taps + tt] : w[((int64_t)co * Cin + ci) * taps + tt];
      }
    }
  }
  return val;
}

struct K3Args {
  const void* x;          // activations: float or bf16_t per the kernel's storage type T
  int64_t x_bstride;
  const float* wp[2];     // per accumulator set: packed [groups][nchunks[s]][NVG][64]
  const float* scale[2];  // per set, indexed by output channel
  const float* shift[2];
  void* y;
  int64_t y_bstride;
  const void* res;
  int64_t res_bstride;
  int B, Cin, Cout, D, H, W;
  int nchunks[2];  // input-channel chunks feeding set 0, then set 1 (consecutive channels of x)
  int relu;
  int tiles_x, tiles_y, tiles_z;
  // up to two consumer 1x1x1 ConvBR_3d fused into the epilogue ("tails"): y_t[j] = act(bn_t(sum_c W_t[j][c] * out[c])).
  // Needs every output channel of a voxel in one lane: one split (gridDim.y == 1), Cout == 4*G <= 16, tail Cout <= 4.
  int ntail;
  int store_main;              // 0: the conv's own output is consumed only by the tails and is not written
  const float* tail_w[2];      // [tail_cout][Cout] row-major
  const float* tail_scale[2];
  const float* tail_shift[2];
  void* tail_y[2];
  int64_t tail_bstride[2];
  int tail_ch0[2], tail_cout[2], tail_relu[2];
  // DOWN-SAMPLING tails (z-marching split-operand form only, conv3d_x3.hip): y_t = act(bn_t(trilinear x0.5, align_corners=True, of
  // W_t * out)) written at HALF resolution [D/2, H/2, W/2] — the 1x1x1 ConvBR of a consumer cell that works one level down
  // (Cell_3d with downup_sample = -1, rag_model.py:146-155), computed conv-first in the producer's epilogue.  They take the slots
  // ntail .. ntail + ndown - 1 of the tail product (4 output channels each).
  int ndown;
  const float* down_w[2];
  const float* down_scale[2];
  const float* down_shift[2];
  void* down_y[2];
  int64_t down_bstride[2];
  int down_ch0[2], down_cout[2], down_relu[2];
  int down_f32;                   // bit k: down-sampling tail k stores fp32 although the launch's storage is bf16 (RAGMI_TAIL_F32)
  int tail_g4;     // 1: the full-resolution tails write channel-group-interleaved tensors (RAGMI_TAIL_G4; include/rag_amd.h)
  int w_in_lds;    // 1: the workgroup's weights (all chunks, both sets) are cached in LDS behind the tile
  int y_ch[RAGMI_MAX_GROUPS];    // destination channel base of each output group
  int res_ch[RAGMI_MAX_GROUPS];
};

constexpr int K3_MAX_WLDS_BYTES = 36 * 1024;   // weight cache budget per workgroup (stem3d0: 6 chunks x 3 groups = 32 KB)

// G output groups (4 channels each) per workgroup, selected by blockIdx.y; NSET accumulator sets:
// NSET=2 fuses two sibling-group convolutions that feed the same destinations from two inputs
// (Cell_3d: out = relu(bn_a(conv_a(s0))) + relu(bn_b(conv_b(s1)))) so the running sum never
// round-trips through HBM.  WPS = waves per SIMD the register budget is sized for.
//
// VCO > 0 selects the VALU form for Cout = VCO <= 2 (last_3_3d has Cout = 1): a 4x4x1 MFMA would idle 3 of its
// 4 rows there, while v_fma with wave-uniform (SGPR) weights does the same per-lane work at 2.5x the issue
// rate.  Same tiles, staging pipeline and epilogue; a.wp[0] is then the RAW weight [Cout][Cin][27].
//
// T = activation storage type (float, or bf16_t for RAGMI_BF16): halo loads convert to fp32 on their way into LDS and
// the epilogue rounds once at the store; LDS tile, MFMA operands, accumulators, BN and the tails are fp32 either way.
// FLAT: the tile form for depth-1 volumes (the 2-D Feature-Net convolutions, rag_model.py:47-111): the four waves split the tile's
// rows instead of its z-planes (TZ = 1, 4x the rows) — with one z-plane per wave three of four waves would idle there.
// TO: storage type of the MAIN output (default T).  float with T = bf16_t keeps a result that is consumed at full precision in fp32 —
// `mat`, the cost the soft-argmin reads — while its input stays bf16.
template <class T, int G, int LOG_TX, int R, int NSET, int WPS, int VCO = 0, bool FLAT = false, class TO = T>
__global__ __launch_bounds__(256, WPS) void conv3d_k3_kernel(K3Args a) {
  static_assert(VCO == 0 || (G == 1 && NSET == 1 && VCO <= 4), "VALU form: one output group, one set");
  constexpr int TX = 1 << LOG_TX;
  constexpr int YS = 64 / TX;      // lane sub-rows per wave
  constexpr int TY = YS * R * (FLAT ? 4 : 1);   // output rows per tile
  constexpr int TZ = FLAT ? 1 : 4; // one z-plane per wave (FLAT: one plane, waves stacked along y)
  constexpr int HX = TX + 2, HY = TY + 2, HZ = TZ + 2;
  constexpr int TILE = CK * HZ * HY * HX;
  // staging map: thread -> (sy, zz, xx) of the halo; passes over compile-time (c, k): yy = k*SY + sy
  constexpr int SY = 256 / (HZ * HX);
  constexpr int KY = (HY + SY - 1) / SY;
  constexpr int NP = CK * KY;      // staging registers per thread
  static_assert(SY >= 1, "tile too wide for the staging map");
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* tile = smem;              // [CK][HZ][HY][HX] halo tile
  float* wlds = smem + TILE;       // optional weight cache: set 0 block then set 1 block, [g][chunk][NVG][64]

  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int HW = a.H * a.W;
  const int64_t DHW = (int64_t)HW * a.D;
  const int ntiles = a.tiles_x * a.tiles_y * a.tiles_z * a.B;
  const int gbase = blockIdx.y * G;                       // first output group of this workgroup
  const int nch0 = a.nchunks[0];
  const int nch = nch0 + (NSET == 2 ? a.nchunks[1] : 0);  // stages per tile

  // compute-side lane geometry
  const int xl = lane & (TX - 1), ysub = lane >> LOG_TX;
  const int wz = FLAT ? 0 : wave, wy = FLAT ? wave * YS * R : 0;   // this wave's plane / first row inside the tile
  const float* rd = tile + (wz * HY + wy + ysub * R) * HX + xl;  // lane's (c=0, dz=0, rr=0, dx=0) tap
  // staging-side thread geometry
  const int sxx = tid % HX, szz = (tid / HX) % HZ, ssy = tid / (HX * HZ);
  const bool sactive = ssy < SY;
  float* wr = tile + (szz * HY + ssy) * HX + sxx;             // + (c*HZ*HY + k*SY) * HX per pass

  // folded-BN parameters of this workgroup's output channels, staged once into LDS: the epilogue must not
  // re-read them from global memory (its stores may alias them, which would serialise every element)
  __shared__ __attribute__((aligned(16))) float bnp[NSET][2][G * 4];
  // (compile-time set / tail index in unrolled loops: a descriptor array of the kernel arguments indexed by a lane-dependent value
  // costs a vector load of the pointer plus a dependent one of the value, serially, in front of the first tile)
#pragma unroll
  for (int s = 0; s < NSET; ++s) {
    const float* const psc = a.scale[s];
    const float* const psh = a.shift[s];
    if (tid < G * 4) {
      const int j = tid, co = gbase * 4 + j;
      const bool ok = psc != nullptr && co < a.Cout;
      bnp[s][0][j] = ok ? psc[co] : 1.f;   // identity affine when there is no BN: fma(x, 1, 0) == x exactly
      bnp[s][1][j] = ok ? psh[co] : 0.f;
    }
  }
  // tail weights + folded BN, staged like bnp (alias-free reads in the epilogue)
  __shared__ __attribute__((aligned(16))) float tailw[2][4][16];
  __shared__ __attribute__((aligned(16))) float tailbn[2][2][4];
  if (a.ntail > 0 && tid < 4 * 16) {
    const int j = tid / 16, c = tid % 16;
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      const float* const pw = a.tail_w[t];
      const float* const psc = a.tail_scale[t];
      const float* const psh = a.tail_shift[t];
      const bool ok = t < a.ntail && j < a.tail_cout[t] && c < a.Cout;
      tailw[t][j][c] = ok ? pw[j * a.Cout + c] : 0.f;
      if (c < 2) {
        const bool okb = t < a.ntail && j < a.tail_cout[t] && psc != nullptr;
        tailbn[t][c][j] = okb ? (c == 0 ? psc[j] : psh[j]) : (c == 0 ? 1.f : 0.f);
      }
    }
  }
  int ych[G], rch[G];
#pragma unroll
  for (int g = 0; g < G; ++g) {
    ych[g] = a.y_ch[gbase + g];
    rch[g] = a.res_ch[gbase + g];
  }
  const bool has_res = a.res != nullptr;
  const bool do_relu = (a.relu & 1) != 0;

  f32x4 acc[NSET][R][G];
#pragma unroll
  for (int s = 0; s < NSET; ++s)
#pragma unroll
    for (int r = 0; r < R; ++r)
#pragma unroll
      for (int g = 0; g < G; ++g) acc[s][r][g] = (f32x4){0.f, 0.f, 0.f, 0.f};

  float st[NP];  // next stage's halo elements, in flight during the MFMA phase

  // tile coordinates are decoded ONCE per tile (three integer divisions by run-time values are a few hundred
  // dependent cycles) and carried in scalars: `cur` for the stage being computed, `nxt` for the one in flight
  struct TC { int b, x0, y0, z0; };
  auto decode_tc = [&](int t) -> TC {
    const int tx_i = t % a.tiles_x; t /= a.tiles_x;
    const int ty_i = t % a.tiles_y; t /= a.tiles_y;
    const int tz_i = t % a.tiles_z;
    return TC{t / a.tiles_z, tx_i * TX, ty_i * TY, tz_i * TZ};
  };
  auto decode = [&](const TC& c, int& b, int& x0, int& y0, int& z0) { b = c.b; x0 = c.x0; y0 = c.y0; z0 = c.z0; };

  // A stage whose halo lies inside the volume and whose 4 channels exist needs no clamping and no zero fill.
  // The vector-issue port is the scarce resource here (a 4x4x1 MFMA holds it for most of its 8 cycles, so every
  // VALU instruction in staging is serialised with the MFMAs): the interior path uses wave-uniform (SGPR) row
  // bases and immediate LDS offsets, i.e. one VMEM + one DS instruction per element and NO VALU work.
  auto interior = [&](const TC& t, int chunk) -> bool {
    int b, x0, y0, z0;
    decode(t, b, x0, y0, z0);
    return x0 >= 1 && x0 + TX + 1 <= a.W && y0 >= 1 && y0 + TY + 1 <= a.H && z0 >= 1 && z0 + TZ + 1 <= a.D &&
           chunk * CK + CK <= a.Cin;
  };

  // issue the global loads of stage (t, chunk): always in-bounds (clamped); validity is applied at write time
  auto prefetch = [&](const TC& t, int chunk) {
    int b, x0, y0, z0;
    decode(t, b, x0, y0, z0);
    const T* xb = static_cast<const T*>(a.x) + (int64_t)b * a.x_bstride;
    if (SY == 1 && interior(t, chunk)) {
      const unsigned zx = (unsigned)((z0 - 1 + szz) * HW + x0 - 1 + sxx);   // the only per-lane quantity
      const T* xr = xb + (int64_t)(chunk * CK) * DHW + (int64_t)(y0 - 1) * a.W;   // wave-uniform
#pragma unroll
      for (int c = 0; c < CK; ++c)
#pragma unroll
        for (int k = 0; k < KY; ++k) st[c * KY + k] = ld(xr + (int64_t)c * DHW + (int64_t)k * a.W + (sactive ? zx : 0u));
      return;
    }
    int gzc = min(max(z0 - 1 + szz, 0), a.D - 1), gxc = min(max(x0 - 1 + sxx, 0), a.W - 1);
#ifdef RAGMI_DIAG   // profiling builds only (make DIAG=1): the shipped library has no switch that changes what is computed
    if (a.relu & 0x800) gxc = min(x0 + sxx, a.W - 1);          // line-aligned halo rows (wrong data)
    if (a.relu & 0x1000) gzc = min(z0, a.D - 1);               // every lane reads the same z-plane
#endif
    const int zx = gzc * HW + gxc;
    const int gy0 = y0 - 1 + ssy;
#pragma unroll
    for (int c = 0; c < CK; ++c) {
      const T* xc = xb + (int64_t)min(chunk * CK + c, a.Cin - 1) * DHW;   // wave-uniform base
#pragma unroll
      for (int k = 0; k < KY; ++k) {
        const int gyc = min(max(gy0 + k * SY, 0), a.H - 1);
        st[c * KY + k] = ld(xc + (unsigned)(zx + gyc * a.W));
      }
    }
  };

  // write the staged stage (t, chunk) into LDS, zeroing everything outside the volume / past Cin
  auto commit = [&](const TC& t, int chunk) {
    if (SY == 1 && interior(t, chunk)) {
      if (sactive) {
#pragma unroll
        for (int c = 0; c < CK; ++c)
#pragma unroll
          for (int k = 0; k < KY; ++k) wr[(c * HZ * HY + k) * HX] = st[c * KY + k];
      }
      return;
    }
    int b, x0, y0, z0;
    decode(t, b, x0, y0, z0);
    const bool zx_ok = sactive && (unsigned)(z0 - 1 + szz) < (unsigned)a.D && (unsigned)(x0 - 1 + sxx) < (unsigned)a.W;
    const int gy0 = y0 - 1 + ssy;
#pragma unroll
    for (int c = 0; c < CK; ++c) {
      const bool c_ok = chunk * CK + c < a.Cin;
#pragma unroll
      for (int k = 0; k < KY; ++k) {
        const bool ok = zx_ok && c_ok && (unsigned)(gy0 + k * SY) < (unsigned)a.H;
        if (sactive && k * SY + ssy < HY) wr[(c * HZ * HY + k * SY) * HX] = ok ? st[c * KY + k] : 0.f;
      }
    }
  };

  float wreg[G][NVG];
  auto load_weights = [&](int chunk) {   // chunk is the tile-level stage index
    const int set = (NSET == 2 && chunk >= nch0) ? 1 : 0;
    const int lc = chunk - (set ? nch0 : 0), nc = a.nchunks[set];
    if (a.w_in_lds) {
      const float* wl = wlds + (set ? G * nch0 * PACK_PER_GC : 0) + lane;
#pragma unroll
      for (int g = 0; g < G; ++g)
#pragma unroll
        for (int v = 0; v < NVG; ++v) wreg[g][v] = wl[((g * nc + lc) * NVG + v) * 64];
    } else {
      const float* wp = a.wp[set];
#pragma unroll
      for (int g = 0; g < G; ++g)
#pragma unroll
        for (int v = 0; v < NVG; ++v)
          wreg[g][v] = wp[(((gbase + g) * nc + lc) * NVG + v) * 64 + lane];
    }
  };

  auto epilogue = [&](const TC& t, auto full_, auto res_) {
    constexpr bool FULL = decltype(full_)::value;
    constexpr bool RES = decltype(res_)::value;
    constexpr int NM = VCO > 0 ? VCO : 4;   // channels of a group actually computed
    int b, x0, y0, z0;
    decode(t, b, x0, y0, z0);
    const int gz = z0 + wz, gx = x0 + xl, gy0 = y0 + wy + ysub * R;
    if (!FULL && (gz >= a.D || gx >= a.W)) return;
    const unsigned off0 = (unsigned)(gz * HW + gy0 * a.W + gx);
    TO* yb = static_cast<TO*>(a.y) + (int64_t)b * a.y_bstride;
    const T* rb = RES ? static_cast<const T*>(a.res) + (int64_t)b * a.res_bstride : nullptr;
    const bool tails = VCO == 0 && a.ntail > 0;
    const bool store_main = a.store_main != 0;
    // row-outer: only one output row's G*4 activated values are live at a time (they feed the tails)
#pragma unroll
    for (int r = 0; r < R; ++r) {
      if (!FULL && gy0 + r >= a.H) continue;
      const unsigned off = off0 + (unsigned)(r * a.W);
      float fin[G * 4];
      float rv[G * 4];
      if (RES) {   // residual loads of this row first, then the arithmetic and the stores
#pragma unroll
        for (int g = 0; g < G; ++g)
#pragma unroll
          for (int m = 0; m < NM; ++m) {
            const bool ok = FULL || (gbase + g) * 4 + m < a.Cout;
            rv[g * 4 + m] = ok ? ld(rb + (int64_t)(rch[g] + m) * DHW + off) : 0.f;
          }
      }
#pragma unroll
      for (int g = 0; g < G; ++g) {
        f32x4 sc[NSET], sh[NSET];
#pragma unroll
        for (int s = 0; s < NSET; ++s) {
          sc[s] = *reinterpret_cast<const f32x4*>(&bnp[s][0][g * 4]);   // LDS broadcast reads
          sh[s] = *reinterpret_cast<const f32x4*>(&bnp[s][1][g * 4]);
        }
#pragma unroll
        for (int m = 0; m < 4; ++m) {
          if (m >= NM) { fin[g * 4 + m] = 0.f; continue; }
          float val = fmaf(acc[0][r][g][m], sc[0][m], sh[0][m]);
          if (do_relu) val = fmaxf(val, 0.f);
          if (NSET == 2) {
            float v1 = fmaf(acc[NSET - 1][r][g][m], sc[NSET - 1][m], sh[NSET - 1][m]);
            if (do_relu) v1 = fmaxf(v1, 0.f);
            val += v1;
          }
          if (RES) val += rv[g * 4 + m];
          fin[g * 4 + m] = val;
          const bool ch_ok = FULL || (gbase + g) * 4 + m < a.Cout;
#ifdef RAGMI_DIAG
          if (ch_ok && store_main && (!(a.relu & 0x100) || val == 12345.678f))   // 0x100: skip stores
#else
          if (ch_ok && store_main)
#endif
            ragmi::st(yb + (int64_t)(ych[g] + m) * DHW + off, val);
        }
      }
      if (tails) {
#pragma unroll
        for (int tt = 0; tt < 2; ++tt) {
          if (tt >= a.ntail) break;
          T* tb = static_cast<T*>(a.tail_y[tt]) + (int64_t)b * a.tail_bstride[tt] + (int64_t)a.tail_ch0[tt] * DHW;
          const f32x4 tsc = *reinterpret_cast<const f32x4*>(&tailbn[tt][0][0]);
          const f32x4 tsh = *reinterpret_cast<const f32x4*>(&tailbn[tt][1][0]);
          const bool trelu = a.tail_relu[tt] != 0;
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            if (j >= a.tail_cout[tt]) break;
            float sacc = 0.f;
#pragma unroll
            for (int g = 0; g < G; ++g) {
              const f32x4 w = *reinterpret_cast<const f32x4*>(&tailw[tt][j][g * 4]);
#pragma unroll
              for (int m = 0; m < 4; ++m) sacc = fmaf(w[m], fin[g * 4 + m], sacc);
            }
            sacc = fmaf(sacc, tsc[j], tsh[j]);
            if (trelu) sacc = fmaxf(sacc, 0.f);
            ragmi::st(tb + (int64_t)j * DHW + off, sacc);
          }
        }
      }
    }
  };

  // 12 (channel, dz) blocks per stage; the (R+2) x 3 LDS operands of block i+1 are fetched into the other
  // half of vbuf while block i's 9*G*R MFMAs issue, so no MFMA waits on an LDS read it has just issued
  auto load_v = [&](float (&v)[R + 2][3], auto blk_) {
    constexpr int blk = decltype(blk_)::value;
    constexpr int c = blk / 3, dz = blk % 3;
#pragma unroll
    for (int rr = 0; rr < R + 2; ++rr)
#pragma unroll
      for (int dx = 0; dx < 3; ++dx) v[rr][dx] = rd[((c * HZ + dz) * HY + rr) * HX + dx];
  };
  auto mfma_block = [&](auto set_) {
    constexpr int S = decltype(set_)::value;
    float vbuf[2][R + 2][3];
    load_v(vbuf[0], std::integral_constant<int, 0>{});
    static_for<CK * 3>([&](auto blk_) {
      constexpr int blk = decltype(blk_)::value;
      constexpr int c = blk / 3, dz = blk % 3;
      if constexpr (blk + 1 < CK * 3) load_v(vbuf[(blk + 1) & 1], std::integral_constant<int, blk + 1>{});
      static_for<3>([&](auto dy_) {
        constexpr int dy = decltype(dy_)::value;
        static_for<3>([&](auto dx_) {
          constexpr int dx = decltype(dx_)::value;
          constexpr int q = c * 27 + (dz * 3 + dy) * 3 + dx;
          static_for<G>([&](auto g_) {
            constexpr int g = decltype(g_)::value;
            static_for<R>([&](auto r_) {
              constexpr int r = decltype(r_)::value;
              acc[S][r][g] = __builtin_amdgcn_mfma_f32_4x4x1f32(wreg[g][q / 16], vbuf[blk & 1][r + dy][dx], acc[S][r][g], 4, q % 16, 0);
            });
          });
        });
      });
    });
  };

  // VALU form of the same block: weights are wave-uniform scalar loads from the raw [Cout][Cin][27] tensor
  auto valu_block = [&](int chunk) {
    // raw weights cached in LDS, the nine taps of one (channel, dz) padded to twelve floats: + ((co * Cin + c) * 3 + dz) * 12 + tap.
    // Three 16-byte broadcast reads fetch a (c, dz) row of weights; read one by one they were 9 of the 27 LDS instructions that
    // feed 36 FMAs
    const float* wc = wlds + chunk * (CK * 3 * 12);
    float vbuf[2][R + 2][3];
    load_v(vbuf[0], std::integral_constant<int, 0>{});
    static_for<CK * 3>([&](auto blk_) {
      constexpr int blk = decltype(blk_)::value;
      if constexpr (blk + 1 < CK * 3) load_v(vbuf[(blk + 1) & 1], std::integral_constant<int, blk + 1>{});
#pragma unroll
      for (int co = 0; co < (VCO > 0 ? VCO : 1); ++co) {
        const float4* wq = reinterpret_cast<const float4*>(wc + co * a.Cin * 36 + blk * 12);
        const float4 w0 = wq[0], w1 = wq[1], w2 = wq[2];
        const float w9[9] = {w0.x, w0.y, w0.z, w0.w, w1.x, w1.y, w1.z, w1.w, w2.x};
        static_for<9>([&](auto tap_) {
          constexpr int dy = decltype(tap_)::value / 3, dx = decltype(tap_)::value % 3;
#pragma unroll
          for (int r = 0; r < R; ++r) acc[0][r][0][co] = fmaf(w9[dy * 3 + dx], vbuf[blk & 1][r + dy][dx], acc[0][r][0][co]);
        });
      }
    });
  };

  // XCD-aware persistent schedule: workgroups are dealt round-robin over the 8 XCDs (blockIdx % 8), each with
  // its own L2.  Every XCD gets one CONTIGUOUS slab of the tile index space (x fastest, then y, z) and its
  // workgroups stride through it together, so halos shared by neighbouring tiles are L2 hits.  Speed only:
  // any placement gives the same results.
  int t, tend, tstep;
  if (gridDim.x % 8 == 0) {
    const int xcd = blockIdx.x & 7, per = (ntiles + 7) / 8;
    t = xcd * per + (blockIdx.x >> 3);
    tend = min((xcd + 1) * per, ntiles);
    tstep = gridDim.x >> 3;
  } else {
    t = blockIdx.x; tend = ntiles; tstep = gridDim.x;
  }
  int chunk = 0;
  if (t >= tend) return;

#ifdef RAGMI_DIAG
  const bool diag_nomfma = (a.relu & 0x200) != 0, diag_nostage = (a.relu & 0x400) != 0;
#else
  constexpr bool diag_nomfma = false, diag_nostage = false;
#endif
  TC cur = decode_tc(t), nxt = cur;
  if (!diag_nostage) prefetch(cur, 0);
  if constexpr (VCO > 0) {
    // VALU form: the whole raw weight tensor [Cout][Cin][27] (<= a few KB) lives in LDS for the kernel's lifetime;
    // global loads of it inside the loop could not be scalar (stores may alias) and would stall every 16 weights
    const int n = a.Cout * a.Cin * 36;         // [co][c][dz][12]: nine taps + three pad floats
    for (int i = tid; i < n; i += 256) {
      const int k = i % 12, row = i / 12;        // row = (co * Cin + c) * 3 + dz
      wlds[i] = k < 9 ? a.wp[0][row * 9 + k] : 0.f;
    }
  } else if (a.w_in_lds) {
    // the G groups of this workgroup are one contiguous block per set in the packed array
#pragma unroll
    for (int s = 0; s < NSET; ++s) {
      const int n = G * a.nchunks[s] * PACK_PER_GC;
      const float* src = a.wp[s] + (int64_t)gbase * a.nchunks[s] * PACK_PER_GC;
      float* dst = wlds + (s ? G * nch0 * PACK_PER_GC : 0);
      for (int i = tid * 4; i < n; i += 1024) *reinterpret_cast<float4*>(dst + i) = *reinterpret_cast<const float4*>(src + i);
    }
  }
  if (VCO == 0 && nch == 1 && !a.w_in_lds) load_weights(0);

  while (true) {
    __syncthreads();  // every wave is done reading the previous stage's tile
    if (!diag_nostage) commit(cur, chunk);
    __syncthreads();

    // next stage: same tile / next chunk, or this workgroup's next tile
    int nt = t, nchunk = chunk + 1;
    if (nchunk == nch) { nchunk = 0; nt += tstep; }
    const bool has_next = nt < tend;
    if (has_next && nt != t) nxt = decode_tc(nt);
    // weights that come from global memory are requested BEFORE the halo prefetch: vector memory returns in order, so waiting
    // for them must not have to wait for the prefetch that is meant to stay in flight under the MFMA phase
    if (VCO == 0 && !diag_nomfma && nch > 1 && !a.w_in_lds) load_weights(chunk);
    if (has_next && !diag_nostage) prefetch(nxt, nchunk);

    if (diag_nomfma) {
    } else if constexpr (VCO > 0) {
      valu_block(chunk);
    } else {
      if (a.w_in_lds) load_weights(chunk);   // from LDS: visible after the barrier above
      if (NSET == 2 && chunk >= nch0) mfma_block(std::integral_constant<int, NSET - 1>{});
      else mfma_block(std::integral_constant<int, 0>{});
    }

    if (chunk == nch - 1) {
      const bool full = cur.x0 + TX <= a.W && cur.y0 + TY <= a.H && cur.z0 + TZ <= a.D &&
                        (VCO > 0 ? VCO == a.Cout : (gbase + G) * 4 <= a.Cout);
      if (full) {
        if (has_res) epilogue(cur, std::true_type{}, std::true_type{}); else epilogue(cur, std::true_type{}, std::false_type{});
      } else {
        if (has_res) epilogue(cur, std::false_type{}, std::true_type{}); else epilogue(cur, std::false_type{}, std::false_type{});
      }
#pragma unroll
      for (int s = 0; s < NSET; ++s)
#pragma unroll
        for (int r = 0; r < R; ++r)
#pragma unroll
          for (int g = 0; g < G; ++g) acc[s][r][g] = (f32x4){0.f, 0.f, 0.f, 0.f};
    }
    if (!has_next) break;
    t = nt;
    cur = nxt;
    chunk = nchunk;
  }
}

// persistent grid: as many workgroups as the chip holds at once (occupancy x CUs) stride over the tiles — LaunchState::slots

// output groups per workgroup: the largest of {4,3,2,1} dividing the group count (12 -> 4, 6 -> 3)
inline int split_groups(int ngroups) {
  for (int g = 4; g > 1; --g)
    if (ngroups % g == 0) return g;
  return 1;
}

template <class T, int G, int LOG_TX, int R, int NSET, int WPS, int VCO = 0, bool FLAT = false, class TO = T>
static int launch_one(K3Args a, int64_t ntiles, int nsplits, hipStream_t s) {
  constexpr int TX = 1 << LOG_TX, TY = (64 / TX) * R * (FLAT ? 4 : 1);
  constexpr size_t tile_bytes = (size_t)CK * (FLAT ? 3 : 6) * (TY + 2) * (TX + 2) * sizeof(float);
  const size_t wbytes = (size_t)G * (a.nchunks[0] + (NSET == 2 ? a.nchunks[1] : 0)) * PACK_PER_GC * sizeof(float);
  // cache the weights in LDS when they fit the budget: only for multi-stage tiles (policy 2; a profiling build, make DIAG=1,
  // reads RAGMI_K3_WLDS = 0 never / 1 whenever they fit, and the RAGMI_K3_DIAG_NOSTORE bit mask 1 no stores, 2 no MFMA block,
  // 4 no staging)
#ifdef RAGMI_DIAG
  static const int diag_nostore = [] { const char* e = getenv("RAGMI_K3_DIAG_NOSTORE"); return e ? atoi(e) : 0; }();
  if (diag_nostore) a.relu |= (diag_nostore << 8);
  static const int policy = [] { const char* e = getenv("RAGMI_K3_WLDS"); return e ? atoi(e) : 2; }();
#else
  constexpr int policy = 2;
#endif
  const int nstages = a.nchunks[0] + (NSET == 2 ? a.nchunks[1] : 0);
  // budget: 36 KB next to the big tiles, or whatever still leaves two workgroups per CU (small tiles of the deep levels: their
  // 48-output weights are ~57 KB; read from global per stage they sit in the vmcnt queue BEHIND the halo prefetch, so waiting
  // for them drains the prefetch that should fly under the MFMAs)
  const size_t wbudget = std::max<size_t>(K3_MAX_WLDS_BYTES, tile_bytes < 78 * 1024 ? 78 * 1024 - tile_bytes : 0);
  a.w_in_lds = (VCO == 0 && policy != 0 && wbytes <= wbudget && (policy == 1 || nstages > 1)) ? 1 : 0;
  const size_t lds = tile_bytes + (VCO > 0 ? (size_t)a.Cout * a.Cin * 36 * sizeof(float) : (a.w_in_lds ? wbytes : 0));
  // per-device, mutex-guarded launch state of this instantiation: the dynamic-LDS attribute applies to the CURRENT device only, and
  // forward (main thread) and backward (autograd worker thread) both come through here
  static LaunchState state;
  const int slots = state.slots((const void*)conv3d_k3_kernel<T, G, LOG_TX, R, NSET, WPS, VCO, FLAT, TO>, 256, lds,
                                std::max<size_t>(tile_bytes + K3_MAX_WLDS_BYTES, 80 * 1024));
  if (slots <= 0) return fail(RAGMI_ELAUNCH, "conv3d_k3: cannot raise the dynamic LDS limit");
  int64_t gx = std::max<int64_t>(1, std::min<int64_t>(ntiles, slots / nsplits));
  if (gx >= 8) gx -= gx % 8;   // the XCD-aware schedule wants a multiple of 8 workgroups per split
  hipLaunchKernelGGL((conv3d_k3_kernel<T, G, LOG_TX, R, NSET, WPS, VCO, FLAT, TO>), dim3((unsigned)gx, (unsigned)nsplits), dim3(256), lds, s, a);
  return RAGMI_OK;
}

// one tile configuration: sets the tile counts and launches with G = split_groups(ngroups)
template <class T, int LOG_TX, int R, int NSET, int WPS, bool FLAT = false>
static int launch_cfg(K3Args a, int ngroups, hipStream_t s) {
  constexpr int TX = 1 << LOG_TX, TY = (64 / TX) * R * (FLAT ? 4 : 1);
  a.tiles_x = (int)ceil_div(a.W, TX);
  a.tiles_y = (int)ceil_div(a.H, TY);
  a.tiles_z = (int)ceil_div(a.D, FLAT ? 1 : 4);
  const int64_t ntiles = (int64_t)a.tiles_x * a.tiles_y * a.tiles_z * a.B;
  if (ntiles > 0x7fffffff) return fail(RAGMI_EUNSUPPORTED, "conv3d_k3: grid too large");
  const int G = split_groups(ngroups);
  const int nsplits = ngroups / G;
  int rc;
  switch (G) {
    case 1: rc = launch_one<T, 1, LOG_TX, R, NSET, WPS, 0, FLAT>(a, ntiles, nsplits, s); break;
    case 2: rc = launch_one<T, 2, LOG_TX, R, NSET, WPS, 0, FLAT>(a, ntiles, nsplits, s); break;
    case 3: rc = launch_one<T, 3, LOG_TX, R, NSET, WPS, 0, FLAT>(a, ntiles, nsplits, s); break;
    default: rc = launch_one<T, 4, LOG_TX, R, NSET, WPS, 0, FLAT>(a, ntiles, nsplits, s); break;
  }
  if (rc != RAGMI_OK) return rc;
  return check_launch("conv3d_k3");
}

// instantiated one per translation unit (conv3d_k3_inst_*.hip, fp32 and bf16 storage) so the build parallelises
#define RAGMI_K3_DECL(name) \
  int launch_k3_##name##_f32(const K3Args& a, int ngroups, hipStream_t s); \
  int launch_k3_##name##_bf16(const K3Args& a, int ngroups, hipStream_t s)
RAGMI_K3_DECL(s1_cfg0);   // TX=32 R=4
RAGMI_K3_DECL(s1_cfg1);   // TX=16 R=2
RAGMI_K3_DECL(s1_cfg2);   // TX=8  R=1
RAGMI_K3_DECL(s2_cfg0);   // dual, TX=32 R=2
RAGMI_K3_DECL(s2_cfg1);
RAGMI_K3_DECL(s2_cfg2);
#undef RAGMI_K3_DECL
// bf16x3 form (conv3d_x3.hip): fp32 accuracy on the bf16 matrix cores, used for the big level-3 volumes
int64_t x3_packed_words(int Cout, int Cin);
// both sections of the packed weights (fp32-MFMA section of `total_k3` floats, then the bf16x3 fragments) in ONE launch
// (`all` = false: the fp32-MFMA section only)
// conv3d_c1.hip: the single-output-channel form (the head's last_3_3d at full resolution)
bool c1_eligible(const K3Args& a, int dtype, int y_dtype);
int c1_launch(const K3Args& k, int dtype, int y_dtype, hipStream_t st);
int pack_both(const float* w, float* packed, int64_t total_k3, int Cout, int Cin, int transpose, int planar, bool all, hipStream_t s);
bool x3_eligible(const K3Args& a, int nset, int dtype);
int x3_g4_caps(const K3Args& a, int nset, int dtype);      // G4 forms (include/rag_amd.h) the kernel this call lands on takes
struct X3StemSrc;
int x3_launch(K3Args a, int nset, int dtype, hipStream_t st, const X3StemSrc* src = nullptr);
// argument marshalling of the 3x3x3 entry points (conv3d.hip), shared with ragmi_costvol_stem_conv3d_fwd (costvol_stem.hip)
int fill_common(K3Args& a, const void* x, int64_t x_bstride, void* y, int64_t y_bstride, const int32_t* y_group_ch, const void* res,
                int64_t res_bstride, const int32_t* res_group_ch, int B, int Cin, int Cout, int D, int H, int W, int relu);
int fill_tails(K3Args& a, int store_main, int ntail, const ragmi_tail_t* tails, int Cout);
// deep-level bf16x3 form (8 / 16 input channels per set, box tiles): levels 6 and 12
// depth-1 volumes (the Feature Net's 2-D convolutions) on the split-operand form: conv2d_x3.hip
bool x2d_eligible(const K3Args& a, int nset, int dtype);
int x2d_launch(K3Args a, int nset, int dtype, hipStream_t st);
bool x3d_eligible(const K3Args& a, int nset, int dtype);
int x3d_launch(K3Args a, int nset, int dtype, hipStream_t st);
int launch_k3_valu_f32(const K3Args& a, int cfg, hipStream_t s);          // Cout <= 2, raw weights
int launch_k3_valu_bf16(const K3Args& a, int cfg, hipStream_t s);
int launch_k3_valu_bf16_f32out(const K3Args& a, int cfg, hipStream_t s);   // bf16 input, fp32 main output

template <class T, int LOG_TX, int R, int VCO, class TO = T>
static int launch_cfg_valu(K3Args a, hipStream_t s) {
  constexpr int TX = 1 << LOG_TX, TY = (64 / TX) * R;
  a.tiles_x = (int)ceil_div(a.W, TX);
  a.tiles_y = (int)ceil_div(a.H, TY);
  a.tiles_z = (int)ceil_div(a.D, 4);
  const int64_t ntiles = (int64_t)a.tiles_x * a.tiles_y * a.tiles_z * a.B;
  if (ntiles > 0x7fffffff) return fail(RAGMI_EUNSUPPORTED, "conv3d_k3: grid too large");
  // one output channel fits 128 VGPRs without spilling: four waves per SIMD hide the LDS-read chains of the VALU form better
  const int rc = launch_one<T, 1, LOG_TX, R, 1, (VCO == 1 ? 4 : 2), VCO, false, TO>(a, ntiles, 1, s);
  if (rc != RAGMI_OK) return rc;
  return check_launch("conv3d_k3_small");
}

}  // namespace ragmi
