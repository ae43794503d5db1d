// K3 — fused 3x3x3 ConvBR_3d (+ Cell_3d running sum / channel concat) on the CDNA4 matrix cores.
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
#include <algorithm>

#include "common.h"

namespace ragmi {

constexpr int CK = 4;                        // input channels per LDS chunk
constexpr int NPAIR = CK * 27;               // (cin, tap) pairs per chunk = 108
constexpr int NVG = (NPAIR + 15) / 16;       // VGPRs per output group per chunk = 7
constexpr int PACK_PER_GC = NVG * 64;        // packed floats per (group, chunk) = 448

struct K3Args {
  const float* x;
  int64_t x_bstride;
  const float* wp;     // packed, already offset to the first group of this launch
  const float* scale;  // indexed by absolute output channel
  const float* shift;
  float* y;
  int64_t y_bstride;
  const float* res;
  int64_t res_bstride;
  int B, Cin, Cout, D, H, W;
  int nchunks;     // ceil(Cin / 4)
  int co0;         // first output channel of this launch (multiple of 4)
  int relu;
  int tiles_x, tiles_y, tiles_z;
  int y_ch[4];     // destination channel base of each group in this launch
  int res_ch[4];
};

// one pack element per thread: packed[((g*nchunks + ch)*NVG + v)*64 + lane]
__global__ void conv3d_k3_pack_kernel(const float* __restrict__ w, float* __restrict__ packed, int Cout, int Cin,
                                      int nchunks, int64_t total) {
  const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (idx >= total) return;
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
    if (ci < Cin && co < Cout) val = w[((int64_t)co * Cin + ci) * 27 + tap];
  }
  packed[idx] = val;
}

template <int G, int LOG_TX, int R>
__global__ __launch_bounds__(256, 2) void conv3d_k3_kernel(K3Args a) {
  constexpr int TX = 1 << LOG_TX;
  constexpr int YS = 64 / TX;      // lane sub-rows per wave
  constexpr int TY = YS * R;       // output rows per tile
  constexpr int TZ = 4;            // one z-plane per wave
  constexpr int HX = TX + 2, HY = TY + 2, HZ = TZ + 2;
  constexpr int TILE = CK * HZ * HY * HX;
  // staging map: thread -> (sy, zz, xx) of the halo; passes over compile-time (c, k): yy = k*SY + sy
  constexpr int SY = 256 / (HZ * HX);
  constexpr int KY = (HY + SY - 1) / SY;
  constexpr int NP = CK * KY;      // staging registers per thread
  static_assert(SY >= 1, "tile too wide for the staging map");
  __shared__ float tile[TILE];

  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int HW = a.H * a.W;
  const int64_t DHW = (int64_t)HW * a.D;
  const int ntiles = a.tiles_x * a.tiles_y * a.tiles_z * a.B;

  // compute-side lane geometry
  const int xl = lane & (TX - 1), ysub = lane >> LOG_TX;
  const float* rd = tile + (wave * HY + ysub * R) * HX + xl;  // lane's (c=0, dz=0, rr=0, dx=0) tap
  // staging-side thread geometry
  const int sxx = tid % HX, szz = (tid / HX) % HZ, ssy = tid / (HX * HZ);
  const bool sactive = ssy < SY;
  float* wr = tile + (szz * HY + ssy) * HX + sxx;             // + (c*HZ*HY + k*SY) * HX per pass

  f32x4 acc[R][G];
#pragma unroll
  for (int r = 0; r < R; ++r)
#pragma unroll
    for (int g = 0; g < G; ++g) acc[r][g] = (f32x4){0.f, 0.f, 0.f, 0.f};

  float st[NP];  // next stage's halo elements, in flight during the MFMA phase

  auto decode = [&](int t, int& b, int& x0, int& y0, int& z0) {
    const int tx_i = t % a.tiles_x; t /= a.tiles_x;
    const int ty_i = t % a.tiles_y; t /= a.tiles_y;
    const int tz_i = t % a.tiles_z;
    b = t / a.tiles_z;
    x0 = tx_i * TX; y0 = ty_i * TY; z0 = tz_i * TZ;
  };

  // issue the global loads of stage (t, chunk): always in-bounds (clamped); validity is applied at write time
  auto prefetch = [&](int t, int chunk) {
    int b, x0, y0, z0;
    decode(t, b, x0, y0, z0);
    const float* xb = a.x + (int64_t)b * a.x_bstride;
    const int gzc = min(max(z0 - 1 + szz, 0), a.D - 1), gxc = min(max(x0 - 1 + sxx, 0), a.W - 1);
    const int zx = gzc * HW + gxc;
    const int gy0 = y0 - 1 + ssy;
#pragma unroll
    for (int c = 0; c < CK; ++c) {
      const float* xc = xb + (int64_t)min(chunk * CK + c, a.Cin - 1) * DHW;   // wave-uniform base
#pragma unroll
      for (int k = 0; k < KY; ++k) {
        const int gyc = min(max(gy0 + k * SY, 0), a.H - 1);
        st[c * KY + k] = xc[(unsigned)(zx + gyc * a.W)];
      }
    }
  };

  // write the staged stage (t, chunk) into LDS, zeroing everything outside the volume / past Cin
  auto commit = [&](int t, int chunk) {
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
  auto load_weights = [&](int chunk) {
#pragma unroll
    for (int g = 0; g < G; ++g)
#pragma unroll
      for (int v = 0; v < NVG; ++v)
        wreg[g][v] = a.wp[((g * a.nchunks + chunk) * NVG + v) * 64 + lane];
  };

  auto epilogue = [&](int t, auto full_) {
    constexpr bool FULL = decltype(full_)::value;
    int b, x0, y0, z0;
    decode(t, b, x0, y0, z0);
    const int gz = z0 + wave, gx = x0 + xl, gy0 = y0 + ysub * R;
    if (!FULL && (gz >= a.D || gx >= a.W)) return;
    const unsigned off0 = (unsigned)(gz * HW + gy0 * a.W + gx);
    float* yb = a.y + (int64_t)b * a.y_bstride;
    const float* rb = a.res ? a.res + (int64_t)b * a.res_bstride : nullptr;
#pragma unroll
    for (int g = 0; g < G; ++g) {
#pragma unroll
      for (int m = 0; m < 4; ++m) {
        const int co = a.co0 + g * 4 + m;
        if (!FULL && co >= a.Cout) continue;
        const float sc = a.scale ? a.scale[co] : 1.f;
        const float sh = a.scale ? a.shift[co] : 0.f;
        float* yc = yb + (int64_t)(a.y_ch[g] + m) * DHW;                    // wave-uniform bases
        const float* rc = rb ? rb + (int64_t)(a.res_ch[g] + m) * DHW : nullptr;
#pragma unroll
        for (int r = 0; r < R; ++r) {
          if (!FULL && gy0 + r >= a.H) continue;
          float val = fmaf(acc[r][g][m], sc, sh);
          if (a.relu) val = fmaxf(val, 0.f);
          if (rc) val += rc[off0 + (unsigned)(r * a.W)];
          yc[off0 + (unsigned)(r * a.W)] = val;
        }
      }
    }
  };

  int t = blockIdx.x, chunk = 0;
  if (t >= ntiles) return;
  prefetch(t, 0);
  if (a.nchunks == 1) load_weights(0);

  while (true) {
    __syncthreads();  // every wave is done reading the previous stage's tile
    commit(t, chunk);
    __syncthreads();

    // next stage: same tile / next chunk, or this workgroup's next tile (grid-stride)
    int nt = t, nchunk = chunk + 1;
    if (nchunk == a.nchunks) { nchunk = 0; nt += gridDim.x; }
    const bool has_next = nt < ntiles;
    if (has_next) prefetch(nt, nchunk);   // global loads stay in flight under the MFMA phase below
    if (a.nchunks > 1) load_weights(chunk);

    static_for<CK>([&](auto c_) {
      constexpr int c = decltype(c_)::value;
      static_for<3>([&](auto dz_) {
        constexpr int dz = decltype(dz_)::value;
        float v[R + 2][3];
#pragma unroll
        for (int rr = 0; rr < R + 2; ++rr)
#pragma unroll
          for (int dx = 0; dx < 3; ++dx) v[rr][dx] = rd[((c * HZ + dz) * HY + rr) * HX + dx];
        static_for<3>([&](auto dy_) {
          constexpr int dy = decltype(dy_)::value;
          static_for<3>([&](auto dx_) {
            constexpr int dx = decltype(dx_)::value;
            constexpr int q = c * 27 + (dz * 3 + dy) * 3 + dx;
            static_for<G>([&](auto g_) {
              constexpr int g = decltype(g_)::value;
              static_for<R>([&](auto r_) {
                constexpr int r = decltype(r_)::value;
                acc[r][g] = __builtin_amdgcn_mfma_f32_4x4x1f32(wreg[g][q / 16], v[r + dy][dx], acc[r][g], 4, q % 16, 0);
              });
            });
          });
        });
      });
    });

    if (chunk == a.nchunks - 1) {
      int b, x0, y0, z0;
      decode(t, b, x0, y0, z0);
      const bool full = x0 + TX <= a.W && y0 + TY <= a.H && z0 + TZ <= a.D && a.co0 + 4 * G <= a.Cout;
      if (full) epilogue(t, std::true_type{}); else epilogue(t, std::false_type{});
#pragma unroll
      for (int r = 0; r < R; ++r)
#pragma unroll
        for (int g = 0; g < G; ++g) acc[r][g] = (f32x4){0.f, 0.f, 0.f, 0.f};
    }
    if (!has_next) break;
    t = nt;
    chunk = nchunk;
  }
}

// tile shape: widest x-tile whose padding waste is small; rows per lane sized so the grid still
// fills 256 CUs a few times over.  0: TX=32,R=4   1: TX=16,R=2   2: TX=8,R=1
static int choose_cfg(int B, int D, int H, int W) {
  const int64_t vol = (int64_t)B * D * H * W;
  auto waste = [&](int tx) { return (double)(ceil_div(W, tx) * tx) / W; };
  if (W > 16 && waste(32) <= waste(16) + 1e-9 && vol >= (1 << 20)) return 0;
  if (W > 8 && waste(16) <= waste(8) + 1e-9 && vol >= (1 << 17)) return 1;
  return 2;
}
// output groups per launch: at most 4, split evenly-ish (6 -> 3+3, 5 -> 3+2, 8 -> 4+4)
// persistent grid: as many workgroups as the chip holds at once (occupancy x CUs) stride over the tiles
template <class K>
static int persistent_slots(K kernel) {
  int dev = 0, cus = 256, per_cu = 2;
  hipDeviceProp_t prop;
  if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount > 0)
    cus = prop.multiProcessorCount;
  if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kernel, 256, 0) != hipSuccess || per_cu < 1) per_cu = 2;
  (void)hipGetLastError();
  return per_cu * cus;
}
static int split_groups(int left) { return left > 4 ? (left == 5 || left == 6 ? 3 : 4) : left; }

template <int LOG_TX, int R>
static int launch_cfg(K3Args a, int B, int ngroups, const int32_t* y_group_ch, const int32_t* res_group_ch,
                      hipStream_t s) {
  constexpr int TX = 1 << LOG_TX, TY = (64 / TX) * R;
  a.tiles_x = (int)ceil_div(a.W, TX);
  a.tiles_y = (int)ceil_div(a.H, TY);
  a.tiles_z = (int)ceil_div(a.D, 4);
  const int64_t nblk = (int64_t)a.tiles_x * a.tiles_y * a.tiles_z * B;
  if (nblk > 0x7fffffff) return fail(RAGMI_EUNSUPPORTED, "conv3d_k3: grid too large");
  const float* wp0 = a.wp;
  int g0 = 0;
  while (g0 < ngroups) {
    const int left = ngroups - g0;
    const int G = split_groups(left);
    a.wp = wp0 + (int64_t)g0 * a.nchunks * PACK_PER_GC;
    a.co0 = g0 * 4;
    for (int g = 0; g < G; ++g) {
      a.y_ch[g] = y_group_ch ? y_group_ch[g0 + g] : (g0 + g) * 4;
      a.res_ch[g] = res_group_ch ? res_group_ch[g0 + g] : (g0 + g) * 4;
    }
    static const int slots[4] = {persistent_slots(conv3d_k3_kernel<1, LOG_TX, R>), persistent_slots(conv3d_k3_kernel<2, LOG_TX, R>),
                                 persistent_slots(conv3d_k3_kernel<3, LOG_TX, R>), persistent_slots(conv3d_k3_kernel<4, LOG_TX, R>)};
    dim3 grid((unsigned)std::min<int64_t>(nblk, slots[G - 1])), blk(256);
    switch (G) {
      case 1: hipLaunchKernelGGL((conv3d_k3_kernel<1, LOG_TX, R>), grid, blk, 0, s, a); break;
      case 2: hipLaunchKernelGGL((conv3d_k3_kernel<2, LOG_TX, R>), grid, blk, 0, s, a); break;
      case 3: hipLaunchKernelGGL((conv3d_k3_kernel<3, LOG_TX, R>), grid, blk, 0, s, a); break;
      default: hipLaunchKernelGGL((conv3d_k3_kernel<4, LOG_TX, R>), grid, blk, 0, s, a); break;
    }
    g0 += G;
  }
  return check_launch("conv3d_k3");
}

}  // namespace ragmi

extern "C" int64_t ragmi_conv3d_k3_packed_elems(int Cout, int Cin) {
  if (Cout <= 0 || Cin <= 0) return 0;
  return (int64_t)((Cout + 3) / 4) * ((Cin + ragmi::CK - 1) / ragmi::CK) * ragmi::PACK_PER_GC;
}

extern "C" int ragmi_conv3d_k3_pack(const void* weight, void* packed, int Cout, int Cin, int dtype, void* stream) {
  using namespace ragmi;
  RAGMI_REQUIRE(weight && packed, RAGMI_EINVAL, "conv3d_k3_pack: null pointer");
  RAGMI_REQUIRE(Cout > 0 && Cin > 0, RAGMI_EINVAL, "conv3d_k3_pack: non-positive size");
  RAGMI_REQUIRE(dtype == RAGMI_F32, RAGMI_EUNSUPPORTED, "conv3d_k3_pack: dtype %d not built", dtype);
  const int64_t total = ragmi_conv3d_k3_packed_elems(Cout, Cin);
  hipLaunchKernelGGL(conv3d_k3_pack_kernel, dim3((unsigned)ceil_div(total, 256)), dim3(256), 0,
                     static_cast<hipStream_t>(stream), (const float*)weight, (float*)packed, Cout, Cin,
                     (Cin + CK - 1) / CK, total);
  return check_launch("conv3d_k3_pack");
}

extern "C" int ragmi_conv3d_k3_fwd(const void* x, int64_t x_bstride, const void* packed_weight, const void* scale,
                                   const void* shift, int relu, void* y, int64_t y_bstride, const int32_t* y_group_ch,
                                   const void* res, int64_t res_bstride, const int32_t* res_group_ch, int B, int Cin,
                                   int Cout, int D, int H, int W, int dtype, void* stream) {
  using namespace ragmi;
  RAGMI_REQUIRE(x && packed_weight && y, RAGMI_EINVAL, "conv3d_k3: null pointer");
  RAGMI_REQUIRE((scale == nullptr) == (shift == nullptr), RAGMI_EINVAL, "conv3d_k3: scale/shift must both be given or both NULL");
  RAGMI_REQUIRE(B > 0 && Cin > 0 && Cout > 0 && D > 0 && H > 0 && W > 0, RAGMI_EINVAL, "conv3d_k3: non-positive size");
  RAGMI_REQUIRE(dtype == RAGMI_F32, RAGMI_EUNSUPPORTED, "conv3d_k3: dtype %d not built", dtype);
  const int ngroups = (Cout + 3) / 4;
  RAGMI_REQUIRE(ngroups <= RAGMI_MAX_GROUPS, RAGMI_EUNSUPPORTED, "conv3d_k3: Cout %d > %d", Cout, 4 * RAGMI_MAX_GROUPS);
  RAGMI_REQUIRE((int64_t)D * H * W < (1ll << 30), RAGMI_EUNSUPPORTED, "conv3d_k3: volume too large (32-bit plane offsets)");
  K3Args a{};
  a.x = (const float*)x; a.x_bstride = x_bstride;
  a.wp = (const float*)packed_weight; a.scale = (const float*)scale; a.shift = (const float*)shift;
  a.y = (float*)y; a.y_bstride = y_bstride;
  a.res = (const float*)res; a.res_bstride = res_bstride;
  a.B = B; a.Cin = Cin; a.Cout = Cout; a.D = D; a.H = H; a.W = W;
  a.nchunks = (Cin + CK - 1) / CK; a.relu = relu;
  hipStream_t s = static_cast<hipStream_t>(stream);
  switch (choose_cfg(B, D, H, W)) {
    case 0: return launch_cfg<5, 4>(a, B, ngroups, y_group_ch, res_group_ch, s);
    case 1: return launch_cfg<4, 2>(a, B, ngroups, y_group_ch, res_group_ch, s);
    default: return launch_cfg<3, 1>(a, B, ngroups, y_group_ch, res_group_ch, s);
  }
}

extern "C" int ragmi_conv3d_k3_plan(int Cout, int B, int D, int H, int W, int32_t* log_tx, int32_t* rows_per_lane,
                                    int32_t* launch_groups, int32_t max_launches) {
  using namespace ragmi;
  RAGMI_REQUIRE(log_tx && rows_per_lane && launch_groups, RAGMI_EINVAL, "conv3d_k3_plan: null pointer");
  RAGMI_REQUIRE(Cout > 0 && B > 0 && D > 0 && H > 0 && W > 0, RAGMI_EINVAL, "conv3d_k3_plan: non-positive size");
  static const int cfgs[3][2] = {{5, 4}, {4, 2}, {3, 1}};
  const int c = choose_cfg(B, D, H, W);
  *log_tx = cfgs[c][0];
  *rows_per_lane = cfgs[c][1];
  int n = 0, g0 = 0;
  const int ngroups = (Cout + 3) / 4;
  while (g0 < ngroups) {
    const int G = split_groups(ngroups - g0);
    if (n < max_launches) launch_groups[n] = G;
    ++n;
    g0 += G;
  }
  return n;
}
