// MEASURED AND NOT SHIPPED (round 4; nothing builds this file) — profiles/r04_strict_fp32.md has the numbers:
//   strict fp32 pass, same box:   stem3d1 288 -> 340 us, level-3 dual cells 231 / 224 / 204 -> 251 / 258 / 225 us, level-6 dual 115 -> 147 us
// The z-marching structure stages every input plane once and has no staging VALU work, but v_mfma_f32_16x16x4_f32 spends 16 rows on
// 12 output channels (25 % of every product idle) while the shipped 4x4x1 form (conv3d_k3.h) packs them exactly: at equal FLOP rate
// per instruction the 4x4x1 kernel's floor is 113 us per dual cell against 150 us here, and that difference is what was measured.
// To try it again: it needs a packed section [cog][cg][tap][64] of plain fp32 weights (fz_frag_words) appended to the pack, the
// x3_plan_segments helper (the segmentation code of x3_launch), and a dispatch line `if (f32z_eligible(...)) return f32z_launch(...)`
// behind the x3 checks of ragmi_conv3d_k3_fwd_ex / _dual_fwd_ex.  All 262 GPU tests passed with it under RAGMI_X3=0.
//
// Strict fp32 (ABI dtype RAGMI_F32) 3x3x3 convolution of the big level-3 volumes: the z-marching structure of conv3d_x3.hip with
// EXACT fp32 arithmetic on v_mfma_f32_16x16x4_f32 — every output is one k-ordered chain of fp32 fused multiply-adds, like the
// 4x4x1 form of conv3d_k3.h (which this replaces for the shapes conv3d_x3.hip takes under RAGMI_F32X3).
//
// Why: the 4x4x1 box-tile kernel runs the level-3 dual cells at 227 us and stem3d1 at 288-302 us (0.50 / 0.55 of the fp32 matrix
// peak) — 3-D box tiles re-stage every input voxel 1.3-1.6x, weights are re-read per chunk, and its staging VALU work is serialised
// with the MFMAs.  Here an input plane is staged ONCE per column (ring of three planes in LDS, next plane in flight under the
// MFMAs), there is no operand split and no scale logic, and the K loop is one 4-byte LDS read + one MFMA per (tap, 4 channels).
//
// Mapping (mfma_f32_16x16x4f32: lane l holds A[row l&15][k = l>>4], B[k = l>>4][col l&15], D[row 4(l>>4)+reg][col l&15]):
// rows = 16 output channels, columns = 16 consecutive voxels along x, K = the 4 channels of one group at one tap.
// LDS: activations PLANAR per channel, [cg][ch][slot][y][x] (a lane quarter reads ITS channel: consecutive lanes = consecutive
// words; the channel stride is 16 mod 32 words, so the two quarters of a 32-lane access sit on disjoint banks); weight fragments
// [set][cg][tap][64 lanes] (A operand: w[co = lane & 15][ci = 4 cg + (lane >> 4)][tap]) staged once per workgroup.
#include "conv3d_x3_common.h"

namespace ragmi {

constexpr int FZ_RS = X3_HX;                          // halo row stride (words)
constexpr int FZ_PLS = 368;                           // words per (channel, slot) plane: >= HY * RS = 340, and 3 * 368 = 16 mod 32
static_assert(FZ_PLS >= X3_HY * FZ_RS && (3 * FZ_PLS) % 32 == 16, "plane stride");

template <int NCG, int NSET, bool TAILS>
__global__ __launch_bounds__(X3_THREADS, (NCG <= 3 ? 4 : 2)) void conv3d_f32z_kernel(K3Args a, X3Extra e) {
  constexpr int NCGS = NCG / NSET;
  constexpr int NPF = (NCG * X3_PL + X3_THREADS - 1) / X3_THREADS;
  constexpr int CGW = 4 * 3 * FZ_PLS;                          // words of one channel group (4 channels x 3 ring slots)
  extern __shared__ __attribute__((aligned(16))) float fz_lds[];
  float* const lx = fz_lds;                                     // [NCG][4 ch][3 slots][FZ_PLS]
  float* const lw = fz_lds + NCG * CGW;                         // [NSET][NCGS][27][64]
  float* const ltail = lw + NSET * NCGS * 27 * 64;              // [4][64]: A operand of the four tail products
  float* const par = ltail + 4 * 64;                            // scale[2][16] | shift[2][16] | tail scale[4 kb][4] | tail shift[4][4]
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, n = lane & 15, kb = lane >> 4;
  const int cog = blockIdx.y;
  const int HW = a.H * a.W;
  const int64_t DHW = (int64_t)HW * a.D;
  float act_floor = (a.relu & 1) ? 0.f : __builtin_nanf("");    // max(u, NaN) = u: the identity
  asm volatile("" : "+v"(act_floor));
  for (int i = tid; i < NSET * NCGS * 27 * 64; i += X3_THREADS) {
    const int set = i / (NCGS * 27 * 64), r = i % (NCGS * 27 * 64);
    lw[i] = reinterpret_cast<const float*>(e.wf[set])[(int64_t)cog * NCGS * 27 * 64 + r];
  }
  for (int i = tid; i < 32; i += X3_THREADS) {
    const int set = i >> 4, co = cog * 16 + (i & 15);
    const bool ok = set < NSET && co < a.Cout;
    par[i] = (ok && a.scale[set]) ? a.scale[set][co] : 1.f;
    par[32 + i] = (ok && a.shift[set]) ? a.shift[set][co] : 0.f;
  }
  // Fused consumer 1x1x1 convs ("tails"): out_t[k][voxel] = sum_c W_t[k][c] * v[c][voxel] as FOUR exact 16x16x4 products — product r
  // takes channel 4 kb + r of every lane quarter (a lane holds channels 4 kb .. 4 kb + 3 of its voxel: no value crosses lanes);
  // rows: tail 0 -> 0..3, tail 1 -> 4..7, ...
  if constexpr (TAILS) {
    if (tid < 64) {
      const int tl = n >> 2, k = n & 3;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int c = cog * 16 + 4 * kb + r;
        float wv = 0.f;
        if (tl < a.ntail && k < a.tail_cout[tl] && c < a.Cout) wv = a.tail_w[tl][k * a.Cout + c];
        ltail[r * 64 + lane] = wv;
      }
    }
    if (tid < 16) {
      const int tk = tid >> 2, r = tid & 3;
      const bool ok = tk < a.ntail && r < a.tail_cout[tk < 2 ? tk : 0];
      par[64 + tid] = (ok && a.tail_scale[tk < 2 ? tk : 0]) ? a.tail_scale[tk < 2 ? tk : 0][r] : 1.f;
      par[80 + tid] = (ok && a.tail_shift[tk < 2 ? tk : 0]) ? a.tail_shift[tk < 2 ? tk : 0][r] : 0.f;
    }
  }
  float pf[NPF][4];
  unsigned valid = 0;
  const float* const x = static_cast<const float*>(a.x);
  // this thread's halo elements of a column, located once per column (as conv3d_x3_kernel): the loads of a plane are
  // `uniform base + lane offset`, unconditional, clamped; zeros are substituted at the commit
  int voff[NPF];
  unsigned vmask = 0;
  auto locate = [&](int y0, int x0) {
    vmask = 0;
#pragma unroll
    for (int p = 0; p < NPF; ++p) {
      const int el = p * X3_THREADS + tid, cg = el / X3_PL, r = el % X3_PL;
      const int xx = r % X3_HX, yy = r / X3_HX;
      const int gy = y0 - 1 + yy, gx = x0 - 1 + xx;
      const bool ok = cg < NCG && (unsigned)gy < (unsigned)a.H && (unsigned)gx < (unsigned)a.W;
      vmask |= (ok ? 1u : 0u) << p;
      voff[p] = (int)(min(cg, NCG - 1) * 4 * DHW) + min(max(gy, 0), a.H - 1) * a.W + min(max(gx, 0), a.W - 1);
    }
  };
  auto prefetch = [&](const float* xb, int gz) {
    valid = (unsigned)gz < (unsigned)a.D ? vmask : 0u;
    const float* const pb = xb + (int64_t)min(max(gz, 0), a.D - 1) * HW;
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      const float* const pc = pb + c * DHW;            // wave-uniform
#pragma unroll
      for (int p = 0; p < NPF; ++p) pf[p][c] = pc[voff[p]];
    }
  };
  auto commit = [&](int slot) {
#pragma unroll
    for (int p = 0; p < NPF; ++p) {
      const int el = p * X3_THREADS + tid;
      if (el >= NCG * X3_PL) continue;
      const int cg = el / X3_PL, r = el % X3_PL;
      float* const d = lx + cg * CGW + slot * FZ_PLS + r;            // (r = yy * HX + xx and RS == HX)
#pragma unroll
      for (int c = 0; c < 4; ++c) d[c * 3 * FZ_PLS] = ((valid >> p) & 1u) ? pf[p][c] : 0.f;
    }
  };
  static_assert(X3_NT % 2 == 0 && FZ_RS == X3_HX, "tile geometry");
  // word base of each of this wave's column tiles: tile i = row (wave * NT + i) / 2, x half (i & 1); + this lane quarter's channel
  int vbt[X3_NT];
#pragma unroll
  for (int i = 0; i < X3_NT; ++i) {
    const int nt = wave * X3_NT + i;
    vbt[i] = (kb * 3 * FZ_PLS + (nt >> 1) * FZ_RS + (nt & 1) * 16 + n) * (int)sizeof(float);
    asm volatile("" : "+v"(vbt[i]));
  }
  const char* const lbytes = reinterpret_cast<const char*>(fz_lds);
  const int g = cog * 4 + kb, ngroups = (a.Cout + 3) >> 2;
  const int my_ych = g < ngroups ? a.y_ch[g < RAGMI_MAX_GROUPS ? g : 0] : 0;
  const int tsel = kb & 1;
  const int chunk = (e.nwork + 7) / 8;
  for (int j = blockIdx.x; j < chunk * 8; j += gridDim.x) {
    const int work = (j & 7) * chunk + (j >> 3);
    if ((j >> 3) >= chunk || work >= e.nwork) continue;
    int t = work, half = -1;
    const int b = t / (e.ngrp * (e.grp + e.nsplit));
    t %= e.ngrp * (e.grp + e.nsplit);
    const int gi = t / (e.grp + e.nsplit), k = t % (e.grp + e.nsplit);
    if (k < e.grp - e.nsplit) t = gi * e.grp + k;
    else { t = gi * e.grp + (e.grp - e.nsplit) + ((k - (e.grp - e.nsplit)) >> 1); half = (k - (e.grp - e.nsplit)) & 1; }
    const int x0 = (t % a.tiles_x) * X3_TX; t /= a.tiles_x;
    const int y0 = (t % a.tiles_y) * X3_TY; t /= a.tiles_y;
    const int seg = t;
    int zs = seg * e.seg_len, ze = min(a.D, zs + e.seg_len);
    if (half >= 0) { const int mid = zs + ((ze - zs + 1) >> 1); if (half) zs = mid; else ze = mid; }
    const float* xb = x + b * a.x_bstride;
    __syncthreads();                                   // the previous column's LDS reads are done (and the tables above are written)
    locate(y0, x0);
    prefetch(xb, zs - 1); commit((zs - 1 + 3) % 3);
    prefetch(xb, zs); commit(zs % 3);
    prefetch(xb, zs + 1);
    for (int z = zs; z < ze; ++z) {
      __syncthreads();                                 // plane z-2 (same ring slot as z+1) is no longer read
      commit((z + 1) % 3);
      __syncthreads();
      prefetch(xb, z + 2);                             // unconditional (clamped): straight-line loads ahead of the MFMA block
      __builtin_amdgcn_sched_barrier(0);
      f32x4 acc[NSET][X3_NT];
#pragma unroll
      for (int st = 0; st < NSET; ++st)
#pragma unroll
        for (int i = 0; i < X3_NT; ++i) acc[st][i] = f32x4{0.f, 0.f, 0.f, 0.f};
      const int ring = (z - 1 + 3) % 3;                // slot of plane z-1; plane z+dz-1 sits in slot (ring + dz) % 3
      int vz[3][X3_NT];                                // per dz: tile base + the slot's offset (3 x NT adds per plane: all the address work)
#pragma unroll
      for (int dz = 0; dz < 3; ++dz) {
        const int sb = ((ring + dz) % 3) * FZ_PLS * (int)sizeof(float);      // wave-uniform
#pragma unroll
        for (int i = 0; i < X3_NT; ++i) vz[dz][i] = vbt[i] + sb;
      }
#pragma unroll
      for (int st = 0; st < NSET; ++st)
#pragma unroll
        for (int cgl = 0; cgl < NCGS; ++cgl)
#pragma unroll
          for (int tap = 0; tap < 27; ++tap) {
            const float wa = lw[((st * NCGS + cgl) * 27 + tap) * 64 + lane];
            const int off = ((st * NCGS + cgl) * CGW + ((tap / 3) % 3) * FZ_RS + tap % 3) * (int)sizeof(float);     // compile time
#pragma unroll
            for (int i = 0; i < X3_NT; ++i) {
              const float bv = *reinterpret_cast<const float*>(lbytes + vz[tap / 9][i] + off);
              acc[st][i] = __builtin_amdgcn_mfma_f32_16x16x4f32(wa, bv, acc[st][i], 0, 0, 0);
            }
          }
      // epilogue: lane holds channels 4 g + reg (g = cog*4 + kb) of voxel n of each column tile
#pragma unroll
      for (int i = 0; i < X3_NT; ++i) {
        const int nt = wave * X3_NT + i;
        const int gy = y0 + (nt >> 1), gx = x0 + (nt & 1) * 16 + n;
        const bool inside = gy < a.H && gx < a.W;
        const int64_t vox = (int64_t)z * HW + gy * a.W + gx;
        float v[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          float sum = 0.f;
#pragma unroll
          for (int st = 0; st < NSET; ++st) {
            const float u = fmaxf(fmaf(acc[st][i][r], par[st * 16 + 4 * kb + r], par[32 + st * 16 + 4 * kb + r]), act_floor);
            sum = st == 0 ? u : sum + u;
          }
          v[r] = sum;
        }
        if (a.store_main && inside && g < ngroups) {
          float* py = static_cast<float*>(a.y) + b * a.y_bstride + (int64_t)my_ych * DHW + vox;
#pragma unroll
          for (int r = 0; r < 4; ++r) py[r * DHW] = v[r];
        }
        if constexpr (TAILS) {
          f32x4 tacc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
          for (int r = 0; r < 4; ++r) tacc = __builtin_amdgcn_mfma_f32_16x16x4f32(ltail[r * 64 + lane], v[r], tacc, 0, 0, 0);
          const int my_tail_cout = kb < a.ntail ? (tsel ? a.tail_cout[1] : a.tail_cout[0]) : 0;
          if (my_tail_cout > 0 && inside) {
            float* const my_tail = static_cast<float*>(tsel ? a.tail_y[1] : a.tail_y[0]);
            const int64_t tb = tsel ? a.tail_bstride[1] : a.tail_bstride[0];
            const int tch0 = tsel ? a.tail_ch0[1] : a.tail_ch0[0], trelu = tsel ? a.tail_relu[1] : a.tail_relu[0];
            const float4 tsc = *reinterpret_cast<const float4*>(par + 64 + 4 * kb), tsh = *reinterpret_cast<const float4*>(par + 80 + 4 * kb);
            const float sc4[4] = {tsc.x, tsc.y, tsc.z, tsc.w}, sh4[4] = {tsh.x, tsh.y, tsh.z, tsh.w};
            float* pt = my_tail + b * tb + (int64_t)tch0 * DHW + vox;
#pragma unroll
            for (int r = 0; r < 4; ++r)
              if (r < my_tail_cout) {
                const float u = fmaf(tacc[r], sc4[r], sh4[r]);
                pt[r * DHW] = trelu ? fmaxf(u, 0.f) : u;
              }
          }
        }
      }
    }
  }
}

// the shapes conv3d_x3.hip's z-marching form takes under RAGMI_F32X3, under the strict contract: fp32 storage, big level-3 volumes
bool f32z_eligible(const K3Args& a, int nset, int dtype) {
#ifdef RAGMI_F32Z_DISABLE      // A/B build: the 4x4x1 box-tile kernel everywhere
  return false;
#endif
  if (dtype != RAGMI_F32) return false;
  return x3_eligible(a, nset, RAGMI_F32X3);
}

template <int NCG, int NSET, bool TAILS>
static int f32z_launch_one(const K3Args& a, const X3Extra& e, dim3 grid, size_t lds, hipStream_t st) {
  static LaunchState state;
  const int slots = state.slots((const void*)conv3d_f32z_kernel<NCG, NSET, TAILS>, X3_THREADS, lds, 160 * 1024);
  if (slots <= 0) return fail(RAGMI_ELAUNCH, "conv3d_f32z: cannot raise the dynamic LDS limit");
  grid.x = (unsigned)std::max<int64_t>(1, std::min<int64_t>(grid.x, std::max(256, slots) / (int)grid.y));
  hipLaunchKernelGGL((conv3d_f32z_kernel<NCG, NSET, TAILS>), grid, dim3(X3_THREADS), lds, st, a, e);
  return check_launch("conv3d_f32z");
}

// a: as filled for conv3d_k3 (wp[s] = packed weights); the exact-fp32 fragments follow the split-operand sections
int f32z_launch(K3Args a, int nset, hipStream_t st) {
  X3Extra e{};
  const int ngroups = (a.Cout + 3) / 4;
  for (int s = 0; s < nset; ++s) {
    const float* base = a.wp[s] + (int64_t)ngroups * a.nchunks[s] * PACK_PER_GC + x3_packed_words(a.Cout, a.nchunks[s] * 4) -
                        fz_frag_words(a.Cout, a.nchunks[s] * 4);
    e.wf[s] = reinterpret_cast<const uint4*>(base);
  }
  const int ncg = a.nchunks[0] + (nset == 2 ? a.nchunks[1] : 0), ncgs = ncg / nset;
  const int ncog = (a.Cout + 15) / 16;
  const int rcp = x3_plan_segments(a, e, ncog, true);
  if (rcp != RAGMI_OK) return rcp;
  const size_t lds = ((size_t)ncg * 4 * 3 * FZ_PLS + (size_t)nset * ncgs * 27 * 64 + 4 * 64 + 96) * sizeof(float);
  RAGMI_REQUIRE(lds <= 160 * 1024, RAGMI_EUNSUPPORTED, "conv3d_f32z: tile does not fit the LDS");
  const dim3 grid((unsigned)std::min<int64_t>(e.nwork, 1 << 20), ncog);
#define RAGMI_FZ(NCG_, NSET_) (a.ntail > 0 ? f32z_launch_one<NCG_, NSET_, true>(a, e, grid, lds, st) : f32z_launch_one<NCG_, NSET_, false>(a, e, grid, lds, st))
  if (nset == 2) {
    switch (ncg) {
      case 2: return RAGMI_FZ(2, 2);
      case 4: return RAGMI_FZ(4, 2);
      default: return fail(RAGMI_EUNSUPPORTED, "conv3d_f32z: dual form with %d channel groups not instantiated", ncg);
    }
  }
  switch (ncg) {
    case 1: return RAGMI_FZ(1, 1);
    case 2: return RAGMI_FZ(2, 1);
    case 3: return RAGMI_FZ(3, 1);
    case 4: return RAGMI_FZ(4, 1);
    case 5: return RAGMI_FZ(5, 1);
    case 6: return RAGMI_FZ(6, 1);
    default: return fail(RAGMI_EUNSUPPORTED, "conv3d_f32z: %d channel groups not instantiated", ncg);
  }
#undef RAGMI_FZ
}

}  // namespace ragmi
