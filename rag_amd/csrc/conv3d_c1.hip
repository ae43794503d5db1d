// 3x3x3 convolution with ONE output channel on the vector ALUs: the Matching-Net head's `last_3_3d`
// (ConvBR_3d(12, 1, 3, 1, 1, bn=False, relu=False), src/models/rag_model.py:269, used :361-365) at the cost volume's full
// resolution — 324 multiply-adds per output voxel and nothing for the matrix cores to hold on to (one output row).
//
// Input-stationary z-marching.  A workgroup (512 threads) owns a 64 x 32 (y, x) tile and a segment of ZSEG output planes; a thread owns
// 4 consecutive x of one row in every plane of the segment (ZSEG x 4 accumulators).  For one input channel at a time the 27
// weights sit in registers while the channel's input planes z-1 .. z+ZSEG of the tile (66 x 34 halo, fetched once per workgroup:
// global -> registers -> LDS, double-buffered in LDS, THREE stages deep in registers — a stage is ~0.3 us of arithmetic, an HBM
// round trip under load several times that) pass through: a plane feeds the three output planes around it — 108 FMAs per
// thread from 18 operand values (9 LDS reads).
// The generic small-Cout kernel (conv3d_k3_kernel<.., VALU>) re-reads every input voxel 1.65x from HBM (3-D box tiles with
// halos on all sides) and reloads weights per chunk: 78 us at the headline shape against the 27 us the input read costs.
#include "conv3d_k3.h"

namespace ragmi {

constexpr int C1_TY = 64, C1_TX = 32, C1_THREADS = 512;
// LDS tile: rows y0-1 .. y0+64 of columns x0 .. x0+31 — a row is exactly 32 banks, so the 16-byte operand reads of 8 lanes per
// row x 8 rows per wave are bank-conflict free (with the halo columns inside the rows, stride 40, they were 3-way conflicted and
// the kernel LDS-bound: 63 us of arithmetic + LDS against 17 us of arithmetic alone); the halo columns x0-1 / x0+32 live in `edge`
constexpr int C1_ROWS = C1_TY + 2, C1_COLS = C1_TX;
constexpr int C1_TILE = C1_ROWS * C1_COLS, C1_EDGE = C1_ROWS * 2;
// staging elements per stage: A = 4 consecutive x of an interior row (64 rows x 8: one per thread, 16 bytes);
// B = one voxel of the halo (rows y0-1 and y0+64: 2 x 32; columns x0-1 and x0+32 of all 66 rows: 132), 4 bytes, threads 0..195
constexpr int C1_NB = 2 * C1_TX + 2 * C1_ROWS;
static_assert(C1_TY * (C1_TX / 4) == C1_THREADS && C1_NB <= C1_THREADS, "staging map");
constexpr int C1_MAX_CIN = 64;
constexpr int64_t C1_MIN_VOXELS = 1 << 18;
#ifndef C1_ZSEG_OVERRIDE
constexpr int C1_ZSEG = 4;
#else
constexpr int C1_ZSEG = C1_ZSEG_OVERRIDE;
#endif

struct C1Args {
  const void* x;
  const float* w;          // [Cin][27], the reference layout of a [1, Cin, 3, 3, 3] weight
  const float* scale;      // optional folded BatchNorm of the single output channel
  const float* shift;
  void* y;
  int64_t x_bstride, y_bstride;
  int y_ch0, relu;
  int B, Cin, D, H, W;
  int tiles_x, tiles_y, nseg, nwork;
};

template <class T, class TO, int ZSEG>
__global__ __launch_bounds__(C1_THREADS, 2) void conv3d_c1_kernel(C1Args a) {
  constexpr int NP = ZSEG + 2;                            // input planes per channel
  // pipeline stages per channel: NP rounded up to a multiple of 6, so that the LDS double buffer (s % 2) and the three register
  // stages (s % 3) are indexed at compile time; the padding stages are dead (no loads, no arithmetic, one barrier)
  constexpr int NPS = (NP + 5) / 6 * 6;
  __shared__ __attribute__((aligned(16))) float tile[2][C1_TILE];
  __shared__ float edge[2][C1_EDGE];
  __shared__ __attribute__((aligned(16))) float wl[C1_MAX_CIN * 28];
  const int tid = threadIdx.x, tx = tid & 7, ty = tid >> 3;
  // XCD-aware order: workgroup j runs on XCD j % 8; every XCD walks one contiguous chunk of the x-fastest tile list, so tiles that
  // share halo rows / columns (and the depth segments that share two planes) meet in the same L2
  const int chunk = (a.nwork + 7) / 8, j = blockIdx.x;
  int t = (j & 7) * chunk + (j >> 3);
  if ((j >> 3) >= chunk || t >= a.nwork) return;
  const int x0 = (t % a.tiles_x) * C1_TX; t /= a.tiles_x;
  const int y0 = (t % a.tiles_y) * C1_TY; t /= a.tiles_y;
  const int seg = t % a.nseg, b = t / a.nseg;
  const int zs = seg * ZSEG, ze = min(a.D, zs + ZSEG);
  const int HW = a.H * a.W;
  const int64_t DHW = (int64_t)HW * a.D;
  for (int i = tid; i < a.Cin * 28; i += C1_THREADS) wl[i] = (i % 28) < 27 ? a.w[(i / 28) * 27 + i % 28] : 0.f;
  // this thread's two staging elements, located once (the same for every plane and channel).  The loads are UNCONDITIONAL
  // (addresses clamped into the tensor, zeros substituted at the commit): with a branch around them the compiler's wait counts
  // collapse to vmcnt(0) and every stage waits out a full memory round trip (measured: 0.6 us per stage, 90 us per launch with
  // the arithmetic removed)
  int offA, dstA, offB, dstB;
  bool okA, okB;
  {
    const int r = 1 + (tid >> 3), q = tid & 7, gy = y0 - 1 + r, gx = x0 + 4 * q;
    dstA = r * C1_COLS + 4 * q;
    okA = gy < a.H && gx + 3 < a.W;                        // W % 4 == 0: all four or none
    offA = okA ? gy * a.W + gx : 0;
  }
  {
    const int e = min(tid, C1_NB - 1);
    int r, gx;
    if (e < 2 * C1_TX) { r = e < C1_TX ? 0 : C1_ROWS - 1; gx = x0 + (e & (C1_TX - 1)); dstB = r * C1_COLS + (e & (C1_TX - 1)); }
    else { r = (e - 2 * C1_TX) >> 1; const int side = (e - 2 * C1_TX) & 1; gx = side ? x0 + C1_TX : x0 - 1; dstB = C1_TILE + r * 2 + side; }
    const int gy = y0 - 1 + r;
    okB = tid < C1_NB && (unsigned)gy < (unsigned)a.H && (unsigned)gx < (unsigned)a.W;
    offB = okB ? gy * a.W + gx : 0;
  }
  const bool hasB = tid < C1_NB;
  float* const pB[2] = {dstB < C1_TILE ? tile[0] + dstB : edge[0] + (dstB - C1_TILE), dstB < C1_TILE ? tile[1] + dstB : edge[1] + (dstB - C1_TILE)};
  const T* const xb = static_cast<const T*>(a.x) + (int64_t)b * a.x_bstride;
  float pfA[3][4], pfB[3];
  // stage s = (channel c, input plane zi): plane z' = zs - 1 + zi.  Planes outside the volume (first / last segment) or past the
  // segment's last output plane + 1 contribute zeros: their (clamped) loads are discarded and their arithmetic skipped.
  auto plane_live = [&](int zi) { const int z = zs - 1 + zi; return zi < NP && z >= 0 && z <= min(ze, a.D - 1); };
  auto prefetch = [&](int c, int zi, int slot) {           // zi, slot: compile time after unrolling
#ifdef C1_DIAG_NOLOAD
    return;
#endif
    const T* const pc = xb + (int64_t)min(c, a.Cin - 1) * DHW + (int64_t)min(max(zs - 1 + zi, 0), a.D - 1) * HW;
    ld4(pc + offA, pfA[slot]);
    pfB[slot] = ld(pc + offB);
  };
  auto commit = [&](int buf, int slot) {
    *reinterpret_cast<float4*>(tile[buf] + dstA) = okA ? make_float4(pfA[slot][0], pfA[slot][1], pfA[slot][2], pfA[slot][3]) : make_float4(0.f, 0.f, 0.f, 0.f);
    if (hasB) *pB[buf] = okB ? pfB[slot] : 0.f;
  };
  float acc[ZSEG][4];
#pragma unroll
  for (int k = 0; k < ZSEG; ++k)
#pragma unroll
    for (int i = 0; i < 4; ++i) acc[k][i] = 0.f;
  // stages 0, 1, 2 are in flight before the first multiply; stage s + 3 is requested when stage s starts
  prefetch(0, 0, 0);
  prefetch(0, 1, 1);
  prefetch(0, 2, 2);
  commit(0, 0);
  __syncthreads();
  // operand window of this thread: rows ty .. ty+2, x0 + 4tx - 1 .. + 4: one aligned 16-byte read per row; the two outer
  // columns come from the neighbouring lanes' reads (DPP row shifts: lanes of one tile row are 8 consecutive lanes), except at
  // the row's ends, which read the halo column (one 4-byte read per row, bank-conflict free)
  const float4* const my4 = reinterpret_cast<const float4*>(&tile[0][0]) + ty * (C1_COLS / 4) + tx;
  const float* const myedge = &edge[0][0] + ty * 2 + (tx == 7 ? 1 : 0);
  const bool left_edge = tx == 0, right_edge = tx == 7;
  for (int c = 0; c < a.Cin; ++c) {
    float w[28];
#pragma unroll
    for (int i = 0; i < 28; i += 4) {
      const float4 q = *reinterpret_cast<const float4*>(wl + c * 28 + i);       // same address in every lane: a broadcast read
      w[i] = q.x; w[i + 1] = q.y; w[i + 2] = q.z; w[i + 3] = q.w;
    }
    static_for<NPS>([&](auto zi_) {
      constexpr int zi = decltype(zi_)::value;             // stage s = c * NPS + zi: s % 2 == zi % 2, s % 3 == zi % 3
      // stage s + 3 -> the register slot stage s has just left (its data went to LDS at the end of stage s - 1)
      if constexpr ((zi + 3) % NPS < NP) prefetch(zi + 3 < NPS ? c : c + 1, (zi + 3) % NPS, zi % 3);
#ifdef C1_DIAG_NOFMA
      if (false) {
#else
      if (plane_live(zi)) {
#endif
        float in[3][6];
#pragma unroll
        for (int r = 0; r < 3; ++r) {
#ifdef C1_DIAG_NOLDS
          const float4 m = make_float4(pfA[0][0] + r, pfA[1][1], pfA[2][2], pfA[0][3] + zi);
          const float e = pfB[0];
#else
          const float4 m = my4[(zi & 1) * (C1_TILE / 4) + r * (C1_COLS / 4)];
          const float e = myedge[(zi & 1) * C1_EDGE + r * 2];
#endif
          const float fl = __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(m.w), 0x111, 0xF, 0xF, false));   // row_shr:1
          const float fr = __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(m.x), 0x101, 0xF, 0xF, false));   // row_shl:1
          in[r][0] = left_edge ? e : fl; in[r][1] = m.x; in[r][2] = m.y; in[r][3] = m.z; in[r][4] = m.w; in[r][5] = right_edge ? e : fr;
        }
        static_for<3>([&](auto dz_) {
          constexpr int dz = decltype(dz_)::value, k = zi - dz;         // output plane zs + k = z' + 1 - dz
          if constexpr (zi < NP && k >= 0 && k < ZSEG) {
            if (zs + k < ze) {                                           // (wave-uniform: a partial last segment)
#pragma unroll
              for (int dy = 0; dy < 3; ++dy)
#pragma unroll
                for (int dx = 0; dx < 3; ++dx)
#pragma unroll
                  for (int i = 0; i < 4; ++i) acc[k][i] = fmaf(w[dz * 9 + dy * 3 + dx], in[dy][i + dx], acc[k][i]);
            }
          }
        });
      }
      // stage s + 1 -> LDS (the buffer stage s - 1 read: every thread is past it since the last barrier)
      if constexpr ((zi + 1) % NPS < NP) commit((zi + 1) & 1, (zi + 1) % 3);
#ifndef C1_DIAG_NOBARRIER
      __syncthreads();
#endif
    });
  }
  const float sc = a.scale ? a.scale[0] : 1.f, sh = a.shift ? a.shift[0] : 0.f;
  TO* const yb = static_cast<TO*>(a.y) + (int64_t)b * a.y_bstride + (int64_t)a.y_ch0 * DHW;
  const int gx = x0 + 4 * tx, gy = y0 + ty;
  if (gy < a.H && gx < a.W) {
#pragma unroll
    for (int k = 0; k < ZSEG; ++k) {
      if (zs + k >= ze) continue;
      float v[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const float u = fmaf(acc[k][i], sc, sh);
        v[i] = a.relu ? fmaxf(u, 0.f) : u;
      }
      st4(yb + (int64_t)(zs + k) * HW + gy * a.W + gx, v);
    }
  }
}

// ---------------------------------------------------------------------------------------------------------------
// The head's last two steps as ONE kernel: mat = last_3_3d(Upsample(scale 2, trilinear, align_corners=True)(y6))
// (rag_model.py:357-365: `upsample_6` followed by the 12 -> 1 channel 3x3x3 convolution).  The upsampled tensor (164 MB at the
// headline shape: 43 us to write, then read back 1.4x by the convolution, which that read bounds at ~56 us) never exists: a
// thread interpolates ITS OWN operand window from the level-6 tensor (20 MB, L2 / memory-side-cache resident) in registers.
//
// For an exact factor 2 with align_corners=True the source index pattern is fixed — output 2k reads inputs (k-1, k), output 2k+1
// reads (k, k+1); only the weights vary (and, in fp32, the very last output of an axis may come out as (in-2, in-1) with a weight of
// ~1 on the second: lin_index is evaluated per thread and that one deviation is honoured) — so a thread's 4 x 6 operand window
// (2 output rows x 4 columns + halo) comes from 3 rows x 4 columns x 2 planes of the level-6 block: 12 8-byte LDS reads, 12 z-,
// 16 y-, 24 x-interpolations (two instructions each) next to the 216 multiply-adds.  Zero padding of the CONVOLUTION (positions
// outside the upsampled volume) is folded into the y / x interpolation weights (both zero).
// The level-6 block of a tile (4 planes x 34 rows x 18 columns per channel, 10 KB) is staged once per channel — one barrier per
// channel instead of one per plane — while the previous channel is multiplied.  Interpolation nesting is z, y, x (ATen: x, y, z):
// the same three linear interpolations in another order.
constexpr int CU_TY = 64, CU_TX = 32, CU_THREADS = 256, CU_ZSEG = 4;
constexpr int CU_PZ = CU_ZSEG / 2 + 2, CU_PY = CU_TY / 2 + 2, CU_PX = CU_TX / 2 + 2, CU_RS = CU_PX + 2;   // block 4 x 34 x 18, row stride 20
constexpr int CU_BLK = CU_PZ * CU_PY * CU_RS;
constexpr int CU_NPF = (CU_PZ * CU_PY * CU_PX + CU_THREADS - 1) / CU_THREADS;      // 10 loads per thread and channel

struct CUArgs {
  const void* x;           // [B, Cin, Di, Hi, Wi]
  const float* w;          // [Cin][27]
  const float* scale;
  const float* shift;
  void* y;                 // [B, *, 2 Di, 2 Hi, 2 Wi]
  int64_t x_bstride, y_bstride;
  int y_ch0, relu;
  int B, Cin, Di, Hi, Wi;
  float sd, sh, sw;        // (in - 1) / (out - 1) per axis, fp32 as lin_scale gives it
  int tiles_x, tiles_y, nseg, nwork;
};

template <class T, class TO>
__global__ __launch_bounds__(CU_THREADS, 2) void upconv3d_c1_kernel(CUArgs a) {
  constexpr int ZSEG = CU_ZSEG, NP = ZSEG + 2;
  __shared__ __attribute__((aligned(16))) float blk[2][CU_BLK];
  __shared__ __attribute__((aligned(16))) float wl[C1_MAX_CIN * 28];
  const int tid = threadIdx.x, tx = tid & 7, ty = tid >> 3;
  const int D = 2 * a.Di, H = 2 * a.Hi, W = 2 * a.Wi;
  const int chunk = (a.nwork + 7) / 8, j = blockIdx.x;      // XCD-aware order (as conv3d_c1_kernel)
  int t = (j & 7) * chunk + (j >> 3);
  if ((j >> 3) >= chunk || t >= a.nwork) return;
  const int x0 = (t % a.tiles_x) * CU_TX; t /= a.tiles_x;
  const int y0 = (t % a.tiles_y) * CU_TY; t /= a.tiles_y;
  const int seg = t % a.nseg, b = t / a.nseg;
  const int zs = seg * ZSEG, ze = min(D, zs + ZSEG);
  const int HWi = a.Hi * a.Wi;
  const int64_t DHWi = (int64_t)HWi * a.Di;
  for (int i = tid; i < a.Cin * 28; i += CU_THREADS) wl[i] = (i % 28) < 27 ? a.w[(i / 28) * 27 + i % 28] : 0.f;
  // level-6 block origin: plane zs/2 - 1, row y0/2 - 1, column x0/2 - 1 (coordinates clamped into the tensor at the load)
  const int pz0 = zs / 2 - 1, py0 = y0 / 2 - 1, px0 = x0 / 2 - 1;
  int soff[CU_NPF];
#pragma unroll
  for (int p = 0; p < CU_NPF; ++p) {
    const int e = min(tid + p * CU_THREADS, CU_PZ * CU_PY * CU_PX - 1);
    const int cx = e % CU_PX, cy = (e / CU_PX) % CU_PY, cz = e / (CU_PX * CU_PY);
    soff[p] = min(max(pz0 + cz, 0), a.Di - 1) * HWi + min(max(py0 + cy, 0), a.Hi - 1) * a.Wi + min(max(px0 + cx, 0), a.Wi - 1);
  }
  const T* const xb = static_cast<const T*>(a.x) + (int64_t)b * a.x_bstride;
  float pf[CU_NPF];
  auto prefetch = [&](int c) {
    const T* const pc = xb + (int64_t)min(c, a.Cin - 1) * DHWi;
#pragma unroll
    for (int p = 0; p < CU_NPF; ++p) pf[p] = ld(pc + soff[p]);
  };
  auto commit = [&](float* buf) {
#pragma unroll
    for (int p = 0; p < CU_NPF; ++p) {
      const int e = tid + p * CU_THREADS;
      if (e < CU_PZ * CU_PY * CU_PX) buf[(e / CU_PX) * CU_RS + e % CU_PX] = pf[p];
    }
  };
  // this thread's interpolation weights (fp32 lin_index, as the standalone upsample kernel evaluates them).  Window: block rows
  // ty .. ty+2 (level-6 rows j-1, j, j+1 with j = y0/2 + ty) and block columns 2tx .. 2tx+3 (k-1 .. k+2, k = x0/2 + 2tx).
  // Expected pairs (window indices): rows 2j-1, 2j -> (0, 1); 2j+1, 2j+2 -> (1, 2); columns 2k-1, 2k -> (0, 1); 2k+1, 2k+2 -> (1, 2);
  // 2k+3, 2k+4 -> (2, 3).  A pair that starts ONE lower than expected (the last output of an axis in fp32) is taken one lower.
  float wy0[4], wy1[4], wx0[6], wx1[6];
  bool ylow[4], xlow[6];
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int gy = y0 + 2 * ty - 1 + r;
    const LinIdx l = lin_index(min(max(gy, 0), H - 1), a.Hi, H, a.sh, 1);
    const int expect = py0 + ty + (r >= 2 ? 1 : 0);
    ylow[r] = l.i0 < expect;
    const bool in = (unsigned)gy < (unsigned)H;
    wy0[r] = in ? l.w0 : 0.f; wy1[r] = in ? l.w1 : 0.f;
  }
#pragma unroll
  for (int i = 0; i < 6; ++i) {
    const int gx = x0 + 4 * tx - 1 + i;
    const LinIdx l = lin_index(min(max(gx, 0), W - 1), a.Wi, W, a.sw, 1);
    const int expect = px0 + 2 * tx + (i >= 4 ? 2 : (i >= 2 ? 1 : 0));
    xlow[i] = l.i0 < expect;
    const bool in = (unsigned)gx < (unsigned)W;
    wx0[i] = in ? l.w0 : 0.f; wx1[i] = in ? l.w1 : 0.f;
  }
  float acc[ZSEG][2][4];
#pragma unroll
  for (int k = 0; k < ZSEG; ++k)
#pragma unroll
    for (int r = 0; r < 2; ++r)
#pragma unroll
      for (int i = 0; i < 4; ++i) acc[k][r][i] = 0.f;
  prefetch(0);
  commit(blk[0]);
  __syncthreads();
  const int wbase = ty * CU_RS + 2 * tx;                    // this thread's window in a block plane (floats; even: 8-byte reads)
  for (int c = 0; c < a.Cin; ++c) {
    const float* const cur = blk[c & 1];
    prefetch(c + 1);                                       // the next channel's block travels under this channel's arithmetic
    float w[28];
#pragma unroll
    for (int i = 0; i < 28; i += 4) {
      const float4 q = *reinterpret_cast<const float4*>(wl + c * 28 + i);
      w[i] = q.x; w[i + 1] = q.y; w[i + 2] = q.z; w[i + 3] = q.w;
    }
    static_for<NP>([&](auto zi_) {
      constexpr int zi = decltype(zi_)::value;
      const int z = zs - 1 + zi;                           // upsampled plane (wave-uniform)
      if (z >= 0 && z <= min(ze, D - 1)) {
        const LinIdx lz = lin_index(z, a.Di, D, a.sd, 1);
        const int pa = lz.i0 - pz0, pb = lz.i1 - pz0;      // block planes (uniform): 0 .. 3
        float in[4][6];
        {
          float rz[3][4];
#pragma unroll
          for (int rr = 0; rr < 3; ++rr) {
            const float* const qa = cur + pa * (CU_PY * CU_RS) + wbase + rr * CU_RS;
            const float* const qb = cur + pb * (CU_PY * CU_RS) + wbase + rr * CU_RS;
            const float2 a0 = *reinterpret_cast<const float2*>(qa), a1 = *reinterpret_cast<const float2*>(qa + 2);
            const float2 b0 = *reinterpret_cast<const float2*>(qb), b1 = *reinterpret_cast<const float2*>(qb + 2);
            rz[rr][0] = lerp2(lz.w0, a0.x, lz.w1, b0.x); rz[rr][1] = lerp2(lz.w0, a0.y, lz.w1, b0.y);
            rz[rr][2] = lerp2(lz.w0, a1.x, lz.w1, b1.x); rz[rr][3] = lerp2(lz.w0, a1.y, lz.w1, b1.y);
          }
          float yv[4][4];
#pragma unroll
          for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
              const float lo = r >= 2 ? (ylow[r] ? rz[0][q] : rz[1][q]) : rz[0][q];
              const float hi = r >= 2 ? (ylow[r] ? rz[1][q] : rz[2][q]) : rz[1][q];
              yv[r][q] = lerp2(wy0[r], lo, wy1[r], hi);
            }
#pragma unroll
          for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int i = 0; i < 6; ++i) {
              const int e = i >= 4 ? 2 : (i >= 2 ? 1 : 0);           // compile time
              const float lo = e > 0 ? (xlow[i] ? yv[r][e - 1] : yv[r][e]) : yv[r][e];
              const float hi = e > 0 ? (xlow[i] ? yv[r][e] : yv[r][e + 1]) : yv[r][e + 1];
              in[r][i] = lerp2(wx0[i], lo, wx1[i], hi);
            }
        }
        static_for<3>([&](auto dz_) {
          constexpr int dz = decltype(dz_)::value, k = zi - dz;
          if constexpr (k >= 0 && k < ZSEG) {
            if (zs + k < ze) {
#pragma unroll
              for (int r = 0; r < 2; ++r)
#pragma unroll
                for (int dy = 0; dy < 3; ++dy)
#pragma unroll
                  for (int dx = 0; dx < 3; ++dx)
#pragma unroll
                    for (int i = 0; i < 4; ++i) acc[k][r][i] = fmaf(w[dz * 9 + dy * 3 + dx], in[r + dy][i + dx], acc[k][r][i]);
            }
          }
        });
      }
    });
    commit(blk[(c + 1) & 1]);      // (the buffer channel c - 1 read: every thread left it before the last barrier)
    __syncthreads();
  }
  const float sc = a.scale ? a.scale[0] : 1.f, sh = a.shift ? a.shift[0] : 0.f;
  const int64_t HW = (int64_t)H * W;
  TO* const yb = static_cast<TO*>(a.y) + (int64_t)b * a.y_bstride + (int64_t)a.y_ch0 * HW * D;
  const int gx = x0 + 4 * tx;
#pragma unroll
  for (int k = 0; k < ZSEG; ++k) {
    if (zs + k >= ze) continue;
#pragma unroll
    for (int r = 0; r < 2; ++r) {
      const int gy = y0 + 2 * ty + r;
      if (gy >= H || gx >= W) continue;
      float v[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const float u = fmaf(acc[k][r][i], sc, sh);
        v[i] = a.relu ? fmaxf(u, 0.f) : u;
      }
      st4(yb + (int64_t)(zs + k) * HW + (int64_t)gy * W + gx, v);
    }
  }
}

template <class T, class TO>
static int upconv_launch_typed(CUArgs a, hipStream_t st) {
  const int D = 2 * a.Di, H = 2 * a.Hi, W = 2 * a.Wi;
  a.tiles_x = (int)ceil_div(W, CU_TX); a.tiles_y = (int)ceil_div(H, CU_TY); a.nseg = (int)ceil_div(D, CU_ZSEG);
  const int64_t nwork = (int64_t)a.tiles_x * a.tiles_y * a.nseg * a.B;
  RAGMI_REQUIRE(nwork < (1ll << 28), RAGMI_EUNSUPPORTED, "upconv3d_c1: too many tiles");
  a.nwork = (int)nwork;
  hipLaunchKernelGGL((upconv3d_c1_kernel<T, TO>), dim3((unsigned)(ceil_div(nwork, 8) * 8)), dim3(CU_THREADS), 0, st, a);
  return check_launch("upconv3d_c1");
}

}  // namespace ragmi

extern "C" int ragmi_upconv3d_c1_supported(int Cin, int Di, int Hi, int Wi) {
  using namespace ragmi;
  // whole 16-byte output rows; an axis of length 1 has no (in - 1) / (out - 1) scale; 32-bit offsets inside a sample
  return (Cin >= 1 && Cin <= C1_MAX_CIN && Di >= 2 && Hi >= 2 && Wi >= 2 && (2 * Wi) % 4 == 0 && (int64_t)Cin * Di * Hi * Wi < (1ll << 31)) ? 1 : 0;
}

extern "C" int ragmi_upconv3d_c1_fwd(const void* x, int64_t x_bstride, const void* weight, const void* scale, const void* shift, int relu,
                                     void* y, int64_t y_bstride, int y_ch0, int B, int Cin, int Di, int Hi, int Wi, int dtype, int y_dtype,
                                     void* stream) {
  using namespace ragmi;
  RAGMI_REQUIRE(x && weight && y, RAGMI_EINVAL, "upconv3d_c1: null pointer");
  RAGMI_REQUIRE((scale == nullptr) == (shift == nullptr), RAGMI_EINVAL, "upconv3d_c1: scale/shift must both be given or both NULL");
  RAGMI_REQUIRE(B > 0 && B <= 65535 && y_ch0 >= 0, RAGMI_EINVAL, "upconv3d_c1: bad size");
  RAGMI_REQUIRE(dtype_ok(dtype) && (y_dtype == dtype || (dtype == RAGMI_BF16 && y_dtype == RAGMI_F32)), RAGMI_EUNSUPPORTED,
                "upconv3d_c1: dtype %d -> %d not built", dtype, y_dtype);
  RAGMI_REQUIRE(ragmi_upconv3d_c1_supported(Cin, Di, Hi, Wi), RAGMI_EUNSUPPORTED,
                "upconv3d_c1: needs 1..%d input channels, every input axis >= 2 and an even input width", C1_MAX_CIN);
  RAGMI_REQUIRE(y_bstride % 4 == 0 && aligned4(y, y_dtype), RAGMI_EUNSUPPORTED, "upconv3d_c1: output rows must be 16-byte aligned");
  CUArgs a{};
  a.x = x; a.w = (const float*)weight; a.scale = (const float*)scale; a.shift = (const float*)shift; a.y = y;
  a.x_bstride = x_bstride; a.y_bstride = y_bstride; a.y_ch0 = y_ch0; a.relu = relu ? 1 : 0;
  a.B = B; a.Cin = Cin; a.Di = Di; a.Hi = Hi; a.Wi = Wi;
  a.sd = lin_scale(Di, 2 * Di, 1); a.sh = lin_scale(Hi, 2 * Hi, 1); a.sw = lin_scale(Wi, 2 * Wi, 1);
  hipStream_t st = static_cast<hipStream_t>(stream);
  if (dtype == RAGMI_BF16) return y_dtype == RAGMI_F32 ? upconv_launch_typed<bf16_t, float>(a, st) : upconv_launch_typed<bf16_t, bf16_t>(a, st);
  return upconv_launch_typed<float, float>(a, st);
}

namespace ragmi {

// Cout == 1, no residual input, rows of whole 16-byte elements (W % 4 == 0 and aligned bases), volumes big enough to fill the chip
bool c1_eligible(const K3Args& a, int dtype, int y_dtype) {
#ifdef C1_DISABLE      // A/B build (tools/build_variant.sh): the generic small-Cout kernel everywhere
  return false;
#endif
  if (a.Cout != 1 || a.res != nullptr || a.Cin > C1_MAX_CIN || a.W % 4 != 0) return false;
  // per SAMPLE (which kernel a pair runs on must not depend on how a batch is split over ranks)
  if ((int64_t)a.D * a.H * a.W < C1_MIN_VOXELS || (int64_t)a.Cin * a.D * a.H * a.W >= (1ll << 31)) return false;
  // (a destination-channel offset moves the base by D*H*W elements, a multiple of 4 with W)
  return a.x_bstride % 4 == 0 && a.y_bstride % 4 == 0 && aligned4(a.x, dtype) && aligned4(a.y, y_dtype);
}

template <class T, class TO>
static int c1_launch_typed(const C1Args& a, hipStream_t st) {
  constexpr int ZSEG = C1_ZSEG;
  C1Args b = a;
  b.tiles_x = (int)ceil_div(a.W, C1_TX); b.tiles_y = (int)ceil_div(a.H, C1_TY); b.nseg = (int)ceil_div(a.D, ZSEG);
  const int64_t nwork = (int64_t)b.tiles_x * b.tiles_y * b.nseg * a.B;
  RAGMI_REQUIRE(nwork < (1ll << 28), RAGMI_EUNSUPPORTED, "conv3d_c1: too many tiles");
  b.nwork = (int)nwork;
  const unsigned grid = (unsigned)(ceil_div(nwork, 8) * 8);
  hipLaunchKernelGGL((conv3d_c1_kernel<T, TO, ZSEG>), dim3(grid), dim3(C1_THREADS), 0, st, b);
  return check_launch("conv3d_c1");
}

// k: as filled for the generic small-Cout kernel (wp[0] = the raw [1][Cin][27] weight, y_ch[0] = destination channel)
int c1_launch(const K3Args& k, int dtype, int y_dtype, hipStream_t st) {
  C1Args a{};
  a.x = k.x; a.w = k.wp[0]; a.scale = k.scale[0]; a.shift = k.shift[0]; a.y = k.y;
  a.x_bstride = k.x_bstride; a.y_bstride = k.y_bstride; a.y_ch0 = k.y_ch[0]; a.relu = k.relu & 1;
  a.B = k.B; a.Cin = k.Cin; a.D = k.D; a.H = k.H; a.W = k.W;
  if (dtype == RAGMI_BF16) return y_dtype == RAGMI_F32 ? c1_launch_typed<bf16_t, float>(a, st) : c1_launch_typed<bf16_t, bf16_t>(a, st);
  return c1_launch_typed<float, float>(a, st);
}

}  // namespace ragmi
