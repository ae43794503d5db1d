// 3x3x3 convolution with fp32 accuracy on the 16-bit matrix cores ("x3", ABI dtype RAGMI_F32X3): every fp32 operand a is split into
// two 16-bit halves, a = hi + lo (+ r), and a product a*b is accumulated in fp32 as hi*hi + hi*lo + lo*hi: three
// v_mfma_f32_16x16x32 per product, which run at 16x the rate of the fp32 MFMA forms.
//
// Round 3: the halves are FP16 (11 significant bits each), not bf16 (8).  With bf16 halves the dropped terms (lo*lo and the rounding of
// lo) are ~2^-17 of a product; seeded random weights make the soft-argmin nearly an argmin, and over weight seeds / genotypes at the
// headline size that error moved the EPE against the CPU reference between 1e-5 and 1.4e-3 px — outside the 1e-3 budget for one seed
// in three (tests/test_hip_parity.py::test_x3_margin_over_seeds_and_genotypes_at_headline_size, tests/analysis_split_precision.py).
// fp16 halves leave ~2^-22 per product — the class of fp32 reassociation itself.  What fp16 lacks is range, so the operands are
// scaled by powers of two chosen on the fly:
//   * weights: per OUTPUT CHANNEL, at pack time: w * 2^k with max |w| * 2^k in [2^9, 2^10]; 2^-k is folded into the BatchNorm scale;
//   * activations: per WORKGROUP COLUMN SEGMENT (z-marching form) / per box (deep form): x * 2^-e with the largest |x| of the
//     planes in the ring at most 2^11 when the scale is chosen (a factor 16 of headroom below fp16's 65504 for the planes that follow).
//     A plane that does not fit makes the workgroup restart its ring at that plane with a larger e (rare: one reload of three
//     planes); 2^e is folded into the epilogue's scale.  Powers of two: the scaling itself is exact.
// bf16 activation storage (RAGMI_BF16) keeps bf16 operands: the activations ARE bf16 there (no lo half), weights hi + lo in bf16.
//
// Mapping (operand layout of mfma_f32_16x16x32_{f16,bf16}: lane l holds A[row l&15][k = 8(l>>4)+j], B[k = 8(l>>4)+j][col l&15],
// D[row 4(l>>4)+reg][col l&15]):  rows = 16 output channels, cols = 16 consecutive voxels along x, K = 8 "pairs" of
// (input-channel group of 4, tap) x 4 channels.  The input halo tile lives in LDS channel-interleaved,
// [cg][z][y][x][4 ch] as 16-bit halves (one hi and one lo copy), so a lane fetches the 4 channels of one (voxel, tap) with one
// 8-byte read; weight fragments are pre-packed per lane in HBM (hi and lo) and read once per K-slice per wave.
#include "conv3d_x3_common.h"

namespace ragmi {

#ifdef RAGMI_DIAG
// In-kernel stamps of the z-marching kernel (profiling builds only, RAGMI_X3_DIAG bit 32): per wave the shader cycles (s_memtime) spent in
// each phase of its plane steps, summed over the launch, plus the wave's first / last s_memtime and s_memrealtime (100 MHz): the clock
// the chip actually held is d(memtime) / d(memrealtime) x 100 MHz (MI355X_MICROARCH.md, DVFS item 6).  The values go to a buffer of
// their own (ragmi_diag_x3_stamp_buffer) that no kernel reads; nothing is computed from them.
__device__ unsigned long long* x3_stamp_buf = nullptr;
constexpr int X3_STAMP_WORDS = 16;
#define X3_STAMP(k) do { if (dg_stamp) { __builtin_amdgcn_sched_barrier(0); const unsigned long long t_ = __builtin_amdgcn_s_memtime(); \
                                         dg_sum[k] += t_ - dg_last; dg_last = t_; __builtin_amdgcn_sched_barrier(0); } } while (0)
#else
#define X3_STAMP(k) do { } while (0)
#endif

// NCG = input-channel groups of 4 over all sets, NSET accumulator sets (2: out = act(bnA(convA(x[:, :C]))) + act(bnB(convB(x[:, C:]))),
// the Cell_3d sibling fusion of conv3d_k3).  Compile-time so that the K loop is fully unrolled (the LDS reads of the next
// K-slice are in flight under the MFMAs of the current one) and the prefetch registers are static.
// T = activation storage: float (three MFMAs per product) or bf16_t (the activations ARE bf16: no lo copy, two MFMAs per
// product — weight hi and lo — and half the LDS operand traffic)
// TAILS: 0 none, 1 fused consumer 1x1x1 convs at full resolution, 2 also DOWN-SAMPLING ones (see K3Args::ndown and the finishing step)
// XSRC == 1 (G4X): x is a channel-group-interleaved tensor [B][C/4][D][H][W][4] (include/rag_amd.h, "G4"; fp32 storage): the four channels of
// a halo voxel are ONE 16-byte load.  The tails write G4 destinations when a.tail_g4 says so (one 16-byte store per voxel).
// XSRC == 2 (round 5, stem3d1 behind the cost-volume-folded stem3d0): there is NO input tensor — a halo voxel's channels are
// act(scale0 * (A + B) + shift0) of stem3d0's variant planes (costvol_stem.hip: the arithmetic of its combine kernel, bit for bit),
// evaluated in the staging; the 164 MB tensor between the two stems is never written or read.
template <class T, int NCG, int NSET, int TAILS, int XSRC = 0>
__global__ __launch_bounds__(X3_THREADS, (NCG <= 3 ? 4 : 2)) void conv3d_x3_kernel(K3Args a, X3Extra e) {
  constexpr bool BF = std::is_same<T, bf16_t>::value;
  constexpr bool G4X = XSRC == 1, ABX = XSRC == 2;
  // (G4 under bf16 storage: four bf16 of a voxel and group = one 8-byte access; the plane source also feeds bf16 storage — its values are
  // rounded as a store would)
  constexpr int NCGS = NCG / NSET, NSLS = (NCGS * 27 + 7) / 8, NSL = NSET * NSLS;
  constexpr int NPF = (NCG * X3_PL + X3_THREADS - 1) / X3_THREADS;
  // LDS row / plane / channel-group strides in records (padded against bank conflicts: conv3d_x3_common.h; staging still enumerates X3_PL voxels)
  constexpr int RS = x3_row_stride(NCG, NSET), PLS = x3_plane_stride(NCG, NSET), CGS = x3_group_stride(NCG, NSET);
  static_assert(NCG % NSET == 0 && NPF <= 32, "bad instantiation");
  extern __shared__ __attribute__((aligned(16))) uint2 x3_lds[];       // hi[NCG][3][PL] | lo[NCG][3][PL] (uint2 = 4 bf16) | weights | offsets | params
  uint2* const lhi = x3_lds;
  uint2* const llo = x3_lds + NCG * CGS;                                            // absent for bf16 storage
  uint4* const lw = reinterpret_cast<uint4*>(x3_lds + (BF ? 1 : 2) * NCG * CGS);    // [set][slice][hi/lo][64 lanes]
  // operand byte offsets, ready to use: [ring phase 0..2][slice][lane quarter kb] -> (pair 2kb, pair 2kb+1) of that slice with the ring
  // rotation already applied, so the K loop spends no VALU work on addresses (one 8-byte table read per slice instead of two
  // plus ~8 instructions of mod-3 arithmetic)
  uint4* const ltail = lw + NSL * 2 * 64;             // fused-tail weight fragments ta1 | ta2 | ta3, [3][64 lanes] (kept out of the registers)
  int2* const loff = reinterpret_cast<int2*>(ltail + 3 * 64);
  // scale[2][16] (fp32 storage: times the column's 2^e, rewritten per column) | shift[2][16] | tail scale[4 kb][4] | tail shift[4][4] |
  // static scale[2][16] (BatchNorm scale x the weights' 2^-k) | the column's running max |x| (float bits)
  float* const par = reinterpret_cast<float*>(loff + 3 * NSL * 4);
  unsigned* const lmaxp = reinterpret_cast<unsigned*>(par + 128);
  // down-sampling tails (TAILS == 2): x-blended tail values of two consecutive planes U[plane parity][down slot 2][8 rows][16][4 ch] |
  // x table [16]{w0, w1: weights of the source pair (2X, 2X+1)} | y table [4]{...}
  float* const ldu = par + 132;
  float4* const lsrc = reinterpret_cast<float4*>(par + 132);      // XSRC == 2 (no down-sampling tails there): stem3d0's scale[NCG] | shift[NCG] per channel group
  float4* const ldxt = reinterpret_cast<float4*>(ldu + 2 * 2 * 4 * X3_TY * (X3_TX / 2));
  float4* const ldyt = ldxt + X3_TX / 2;
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, n = lane & 15, kb = lane >> 4;
  const int cog = blockIdx.y;
  const int HW = a.H * a.W;
  float act_floor = (a.relu & 1) ? 0.f : __builtin_nanf("");   // max(u, NaN) = u: the identity, NaN inputs included
  if constexpr (ABX) {      // rows 12..15 of the product = stem3d0's fused tail (RAGMI_TAIL_ROWS): its own activation
    if (e.src.tail_rows && kb == 3) act_floor = e.src.tail_relu ? 0.f : __builtin_nanf("");
  }
  asm volatile("" : "+v"(act_floor));      // opaque: otherwise the compiler turns max(u, floor) back into max(u, 0) + a select per value
  const int64_t DHW = (int64_t)HW * a.D;
#ifdef RAGMI_DIAG
  const bool dg_nostore = (a.relu & 0x100) != 0, dg_nomfma = (a.relu & 0x200) != 0, dg_nocommit = (a.relu & 0x400) != 0, dg_noload = (a.relu & 0x800) != 0, dg_noread = (a.relu & 0x1000) != 0;
  const bool dg_stamp = (a.relu & 0x2000) != 0 && x3_stamp_buf != nullptr;
  unsigned long long dg_sum[8] = {0, 0, 0, 0, 0, 0, 0, 0}, dg_last = 0, dg_t0 = 0, dg_r0 = 0;
  unsigned dg_steps = 0;
  if (dg_stamp) { dg_t0 = dg_last = __builtin_amdgcn_s_memtime(); dg_r0 = __builtin_amdgcn_s_memrealtime(); }
#else
  constexpr bool dg_nostore = false, dg_nomfma = false, dg_nocommit = false, dg_noload = false, dg_noread = false;
#endif
  // weight fragments, 8 bytes (one pair's four channels) at a time: lane quarter q of slice s holds the pairs x3_pair_perm(.., 2q + j)
  // of the packed slice (packed: quarter p >> 1, half p & 1)
  // (every descriptor array of the kernel arguments below is indexed by a COMPILE-TIME index in an unrolled loop, the lanes pick by
  // comparison: indexed by a lane-dependent value the compiler fetches the pointer itself with a vector load from the argument
  // segment and the value with a second, dependent one — ~25 serial memory round trips in front of the first plane of a launch)
  // (and all loads of a set first, then its LDS stores: as load -> store per element the loop waits out a memory round trip per
  // iteration, up to eleven of them for the 22 KB of a 12-channel set)
#pragma unroll
  for (int set = 0; set < NSET; ++set) {
    const uint2* const src = reinterpret_cast<const uint2*>(e.wf[set] + (int64_t)cog * NSLS * 2 * 64);
    constexpr int PER_SET = NSLS * 2 * 64 * 2, ITER = (PER_SET + X3_THREADS - 1) / X3_THREADS;
    uint2 wv[ITER];
#pragma unroll
    for (int it = 0; it < ITER; ++it) {
      const int i = min(it * X3_THREADS + tid, PER_SET - 1);
      const int j = i & 1, ln = (i >> 1) & 63, sh = i >> 7, sl = sh >> 1;
      const int p = x3_pair_perm(NCGS, sl, 2 * (ln >> 4) + j);
      wv[it] = src[((sh * 64) + (p >> 1) * 16 + (ln & 15)) * 2 + (p & 1)];
    }
#pragma unroll
    for (int it = 0; it < ITER; ++it) {
      const int i = it * X3_THREADS + tid;
      if (i < PER_SET) reinterpret_cast<uint2*>(lw)[set * PER_SET + i] = wv[it];
    }
  }
  for (int i = tid; i < 3 * NSL * 4; i += X3_THREADS) {
    const int ring = i / (NSL * 4), q = i % (NSL * 4), set = q / (NSLS * 4);
    const int sl = (q % (NSLS * 4)) >> 2, qt = q & 3;
    auto pair_offset = [&](int quarter, int j) {
      const int P = 8 * sl + x3_pair_perm(NCGS, sl, 2 * quarter + j), cgl = P % NCGS, tap = P / NCGS;      // tap-major pairs (x3_pack_one)
      const int slot = (ring + tap / 9) % 3;          // plane z+dz-1 sits in slot (ring + dz) % 3
      return tap < 27 ? ((set * NCGS + cgl) * CGS + slot * PLS + ((tap / 3) % 3) * RS + tap % 3) * (int)sizeof(uint2) : -1;
    };
    int o[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      // (padding pairs — zero weights — read what the quarter they share the pass with reads: a voxel inside the 3x3x3 window, and
      // no bank conflict; offset 0 — also inside the window — when that one is padding too)
      o[j] = pair_offset(qt, j);
      if (o[j] < 0) o[j] = max(pair_offset(qt ^ 1, j), 0);
    }
    loff[i] = make_int2(o[0], o[1]);
  }
  if (tid < 32) {
    const int co = cog * 16 + (tid & 15);
    float sc = 1.f, sh = 0.f;
#pragma unroll
    for (int set = 0; set < NSET; ++set) {
      const float* const psc = a.scale[set];
      const float* const psh = a.shift[set];
      if ((tid >> 4) == set && co < a.Cout) {
        sc = psc ? psc[co] : 1.f;
        if constexpr (!BF) sc *= e.wmul[set][co];          // undo the per-channel weight scale 2^k
        sh = psh ? psh[co] : 0.f;
      }
    }
    if constexpr (ABX) {      // rows 12..15: the fused tail's BatchNorm x its rows' weight scale 2^-k (the pack is a 16-channel one)
      if (e.src.tail_rows && tid >= 12 && tid < 16) {
        const float* const tsc = e.src.tail_scale;
        const float* const tsh = e.src.tail_shift;
        sc = tsc ? tsc[tid - 12] : 1.f;
        if constexpr (!BF) sc *= e.wmul[0][tid];
        sh = tsh ? tsh[tid - 12] : 0.f;
      }
    }
    par[tid] = sc;
    par[96 + tid] = sc;
    par[32 + tid] = sh;
  }
  if (tid < 2) lmaxp[tid] = 0u;
  if constexpr (ABX) {      // stem3d0's folded BatchNorm per channel group (identity when absent)
    if (tid < 2 * NCG) {
      const bool isshift = tid >= NCG;
      const int cg = isshift ? tid - NCG : tid;
      const float* const pp = isshift ? e.src.shift : e.src.scale;
      const float idv = isshift ? 0.f : 1.f;
      lsrc[tid] = pp ? make_float4(pp[4 * cg], pp[4 * cg + 1], pp[4 * cg + 2], pp[4 * cg + 3]) : make_float4(idv, idv, idv, idv);
    }
  }
  // Fused consumer 1x1x1 convs ("tails") on the matrix cores: out_t[k][voxel] = sum_c W_t[k][c] * v[c][voxel] as four 16x16x4 fp32
  // products laid out so that every lane quarter feeds ITS OWN four channels (product r: channel 4 kb + r of each quarter), so no
  // value crosses lanes.  Rows: tail 0 -> 0..3, tail 1 -> 4..7, down-sampling tails after them.
  // (TAILS is compile time; the tail fragments and parameters live in LDS and are fetched in the epilogue — in registers they cost
  // ~20 VGPRs of a 128-VGPR budget and the main loop spilled)
  if constexpr (TAILS) {
    // (round 4: the tail products run on v_mfma_f32_16x16x4_f32 — four EXACT fp32 products, product r taking channel 4 kb + r of
    // every lane quarter — instead of three bf16 products of three-way split operands: the splits were ~28 vector instructions per
    // tile in a kernel bound by vector issue; the fp32 MFMA holds the issue port for 8 of its 32 cycles)
    if (tid < 64) {
      const int row = n, tl = row >> 2, k = row & 3;     // n = lane & 15 is the A row: tail slot tl, its output channel k
      float wv[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        const float* const pw = a.tail_w[t];
        if (t < a.ntail && tl == t && k < a.tail_cout[t]) {
#pragma unroll
          for (int j = 0; j < 4; ++j) { const int c = cog * 16 + 4 * kb + j; if (c < a.Cout) wv[j] = pw[k * a.Cout + c]; }
        }
        if constexpr (TAILS == 2) {
          const float* const pd = a.down_w[t];
          if (t < a.ndown && tl == a.ntail + t && k < a.down_cout[t]) {
#pragma unroll
            for (int j = 0; j < 4; ++j) { const int c = cog * 16 + 4 * kb + j; if (c < a.Cout) wv[j] = pd[k * a.Cout + c]; }
          }
        }
      }
#pragma unroll
      for (int j = 0; j < 4; ++j) reinterpret_cast<float*>(ltail)[j * 64 + lane] = wv[j];
    }
    // this lane's tail outputs after that product: rows 4 kb + r -> tail kb, output r
    if (tid < 16) {
      const int tk = tid >> 2, r = tid & 3;
      float sc = 1.f, sh = 0.f;
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        const float* const psc = a.tail_scale[t];
        const float* const psh = a.tail_shift[t];
        if (t < a.ntail && tk == t && r < a.tail_cout[t] && psc) { sc = psc[r]; sh = psh[r]; }
        if constexpr (TAILS == 2) {
          const float* const dsc = a.down_scale[t];
          const float* const dsh = a.down_shift[t];
          if (t < a.ndown && tk == a.ntail + t && r < a.down_cout[t] && dsc) { sc = dsc[r]; sh = dsh[r]; }
        }
      }
      par[64 + tid] = sc;
      par[80 + tid] = sh;
    }
  }
  // staging elements of a thread: (channel group, halo voxel) pairs numbered p * X3_THREADS + tid through the groups — or, with the
  // plane source (XSRC == 2, three groups), TWO elements per thread that share as much of the plane addressing as possible: thread
  // t < 340 stages groups 0 and 1 of halo voxel t (one address computation, four 16-byte loads), thread 340 + u (u < 170) group 2 of
  // the voxels 2u and 2u + 1.  (Until stem3d0's tail moved into the matrix product's idle rows — RAGMI_TAIL_ROWS — a voxel's three
  // groups had to meet in ONE thread, 340 of 512 threads staging three elements each.)
  static_assert(!ABX || (NCG == 3 && 2 * (X3_THREADS - X3_PL) >= X3_PL), "plane source: the element mapping below");
  constexpr int NPE = ABX ? 2 : NPF;
  const bool abx_pair = tid >= X3_PL;                        // XSRC == 2: this thread stages group 2 of two voxels
  auto el_cg = [&](int p) { return ABX ? (abx_pair ? 2 : p) : (p * X3_THREADS + tid) / X3_PL; };
  auto el_r = [&](int p) { return ABX ? (abx_pair ? min(2 * (tid - X3_PL) + p, X3_PL - 1) : tid) : (p * X3_THREADS + tid) % X3_PL; };
  auto el_on = [&](int p) { return ABX ? (!abx_pair || 2 * (tid - X3_PL) + p < X3_PL) : p * X3_THREADS + tid < NCG * X3_PL; };
  float pf[NPE][4];
  unsigned valid = 0;
  const T* const x = static_cast<const T*>(a.x);
  // this thread's halo elements of a column, located ONCE per column: element offset of each of the four channels inside the
  // plane (channel clamp and (y, x) clamp folded in) and whether the voxel lies inside the plane.  Per plane only the wave-uniform
  // plane base changes, so the loads are `uniform base + lane offset` with no address arithmetic left in the z loop (per plane it
  // was ~100 VALU instructions per wave next to 48 MFMAs, and VALU issue is additive to MFMA issue on this chip)
  int voff[NPF];
  int vgy[ABX ? 2 : 1] = {}, vgx[ABX ? 2 : 1] = {};      // XSRC == 2: the elements' voxels (clamped row, column)
  float4 pa[ABX ? 2 : 1], pbv[ABX ? 2 : 1];              // XSRC == 2: the A and B plane values of the plane in flight, per element
  unsigned hflags = 0;                 // XSRC == 2: bits 2p, 2p + 1: element p's voxel has an A term in that plane, a B term
  unsigned vmask = 0;
  auto locate = [&](int y0, int x0) {
    vmask = 0;
    if constexpr (ABX) {
#pragma unroll
      for (int p = 0; p < 2; ++p) {
        const int r = el_r(p), xx = r % X3_HX, yy = r / X3_HX;
        const int gy = y0 - 1 + yy, gx = x0 - 1 + xx;
        const bool ok = el_on(p) && (unsigned)gy < (unsigned)a.H && (unsigned)gx < (unsigned)a.W;
        vmask |= (ok ? 1u : 0u) << p;
        vgy[p] = min(max(gy, 0), a.H - 1); vgx[p] = min(max(gx, 0), a.W - 1);
      }
      return;
    }
#pragma unroll
    for (int p = 0; p < NPF; ++p) {
      const int el = p * X3_THREADS + tid, cg = el / X3_PL, r = el % X3_PL;
      const int xx = r % X3_HX, yy = r / X3_HX;
      const int gy = y0 - 1 + yy, gx = x0 - 1 + xx;
      const bool ok = cg < NCG && (unsigned)gy < (unsigned)a.H && (unsigned)gx < (unsigned)a.W;
      vmask |= (ok ? 1u : 0u) << p;
      // whole 4-channel groups only (x3_eligible): the group's channels are c * DHW apart, a wave-uniform step.  Cin * DHW < 2^31.
      voff[p] = (int)(min(cg, NCG - 1) * 4 * DHW) + (min(max(gy, 0), a.H - 1) * a.W + min(max(gx, 0), a.W - 1)) * (G4X ? 4 : 1);
    }
  };
  // issue the loads of input plane gz of the located column: unconditional, the plane index clamped into the volume
  const float* wsb = nullptr;          // XSRC == 2: this sample's planes
  T* tdst = nullptr;                   // ... and the sample of the fused tail's destination
  auto prefetch = [&](const T* xb, int gz) {
    if (dg_noload) return;
    valid = (unsigned)gz < (unsigned)a.D ? vmask : 0u;
    if constexpr (ABX) {
      // the combine kernel's addressing (costvol_stem.hip), once per voxel: cls = z border, tc = clamp(x - i, -3, 2) picks the A plane
      // (full / band / none), B is indexed by x - i (right-border variant at x = W - 1); unconditional loads — an absent term reads
      // the set's base and is discarded by a select in mat()
      const int gzc = min(max(gz, 0), a.D - 1), cls = (gzc == 0 ? 1 : 0) + (gzc == a.D - 1 ? 2 : 0);
      auto pick = [&](const int (&o)[4]) { return cls == 0 ? o[0] : cls == 1 ? o[1] : cls == 2 ? o[2] : o[3]; };      // (scalar selects: no argument-segment loads)
      const int ofull = pick(e.src.off_afull), oband = pick(e.src.off_aband), ob0 = pick(e.src.off_b0), ob1 = pick(e.src.off_b1);
      constexpr int CM = 4 * NCG;
      hflags = 0;
      int oa[2], ob[2];
#pragma unroll
      for (int p = 0; p < 2; ++p) {
        if (p == 1 && !abx_pair) { oa[1] = oa[0]; ob[1] = ob[0]; hflags |= (hflags & 3u) << 2; continue; }      // (same voxel, the next group)
        const int t = vgx[p] - gzc, tc = min(max(t, -3), 2);
        oa[p] = ofull;
        if (tc == 2) oa[p] = ofull + (vgy[p] * a.W + vgx[p]) * CM;
        else if (tc > -3) oa[p] = oband + (tc + 2) * CM * a.H * e.src.wband + (vgy[p] * e.src.wband + vgx[p] - (tc == 1 ? 1 : 0)) * CM;
        ob[p] = ob0;
        if (t >= -2) ob[p] = vgx[p] == a.W - 1 ? ob1 + (vgy[p] * e.src.wb1 + (t - e.src.u1_0)) * CM : ob0 + (vgy[p] * (a.W + 2) + (t + 2)) * CM;
        hflags |= ((tc > -3 ? 1u : 0u) | (t >= -2 ? 2u : 0u)) << (2 * p);
      }
      if (tid < X3_PL + (X3_PL + 1) / 2) {
#pragma unroll
        for (int p = 0; p < 2; ++p) {
          pa[p] = *reinterpret_cast<const float4*>(wsb + oa[p] + 4 * el_cg(p));
          pbv[p] = *reinterpret_cast<const float4*>(wsb + ob[p] + 4 * el_cg(p));
        }
      }
      return;
    }
    const T* const pb = xb + (int64_t)min(max(gz, 0), a.D - 1) * HW * (G4X ? 4 : 1);
    if constexpr (G4X) {
#pragma unroll
      for (int p = 0; p < NPF; ++p) {
        if constexpr (BF) {
          const uint2 v = *reinterpret_cast<const uint2*>(pb + voff[p]);
          pf[p][0] = __uint_as_float(v.x << 16); pf[p][1] = __uint_as_float(v.x & 0xffff0000u);
          pf[p][2] = __uint_as_float(v.y << 16); pf[p][3] = __uint_as_float(v.y & 0xffff0000u);
        } else {
          const float4 v = *reinterpret_cast<const float4*>(pb + voff[p]);
          pf[p][0] = v.x; pf[p][1] = v.y; pf[p][2] = v.z; pf[p][3] = v.w;
        }
      }
    } else {
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        const T* const pc = pb + c * DHW;            // wave-uniform
#pragma unroll
        for (int p = 0; p < NPF; ++p) pf[p][c] = ld(pc + voff[p]);
      }
    }
  };
  // XSRC == 2: the plane values in flight -> the element's four channels, exactly as costvol_stem_combine_kernel computes them:
  // s = (B term or 0) + (A term or 0); s = fma(s, scale, shift); ReLU.  Called where the loads are first needed.
  // (stem3d0's fused tail — cell 0's pre_preprocess, rag_model.py:125,154 — is NOT formed here: it rides in rows 12..15 of the matrix
  // product, RAGMI_TAIL_ROWS.  Round 5's first form evaluated it here as a 48-term chain by the voxel's owner; built as v_pk_fma_f32
  // that chain gave wrong low halves in lanes 48..63 now and then — NOTES.md, round-5 log — and it cost 16 us of the launch.)
  auto mat = [&](int) {
    if constexpr (ABX) {
#pragma unroll
      for (int p = 0; p < 2; ++p) {
        const bool ha = (hflags >> (2 * p)) & 1u, hb = (hflags >> (2 * p + 1)) & 1u;
        const int cg = el_cg(p);
        const float4 sc = lsrc[cg], sh = lsrc[NCG + cg];
        const float av[4] = {pa[p].x, pa[p].y, pa[p].z, pa[p].w}, bv[4] = {pbv[p].x, pbv[p].y, pbv[p].z, pbv[p].w};
        const float s4[4] = {sc.x, sc.y, sc.z, sc.w}, h4[4] = {sh.x, sh.y, sh.z, sh.w};
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          float sv = (hb ? bv[c] : 0.f) + (ha ? av[c] : 0.f);
          sv = fmaf(sv, s4[c], h4[c]);
          pf[p][c] = e.src.relu ? fmaxf(sv, 0.f) : sv;
        }
      }
    }
  };
  float mul = 1.f;                       // fp32 storage: the column's operand scale 2^-e (wave-uniform)
  unsigned cap_bits = 0x7f7fffffu;       // bit pattern of X3_F16_CAP / mul: an element above it does not fit the column's scale
  auto commit = [&](int slot) {          // registers -> ring plane `slot` (16-bit hi / lo halves), zeros outside the volume / past Cin
    if (dg_nocommit) return;
#pragma unroll
    for (int p = 0; p < NPE; ++p) {
      if (!el_on(p)) continue;
      const int cg = el_cg(p), r = el_r(p);
      float v[4];
#pragma unroll
      for (int c = 0; c < 4; ++c) v[c] = ((valid >> p) & 1u) ? pf[p][c] : 0.f;
      unsigned l01, l23, h01, h23;
      if constexpr (BF) { h01 = x3_split2(v[0], v[1], l01); h23 = x3_split2(v[2], v[3], l23); }
      else { h01 = x3_split2h(v[0], v[1], mul, l01); h23 = x3_split2h(v[2], v[3], mul, l23); }
      const int d = cg * CGS + slot * PLS + (r / X3_HX) * RS + r % X3_HX;
      lhi[d] = make_uint2(h01, h23);
      if constexpr (!BF) llo[d] = make_uint2(l01, l23);
    }
  };
  // largest |value| among this thread's valid halo elements of the plane in flight (fp32 storage: feeds the operand scale)
  // (elements no scale can bring into fp16 — Inf, NaN, |x| >= 2^115 — take no part: they become fp16 Inf / NaN on their own and
  // disturb the outputs whose 3x3x3 window holds them, like the reference's arithmetic; letting them pick the scale would flush
  // every ordinary activation of the tile to zero for the rest of the segment)
  auto local_max = [&]() {
    float m = 0.f;
#pragma unroll
    for (int p = 0; p < NPE; ++p) {
      const float mp = x3_scalable_max4(pf[p][0], pf[p][1], pf[p][2], pf[p][3]);
      m = fmaxf(m, ((valid >> p) & 1u) ? mp : 0.f);
    }
    return m;
  };
  // a plane that does not fit the column's scale: record its magnitude; the workgroup restarts its ring behind the next barrier
  // (lmaxp[1], never the word the scale is derived from: lmaxp[0] only changes between the two barriers that open a ring pass, so
  // `mul` and the restart test below are workgroup-uniform by construction)
  // Fast test first (this runs once per plane step, in a loop bound by vector issue): the largest |bit pattern| of the thread's valid
  // elements against the pattern of CAP / mul (mul is a power of two: exact) — 8 ands, a max tree and one compare instead of the
  // ~50 instructions of the filtered maximum.  Only a thread that holds an element above the threshold (an overflow, or an
  // unscalable element: Inf, NaN, >= 2^115) takes the slow path, which decides exactly as before.
  auto note_overflow = [&]() {
    unsigned mb = 0u;
#pragma unroll
    for (int p = 0; p < NPE; ++p) {
      const unsigned mp = max(max(__float_as_uint(pf[p][0]) & 0x7fffffffu, __float_as_uint(pf[p][1]) & 0x7fffffffu),
                              max(__float_as_uint(pf[p][2]) & 0x7fffffffu, __float_as_uint(pf[p][3]) & 0x7fffffffu));
      mb = max(mb, ((valid >> p) & 1u) ? mp : 0u);
    }
    if (mb > cap_bits) {
      const float m = local_max();
      if (m * mul > X3_F16_CAP) atomicMax(lmaxp + 1, __float_as_uint(m));
    }
  };
  // this wave's column tiles of a plane: nt = wave * X3_NT + i -> (row y = nt / 2, x half = nt % 2); X3_NT is even, so tile i
  // sits a compile-time distance behind tile 0 (an immediate offset of the LDS read)
  static_assert(X3_NT % 2 == 0, "tile deltas below assume an even number of column tiles per wave");
  const int vb0 = ((wave * X3_NT >> 1) * RS + n) * (int)sizeof(uint2);
  int vbt[X3_NT];                                      // byte base of each of this wave's column tiles (see the K loop)
#pragma unroll
  for (int i = 0; i < X3_NT; ++i) {
    vbt[i] = vb0 + ((i >> 1) * RS + (i & 1) * 16) * (int)sizeof(uint2);
    asm volatile("" : "+v"(vbt[i]));
  }
  const char* const lbytes = reinterpret_cast<const char*>(x3_lds);
  constexpr int LO_BYTES = NCG * CGS * (int)sizeof(uint2);
  const int g = cog * 4 + kb, ngroups = (a.Cout + 3) >> 2;
  // per-lane destinations, read ONCE: indexing the kernel-argument arrays with a lane-dependent index inside the loop is a
  // vector memory load per use, and its s_waitcnt vmcnt drains the prefetch that is supposed to fly under the MFMAs
  const int my_ych = g < ngroups ? a.y_ch[g < RAGMI_MAX_GROUPS ? g : 0] : 0;
  const int tsel = kb & 1;
  const int my_tail_cout = TAILS ? (kb < a.ntail ? (tsel ? a.tail_cout[1] : a.tail_cout[0]) : 0) : 0;
  const int my_trelu = tsel ? a.tail_relu[1] : a.tail_relu[0];
  T* tbase = nullptr;
  // XCD-aware schedule: workgroup j runs on XCD j % 8 (round-robin dispatch); give every XCD one contiguous chunk of the
  // (x-fastest) work list so that neighbouring columns — which share halo rows and cache lines — meet in the same L2
  // Down-sampling tails, second half: the planes 2Z (LDS parity 0) and 2Z+1 (parity 1) of the x-blended tail values are complete ->
  // blend along y, then z (ATen's nesting: x innermost), BatchNorm + ReLU, store the half-resolution tile (4 rows x 16 columns per slot
  // channel): one output per thread and plane pair.
  auto down_finish = [&](int zodd, int b, int y0, int x0) {
    if constexpr (TAILS == 2) {
      const int Z = zodd >> 1, Do = a.D >> 1, Ho = a.H >> 1, Wo = a.W >> 1;
      const LinIdx lz = lin_index(min(Z, Do - 1), a.D, Do, e.dsd, 1);            // wave-uniform
      const float wz0 = lz.i0 == 2 * Z ? lz.w0 : 0.f, wz1 = lz.i0 == 2 * Z ? lz.w1 : 1.f;
      const int64_t ovol = (int64_t)Do * Ho * Wo;
      // one output VALUE per thread: (slot, row pair, column) x channel — all eight waves take part (round 4, first form: one thread
      // per four channels, i.e. two waves worked while six waited at the next barrier)
      // (the down slot is wave-uniform — waves 0..3 finish slot 0, waves 4..7 slot 1 — and SAID to be: picked by a lane-dependent index
      // the descriptors below were three serial vector loads from the argument segment per finishing step, each behind an
      // s_waitcnt vmcnt(0) that also drained the halo prefetch)
      const int o = tid >> 2, r = tid & 3;
      const int dl = __builtin_amdgcn_readfirstlane(tid >> 8);
      if (dl < a.ndown) {
        const int yp = (o >> 4) & 3, xp = o & 15;
        const float4 yt = ldyt[yp];
        const float* const u0 = ldu + (((0 * 2 + dl) * X3_TY + 2 * yp) * (X3_TX / 2) + xp) * 4 + r;
        const float* const u1 = u0 + 2 * X3_TY * (X3_TX / 2) * 4;
        const float e0 = u0[0], e1 = u0[(X3_TX / 2) * 4], o0 = u1[0], o1 = u1[(X3_TX / 2) * 4];
        const int slot = a.ntail + dl;
        const float sc = par[64 + 4 * slot + r], sh = par[80 + 4 * slot + r];
        const int dcout = dl ? a.down_cout[1] : a.down_cout[0], drelu = dl ? a.down_relu[1] : a.down_relu[0];
        const int Y = (y0 >> 1) + yp, X = (x0 >> 1) + xp;
        if (Y < Ho && X < Wo && Z < Do && r < dcout) {
          void* const dbase = dl ? a.down_y[1] : a.down_y[0];
          const int64_t doff = b * (dl ? a.down_bstride[1] : a.down_bstride[0]) +
                               (int64_t)((dl ? a.down_ch0[1] : a.down_ch0[0]) + r) * ovol + ((int64_t)Z * Ho + Y) * Wo + X;
          const bool yclamp = yt.z != 0.f, zclamp = lz.i0 != 2 * Z;      // clamped pairs read the odd source twice
          const float ye = lerp2(yt.x, yclamp ? e1 : e0, yt.y, e1), yo = lerp2(yt.x, yclamp ? o1 : o0, yt.y, o1);
          const float uu = fmaf(lerp2(wz0, zclamp ? yo : ye, wz1, yo), sc, sh);
          // (mixed storage: a bf16 launch whose half-resolution consumer keeps fp32 — RAGMI_TAIL_F32; wave-uniform)
          if (BF && ((a.down_f32 >> dl) & 1)) static_cast<float*>(dbase)[doff] = drelu ? fmaxf(uu, 0.f) : uu;
          else st(static_cast<T*>(dbase) + doff, drelu ? fmaxf(uu, 0.f) : uu);
        }
      }
    }
  };
  const int chunk = (e.nwork + 7) / 8;
  for (int j = blockIdx.x; j < chunk * 8; j += gridDim.x) {
    const int work = (j & 7) * chunk + (j >> 3);
    if ((j >> 3) >= chunk || work >= e.nwork) continue;
    // virtual item -> (sample, column, depth segment[, half]): a sample's items form `ngrp` groups (8: one per XCD at batch 1) of `grp`
    // (column, segment) pairs; the last `nsplit` pairs of a group are TWO items, the halves of the segment (x3_launch)
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
    const T* xb = x + b * a.x_bstride;
    if constexpr (TAILS) {
      T* const my_tail = static_cast<T*>(tsel ? a.tail_y[1] : a.tail_y[0]);
      const int64_t tb = tsel ? a.tail_bstride[1] : a.tail_bstride[0];
      const int tch0 = tsel ? a.tail_ch0[1] : a.tail_ch0[0];
      const int64_t v0 = (int64_t)(y0 + wave) * a.W + x0 + n;      // this wave's row of the tile (X3_NT == 2: tile i = x half i)
      // (G4: the group tch0 / 4 of an interleaved tensor, four elements per voxel)
      tbase = a.tail_g4 ? my_tail + b * tb + ((int64_t)(tch0 >> 2) * DHW + v0) * 4 : my_tail + b * tb + (int64_t)tch0 * DHW + v0;
    }
    if constexpr (ABX) {
      wsb = e.src.ws + b * e.src.ws_bstride;
      tdst = reinterpret_cast<T*>(e.src.tail_y) + b * e.src.tail_bstride;
    }
    __syncthreads();                                   // the previous column's LDS reads are done (and the tables above are written)
    locate(y0, x0);
    if constexpr (!BF) { if (tid < 2) lmaxp[tid] = 0u; }  // (ordered before the first atomicMax below by the barrier that follows)
    if constexpr (TAILS == 2) {
      // interpolation tables of this column's half-resolution outputs: (w0, w1, first operand from the ODD source, second from the ODD
      // source) — the source pair of output o is (2o, 2o+1) except where the last output of an axis clamps (host-checked)
      if (tid < X3_TX / 2 + X3_TY / 2) {
        const bool isx = tid < X3_TX / 2;
        const int o = isx ? (x0 >> 1) + tid : (y0 >> 1) + (tid - X3_TX / 2), in = isx ? a.W : a.H;
        const LinIdx l = lin_index(min(o, (in >> 1) - 1), in, in >> 1, isx ? e.dsw : e.dsh, 1);
        // the pair (source 2o, source 2o+1) with weights (w0, w1); where the last output of an axis clamps (i0 = i1 = 2o+1, fp32
        // source index a hair above in-1) the reference blends the odd source with itself: weights (0, 1)
        // (.z != 0: the clamped pair — BOTH taps are the odd source, as in the reference; blending the even one with weight 0 would turn
        // a non-finite even source into NaN where the reference stays finite: ADVICE r04)
        const float4 ent = l.i0 == 2 * o ? make_float4(l.w0, l.w1, 0.f, 0.f) : make_float4(0.f, 1.f, 1.f, 0.f);
        if (isx) ldxt[tid] = ent; else ldyt[tid - X3_TX / 2] = ent;
      }
    }
    int zfirst = zs;
    // fp32 storage: the ring (re)starts at plane zfirst with the operand scale chosen from that plane — its largest |x| lands at
    // 2^10..2^11, a factor >= 16 below fp16's range for the planes that follow; one that still does not fit restarts the ring at
    // the current plane with a larger scale (the running maximum only grows within a column segment).  bf16 storage: one pass.
    for (;;) {
      if constexpr (!BF) {
        __syncthreads();
        // lmaxp[0] = the maximum the scale is chosen from: running maximum of the segment so far (a restart folds the overflow
        // note in) joined by plane zfirst.  Written only here, between this pass's two barriers; lmaxp[1] = overflow notes, written
        // only AFTER the second barrier and read behind the z loop's barriers.
        if (tid == 0) { const unsigned note = lmaxp[1]; lmaxp[1] = 0u; if (note) atomicMax(lmaxp, note); }
        prefetch(xb, zfirst); mat(zfirst);
        const float wm = x3_wave_max(local_max());
        if (lane == 0) atomicMax(lmaxp, __float_as_uint(wm));
        __syncthreads();
        mul = x3_pow2_scale(__uint_as_float(lmaxp[0]), X3_ACT_TARGET);
        cap_bits = __float_as_uint(X3_F16_CAP / mul);      // (mul in [2^-101, 2^99]: finite)
        if (tid < 32) par[tid] = par[96 + tid] * (1.f / mul);       // the epilogue's scale undoes the column's 2^-e
        commit(zfirst % 3);
        prefetch(xb, zfirst - 1); mat(zfirst - 1); note_overflow(); commit((zfirst - 1 + 3) % 3);
        prefetch(xb, zfirst + 1);
      } else {
        prefetch(xb, zs - 1); mat(zs - 1); commit((zs - 1 + 3) % 3);
        prefetch(xb, zs); mat(zs); commit(zs % 3);
        prefetch(xb, zs + 1);
      }
      bool again = false;
    for (int z = zfirst; z < ze; ++z) {
      X3_STAMP(6);                                     // (profiling builds: everything outside the plane steps — ring start, item decode)
      __syncthreads();                                 // plane z-2 (same ring slot as z+1) is no longer read
      X3_STAMP(0);
      if constexpr (TAILS == 2) { if (z > zs && !(z & 1)) down_finish(z - 1, b, y0, x0); }     // planes z-2, z-1 are complete (segments start even)
      mat(z + 1);
      if constexpr (!BF) note_overflow();
      commit((z + 1) % 3);
      X3_STAMP(1);
      __syncthreads();
      X3_STAMP(2);
      if constexpr (!BF) {
        // (workgroup-uniform.  No restart once the scale sits at its floor: an Inf — or an operand above ~2^110 — can never be made
        // to fit, and restarting for it would never end; such inputs give non-finite outputs, as include/rag_amd.h says)
        if (lmaxp[1] != 0u && __float_as_uint(mul) > X3_SCALE_FLOOR_BITS) { zfirst = z; again = true; break; }
      }
      // unconditional (also past the segment end: the addresses are clamped): the loads stay straight-line code ahead of the
      // MFMA block, a branch here made the compiler drain them (s_waitcnt vmcnt(0)) before the first LDS read
      if constexpr (!ABX) prefetch(xb, z + 2);
      __builtin_amdgcn_sched_barrier(0);               // ...and the scheduler must not sink them below the MFMAs either
      X3_STAMP(3);
      f32x4 acc[NSET][X3_NT];
#pragma unroll
      for (int st = 0; st < NSET; ++st)
#pragma unroll
        for (int i = 0; i < X3_NT; ++i) acc[st][i] = f32x4{0.f, 0.f, 0.f, 0.f};
      const int ring = (z - 1 + 3) % 3;                // slot of plane z-1; plane z+dz-1 (dz = 0..2) sits in slot (ring + dz) % 3
      const int2* const lo_r = loff + ring * (NSL * 4) + kb;      // this lane's offset pairs for the current ring phase
      if (!dg_nomfma)
#pragma unroll
      for (int s = 0; s < NSL; ++s) {
        const int st = s / NSLS;                        // compile time after unrolling
        const uint4 ah = lw[(s * 2 + 0) * 64 + lane];
        const uint4 al = lw[(s * 2 + 1) * 64 + lane];
        const int2 po = lo_r[s * 4];
        uint4 bh[X3_NT], bl[X3_NT];
#pragma unroll
        for (int i = 0; i < X3_NT; ++i) {
          // per-tile base registers whose relation the compiler cannot see (vbt): at a visible constant distance it fuses the
          // SAME pair of two tiles into one ds_read2_b64 — half rate, and its result lands as (tile 0, tile 1) where the MFMA
          // operand wants (pair 0, pair 1) of ONE tile: 96 of the 258 VALU instructions of a plane were the v_movs that undo it
          const char* const a0 = lbytes + (dg_noread ? 0 : vbt[i] + po.x);
          const char* const a1 = lbytes + (dg_noread ? 0 : vbt[i] + po.y);
          const uint2 h0 = *reinterpret_cast<const uint2*>(a0), h1 = *reinterpret_cast<const uint2*>(a1);
          bh[i] = make_uint4(h0.x, h0.y, h1.x, h1.y);
          if constexpr (!BF) {
            const uint2 l0 = *reinterpret_cast<const uint2*>(a0 + LO_BYTES), l1 = *reinterpret_cast<const uint2*>(a1 + LO_BYTES);
            bl[i] = make_uint4(l0.x, l0.y, l1.x, l1.y);
          }
        }
        // term-major order: consecutive MFMAs hit different accumulators (a dependent pair is X3_NT instructions apart)
#pragma unroll
        for (int i = 0; i < X3_NT; ++i) acc[st][i] = x3_mma<BF>(ah, bh[i], acc[st][i]);
        if constexpr (!BF)
#pragma unroll
        for (int i = 0; i < X3_NT; ++i) acc[st][i] = x3_mma<BF>(ah, bl[i], acc[st][i]);
#pragma unroll
        for (int i = 0; i < X3_NT; ++i) acc[st][i] = x3_mma<BF>(al, bh[i], acc[st][i]);
      }
      X3_STAMP(4);
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
            const float u = fmaxf(fmaf(acc[st][i][r], par[st * 16 + 4 * kb + r], par[32 + st * 16 + 4 * kb + r]), act_floor);   // ReLU, or the identity (floor = NaN)
            sum = st == 0 ? u : sum + u;                 // (no `0 + u`: the compiler keeps that add for the sign of zero)
          }
          v[r] = sum;
        }
        if (a.store_main && inside && g < ngroups && !(dg_nostore && v[0] != 12345.f)) {
          T* py = static_cast<T*>(a.y) + b * a.y_bstride + (int64_t)my_ych * DHW + vox;
#pragma unroll
          for (int r = 0; r < 4; ++r) st(py + r * DHW, v[r]);      // whole 4-channel output groups only (x3_eligible)
        }
        if constexpr (ABX) {
          // rows 12..15 (lane quarter 3): stem3d0's fused tail, already through its BatchNorm and activation above
          if (e.src.tail_rows && g == 3 && inside && !(dg_nostore && v[0] != 12345.f)) {
            if (e.src.tail_g4) {
              st4(tdst + ((int64_t)(e.src.tail_ch0 >> 2) * DHW + vox) * 4, v);
            } else {
#pragma unroll
              for (int r = 0; r < 4; ++r) st(tdst + (int64_t)(e.src.tail_ch0 + r) * DHW + vox, v[r]);
            }
          }
        }
        if constexpr (TAILS) {
          f32x4 tacc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
          for (int r = 0; r < 4; ++r)
            tacc = __builtin_amdgcn_mfma_f32_16x16x4f32(reinterpret_cast<const float*>(ltail)[r * 64 + lane], v[r], tacc, 0, 0, 0);
          // destination of this lane quarter's tail: `tbase` (per item: sample, first channel, this lane's voxel of tile 0 in plane 0)
          // + the plane and the tile — one 64-bit pointer in registers instead of the descriptor fields it was built from
          if (my_tail_cout > 0 && inside && !(dg_nostore && tacc[0] != 12345.f)) {
            const int trelu = my_trelu;
            const float4 tsc = *reinterpret_cast<const float4*>(par + 64 + 4 * kb), tsh = *reinterpret_cast<const float4*>(par + 80 + 4 * kb);
            const float sc4[4] = {tsc.x, tsc.y, tsc.z, tsc.w}, sh4[4] = {tsh.x, tsh.y, tsh.z, tsh.w};
            if (a.tail_g4) {      // G4 destination (four output channels: fill_tails): one 16-byte (bf16: 8-byte) store per voxel
              float u4[4];
#pragma unroll
              for (int r = 0; r < 4; ++r) { const float u = fmaf(tacc[r], sc4[r], sh4[r]); u4[r] = trelu ? fmaxf(u, 0.f) : u; }
              st4(tbase + ((int64_t)z * HW + (nt & 1) * 16) * 4, u4);
            } else {
              T* pt = tbase + (int64_t)z * HW + (nt & 1) * 16;
#pragma unroll
              for (int r = 0; r < 4; ++r)
                if (r < my_tail_cout) {
                  float u = fmaf(tacc[r], sc4[r], sh4[r]);
                  st(pt + r * DHW, trelu ? fmaxf(u, 0.f) : u);
                }
            }
          }
          if constexpr (TAILS == 2) {
            // down-sampling tail, first half: blend the raw tail values of the source pair (x = 2X, 2X+1: this lane and the next)
            // and park the result in the LDS for the finishing step (even lanes: X' = 8 (x half) + n / 2)
            const int dl = kb - a.ntail;
            if (dl >= 0 && dl < a.ndown) {
              const float4 xt = ldxt[8 * (nt & 1) + (n >> 1)];
              float ux[4];
#pragma unroll
              for (int r = 0; r < 4; ++r) {
                // w0 * (this lane's value) + w1 * (the next lane's): the row shift rides on the multiply's operand (v_mul_f32_dpp)
                const float p1 = __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(tacc[r]), 0x101, 0xF, 0xF, false));   // row_shl:1
                ux[r] = lerp2(xt.x, xt.z != 0.f ? p1 : tacc[r], xt.y, p1);
              }
              if (!(n & 1))
                reinterpret_cast<float4*>(ldu)[(((z & 1) * 2 + dl) * X3_TY + (nt >> 1)) * (X3_TX / 2) + 8 * (nt & 1) + (n >> 1)] =
                    make_float4(ux[0], ux[1], ux[2], ux[3]);
            }
          }
        }
      }
      if constexpr (ABX) {
        // the plane source's six 16-byte loads per voxel come from L2 (the planes are a few MB): requested HERE, behind the epilogue,
        // their 24 registers are live neither under the MFMAs nor in the epilogue (in front of the matrix block, where the tensor
        // loads sit, the kernel spilled 13 registers; behind it, 15, two of them reloaded in every epilogue.  Half of them — the A
        // terms — in front of or behind the matrix block does fit, and measured the same step time: the staging's cost is its
        // arithmetic, not the wait)
        __builtin_amdgcn_sched_barrier(0);
        prefetch(xb, z + 2);
      }
      X3_STAMP(5);
#ifdef RAGMI_DIAG
      ++dg_steps;
#endif
    }
      if (!again) break;
    }
    if constexpr (TAILS == 2) {
      __syncthreads();                                 // the segment's last (odd) plane is in the LDS
      down_finish(ze - 1, b, y0, x0);
    }
  }
#ifdef RAGMI_DIAG
  if (dg_stamp) {
    X3_STAMP(6);
    const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    if (lane == 0) {
      unsigned long long* const o = x3_stamp_buf + ((int64_t)(blockIdx.y * gridDim.x + blockIdx.x) * X3_WAVES + wave) * X3_STAMP_WORDS;
      for (int k = 0; k < 7; ++k) o[k] = dg_sum[k];
      o[7] = dg_steps; o[8] = dg_t0; o[9] = t1; o[10] = dg_r0; o[11] = r1;
      o[12] = __builtin_amdgcn_s_getreg((4 << 11) | 20);      // HW_REG_XCC_ID (speed diagnostics only)
    }
  }
#endif
}

// ---------------------------------------------------------------------------------------------------------------
// Deep-level form (Matching-Net levels 6 and 12, rag_model.py:236-268: 8 or 16 input channels per set, volumes of 2^19 .. 2^16
// voxels).  The z-marching kernel above starves there: its 8 x 32 columns give a level-12 volume (16 x 32 x 104) 16 columns, and
// with 4-channel operand records every K-slice costs four 8-byte LDS reads, a table read and address arithmetic per lane.
//  * Operand records are 8 CHANNELS of one voxel (16 bytes), [cg8][z][y][x] in LDS: one ds_read_b128 per operand.  A K-slice (32) is
//    4 taps x 8 channels or 2 taps x 16 channels, taps in their linear order (dz, dy, dx) — the tap-major pair order of the packed
//    fragments (x3_pack_one), so the fragments are used as packed: 27 taps in 28 K slots (7 / 14 slices per set; round 3 used
//    {dx .. dx+3} of one row per slice: 36 slots, 9 / 18 slices, and a fragment gather at every launch).  A lane quarter's record
//    address for slice s is a per-lane constant of the launch (NSLS registers set up once): base + immediate in the K loop, no VALU
//    work.  The padding slot (tap 27, zero weights) re-reads tap 26: a voxel INSIDE the 3x3x3 window, so a non-finite value can only
//    reach outputs whose window holds it (round 3's fourth dx read one voxel past the window: Inf x 0 = NaN one voxel further).
//  * Box tiles 2 x 8 x 16 (z, y, x): 16 column tiles = 8 waves x 2, so a level-12 volume is 224 boxes per output-channel block
//    instead of 16 columns; the halo (4 x 10 x 18) comes from L2 (these volumes are 1.7-14 MB).
//  * One SET (input tensor) per stage: the box of set 0 is multiplied while set 1's travels HBM/L2 -> registers, and the LDS holds
//    one set's box (23-46 KB) next to the workgroup's weight fragments (28-56 KB), which stay resident for the whole launch
//    (persistent workgroups; copied as packed by ragmi_conv3d_k3_pack).
//  * Output-channel blocks of 16 ride blockIdx.y.  (Two blocks per workgroup sharing the operand reads were measured at level 6:
//    64 us against 58 — their weights push the workgroup to 98 KB of LDS, one per CU, and the stages serialise.)
constexpr int XD_TY = 8, XD_TX = 16;
constexpr int XD_HY = XD_TY + 2, XD_HX = XD_TX + 2;
constexpr int64_t XD_MIN_VOXELS = 1 << 14;
constexpr int XD_THREADS = 512;                        // 8 waves: wave w owns rows (w & 3) * 2, +1 of planes w >> 2 (, + 2)

// COGS = output-channel blocks of 16 per workgroup (they share every operand read).  WS = the weight fragments of ONE set live in
// LDS and are re-staged with every stage's box (from L2, 37 KB) instead of all sets' staying resident: what lets two blocks
// per workgroup (8 channels per set) still fit two workgroups per CU.
// W16 = 16 waves per workgroup, ONE column tile each (wave w: row w & 7 of plane w >> 3) instead of 8 waves with two: for the
// level-12 cells, whose 103 KB of LDS allow one workgroup per CU — 8 waves, two per SIMD, nothing to hide a stage's latencies
// behind; with 16 the same box is staged by twice the threads and every wave's MFMA chain is half as long.
template <class T, int CH8, int NSET, int COGS, bool WS, bool W16 = false>
__global__ __launch_bounds__(W16 ? 1024 : XD_THREADS, W16 ? 1 : (WS ? 4 : 2)) void conv3d_x3d_kernel(K3Args a, X3Extra e) {
  constexpr int THREADS = W16 ? 1024 : XD_THREADS;
  static_assert(!WS || NSET == 2, "per-stage weights only pay with two sets");
  constexpr int XD_TZ = 2, XD_HZ = XD_TZ + 2, XD_PL = XD_HZ * XD_HY * XD_HX, XD_NT = W16 ? 1 : XD_TZ;
  constexpr bool BF = std::is_same<T, bf16_t>::value;
  constexpr int NCG4 = 2 * CH8;                         // 4-channel groups per set: the staging unit and the packed fragments' unit
  constexpr int NSLS = (NCG4 * 27 + 7) / 8;             // K-slices per set, as packed: 7 (8 channels: 4 taps each) or 14 (16: 2 taps)
  constexpr int NPF = (NCG4 * XD_PL + THREADS - 1) / THREADS;
  constexpr int REC = CH8 * XD_PL;                      // 16-byte records per copy (hi or lo)
  extern __shared__ __attribute__((aligned(16))) uint4 xd_lds[];
  uint4* const lw = xd_lds + (BF ? 1 : 2) * REC;        // [set][block][slice][hi/lo][64 lanes]
  float* const par = reinterpret_cast<float*>(lw + (WS ? 1 : NSET) * COGS * NSLS * 2 * 64);   // scale[set][block][16] | shift[...] | 3 max slots
  unsigned* const lmaxp = reinterpret_cast<unsigned*>(par + 2 * NSET * COGS * 16);   // fp32 storage: largest |x| of a stage's box, three rotating slots
  uint2* const lhi2 = reinterpret_cast<uint2*>(xd_lds);
  uint2* const llo2 = reinterpret_cast<uint2*>(xd_lds + REC);
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, n = lane & 15, kb = lane >> 4;
  const int cog0 = blockIdx.y * COGS, ncog = (a.Cout + 15) >> 4;
  const int HW = a.H * a.W;
  float act_floor = (a.relu & 1) ? 0.f : __builtin_nanf("");   // max(u, NaN) = u: the identity, NaN inputs included
  asm volatile("" : "+v"(act_floor));      // opaque: otherwise the compiler turns max(u, floor) back into max(u, 0) + a select per value
  const int64_t DHW = (int64_t)HW * a.D;
  // weight fragments of this workgroup's output blocks, as packed (tap-major pairs: lane quarter kb of slice s holds the 8 channels
  // 8 (kb & 1) .. +7 of tap 2 s + (kb >> 1) for 16 channels per set, all 8 channels of tap 4 s + kb for 8)
  constexpr int NWS = COGS * NSLS * 2 * 64;              // uint4 words of one set's fragments
  // (all loads of a set first, then the LDS stores: written as load -> store per element the loop waits out one L2 round trip per
  // iteration — four to seven of them in front of the first box of every deep-level launch)
  auto wcopy = [&](int set, int region) {
    constexpr int ITER = (NWS + THREADS - 1) / THREADS;
    uint4 wv[ITER];
#pragma unroll
    for (int it = 0; it < ITER; ++it) {
      const int i = min(it * THREADS + tid, NWS - 1);
      const int cl = i / (NSLS * 2 * 64), r = i % (NSLS * 2 * 64);
      const uint4 v = e.wf[set][(int64_t)min(cog0 + cl, ncog - 1) * NSLS * 2 * 64 + r];      // (unconditional, clamped: no branch between the loads)
      wv[it] = cog0 + cl < ncog ? v : make_uint4(0u, 0u, 0u, 0u);
    }
#pragma unroll
    for (int it = 0; it < ITER; ++it) {
      const int i = it * THREADS + tid;
      if (i < NWS) lw[region * NWS + i] = wv[it];
    }
  };
#ifdef RAGMI_DIAG
  const bool diag_nowcopy = (a.relu & 0x800) != 0;
#else
  constexpr bool diag_nowcopy = false;
#endif
  float pf[NPF][4];
  unsigned valid = 0;
  const T* const x = static_cast<const T*>(a.x);
  // this thread's halo elements of a box: (4-channel group, voxel) -> element offset from the set's first channel, and whether the
  // voxel lies inside the volume.  Located once per box and used for every set (the index arithmetic is ~25 VALU instructions per
  // element: per stage it stood in front of the MFMA block of both waves of a SIMD at once)
  int offs[NPF];
  auto locate = [&](int z0, int y0, int x0) {
    valid = 0;
#pragma unroll
    for (int p = 0; p < NPF; ++p) {
      const int el = p * THREADS + tid, cg4 = el / XD_PL, r = el % XD_PL;
      const int xx = r % XD_HX, yy = (r / XD_HX) % XD_HY, zz = r / (XD_HX * XD_HY);
      const int gz = z0 - 1 + zz, gy = y0 - 1 + yy, gx = x0 - 1 + xx;
      const bool ok = cg4 < NCG4 && (unsigned)gz < (unsigned)a.D && (unsigned)gy < (unsigned)a.H && (unsigned)gx < (unsigned)a.W;
      valid |= (ok ? 1u : 0u) << p;
      offs[p] = (int)(min(cg4, NCG4 - 1) * 4 * DHW) + min(max(gz, 0), a.D - 1) * HW + min(max(gy, 0), a.H - 1) * a.W + min(max(gx, 0), a.W - 1);
    }
  };
  // loads of one set's halo box: unconditional, addresses clamped into the volume (zeros are substituted at the commit)
  auto prefetch = [&](const T* xb, int set) {
    const T* xs = xb + (int64_t)set * (8 * CH8) * DHW;
#pragma unroll
    for (int p = 0; p < NPF; ++p)
#pragma unroll
      for (int c = 0; c < 4; ++c) pf[p][c] = ld(xs + c * DHW + offs[p]);
  };
  auto commit = [&](float mul) {
#pragma unroll
    for (int p = 0; p < NPF; ++p) {
      const int el = p * THREADS + tid;
      if (el >= NCG4 * XD_PL) continue;
      const int cg4 = el / XD_PL, r = el % XD_PL;
      const bool ok = (valid >> p) & 1u;
      unsigned l01, l23, h01, h23;
      if constexpr (BF) {
        h01 = x3_split2(ok ? pf[p][0] : 0.f, ok ? pf[p][1] : 0.f, l01); h23 = x3_split2(ok ? pf[p][2] : 0.f, ok ? pf[p][3] : 0.f, l23);
      } else {
        h01 = x3_split2h(ok ? pf[p][0] : 0.f, ok ? pf[p][1] : 0.f, mul, l01); h23 = x3_split2h(ok ? pf[p][2] : 0.f, ok ? pf[p][3] : 0.f, mul, l23);
      }
      const int d = ((cg4 >> 1) * XD_PL + r) * 2 + (cg4 & 1);
      lhi2[d] = make_uint2(h01, h23);
      if constexpr (!BF) llo2[d] = make_uint2(l01, l23);
    }
  };
  // this wave's two column tiles: rows (tz, ty0) and (tz, ty0 + 1) of the box; lane quarter -> (channel half, dx shift)
  const int tz = W16 ? wave >> 3 : wave >> 2, ty0 = W16 ? (wave & 7) : (wave & 3) * 2;
  const int vb0 = (((CH8 == 2 ? (kb & 1) : 0) * XD_PL) + (tz * XD_HY + ty0) * XD_HX + n) * (int)sizeof(uint4);
  // this lane quarter's operand record of every K-slice: its tap's (dz, dy, dx) offset in the box, fixed for the launch
  int vsl[NSLS];
#pragma unroll
  for (int sl = 0; sl < NSLS; ++sl) {
    const int tap = min(CH8 == 2 ? 2 * sl + (kb >> 1) : 4 * sl + kb, 26);         // the padding slot re-reads tap 26 (zero weights)
    vsl[sl] = vb0 + ((tap / 9 * XD_HY + (tap / 3) % 3) * XD_HX + tap % 3) * (int)sizeof(uint4);
  }
  const char* const lbytes = reinterpret_cast<const char*>(xd_lds);
  constexpr int LO_BYTES = REC * (int)sizeof(uint4);
  const int ngroups = (a.Cout + 3) >> 2;
  int my_ych[COGS];
#pragma unroll
  for (int cl = 0; cl < COGS; ++cl) {
    const int g = (cog0 + cl) * 4 + kb;
    my_ych[cl] = g < ngroups ? a.y_ch[g < RAGMI_MAX_GROUPS ? g : 0] : 0;
  }
  auto decode = [&](int work, int& b, int& z0, int& y0, int& x0) {
    int t = work;
    x0 = (t % a.tiles_x) * XD_TX; t /= a.tiles_x;
    y0 = (t % a.tiles_y) * XD_TY; t /= a.tiles_y;
    z0 = (t % a.tiles_z) * XD_TZ; b = t / a.tiles_z;
  };
  // XCD-aware schedule (as above): every XCD walks one contiguous chunk of the x-fastest box list
  const int chunk = (e.nwork + 7) / 8;
  auto work_of = [&](int j) { return (j >> 3) < chunk ? (j & 7) * chunk + (j >> 3) : e.nwork; };
  int j = blockIdx.x;
  while (j < chunk * 8 && work_of(j) >= e.nwork) j += gridDim.x;
  int b = 0, z0 = 0, y0 = 0, x0 = 0, stage = 0;
  if (j < chunk * 8) {
    decode(work_of(j), b, z0, y0, x0);
    locate(z0, y0, x0);
    prefetch(x + b * a.x_bstride, 0);
  }
  // (the first box's loads are in flight: the weight fragments and parameters travel L2 -> LDS beside them, not in front of them)
  if constexpr (!WS) {
    if (!diag_nowcopy)
#pragma unroll
    for (int set = 0; set < NSET; ++set) wcopy(set, set);
  }
  // (compile-time `set` in an unrolled loop: a descriptor array of the kernel arguments indexed by a lane-dependent value costs a
  // vector load of the pointer plus a dependent one of the value — serial memory round trips in front of the first box)
#pragma unroll
  for (int set = 0; set < NSET; ++set) {
    const float* const psc = a.scale[set];
    const float* const psh = a.shift[set];
    if (tid < COGS * 16) {      // (THREADS >= 512 > COGS * 16; loads unconditional and clamped: no branch, one wait for all of them)
      const int co = cog0 * 16 + tid, cc = min(co, a.Cout - 1);
      const float vsc = psc ? psc[cc] : 1.f, vsh = psh ? psh[cc] : 0.f;
      float sc = co < a.Cout ? vsc : 1.f;
      if constexpr (!BF) { const float wm = e.wmul[set][cc]; sc *= (co < a.Cout ? wm : 1.f); }          // undo the per-channel weight scale 2^k
      par[set * COGS * 16 + tid] = sc;
      par[NSET * COGS * 16 + set * COGS * 16 + tid] = co < a.Cout ? vsh : 0.f;
    }
  }
  if (tid < 3) lmaxp[tid] = 0u;
  __syncthreads();      // the slots are zero before any wave's first atomicMax (waves that skip the loops above arrive there early)
#ifdef RAGMI_DIAG
  const bool diag_nostore = (a.relu & 0x100) != 0, diag_nomfma = (a.relu & 0x200) != 0, diag_nostage = (a.relu & 0x400) != 0;
#else
  constexpr bool diag_nostore = false, diag_nomfma = false, diag_nostage = false;
#endif
  while (j < chunk * 8) {
    int jn = j + gridDim.x;
    while (jn < chunk * 8 && work_of(jn) >= e.nwork) jn += gridDim.x;
    int bn = b, zn = z0, yn = y0, xn = x0;              // the next box (this one again after the last: its loads are discarded)
    if (jn < chunk * 8) decode(work_of(jn), bn, zn, yn, xn);
    f32x4 acc[NSET][COGS][XD_NT];
    float inv_mul[NSET];                               // fp32 storage: 2^e of each set's box (the epilogue undoes the operand scale)
#pragma unroll
    for (int st = 0; st < NSET; ++st) {
      inv_mul[st] = 1.f;
#pragma unroll
      for (int cl = 0; cl < COGS; ++cl)
#pragma unroll
        for (int i = 0; i < XD_NT; ++i) acc[st][cl][i] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
    static_for<NSET>([&](auto st_) {
      constexpr int st = decltype(st_)::value;
      float mul = 1.f;
      // (nothing of this stage may move up into the previous stage's MFMA block: hipcc hoisted the maximum of the values that block's
      // prefetch had just requested into it — an s_waitcnt vmcnt on loads a few instructions old, a memory round trip per stage)
      __builtin_amdgcn_sched_barrier(0);
      if constexpr (!BF) {
        // the box's operand scale: its largest |x| lands at 2^13..2^14 (exact maximum: no headroom needed).  Slot k % 3 of three:
        // the slot of stage k + 2 is cleared behind this stage's second barrier, two barriers ahead of its next use
        float m = 0.f;
#pragma unroll
        for (int p = 0; p < NPF; ++p) {
          const float mp = x3_scalable_max4(pf[p][0], pf[p][1], pf[p][2], pf[p][3]);     // (Inf / NaN / >= 2^115 take no part)
          m = fmaxf(m, ((valid >> p) & 1u) ? mp : 0.f);
        }
        m = x3_wave_max(m);
        if (lane == 0) atomicMax(lmaxp + stage % 3, __float_as_uint(m));
      }
      __syncthreads();                                 // the previous stage's operand reads are done (first pass: the tables are written)
      if constexpr (!BF) {
        mul = x3_pow2_scale(__uint_as_float(lmaxp[stage % 3]), 16384.f);
        inv_mul[st] = 1.f / mul;
      }
      if (!diag_nostage) commit(mul);
      if constexpr (WS) wcopy(st, 0);
      __syncthreads();
      if constexpr (!BF) {
        if (tid == 0) lmaxp[(stage + 2) % 3] = 0u;
        ++stage;
      }
      if (!diag_nostage) {
        if constexpr (st + 1 < NSET) prefetch(x + b * a.x_bstride, st + 1);
        else { locate(zn, yn, xn); prefetch(x + bn * a.x_bstride, 0); }
      }
      __builtin_amdgcn_sched_barrier(0);               // the loads stay ahead of the MFMA block
      if (!diag_nomfma)
#pragma unroll
      for (int sl = 0; sl < NSLS; ++sl) {
        constexpr int ROWB = XD_HX * (int)sizeof(uint4);
        uint4 bh[XD_NT], bl[XD_NT];
#pragma unroll
        for (int i = 0; i < XD_NT; ++i) {
          const int d = (i & 1) * ROWB + (i >> 1) * 2 * XD_HY * ROWB;        // compile time: tile i = row +(i & 1), plane +2 (i >> 1)
          bh[i] = *reinterpret_cast<const uint4*>(lbytes + vsl[sl] + d);
          if constexpr (!BF) bl[i] = *reinterpret_cast<const uint4*>(lbytes + vsl[sl] + d + LO_BYTES);
        }
#pragma unroll
        for (int cl = 0; cl < COGS; ++cl) {
          constexpr int wr = WS ? 0 : st;           // LDS region of this stage's set
          const uint4 ah = lw[(((wr * COGS + cl) * NSLS + sl) * 2 + 0) * 64 + lane];
          const uint4 al = lw[(((wr * COGS + cl) * NSLS + sl) * 2 + 1) * 64 + lane];
#pragma unroll
          for (int i = 0; i < XD_NT; ++i) acc[st][cl][i] = x3_mma<BF>(ah, bh[i], acc[st][cl][i]);
          if constexpr (!BF)
#pragma unroll
          for (int i = 0; i < XD_NT; ++i) acc[st][cl][i] = x3_mma<BF>(ah, bl[i], acc[st][cl][i]);
#pragma unroll
          for (int i = 0; i < XD_NT; ++i) acc[st][cl][i] = x3_mma<BF>(al, bh[i], acc[st][cl][i]);
        }
      }
    });
    // epilogue: lane holds channels 4 g + reg (g = 4 block + kb) of voxel n of each column tile
#pragma unroll
    for (int cl = 0; cl < COGS; ++cl) {
      const int g = (cog0 + cl) * 4 + kb;
#pragma unroll
      for (int i = 0; i < XD_NT; ++i) {
        const int gz = z0 + tz + 2 * (i >> 1), gy = y0 + ty0 + (i & 1), gx = x0 + n;
        const bool inside = gz < a.D && gy < a.H && gx < a.W;
        float v[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          float sum = 0.f;
#pragma unroll
          for (int st = 0; st < NSET; ++st) {
            const float u = fmaxf(fmaf(acc[st][cl][i][r], par[(st * COGS + cl) * 16 + 4 * kb + r] * inv_mul[st], par[NSET * COGS * 16 + (st * COGS + cl) * 16 + 4 * kb + r]), act_floor);
            sum = st == 0 ? u : sum + u;                 // ReLU, or the identity (floor = NaN); no `0 + u`
          }
          v[r] = sum;
        }
        if (inside && g < ngroups && !(diag_nostore && v[0] != 12345.f)) {
          T* py = static_cast<T*>(a.y) + b * a.y_bstride + (int64_t)my_ych[cl] * DHW + ((int64_t)gz * HW + gy * a.W + gx);
#pragma unroll
          for (int r = 0; r < 4; ++r) st(py + r * DHW, v[r]);      // whole 4-channel output groups only (x3d_eligible)
        }
      }
    }
    j = jn; b = bn; z0 = zn; y0 = yn; x0 = xn;
  }
}

bool x3d_eligible(const K3Args& a, int nset, int dtype) {
  if ((dtype != RAGMI_F32X3 && dtype != RAGMI_BF16) || a.res != nullptr || a.ntail > 0 || a.ndown > 0 || !a.store_main) return false;
  const int nc = a.nchunks[0];
  if ((nc != 2 && nc != 4) || (nset == 2 && a.nchunks[1] != nc) || a.Cin != nset * nc * 4 || a.Cout % 4 != 0) return false;
  // volumes only (the depth-1 Feature-Net convolutions would idle half of every 2-deep box), and big enough that the persistent
  // grid has boxes to walk: below 2^14 voxels (128 boxes) the fp32 kernel's latency is the same and its arithmetic exact
  // (per SAMPLE, not per batch: which kernel a pair runs on must not depend on how the batch is split over ranks)
  if (a.D < 2 || a.W < 8 || (int64_t)a.D * a.H * a.W < XD_MIN_VOXELS || (int64_t)a.Cin * a.D * a.H * a.W >= (1ll << 31)) return false;
  return true;
}

template <class T, int CH8, int NSET, int COGS, bool WS, bool W16 = false>
static int x3d_launch_one(K3Args a, X3Extra e, hipStream_t st) {
  constexpr int THREADS = W16 ? 1024 : XD_THREADS;
  constexpr int PL = 4 * XD_HY * XD_HX;
  constexpr size_t lds = (size_t)(std::is_same<T, bf16_t>::value ? 1 : 2) * CH8 * PL * sizeof(uint4) +
                         (size_t)(WS ? 1 : NSET) * COGS * ((2 * CH8 * 27 + 7) / 8) * 2 * 64 * sizeof(uint4) + (size_t)(2 * NSET * COGS * 16 + 4) * sizeof(float);
  static_assert(lds <= 160 * 1024, "deep-level tile does not fit the LDS");
  a.tiles_x = (int)ceil_div(a.W, XD_TX); a.tiles_y = (int)ceil_div(a.H, XD_TY); a.tiles_z = (int)ceil_div(a.D, 2);
  const int64_t nwork = (int64_t)a.tiles_x * a.tiles_y * a.tiles_z * a.B;
  RAGMI_REQUIRE(nwork < (1ll << 28), RAGMI_EUNSUPPORTED, "conv3d_x3d: too many tiles");
  e.nwork = (int)nwork;
  static LaunchState state;
  const int slots = state.slots((const void*)conv3d_x3d_kernel<T, CH8, NSET, COGS, WS, W16>, THREADS, lds, 160 * 1024);
  if (slots <= 0) return fail(RAGMI_ELAUNCH, "conv3d_x3d: cannot raise the dynamic LDS limit");
  const int ny = (int)ceil_div((a.Cout + 15) / 16, COGS);
  int64_t gx = std::max<int64_t>(1, std::min<int64_t>(e.nwork, slots / ny));
  if (gx >= 8) gx -= gx % 8;
  hipLaunchKernelGGL((conv3d_x3d_kernel<T, CH8, NSET, COGS, WS, W16>), dim3((unsigned)gx, (unsigned)ny), dim3(THREADS), lds, st, a, e);
  return check_launch("conv3d_x3d");
}

int x3d_launch(K3Args a, int nset, int dtype, hipStream_t st) {
#ifdef RAGMI_DIAG
  static const int diag_xd = [] { const char* v = getenv("RAGMI_XD_DIAG"); return v ? atoi(v) : 0; }();   // 1 no stores, 2 no MFMA block, 4 no staging
  a.relu |= diag_xd << 8;
#endif
  X3Extra e{};
  x3_weight_sections(e, a, nset, dtype);
  const bool bf = dtype == RAGMI_BF16;
  const int ch8 = a.nchunks[0] / 2;
  // Measured and not shipped (same box, tools/ab_bench.sh): for 8 channels per set and two sets, TWO output blocks per workgroup
  // sharing every operand read (a third fewer LDS bytes per MFMA) with one set's weights at a time in LDS (WS = true: re-staged
  // from L2 with every box, so that two workgroups still fit a CU): 72.8 us against 57.5 for the level-6 launch — at the 128-VGPR
  // cap of four waves per SIMD the kernel spills 37 registers, and the per-stage weight loads sit in front of every stage.
#define RAGMI_XD(CH8_, NSET_, COGS_, WS_) (bf ? x3d_launch_one<bf16_t, CH8_, NSET_, COGS_, WS_>(a, e, st) : x3d_launch_one<float, CH8_, NSET_, COGS_, WS_>(a, e, st))
  // 8 channels per set, two sets, more than one output block (the level-6 dual cells: 8 + 8 -> 24): BOTH blocks in one workgroup —
  // every operand record is read once for two blocks' MFMAs (0.67 LDS reads per MFMA instead of 1) and the box is staged once
  // instead of twice.  With the fragments as packed (7 slices per set) that is 56 + 23 KB of LDS: two workgroups per CU still fit.
  // Same box, both builds (tools/ab_bench.sh): level-6 launch 58.9 -> 46.0 us, step 1.225 -> 1.200 ms.  (Round 2 measured the same
  // idea at 9 slices per set — 98 KB, one workgroup per CU — and lost: 64 vs 58 us.)
#ifndef RAGMI_XD_COGS1
  if (ch8 == 1 && nset == 2 && a.Cout > 16) return RAGMI_XD(1, 2, 2, false);
#endif
  if (ch8 == 1) return nset == 2 ? RAGMI_XD(1, 2, 1, false) : RAGMI_XD(1, 1, 1, false);
  // 16 channels per set, two sets (the level-12 dual cells): one workgroup per CU by LDS -> 16 waves (W16 above)
#ifndef RAGMI_XD_NOW16
  if (nset == 2) return bf ? x3d_launch_one<bf16_t, 2, 2, 1, false, true>(a, e, st) : x3d_launch_one<float, 2, 2, 1, false, true>(a, e, st);
#endif
  return nset == 2 ? RAGMI_XD(2, 2, 1, false) : RAGMI_XD(2, 1, 1, false);
#undef RAGMI_XD
}

// ---------------------------------------------------------------------------------------------------------------
int64_t x3_packed_words(int Cout, int Cin) {
  return 2 * x3_frag_words(Cout, Cin) + (int64_t)((Cout + 15) / 16) * 16;     // bf16 fragments | fp16 fragments | multipliers
}

// the sections of ragmi_conv3d_k3_pack_ex in one launch: workgroups [0, nb_k3) fill the fp32-MFMA section, the next nb_x3 the
// bf16 fragments, the last nb_x3 the scaled fp16 fragments and their multipliers (a training step packs ~150 weights; each launch
// it does not make is ~3.5 us)
// poison != 0 (ragmi_conv3d_k3_pack_for(RAGMI_F32): only the fp32-MFMA section is filled): the first fragment of the bf16 and of
// the fp16 section and every multiplier become NaN, so a convolution that consumes such a pack under another contract returns NaN
// everywhere instead of the products of uninitialised memory.
__global__ void pack_both_kernel(const float* __restrict__ w, float* __restrict__ packed, int64_t total_k3, int nb_k3, int nb_x3, int Cout, int Cin,
                                 int nchunks, int nsls, int ncog, int transpose, int planar, int poison) {
  if ((int)blockIdx.x < nb_k3) {
    const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (idx < total_k3) packed[idx] = k3_pack_value(w, Cout, Cin, nchunks, idx, transpose, planar);
    if (poison && blockIdx.x == 0) {
      const int64_t fw = (int64_t)ncog * nsls * 2 * 64 * 4;
      unsigned* const sec = reinterpret_cast<unsigned*>(packed + total_k3);
      for (int c = 0; c < ncog; ++c)        // hi fragment of slice 0 of every output block: 64 lanes x 4 words
        for (int i = threadIdx.x; i < 256; i += 256) {
          sec[(int64_t)c * nsls * 2 * 64 * 4 + i] = 0x7fc07fc0u;             // bf16 NaN pairs
          sec[fw + (int64_t)c * nsls * 2 * 64 * 4 + i] = 0x7e007e00u;        // fp16 NaN pairs
        }
      for (int i = threadIdx.x; i < ncog * 16; i += 256) sec[2 * fw + i] = 0x7fc00000u;
    }
  } else {
    const int half = ((int)blockIdx.x - nb_k3) >= nb_x3 ? 1 : 0;      // block-uniform
    __shared__ unsigned rowmax[64];
    if (half) x3_row_max(w, rowmax, Cout, Cin, transpose, planar, (int)threadIdx.x, 256);
    const int64_t fw = (int64_t)ncog * nsls * 2 * 64 * 4;
    x3_pack_one(w, reinterpret_cast<uint4*>(packed + total_k3 + half * fw), packed + total_k3 + 2 * fw, rowmax, Cout, Cin, nsls, ncog, transpose, planar,
                half, ((int)blockIdx.x - nb_k3 - half * nb_x3) * 256 + threadIdx.x);
  }
}
int pack_both(const float* w, float* packed, int64_t total_k3, int Cout, int Cin, int transpose, int planar, bool all, hipStream_t s) {
  const int ncgs = (Cin + 3) / 4, nsls = (ncgs * 27 + 7) / 8, ncog = (Cout + 15) / 16;
  const int nb_k3 = (int)ceil_div(total_k3, 256), nb_x3 = (int)ceil_div((int64_t)ncog * nsls * 64, 256);
  hipLaunchKernelGGL(pack_both_kernel, dim3((unsigned)(nb_k3 + (all ? 2 * nb_x3 : 0))), dim3(256), 0, s, w, packed, total_k3, nb_k3, nb_x3, Cout, Cin,
                     (Cin + CK - 1) / CK, nsls, ncog, transpose, planar, all ? 0 : 1);
  return RAGMI_OK;
}

// The bf16x3 form pays off on the big level-3 volumes (z-marching columns need many (column, segment) work items to fill the
// chip) with fp32 storage, no residual input and equal-sized sets; everything else stays on the fp32-MFMA kernel.
bool x3_eligible(const K3Args& a, int nset, int dtype) {
  // the caller asks for it through the dtype argument (include/rag_amd.h): RAGMI_F32 never comes here
  if ((dtype != RAGMI_F32X3 && dtype != RAGMI_BF16) || a.res != nullptr) return false;
  if (a.Cin % 4 != 0 || a.Cout % 4 != 0) return false;   // whole 4-channel groups in and out (the reference's channel counts all are)
  const int ncg = a.nchunks[0] + (nset == 2 ? a.nchunks[1] : 0);
  if (nset == 2 && (a.nchunks[0] != a.nchunks[1] || a.nchunks[0] > 2)) return false;
  if (nset == 1 && ncg > 6) return false;
  // voxels per SAMPLE: the choice of kernel (hence the rounding) must not depend on how a batch is split over ranks
  if ((int64_t)a.D * a.H * a.W < X3_MIN_VOXELS || a.W < 32 || a.D < 8) return false;
  if ((a.ntail > 0 || a.ndown > 0) && a.Cout > 16) return false;
  if ((int64_t)a.Cin * a.D * a.H * a.W >= (1ll << 31)) return false;
  return true;
}

// which G4 forms (include/rag_amd.h) the kernel this call lands on takes: bit 0 a G4 input, bit 1 G4 full-resolution tails
int x3_g4_caps(const K3Args& a, int nset, int dtype) {
  if ((dtype != RAGMI_F32X3 && dtype != RAGMI_BF16) || x2d_eligible(a, nset, dtype) || x3d_eligible(a, nset, dtype) || !x3_eligible(a, nset, dtype)) return 0;
#ifndef RAGMI_NO_X3Q
  if (xq_takes(a, nset, dtype)) return 3;
#endif
  const int ncg = a.nchunks[0] + (nset == 2 ? a.nchunks[1] : 0);
  // a G4 input: stem3d1's shape (12 channels, one set) in either storage type; under bf16 storage also the level-3 dual cells (4 + 4
  // channels — fp32 storage runs those on conv3d_x3q_kernel, above)
  const bool x_g4 = (nset == 1 && ncg == 3 && a.ndown == 0) || (dtype == RAGMI_BF16 && nset == 2 && ncg == 2);
  return (a.Cout <= 16 ? 2 : 0) | (x_g4 ? 1 : 0);
}

template <class T, int NCG, int NSET, int TAILS, int XSRC = 0>
static int x3_launch_tails(const K3Args& a, const X3Extra& e, dim3 grid, size_t lds, hipStream_t st) {
  if constexpr (XSRC == 0 && NCG == 3 && NSET == 1 && TAILS < 2) {
    if (e.src.ws != nullptr) return x3_launch_tails<T, NCG, NSET, TAILS, 2>(a, e, grid, lds + 2 * NCG * sizeof(float4), st);      // stem3d1 expanding stem3d0's planes (+ stem3d0's BatchNorm per group)
    if (a.relu & RAGMI_CONV_X_G4) return x3_launch_tails<T, NCG, NSET, TAILS, 1>(a, e, grid, lds, st);      // stem3d1 on a G4 input
  }
  if constexpr (XSRC == 0 && NCG == 2 && NSET == 2 && std::is_same<T, bf16_t>::value) {
    if (a.relu & RAGMI_CONV_X_G4) return x3_launch_tails<T, NCG, NSET, TAILS, 1>(a, e, grid, lds, st);      // a level-3 dual cell on a G4 input, bf16 storage
  }
  if ((a.relu & RAGMI_CONV_X_G4) && XSRC != 1) return fail(RAGMI_EUNSUPPORTED, "conv3d_x3: this launch does not take a G4 input (ragmi_conv3d_k3_g4_caps)");
  if (e.src.ws != nullptr && XSRC != 2) return fail(RAGMI_EUNSUPPORTED, "conv3d_x3: the plane source is built for 12 -> Cout fp32 launches without down-sampling tails");
  static LaunchState state;     // per device, mutex-guarded (common.h)
  // persistent grid = the workgroups the chip holds at once (occupancy x CUs): measured on the level-3 launches (1664 work items)
  // with both on one box: 512 workgroups 1.219 ms per step, 1024 1.226, 768 1.296, 1536 1.250
  const int slots = state.slots((const void*)conv3d_x3_kernel<T, NCG, NSET, TAILS, XSRC>, X3_THREADS, lds, 160 * 1024);
  if (slots <= 0) return fail(RAGMI_ELAUNCH, "conv3d_x3: cannot raise the dynamic LDS limit");
  grid.x = (unsigned)std::max<int64_t>(1, std::min<int64_t>(grid.x, std::max(256, slots) / (int)grid.y));
  hipLaunchKernelGGL((conv3d_x3_kernel<T, NCG, NSET, TAILS, XSRC>), grid, dim3(X3_THREADS), lds, st, a, e);
  return check_launch("conv3d_x3");
}
template <class T, int NCG, int NSET>
static int x3_launch_typed(const K3Args& a, const X3Extra& e, dim3 grid, size_t lds, hipStream_t st) {
  if constexpr (NCG <= 3) {      // down-sampling tails: the level-3 launches (<= 3 channel groups)
    if (a.ndown > 0) return x3_launch_tails<T, NCG, NSET, 2>(a, e, grid, lds, st);
  }
  if (a.ndown > 0) return fail(RAGMI_EUNSUPPORTED, "conv3d_x3: down-sampling tails are built for <= 12 input channels");
  return a.ntail > 0 ? x3_launch_tails<T, NCG, NSET, 1>(a, e, grid, lds, st) : x3_launch_tails<T, NCG, NSET, 0>(a, e, grid, lds, st);
}
template <int NCG, int NSET>
static int x3_launch_one(const K3Args& a, const X3Extra& e, dim3 grid, size_t lds, hipStream_t st) {
  return e.bf16 ? x3_launch_typed<bf16_t, NCG, NSET>(a, e, grid, lds, st) : x3_launch_typed<float, NCG, NSET>(a, e, grid, lds, st);
}

#ifdef RAGMI_DIAG
// profiling builds: where the stamps of the next z-marching launches go (X3_STAMP_WORDS 64-bit words per wave, grid x 8 waves); null = off
extern "C" __attribute__((visibility("default"))) int ragmi_diag_x3_stamp_buffer(void* buf) {
  unsigned long long* p = static_cast<unsigned long long*>(buf);
  return hipMemcpyToSymbol(HIP_SYMBOL(x3_stamp_buf), &p, sizeof(p)) == hipSuccess ? 0 : -1;
}
#endif

// a: as filled for the fp32 kernel (wp[s] = packed weights: fp32-MFMA section followed by the bf16x3 fragments)
int x3_launch(K3Args a, int nset, int dtype, hipStream_t st, const X3StemSrc* src) {
#ifdef RAGMI_DIAG
  static const int diag_x3 = [] { const char* v = getenv("RAGMI_X3_DIAG"); return v ? atoi(v) : 0; }();   // 1 no stores, 2 no MFMA block, 4 no commit, 8 no loads, 16 operand reads at one address, 32 in-kernel stamps, 64 / 128 (quad-ring kernel) no finishing step / no parking of the down-sampling tails
  a.relu |= diag_x3 << 8;
#endif
  X3Extra e{};
  if (src) e.src = *src;
  x3_weight_sections(e, a, nset, dtype);
  if (src && src->tail_rows) { K3Args t16 = a; t16.Cout = 16; x3_weight_sections(e, t16, nset, dtype); }      // (the caller's pack is a 16-channel one: RAGMI_TAIL_ROWS)
  const int ncg = a.nchunks[0] + (nset == 2 ? a.nchunks[1] : 0), ncgs = ncg / nset, nsls = (ncgs * 27 + 7) / 8, nsl = nset * nsls;
  a.tiles_x = (int)ceil_div(a.W, X3_TX); a.tiles_y = (int)ceil_div(a.H, X3_TY);
  const int ncog = (a.Cout + 15) / 16;
  // depth segments: enough independent (column, segment) work items to fill several workgroups per CU, at least 8 planes each.
  // (Measured on the level-3 volumes, round 2, both builds on one box: 8 segments 1.214 ms per step; 4: 1.253, 5: 1.262, 6: 1.213,
  // 7: 1.256, 10: 1.305, 13: 1.224, 16: 1.299.  A model that minimises rounds x (planes + 2 halo planes) does not predict this.)
  // fp32 storage: the segmentation must not depend on the batch size — the operand scale is chosen per (column, segment), so where
  // the segments end enters the rounding, and a pair's result must not depend on how a batch is split over ranks
  const int64_t cols = (int64_t)a.tiles_x * a.tiles_y * a.B;
  const int64_t cols_seg = dtype == RAGMI_BF16 ? cols : (int64_t)a.tiles_x * a.tiles_y;
  int nseg = (int)std::max<int64_t>(1, std::min<int64_t>(ceil_div(1536, cols_seg * ncog), ceil_div(a.D, 8)));
#ifdef RAGMI_DIAG
  static const int diag_nseg = [] { const char* v = getenv("RAGMI_X3_NSEG"); return v ? atoi(v) : 0; }();      // profiling builds: depth segments per column
  if (diag_nseg > 0) nseg = std::min(diag_nseg, a.D);
#endif
  e.seg_len = (int)ceil_div(a.D, nseg);
  if (a.ndown > 0) e.seg_len += e.seg_len & 1;         // down-sampling tails pair the planes (2Z, 2Z+1): segments start and end even
  e.nseg = (int)ceil_div(a.D, e.seg_len);
  // Work items.  The persistent grid holds 64 workgroups per XCD (two per CU) and each XCD walks one contiguous eighth of the list, so a
  // sample's (column, segment) pairs are dealt in 8 groups of `grp`; 208 per group at the headline shape = 3 full rounds of 64 and a
  // fourth with 16 of 64 slots busy.  When that remainder is at most half a round its pairs are split into the two halves of their
  // depth segment (`nsplit` per group): twice the items at half the length fill twice the slots.  The cut depends on the sample's
  // shape only, never on the batch size (fp32 storage: the operand scale is chosen per item, so where items end enters the rounding).
  const int64_t per_sample = (int64_t)a.tiles_x * a.tiles_y * e.nseg;
  e.ngrp = per_sample % 8 == 0 ? 8 : 1;
  e.grp = (int)(per_sample / e.ngrp);
  e.nsplit = 0;
  if (e.ngrp == 8 && e.seg_len >= 4) {
    const int rem = e.grp % 64;
    if (rem > 0 && rem <= 32) e.nsplit = rem;
  }
  if (a.ndown > 0) {
    // the halves of a split item must start and end on even planes too (down_finish pairs the planes 2Z, 2Z+1): EVERY item's length
    // a multiple of 4 — also the last segment's, which is D - (nseg - 1) * seg_len planes (D % 4 == 2 left it at 2 mod 4,
    // its odd mid-point made both halves finish the same half-resolution voxel from a stale slot)
    if (e.seg_len % 4 != 0 || (a.D - (e.nseg - 1) * e.seg_len) % 4 != 0) e.nsplit = 0;
    e.dsd = lin_scale(a.D, a.D / 2, 1); e.dsh = lin_scale(a.H, a.H / 2, 1); e.dsw = lin_scale(a.W, a.W / 2, 1);
  }
  const int64_t nwork = (int64_t)a.B * e.ngrp * (e.grp + e.nsplit);
  RAGMI_REQUIRE(nwork < (1ll << 31) && per_sample < (1ll << 28), RAGMI_EUNSUPPORTED, "conv3d_x3: too many tiles");
  e.nwork = (int)nwork;
  e.bf16 = dtype == RAGMI_BF16 ? 1 : 0;
  const size_t lds = (size_t)(dtype == RAGMI_BF16 ? 1 : 2) * ncg * x3_group_stride(ncg, nset) * sizeof(uint2) + (size_t)nsl * 2 * 64 * sizeof(uint4) + (size_t)3 * nsl * 4 * sizeof(int2) +
                     3 * 64 * sizeof(uint4) + 132 * sizeof(float) +
                     (a.ndown > 0 ? (size_t)(2 * 2 * 4 * X3_TY * (X3_TX / 2)) * sizeof(float) + (size_t)(X3_TX / 2 + X3_TY / 2) * sizeof(float4) : 0);
  RAGMI_REQUIRE(lds <= 160 * 1024, RAGMI_EUNSUPPORTED, "conv3d_x3: tile does not fit the LDS");
  const dim3 grid((unsigned)std::min<int64_t>(nwork, 1 << 20), ncog);      // x is cut to the resident slots where the kernel is known
#ifndef RAGMI_NO_X3Q
  if (!src && xq_takes(a, nset, dtype)) return xq_launch(a, e, nset, grid, st);     // one 4-channel group per set, fp32 storage: conv3d_x3q.hip
#endif
  if (nset == 2) {
    switch (ncg) {
      case 2: return x3_launch_one<2, 2>(a, e, grid, lds, st);
      case 4: return x3_launch_one<4, 2>(a, e, grid, lds, st);
      default: return fail(RAGMI_EUNSUPPORTED, "conv3d_x3: dual form with %d channel groups not instantiated", ncg);
    }
  }
  switch (ncg) {
    case 1: return x3_launch_one<1, 1>(a, e, grid, lds, st);
    case 2: return x3_launch_one<2, 1>(a, e, grid, lds, st);
    case 3: return x3_launch_one<3, 1>(a, e, grid, lds, st);
    case 4: return x3_launch_one<4, 1>(a, e, grid, lds, st);
    case 5: return x3_launch_one<5, 1>(a, e, grid, lds, st);
    case 6: return x3_launch_one<6, 1>(a, e, grid, lds, st);
    default: return fail(RAGMI_EUNSUPPORTED, "conv3d_x3: %d channel groups not instantiated", ncg);
  }
}

}  // namespace ragmi
