// Plane-stationary form of the split-operand 3x3x3 convolution (fp32 storage, scaled fp16 halves; conv3d_x3_common.h) for the
// level-3 volumes of the Matching Net (rag_model.py:234-261: stem3d1 12 -> 12 and the dual cells 4 + 4 -> 12): <= 3 input-channel
// groups of 4 per accumulator set.
//
// The z-marching kernel of conv3d_x3.hip computes output plane z from the three input planes z-1, z, z+1 in its LDS ring: every
// (voxel, tap) operand is read from LDS three times (once per output plane it feeds) and the weight fragments of all 27 taps once
// per plane — 59 KB of LDS reads per wave and plane for 48 MFMAs, and the LDS array, not the matrix pipe, sets its pace (counters:
// LDS active 55 % of the launch, matrix pipe a third).  Here input plane s is multiplied ONCE, as it arrives:
//   * its (voxel, (dy, dx)) operands are read once and used against the weight slices of all three dz, accumulating into three
//     ROTATING accumulator sets (outputs s+1, s, s-1): 8 of the 9 in-plane taps of a channel group fill one K-slice of 8 pairs;
//   * the ninth tap (dy, dx) = (2, 2) of the planes z-1, z, z+1 of an OUTPUT plane z shares one K = 16 product per channel group
//     (lane quarter <-> plane), read from the ring when output z is finished — so the matrix-core work per output is unchanged
//     (3 x 3 + 3 products per (set, column tile) against 4 x 3 of the z-marching form);
//   * operand addresses are per-lane CONSTANTS (base + immediate): no offset table, no address arithmetic in the loop (the
//     z-marching form spends 34 of its ~210 vector instructions per plane on them);
//   * the plane loop is unrolled three times, so ring slots and accumulator roles are compile-time names (no register moves).
// LDS reads per wave and plane: 24 operand reads of 8 bytes + the weight fragments (14 KB, shared by the wave's two column
// tiles) = 26 KB.  Work decomposition, halo staging, operand scaling (per column segment, restart on overflow), epilogue and
// fused tails are those of the z-marching kernel.
//
// STATUS (round 3): MEASURED AND NOT SHIPPED — profiles/r03_x3p_investigation.md.  Bit-exact contract and parity as the z-marching
// kernel (86 parity tests green through the C ABI), but on the same box 134 us against 125 us for the dual level-3 launch with its
// fused tails (123 against 127 without tails), 163 against 156 us for stem3d1, equal at batch 8.  To build it into the library
// again: copy this file to rag_amd/csrc/, declare
//     bool x3p_eligible(const K3Args& a, int nset, int dtype);  int x3p_launch(K3Args a, X3Extra e, int nset, hipStream_t st);
// in conv3d_x3_common.h and call `if (x3p_eligible(a, nset, dtype)) return x3p_launch(a, e, nset, st);` in x3_launch
// (conv3d_x3.hip) before the z-marching launch; `make EXTRA=-DRAGMI_X3P_STAMPS` adds the in-kernel stamps and the RAGMI_X3P=0
// switch that tools/experiments/x3p_time.py and x3p_ab.sh use.
#include "conv3d_x3_common.h"

namespace ragmi {

#ifdef RAGMI_X3P_STAMPS
__device__ __forceinline__ unsigned long long x3p_now() {
  unsigned long long t;
  __builtin_amdgcn_sched_barrier(0);
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");
  __builtin_amdgcn_sched_barrier(0);
  return t;
}
#define X3P_STAMP(k) do { const unsigned long long t_ = x3p_now(); tsum[k] += t_ - tlast; tlast = t_; } while (0)
#else
#define X3P_STAMP(k) do { } while (0)
#endif

// Geometry of this form: workgroups of FOUR waves (one per SIMD) owning a 4 x 32 (y, x) tile, two column tiles per wave, two
// workgroups per CU.  With eight waves per workgroup the two waves of a SIMD ran in lockstep between the barriers — both in their
// operand-read latency, then both contending for the matrix pipe (stamps: the first wave of a SIMD spent 29-35 % of its time at the
// barrier waiting for its partner); two independent workgroups fall out of step and fill each other's bubbles.
constexpr int XP_TY = 8, XP_TX = 32, XP_HY = XP_TY + 2, XP_HX = XP_TX + 2, XP_PL = XP_HY * XP_HX;
constexpr int XP_THREADS = 512, XP_WAVES = XP_THREADS / 64, XP_NT = XP_TY * XP_TX / 16 / XP_WAVES;
constexpr int X3P_PARTS = 256;      // equal runs of output planes per sample (= the CUs of an MI355X: one run per CU at batch 1)
constexpr unsigned X3P_OOB = 0x80000000u;      // a buffer offset past every descriptor range used here: loads return 0, stores are dropped
constexpr int X3P_WPS = 2;      // waves per SIMD the register budget is sized for = workgroups per CU (one wave per SIMD each)
typedef _Float16 x3_f16x4 __attribute__((ext_vector_type(4)));
// 16x16x16 product (K = 16: one 4-channel pair per lane quarter) for the ninth-tap slice
__device__ __forceinline__ f32x4 x3_mma16(const uint2& a, const uint2& b, const f32x4& c) {
  return __builtin_amdgcn_mfma_f32_16x16x16f16(__builtin_bit_cast(x3_f16x4, a), __builtin_bit_cast(x3_f16x4, b), c, 0, 0, 0);
}

// NCG = input-channel groups of 4 over all sets, NSET accumulator sets (2: the Cell_3d sibling fusion)
template <int NCG, int NSET, bool TAILS>
__global__ __launch_bounds__(XP_THREADS, X3P_WPS) void conv3d_x3p_kernel(K3Args a, X3Extra e) {
  using T = float;
  constexpr int NCGS = NCG / NSET;
  constexpr int NSLS_V1 = (NCGS * 27 + 7) / 8;          // slices per set in the packed (z-marching) fragment layout
  constexpr int NPF = (NCG * XP_PL + XP_THREADS - 1) / XP_THREADS;
  constexpr int RS = x3_row_stride(NCG), PLS = XP_HY * RS + 2;   // + a spare pair of records per plane: where staging threads without an element write
  static_assert(NCG % NSET == 0 && NCGS <= 3 && NPF <= 32, "bad instantiation");
  extern __shared__ __attribute__((aligned(16))) uint2 x3p_lds[];      // hi[NCG][4][PLS] | lo[NCG][4][PLS] | weights | tails | params
  uint2* const lhi = x3p_lds;
  // records per copy (hi or lo), padded so that the lo copy does NOT sit a multiple of 512 bytes behind the hi copy: at such a
  // distance hipcc fuses a pair's hi and lo read into one ds_read2st64_b64, whose result registers (hi, lo) then have to be moved
  // into the (pair 0, pair 1) operand tuples — 25 v_mov per step
  constexpr int COPY = NCG * 4 * PLS + ((NCG * 4 * PLS) % 64 == 0 ? 2 : 0);
  uint2* const llo = x3p_lds + COPY;
  uint4* const lwm = reinterpret_cast<uint4*>(x3p_lds + 2 * COPY);             // main slices [set][cg][dz][hi/lo][64 lanes]
  uint2* const lwl = reinterpret_cast<uint2*>(lwm + NCG * 3 * 2 * 64);                   // ninth-tap slices [set][cg][hi/lo][64 lanes]
  uint4* const ltail = reinterpret_cast<uint4*>(lwl + NCG * 2 * 64);                     // fused-tail fragments [3][64 lanes]
  // scale[2][16] (times the column's 2^e, rewritten per column) | shift[2][16] | tail scale[4][4] | tail shift[4][4] | static scale[2][16]
  // | the column's running max |x| (float bits)
  float* const par = reinterpret_cast<float*>(ltail + 3 * 64);
  unsigned* const lmaxp = reinterpret_cast<unsigned*>(par + 128);
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, n = lane & 15, kb = lane >> 4;
  const int cog = blockIdx.y;
  const int HW = a.H * a.W;
  float act_floor = (a.relu & 1) ? 0.f : __builtin_nanf("");   // max(u, NaN) = u: the identity, NaN inputs included
  asm volatile("" : "+v"(act_floor));
  const int64_t DHW = (int64_t)HW * a.D;
  // weight fragments, gathered in 8-byte halves (4 channels of one tap) from the packed z-marching layout:
  // main slice (set, cg, dz): A[row m][k = 8 kb + 4 j + c] = w[16 cog + m][4 cg + c][tap 9 dz + 2 kb + j]
  for (int i = tid; i < NCG * 3 * 2 * 64 * 2; i += XP_THREADS) {
    const int half = i & 1;
    int q = i >> 1;
    const int ln = q & 63; q >>= 6;
    const int hl = q & 1; q >>= 1;
    const int dz = q % 3; q /= 3;
    const int cgl = q % NCGS, set = q / NCGS;
    const int P = cgl * 27 + 9 * dz + 2 * (ln >> 4) + half;
    const uint2* const src = reinterpret_cast<const uint2*>(e.wf[set]);
    reinterpret_cast<uint2*>(lwm)[i] = src[((((int64_t)cog * NSLS_V1 + (P >> 3)) * 2 + hl) * 64 + ((P & 7) >> 1) * 16 + (ln & 15)) * 2 + (P & 1)];
  }
  // ninth-tap slice (set, cg): A[row m][k = 4 kb + c] = w[16 cog + m][4 cg + c][tap 9 kb + 8] (kb = dz; quarter 3: zeros)
  for (int i = tid; i < NCG * 2 * 64; i += XP_THREADS) {
    int q = i;
    const int ln = q & 63; q >>= 6;
    const int hl = q & 1; q >>= 1;
    const int cgl = q % NCGS, set = q / NCGS;
    const int dz = ln >> 4;
    const int P = cgl * 27 + 9 * min(dz, 2) + 8;
    const uint2* const src = reinterpret_cast<const uint2*>(e.wf[set]);
    const uint2 v = src[((((int64_t)cog * NSLS_V1 + (P >> 3)) * 2 + hl) * 64 + ((P & 7) >> 1) * 16 + (ln & 15)) * 2 + (P & 1)];
    lwl[i] = dz < 3 ? v : make_uint2(0u, 0u);
  }
  for (int i = tid; i < 32; i += XP_THREADS) {
    const int set = i >> 4, co = cog * 16 + (i & 15);
    const bool ok = set < NSET && co < a.Cout;
    float sc = (ok && a.scale[set]) ? a.scale[set][co] : 1.f;
    sc *= (ok ? e.wmul[set][co] : 1.f);          // undo the per-channel weight scale 2^k
    par[i] = sc;
    par[96 + i] = sc;
    par[32 + i] = (ok && a.shift[set]) ? a.shift[set][co] : 0.f;
  }
  if (tid == 0) *lmaxp = 0u;
  // fused tails: three bf16 parts of the epilogue's values against three parts of the tail weights (conv3d_x3.hip)
  if constexpr (TAILS) {
    unsigned short wh[4], wm[4], wl[4];
    const int row = n, tl = row >> 2, k = row & 3;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int c = cog * 16 + 4 * kb + j;
      float wv = 0.f;
      if (tl < a.ntail && k < a.tail_cout[tl] && c < a.Cout) wv = a.tail_w[tl][k * a.Cout + c];
      wh[j] = x3_bf16_rn(wv);
      const float r1 = wv - __uint_as_float((unsigned)wh[j] << 16);
      wm[j] = x3_bf16_rn(r1);
      wl[j] = x3_bf16_rn(r1 - __uint_as_float((unsigned)wm[j] << 16));
    }
    auto pk2 = [](const unsigned short* p, const unsigned short* q) {
      return make_uint4(p[0] | ((unsigned)p[1] << 16), p[2] | ((unsigned)p[3] << 16), q[0] | ((unsigned)q[1] << 16), q[2] | ((unsigned)q[3] << 16));
    };
    if (tid < 64) {
      ltail[lane] = pk2(wh, wh);
      ltail[64 + lane] = pk2(wh, wm);
      ltail[128 + lane] = pk2(wm, wl);
    }
    if (tid < 16) {
      const int tk = tid >> 2, r = tid & 3;
      const bool ok = tk < a.ntail && r < a.tail_cout[tk < 2 ? tk : 0];
      par[64 + tid] = (ok && a.tail_scale[tk < 2 ? tk : 0]) ? a.tail_scale[tk < 2 ? tk : 0][r] : 1.f;
      par[80 + tid] = (ok && a.tail_shift[tk < 2 ? tk : 0]) ? a.tail_shift[tk < 2 ? tk : 0][r] : 0.f;
    }
  }
  // Halo staging through raw buffer loads: an offset past the descriptor's range returns 0, so voxels outside the plane (offset
  // X3P_OOB, fixed per column), planes outside the depth range (descriptor with range 0, a scalar select per plane) and the zero
  // padding cost no clamping, no validity masks and no selects — per plane the loads are `descriptor + lane offset + scalar offset`.
  float pf[2][NPF][4];                   // two planes in flight: plane q of a ring pass travels in set (q - s0) & 1
  unsigned voff[NPF];
  const T* const x = static_cast<const T*>(a.x);
  auto locate = [&](int y0, int x0) {
#pragma unroll
    for (int p = 0; p < NPF; ++p) {
      const int el = p * XP_THREADS + tid, cg = el / XP_PL, r = el % XP_PL;
      const int xx = r % XP_HX, yy = r / XP_HX;
      const int gy = y0 - 1 + yy, gx = x0 - 1 + xx;
      const bool ok = cg < NCG && (unsigned)gy < (unsigned)a.H && (unsigned)gx < (unsigned)a.W;
      // whole 4-channel groups only (x3_eligible); Cin * DHW * 4 < 2^31 (x3p_eligible)
      voff[p] = ok ? (unsigned)((cg * 4 * DHW + gy * a.W + gx) * 4) : X3P_OOB;
    }
  };
  auto prefetch = [&](const T* xb, int gz, auto set_) {
    constexpr int SET = decltype(set_)::value;
    const bool inr = (unsigned)gz < (unsigned)a.D;
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<T*>(xb), 0, inr ? e.xbytes : 0u, 0x00020000);
    const unsigned zoff = inr ? (unsigned)gz * (unsigned)HW * 4u : 0u;
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      const unsigned soff = zoff + (unsigned)c * (unsigned)DHW * 4u;        // wave-uniform
#pragma unroll
      for (int p = 0; p < NPF; ++p) pf[SET][p][c] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rs, voff[p], soff, 0));
    }
  };
  // LDS record each staging element of this thread lands in (ring slot 0): a thread constant; elements past the last channel group
  // go to a spare record behind the planes, so the staging pieces carry no branch
  int cdst[NPF];
#pragma unroll
  for (int p = 0; p < NPF; ++p) {
    const int el = p * XP_THREADS + tid, cg = el / XP_PL, r = el % XP_PL;
    cdst[p] = el < NCG * XP_PL ? cg * 4 * PLS + (r / XP_HX) * RS + r % XP_HX : XP_HY * RS;     // (the spare record of channel group 0's plane)
  }
  float mul = 1.f;                       // the column's operand scale 2^-e (wave-uniform)
  auto commit_all = [&](int slot) {      // registers (set 0) -> ring plane `slot` (fp16 hi / lo halves); the steps stage in pieces instead
#pragma unroll
    for (int p = 0; p < NPF; ++p) {
      const int el = p * XP_THREADS + tid;
      if (el >= NCG * XP_PL) continue;
      const int cg = el / XP_PL, r = el % XP_PL;
      unsigned l01, l23;
      const unsigned h01 = x3_split2h(pf[0][p][0], pf[0][p][1], mul, l01), h23 = x3_split2h(pf[0][p][2], pf[0][p][3], mul, l23);
      const int d = (cg * 4 + slot) * PLS + (r / XP_HX) * RS + r % XP_HX;
      lhi[d] = make_uint2(h01, h23);
      llo[d] = make_uint2(l01, l23);
    }
  };
  auto local_max = [&](auto set_) {
    constexpr int SET = decltype(set_)::value;
    float m = 0.f;
#pragma unroll
    for (int p = 0; p < NPF; ++p) m = fmaxf(m, fmaxf(fmaxf(fabsf(pf[SET][p][0]), fabsf(pf[SET][p][1])), fmaxf(fabsf(pf[SET][p][2]), fabsf(pf[SET][p][3]))));
    return m;
  };
  auto note_overflow = [&](auto set_) {
    const float m = local_max(set_);
    if (m * mul > X3_F16_CAP) atomicMax(lmaxp, __float_as_uint(m));
  };
  // operand addresses of this lane (bytes from the LDS base, ring slot 0, channel group 0, hi copy): the two pairs of the main
  // slice — taps 2 kb and 2 kb + 1 of the plane — for each of the wave's column tiles, and the ninth tap (2, 2)
  static_assert(XP_NT == 2, "tile addressing below is written for two column tiles per wave");
  auto tap_off = [&](int t) { return (t / 3) * RS + t % 3; };
  const int row0 = (wave * XP_NT) >> 1;                       // both tiles of a wave sit in one row: x halves 0 and 1
  int va0[XP_NT], va1[XP_NT], va8[XP_NT];
#pragma unroll
  for (int i = 0; i < XP_NT; ++i) {
    const int base = (row0 * RS + i * 16 + n) * (int)sizeof(uint2);
    va0[i] = base + tap_off(2 * kb) * (int)sizeof(uint2);
    va1[i] = base + tap_off(2 * kb + 1) * (int)sizeof(uint2);
    va8[i] = base + tap_off(8) * (int)sizeof(uint2);
    asm volatile("" : "+v"(va0[i]), "+v"(va1[i]), "+v"(va8[i]));       // opaque: no ds_read2 fusion across tiles (conv3d_x3.hip)
  }
  const char* const lbytes = reinterpret_cast<const char*>(x3p_lds);
  constexpr int LO_BYTES = COPY * (int)sizeof(uint2);
  constexpr int SLOT_BYTES = PLS * (int)sizeof(uint2), CG_BYTES = 4 * SLOT_BYTES;
  // the weight fragments live in REGISTERS for the whole launch (NCG x 28 VGPRs: the kernel is sized for two waves per SIMD):
  // with them in LDS every step re-read 14 KB per wave and waited for it in front of each group of MFMAs
  __syncthreads();
  uint4 wmh[NCG][3], wml[NCG][3];
  uint2 w8h[NCG], w8l[NCG];
#pragma unroll
  for (int cgi = 0; cgi < NCG; ++cgi) {
#pragma unroll
    for (int dz = 0; dz < 3; ++dz) {
      wmh[cgi][dz] = lwm[((cgi * 3 + dz) * 2 + 0) * 64 + lane];
      wml[cgi][dz] = lwm[((cgi * 3 + dz) * 2 + 1) * 64 + lane];
    }
    w8h[cgi] = lwl[(cgi * 2 + 0) * 64 + lane];
    w8l[cgi] = lwl[(cgi * 2 + 1) * 64 + lane];
  }
  const int g = cog * 4 + kb, ngroups = (a.Cout + 3) >> 2;
  const int my_ych = g < ngroups ? a.y_ch[g < RAGMI_MAX_GROUPS ? g : 0] : 0;
  const int tsel = kb & 1;
  const unsigned plane_bytes = (unsigned)HW * 4u, chan_bytes = (unsigned)DHW * 4u;
#ifdef RAGMI_X3P_STAMPS
  unsigned long long tsum[6] = {0, 0, 0, 0, 0, 0}, tlast = x3p_now();
#endif
  // Work: every SAMPLE's (column, plane) space — columns x-fastest, planes innermost — is cut into X3P_PARTS equal runs of output
  // planes; a work item is one run of one sample: the tail of a column, whole columns, the head of another, each piece a ring pass
  // of its own (a "segment").  Equal runs instead of fixed depth segments: 208 columns x 64 planes over 256 CUs leave no whole
  // number of fixed segments per workgroup (the z-marching kernel's last round is a quarter full).  The cut depends on the sample's
  // shape only, never on the batch size: the operand scale is chosen per segment, so where segments end enters the rounding.
  const int chunk = (e.nwork + 7) / 8;
  const int64_t G = (int64_t)a.tiles_x * a.tiles_y * a.D;      // output (column, plane) pairs of one sample
  for (int j = blockIdx.x; j < chunk * 8; j += gridDim.x) {
    const int work = (j & 7) * chunk + (j >> 3);
    if ((j >> 3) >= chunk || work >= e.nwork) continue;
    const int b = work / X3P_PARTS, part = work % X3P_PARTS;
    const int64_t g1 = G * (part + 1) / X3P_PARTS;
    for (int64_t gq = G * part / X3P_PARTS; gq < g1;) {
    const int col = (int)(gq / a.D), zs = (int)(gq % a.D), ze = (int)std::min<int64_t>(a.D, zs + (g1 - gq));
    gq += ze - zs;
    const int x0 = (col % a.tiles_x) * XP_TX, y0 = (col / a.tiles_x) * XP_TY;
    const T* xb = x + b * a.x_bstride;
    // destinations through buffer descriptors too: a lane that must not store (outside the volume, an absent output group, a
    // plane outside the segment, a restarted ring) gets the offset X3P_OOB and the store is dropped — no divergent branches
    const __amdgpu_buffer_rsrc_t yrs = __builtin_amdgcn_make_buffer_rsrc(static_cast<T*>(a.y) + b * a.y_bstride, 0, 0x7fffffffu, 0x00020000);
    unsigned yoff[XP_NT], toff[2][XP_NT];    // byte offset of this lane's voxel at z = 0 in its first destination channel
    __amdgpu_buffer_rsrc_t trs[2] = {yrs, yrs};
    int my_tail_cout = 0, trelu = 0;
    if constexpr (TAILS) {
      // the two tails write different tensors: one descriptor each, and every tail store is issued against both with the lanes
      // of the other tail's quarter masked through the offset (which kernel a call runs on must not depend on how far apart the
      // caller's buffers happen to lie)
      my_tail_cout = kb < a.ntail ? (tsel ? a.tail_cout[1] : a.tail_cout[0]) : 0;
      trelu = tsel ? a.tail_relu[1] : a.tail_relu[0];
#pragma unroll
      for (int tq = 0; tq < 2; ++tq)
        trs[tq] = __builtin_amdgcn_make_buffer_rsrc(static_cast<T*>(a.tail_y[tq < a.ntail ? tq : 0]) + b * a.tail_bstride[tq < a.ntail ? tq : 0], 0,
                                                    tq < a.ntail ? 0x7fffffffu : 0u, 0x00020000);
    }
#pragma unroll
    for (int i = 0; i < XP_NT; ++i) {
      const int nt = wave * XP_NT + i;
      const int gy = y0 + (nt >> 1), gx = x0 + (nt & 1) * 16 + n;
      const bool inside = gy < a.H && gx < a.W;
      yoff[i] = (a.store_main && inside && g < ngroups) ? ((unsigned)my_ych * (unsigned)DHW + (unsigned)(gy * a.W + gx)) * 4u : X3P_OOB;
#pragma unroll
      for (int tq = 0; tq < 2; ++tq)
        toff[tq][i] = (TAILS && my_tail_cout > 0 && inside && tsel == tq) ? ((unsigned)a.tail_ch0[tq] * (unsigned)DHW + (unsigned)(gy * a.W + gx)) * 4u : X3P_OOB;
    }
    const float tfloor = trelu ? 0.f : -__builtin_inff();
    __syncthreads();                                   // the previous column's LDS reads are done (and the tables above are written)
    locate(y0, x0);
    if (tid == 0) *lmaxp = 0u;
    int zfirst = zs;                                    // first output plane of the (re)started ring
    for (;;) {
      // the operand scale, from the first output plane's input (largest |x| -> 2^10..2^11: 16x of headroom for the planes that
      // follow; one that still does not fit restarts the ring with a larger scale — conv3d_x3.hip)
      __syncthreads();
      prefetch(xb, zfirst, std::integral_constant<int, 0>{});
      const float wmx = x3_wave_max(local_max(std::integral_constant<int, 0>{}));
      if (lane == 0) atomicMax(lmaxp, __float_as_uint(wmx));
      __syncthreads();
      mul = x3_pow2_scale(__uint_as_float(*lmaxp), X3_ACT_TARGET);
      if (tid < 32) par[tid] = par[96 + tid] * (1.f / mul);
      const int zlo = zfirst, s0 = zfirst - 1;          // first output / input plane of this pass; plane p lives in ring slot (p - s0) % 4
      prefetch(xb, s0, std::integral_constant<int, 0>{});
      note_overflow(std::integral_constant<int, 0>{});
      commit_all(0);
      prefetch(xb, s0 + 1, std::integral_constant<int, 1>{});
      int zhi = ze;                                     // outputs below zhi may be stored; lowered when a plane overflows the scale
      int again = 0;
      // One scheduling region and ONE barrier per step.  While plane sp (ring slot PH) is multiplied — by all three dz slices, also
      // at the segment's ends: what it contributes beyond [zfirst, ze) lands in accumulators that are never stored — the SAME
      // instruction stream carries, between the MFMAs, the staging of plane sp + 1 into the free fourth ring slot and the epilogue of
      // output sp - 2, which the previous step completed: on this chip an MFMA leaves half of its issue cycles to other vector
      // instructions of the same wave, but phases of vector-only code (staging, epilogue) next to phases of MFMA-only code add up
      // (measured on this kernel: 145 us = 56 MFMA + 29 epilogue + 18 staging + 9 prefetch issue + 21 reads / barriers + ...).
      // Four ring slots / accumulator sets, unrolled four times: slots and accumulator roles are compile-time names.
      f32x4 acc[4][NSET][XP_NT];                        // output plane o accumulates in acc[(o - s0) % 4]
#pragma unroll
      for (int k4 = 0; k4 < 4; ++k4)
#pragma unroll
        for (int st = 0; st < NSET; ++st)
#pragma unroll
          for (int i = 0; i < XP_NT; ++i) acc[k4][st][i] = f32x4{0.f, 0.f, 0.f, 0.f};
      for (int sb = s0; sb <= ze + 1; sb += 4) {
        static_for<4>([&](auto ph_) {
          constexpr int PH = decltype(ph_)::value;      // ring slot of the plane multiplied in this step
          const int sp = sb + PH;
          X3P_STAMP(3);
          // LDS-only barrier: __syncthreads() also waits for every outstanding global load and store of the wave (vmcnt(0)), i.e.
          // for the prefetch issued a few hundred cycles earlier — measured: 28-35 % of the wave's time sat in that wait
          asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");   // plane sp (staged during the previous step) is visible; slot PH + 1 is free
          X3P_STAMP(0);
          constexpr int AN = (PH + 1) % 4, AC = PH, AP = (PH + 3) % 4, AE = (PH + 2) % 4;   // outputs sp + 1, sp, sp - 1 | sp - 2 (epilogue)
          // plane sp + 2 starts its journey first: into the register set plane sp left when it was staged (two planes in flight: a
          // plane has more than a step to arrive — with one set the step began by waiting for loads issued half a step earlier)
          prefetch(xb, sp + 2, std::integral_constant<int, PH & 1>{});
          // LDS operands: the plane's two pairs per (channel group, column tile), hi and lo, and the ninth tap of output z1 = sp - 1
          // (lane quarter kb reads plane z1 - 1 + kb, ring slot (PH + 2 + kb) % 4).  The MFMAs run in groups of one (channel group,
          // column tile); the operands of group g + 1 are requested between the MFMAs of group g (an LDS read issued in an MFMA's
          // shadow is nearly free; a burst of 32 of them at the head of the step cost a fifth of the step), group 0's here
          const int s8 = ((PH + 2 + (kb < 3 ? kb : 0)) % 4) * SLOT_BYTES;
          uint4 bh[NCG][XP_NT], bl[NCG][XP_NT];
          uint2 b8h[NCG][XP_NT], b8l[NCG][XP_NT];
          auto request = [&](auto g_) {
            constexpr int cgi = decltype(g_)::value / XP_NT, i = decltype(g_)::value % XP_NT;
            const char* const p0 = lbytes + va0[i] + (cgi * CG_BYTES + PH * SLOT_BYTES);
            const char* const p1 = lbytes + va1[i] + (cgi * CG_BYTES + PH * SLOT_BYTES);
            const uint2 h0 = *reinterpret_cast<const uint2*>(p0), h1 = *reinterpret_cast<const uint2*>(p1);
            const uint2 l0 = *reinterpret_cast<const uint2*>(p0 + LO_BYTES), l1 = *reinterpret_cast<const uint2*>(p1 + LO_BYTES);
            bh[cgi][i] = make_uint4(h0.x, h0.y, h1.x, h1.y);
            bl[cgi][i] = make_uint4(l0.x, l0.y, l1.x, l1.y);
            const char* const p8 = lbytes + va8[i] + s8 + cgi * CG_BYTES;
            b8h[cgi][i] = *reinterpret_cast<const uint2*>(p8);
            b8l[cgi][i] = *reinterpret_cast<const uint2*>(p8 + LO_BYTES);
          };
          request(std::integral_constant<int, 0>{});
          float4 esc[NSET], esh[NSET];                  // this lane's folded BatchNorm scale (x 2^e) and shift: four channels per set
#pragma unroll
          for (int st = 0; st < NSET; ++st) {
            esc[st] = *reinterpret_cast<const float4*>(par + st * 16 + 4 * kb);
            esh[st] = *reinterpret_cast<const float4*>(par + 32 + st * 16 + 4 * kb);
          }
          // verdict on plane sp: if it did not fit the scale, outputs from sp - 1 on are recomputed by a restarted ring
          if (__uint_as_float(*lmaxp) * mul > X3_F16_CAP && __float_as_uint(mul) > X3_SCALE_FLOOR_BITS && !again) { again = 1; zhi = min(zhi, sp - 1); zfirst = max(zs, sp - 1); }
          __builtin_amdgcn_sched_barrier(0);
          X3P_STAMP(1);
          // the vector work that rides between the MFMAs, cut into pieces of a few instructions each
          const int ze2 = sp - 2;                       // the output whose epilogue runs in this step
          const unsigned emask = (ze2 >= zlo && ze2 < zhi) ? 0u : X3P_OOB;
          const unsigned zb = (unsigned)ze2 * plane_bytes;
          unsigned ch[NPF][2], cl[NPF][2];
          float ev[XP_NT][4];
          unsigned th[XP_NT][2], tm[XP_NT][2], tl[XP_NT][2];
          f32x4 tacc[XP_NT];
          constexpr int NCP = 3 * NPF, NEP = TAILS ? 9 : 5;                  // staging pieces; epilogue pieces per column tile
          constexpr int NPIECE = NCP + 1 + XP_NT * NEP;
          auto piece = [&](auto j_) {
            constexpr int J = decltype(j_)::value;
            if constexpr (J < NCP) {                    // staging of plane sp + 1 into ring slot (PH + 1) % 4
              constexpr int p = J / 3, q = J % 3;
              constexpr int SET = (PH + 1) & 1;
              if constexpr (J == 0) note_overflow(std::integral_constant<int, SET>{});       // plane sp + 1: examined by the next step
              if constexpr (q == 0) ch[p][0] = x3_split2h(pf[SET][p][0], pf[SET][p][1], mul, cl[p][0]);
              else if constexpr (q == 1) ch[p][1] = x3_split2h(pf[SET][p][2], pf[SET][p][3], mul, cl[p][1]);
              else {
                lhi[cdst[p] + ((PH + 1) % 4) * PLS] = make_uint2(ch[p][0], ch[p][1]);
                llo[cdst[p] + ((PH + 1) % 4) * PLS] = make_uint2(cl[p][0], cl[p][1]);
              }
            } else if constexpr (J == NCP) {
            } else {
              constexpr int i = (J - NCP - 1) / NEP, q = (J - NCP - 1) % NEP;
              if constexpr (q < 4) {                    // folded BatchNorm + ReLU (+ sibling sum) of register q
                float sum = 0.f;
#pragma unroll
                for (int st = 0; st < NSET; ++st) {
                  const float scq = q == 0 ? esc[st].x : q == 1 ? esc[st].y : q == 2 ? esc[st].z : esc[st].w;
                  const float shq = q == 0 ? esh[st].x : q == 1 ? esh[st].y : q == 2 ? esh[st].z : esh[st].w;
                  const float u = fmaxf(fmaf(acc[AE][st][i][q], scq, shq), act_floor);
                  sum = st == 0 ? u : sum + u;
                }
                ev[i][q] = sum;
              } else if constexpr (q == 4) {
                unsigned off = (yoff[i] + zb) | emask;       // an X3P_OOB offset keeps its top bit: z * plane_bytes < 2^31
                asm volatile("" : "+v"(off));
#pragma unroll
                for (int r = 0; r < 4; ++r) __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(ev[i][r]), yrs, off, (unsigned)r * chan_bytes, 0);
              } else if constexpr (q == 5) {
                th[i][0] = x3_split3(ev[i][0], ev[i][1], tm[i][0], tl[i][0]);
              } else if constexpr (q == 6) {
                th[i][1] = x3_split3(ev[i][2], ev[i][3], tm[i][1], tl[i][1]);
              } else if constexpr (q == 7) {
                tacc[i] = x3_mma<true>(ltail[lane], make_uint4(th[i][0], th[i][1], tm[i][0], tm[i][1]), f32x4{0.f, 0.f, 0.f, 0.f});
                tacc[i] = x3_mma<true>(ltail[64 + lane], make_uint4(tl[i][0], tl[i][1], th[i][0], th[i][1]), tacc[i]);
                tacc[i] = x3_mma<true>(ltail[128 + lane], make_uint4(tm[i][0], tm[i][1], th[i][0], th[i][1]), tacc[i]);
              } else {
                const float4 tsc = *reinterpret_cast<const float4*>(par + 64 + 4 * kb), tsh = *reinterpret_cast<const float4*>(par + 80 + 4 * kb);
                const float sc4[4] = {tsc.x, tsc.y, tsc.z, tsc.w}, sh4[4] = {tsh.x, tsh.y, tsh.z, tsh.w};
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                  const unsigned val = __float_as_uint(fmaxf(fmaf(tacc[i][r], sc4[r], sh4[r]), tfloor));
#pragma unroll
                  for (int tq = 0; tq < 2; ++tq) {
                    unsigned off = (toff[tq][i] + zb) | emask | (r < my_tail_cout ? 0u : X3P_OOB);
                    asm volatile("" : "+v"(off));
                    __builtin_amdgcn_raw_buffer_store_b32(val, trs[tq], off, (unsigned)r * chan_bytes, 0);
                  }
                }
              }
            }
          };
          // MFMA K of the step: groups of 12 — one (channel group, column tile): the three terms of dz = 2, 1, 0 round-robin over
          // their three accumulators (a dependent pair is three instructions apart), then the three terms of the ninth-tap slice,
          // which completes output sp - 1
          constexpr int NGRP = NCG * XP_NT, NMFMA = 12 * NGRP;
          static_for<NMFMA>([&](auto k_) {
            constexpr int K = decltype(k_)::value;
            if constexpr (K == NMFMA / 2) X3P_STAMP(2);
            constexpr int grp = K / 12, r_ = K % 12, cgi = grp / XP_NT, i = grp % XP_NT, st = cgi / NCGS;
            if constexpr (r_ >= 9) {
              constexpr int term = r_ - 9;
              acc[AP][st][i] = x3_mma16(term == 2 ? w8l[cgi] : w8h[cgi], term == 1 ? b8l[cgi][i] : b8h[cgi][i], acc[AP][st][i]);
            } else {
              constexpr int term = r_ / 3, dz = 2 - r_ % 3, k4 = dz == 2 ? AP : dz == 1 ? AC : AN;
              constexpr bool open = dz == 0 && term == 0 && cgi % NCGS == 0;     // dz = 0 of the set's first channel group opens the sum
              acc[k4][st][i] = x3_mma<false>(term == 2 ? wml[cgi][dz] : wmh[cgi][dz], term == 1 ? bl[cgi][i] : bh[cgi][i],
                                             open ? f32x4{0.f, 0.f, 0.f, 0.f} : acc[k4][st][i]);
            }
            if constexpr (r_ == 1 && grp + 1 < NGRP) request(std::integral_constant<int, grp + 1>{});
            // pieces [K * NPIECE / NMFMA, (K + 1) * NPIECE / NMFMA) follow MFMA K
            static_for<((K + 1) * NPIECE) / NMFMA - (K * NPIECE) / NMFMA>([&](auto d_) { piece(std::integral_constant<int, (K * NPIECE) / NMFMA + decltype(d_)::value>{}); });
            __builtin_amdgcn_sched_barrier(0);
          });
        });
        if (again) break;
      }
      if (!again) break;
    }
    }
  }
#ifdef RAGMI_X3P_STAMPS
  if (e.dbg && lane == 0 && wave == 1)
    for (int k = 0; k < 6; ++k) atomicAdd(e.dbg + k, tsum[k]);
#endif
}

bool x3p_eligible(const K3Args& a, int nset, int dtype) {
  if (dtype != RAGMI_F32X3) return false;
#ifdef RAGMI_X3P_STAMPS     // measurement builds only (tools/x3p_time.py): RAGMI_X3P=0 sends every call to the z-marching kernel
  { static const int on = [] { const char* v = getenv("RAGMI_X3P"); return v ? atoi(v) : 1; }(); if (!on) return false; }
#endif
  const int ncgs = a.nchunks[0];
  // buffer offsets are 31-bit byte offsets from a batch item's base: the input channels, the destination channels of the main
  // output and of each tail must fit
  const int64_t chan = (int64_t)a.D * a.H * a.W * 4, lim = (1ll << 31) - 4096;
  if ((int64_t)a.Cin * chan >= lim) return false;
  for (int g = 0; g < (a.Cout + 3) / 4; ++g)
    if ((int64_t)(a.y_ch[g] + 4) * chan >= lim) return false;
  for (int tq = 0; tq < a.ntail; ++tq)
    if ((int64_t)(a.tail_ch0[tq] + 4) * chan >= lim) return false;
  if (nset == 2) return ncgs == 1 && a.nchunks[1] == 1;          // the dual level-3 cell: <2, 2>
  return ncgs >= 1 && ncgs <= 3;                                  // <1, 1>, <2, 1>, <3, 1>
}

template <int NCG, int NSET, bool TAILS>
static int x3p_launch_one(const K3Args& a, const X3Extra& e, dim3 grid, hipStream_t st) {
  constexpr size_t lds = (size_t)2 * (NCG * 4 * (XP_HY * x3_row_stride(NCG) + 2) + 2) * sizeof(uint2) + (size_t)NCG * 3 * 2 * 64 * sizeof(uint4) +
                         (size_t)NCG * 2 * 64 * sizeof(uint2) + 3 * 64 * sizeof(uint4) + 132 * sizeof(float);
  static_assert(lds <= 160 * 1024, "plane-stationary tile does not fit the LDS");
  static LaunchState state;     // per device, mutex-guarded (common.h)
  const int slots = state.slots((const void*)conv3d_x3p_kernel<NCG, NSET, TAILS>, XP_THREADS, lds, 160 * 1024);
  if (slots <= 0) return fail(RAGMI_ELAUNCH, "conv3d_x3p: cannot raise the dynamic LDS limit");
  grid.x = (unsigned)std::max<int64_t>(1, std::min<int64_t>(grid.x, std::max(256, slots) / (int)grid.y));
  hipLaunchKernelGGL((conv3d_x3p_kernel<NCG, NSET, TAILS>), grid, dim3(XP_THREADS), lds, st, a, e);
  return check_launch("conv3d_x3p");
}

// a, e: as prepared by x3_launch (tiles, segments, weight sections)
int x3p_launch(K3Args a, X3Extra e, int nset, hipStream_t st) {
  const int ncg = a.nchunks[0] * nset;
  // work items: X3P_PARTS equal runs of output planes per sample (see the kernel)
  a.tiles_x = (int)ceil_div(a.W, XP_TX); a.tiles_y = (int)ceil_div(a.H, XP_TY);
  e.seg_len = a.D; e.nseg = 1;
  e.nwork = X3P_PARTS * a.B;
  e.xbytes = (unsigned)((int64_t)a.Cin * a.D * a.H * a.W * 4);
#ifdef RAGMI_X3P_STAMPS
  { const char* v = getenv("RAGMI_X3P_DBG"); e.dbg = v ? reinterpret_cast<unsigned long long*>(strtoull(v, nullptr, 16)) : nullptr; }
#endif
  const dim3 grid((unsigned)std::min<int64_t>(e.nwork, 1 << 20), (a.Cout + 15) / 16);
#define RAGMI_X3P(NCG_, NSET_) (a.ntail > 0 ? x3p_launch_one<NCG_, NSET_, true>(a, e, grid, st) : x3p_launch_one<NCG_, NSET_, false>(a, e, grid, st))
  if (nset == 2) return RAGMI_X3P(2, 2);
  switch (ncg) {
    case 1: return RAGMI_X3P(1, 1);
    case 2: return RAGMI_X3P(2, 1);
    default: return RAGMI_X3P(3, 1);
  }
#undef RAGMI_X3P
}

}  // namespace ragmi
