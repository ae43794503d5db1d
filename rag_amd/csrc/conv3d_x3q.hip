// Level-3 dual cells (Cell_3d sibling pair 4 + 4 -> 12 channels, rag_model.py:134-137,160-172 via operations_3d.py:31-47) on the split-operand
// form of conv3d_x3.hip, restructured (round 5) so that the plane step issues fewer instructions and meets ONE barrier:
//
//  * Ring of FOUR halo planes in LDS (conv3d_x3_kernel: three).  Plane z+2 is committed during step z into the slot that plane z-2
//    left at the previous step, one whole step ahead of its first use: a step needs a single barrier (all reads of step z-1 done /
//    all commits of plane z+1 visible) instead of two.  Rows of 40 records (conv3d_x3_kernel: 49) make the four slots cheaper
//    than its three: 51.2 KB.
//  * Every operand address is a per-lane constant of the launch plus an IMMEDIATE.  The 27 taps of a channel group are dealt into
//    K-slices so that a slice reads ONE plane (dz): three "plane slices" of 7 taps + 1 padding slot, and one "leftover slice" holding
//    the two remaining taps of each of the three planes.  The z loop is unrolled over the ring phase (z & 3), so the slot of a plane
//    slice is a compile-time immediate; in the leftover slice lane quarter q reads plane z-1+q, whose slot advances by one per step
//    (one add, one and, two multiply-adds per step).  conv3d_x3_kernel read an offset table (a ds_read per slice) and added a table
//    value to a tile base for every operand: 48 vector instructions and 8 LDS reads per step, on the critical path of every slice.
//  * Lane quarters that share an LDS pass (0,1) / (2,3) read taps (dy = 0, dx) and (dy = 2, dx) — 80 records = 128 bytes mod 256
//    apart: the two halves of the 64 banks — or the SAME address (padding slots: zero weights against a voxel inside the 3x3x3
//    window, include/rag_amd.h's non-finite contract); planes are 400 records = 128 mod 256 apart too (leftover slice).  No pass
//    is conflicted.
// Arithmetic, operand scaling, restart on overflow, fused tails and down-sampling tails are those of conv3d_x3_kernel (same
// helpers); only the order of the K sum differs (plane slices first).  fp32 storage only; bf16 storage stays on conv3d_x3_kernel.
#include "conv3d_x3_common.h"

namespace ragmi {

// (2 records behind every channel group and behind the hi copy: with strides that are multiples of 512 bytes hipcc fuses the reads of
// the two sets / of a hi and its lo operand into ds_read2st64_b64 — half rate, and its result registers are not an MFMA operand)
constexpr int XQ_RS = 40, XQ_PLS = X3_HY * XQ_RS, XQ_SLOTS = 4, XQ_CGS = XQ_SLOTS * XQ_PLS + 2, XQ_LOPAD = 2;      // records of 8 bytes
constexpr int XQ_DU_SLOTS = 3, XQ_DU_PLANE = 2 * 4 * X3_TY * (X3_TX / 2);      // down-sampling tails: x-blended tail values, floats per plane copy
static_assert(XQ_RS >= X3_HX && (2 * XQ_RS) % 32 == 16 && XQ_PLS % 32 == 16, "bank layout: partners of an LDS pass 128 bytes mod 256 apart");
static_assert((XQ_CGS * 8) % 512 != 0 && (XQ_PLS * 8) % 512 != 0 && XQ_PLS > 255, "strides that ds_read2(st64)_b64 cannot span");
static_assert(2 * (2 * XQ_CGS + XQ_LOPAD) * 8 + 8 * 2 * 64 * 16 + 4 * 64 * 4 + 132 * 4 + XQ_DU_SLOTS * XQ_DU_PLANE * 4 + 20 * 16 <= 80 * 1024,
              "the dual launch with down-sampling tails must leave room for two workgroups per CU");
// tap (dy * 3 + dx) that lane quarter q holds as operand j of a plane slice; the padding slot (q = 3, j = 1) re-reads its pass
// partner's voxel.  Left over per plane: taps 4 = (1,1) and 5 = (1,2).
__host__ __device__ constexpr int xq_tap7(int q, int j) {
  constexpr int t[4][2] = {{0, 2}, {6, 8}, {1, 3}, {7, 3}};
  return t[q][j];
}
__host__ __device__ constexpr bool xq_pad7(int q, int j) { return q == 3 && j == 1; }

#ifdef RAGMI_DIAG
// profiling builds: in-kernel stamps, as conv3d_x3.hip's (a buffer of its own per translation unit: no relocatable device code)
__device__ unsigned long long* xq_stamp_buf = nullptr;
extern "C" __attribute__((visibility("default"))) int ragmi_diag_x3q_stamp_buffer(void* buf) {
  unsigned long long* p = static_cast<unsigned long long*>(buf);
  return hipMemcpyToSymbol(HIP_SYMBOL(xq_stamp_buf), &p, sizeof(p)) == hipSuccess ? 0 : -1;
}
#endif

// NSET accumulator sets of ONE 4-channel group each; TAILS as conv3d_x3_kernel.
// G4X / G4T: the input / the destinations of the full-resolution tails are CHANNEL-GROUP-INTERLEAVED tensors [B][C/4][D][H][W][4]
// ("G4", include/rag_amd.h) instead of channel planes [B][C][D][H][W].  The level-3 tensors this kernel exchanges with its
// neighbours are private to the fused executor (MatchingNet._run_chain): in G4 a lane's four channels of a voxel are ONE 16-byte
// access — a halo voxel is one buffer_load_dwordx4 per group instead of four dword loads, a tail one global_store_dwordx4 instead of
// four dword stores into four planes (64-byte segments).  The vector memory path costs per INSTRUCTION here (the plane step issued
// ~190 of them per workgroup, ~20 cycles each with two workgroups per CU: MI355X_MICROARCH.md, store tail; profiles/r05_x3_stamps.md).
typedef unsigned xq_u32x4 __attribute__((ext_vector_type(4)));
template <int NSET, int TAILS, bool G4X, bool G4T>
__global__ __launch_bounds__(X3_THREADS, 4) void conv3d_x3q_kernel(K3Args a, X3Extra e) {
  using T = float;
  constexpr int NCG = NSET, NSL = 4 * NSET;
  static_assert(X3_PL <= X3_THREADS, "one halo voxel per thread");
  extern __shared__ __attribute__((aligned(16))) uint2 xq_lds[];       // hi[NCG][4 slots][PLS] | lo[...] | weights | tail weights | params | down staging
  uint2* const lhi = xq_lds;
  uint2* const llo = xq_lds + NCG * XQ_CGS + XQ_LOPAD;
  uint4* const lw = reinterpret_cast<uint4*>(xq_lds + 2 * (NCG * XQ_CGS + XQ_LOPAD));           // [slice][hi/lo][64 lanes]
  float* const ltail = reinterpret_cast<float*>(lw + NSL * 2 * 64);               // fused-tail A fragments [4 products][64 lanes] (fp32)
  // scale[2][16] (times the column's 2^e, rewritten per ring pass) | shift[2][16] | tail scale[4 kb][4] | tail shift[4][4] |
  // static scale[2][16] (BatchNorm scale x the weights' 2^-k) | lmaxp[3]: the maximum the scale is chosen from, two words of overflow notes
  float* const par = ltail + 4 * 64;
  unsigned* const lmaxp = reinterpret_cast<unsigned*>(par + 128);
  float* const ldu = par + 132;
  float4* const ldxt = reinterpret_cast<float4*>(ldu + XQ_DU_SLOTS * XQ_DU_PLANE);
  float4* const ldyt = ldxt + X3_TX / 2;
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, n = lane & 15, kb = lane >> 4;
  const int cog = blockIdx.y;
  const int HW = a.H * a.W;
  float act_floor = (a.relu & 1) ? 0.f : __builtin_nanf("");   // max(u, NaN) = u: the identity, NaN inputs included
  asm volatile("" : "+v"(act_floor));
  const int64_t DHW = (int64_t)HW * a.D;
#ifdef RAGMI_DIAG
  // profiling builds (RAGMI_X3_DIAG bits, as conv3d_x3_kernel): 1 no stores, 2 no MFMA block, 4 no commit, 8 no loads, 16 operand reads at one address
  const bool dg_nostore = (a.relu & 0x100) != 0, dg_nomfma = (a.relu & 0x200) != 0, dg_nocommit = (a.relu & 0x400) != 0, dg_noload = (a.relu & 0x800) != 0, dg_noread = (a.relu & 0x1000) != 0;
  const bool dg_nofinish = (a.relu & 0x4000) != 0, dg_nopark = (a.relu & 0x8000) != 0;      // 64: no finishing step of the down-sampling tails, 128: no x blend + LDS parking
  const bool dg_stamp = (a.relu & 0x2000) != 0 && xq_stamp_buf != nullptr;
  unsigned long long dg_sum[8] = {0, 0, 0, 0, 0, 0, 0, 0}, dg_last = 0, dg_t0 = 0, dg_r0 = 0;
  unsigned dg_steps = 0;
  if (dg_stamp) { dg_t0 = dg_last = __builtin_amdgcn_s_memtime(); dg_r0 = __builtin_amdgcn_s_memrealtime(); }
#define XQ_STAMP(k) do { if (dg_stamp) { __builtin_amdgcn_sched_barrier(0); const unsigned long long t_ = __builtin_amdgcn_s_memtime(); \
                                         dg_sum[k] += t_ - dg_last; dg_last = t_; __builtin_amdgcn_sched_barrier(0); } } while (0)
#else
  constexpr bool dg_nostore = false, dg_nomfma = false, dg_nocommit = false, dg_noload = false, dg_noread = false, dg_nofinish = false, dg_nopark = false;
#define XQ_STAMP(k) do { } while (0)
#endif
  // weight fragments, 8 bytes (one pair's four channels) at a time, from the packed tap-major slices (x3_pack_one: pair P = tap, one
  // channel group per set) into this kernel's slices: [set][plane 0..2 | leftover]
  {
    // (all loads of the staging first, then the LDS stores: written as load -> store per element the loop waits out one memory round
    // trip per iteration; compile-time `set`: e.wf[set] stays a scalar — see the parameters below)
    constexpr int PER_SET = 4 * 2 * 64 * 2, ITER = PER_SET / X3_THREADS;
    static_assert(PER_SET % X3_THREADS == 0, "weight staging: whole rounds");
    uint2 wv[NSET][ITER];
#pragma unroll
    for (int set = 0; set < NSET; ++set) {
      const uint2* const src = reinterpret_cast<const uint2*>(e.wf[set] + (int64_t)cog * 4 * 2 * 64);
#pragma unroll
      for (int it = 0; it < ITER; ++it) {
        const int i = it * X3_THREADS + tid;
        const int j = i & 1, ln = (i >> 1) & 63, sh = i >> 7, hl = sh & 1, sl = sh >> 1, q = ln >> 4;
        int tap = -1;
        if (sl < 3) { if (!xq_pad7(q, j)) tap = sl * 9 + xq_tap7(q, j); }
        else if (q < 3) tap = q * 9 + 4 + j;
        const int ps = max(tap, 0) >> 3, pp = max(tap, 0) & 7;
        const uint2 v = src[(((ps * 2 + hl) * 64) + (pp >> 1) * 16 + (ln & 15)) * 2 + (pp & 1)];      // (unconditional: a padding slot reads tap 0 and drops it)
        wv[set][it] = tap >= 0 ? v : make_uint2(0u, 0u);
      }
    }
#pragma unroll
    for (int set = 0; set < NSET; ++set)
#pragma unroll
      for (int it = 0; it < ITER; ++it) reinterpret_cast<uint2*>(lw)[set * PER_SET + it * X3_THREADS + tid] = wv[set][it];
  }
  // Parameters -> LDS.  Every descriptor array of the kernel arguments is indexed by a COMPILE-TIME index inside an unrolled loop and
  // the lanes pick by comparison: indexed by a lane-dependent value (set = i >> 4, tail = row >> 2) the compiler fetches the pointer
  // itself with a vector load from the argument segment and the value with a second, dependent one — ~25 serial memory round trips
  // in front of the first plane of every level-3 launch (round 5: found in the listing of the finishing step, below).
  if (tid < 32) {
    const int co = cog * 16 + (tid & 15);
    float sc = 1.f, sh = 0.f;
#pragma unroll
    for (int set = 0; set < NSET; ++set) {
      const float* const psc = a.scale[set];
      const float* const psh = a.shift[set];
      const float* const pwm = e.wmul[set];
      if ((tid >> 4) == set && co < a.Cout) {
        sc = (psc ? psc[co] : 1.f) * pwm[co];          // BatchNorm scale x 2^-k (undoes the per-channel weight scale)
        sh = psh ? psh[co] : 0.f;
      }
    }
    par[tid] = sc;
    par[96 + tid] = sc;
    par[32 + tid] = sh;
  }
  if (tid < 3) lmaxp[tid] = 0u;
  if constexpr (TAILS) {
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
      for (int j = 0; j < 4; ++j) ltail[j * 64 + lane] = wv[j];
    }
    if (tid < 16) {
      const int tk = tid >> 2, r = tid & 3;      // tail slot tk (full-resolution tails first, then the down-sampling ones), output r
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
  // Halo staging: thread t < 340 owns voxel t of the 10 x 34 halo plane for EVERY channel group (the groups are a wave-uniform
  // distance apart: one lane offset, one validity bit, one LDS record address); the threads past 340 stage nothing.
  const T* const x = static_cast<const T*>(a.x);
  float pf[NCG][4];
  bool valid = false, vmask = false;
  const bool stager = tid < X3_PL;
  unsigned voff = 0;                     // byte offset of the thread's voxel inside a channel plane
  auto locate = [&](int y0, int x0) {
    const int r = min(tid, X3_PL - 1), xx = r % X3_HX, yy = r / X3_HX;
    const int gy = y0 - 1 + yy, gx = x0 - 1 + xx;
    vmask = (unsigned)gy < (unsigned)a.H && (unsigned)gx < (unsigned)a.W;
    voff = (unsigned)(min(max(gy, 0), a.H - 1) * a.W + min(max(gx, 0), a.W - 1)) * (unsigned)sizeof(T) * (G4X ? 4u : 1u);
  };
  // halo loads and main stores through raw buffer descriptors: address = descriptor base (the sample) + a wave-uniform scalar offset
  // (plane, channel) + this lane's 32-bit byte offset — no vector address arithmetic in the plane step, 32-bit lane offsets
  // (xq_takes: Cin and Cout * D * H * W * 4 bytes < 2^31)
  __amdgpu_buffer_rsrc_t xrs = __builtin_amdgcn_make_buffer_rsrc(const_cast<T*>(x), 0, -1, 0x00020000);
  __amdgpu_buffer_rsrc_t yrs = __builtin_amdgcn_make_buffer_rsrc(static_cast<T*>(a.y), 0, -1, 0x00020000);
  auto prefetch_to = [&](float (&q)[NCG][4], bool& vld, int gz) {
    vld = vmask && (unsigned)gz < (unsigned)a.D;
    const int pb = min(max(gz, 0), a.D - 1) * HW;
    if (stager && !dg_noload) {
#pragma unroll
      for (int cg = 0; cg < NCG; ++cg) {
        if constexpr (G4X) {
          const xq_u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(xrs, (int)voff, (pb + cg * (int)DHW) * 4 * (int)sizeof(T), 0);
#pragma unroll
          for (int c = 0; c < 4; ++c) q[cg][c] = __uint_as_float(v[c]);
        } else {
#pragma unroll
          for (int c = 0; c < 4; ++c)
            q[cg][c] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(xrs, (int)voff, (pb + (4 * cg + c) * (int)DHW) * (int)sizeof(T), 0));
        }
      }
    }
  };
  auto prefetch = [&](int gz) { prefetch_to(pf, valid, gz); };
  float mul = 1.f;                       // the column's operand scale 2^-e (wave-uniform)
  unsigned cap_bits = 0x7f7fffffu;       // bit pattern of X3_F16_CAP / mul
  // this thread's record of a halo plane (slot 0, channel group 0), located once
  const int cdst = (min(tid, X3_PL - 1) / X3_HX) * XQ_RS + min(tid, X3_PL - 1) % X3_HX;
  auto commit_from = [&](const float (&q)[NCG][4], bool vld, int slot) {          // registers -> ring slot (fp16 hi / lo halves of x * mul), zeros outside the volume
    if (stager && !dg_nocommit) {
#pragma unroll
      for (int cg = 0; cg < NCG; ++cg) {
        float v[4];
#pragma unroll
        for (int c = 0; c < 4; ++c) v[c] = vld ? q[cg][c] : 0.f;
        unsigned l01, l23;
        const unsigned h01 = x3_split2h(v[0], v[1], mul, l01), h23 = x3_split2h(v[2], v[3], mul, l23);
        const int d = cdst + cg * XQ_CGS + slot * XQ_PLS;
        lhi[d] = make_uint2(h01, h23);
        llo[d] = make_uint2(l01, l23);
      }
    }
  };
  auto commit = [&](int slot) { commit_from(pf, valid, slot); };
  auto local_max_of = [&](const float (&q)[NCG][4], bool vld) {
    float m = 0.f;
#pragma unroll
    for (int cg = 0; cg < NCG; ++cg) m = fmaxf(m, x3_scalable_max4(q[cg][0], q[cg][1], q[cg][2], q[cg][3]));
    return (stager && vld) ? m : 0.f;
  };
  // Notes go to one of TWO words, by the parity of the step that writes them: step z writes word 1 + (z & 1) and reads word
  // 1 + ((z - 1) & 1), which nobody writes between the barriers of steps z and z + 1 — with ONE barrier per step a single word would be
  // written (this step's notes) and read (this step's restart test) in the same interval, and the test would not be workgroup-uniform.
  // The words are only cleared at a ring start, between its two barriers.
  auto note_of = [&](const float (&q)[NCG][4], bool vld, int word) {
    unsigned mb = 0u;
#pragma unroll
    for (int cg = 0; cg < NCG; ++cg)
      mb = max(mb, max(max(__float_as_uint(q[cg][0]) & 0x7fffffffu, __float_as_uint(q[cg][1]) & 0x7fffffffu),
                       max(__float_as_uint(q[cg][2]) & 0x7fffffffu, __float_as_uint(q[cg][3]) & 0x7fffffffu)));
    if (stager && vld && mb > cap_bits) {
      const float m = local_max_of(q, vld);
      if (m * mul > X3_F16_CAP) atomicMax(lmaxp + 1 + word, __float_as_uint(m));
    }
  };
  auto note_overflow = [&](int word) { note_of(pf, valid, word); };
  // operand addresses (bytes): wave w owns row y = w of the tile, column tiles x = 0..15 and 16..31.  Plane slices: ad[tile][j] + an
  // immediate (set, slot[, lo copy]); leftover slice: a3[tile][j], rebuilt per step from b3 (slot 0) and the lane's plane
  static_assert(X3_NT == 2, "two column tiles per wave");
  int ad[X3_NT][2], b3[X3_NT][2];
#pragma unroll
  for (int i = 0; i < X3_NT; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int t7 = xq_tap7(kb, j);
      ad[i][j] = ((wave + t7 / 3) * XQ_RS + n + 16 * i + t7 % 3) * (int)sizeof(uint2);
      b3[i][j] = ((wave + 1) * XQ_RS + n + 16 * i + 1 + j) * (int)sizeof(uint2);
      asm volatile("" : "+v"(ad[i][j]));     // opaque: at a visible constant distance the compiler fuses two reads into one ds_read2_b64
      asm volatile("" : "+v"(b3[i][j]));     // (half rate, and its result registers are not the MFMA operand's)
    }
  const int qq3 = min(kb, 2) + 3;            // leftover slice: this lane quarter reads plane z - 1 + min(kb, 2) (quarter 3: padding, its partner's voxel)
  const char* const lbytes = reinterpret_cast<const char*>(xq_lds);
  constexpr int LO_BYTES = (NCG * XQ_CGS + XQ_LOPAD) * (int)sizeof(uint2);
  static_assert(LO_BYTES % 512 != 0, "hi / lo copies a ds_read2st64_b64 apart");
  const int g = cog * 4 + kb, ngroups = (a.Cout + 3) >> 2;
  const int my_ych = g < ngroups ? a.y_ch[g < RAGMI_MAX_GROUPS ? g : 0] : 0;
  const int tsel = kb & 1;
  int b = 0, y0 = 0, x0 = 0;
  auto down_finish = [&](int zodd) {
    if constexpr (TAILS == 2) {
      if (dg_nofinish) return;
      const int Z = zodd >> 1, Do = a.D >> 1, Ho = a.H >> 1, Wo = a.W >> 1;
      const LinIdx lz = lin_index(min(Z, Do - 1), a.D, Do, e.dsd, 1);            // wave-uniform
      const float wz0 = lz.i0 == 2 * Z ? lz.w0 : 0.f, wz1 = lz.i0 == 2 * Z ? lz.w1 : 1.f;
      const int64_t ovol = (int64_t)Do * Ho * Wo;
      // (the down slot is wave-uniform — waves 0..3 finish slot 0, waves 4..7 slot 1 — and SAID to be: picked by a lane-dependent
      // index the descriptors below were three serial vector loads from the argument segment per finishing step, each behind an
      // s_waitcnt vmcnt(0) that also drained the halo prefetch: 196 -> 150 us for cell 1's launch with this step switched off)
      const int o = tid >> 2, r = tid & 3;
      const int dl = __builtin_amdgcn_readfirstlane(tid >> 8);
      if (dl < a.ndown) {
        const int yp = (o >> 4) & 3, xp = o & 15;
        const float4 yt = ldyt[yp];
        // (THREE plane copies, plane z in copy z % 3: this step — after ONE barrier — reads the copies of planes zodd-1 and zodd while
        // the epilogues of this step already write plane zodd+1 into the third; with two copies a second barrier stood here)
        const float* const u0 = ldu + ((zodd - 1) % 3) * XQ_DU_PLANE + ((dl * X3_TY + 2 * yp) * (X3_TX / 2) + xp) * 4 + r;
        const float* const u1 = ldu + (zodd % 3) * XQ_DU_PLANE + ((dl * X3_TY + 2 * yp) * (X3_TX / 2) + xp) * 4 + r;
        const float e0 = u0[0], e1 = u0[(X3_TX / 2) * 4], o0 = u1[0], o1 = u1[(X3_TX / 2) * 4];
        const int slot = a.ntail + dl;
        const float sc = par[64 + 4 * slot + r], sh = par[80 + 4 * slot + r];
        const int dcout = dl ? a.down_cout[1] : a.down_cout[0], drelu = dl ? a.down_relu[1] : a.down_relu[0];
        const int Y = (y0 >> 1) + yp, X = (x0 >> 1) + xp;
        if (Y < Ho && X < Wo && Z < Do && r < dcout) {
          T* const dy = static_cast<T*>(dl ? a.down_y[1] : a.down_y[0]) + b * (dl ? a.down_bstride[1] : a.down_bstride[0]) +
                        (int64_t)((dl ? a.down_ch0[1] : a.down_ch0[0]) + r) * ovol + ((int64_t)Z * Ho + Y) * Wo + X;
          const bool yclamp = yt.z != 0.f, zclamp = lz.i0 != 2 * Z;      // clamped pairs read the odd source twice
          const float ye = lerp2(yt.x, yclamp ? e1 : e0, yt.y, e1), yo = lerp2(yt.x, yclamp ? o1 : o0, yt.y, o1);
          const float uu = fmaf(lerp2(wz0, zclamp ? yo : ye, wz1, yo), sc, sh);
          st(dy, drelu ? fmaxf(uu, 0.f) : uu);
        }
      }
    }
  };
  int zs = 0;
  // destinations, per item: the main output as a 32-bit lane offset against a wave-uniform base (Cout * D * H * W < 2^31 elements is the
  // library's rule); the lane quarter's tail as a running 64-bit pointer (two tails may live in different tensors)
  unsigned yoff[X3_NT];
  bool inside[X3_NT];
  T* tptr = nullptr;                      // this lane quarter's tail, channel 0, tile 0's voxel of plane z (advanced by a plane per step)
  int my_tail_cout = 0, my_trelu = 0;
  if constexpr (TAILS) {
    my_tail_cout = kb < a.ntail ? (tsel ? a.tail_cout[1] : a.tail_cout[0]) : 0;
    my_trelu = tsel ? a.tail_relu[1] : a.tail_relu[0];
  }
  f32x4 acc[NSET][X3_NT];
  // epilogue of plane z: lane holds channels 4 g + reg (g = cog*4 + kb) of voxel n of each column tile
  auto epilogue = [&](int z) {
    const int ybase = z * HW * (int)sizeof(T);      // wave-uniform
    // BatchNorm + ReLU + sum of the sets for BOTH column tiles first, then the two tiles' tail products INTERLEAVED (product r of tile
    // 0, of tile 1, r + 1 ...: a dependent chain of four fp32 MFMAs per tile, 40 cycles of latency a link — issued tile after tile
    // the second chain waited out the first), the stores last.
    float v[X3_NT][4];
#pragma unroll
    for (int i = 0; i < X3_NT; ++i)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        float sum = 0.f;
#pragma unroll
        for (int st = 0; st < NSET; ++st) {
          const float u = fmaxf(fmaf(acc[st][i][r], par[st * 16 + 4 * kb + r], par[32 + st * 16 + 4 * kb + r]), act_floor);
          sum = st == 0 ? u : sum + u;
        }
        v[i][r] = sum;
      }
    f32x4 tacc[X3_NT];
    if constexpr (TAILS) {
#pragma unroll
      for (int i = 0; i < X3_NT; ++i) tacc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float wt = ltail[r * 64 + lane];
#pragma unroll
        for (int i = 0; i < X3_NT; ++i) tacc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(wt, v[i][r], tacc[i], 0, 0, 0);
      }
    }
#pragma unroll
    for (int i = 0; i < X3_NT; ++i) {
      if (a.store_main && inside[i] && g < ngroups && !(dg_nostore && v[i][0] != 12345.f)) {
#pragma unroll
        for (int r = 0; r < 4; ++r)
          __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v[i][r]), yrs, (int)yoff[i], ybase + r * (int)DHW * (int)sizeof(T), 0);
      }
    }
    if constexpr (TAILS) {
#pragma unroll
      for (int i = 0; i < X3_NT; ++i) {
        const int nt = wave * X3_NT + i;
        if (my_tail_cout > 0 && inside[i] && !(dg_nostore && tacc[i][0] != 12345.f)) {
          const float4 tsc = *reinterpret_cast<const float4*>(par + 64 + 4 * kb), tsh = *reinterpret_cast<const float4*>(par + 80 + 4 * kb);
          const float sc4[4] = {tsc.x, tsc.y, tsc.z, tsc.w}, sh4[4] = {tsh.x, tsh.y, tsh.z, tsh.w};
          float u4[4];
#pragma unroll
          for (int r = 0; r < 4; ++r) {      // (xq_takes: every fused tail has exactly four output channels)
            const float u = fmaf(tacc[i][r], sc4[r], sh4[r]);
            u4[r] = my_trelu ? fmaxf(u, 0.f) : u;
          }
          if constexpr (G4T) {
            *reinterpret_cast<float4*>(tptr + 64 * i) = make_float4(u4[0], u4[1], u4[2], u4[3]);
          } else {
            T* const pt = tptr + 16 * i;
#pragma unroll
            for (int r = 0; r < 4; ++r) pt[r * DHW] = u4[r];
          }
        }
        if constexpr (TAILS == 2) {
          const int dl = kb - a.ntail;
          if (dl >= 0 && dl < a.ndown && !dg_nopark) {
            const float4 xt = ldxt[8 * (nt & 1) + (n >> 1)];
            float ux[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              const float p1 = __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(tacc[i][r]), 0x101, 0xF, 0xF, false));   // row_shl:1
              ux[r] = lerp2(xt.x, xt.z != 0.f ? p1 : tacc[i][r], xt.y, p1);
            }
            if (!(n & 1))
              reinterpret_cast<float4*>(ldu + (z % 3) * XQ_DU_PLANE)[(dl * X3_TY + (nt >> 1)) * (X3_TX / 2) + 8 * (nt & 1) + (n >> 1)] =
                  make_float4(ux[0], ux[1], ux[2], ux[3]);
          }
        }
      }
    }
    if constexpr (TAILS) tptr += G4T ? 4 * HW : HW;
  };
  // One plane step, ring phase PH = z & 3 at compile time.  Returns true when the ring has to restart at this plane (a plane
  // committed since the last check does not fit the column's operand scale): workgroup-uniform.
  auto step = [&](auto ph_, int z) -> bool {
    constexpr int PH = decltype(ph_)::value;
    XQ_STAMP(6);
    __syncthreads();                     // step z-1's operand reads are done (slot of plane z-2 is free); plane z+1 is in the LDS
    XQ_STAMP(0);
    if constexpr (TAILS == 2) {
      if (z > zs && !(z & 1)) down_finish(z - 1);      // planes z-2, z-1 of the x-blended tail values are complete (segments start even)
    }
    // (workgroup-uniform: the word of step z-1 holds the notes of the commits BEFORE the barrier above and is not written during this
    // step.  No restart once the scale sits at its floor: an Inf — or an operand above ~2^110 — can never be made to fit; such inputs
    // give non-finite outputs, include/rag_amd.h)
    if (lmaxp[1 + ((PH + 1) & 1)] != 0u && __float_as_uint(mul) > X3_SCALE_FLOOR_BITS) return true;
#ifdef XQ_COMMIT_FIRST
    note_overflow(PH & 1);
    commit((PH + 2) & 3);                // plane z+2, first read at step z+1
    XQ_STAMP(1);
    prefetch(z + 3);                 // unconditional (addresses clamped): straight-line code ahead of the MFMA block
    __builtin_amdgcn_sched_barrier(0);
#endif
    XQ_STAMP(3);
#pragma unroll
    for (int st = 0; st < NSET; ++st)
#pragma unroll
      for (int i = 0; i < X3_NT; ++i) acc[st][i] = f32x4{0.f, 0.f, 0.f, 0.f};
    // leftover slice: slot of plane z - 1 + min(kb, 2) = (PH + 3 + min(kb, 2)) & 3
    const int lslot = ((qq3 + PH) & 3) * (XQ_PLS * (int)sizeof(uint2));
    if (!dg_nomfma)
#pragma unroll
    for (int s = 0; s < NSL; ++s) {
      const int st = s >> 2, sl = s & 3;             // compile time after unrolling
      const uint4 ah = lw[(s * 2 + 0) * 64 + lane];
      const uint4 al = lw[(s * 2 + 1) * 64 + lane];
      const int imm = (st * XQ_CGS + (sl < 3 ? ((PH + 3 + sl) & 3) * XQ_PLS : 0)) * (int)sizeof(uint2);
      uint4 bh[X3_NT], bl[X3_NT];
#pragma unroll
      for (int i = 0; i < X3_NT; ++i) {
        const char* const a0 = dg_noread ? lbytes : lbytes + (sl < 3 ? ad[i][0] : b3[i][0] + lslot) + imm;
        const char* const a1 = dg_noread ? lbytes : lbytes + (sl < 3 ? ad[i][1] : b3[i][1] + lslot) + imm;
        const uint2 h0 = *reinterpret_cast<const uint2*>(a0), h1 = *reinterpret_cast<const uint2*>(a1);
        const uint2 l0 = *reinterpret_cast<const uint2*>(a0 + LO_BYTES), l1 = *reinterpret_cast<const uint2*>(a1 + LO_BYTES);
        bh[i] = make_uint4(h0.x, h0.y, h1.x, h1.y);
        bl[i] = make_uint4(l0.x, l0.y, l1.x, l1.y);
      }
#pragma unroll
      for (int i = 0; i < X3_NT; ++i) acc[st][i] = x3_mma<false>(ah, bh[i], acc[st][i]);
#pragma unroll
      for (int i = 0; i < X3_NT; ++i) acc[st][i] = x3_mma<false>(ah, bl[i], acc[st][i]);
#pragma unroll
      for (int i = 0; i < X3_NT; ++i) acc[st][i] = x3_mma<false>(al, bh[i], acc[st][i]);
    }
    XQ_STAMP(4);
#ifndef XQ_COMMIT_FIRST
    // Plane z+2 (in the registers since step z-1) goes to the ring AFTER the matrix block, not in front of it: vmcnt counts loads and
    // stores together and in order, and across the loop's back edge the compiler can only wait for vmcnt(0) — in front of the matrix
    // block that wait also sat out the acknowledgement of the stores the previous epilogue had issued a few hundred cycles earlier,
    // on the critical path of the slowest wave behind every barrier.  Here loads and stores are a matrix block old.
    __builtin_amdgcn_sched_barrier(0);
    note_overflow(PH & 1);
    commit((PH + 2) & 3);                // plane z+2, first read at step z+1 (its slot was plane z-2's: free since the barrier above)
    XQ_STAMP(1);
    prefetch(z + 3);                     // unconditional (addresses clamped); consumed a whole step later
    __builtin_amdgcn_sched_barrier(0);
#endif
    epilogue(z);
    XQ_STAMP(5);
#ifdef RAGMI_DIAG
    ++dg_steps;
#endif
    return false;
  };
#ifdef XQ_PHASE_OFFSET
  // experiment: the two workgroups of a CU (dispatched a grid-half apart) start half a plane step apart
  if (blockIdx.x >= gridDim.x / 2) __builtin_amdgcn_s_sleep(XQ_PHASE_OFFSET);
#endif
  const int chunk = (e.nwork + 7) / 8;
  for (int j = blockIdx.x; j < chunk * 8; j += gridDim.x) {
    const int work = (j & 7) * chunk + (j >> 3);
    if ((j >> 3) >= chunk || work >= e.nwork) continue;
    // virtual item -> (sample, column, depth segment[, half]): conv3d_x3_kernel's schedule (x3_launch)
    int t = work, half = -1;
    b = t / (e.ngrp * (e.grp + e.nsplit));
    t %= e.ngrp * (e.grp + e.nsplit);
    const int gi = t / (e.grp + e.nsplit), k = t % (e.grp + e.nsplit);
    if (k < e.grp - e.nsplit) t = gi * e.grp + k;
    else { t = gi * e.grp + (e.grp - e.nsplit) + ((k - (e.grp - e.nsplit)) >> 1); half = (k - (e.grp - e.nsplit)) & 1; }
    x0 = (t % a.tiles_x) * X3_TX; t /= a.tiles_x;
    y0 = (t % a.tiles_y) * X3_TY; t /= a.tiles_y;
    const int seg = t;
    zs = seg * e.seg_len;
    int ze = min(a.D, zs + e.seg_len);
    if (half >= 0) { const int mid = zs + ((ze - zs + 1) >> 1); if (half) zs = mid; else ze = mid; }
    xrs = __builtin_amdgcn_make_buffer_rsrc(const_cast<T*>(x + b * a.x_bstride), 0, -1, 0x00020000);
    yrs = __builtin_amdgcn_make_buffer_rsrc(static_cast<T*>(a.y) + b * a.y_bstride, 0, -1, 0x00020000);
    __syncthreads();                                   // the previous column's LDS reads are done (and the tables above are written)
    locate(y0, x0);
#pragma unroll
    for (int i = 0; i < X3_NT; ++i) {
      const int gy = y0 + wave, gx = x0 + 16 * i + n;  // this wave's row of the tile, column tile i
      inside[i] = gy < a.H && gx < a.W;
      yoff[i] = (unsigned)(my_ych * (int)DHW + gy * a.W + gx) * (unsigned)sizeof(T);
    }
    if constexpr (TAILS) {
      T* const my_tail = static_cast<T*>(tsel ? a.tail_y[1] : a.tail_y[0]);
      const int64_t tb = tsel ? a.tail_bstride[1] : a.tail_bstride[0];
      const int tch0 = tsel ? a.tail_ch0[1] : a.tail_ch0[0];
      // (+ z * HW at the ring start below; G4: the group tch0 / 4 of an interleaved tensor, four floats per voxel)
      tptr = G4T ? my_tail + b * tb + ((int64_t)(tch0 >> 2) * DHW + (y0 + wave) * a.W + x0 + n) * 4
                 : my_tail + b * tb + (int64_t)tch0 * DHW + (y0 + wave) * a.W + x0 + n;
    }
    if (tid < 3) lmaxp[tid] = 0u;                      // (ordered before the first atomicMax below by the barrier that follows)
    if constexpr (TAILS == 2) {
      if (tid < X3_TX / 2 + X3_TY / 2) {
        const bool isx = tid < X3_TX / 2;
        const int o = isx ? (x0 >> 1) + tid : (y0 >> 1) + (tid - X3_TX / 2), in = isx ? a.W : a.H;
        const LinIdx l = lin_index(min(o, (in >> 1) - 1), in, in >> 1, isx ? e.dsw : e.dsh, 1);
        // (.z != 0: the clamped pair — BOTH taps are the odd source, as in the reference; blending the even one with weight 0 would turn
        // a non-finite even source into NaN where the reference stays finite: ADVICE r04)
        const float4 ent = l.i0 == 2 * o ? make_float4(l.w0, l.w1, 0.f, 0.f) : make_float4(0.f, 1.f, 1.f, 0.f);
        if (isx) ldxt[tid] = ent; else ldyt[tid - X3_TX / 2] = ent;
      }
    }
    int zfirst = zs;
    bool fresh = true;
    // the ring (re)starts at plane zfirst with the operand scale chosen from that plane (conv3d_x3_kernel: same rule, same helpers)
    for (;;) {
      __syncthreads();
      if (tid == 0) { const unsigned note = max(lmaxp[1], lmaxp[2]); lmaxp[1] = 0u; lmaxp[2] = 0u; if (note) atomicMax(lmaxp, note); }
      prefetch(zfirst);
      const float wm = x3_wave_max(local_max_of(pf, valid));
      if (lane == 0) atomicMax(lmaxp, __float_as_uint(wm));
      __syncthreads();
      mul = x3_pow2_scale(__uint_as_float(lmaxp[0]), X3_ACT_TARGET);
      cap_bits = __float_as_uint(X3_F16_CAP / mul);
      if (tid < 32) par[tid] = par[96 + tid] * (1.f / mul);       // the epilogue's scale undoes the column's 2^-e
      commit(zfirst & 3);
      // (the ring start's notes are read by step zfirst: the word of "step zfirst - 1")
      prefetch(zfirst - 1); note_overflow((zfirst + 1) & 1); commit((zfirst + 3) & 3);
      prefetch(zfirst + 1); note_overflow((zfirst + 1) & 1); commit((zfirst + 1) & 3);
      prefetch(zfirst + 2);
      bool again = false;
      int z = zfirst;
      if constexpr (TAILS) { if (fresh) tptr += (int64_t)zs * HW * (G4T ? 4 : 1); }      // (a restart re-enters with tptr already at plane zfirst)
      fresh = false;
      while (z < ze) {
        bool r;
        switch (z & 3) {
          case 0: r = step(std::integral_constant<int, 0>{}, z); break;
          case 1: r = step(std::integral_constant<int, 1>{}, z); break;
          case 2: r = step(std::integral_constant<int, 2>{}, z); break;
          default: r = step(std::integral_constant<int, 3>{}, z); break;
        }
        if (r) { zfirst = z; again = true; break; }
        ++z;
      }
      if (!again) break;
    }
    if constexpr (TAILS == 2) {
      __syncthreads();                                 // the segment's last (odd) plane is in the LDS
      down_finish(ze - 1);
    }
  }
#ifdef RAGMI_DIAG
  if (dg_stamp) {
    XQ_STAMP(6);
    const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    if (lane == 0) {
      unsigned long long* const o = xq_stamp_buf + ((int64_t)(blockIdx.y * gridDim.x + blockIdx.x) * X3_WAVES + wave) * 16;
      for (int k = 0; k < 7; ++k) o[k] = dg_sum[k];
      o[7] = dg_steps; o[8] = dg_t0; o[9] = t1; o[10] = dg_r0; o[11] = r1;
    }
  }
#endif
}

size_t xq_lds_bytes(int nset, bool down) {
  return (size_t)2 * (nset * XQ_CGS + XQ_LOPAD) * sizeof(uint2) + (size_t)4 * nset * 2 * 64 * sizeof(uint4) + 4 * 64 * sizeof(float) + 132 * sizeof(float) +
         (down ? (size_t)(XQ_DU_SLOTS * XQ_DU_PLANE) * sizeof(float) + (size_t)(X3_TX / 2 + X3_TY / 2) * sizeof(float4) : 0);
}

// the shapes conv3d_x3_kernel<float, NSET, NSET, *> serves (x3_launch decides eligibility and the work list)
bool xq_takes(const K3Args& a, int nset, int dtype) {
  const int64_t vol = (int64_t)a.D * a.H * a.W;
  int ych = 0;                                     // buffer addressing: every byte offset inside a sample below 2^31
  for (int g = 0; g < (a.Cout + 3) / 4 && g < RAGMI_MAX_GROUPS; ++g) ych = std::max(ych, a.y_ch[g]);
  for (int t = 0; t < a.ntail; ++t) if (a.tail_cout[t] != 4) return false;
  for (int t = 0; t < a.ndown; ++t) if (a.down_cout[t] != 4) return false;
  return dtype == RAGMI_F32X3 && a.Cout <= 16 && a.Cin * vol * 4 < (1ll << 31) && (ych + 4) * vol * 4 < (1ll << 31) && nset == 2 && a.nchunks[0] == 1 && a.nchunks[1] == 1;
}

template <int NSET, int TAILS, bool G4X, bool G4T>
static int xq_launch_one(const K3Args& a, const X3Extra& e, dim3 grid, hipStream_t st) {
  static LaunchState state;
  const size_t lds = xq_lds_bytes(NSET, TAILS == 2);
  const int slots = state.slots((const void*)conv3d_x3q_kernel<NSET, TAILS, G4X, G4T>, X3_THREADS, lds, 160 * 1024);
  if (slots <= 0) return fail(RAGMI_ELAUNCH, "conv3d_x3q: cannot raise the dynamic LDS limit");
  grid.x = (unsigned)std::max<int64_t>(1, std::min<int64_t>(grid.x, std::max(256, slots) / (int)grid.y));
#ifdef RAGMI_DIAG
  static const int diag_grid = [] { const char* v = getenv("RAGMI_X3_GRID"); return v ? atoi(v) : 0; }();      // profiling builds: persistent grid size
  if (diag_grid > 0) grid.x = (unsigned)std::min<int64_t>(grid.x, diag_grid);
#endif
  hipLaunchKernelGGL((conv3d_x3q_kernel<NSET, TAILS, G4X, G4T>), grid, dim3(X3_THREADS), lds, st, a, e);
  return check_launch("conv3d_x3q");
}
template <int TAILS>
static int xq_launch_layout(const K3Args& a, const X3Extra& e, dim3 grid, hipStream_t st) {
  const bool g4x = (a.relu & RAGMI_CONV_X_G4) != 0, g4t = a.tail_g4 != 0;
  if constexpr (TAILS == 0) return g4x ? xq_launch_one<2, 0, true, false>(a, e, grid, st) : xq_launch_one<2, 0, false, false>(a, e, grid, st);
  else return g4x ? (g4t ? xq_launch_one<2, TAILS, true, true>(a, e, grid, st) : xq_launch_one<2, TAILS, true, false>(a, e, grid, st))
                  : (g4t ? xq_launch_one<2, TAILS, false, true>(a, e, grid, st) : xq_launch_one<2, TAILS, false, false>(a, e, grid, st));
}

int xq_launch(const K3Args& a, const X3Extra& e, int nset, dim3 grid, hipStream_t st) {
  if (nset != 2) return fail(RAGMI_EUNSUPPORTED, "conv3d_x3q: dual launches only");
  return a.ndown > 0 ? xq_launch_layout<2>(a, e, grid, st) : a.ntail > 0 ? xq_launch_layout<1>(a, e, grid, st) : xq_launch_layout<0>(a, e, grid, st);
}

}  // namespace ragmi
