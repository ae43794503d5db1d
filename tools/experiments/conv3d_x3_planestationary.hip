// 3x3x3 convolution with fp32 accuracy on the bf16 matrix cores ("bf16x3", ABI dtype RAGMI_F32X3): every fp32 operand is split into
// hi = bf16(x) and lo = bf16(x - hi), and a product a*b is accumulated in fp32 as hi*hi + hi*lo + lo*hi (error bound: include/rag_amd.h).
// v_mfma_f32_16x16x32_bf16 runs at 16x the rate of the fp32 MFMA forms, so three of them per product still leave ~5x.
//
// MFMA mapping (operand layout of mfma_f32_16x16x32_bf16: lane l holds A[row l&15][k = 8(l>>4)+j], B[k = 8(l>>4)+j][col l&15],
// D[row 4(l>>4)+reg][col l&15]):  rows = 16 output channels, cols = 16 consecutive voxels along x, K = 8 "pairs" of
// ((input-channel group of 4, tap), 4 channels): every lane quarter kb = l>>4 feeds two pairs.
//
// Plane-stationary z-march (round 2).  A workgroup owns an 8 x 32 (y, x) tile and walks a depth segment.  When input plane s
// arrives, the operand of a (voxel, (dy,dx) tap) is read from LDS ONCE and multiplied by the weight slices of all three dz into
// three ROTATING accumulator sets (outputs s+1, s, s-1) — 9 operand reads per voxel and plane instead of 27 (round 1 read every
// (voxel, tap) operand again for each of the three outputs it feeds; LDS operand bandwidth, not the matrix pipe, set its speed).
// Eight of the nine (dy,dx) taps of a channel group fill one K-slice of 8 pairs; the ninth tap (2,2) of the three planes s-2, s-1, s
// shares one "leftover" K-slice per output plane, so the MFMA count is unchanged: 3 full + 1 leftover slice per output = ceil(27/8).
// The tap -> lane-quarter assignment and the row stride of the LDS halo planes (48 records = 384 B) make every 8-byte operand read
// of a full slice conflict-free: the two quarters of a 32-lane half read either the same halo row (overlapping addresses broadcast)
// or rows 384 B = 128 (mod 256) apart.  One barrier per plane: the ring holds four planes (s-2, s-1, s read, s+1 written).
// Halo voxels outside the volume, planes outside the depth range and channels past Cin are out-of-range buffer-load offsets (the
// hardware returns 0): the staging path has no clamping and no zero-fill arithmetic.
#include <cstdlib>

#include "../../rag_amd/csrc/conv3d_k3.h"

namespace ragmi {

typedef __bf16 x3_bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 x3_bf16x2 __attribute__((ext_vector_type(2)));
typedef float x3_f32x2 __attribute__((ext_vector_type(2)));
typedef unsigned x3_u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned x3_u32x2 __attribute__((ext_vector_type(2)));
typedef short x3_s16x4 __attribute__((ext_vector_type(4)));

// LDS operand reads as inline asm (byte address in a VGPR + immediate offset).  The compiler does not count them: every consumer
// sits behind an explicit `s_waitcnt lgkmcnt(0)` statement that names the destination registers.
template <int OFF>
__device__ __forceinline__ x3_u32x2 x3_lds_read64(unsigned addr) {
  static_assert(OFF >= 0 && OFF < 65536, "ds_read offset is 16 bits");
  x3_u32x2 v;
  asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(OFF));
  return v;
}
template <int OFF>
__device__ __forceinline__ x3_u32x4 x3_lds_read128(unsigned addr) {
  static_assert(OFF >= 0 && OFF < 65536, "ds_read offset is 16 bits");
  x3_u32x4 v;
  asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(OFF));
  return v;
}

// the wait for every outstanding LDS read of this wave; the register operands order consumers behind it (volatile asm statements
// keep their program order, so x3_pin() calls after the wait extend the same guarantee to more registers)
__device__ __forceinline__ void x3_lds_wait(x3_u32x4& a, x3_u32x4& b, x3_u32x4& c, x3_u32x4& d, x3_u32x4& e, x3_u32x4& f) {
  asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(e), "+v"(f));
}
__device__ __forceinline__ void x3_pin(x3_u32x2& v) { asm volatile("" : "+v"(v)); }
__device__ __forceinline__ void x3_pin(x3_u32x4& v) { asm volatile("" : "+v"(v)); }

#ifdef RAGMI_X3_BENCH      // tools/x3_bench.hip only: timing-only builds that skip parts of the kernel (bit mask in X3Extra::diag)
#define X3_DIAG(e) ((e).diag)
// in-kernel stamps (diag bit 16): cycles between phase boundaries, summed per wave over its steps (cdna_hip_programming.md §7)
__device__ __forceinline__ unsigned long long x3_now() {
  unsigned long long t;
  __builtin_amdgcn_sched_barrier(0);
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");
  __builtin_amdgcn_sched_barrier(0);
  return t;
}
#else
#define X3_DIAG(e) 0
#endif
#if defined(RAGMI_X3_BENCH) && defined(RAGMI_X3_STAMPS)     // a separate build of the tool: the stamps cost registers
#define X3_STAMP(k) do { const unsigned t_ = (unsigned)x3_now(); tsum[k] += t_ - tlast; tlast = t_; } while (0)
#else
#define X3_STAMP(k) do { } while (0)
#endif

constexpr int X3_TY = 8, X3_TX = 32;
constexpr int X3_HY = X3_TY + 2;                 // halo rows of a plane
constexpr int X3_NQ = 10;                        // 16-byte quads per halo row: voxels x0-4 .. x0+35 (34 are used; aligned for dwordx4 loads)
constexpr int X3_RING = 4;                       // planes in LDS
constexpr int X3_LO_PAD = 16;                    // pad records between the hi and the lo planes (see LO_REC in the kernel)
constexpr int64_t X3_MIN_VOXELS = 1 << 18;       // below this the z-marching columns do not fill the chip
constexpr unsigned X3_OOB = 0x80000000u;         // buffer-load offset past every descriptor: the hardware range check returns 0
constexpr unsigned X3_MASK = 0x40000000u;        // output descriptors span 2^30 bytes: adding this to a store offset drops the lane

// K-slices of ONE accumulator set with ncgs input-channel groups: 3 per group (dz = 0,1,2: the eight taps (dy,dx) != (2,2)),
// then ceil(ncgs/2) leftover slices (tap (2,2) of the three planes for two groups each)
__host__ __device__ constexpr int x3_nsls(int ncgs) { return 3 * ncgs + (ncgs + 1) / 2; }

// (dy, dx) of pair slot jj (0/1) of lane quarter kb in a full slice.  Same-row taps sit in the two quarters of one 32-lane half
// (their overlapping reads broadcast); (0,2) / (1,2) are one row = 128 (mod 256) bytes apart.
__host__ __device__ constexpr int x3_tap_dy(int jj, int kb) { return jj == 0 ? (kb >> 1) : (kb < 2 ? 2 : kb - 2); }
__host__ __device__ constexpr int x3_tap_dx(int jj, int kb) { return jj == 0 ? (kb & 1) : (kb < 2 ? kb : 2); }

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
__device__ __forceinline__ unsigned x3_split2(float v0, float v1, unsigned& lo) {
  const x3_bf16x2 h = __builtin_convertvector(x3_f32x2{v0, v1}, x3_bf16x2);
  const unsigned hb = __builtin_bit_cast(unsigned, h);
  const float r0 = v0 - __uint_as_float(hb << 16), r1 = v1 - __uint_as_float(hb & 0xffff0000u);
  lo = __builtin_bit_cast(unsigned, __builtin_convertvector(x3_f32x2{r0, r1}, x3_bf16x2));
  return hb;
}

// packed weight fragments of ONE accumulator set (a conv with Cout outputs and Cin = 4 * ncgs inputs):
// wf[((cog * nsls + s) * 2 + hl) * 64 + lane] (uint4 = 8 bf16) = A[row = lane & 15][k = 8 (lane>>4) + j] of K-slice s:
//   s = 3 cg + dz (full slice):   pair slot jj = j>>2 -> tap (dz, x3_tap_dy(jj, kb), x3_tap_dx(jj, kb)), channel 4 cg + (j&3)
//   s = 3 ncgs + m (leftover):    pair slot jj -> group 2m + jj, tap (dz = kb, 2, 2) for kb < 3, zero for kb = 3
// The source is indexed like the fp32 pack (transpose / planar options of ragmi_conv3d_k3_pack_ex).
__device__ __forceinline__ void x3_pack_one(const float* __restrict__ w, uint4* __restrict__ wf, int Cout, int Cin, int nsls, int ncog,
                                            int transpose, int planar, int idx) {
  if (idx >= ncog * nsls * 64) return;
  const int lane = idx & 63, s = (idx >> 6) % nsls, cog = idx / (64 * nsls);
  const int co = cog * 16 + (lane & 15), kb = lane >> 4;
  const int ncgs = (Cin + 3) / 4;
  unsigned short hi[8], lo[8];
  for (int j = 0; j < 8; ++j) {
    const int jj = j >> 2;
    int cg, tap;
    bool live = true;
    if (s < 3 * ncgs) {
      cg = s / 3;
      tap = (s % 3) * 9 + x3_tap_dy(jj, kb) * 3 + x3_tap_dx(jj, kb);
    } else {
      cg = 2 * (s - 3 * ncgs) + jj;
      tap = kb * 9 + 8;
      live = kb < 3 && cg < ncgs;
    }
    const int ci = 4 * cg + (j & 3);
    float v = 0.f;
    if (live && co < Cout && ci < Cin) {
      const int taps = planar ? 9 : 27;
      int t = planar ? tap - 9 : tap;
      if (t >= 0 && t < taps) {
        if (transpose) t = taps - 1 - t;
        v = transpose ? w[((int64_t)ci * Cout + co) * taps + t] : w[((int64_t)co * Cin + ci) * taps + t];
      }
    }
    x3_split(v, hi[j], lo[j]);
  }
  auto pk = [](const unsigned short* h) {
    return make_uint4(h[0] | ((unsigned)h[1] << 16), h[2] | ((unsigned)h[3] << 16), h[4] | ((unsigned)h[5] << 16), h[6] | ((unsigned)h[7] << 16));
  };
  wf[((int64_t)(cog * nsls + s) * 2 + 0) * 64 + lane] = pk(hi);
  wf[((int64_t)(cog * nsls + s) * 2 + 1) * 64 + lane] = pk(lo);
}

struct X3Extra {
  const uint4* wf[2];        // packed fragments per accumulator set
  int nseg, seg_len, nwork, bf16;   // bf16 != 0: bf16 activation storage (kernel instantiation selector)
  unsigned xbytes;           // bytes of one batch item's input channels (buffer-descriptor range: offsets past it load 0)
  int diag;                  // honoured by tools/x3_bench.hip builds only: 1 no epilogue, 2 no MFMA, 4 no staging, 16 stamps
  unsigned long long* stamps;   // [workgroup][wave][16] cycle sums (diag bit 16)
};

// raw buffer loads as inline asm (out-of-range offsets return 0): one (element, channel) of the plane in flight
__device__ __forceinline__ void x3_bload_asm(x3_u32x4& v, unsigned off, __amdgpu_buffer_rsrc_t r, float) {
  asm volatile("buffer_load_dwordx4 %0, %1, %2, 0 offen" : "=v"(v) : "v"(off), "s"(r));
}
__device__ __forceinline__ void x3_bload_asm(x3_u32x2& v, unsigned off, __amdgpu_buffer_rsrc_t r, bf16_t) {
  asm volatile("buffer_load_dwordx2 %0, %1, %2, 0 offen" : "=v"(v) : "v"(off), "s"(r));
}
__device__ __forceinline__ void x3_bload_asm(unsigned& v, unsigned off, __amdgpu_buffer_rsrc_t r, float) {
  asm volatile("buffer_load_dword %0, %1, %2, 0 offen" : "=v"(v) : "v"(off), "s"(r));
}
__device__ __forceinline__ void x3_bload_asm(unsigned& v, unsigned off, __amdgpu_buffer_rsrc_t r, bf16_t) {
  asm volatile("buffer_load_ushort %0, %1, %2, 0 offen" : "=v"(v) : "v"(off), "s"(r));
}
// The wait for the plane in flight: all but the N youngest vector-memory operations of the wave are done.  ONE statement names
// every destination register of the loads, so that no consumer — and no register copy the allocator might want — can be placed
// between a load and its wait (a copy hoisted above the wait reads the register before the data has landed).
#define X3_W4(a) "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3])
template <int N, class R> __device__ __forceinline__ void x3_vm_wait(R (&p)[1][4]) { asm volatile("s_waitcnt vmcnt(%4)" : X3_W4(p[0]) : "n"(N)); }
template <int N, class R> __device__ __forceinline__ void x3_vm_wait(R (&p)[2][4]) { asm volatile("s_waitcnt vmcnt(%8)" : X3_W4(p[0]), X3_W4(p[1]) : "n"(N)); }
template <int N, class R> __device__ __forceinline__ void x3_vm_wait(R (&p)[3][4]) {
  asm volatile("s_waitcnt vmcnt(%12)" : X3_W4(p[0]), X3_W4(p[1]), X3_W4(p[2]) : "n"(N));
}
template <int N, class R> __device__ __forceinline__ void x3_vm_wait(R (&p)[4][4]) {
  asm volatile("s_waitcnt vmcnt(%16)" : X3_W4(p[0]), X3_W4(p[1]), X3_W4(p[2]), X3_W4(p[3]) : "n"(N));
}
template <int N, class R> __device__ __forceinline__ void x3_vm_wait(R (&p)[5][4]) {
  asm volatile("s_waitcnt vmcnt(%20)" : X3_W4(p[0]), X3_W4(p[1]), X3_W4(p[2]), X3_W4(p[3]), X3_W4(p[4]) : "n"(N));
}
template <int N, class R> __device__ __forceinline__ void x3_vm_wait(R (&p)[6][4]) {
  asm volatile("s_waitcnt vmcnt(%24)" : X3_W4(p[0]), X3_W4(p[1]), X3_W4(p[2]), X3_W4(p[3]), X3_W4(p[4]), X3_W4(p[5]) : "n"(N));
}
template <int N, class R> __device__ __forceinline__ void x3_vm_wait(R (&p)[7][4]) {
  asm volatile("s_waitcnt vmcnt(%28)" : X3_W4(p[0]), X3_W4(p[1]), X3_W4(p[2]), X3_W4(p[3]), X3_W4(p[4]), X3_W4(p[5]), X3_W4(p[6]) : "n"(N));
}
#undef X3_W4
__device__ __forceinline__ void x3_pin(unsigned& v) { asm volatile("" : "+v"(v)); }
// raw buffer stores (out-of-range offsets are dropped): every lane always executes them, so a step issues a FIXED number of
// store instructions and the staging wait can count them
// (the offset is pinned in ONE register first: left as a select feeding the store, hipcc turns `valid ? offset : X3_OOB` into
// divergent control flow with a store on each side — two instructions whenever the lanes of a wave disagree)
__device__ __forceinline__ void x3_bstore(float v, unsigned off, unsigned soff, __amdgpu_buffer_rsrc_t r, float) {
  asm volatile("" : "+v"(off));
  __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v), r, off, soff, 0);
}
__device__ __forceinline__ void x3_bstore(float v, unsigned off, unsigned soff, __amdgpu_buffer_rsrc_t r, bf16_t) {
  asm volatile("" : "+v"(off));
  __builtin_amdgcn_raw_buffer_store_b16(to_bf16(v), r, off, soff, 0);
}

// T = activation storage: float (three MFMAs per product) or bf16_t (the activations ARE bf16: no lo copy, two MFMAs per
// product — weight hi and lo — and half the LDS operand traffic).  NCG = input-channel groups of 4 over all sets, NSET accumulator
// sets (2: out = act(bnA(convA(x[:, :C]))) + act(bnB(convB(x[:, C:]))), the Cell_3d sibling fusion).  NW waves per workgroup, each
// owning 16 / NW column tiles of 16 voxels; ROWV = voxel records per LDS halo row (48: conflict-free; 40: smaller planes).
// VEC: 16-byte aligned quads along x (W % 4 == 0, aligned bases) instead of single voxels in the staging path.
//
// The step (one input plane) is ONE basic block — no data-dependent or position-dependent branch: every step multiplies all three
// dz slices and the leftover slice (the accumulators of output planes outside the depth segment are simply never stored; a
// segment of L planes costs L + 2 steps of MFMAs), masked stores replace exec masking, staging lanes past the plane repeat its last
// element.  hipcc can then interleave the staging / epilogue VALU work and the operand reads of the next column tile with the
// MFMAs of the current one: measured on this chip, an instruction issued between two MFMAs of the same wave costs ~2.5 cycles,
// in a phase of its own ~13 (the partner wave's MFMAs own the issue port), and the phase-separated version of this kernel ran at
// a third of the matrix pipe's rate for exactly that reason.
template <class T, int NCG, int NSET, bool TAILS, int NW, int ROWV, bool VEC>
__global__ __launch_bounds__(NW * 64, NW / 2) void conv3d_x3_kernel(K3Args a, X3Extra e) {
  constexpr bool BF = std::is_same<T, bf16_t>::value;
  constexpr int THREADS = NW * 64, NT = 16 / NW;
  constexpr int NCGS = NCG / NSET, NLS = (NCGS + 1) / 2, NSLS = 3 * NCGS + NLS, NSL = NSET * NSLS;
  constexpr int PL = X3_HY * ROWV;                                        // voxel records of one (group, slot) plane
  constexpr int NEL = NCG * X3_HY * (VEC ? X3_NQ : 34);                   // staged elements of a plane (quads or voxels)
  constexpr int NP = (NEL + THREADS - 1) / THREADS, VPE = VEC ? 4 : 1;    // elements per thread, voxels per element
  // weight fragments resident in registers for the whole kernel when they fit (the level-3 cells: 8 slices x 2 x 4 VGPRs)
  constexpr bool ARES = NSL * 8 <= 64 && NT >= 4;
  // (NP <= 7: one asm statement can name at most 30 operands — the staging wait names every load destination)
  static_assert(NCG % NSET == 0 && NT >= 1 && NP <= 7, "bad instantiation");
  extern __shared__ __attribute__((aligned(16))) uint2 x3_lds[];          // hi[NCG][RING][PL] | lo[...] (uint2 = 4 bf16 of a voxel) | weights | params
  // record offset of the lo planes (absent for bf16 storage).  The 16 pad records keep the hi -> lo distance off every multiple of
  // 512 B: at such a distance hipcc fuses the hi and lo reads of one address into ds_read2st64_b64, whose halves then need moves
  // into the MFMA operand tuples (60 v_mov per plane) and which runs at half the LDS rate of two ds_read_b64
  constexpr int LO_REC = NCG * X3_RING * PL + X3_LO_PAD;
  constexpr int LO_BYTES = LO_REC * (int)sizeof(uint2);
  static_assert(LO_BYTES % 512 != 0 && LO_BYTES > 2040, "hi / lo reads would fuse");
  uint4* const lw = reinterpret_cast<uint4*>(x3_lds + (BF ? NCG * X3_RING * PL : LO_REC + NCG * X3_RING * PL));   // [set][slice][hi/lo][64 lanes]
  float* const par = reinterpret_cast<float*>(lw + NSL * 2 * 64);         // BatchNorm: [set][scale | shift][16]; tails at 64
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, n = lane & 15, kb = lane >> 4;
  const int cog = blockIdx.y;
  const int HW = a.H * a.W;
  const int64_t DHW = (int64_t)HW * a.D;
  constexpr unsigned ESZ = (unsigned)sizeof(T);
  x3_u32x4 wres[ARES ? NSL : 1][2];
  if constexpr (ARES) {
#pragma unroll
    for (int sl = 0; sl < NSL; ++sl)
#pragma unroll
      for (int hl = 0; hl < 2; ++hl) {
        const uint4 t = e.wf[sl / NSLS][((int64_t)(cog * NSLS + sl % NSLS) * 2 + hl) * 64 + lane];
        wres[sl][hl] = x3_u32x4{t.x, t.y, t.z, t.w};
      }
  } else {
    for (int i = tid; i < NSL * 2 * 64; i += THREADS) {
      const int set = i / (NSLS * 2 * 64), r = i % (NSLS * 2 * 64);
      lw[i] = e.wf[set][(int64_t)cog * NSLS * 2 * 64 + r];
    }
  }
  for (int i = tid; i < NSET * 32; i += THREADS) {
    const int st = i >> 5, which = (i >> 4) & 1, co = cog * 16 + (i & 15);
    const float* src = which ? a.shift[st] : a.scale[st];
    par[i] = (co < a.Cout && src) ? src[co] : (which ? 0.f : 1.f);
  }
  // Fused consumer 1x1x1 convs ("tails") on the matrix cores: out_t[k][voxel] = sum_c W_t[k][c] * v[c][voxel] is one more
  // 16x16x32 product whose K slots are laid out so that every lane quarter feeds ITS OWN four channels — slots 8kb..8kb+3 carry
  // v_hi, slots 8kb+4..8kb+7 carry v_lo of channels 4kb..4kb+3 — so no value crosses lanes.  Rows: tail 0 -> 0..3, tail 1 -> 4..7.
  //   ta1 = W_hi in all eight slots (W_hi * (v_hi + v_lo)),  ta2 = W_lo in the hi slots only (W_lo * v_hi)
  x3_bf16x8 ta1 = {}, ta2 = {};
  if constexpr (TAILS) {
    unsigned short h1[8], h2[8];
    const int row = n, tl = row >> 2, k = row & 3;     // n = lane & 15 is the A row
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int c = cog * 16 + 4 * kb + (j & 3);
      float wv = 0.f;
      if (tl < a.ntail && k < a.tail_cout[tl < 2 ? tl : 0] && c < a.Cout) wv = a.tail_w[tl < 2 ? tl : 0][k * a.Cout + c];
      unsigned short hi, lo;
      x3_split(wv, hi, lo);
      h1[j] = hi;
      h2[j] = j < 4 ? lo : (unsigned short)0;
    }
    ta1 = __builtin_bit_cast(x3_bf16x8, make_uint4(h1[0] | ((unsigned)h1[1] << 16), h1[2] | ((unsigned)h1[3] << 16), h1[4] | ((unsigned)h1[5] << 16), h1[6] | ((unsigned)h1[7] << 16)));
    ta2 = __builtin_bit_cast(x3_bf16x8, make_uint4(h2[0] | ((unsigned)h2[1] << 16), h2[2] | ((unsigned)h2[3] << 16), h2[4] | ((unsigned)h2[5] << 16), h2[6] | ((unsigned)h2[7] << 16)));
    if (tid < 16) {          // tail BatchNorm: par[64 + 8 t + (scale: 0..3 | shift: 4..7)] for tail t
      const int t = tid >> 3, which = (tid >> 2) & 1, r = tid & 3;
      const bool ok = t < a.ntail && r < a.tail_cout[t];
      const float* src = which ? a.tail_shift[t] : a.tail_scale[t];
      par[64 + tid] = (ok && src) ? src[r] : (which ? 0.f : 1.f);
    }
  }

  // ---- staging geometry, per thread, fixed for the kernel: element p -> (group, halo row, quad or voxel) and its LDS record.
  // Lanes past the last element repeat it (same loads, same LDS writes): no branch in the staging path.
  int st_rec[NP];          // record index inside slot 0 (hi planes)
  int st_cg[NP], st_yy[NP], st_xo[NP];
#pragma unroll
  for (int p = 0; p < NP; ++p) {
    const int el = min(p * THREADS + tid, NEL - 1);
    constexpr int PER_ROW = VEC ? X3_NQ : 34, PER_CG = X3_HY * PER_ROW;
    const int cg = el / PER_CG, r = el % PER_CG, yy = r / PER_ROW, xq = r % PER_ROW;
    st_cg[p] = cg;
    st_yy[p] = yy;
    st_xo[p] = VEC ? 4 * xq - 4 : xq - 1;                                // x offset of the element's first voxel from x0
    st_rec[p] = (cg * X3_RING * X3_HY + yy) * ROWV + (VEC ? 4 * xq : xq + 3);
  }
  // the plane in flight, as loaded: one register tuple per (element, channel) — 4 fp32 / 4 bf16 voxels (VEC) or one voxel
  using PfT = std::conditional_t<VEC, std::conditional_t<BF, x3_u32x2, x3_u32x4>, unsigned>;
  PfT pf[NP][4];
  unsigned poff[NP];       // per column: byte offset of the element inside the batch item at z = 0 (X3_OOB: outside in y / x)

  const T* const x = static_cast<const T*>(a.x);
  // LDS byte addresses of this lane's operand reads inside slot 0 / group 0, per column tile of the wave: pair slots 0 and 1 of the
  // full slices, the leftover tap.  Each is made opaque to the compiler: known to differ by constants, hipcc fuses the 8-byte reads
  // of two tiles into ds_read2_b64 — half the LDS rate, and the halves then need moves into the MFMA operand tuples.
  const unsigned lds_base = (unsigned)reinterpret_cast<uintptr_t>(x3_lds);
  const int lrow0 = wave * NT >> 1;                                       // first output row of this wave inside the tile
  unsigned tb0[NT], tb1[NT], tbl[NT];
#pragma unroll
  for (int i = 0; i < NT; ++i) {
    const int lb = (lrow0 + (i >> 1)) * ROWV + (i & 1) * 16 + n + 3;
    tb0[i] = lds_base + (unsigned)(lb + x3_tap_dy(0, kb) * ROWV + x3_tap_dx(0, kb)) * 8u;
    tb1[i] = lds_base + (unsigned)(lb + x3_tap_dy(1, kb) * ROWV + x3_tap_dx(1, kb)) * 8u;
    tbl[i] = lds_base + (unsigned)(lb + 2 * ROWV + 2) * 8u;
    asm volatile("" : "+v"(tb0[i]), "+v"(tb1[i]), "+v"(tbl[i]));
  }
  const int g = cog * 4 + kb, ngroups = (a.Cout + 3) >> 2;
  // per-lane destinations and epilogue parameters, read ONCE (indexing the kernel-argument arrays with a lane-dependent index inside
  // the loop is a vector memory load per use).  Offsets of masked lanes / channels are X3_MASK: sums of up to two masks and a valid
  // offset stay in [2^30, 2^32), past the 2^30-byte range of the output descriptors.
  const int my_ych = g < ngroups ? a.y_ch[g < RAGMI_MAX_GROUPS ? g : 0] : 0;
  // channel r of the group sits r planes further (a wave-uniform `soffset` of the store); a partial last group masks its
  // missing channels per store
  const unsigned ych_off = (a.store_main && g < ngroups) ? (unsigned)my_ych * (unsigned)DHW * ESZ : X3_MASK;
  const int my_nch = a.Cout - 4 * g;                 // >= 4: every channel of this lane's group exists
  const unsigned plane_b = (unsigned)DHW * ESZ;
  const float relu_lo = a.relu ? 0.f : -__builtin_inff();
  // tails: lane quarter kb computes tail kb (rows 4 kb + r = its output r); tail_cnt = its channel count (0: no tail here)
  unsigned tl_off0 = X3_MASK, tl_off1 = X3_MASK;
  int tail_cnt = 0;
  float tl_lo = 0.f;
  if constexpr (TAILS) {
    const int tsel = kb & 1;
    tail_cnt = kb < a.ntail ? (tsel ? a.tail_cout[1] : a.tail_cout[0]) : 0;
    const unsigned o = tail_cnt > 0 ? (unsigned)(tsel ? a.tail_ch0[1] : a.tail_ch0[0]) * (unsigned)DHW * ESZ : X3_MASK;
    tl_off0 = tsel == 0 ? o : X3_MASK;               // the two tails live in different buffers: one descriptor each
    tl_off1 = tsel == 1 ? o : X3_MASK;
    tl_lo = (tsel ? a.tail_relu[1] : a.tail_relu[0]) ? 0.f : -__builtin_inff();
  }

  f32x4 acc[3][NSET][NT];   // rotating: output plane z lives in acc[(z - zs + 1) % 3] (indices are compile-time after unrolling)

  // XCD-aware schedule: workgroup j runs on XCD j % 8 (round-robin dispatch); give every XCD one contiguous chunk of the
  // (x-fastest) work list so that neighbouring columns — which share halo rows and cache lines — meet in the same L2
  const int chunk = (e.nwork + 7) / 8;
  for (int j = blockIdx.x; j < chunk * 8; j += gridDim.x) {
    const int work = (j & 7) * chunk + (j >> 3);
    if ((j >> 3) >= chunk || work >= e.nwork) continue;
    int t = work;
    const int x0 = (t % a.tiles_x) * X3_TX; t /= a.tiles_x;
    const int y0 = (t % a.tiles_y) * X3_TY; t /= a.tiles_y;
    const int seg = t % e.nseg, b = t / e.nseg;
    const int zs = seg * e.seg_len, ze = min(a.D, zs + e.seg_len);
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<T*>(x + (int64_t)b * a.x_bstride), 0, e.xbytes, 0x00020000);
    // destinations: offsets below 2^30 are in range (x3_eligible checks the extents); X3_MASK added to an offset masks the lane
    const __amdgpu_buffer_rsrc_t yrsrc = __builtin_amdgcn_make_buffer_rsrc(static_cast<T*>(a.y) + (int64_t)b * a.y_bstride, 0, X3_MASK, 0x00020000);
    const __amdgpu_buffer_rsrc_t trsrc0 = __builtin_amdgcn_make_buffer_rsrc(static_cast<T*>(a.tail_y[0]) + (int64_t)b * a.tail_bstride[0], 0, TAILS ? X3_MASK : 0u, 0x00020000);
    const __amdgpu_buffer_rsrc_t trsrc1 = __builtin_amdgcn_make_buffer_rsrc(static_cast<T*>(a.tail_y[a.ntail > 1 ? 1 : 0]) + (int64_t)b * a.tail_bstride[a.ntail > 1 ? 1 : 0], 0, (TAILS && a.ntail > 1) ? X3_MASK : 0u, 0x00020000);
    unsigned tile_off[NT];     // byte offset of this lane's voxel of column tile i inside a channel plane at z = 0 (masked outside the volume)
#pragma unroll
    for (int i = 0; i < NT; ++i) {
      const int nt = wave * NT + i;
      const int gy = y0 + (nt >> 1), gx = x0 + (nt & 1) * 16 + n;
      tile_off[i] = (gy < a.H && gx < a.W) ? (unsigned)(gy * a.W + gx) * ESZ : X3_MASK;
    }
#pragma unroll
    for (int p = 0; p < NP; ++p) {
      const int gy = y0 - 1 + st_yy[p], gx = x0 + st_xo[p];
      const bool ok = (unsigned)gy < (unsigned)a.H && (unsigned)gx < (unsigned)a.W;
      poff[p] = ok ? ((unsigned)(st_cg[p] * 4) * (unsigned)DHW + (unsigned)(gy * a.W + gx)) * ESZ : X3_OOB;
    }
    // The loads are inline asm: hipcc does not count them, so no compiler-inserted `s_waitcnt vmcnt(0)` ever drains them early.
    // staged() is their wait; it names every destination register in ONE statement.
    auto prefetch = [&](int gz) {
      const unsigned zoff = (unsigned)gz < (unsigned)a.D ? (unsigned)(gz * HW) * ESZ : X3_OOB;
#pragma unroll
      for (int p = 0; p < NP; ++p) {
        const unsigned o = (poff[p] | zoff) >= X3_OOB ? X3_OOB : poff[p] + zoff;
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          const unsigned oc = o + (unsigned)c * (unsigned)DHW * ESZ;      // (o = X3_OOB stays out of range: the channel offset is < 2^30)
          x3_bload_asm(pf[p][c], oc, rsrc, T{});
        }
      }
    };
    auto staged = [&]() { x3_vm_wait<0>(pf); };
    auto pfval = [&](int p, int c, int k) -> float {        // channel c of voxel k of element p, as fp32
      if constexpr (VEC && !BF) return __uint_as_float(pf[p][c][k]);
      else if constexpr (VEC && BF) return __uint_as_float(k & 1 ? pf[p][c][k >> 1] & 0xffff0000u : pf[p][c][k >> 1] << 16);
      else if constexpr (BF) return __uint_as_float(pf[p][c] << 16);
      else return __uint_as_float(pf[p][c]);
    };
    auto commit = [&](int slot) {          // registers -> ring plane `slot` (bf16 hi / lo records)
#pragma unroll
      for (int p = 0; p < NP; ++p) {
        uint2* const dh = x3_lds + st_rec[p] + slot * PL;
#pragma unroll
        for (int k = 0; k < VPE; ++k) {
          unsigned l01, l23;
          const unsigned h01 = x3_split2(pfval(p, 0, k), pfval(p, 1, k), l01), h23 = x3_split2(pfval(p, 2, k), pfval(p, 3, k), l23);
          dh[k] = make_uint2(h01, h23);
          if constexpr (!BF) dh[k + LO_REC] = make_uint2(l01, l23);
        }
      }
    };
    // epilogue of output plane z (accumulator AI): lane holds channels 4 g + reg (g = cog*4 + kb) of voxel n of each column tile.
    // Every store is a buffer store that ALL lanes execute: a masked lane (or a plane outside the segment) adds X3_MASK to its
    // offset, which puts it past the descriptor's range — the hardware drops it.  No exec masking, no selects.
    auto epilogue = [&](auto ai_, int z, bool zvalid) {
      constexpr int AI = decltype(ai_)::value;
      const unsigned zo = zvalid ? (unsigned)(z * HW) * ESZ : X3_MASK;
      f32x4 bn_sc[NSET], bn_sh[NSET];
#pragma unroll
      for (int st = 0; st < NSET; ++st) {
        bn_sc[st] = *reinterpret_cast<const f32x4*>(par + st * 32 + 4 * kb);
        bn_sh[st] = *reinterpret_cast<const f32x4*>(par + st * 32 + 16 + 4 * kb);
      }
      static_for<NT>([&](auto i_) {
        constexpr int i = decltype(i_)::value;
        const unsigned vo = tile_off[i] + zo;       // (up to three masks add up below: 3 * 2^30 does not wrap)
        float v[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float u0 = fmaxf(fmaf(acc[AI][0][i][r], bn_sc[0][r], bn_sh[0][r]), relu_lo);
          if constexpr (NSET == 2) v[r] = u0 + fmaxf(fmaf(acc[AI][NSET - 1][i][r], bn_sc[NSET - 1][r], bn_sh[NSET - 1][r]), relu_lo);
          else v[r] = u0;
        }
        const unsigned yo = vo + ych_off;
#pragma unroll
        for (int r = 0; r < 4; ++r) x3_bstore(v[r], r < my_nch ? yo : X3_MASK, (unsigned)r * plane_b, yrsrc, T{});
        if constexpr (TAILS) {
          unsigned l01, l23;
          const unsigned h01 = x3_split2(v[0], v[1], l01), h23 = x3_split2(v[2], v[3], l23);
          const x3_bf16x8 bv = __builtin_bit_cast(x3_bf16x8, make_uint4(h01, h23, l01, l23));
          f32x4 tacc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ta1, bv, f32x4{0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
          tacc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ta2, bv, tacc, 0, 0, 0);
          const f32x4 tsc = *reinterpret_cast<const f32x4*>(par + 64 + 8 * (kb & 1)), tsh = *reinterpret_cast<const f32x4*>(par + 68 + 8 * (kb & 1));
          const unsigned t0 = vo + tl_off0, t1 = vo + tl_off1;
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const float u = fmaxf(fmaf(tacc[r], tsc[r], tsh[r]), tl_lo);
            x3_bstore(u, r < tail_cnt ? t0 : X3_MASK, (unsigned)r * plane_b, trsrc0, T{});
            x3_bstore(u, r < tail_cnt ? t1 : X3_MASK, (unsigned)r * plane_b, trsrc1, T{});
          }
        }
      });
    };

    __syncthreads();                                   // the previous column's LDS reads are done (and the tables above are written)
    prefetch(zs - 1);
    staged();
    commit((zs - 1) & 3);
    prefetch(zs);

    // one step = one input plane s (zs-1 <= s <= ze): PH = (s - (zs-1)) % 3 fixes the accumulator rotation at compile time
    auto step = [&](auto ph_, int s) {
      constexpr int PH = decltype(ph_)::value;
      constexpr int A0 = (PH + 1) % 3, A1 = PH, A2 = (PH + 2) % 3;         // accumulators of outputs s+1 (dz=0), s (dz=1), s-1 (dz=2)
      __syncthreads();                                 // plane s is in LDS; every wave is done with plane s-3 (the slot of s+1)
      // everything this wave has in flight — the loads of plane s+1 and the stores of the previous step's epilogue — was issued
      // about one step ago: the wait is short.  (Loads and stores do not retire in issue order with respect to each other on this
      // chip: a counted vmcnt(N) that lets N younger stores fly returned before the loads had.)
      staged();
      commit((s + 1) & 3);
      prefetch(s + 2);
      // output plane s-2 was completed by the previous step; its accumulator is the one this step's dz = 0 products reopen, so it is
      // stored first — the stores then have this step's whole MFMA phase to drain before the next wait
      epilogue(std::integral_constant<int, A0>{}, s - 2, s - 2 >= zs);
      const unsigned so = (unsigned)((s & 3) * PL * 8);
      const unsigned sq = (unsigned)(((s - 2 + (kb < 3 ? kb : 2)) & 3) * PL * 8);        // leftover tap: plane s-2+kb
      constexpr int CGB = X3_RING * PL * 8;
      // Units = (accumulator set, channel group, column tile), walked in that order.  The operands of unit u+1 are read from LDS
      // while the MFMAs of unit u issue (two register buffers); a scheduling barrier after each unit keeps hipcc from hoisting
      // every unit's reads to the top of the step (it does, and then spills).
      constexpr int NU = NSET * NCGS * NT;
      struct Ops { uint2 h0, h1, l0, l1, g0, g1, k0, k1; };      // full slices: pair slots 0/1 hi, lo; leftover: groups 2m/2m+1 hi, lo
      Ops ops[2];
      auto load_unit = [&](auto u_, Ops& o) {
        constexpr int u = decltype(u_)::value, st = u / (NCGS * NT), cgl = (u / NT) % NCGS, i = u % NT;
        constexpr int cgo = (st * NCGS + cgl) * CGB;
        const char* const p0 = reinterpret_cast<const char*>(x3_lds) + (tb0[i] - lds_base) + so + cgo;
        const char* const p1 = reinterpret_cast<const char*>(x3_lds) + (tb1[i] - lds_base) + so + cgo;
        o.h0 = *reinterpret_cast<const uint2*>(p0);
        o.h1 = *reinterpret_cast<const uint2*>(p1);
        if constexpr (!BF) {
          o.l0 = *reinterpret_cast<const uint2*>(p0 + LO_BYTES);
          o.l1 = *reinterpret_cast<const uint2*>(p1 + LO_BYTES);
        }
        if constexpr (cgl == NCGS - 1 && NLS == 1) {         // the (single) leftover slice rides on the last group's unit
          constexpr bool TWO = NCGS > 1;
          const char* const pl = reinterpret_cast<const char*>(x3_lds) + (tbl[i] - lds_base) + sq + st * NCGS * CGB;
          o.g0 = *reinterpret_cast<const uint2*>(pl);
          if constexpr (TWO) o.g1 = *reinterpret_cast<const uint2*>(pl + CGB);
          if constexpr (!BF) {
            o.k0 = *reinterpret_cast<const uint2*>(pl + LO_BYTES);
            if constexpr (TWO) o.k1 = *reinterpret_cast<const uint2*>(pl + CGB + LO_BYTES);
          }
        }
      };
      auto frag = [&](auto sl_, int hl) -> x3_bf16x8 {
        constexpr int sl = decltype(sl_)::value;
        if constexpr (ARES) return __builtin_bit_cast(x3_bf16x8, wres[sl][hl]);
        else return __builtin_bit_cast(x3_bf16x8, lw[(sl * 2 + hl) * 64 + lane]);
      };
      auto mfma_unit = [&](auto u_, const Ops& o) {
        constexpr int u = decltype(u_)::value, st = u / (NCGS * NT), cgl = (u / NT) % NCGS, i = u % NT;
        const x3_bf16x8 bh = __builtin_bit_cast(x3_bf16x8, make_uint4(o.h0.x, o.h0.y, o.h1.x, o.h1.y));
        x3_bf16x8 bl = bh;
        if constexpr (!BF) bl = __builtin_bit_cast(x3_bf16x8, make_uint4(o.l0.x, o.l0.y, o.l1.x, o.l1.y));
        auto block = [&](auto dz_, auto acc_) {
          constexpr int DZ = decltype(dz_)::value, AI = decltype(acc_)::value;
          const x3_bf16x8 ah = frag(std::integral_constant<int, st * NSLS + cgl * 3 + DZ>{}, 0);
          const x3_bf16x8 al = frag(std::integral_constant<int, st * NSLS + cgl * 3 + DZ>{}, 1);
          // dz = 0 opens the accumulator of output s+1 (its first contribution): C = 0
          f32x4 c = (DZ == 0 && cgl == 0) ? f32x4{0.f, 0.f, 0.f, 0.f} : acc[AI][st][i];
          c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bh, c, 0, 0, 0);
          if constexpr (!BF) c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bl, c, 0, 0, 0);
          acc[AI][st][i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al, bh, c, 0, 0, 0);
        };
        block(std::integral_constant<int, 1>{}, std::integral_constant<int, A1>{});
        block(std::integral_constant<int, 2>{}, std::integral_constant<int, A2>{});
        block(std::integral_constant<int, 0>{}, std::integral_constant<int, A0>{});
        // leftover slice: tap (2,2) of planes s-2, s-1, s (lane quarters 0, 1, 2) completes output plane s-1
        if constexpr (cgl == NCGS - 1 && NLS == 1) {
          const x3_bf16x8 wh = frag(std::integral_constant<int, st * NSLS + 3 * NCGS>{}, 0), wl = frag(std::integral_constant<int, st * NSLS + 3 * NCGS>{}, 1);
          if constexpr (NCGS > 1) {
            const x3_bf16x8 h = __builtin_bit_cast(x3_bf16x8, make_uint4(o.g0.x, o.g0.y, o.g1.x, o.g1.y));
            f32x4 c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wh, h, acc[A2][st][i], 0, 0, 0);
            if constexpr (!BF) c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wh, __builtin_bit_cast(x3_bf16x8, make_uint4(o.k0.x, o.k0.y, o.k1.x, o.k1.y)), c, 0, 0, 0);
            acc[A2][st][i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wl, h, c, 0, 0, 0);
          } else {
            // one group in the slice: a K = 16 product (pair slot 0 only): the fragment's pair-slot-1 half is zero and is not read
            const x3_s16x4 ah = __builtin_bit_cast(x3_s16x4, __builtin_shufflevector(wh, wh, 0, 1, 2, 3));
            const x3_s16x4 al = __builtin_bit_cast(x3_s16x4, __builtin_shufflevector(wl, wl, 0, 1, 2, 3));
            const x3_s16x4 h = __builtin_bit_cast(x3_s16x4, o.g0);
            f32x4 c = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(ah, h, acc[A2][st][i], 0, 0, 0);
            if constexpr (!BF) c = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(ah, __builtin_bit_cast(x3_s16x4, o.k0), c, 0, 0, 0);
            acc[A2][st][i] = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(al, h, c, 0, 0, 0);
          }
        }
      };
      load_unit(std::integral_constant<int, 0>{}, ops[0]);
      static_for<NU>([&](auto u_) {
        constexpr int u = decltype(u_)::value;
        if constexpr (u + 1 < NU) load_unit(std::integral_constant<int, u + 1>{}, ops[(u + 1) & 1]);
        mfma_unit(u_, ops[u & 1]);
        __builtin_amdgcn_sched_barrier(0);
      });
      // more than one leftover slice per set (three channel groups and up): their own pass after the full slices
      if constexpr (NLS > 1) {
        static_for<NSET * NLS * NT>([&](auto q_) {
          constexpr int q = decltype(q_)::value, st = q / (NLS * NT), m = (q / NT) % NLS, i = q % NT;
          constexpr bool TWO = 2 * m + 1 < NCGS;
          const char* const pl = reinterpret_cast<const char*>(x3_lds) + (tbl[i] - lds_base) + sq + (st * NCGS + 2 * m) * CGB;
          const x3_bf16x8 wh = frag(std::integral_constant<int, st * NSLS + 3 * NCGS + m>{}, 0), wl = frag(std::integral_constant<int, st * NSLS + 3 * NCGS + m>{}, 1);
          const uint2 g0 = *reinterpret_cast<const uint2*>(pl);
          if constexpr (TWO) {
            const uint2 g1 = *reinterpret_cast<const uint2*>(pl + CGB);
            const x3_bf16x8 h = __builtin_bit_cast(x3_bf16x8, make_uint4(g0.x, g0.y, g1.x, g1.y));
            f32x4 c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wh, h, acc[A2][st][i], 0, 0, 0);
            if constexpr (!BF) {
              const uint2 k0 = *reinterpret_cast<const uint2*>(pl + LO_BYTES), k1 = *reinterpret_cast<const uint2*>(pl + CGB + LO_BYTES);
              c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wh, __builtin_bit_cast(x3_bf16x8, make_uint4(k0.x, k0.y, k1.x, k1.y)), c, 0, 0, 0);
            }
            acc[A2][st][i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wl, h, c, 0, 0, 0);
          } else {
            const x3_s16x4 ah = __builtin_bit_cast(x3_s16x4, __builtin_shufflevector(wh, wh, 0, 1, 2, 3));
            const x3_s16x4 al = __builtin_bit_cast(x3_s16x4, __builtin_shufflevector(wl, wl, 0, 1, 2, 3));
            const x3_s16x4 h = __builtin_bit_cast(x3_s16x4, g0);
            f32x4 c = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(ah, h, acc[A2][st][i], 0, 0, 0);
            if constexpr (!BF) {
              const uint2 k0 = *reinterpret_cast<const uint2*>(pl + LO_BYTES);
              c = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(ah, __builtin_bit_cast(x3_s16x4, k0), c, 0, 0, 0);
            }
            acc[A2][st][i] = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(al, h, c, 0, 0, 0);
          }
          if constexpr (i == NT - 1) __builtin_amdgcn_sched_barrier(0);
        });
      }
    };

    int s = zs - 1;
    while (true) {
      step(std::integral_constant<int, 0>{}, s); if (++s > ze) break;
      step(std::integral_constant<int, 1>{}, s); if (++s > ze) break;
      step(std::integral_constant<int, 2>{}, s); if (++s > ze) break;
    }
    // the last output plane ze-1 was completed by the last step (phase (ze - zs + 1) % 3, accumulator (phase + 2) % 3)
    switch ((ze - zs + 1) % 3) {
      case 0: epilogue(std::integral_constant<int, 2>{}, ze - 1, true); break;
      case 1: epilogue(std::integral_constant<int, 0>{}, ze - 1, true); break;
      default: epilogue(std::integral_constant<int, 1>{}, ze - 1, true); break;
    }
  }
}

// ---------------------------------------------------------------------------------------------------------------
int64_t x3_packed_words(int Cout, int Cin) {
  const int ncgs = (Cin + 3) / 4, nsls = x3_nsls(ncgs), ncog = (Cout + 15) / 16;
  return (int64_t)ncog * nsls * 2 * 64 * 4;
}

// the two sections of ragmi_conv3d_k3_pack_ex in one launch: workgroups [0, nb_k3) fill the fp32-MFMA section, the rest the
// bf16x3 fragments (a training step packs ~150 weights; each launch it does not make is ~3.5 us)
__global__ void pack_both_kernel(const float* __restrict__ w, float* __restrict__ packed, int64_t total_k3, int nb_k3, int Cout, int Cin,
                                 int nchunks, int nsls, int ncog, int transpose, int planar) {
  if ((int)blockIdx.x < nb_k3) {
    const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (idx < total_k3) packed[idx] = k3_pack_value(w, Cout, Cin, nchunks, idx, transpose, planar);
  } else {
    x3_pack_one(w, reinterpret_cast<uint4*>(packed + total_k3), Cout, Cin, nsls, ncog, transpose, planar,
                ((int)blockIdx.x - nb_k3) * 256 + threadIdx.x);
  }
}
int pack_both(const float* w, float* packed, int64_t total_k3, int Cout, int Cin, int transpose, int planar, hipStream_t s) {
  const int ncgs = (Cin + 3) / 4, nsls = x3_nsls(ncgs), ncog = (Cout + 15) / 16;
  const int nb_k3 = (int)ceil_div(total_k3, 256), nb_x3 = (int)ceil_div((int64_t)ncog * nsls * 64, 256);
  hipLaunchKernelGGL(pack_both_kernel, dim3((unsigned)(nb_k3 + nb_x3)), dim3(256), 0, s, w, packed, total_k3, nb_k3, Cout, Cin,
                     (Cin + CK - 1) / CK, nsls, ncog, transpose, planar);
  return RAGMI_OK;
}

// The bf16x3 form pays off on the big level-3 / level-6 volumes (z-marching columns need many (column, segment) work items to fill
// the chip) without a residual input and with equal-sized sets; everything else stays on the fp32-MFMA kernel.
// staging in 16-byte quads along x needs whole quads inside / outside the volume and aligned channel planes
static bool x3_quad_aligned(const K3Args& a, int dtype) {
  const size_t esz = dtype == RAGMI_BF16 ? 2 : 4;
  return a.W % 4 == 0 && a.x_bstride % 4 == 0 && (reinterpret_cast<uintptr_t>(a.x) & (4 * esz - 1)) == 0;
}

bool x3_eligible(const K3Args& a, int nset, int dtype) {
  // the caller asks for it through the dtype argument (include/rag_amd.h): RAGMI_F32 never comes here
  if ((dtype != RAGMI_F32X3 && dtype != RAGMI_BF16) || a.res != nullptr) return false;
  const int ncg = a.nchunks[0] + (nset == 2 ? a.nchunks[1] : 0);
  if (nset == 2 && (a.nchunks[0] != a.nchunks[1] || a.nchunks[0] > 2)) return false;
  if (nset == 1 && ncg > 6) return false;
  // six groups stage too many single voxels per thread for the scalar path: they need the quad path (W % 4 == 0, aligned planes)
  if (ncg == 6 && !x3_quad_aligned(a, dtype)) return false;
  if ((int64_t)a.B * a.D * a.H * a.W < X3_MIN_VOXELS || a.W < 32 || a.D < 8) return false;
  if (a.ntail > 0 && a.Cout > 16) return false;
  // buffer-descriptor offsets are 32-bit byte offsets inside one batch item (input, output and tail buffers alike); offsets
  // from 2^31 on are the "masked lane" range
  const int64_t plane4 = (int64_t)a.D * a.H * a.W * 4;
  if ((int64_t)(ncg * 4) * plane4 >= (1ll << 31)) return false;
  for (int g = 0; g < (a.Cout + 3) / 4; ++g)
    if ((int64_t)(a.y_ch[g] + 4) * plane4 >= (1ll << 30)) return false;
  for (int t = 0; t < a.ntail; ++t)
    if ((int64_t)(a.tail_ch0[t] + 4) * plane4 >= (1ll << 30)) return false;
  return true;
}

template <class T, int NCG, int NSET, bool TAILS, int NW, int ROWV, bool VEC>
static int x3_launch_final(const K3Args& a, const X3Extra& e, dim3 grid, size_t lds, hipStream_t st) {
  static LaunchState state;     // per device, mutex-guarded (common.h)
  if (!state.ensure_attr((const void*)conv3d_x3_kernel<T, NCG, NSET, TAILS, NW, ROWV, VEC>, 160 * 1024))
    return fail(RAGMI_ELAUNCH, "conv3d_x3: cannot raise the dynamic LDS limit");
  hipLaunchKernelGGL((conv3d_x3_kernel<T, NCG, NSET, TAILS, NW, ROWV, VEC>), grid, dim3(NW * 64), lds, st, a, e);
  return check_launch("conv3d_x3");
}


// workgroup shape per instantiation: 4 waves x 4 column tiles at <= 256 registers (two workgroups per CU = 2 waves per SIMD from
// different workgroups); rows of 48 records (conflict-free operand reads) when two workgroups then fit the LDS, else 40
constexpr size_t x3_lds_bytes_c(int ncg, int nset, int rowv, bool bf) {
  const int ncgs = ncg / nset, nsl = nset * x3_nsls(ncgs);
  return (size_t)(bf ? 1 : 2) * ncg * X3_RING * X3_HY * rowv * sizeof(uint2) + (bf ? 0 : X3_LO_PAD * sizeof(uint2)) + (size_t)nsl * 2 * 64 * sizeof(uint4) +
         80 * sizeof(float);
}
template <class T, int NCG, int NSET>
struct X3Shape {
  static constexpr bool BF = std::is_same<T, bf16_t>::value;
  static constexpr int NW = 4;
  static constexpr int ROWV = x3_lds_bytes_c(NCG, NSET, 48, BF) <= 80 * 1024 ? 48 : 40;
};

#ifndef RAGMI_X3_NO_DISPATCH
template <class T, int NCG, int NSET, bool TAILS>
static int x3_launch_vec(const K3Args& a, const X3Extra& e, bool vec, dim3 grid, hipStream_t st) {
  using S = X3Shape<T, NCG, NSET>;
  const size_t lds = x3_lds_bytes_c(NCG, NSET, S::ROWV, S::BF);
  if (vec) return x3_launch_final<T, NCG, NSET, TAILS, S::NW, S::ROWV, true>(a, e, grid, lds, st);
  if constexpr (NCG <= 5) return x3_launch_final<T, NCG, NSET, TAILS, S::NW, S::ROWV, false>(a, e, grid, lds, st);
  else return fail(RAGMI_EUNSUPPORTED, "conv3d_x3: %d channel groups need quad-aligned rows (x3_eligible excludes this)", NCG);
}
template <class T, int NCG, int NSET>
static int x3_launch_tails(const K3Args& a, const X3Extra& e, bool vec, dim3 grid, hipStream_t st) {
  return a.ntail > 0 ? x3_launch_vec<T, NCG, NSET, true>(a, e, vec, grid, st) : x3_launch_vec<T, NCG, NSET, false>(a, e, vec, grid, st);
}
template <int NCG, int NSET>
static int x3_launch_typed(const K3Args& a, const X3Extra& e, bool vec, dim3 grid, hipStream_t st) {
  return e.bf16 ? x3_launch_tails<bf16_t, NCG, NSET>(a, e, vec, grid, st) : x3_launch_tails<float, NCG, NSET>(a, e, vec, grid, st);
}
#endif

// launch geometry shared by the dispatcher and tools/x3_bench.hip: tiles, depth segments, buffer range; returns the grid
static int x3_prepare(K3Args& a, X3Extra& e, int nset, int dtype, int nseg_override, dim3& grid, bool& vec) {
  const int ngroups = (a.Cout + 3) / 4;
  for (int s = 0; s < nset; ++s)
    e.wf[s] = reinterpret_cast<const uint4*>(a.wp[s] + (int64_t)ngroups * a.nchunks[s] * PACK_PER_GC);
  a.tiles_x = (int)ceil_div(a.W, X3_TX); a.tiles_y = (int)ceil_div(a.H, X3_TY);
  const int ncog = (a.Cout + 15) / 16;
  const bool bf = dtype == RAGMI_BF16;
  const size_t esz = bf ? 2 : 4;
  e.bf16 = bf ? 1 : 0;
  e.xbytes = (unsigned)((int64_t)a.Cin * a.D * a.H * a.W * esz);
  vec = x3_quad_aligned(a, dtype);
  // depth segments: enough independent (column, segment) work items to fill the resident workgroups a few times over
  const int64_t cols = (int64_t)a.tiles_x * a.tiles_y * a.B;
  int nseg = (int)std::max<int64_t>(1, std::min<int64_t>(ceil_div(1536, cols * ncog), ceil_div(a.D, 8)));
  if (nseg_override > 0) nseg = nseg_override;
  e.seg_len = (int)ceil_div(a.D, nseg);
  e.nseg = (int)ceil_div(a.D, e.seg_len);
  const int64_t nwork = cols * e.nseg;
  RAGMI_REQUIRE(nwork < (1ll << 31), RAGMI_EUNSUPPORTED, "conv3d_x3: too many tiles");
  e.nwork = (int)nwork;
  grid = dim3((unsigned)std::min<int64_t>(nwork, 1024), ncog);
  return RAGMI_OK;
}

#ifndef RAGMI_X3_NO_DISPATCH
// a: as filled for the fp32 kernel (wp[s] = packed weights: fp32-MFMA section followed by the bf16x3 fragments)
int x3_launch(K3Args a, int nset, int dtype, hipStream_t st) {
  X3Extra e{};
  dim3 grid;
  bool vec = false;
  const int rc = x3_prepare(a, e, nset, dtype, 0, grid, vec);
  if (rc != RAGMI_OK) return rc;
  const int ncg = a.nchunks[0] + (nset == 2 ? a.nchunks[1] : 0);
  if (nset == 2) {
    switch (ncg) {
      case 2: return x3_launch_typed<2, 2>(a, e, vec, grid, st);
      case 4: return x3_launch_typed<4, 2>(a, e, vec, grid, st);
      default: return fail(RAGMI_EUNSUPPORTED, "conv3d_x3: dual form with %d channel groups not instantiated", ncg);
    }
  }
  switch (ncg) {
    case 1: return x3_launch_typed<1, 1>(a, e, vec, grid, st);
    case 2: return x3_launch_typed<2, 1>(a, e, vec, grid, st);
    case 3: return x3_launch_typed<3, 1>(a, e, vec, grid, st);
    case 4: return x3_launch_typed<4, 1>(a, e, vec, grid, st);
    case 5: return x3_launch_typed<5, 1>(a, e, vec, grid, st);
    case 6: return x3_launch_typed<6, 1>(a, e, vec, grid, st);
    default: return fail(RAGMI_EUNSUPPORTED, "conv3d_x3: %d channel groups not instantiated", ncg);
  }
}
#endif

}  // namespace ragmi
