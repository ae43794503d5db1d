// K3 (host side: weight pre-pack, C ABI, tile-configuration dispatch) — fused 3x3x3 ConvBR_3d (+ Cell_3d running sum / channel concat) on the CDNA4 matrix cores.
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
#include <cstdlib>

#include "conv3d_k3.h"

namespace ragmi {

// one pack element per thread: packed[((g*nchunks + ch)*NVG + v)*64 + lane]
// transpose: the source is the weight of the FORWARD conv, [Cin][Cout][taps]; pack its data-gradient conv
// W'[co][ci][tap] = W[ci][co][taps-1-tap].  planar: the source has 9 taps (a 2-D 3x3 weight) living in the dz = 1 plane.
// tile shape: widest x-tile whose padding waste is small; rows per lane sized so the grid still
// fills 256 CUs a few times over.  0: TX=32,R=4   1: TX=16,R=2   2: TX=8,R=1
static int choose_cfg(int B, int D, int H, int W, int Cout = 0) {
#ifdef RAGMI_DIAG
  static const int forced = [] { const char* v = getenv("RAGMI_K3_CFG"); return v ? atoi(v) : -1; }();      // tuning sweeps only
  if (forced >= 0) return forced;
#endif
  const int64_t vol = (int64_t)B * D * H * W;
  auto waste = [&](int tx) { return (double)(ceil_div(W, tx) * tx) / W; };
  if (W > 16 && waste(32) <= waste(16) + 1e-9 && vol >= (1 << 20)) return 0;
  // (round 4 sweep, tools/bench_deep.py under RAGMI_K3_CFG: the level-6 dual cell — 24 output channels on 2^18.7 voxels — runs 123.7 us
  // on the narrow tile against 129.3 on the middle one; the 8-channel launches of that volume prefer the middle one)
  if (W > 8 && waste(16) <= waste(8) + 1e-9 && vol >= (1 << 17) && !(Cout >= 20 && vol < (1 << 19))) return 1;
  return 2;
}
int fill_common(K3Args& a, const void* x, int64_t x_bstride, void* y, int64_t y_bstride,
                       const int32_t* y_group_ch, const void* res, int64_t res_bstride, const int32_t* res_group_ch,
                       int B, int Cin, int Cout, int D, int H, int W, int relu) {
  RAGMI_REQUIRE(B > 0 && Cin > 0 && Cout > 0 && D > 0 && H > 0 && W > 0, RAGMI_EINVAL, "conv3d_k3: non-positive size");
  const int ngroups = (Cout + 3) / 4;
  RAGMI_REQUIRE(ngroups <= RAGMI_MAX_GROUPS, RAGMI_EUNSUPPORTED, "conv3d_k3: Cout %d > %d", Cout, 4 * RAGMI_MAX_GROUPS);
  RAGMI_REQUIRE((int64_t)D * H * W < (1ll << 30), RAGMI_EUNSUPPORTED, "conv3d_k3: volume too large (32-bit plane offsets)");
  a.x = x; a.x_bstride = x_bstride;
  a.y = y; a.y_bstride = y_bstride;
  a.res = res; a.res_bstride = res_bstride;
  a.B = B; a.Cin = Cin; a.Cout = Cout; a.D = D; a.H = H; a.W = W; a.relu = relu;
  for (int g = 0; g < ngroups; ++g) {
    a.y_ch[g] = y_group_ch ? y_group_ch[g] : g * 4;
    a.res_ch[g] = res_group_ch ? res_group_ch[g] : g * 4;
  }
  return RAGMI_OK;
}

// host twin of lin_index for align_corners=True (same fp32 operations): every output o but the last of an axis must read the
// source pair (2o, 2o+1) — what the down-sampling tails of conv3d_x3.hip rely on (the last may clamp; the kernel honours it)
static bool down2_pairs_aligned(int in_size) {
  if (in_size < 2 || in_size % 2) return false;
  const int out_size = in_size / 2;
  const float scale = lin_scale(in_size, out_size, 1);
  for (int o = 0; o + 1 < out_size; ++o) {
    volatile float src = scale * (float)o;
    if ((int)src != 2 * o) return false;
  }
  volatile float last = scale * (float)(out_size - 1);
  const int i0 = std::min((int)last, in_size - 1);
  return i0 == in_size - 2 || i0 == in_size - 1;
}

int fill_tails(K3Args& a, int store_main, int ntail, const ragmi_tail_t* tails, int Cout) {
  a.store_main = 1;
  a.ntail = 0;
  a.ndown = 0;
  a.tail_g4 = 0;
  a.down_f32 = 0;
  if (ntail == 0 && store_main) return RAGMI_OK;
  RAGMI_REQUIRE(ntail >= 0 && ntail <= 4 && (ntail == 0 || tails != nullptr), RAGMI_EINVAL, "conv3d_k3: 0..4 tails");
  RAGMI_REQUIRE(ntail > 0 || store_main, RAGMI_EINVAL, "conv3d_k3: store_main = 0 needs at least one tail");
  const int ngroups = (Cout + 3) / 4;
  RAGMI_REQUIRE(ntail == 0 || (Cout % 4 == 0 && Cout <= 16 && split_groups(ngroups) == ngroups), RAGMI_EUNSUPPORTED,
                "conv3d_k3: fused tails need Cout in {4, 8, 12, 16} handled by one workgroup (got %d)", Cout);
  for (int t = 0; t < ntail; ++t) {
    RAGMI_REQUIRE(tails[t].weight && tails[t].y && tails[t].cout >= 1 && tails[t].cout <= 4 && tails[t].y_ch0 >= 0 &&
                      ((tails[t].scale == nullptr) == (tails[t].shift == nullptr)),
                  RAGMI_EINVAL, "conv3d_k3: bad tail %d (1..4 output channels)", t);
    if (tails[t].relu & 2) {                       // down-sampling tail (ragmi_tail_t.relu bit 1)
      const int k = a.ndown;
      RAGMI_REQUIRE(k < 2, RAGMI_EUNSUPPORTED, "conv3d_k3: at most two down-sampling tails");
      RAGMI_REQUIRE(down2_pairs_aligned(a.D) && down2_pairs_aligned(a.H) && down2_pairs_aligned(a.W), RAGMI_EUNSUPPORTED,
                    "conv3d_k3: a down-sampling tail needs even D, H, W whose x0.5 source pairs are aligned (got %d x %d x %d)", a.D, a.H, a.W);
      a.down_w[k] = (const float*)tails[t].weight; a.down_scale[k] = (const float*)tails[t].scale;
      a.down_shift[k] = (const float*)tails[t].shift; a.down_y[k] = tails[t].y;
      a.down_bstride[k] = tails[t].y_bstride; a.down_ch0[k] = tails[t].y_ch0; a.down_cout[k] = tails[t].cout;
      a.down_relu[k] = tails[t].relu & 1;
      if (tails[t].relu & RAGMI_TAIL_F32) a.down_f32 |= 1 << k;
      ++a.ndown;
      continue;
    }
    const int k = a.ntail;
    RAGMI_REQUIRE(k < 2, RAGMI_EUNSUPPORTED, "conv3d_k3: at most two full-resolution tails");
    RAGMI_REQUIRE(!(tails[t].relu & RAGMI_TAIL_F32), RAGMI_EINVAL, "conv3d_k3: RAGMI_TAIL_F32 is for down-sampling tails");
    const int g4 = (tails[t].relu & RAGMI_TAIL_G4) ? 1 : 0;
    RAGMI_REQUIRE(k == 0 || g4 == a.tail_g4, RAGMI_EUNSUPPORTED, "conv3d_k3: the full-resolution tails of a call share one layout");
    RAGMI_REQUIRE(!g4 || (tails[t].cout == 4 && tails[t].y_ch0 % 4 == 0), RAGMI_EINVAL, "conv3d_k3: a G4 tail has 4 output channels and a group-aligned y_ch0");
    a.tail_g4 = g4;
    a.tail_w[k] = (const float*)tails[t].weight; a.tail_scale[k] = (const float*)tails[t].scale;
    a.tail_shift[k] = (const float*)tails[t].shift; a.tail_y[k] = tails[t].y;
    a.tail_bstride[k] = tails[t].y_bstride; a.tail_ch0[k] = tails[t].y_ch0; a.tail_cout[k] = tails[t].cout;
    a.tail_relu[k] = tails[t].relu & 1;
    ++a.ntail;
  }
  a.store_main = store_main ? 1 : 0;
  return RAGMI_OK;
}

}  // namespace ragmi

// packed weights = [fp32-MFMA section: groups x chunks x PACK_PER_GC floats][bf16x3 fragments of conv3d_x3.hip]
static int64_t k3_section_elems(int Cout, int Cin) {
  return (int64_t)((Cout + 3) / 4) * ((Cin + ragmi::CK - 1) / ragmi::CK) * ragmi::PACK_PER_GC;
}
extern "C" int64_t ragmi_conv3d_k3_packed_elems(int Cout, int Cin) {
  if (Cout <= 0 || Cin <= 0) return 0;
  return k3_section_elems(Cout, Cin) + ragmi::x3_packed_words(Cout, Cin);
}

extern "C" int ragmi_conv3d_k3_pack(const void* weight, void* packed, int Cout, int Cin, int dtype, void* stream) {
  return ragmi_conv3d_k3_pack_ex(weight, packed, Cout, Cin, 0, 0, dtype, stream);
}

static int pack_sections(const void* weight, void* packed, int Cout, int Cin, int transpose, int planar2d, bool all, void* stream) {
  using namespace ragmi;
  RAGMI_REQUIRE(weight && packed, RAGMI_EINVAL, "conv3d_k3_pack: null pointer");
  RAGMI_REQUIRE(Cout > 0 && Cin > 0, RAGMI_EINVAL, "conv3d_k3_pack: non-positive size");
  RAGMI_REQUIRE(Cout <= 4 * RAGMI_MAX_GROUPS, RAGMI_EUNSUPPORTED, "conv3d_k3_pack: Cout %d > %d", Cout, 4 * RAGMI_MAX_GROUPS);
  const int64_t total = k3_section_elems(Cout, Cin);
  pack_both((const float*)weight, (float*)packed, total, Cout, Cin, transpose ? 1 : 0, planar2d ? 1 : 0, all, static_cast<hipStream_t>(stream));
  return check_launch("conv3d_k3_pack");
}

extern "C" int ragmi_conv3d_k3_pack_ex(const void* weight, void* packed, int Cout, int Cin, int transpose, int planar2d, int dtype,
                                       void* stream) {
  RAGMI_REQUIRE(dtype == RAGMI_F32, RAGMI_EUNSUPPORTED, "conv3d_k3_pack: dtype %d not built", dtype);
  return pack_sections(weight, packed, Cout, Cin, transpose, planar2d, true, stream);
}

extern "C" int ragmi_conv3d_k3_pack_for(const void* weight, void* packed, int Cout, int Cin, int transpose, int planar2d, int for_dtype,
                                        void* stream) {
  RAGMI_REQUIRE(ragmi::conv_dtype_ok(for_dtype), RAGMI_EUNSUPPORTED, "conv3d_k3_pack_for: dtype %d not built", for_dtype);
  return pack_sections(weight, packed, Cout, Cin, transpose, planar2d, for_dtype != RAGMI_F32, stream);
}

extern "C" int ragmi_conv3d_k3_fwd(const void* x, int64_t x_bstride, const void* packed_weight, const void* scale,
                                   const void* shift, int relu, void* y, int64_t y_bstride, const int32_t* y_group_ch,
                                   const void* res, int64_t res_bstride, const int32_t* res_group_ch, int B, int Cin,
                                   int Cout, int D, int H, int W, int dtype, void* stream) {
  return ragmi_conv3d_k3_fwd_ex(x, x_bstride, packed_weight, scale, shift, relu, y, y_bstride, y_group_ch, res, res_bstride,
                                res_group_ch, B, Cin, Cout, D, H, W, 1, 0, nullptr, dtype, stream);
}

extern "C" int ragmi_conv3d_k3_fwd_ex(const void* x, int64_t x_bstride, const void* packed_weight, const void* scale,
                                      const void* shift, int relu, void* y, int64_t y_bstride, const int32_t* y_group_ch,
                                      const void* res, int64_t res_bstride, const int32_t* res_group_ch, int B, int Cin,
                                      int Cout, int D, int H, int W, int store_main, int ntail, const ragmi_tail_t* tails,
                                      int dtype, void* stream) {
  using namespace ragmi;
  RAGMI_REQUIRE(x && packed_weight && y, RAGMI_EINVAL, "conv3d_k3: null pointer");
  RAGMI_REQUIRE((scale == nullptr) == (shift == nullptr), RAGMI_EINVAL, "conv3d_k3: scale/shift must both be given or both NULL");
  RAGMI_REQUIRE(conv_dtype_ok(dtype), RAGMI_EUNSUPPORTED, "conv3d_k3: dtype %d not built", dtype);
  K3Args a{};
  const int rc = fill_common(a, x, x_bstride, y, y_bstride, y_group_ch, res, res_bstride, res_group_ch, B, Cin, Cout, D, H, W, relu);
  if (rc != RAGMI_OK) return rc;
  a.wp[0] = (const float*)packed_weight; a.scale[0] = (const float*)scale; a.shift[0] = (const float*)shift;
  a.nchunks[0] = (Cin + CK - 1) / CK;
  const int rt = fill_tails(a, store_main, ntail, tails, Cout);
  if (rt != RAGMI_OK) return rt;
  hipStream_t s = static_cast<hipStream_t>(stream);
  const int ng = (Cout + 3) / 4;
  const bool bf = dtype == RAGMI_BF16;
  {
    const int want = ((relu & RAGMI_CONV_X_G4) ? 1 : 0) | (a.tail_g4 ? 2 : 0);
    RAGMI_REQUIRE(!(relu & RAGMI_CONV_Y_G4), RAGMI_EUNSUPPORTED, "conv3d_k3: the main output is channel planes (RAGMI_CONV_Y_G4 not built here)");
    RAGMI_REQUIRE((x3_g4_caps(a, 1, dtype) & want) == want, RAGMI_EUNSUPPORTED,
                  "conv3d_k3: this shape / dtype does not take G4 tensors (ragmi_conv3d_k3_g4_caps)");
  }
  if (x2d_eligible(a, 1, dtype)) return x2d_launch(a, 1, dtype, s);
  if (x3d_eligible(a, 1, dtype)) return x3d_launch(a, 1, dtype, s);
  if (x3_eligible(a, 1, dtype)) return x3_launch(a, 1, dtype, s);
  RAGMI_REQUIRE(a.ndown == 0, RAGMI_EUNSUPPORTED, "conv3d_k3: down-sampling tails need the z-marching split-operand form (ragmi_conv3d_k3_uses_x3)");
  switch (choose_cfg(B, D, H, W, Cout)) {
    case 0: return bf ? launch_k3_s1_cfg0_bf16(a, ng, s) : launch_k3_s1_cfg0_f32(a, ng, s);
    case 1: return bf ? launch_k3_s1_cfg1_bf16(a, ng, s) : launch_k3_s1_cfg1_f32(a, ng, s);
    default: return bf ? launch_k3_s1_cfg2_bf16(a, ng, s) : launch_k3_s1_cfg2_f32(a, ng, s);
  }
}

extern "C" int ragmi_conv3d_k3_small_fwd(const void* x, int64_t x_bstride, const void* weight, const void* scale,
                                         const void* shift, int relu, void* y, int64_t y_bstride, int y_ch0,
                                         const void* res, int64_t res_bstride, int res_ch0, int B, int Cin, int Cout,
                                         int D, int H, int W, int dtype, void* stream) {
  return ragmi_conv3d_k3_small_fwd_ex(x, x_bstride, weight, scale, shift, relu, y, y_bstride, y_ch0, res, res_bstride, res_ch0, B, Cin,
                                      Cout, D, H, W, dtype, dtype, stream);
}

extern "C" int ragmi_conv3d_k3_small_fwd_ex(const void* x, int64_t x_bstride, const void* weight, const void* scale,
                                            const void* shift, int relu, void* y, int64_t y_bstride, int y_ch0,
                                            const void* res, int64_t res_bstride, int res_ch0, int B, int Cin, int Cout,
                                            int D, int H, int W, int dtype, int y_dtype, void* stream) {
  using namespace ragmi;
  RAGMI_REQUIRE(y_dtype == dtype || (dtype == RAGMI_BF16 && y_dtype == RAGMI_F32), RAGMI_EUNSUPPORTED,
                "conv3d_k3_small: output dtype %d with input dtype %d not built (same, or fp32 out of bf16)", y_dtype, dtype);
  RAGMI_REQUIRE(x && weight && y, RAGMI_EINVAL, "conv3d_k3_small: null pointer");
  RAGMI_REQUIRE((scale == nullptr) == (shift == nullptr), RAGMI_EINVAL, "conv3d_k3_small: scale/shift must both be given or both NULL");
  RAGMI_REQUIRE(dtype_ok(dtype), RAGMI_EUNSUPPORTED, "conv3d_k3_small: dtype %d not built", dtype);
  RAGMI_REQUIRE(Cout >= 1 && Cout <= 2 && Cin % CK == 0 && (size_t)Cout * Cin * 36 * sizeof(float) <= (size_t)K3_MAX_WLDS_BYTES,
                RAGMI_EUNSUPPORTED, "conv3d_k3_small: needs Cout <= 2 and Cin a multiple of %d, <= %d (use ragmi_conv3d_k3_fwd otherwise)",
                CK, K3_MAX_WLDS_BYTES / (2 * 36 * 4));
  K3Args a{};
  const int32_t ych = y_ch0, rch = res_ch0;
  const int rc = fill_common(a, x, x_bstride, y, y_bstride, &ych, res, res_bstride, &rch, B, Cin, Cout, D, H, W, relu);
  if (rc != RAGMI_OK) return rc;
  a.wp[0] = (const float*)weight; a.scale[0] = (const float*)scale; a.shift[0] = (const float*)shift;
  a.nchunks[0] = Cin / CK;
  a.store_main = 1;
  if (c1_eligible(a, dtype, y_dtype)) return c1_launch(a, dtype, y_dtype, static_cast<hipStream_t>(stream));     // conv3d_c1.hip
  if (dtype == RAGMI_BF16 && y_dtype == RAGMI_F32) return launch_k3_valu_bf16_f32out(a, choose_cfg(B, D, H, W), static_cast<hipStream_t>(stream));
  return dtype == RAGMI_BF16 ? launch_k3_valu_bf16(a, choose_cfg(B, D, H, W), static_cast<hipStream_t>(stream))
                             : launch_k3_valu_f32(a, choose_cfg(B, D, H, W), static_cast<hipStream_t>(stream));
}

extern "C" int ragmi_conv3d_k3_dual_fwd(const void* x, int64_t x_bstride, int CinA, const void* packedA,
                                        const void* scaleA, const void* shiftA, int CinB, const void* packedB,
                                        const void* scaleB, const void* shiftB, int relu, void* y, int64_t y_bstride,
                                        const int32_t* y_group_ch, const void* res, int64_t res_bstride,
                                        const int32_t* res_group_ch, int B, int Cout, int D, int H, int W, int dtype,
                                        void* stream) {
  return ragmi_conv3d_k3_dual_fwd_ex(x, x_bstride, CinA, packedA, scaleA, shiftA, CinB, packedB, scaleB, shiftB, relu, y, y_bstride,
                                     y_group_ch, res, res_bstride, res_group_ch, B, Cout, D, H, W, 1, 0, nullptr, dtype, stream);
}

extern "C" int ragmi_conv3d_k3_dual_fwd_ex(const void* x, int64_t x_bstride, int CinA, const void* packedA,
                                           const void* scaleA, const void* shiftA, int CinB, const void* packedB,
                                           const void* scaleB, const void* shiftB, int relu, void* y, int64_t y_bstride,
                                           const int32_t* y_group_ch, const void* res, int64_t res_bstride,
                                           const int32_t* res_group_ch, int B, int Cout, int D, int H, int W,
                                           int store_main, int ntail, const ragmi_tail_t* tails, int dtype, void* stream) {
  using namespace ragmi;
  RAGMI_REQUIRE(x && packedA && packedB && y, RAGMI_EINVAL, "conv3d_k3_dual: null pointer");
  RAGMI_REQUIRE((scaleA == nullptr) == (shiftA == nullptr) && (scaleB == nullptr) == (shiftB == nullptr), RAGMI_EINVAL,
                "conv3d_k3_dual: scale/shift must both be given or both NULL");
  RAGMI_REQUIRE(conv_dtype_ok(dtype), RAGMI_EUNSUPPORTED, "conv3d_k3_dual: dtype %d not built", dtype);
  RAGMI_REQUIRE(CinA > 0 && CinB > 0 && CinA % CK == 0, RAGMI_EINVAL,
                "conv3d_k3_dual: CinA must be a positive multiple of %d (B's channels start on a chunk boundary)", CK);
  K3Args a{};
  const int rc = fill_common(a, x, x_bstride, y, y_bstride, y_group_ch, res, res_bstride, res_group_ch, B, CinA + CinB, Cout, D, H, W, relu);
  if (rc != RAGMI_OK) return rc;
  a.wp[0] = (const float*)packedA; a.scale[0] = (const float*)scaleA; a.shift[0] = (const float*)shiftA;
  a.wp[1] = (const float*)packedB; a.scale[1] = (const float*)scaleB; a.shift[1] = (const float*)shiftB;
  a.nchunks[0] = CinA / CK;
  a.nchunks[1] = (CinB + CK - 1) / CK;
  const int rt = fill_tails(a, store_main, ntail, tails, Cout);
  if (rt != RAGMI_OK) return rt;
  hipStream_t s = static_cast<hipStream_t>(stream);
  const int ng = (Cout + 3) / 4;
  const bool bf = dtype == RAGMI_BF16;
  {
    const int want = ((relu & RAGMI_CONV_X_G4) ? 1 : 0) | (a.tail_g4 ? 2 : 0);
    RAGMI_REQUIRE(!(relu & RAGMI_CONV_Y_G4), RAGMI_EUNSUPPORTED, "conv3d_k3_dual: the main output is channel planes (RAGMI_CONV_Y_G4 not built here)");
    RAGMI_REQUIRE((x3_g4_caps(a, 2, dtype) & want) == want, RAGMI_EUNSUPPORTED,
                  "conv3d_k3_dual: this shape / dtype does not take G4 tensors (ragmi_conv3d_k3_g4_caps)");
  }
  if (x2d_eligible(a, 2, dtype)) return x2d_launch(a, 2, dtype, s);
  if (x3d_eligible(a, 2, dtype)) return x3d_launch(a, 2, dtype, s);
  if (x3_eligible(a, 2, dtype)) return x3_launch(a, 2, dtype, s);
  RAGMI_REQUIRE(a.ndown == 0, RAGMI_EUNSUPPORTED, "conv3d_k3_dual: down-sampling tails need the z-marching split-operand form (ragmi_conv3d_k3_uses_x3)");
  switch (choose_cfg(B, D, H, W, Cout)) {
    case 0: return bf ? launch_k3_s2_cfg0_bf16(a, ng, s) : launch_k3_s2_cfg0_f32(a, ng, s);
    case 1: return bf ? launch_k3_s2_cfg1_bf16(a, ng, s) : launch_k3_s2_cfg1_f32(a, ng, s);
    default: return bf ? launch_k3_s2_cfg2_bf16(a, ng, s) : launch_k3_s2_cfg2_f32(a, ng, s);
  }
}

extern "C" int ragmi_down2_tail_supported(int D, int H, int W) {
  return (ragmi::down2_pairs_aligned(D) && ragmi::down2_pairs_aligned(H) && ragmi::down2_pairs_aligned(W)) ? 1 : 0;
}

extern "C" int ragmi_conv3d_k3_uses_x3(int Cin, int Cout, int B, int D, int H, int W, int nset, int has_res, int ntail, int dtype) {
  using namespace ragmi;
  if (Cin <= 0 || Cout <= 0 || B <= 0 || D <= 0 || H <= 0 || W <= 0 || (nset != 1 && nset != 2)) return 0;
  K3Args a{};
  a.B = B; a.Cin = Cin; a.Cout = Cout; a.D = D; a.H = H; a.W = W; a.ntail = ntail;
  a.res = has_res ? (const void*)&a : nullptr;
  if (nset == 2) { if (Cin % (2 * CK)) return 0; a.nchunks[0] = a.nchunks[1] = Cin / (2 * CK); }
  else a.nchunks[0] = (Cin + CK - 1) / CK;
  a.store_main = 1;
  return (x2d_eligible(a, nset, dtype) || x3d_eligible(a, nset, dtype) || x3_eligible(a, nset, dtype)) ? 1 : 0;
}

extern "C" int ragmi_conv3d_k3_g4_caps(int Cin, int Cout, int B, int D, int H, int W, int nset, int ntail, int ndown, int dtype) {
  using namespace ragmi;
  if (Cin <= 0 || Cout <= 0 || B <= 0 || D <= 0 || H <= 0 || W <= 0 || (nset != 1 && nset != 2)) return 0;
  K3Args a{};
  a.B = B; a.Cin = Cin; a.Cout = Cout; a.D = D; a.H = H; a.W = W; a.ntail = ntail; a.ndown = ndown;
  for (int t = 0; t < 2; ++t) a.tail_cout[t] = a.down_cout[t] = 4;
  for (int g = 0; g < (Cout + 3) / 4 && g < RAGMI_MAX_GROUPS; ++g) a.y_ch[g] = 4 * g;
  if (nset == 2) { if (Cin % (2 * CK)) return 0; a.nchunks[0] = a.nchunks[1] = Cin / (2 * CK); }
  else a.nchunks[0] = (Cin + CK - 1) / CK;
  a.store_main = 1;
  return x3_g4_caps(a, nset, dtype);
}

extern "C" int ragmi_conv3d_k3_plan(int Cout, int B, int D, int H, int W, int nset, int32_t* log_tx,
                                    int32_t* rows_per_lane, int32_t* launch_groups, int32_t max_launches) {
  using namespace ragmi;
  RAGMI_REQUIRE(log_tx && rows_per_lane && launch_groups, RAGMI_EINVAL, "conv3d_k3_plan: null pointer");
  RAGMI_REQUIRE(Cout > 0 && B > 0 && D > 0 && H > 0 && W > 0 && max_launches > 0 && (nset == 1 || nset == 2), RAGMI_EINVAL,
                "conv3d_k3_plan: bad argument");
  static const int cfgs[3][2] = {{5, 4}, {4, 2}, {3, 1}};
  const int c = choose_cfg(B, D, H, W, Cout);
  *log_tx = cfgs[c][0];
  *rows_per_lane = (nset == 2 && c == 0) ? 2 : cfgs[c][1];   // the level-3 dual kernel uses 2 rows/lane (register budget)
  launch_groups[0] = split_groups((Cout + 3) / 4);   // one launch; blockIdx.y covers ngroups / G splits
  return 1;
}
