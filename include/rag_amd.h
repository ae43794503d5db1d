/*
 * rag_amd.h — C ABI of librag_amd.so: the MI355X (gfx950) kernels behind the
 * RAG stereo Matching-Net forward path.
 *
 * The reference (chzhang18/RAG) is pure PyTorch and offers no FFI; its "plugin
 * API" for this path is the nn.Module composition in src/models/rag_model.py.
 * Each entry point below replaces the ATen op sequence of one reference
 * construct (cited per function); rag_amd/modules.py mirrors the reference's
 * module names/signatures on top of these calls.
 *
 * Conventions:
 *  - plain pointers and sizes, no torch types; all tensors are device pointers
 *    owned by the caller (PyTorch allocates), contiguous N-C-D-H-W planes;
 *    "bstride" arguments are batch strides in ELEMENTS so that a call may read
 *    or write a channel slice of a wider buffer (this is how torch.cat and the
 *    running sum in Cell_3d are folded into the producing kernels);
 *  - every function returns 0 on success and a negative RAGMI_E* code on
 *    failure; no C++ exception crosses the ABI; ragmi_last_error() gives a
 *    thread-local message for the last failure on this thread;
 *  - stateless and re-entrant; kernels are enqueued on the hipStream_t passed
 *    as `stream` (void*), never synchronise, never allocate device memory;
 *  - dtype selects activation storage AND the arithmetic contract of the call (no environment variable changes either):
 *      RAGMI_F32    fp32 storage, fp32 arithmetic: the contractions run on the fp32-input MFMA forms, bitwise an fmaf chain
 *                   per output (one rounding per product, fp32 accumulate);
 *      RAGMI_F32X3  fp32 storage; accepted by the 3x3x3 convolution entry points only (ragmi_conv3d_k3_fwd(_ex),
 *                   ragmi_conv3d_k3_dual_fwd(_ex)).  On shapes for which ragmi_conv3d_k3_uses_x3() answers 1 every fp32
 *                   operand is scaled by a power of two and split into two FP16 halves, a * 2^s = hi + lo + r with hi = fp16(a 2^s),
 *                   lo = fp16(a 2^s - hi), |lo| <= 2^-11 |hi|, and a product a*b is accumulated in fp32 as hi_a*hi_b + hi_a*lo_b +
 *                   lo_a*hi_b on the fp16 matrix cores (16x the fp32 MFMA rate).  The scales are exact (powers of two) and chosen by
 *                   the library: per output channel for the weights (largest |w| of the channel -> [2^9, 2^10]), per workgroup
 *                   tile for the activations (largest |x| of the tile's halo planes -> at most 2^15, 2^10..2^11 when chosen; a plane
 *                   that would overflow makes the tile restart with a larger scale — fp16's range is never exceeded).  Dropped per
 *                   product: lo_a*lo_b and the roundings of the lo halves, each <= 2^-22 |a b|, plus — fp16 has no exponents below
 *                   2^-24 — an ABSOLUTE 2^-25 of the scaled operand, i.e. 2^-35 of the tile's largest |x| (2^-34 of the channel's
 *                   largest |w|).  Per output:
 *                        |y_x3 - y_exact| <= 2^-20 * sum_k |w_k x_k| + 2^-33 * (Xmax * sum_k |w_k| + Wmax * sum_k |x_k|)
 *                                            + fp32 accumulation error
 *                   with Xmax the largest |x| in the workgroup's tile and Wmax the largest |w| of the output channel.  On
 *                   activations of one magnitude class (anything a network produces) the second term vanishes and the error is
 *                   ~1e-7 * sum |w x| (measured 9e-8 on N(0,1) data with a common offset of 1000: the class of RAGMI_F32 itself, 1.1e-7);
 *                   it matters only when operands ~2^-25 smaller than their tile's largest carry the sum (measured 1e-3 * sum |w x|
 *                   on operands spread element by element over 2^-20..2^20).  The bound is relative to the sum of |products|, not to
 *                   |y|: cancelling sums lose that many ABSOLUTE digits, as in fp32.  Non-finite inputs give non-finite outputs (an
 *                   Inf may come out as NaN); so do operands above ~2^110 in magnitude, which no scale brings into fp16's range.  Round 2 used bf16 halves (8 bits each, ~2^-17 per product): measured over weight
 *                   seeds at the headline size that left the EPE against the CPU reference between 1e-5 and 1.4e-3 px; the fp16
 *                   halves sit on the strict fp32 path's EPE for every seed (tests/test_hip_parity.py::
 *                   test_x3_margin_over_seeds_and_genotypes_at_headline_size).
 *                   Elsewhere (small volumes, a residual input, unsupported channel counts) the call is computed exactly as
 *                   RAGMI_F32.  tests/test_hip_parity.py::test_x3_error_bound_adversarial enforces the bound.
 *      RAGMI_BF16   bf16 activation storage (BASELINE config 3), fp32 on-chip accumulation, BN and tails; the contraction may run
 *                   on the bf16 matrix cores with the (exact) bf16 activations and hi+lo split fp32 weights — an error of
 *                   <= 2^-16 per weight, far below the 2^-8 of the storage format.
 *    Weights, scale/shift and the final disparity are fp32 in every case.
 */
#ifndef RAG_AMD_H
#define RAG_AMD_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define RAGMI_OK 0
#define RAGMI_EINVAL (-1)       /* bad argument (null pointer, non-positive size, ...) */
#define RAGMI_EUNSUPPORTED (-2) /* shape / dtype combination not built */
#define RAGMI_ELAUNCH (-3)      /* hipLaunch / runtime error */

#define RAGMI_F32 0
#define RAGMI_BF16 1 /* activations stored as bf16; fp32 on chip (LDS, MFMA, accumulate, BN); weights/BN stay fp32 */
#define RAGMI_F32X3 2 /* fp32 storage, scaled-fp16 split products (hi*hi + hi*lo + lo*hi) in the eligible 3x3x3 convolutions (see the dtype note above) */

#define RAGMI_MAX_GROUPS 16 /* output-channel groups of 4 per conv call */

/* Library version (major*10000 + minor*100 + patch) and last error text. */
int ragmi_version(void);
const char* ragmi_last_error(void);

/*
 * Cost-volume build.  Replaces the inline slice-copy loop of
 * src/models/rag_model.py:375-383 (dups :694-702, src/automl/mdenas_basicmodel.py:83-91):
 *   cost[b, c,   i, y, x] = L[b, c, y, x]      if x >= i else 0
 *   cost[b, C+c, i, y, x] = R[b, c, y, x - i]  if x >= i else 0,      i in [0, d)
 * L, R: [B, C, h, w]; cost: [B, 2C, d, h, w] (fully written, no memset needed).
 */
int ragmi_costvol_fwd(const void* left_fea, const void* right_fea, void* cost,
                      int B, int C, int d, int h, int w, int dtype, void* stream);

/*
 * Weight pre-pack for ragmi_conv3d_k3_fwd: reorders an nn.Conv3d weight
 * [Cout, Cin, 3, 3, 3] (src/automl/operations_3d.py:37), Cout <= 64, into the operand
 * fragments of the matrix-core kernels.  `packed` must hold
 * ragmi_conv3d_k3_packed_elems(Cout, Cin) floats: the per-lane broadcast fragments of
 * v_mfma_f32_4x4x1_16b_f32 (RAGMI_F32), then the bf16 hi / lo fragments (RAGMI_BF16),
 * then the fp16 hi / lo fragments of w * 2^k[co] and the per-output-channel
 * multipliers 2^-k[co] (RAGMI_F32X3).  One pack serves every dtype of the convolution
 * entry points (ragmi_conv3d_k3_pack_for fills only what one contract reads).
 */
int64_t ragmi_conv3d_k3_packed_elems(int Cout, int Cin);
int ragmi_conv3d_k3_pack(const void* weight, void* packed, int Cout, int Cin, int dtype, void* stream);

/*
 * Fused ConvBR_3d, 3x3x3 / stride 1 / pad 1 (src/automl/operations_3d.py:31-47),
 * plus the running sum and channel concat of Cell_3d (src/models/rag_model.py:160-176):
 *   v    = conv3d(x, W)[co]                      (fp32, bias-free)
 *   v    = v * scale[co] + shift[co]             (folded eval BatchNorm3d; NULL scale => skip)
 *   v    = max(v, 0)                  if relu
 *   v   += res[b, res_ch(co), ...]    if res != NULL   (res may alias y: accumulate in place)
 *   y[b, y_ch(co), z, y, x] = v
 * Output channel co belongs to group g = co/4; y_ch(co) = y_group_ch[g] + co%4 when
 * y_group_ch != NULL (host array of ceil(Cout/4) ints), else co; same for res_group_ch.
 * Sibling convolutions that read the same input are fused by concatenating their
 * weights along Cout and pointing each group at its own destination channels.
 * x: [B, Cin, D, H, W] with batch stride x_bstride; y/res likewise.
 */
int ragmi_conv3d_k3_fwd(const void* x, int64_t x_bstride,
                        const void* packed_weight, const void* scale, const void* shift, int relu,
                        void* y, int64_t y_bstride, const int32_t* y_group_ch,
                        const void* res, int64_t res_bstride, const int32_t* res_group_ch,
                        int B, int Cin, int Cout, int D, int H, int W,
                        int dtype, void* stream);

/*
 * Consumer 1x1x1 ConvBR_3d fused into the producing 3x3x3 kernel's epilogue (a "tail"):
 *   y_t[b, y_ch0 + j] = act(bn_t(sum_c W_t[j][c] * out[b, c]))       j < cout <= 4
 * where out is the (activated, summed) result of the 3x3x3 call it is attached to.  This is how the level-3
 * Cell_3d.pre_preprocess / preprocess convs (rag_model.py:125-126, 154-155) of the NEXT cells are computed by the
 * stem / cell that produces their input, instead of as separate HBM-bound passes.  weight: raw [cout][Cout_main].
 */
typedef struct {
  const void* weight;
  const void* scale;   /* folded BN of the tail, or NULL/NULL */
  const void* shift;
  int32_t relu;
  void* y;
  int64_t y_bstride;
  int32_t y_ch0;
  int32_t cout;
} ragmi_tail_t;
/* `relu` bit 0: ReLU after the BatchNorm.  `relu` bit 1 (value 2): a DOWN-SAMPLING tail — the 1x1x1 ConvBR_3d of a consumer cell that
 * works one level down (Cell_3d with downup_sample = -1: F.interpolate(x, half size, 'trilinear', align_corners=True) followed by
 * pre_preprocess / preprocess, rag_model.py:146-155), computed conv-first in the producer's epilogue: y is [B, *, D/2, H/2, W/2].
 * Up to two of them (4 output channels each) beside up to two full-resolution tails; only the z-marching split-operand form takes
 * them (ragmi_conv3d_k3_uses_x3 with RAGMI_F32X3 or RAGMI_BF16, <= 12 input channels) and only for even D, H, W whose x0.5 source
 * pairs are aligned (ragmi_down2_tail_supported: output o reads inputs (2o, 2o+1) on every axis; the last may clamp). */
int ragmi_down2_tail_supported(int D, int H, int W);

/* Channel-group-interleaved tensors ("G4"): [B][C/4][D][H][W][4] fp32 — the four channels of a group of a voxel are 16 contiguous
 * bytes — instead of channel planes [B][C][D][H][W].  A PRIVATE layout of the fused executor (rag_amd.MatchingNet._run_chain) for
 * the level-3 tensors its own kernels exchange (the 8-channel s0|s1 inputs of the dual cells, stem3d0's output): every module
 * boundary of the reference (rag_model.py:143-177, 325-366) keeps channel planes.  Why: on gfx950 the vector memory path costs per
 * INSTRUCTION; in G4 a halo voxel of a group is one 16-byte load and a fused tail one 16-byte store instead of four 4-byte ones into
 * four planes (profiles/r05_x3_stamps.md).  Flags (fp32 storage, or bf16 storage: four bf16 = one 8-byte access; RAGMI_EUNSUPPORTED where the kernel a call lands on does not
 * take them — ask ragmi_conv3d_k3_g4_caps first):
 *   `relu` argument of ragmi_conv3d_k3_fwd_ex / _dual_fwd_ex:  bit 1 (RAGMI_CONV_X_G4): x is G4 (x_bstride in floats, as ever);
 *   ragmi_tail_t.relu bit 2 (RAGMI_TAIL_G4): the tail's destination y is G4; y_ch0 (a multiple of 4) names the group y_ch0 / 4; the
 *                  tail must have exactly 4 output channels.  All full-resolution tails of a call agree; down-sampling tails are planes.
 *   `relu` argument of ragmi_costvol_stem_fwd: bit 2 (RAGMI_CONV_Y_G4): y (Cout a multiple of 4) is G4; its tails as above. */
#define RAGMI_CONV_RELU 1
#define RAGMI_CONV_X_G4 2
#define RAGMI_CONV_Y_G4 4
#define RAGMI_TAIL_G4 4
/* Mixed storage (round 5, BASELINE configs[2]): bf16 storage is kept for the full-resolution (level-3) tensors only — the deep levels and
 * the head are fp32 (tests/analysis_bf16_stage_epe.py: they carry most of the bf16 error and almost none of the bytes).  Two edges cross:
 *   ragmi_tail_t.relu bit 3 (RAGMI_TAIL_F32): the destination of this DOWN-SAMPLING tail is fp32 (y_bstride in floats) although the
 *                  call's storage is RAGMI_BF16 (ignored under fp32 storage; RAGMI_EINVAL on a full-resolution tail);
 *   dtype RAGMI_BF16 | RAGMI_OUT_F32 of ragmi_conv3d_k1_resample_fwd: x is bf16, y is fp32 (same arithmetic: fp32 on chip). */
#define RAGMI_TAIL_F32 8
/* ragmi_costvol_stem_conv3d_fwd only, tails0[0].relu bit 4 (RAGMI_TAIL_ROWS): this tail (4 output channels on stem3d0's 12) is NOT evaluated
 * from `weight` by the staging thread but falls out of stem3d1's matrix product: the caller packed stem3d1's weight as a SIXTEEN-channel
 * convolution whose rows 12..15 hold the tail's weights at the centre tap (zero elsewhere) — rows the 12-channel product leaves idle.  Cout
 * stays 12; the tail's scale / shift / relu / y are used as ever, its products are split-operand ones like the convolution's (the
 * RAGMI_F32X3 bound) instead of an exact fp32 chain. */
#define RAGMI_TAIL_ROWS 16
#define RAGMI_OUT_F32 0x100
/* bit mask: 1 = this call accepts a G4 input, 2 = it can write G4 full-resolution tails (arguments as ragmi_conv3d_k3_uses_x3) */
int ragmi_conv3d_k3_g4_caps(int Cin, int Cout, int B, int D, int H, int W, int nset, int ntail, int ndown, int dtype);

/*
 * ragmi_conv3d_k3_fwd / ragmi_conv3d_k3_dual_fwd with up to two full-resolution tails (plus up to two down-sampling ones).  store_main = 0 skips writing the 3x3x3
 * result itself (only the tails consume it).  Tails need Cout in {4, 8, 12, 16} (all channels in one workgroup).
 */
int ragmi_conv3d_k3_fwd_ex(const void* x, int64_t x_bstride,
                           const void* packed_weight, const void* scale, const void* shift, int relu,
                           void* y, int64_t y_bstride, const int32_t* y_group_ch,
                           const void* res, int64_t res_bstride, const int32_t* res_group_ch,
                           int B, int Cin, int Cout, int D, int H, int W,
                           int store_main, int ntail, const ragmi_tail_t* tails, int dtype, void* stream);
int ragmi_conv3d_k3_dual_fwd_ex(const void* x, int64_t x_bstride,
                                int CinA, const void* packedA, const void* scaleA, const void* shiftA,
                                int CinB, const void* packedB, const void* scaleB, const void* shiftB,
                                int relu, void* y, int64_t y_bstride, const int32_t* y_group_ch,
                                const void* res, int64_t res_bstride, const int32_t* res_group_ch,
                                int B, int Cout, int D, int H, int W,
                                int store_main, int ntail, const ragmi_tail_t* tails, int dtype, void* stream);

/*
 * Same operation as ragmi_conv3d_k3_fwd for Cout <= 2 (last_3_3d: 12 -> 1, rag_model.py:269), computed with
 * v_fma and wave-uniform weights instead of MFMA (a 4-row MFMA tile would idle 3 rows at Cout = 1).
 * `weight` is the RAW nn.Conv3d weight [Cout, Cin, 3, 3, 3]; Cin must be a multiple of 4.
 * Writes y[b, y_ch0 + co]; res (optional) is read at res_ch0 + co.
 */
int ragmi_conv3d_k3_small_fwd(const void* x, int64_t x_bstride, const void* weight, const void* scale,
                              const void* shift, int relu, void* y, int64_t y_bstride, int y_ch0,
                              const void* res, int64_t res_bstride, int res_ch0,
                              int B, int Cin, int Cout, int D, int H, int W, int dtype, void* stream);

/* The same with the output stored in its own dtype: y_dtype == dtype, or RAGMI_F32 out of RAGMI_BF16 input — the head's `mat`
 * (rag_model.py:365), which the soft-argmin consumes at |cost| ~ 1e4, stays fp32 under bf16 activation storage.
 * res, when given, has the INPUT dtype. */
int ragmi_conv3d_k3_small_fwd_ex(const void* x, int64_t x_bstride, const void* weight, const void* scale,
                                 const void* shift, int relu, void* y, int64_t y_bstride, int y_ch0,
                                 const void* res, int64_t res_bstride, int res_ch0,
                                 int B, int Cin, int Cout, int D, int H, int W, int dtype, int y_dtype, void* stream);

/*
 * Two sibling ConvBR_3d groups fused into one launch (Cell_3d with two conv branches per new
 * state, rag_model.py:160-172):
 *   y[b, y_ch(co)] = act(bnA(convA(x[:, 0:CinA])))[co] + act(bnB(convB(x[:, CinA:CinA+CinB])))[co] (+ res)
 * x holds both inputs as consecutive channels (CinA a multiple of 4); packedA / packedB are
 * ragmi_conv3d_k3_pack outputs for [Cout, CinA, 3,3,3] and [Cout, CinB, 3,3,3].  Other arguments
 * as ragmi_conv3d_k3_fwd.  The running sum never touches HBM.
 */
int ragmi_conv3d_k3_dual_fwd(const void* x, int64_t x_bstride,
                             int CinA, const void* packedA, const void* scaleA, const void* shiftA,
                             int CinB, const void* packedB, const void* scaleB, const void* shiftB,
                             int relu, void* y, int64_t y_bstride, const int32_t* y_group_ch,
                             const void* res, int64_t res_bstride, const int32_t* res_group_ch,
                             int B, int Cout, int D, int H, int W, int dtype, void* stream);

/*
 * Introspection for profiling: which kernel instantiation(s) ragmi_conv3d_k3_fwd will launch
 * for this shape (nset = 1, or 2 for ragmi_conv3d_k3_dual_fwd).  Writes the x-tile log2 width and rows per lane, and the output groups per
 * workgroup (G) into launch_groups[0]; returns the number of launches (1) or a negative error
 * code.  Kernel name: conv3d_k3_kernel<G, log_tx, rows_per_lane, NSET> (NSET 1, or 2 for _dual).
 */
int ragmi_conv3d_k3_plan(int Cout, int B, int D, int H, int W, int nset, int32_t* log_tx,
                         int32_t* rows_per_lane, int32_t* launch_groups, int32_t max_launches);

/*
 * Fused ConvBR_3d, 1x1x1 (Cell_3d.pre_preprocess / preprocess, rag_model.py:125-126,
 * last_6_3d / last_12_3d :270-271): channel mix + folded BN + ReLU, HBM-bound.
 * weight: raw nn.Conv3d weight [Cout, Cin] (1x1x1 squeezed), row-major.
 * Writes y[b, y_ch0 + co, ...].
 */
int ragmi_conv3d_k1_fwd(const void* x, int64_t x_bstride,
                        const void* weight, const void* scale, const void* shift, int relu,
                        void* y, int64_t y_bstride, int y_ch0,
                        int B, int Cin, int Cout, int64_t DHW,
                        int dtype, void* stream);

/* Two 1x1x1 ConvBR_3d in a row as one launch: y = act2(bn2(W2 . act1(bn1(W1 . x)))), the intermediate (Cmid channels) in registers —
 * the head's last_12_3d followed by the channel mix of last_6_3d (rag_model.py:358-365, :270-271).  Results are those of two
 * ragmi_conv3d_k1_fwd calls, bit for bit.  Built for Cin <= 64 -> 24 -> 12 channels (ragmi_conv3d_k1_chain_supported). */
int ragmi_conv3d_k1_chain_supported(int Cin, int Cmid, int Cout);
int ragmi_conv3d_k1_chain_fwd(const void* x, int64_t x_bstride, const void* weight1, const void* scale1, const void* shift1, int relu1,
                              int Cmid, const void* weight2, const void* scale2, const void* shift2, int relu2,
                              void* y, int64_t y_bstride, int y_ch0, int B, int Cin, int Cout, int64_t DHW, int dtype, void* stream);

/* The same with the weight given TRANSPOSED in memory when w_transposed != 0 (weight[ci][co], i.e. [Cin][Cout] row-major): the
 * data gradient of a 1x1x1 conv is this conv with the forward weight read in place — no transposed copy per step. */
int ragmi_conv3d_k1_fwd_ex(const void* x, int64_t x_bstride, const void* weight, int w_transposed, const void* scale,
                           const void* shift, int relu, void* y, int64_t y_bstride, int y_ch0, int B, int Cin,
                           int Cout, int64_t DHW, int dtype, void* stream);

/*
 * Trilinear resample fused into the 1x1x1 ConvBR_3d that consumes it:
 *   y[b, y_ch0 + co] = act(bn(W . interpolate(x, [Do,Ho,Wo], 'trilinear', align_corners)))
 * (Cell_3d down/up-sampling + preprocess, rag_model.py:146-155; head last_6_3d(upsample_12(.)), :358-365).
 * x: [B, Cin, Di, Hi, Wi]; the interpolated tensor is never materialised.
 */
int ragmi_conv3d_k1_resample_fwd(const void* x, int64_t x_bstride, int Di, int Hi, int Wi,
                                 const void* weight, const void* scale, const void* shift, int relu,
                                 void* y, int64_t y_bstride, int y_ch0,
                                 int B, int Cin, int Cout, int Do, int Ho, int Wo, int align_corners,
                                 int dtype, void* stream);

/*
 * Two such resample + 1x1x1 ConvBR_3d calls that write into the same buffer at the same output size (a cell's
 * pre_preprocess and preprocess, each reading its own input at its own size) as ONE launch.
 */
typedef struct {
  const void* x;
  int64_t x_bstride;
  int32_t Di, Hi, Wi;
  const void* weight; /* [Cout][Cin] */
  const void* scale;
  const void* shift;
  int32_t relu;
  int32_t y_ch0;
  int32_t Cin, Cout;
} ragmi_k1r_t;
int ragmi_conv3d_k1_resample_pair_fwd(const ragmi_k1r_t* a, const ragmi_k1r_t* b, void* y, int64_t y_bstride,
                                      int B, int Do, int Ho, int Wo, int align_corners, int dtype, void* stream);
/* the same with up to three descriptors and ONE DESTINATION BUFFER PER DESCRIPTOR (ys[i], y_bstrides[i]; y_ch0 inside it), all at
 * the output size (Do, Ho, Wo): a cell's pre_preprocess + preprocess together with the NEXT cell's pre_preprocess when that one
 * resamples the same tensor to the same size (rag_model.py:146-155 of two consecutive cells: the level-6 tensor both level-12
 * cells 6 and 7 read) — the tensor is gathered by both in one launch instead of once per launch. */
int ragmi_conv3d_k1_resample_multi_fwd(const ragmi_k1r_t* const* specs, void* const* ys, const int64_t* y_bstrides, int n, int B,
                                       int Do, int Ho, int Wo, int align_corners, int dtype, void* stream);

/*
 * One launch per Cell_2d of the Feature Net (src/models/rag_model.py:143-177 with the 2-D operations of
 * src/automl/operations_2d.py; the cells built by rag_model.py:236-247), for cells in which every new state is the sum of one
 * 3x3 ConvBR_2d of s0 and one of s1 (the all-conv genotype):
 *     s0 = pre_preprocess(F.interpolate(prev_prev, (H, W), mode='bilinear', align_corners=True))      (1x1 ConvBR_2d)
 *     s1 = preprocess(F.interpolate(prev, (H, W), ...))                                                (1x1 ConvBR_2d)
 *     y[:, group g] = relu(bnA(convA(s0))) + relu(bnB(convB(s1)))          (the stacked sibling convs, as ragmi_conv3d_k3_dual_fwd)
 * s0 / s1 and the interpolated tensors are never written: the two 1x1 convs (resample first, then an fmaf chain over the input
 * channels in order, folded BatchNorm, ReLU — the arithmetic of ragmi_conv3d_k1_resample_fwd) run in the staging of the 3x3 launch.
 * An input already at (H, W) is read as is.  packedA / packedB: ragmi_conv3d_k3_pack(_ex) of the [Cout, C, 3, 3] weights
 * (planar2d) or of any [Cout, C, 3, 3, 3] weight (depth 1: only the middle slice acts).  dtype: RAGMI_F32X3 only (fp32 storage,
 * split products — the contract of ragmi_conv3d_k3_fwd on depth-1 volumes); C = 4 or 8 channels per set, Cin <= 48 per input,
 * W >= 16: ragmi_cell2d_supported answers 1 for what is built, and the host runs the separate launches otherwise.
 */
typedef struct {
  const void* x; /* [B, Cin, Hi, Wi] */
  int64_t x_bstride;
  int32_t Cin, Hi, Wi;
  const void* weight; /* [C][Cin] */
  const void* scale;  /* folded BatchNorm [C], or NULL */
  const void* shift;
  int32_t relu;
} ragmi_cell2d_in_t;
int ragmi_cell2d_supported(int C, int Cin0, int Cin1, int Cout, int H, int W, int dtype);
int ragmi_cell2d_fwd(const ragmi_cell2d_in_t* s0, const ragmi_cell2d_in_t* s1, int C, const void* packedA, const void* scaleA,
                     const void* shiftA, const void* packedB, const void* scaleB, const void* shiftB, int relu, void* y,
                     int64_t y_bstride, const int32_t* y_group_ch, int B, int Cout, int H, int W, int dtype, void* stream);

/*
 * Trilinear resample, F.interpolate(mode='trilinear') with ATen's source-index
 * rule for align_corners = 1 (rag_model.py:150-153, 357-358) or 0.
 * x: [B, C, Di, Hi, Wi] -> y: [B, C, Do, Ho, Wo] (contiguous).
 */
int ragmi_trilinear3d_fwd(const void* x, void* y, int B, int C,
                          int Di, int Hi, int Wi, int Do, int Ho, int Wo,
                          int align_corners, int dtype, void* stream);

/* The same resample reading a channel slice (x_bstride elements between batch items) and writing y[b, y_ch0 + c] of a wider
 * buffer (y_bstride), with an optional ReLU after the interpolation.  It is the second half of an UPSAMPLING
 * `ConvBR_3d(1x1x1)(F.interpolate(x))` (rag_model.py:150-155, 358-365) run conv-first: the channel mix and the folded BatchNorm
 * are affine and the taps sum to one, so act(bn(conv(interp(x)))) == act(interp(bn(conv(x)))) and the mix runs on 1/8 of
 * the voxels. */
/* The Matching-Net head's last two steps in one kernel (rag_model.py:357-365: `upsample_6` = nn.Upsample(scale_factor=2,
 * mode='trilinear', align_corners=True), then last_3_3d = ConvBR_3d(C, 1, 3, 1, 1, bn=False, relu=False)):
 *   y[:, y_ch0] = act(scale * conv3x3x3_pad1(upsample2(x))[:, 0] + shift)   (scale/shift optional, one output channel)
 * x: [B, Cin, Di, Hi, Wi] (dtype, batch stride x_bstride), weight: the raw [1, Cin, 3, 3, 3] fp32 tensor, y: [B, *, 2Di, 2Hi, 2Wi]
 * (y_dtype: the input's, or RAGMI_F32 for a RAGMI_BF16 input).  The upsampled tensor is never materialised: every operand of the
 * convolution is interpolated in registers with the same fp32 source-index rule as ragmi_trilinear3d_fwd (nesting z, y, x instead
 * of ATen's x, y, z: reassociation only).  ragmi_upconv3d_c1_supported: Cin <= 64, every input axis >= 2; otherwise run
 * ragmi_trilinear3d_fwd + ragmi_conv3d_k3_small_fwd_ex. */
int ragmi_upconv3d_c1_supported(int Cin, int Di, int Hi, int Wi);
int ragmi_upconv3d_c1_fwd(const void* x, int64_t x_bstride, const void* weight, const void* scale, const void* shift, int relu, void* y,
                          int64_t y_bstride, int y_ch0, int B, int Cin, int Di, int Hi, int Wi, int dtype, int y_dtype, void* stream);

int ragmi_trilinear3d_act_fwd(const void* x, int64_t x_bstride, void* y, int64_t y_bstride, int y_ch0, int relu, int B, int C,
                              int Di, int Hi, int Wi, int Do, int Ho, int Wo, int align_corners, int dtype, void* stream);

/*
 * Feature-Net stem: 2-D 3x3 / pad 1 / stride `stride` ConvBR (stem2d1 = ConvBR_2d(6, 12, 3,
 * stride=3, padding=1), rag_model.py:201).  x: [B, Cin, H, W] -> y: [B, Cout, (H-1)/stride+1, (W-1)/stride+1].
 * weight: raw nn.Conv2d weight [Cout, Cin, 3, 3].  The stride-1 2-D convs of the Feature Net run on the 3-D
 * kernels over a depth-1 volume (weights embedded in the middle z-slice).
 */
int ragmi_conv2d_k3_strided_fwd(const void* x, const void* weight, const void* scale, const void* shift, int relu,
                                void* y, int B, int Cin, int Cout, int H, int W, int stride, int dtype, void* stream);

/*
 * y[b, y_ch0 + c] = a[b, a_ch0 + c] + b[b, b_ch0 + c]   (sum of two Identity_3d branches,
 * rag_model.py:172 with operations_3d.py:84-90).
 */
int ragmi_add_fwd(const void* a, int64_t a_bstride, int a_ch0,
                  const void* b, int64_t b_bstride, int b_ch0,
                  void* y, int64_t y_bstride, int y_ch0,
                  int B, int C, int64_t DHW, int dtype, void* stream);

/*
 * Fused Disp.forward (rag_model.py:39-44): trilinear upsample of cost[B,1,d,h,w] to
 * [maxdisp, Ho, Wo] (align_corners=False) -> Softmin over the disparity axis ->
 * DisparityRegression (sum_d p_d * d).  The reference always uses Ho=3h, Wo=3w.
 * out: [B, Ho, Wo] fp32.  Nothing of size maxdisp*Ho*Wo is materialised.
 */
int ragmi_disp_softargmin_fwd(const void* cost, void* out, int B, int d, int h, int w,
                              int maxdisp, int Ho, int Wo, int dtype, void* stream);

/*
 * DisparityRegression.forward (rag_model.py:23-29): out[b,y,x] = sum_d prob[b,d,y,x] * d.
 * prob: [B, D, H, W] contiguous -> out [B, H, W] fp32.
 */
int ragmi_disparity_regression_fwd(const void* prob, void* out, int B, int D, int H, int W,
                                   int dtype, void* stream);

/* ------------------------------------------------------------------------------------------------------------------
 * Cost volume + stem3d0 in one step, without materialising the cost volume (rag_model.py:375-383 followed by
 * ConvBR_3d(2C, Cout, 3, 1, 1), rag_model.py:234/341).  The left half of the cost volume does not depend on the disparity
 * plane and the right half depends on x - i only, so the 3-D convolution equals A[cls,tc][co,y,x] + B[cls,xr][co,y,x-i] with
 * A / B two-dimensional convolutions of the feature maps with pre-summed weights (classes: z border, distance to the x = i
 * diagonal, right border) — the same products, summed in another order.
 *   ragmi_costvol_stem_prepare: weight [Cout, 2C, 3,3,3] fp32 -> `variants` (ragmi_costvol_stem_weights_elems floats)
 *   ragmi_costvol_stem_fwd:     left/right [B,C,H,W] (dtype) -> y[B, Cout(+), D, H, W] = act(scale*(conv)+shift) (channels 0..Cout-1,
 *                               batch stride y_bstride), optional fused consumer 1x1x1 tails as in ragmi_conv3d_k3_fwd_ex;
 *                               `workspace`: ragmi_costvol_stem_workspace_elems floats.  C, Cout <= 16.
 *                               dtype RAGMI_F32 / RAGMI_BF16: the A / B planes by fp32 FMAs (exact products); RAGMI_F32X3 (fp32
 *                               storage): the planes as split-operand products on the 16-bit matrix cores under the bound
 *                               documented for the 3x3x3 convolutions above, when C % 4 == 0 and C <= 12 (else as RAGMI_F32). */
int64_t ragmi_costvol_stem_weights_elems(int C, int Cout);
int ragmi_costvol_stem_prepare(const void* weight, void* variants, int C, int Cout, void* stream);
int64_t ragmi_costvol_stem_workspace_elems(int B, int C, int Cout, int D, int H, int W);
int ragmi_costvol_stem_fwd(const void* left, const void* right, const void* variants, const void* scale, const void* shift, int relu,
                           void* y, int64_t y_bstride, void* workspace, int B, int C, int Cout, int D, int H, int W,
                           int ntail, const ragmi_tail_t* tails, int dtype, void* stream);

/* stem3d0 AND stem3d1 (rag_model.py:234-235, 341-343: ConvBR_3d(2C, 12, 3, 1, 1) on the cost volume of rag_model.py:375-383, then
 * ConvBR_3d(12, Cout, 3, 1, 1)) with stem3d0's OUTPUT never written either: ragmi_costvol_stem_fwd's variant planes; its fused tails
 * (ntail0 consumer 1x1x1 convs of stem3d0's output — cell 0's pre_preprocess, rag_model.py:125,154 — by the combine kernel without its
 * main store; ragmi_costvol_stem_fwd itself takes y = NULL for that); then stem3d1's 3x3x3 convolution on the z-marching
 * split-operand kernel, whose halo staging evaluates act(scale0 * (A + B) + shift0) from the planes — the combine kernel's
 * arithmetic, bit for bit — instead of loading a tensor.  What it saves at the headline shape: a 164 MB write and its read-back
 * (stem3d0's combine kernel was bound by exactly that write).  RAGMI_F32X3 only, Cmid = 12 (stem3d0's output channels), C <= 12 a
 * multiple of 4; y / store_main / tails / y_group_ch as ragmi_conv3d_k3_fwd_ex (tails may be RAGMI_TAIL_G4); workspace as
 * ragmi_costvol_stem_workspace_elems(B, C, Cmid, D, H, W).  ragmi_costvol_stem_conv3d_supported: 1 when the stem3d1 launch of this shape
 * runs on that kernel. */
int ragmi_costvol_stem_conv3d_supported(int C, int Cmid, int Cout, int B, int D, int H, int W, int ntail, int dtype);
int ragmi_costvol_stem_conv3d_fwd(const void* left, const void* right, const void* variants, const void* scale0, const void* shift0, int relu0,
                                  void* workspace, int ntail0, const ragmi_tail_t* tails0,
                                  const void* packed_weight, const void* scale, const void* shift, int relu,
                                  void* y, int64_t y_bstride, const int32_t* y_group_ch, int store_main, int ntail, const ragmi_tail_t* tails,
                                  int B, int C, int Cmid, int Cout, int D, int H, int W, int dtype, void* stream);

/* 1 when ragmi_conv3d_k3_fwd(_ex) (nset = 1) / ragmi_conv3d_k3_dual_fwd(_ex) (nset = 2, Cin = both inputs) called with this
 * dtype (RAGMI_F32X3 or RAGMI_BF16) runs this shape on the 16-bit matrix cores (fp16 / bf16 operands), 0 when it runs on the fp32-MFMA kernel.  Three forms
 * (conv3d_x3.hip, conv2d_x3.hip), all without a residual input and with whole 4-channel groups in and out (Cin % 4 == 0, Cout % 4 == 0): the z-marching form for
 * D*H*W >= 2^18 voxels PER SAMPLE, W >= 32, D >= 8, <= 24 input channels; the deep-level form for 8 or 16 input channels per
 * set, no fused tails, D >= 2 and D*H*W >= 2^14 voxels per sample; the depth-1 form (RAGMI_F32X3 only: the Feature Net's 2-D
 * convolutions, rag_model.py:285-323) for D == 1, W >= 16, H >= 2, no fused tails and 4..16 input channels (4 or 8 per set of a
 * dual launch) — there the taps dz != 1 meet only zero padding and only the middle slice of the packed fragments is issued.
 * B never enters: the kernel (hence the rounding) a sample gets does not depend on how a batch is split over ranks.  Always 0 for RAGMI_F32. */
int ragmi_conv3d_k3_uses_x3(int Cin, int Cout, int B, int D, int H, int W, int nset, int has_res, int ntail, int dtype);

/* ragmi_conv3d_k3_pack with two options used by the training step: transpose != 0 packs the DATA-GRADIENT conv of a forward
 * weight (source [Cin][Cout][taps], taps flipped), Cout/Cin being those of the packed conv; planar2d != 0: the source is a 2-D
 * [.,.,3,3] weight living in the dz = 1 plane (Feature Net on depth-1 volumes). */
int ragmi_conv3d_k3_pack_ex(const void* weight, void* packed, int Cout, int Cin, int transpose, int planar2d, int dtype, void* stream);

/* ragmi_conv3d_k3_pack_ex for a weight that will be used under ONE known arithmetic contract (the training step packs ~150 weights
 * per step, each for the call that follows): for_dtype == RAGMI_F32 fills only the fp32-MFMA section of `packed` (same buffer size);
 * such a buffer must be passed to the convolution entry points with dtype RAGMI_F32 only — the sections it skips are POISONED (NaN
 * fragments and multipliers), so a call under another dtype returns NaN everywhere instead of reading uninitialised memory.  Any
 * other for_dtype fills every section, like ragmi_conv3d_k3_pack_ex. */
int ragmi_conv3d_k3_pack_for(const void* weight, void* packed, int Cout, int Cin, int transpose, int planar2d, int for_dtype, void* stream);

/* ------------------------------------------------------------------------------------------------------------------
 * Training step (BASELINE config 5; the reference runs autograd through the same modules, approaches/rag.py:155-219).
 * fp32 only.  Data-gradients of the convolutions are the forward kernels applied to the output gradient with the
 * weight transposed (and its taps flipped); the functions below are the pieces with no forward twin.
 * Buffers marked (+=) are accumulated into with atomics and must be zeroed by the caller.
 */

/* Train-mode BatchNorm3d statistics (operations_3d.py:44) of the raw conv output x[B,C,DHW]: batch mean / biased variance ->
 * mean[c], invstd[c], scale[c] = gamma*invstd, shift[c] = beta - mean*scale, and (running_mean != NULL) the momentum update of
 * running_mean / running_var (unbiased) / num_batches_tracked (int64, may be NULL) with nn.BatchNorm semantics.  Two-level
 * deterministic reduction; `workspace` holds ragmi_bn_workspace_elems(B, C, DHW) floats (no initialisation needed). */
int64_t ragmi_bn_workspace_elems(int B, int C, int64_t DHW);
int ragmi_bn_train_stats_fwd(const void* x, int64_t x_bstride, int B, int C, int64_t DHW, const void* gamma, const void* beta,
                             void* running_mean, void* running_var, void* num_batches_tracked, float momentum, float eps,
                             void* workspace, void* mean, void* invstd, void* scale, void* shift, void* stream);

/* the two launches of a train-mode BatchNorm + ReLU forward: statistics, then finalize (as ragmi_bn_train_stats_fwd) fused with the
 * affine + ReLU pass y[b, y_ch0+c] = act(x[b,c]*scale[c] + shift[c]) (+ res[b, res_ch0+c] when res != NULL: a cell's running sum of
 * branches, rag_model.py:163-172, without a separate add launch) */
int ragmi_bn_train_act_fwd(const void* x, int64_t x_bstride, int B, int C, int64_t DHW, const void* gamma, const void* beta,
                           void* running_mean, void* running_var, void* num_batches_tracked, float momentum, float eps, int relu,
                           void* workspace, void* mean, void* invstd, void* scale, void* shift, void* y, int64_t y_bstride, int y_ch0,
                           const void* res, int64_t res_bstride, int res_ch0, void* stream);

/* the two launches of its adjoint: the reduction of ragmi_bn_act_bwd_coeffs, then coefficients + dx = g*c1 + x*c2 + c3 in one pass
 * (dgamma / dbeta may be NULL; accumulate != 0: +=) */
int ragmi_bn_act_bwd(const void* dy, int64_t dy_bstride, int dy_ch0, const void* x, int64_t x_bstride, const void* scale,
                     const void* shift, int relu, const void* mean, const void* invstd, int training, int B, int C, int64_t DHW,
                     void* workspace, void* dx, int64_t dx_bstride, void* dgamma, void* dbeta, int accumulate, void* stream);

/* y[b, y_ch0+c] = act(x[b,c] * scale[c] + shift[c]) (+ res[b, res_ch0+c]) — BN affine + ReLU (+ running sum) as one pass */
int ragmi_bn_act_fwd(const void* x, int64_t x_bstride, const void* scale, const void* shift, int relu,
                     const void* res, int64_t res_bstride, int res_ch0,
                     void* y, int64_t y_bstride, int y_ch0, int B, int C, int64_t DHW, void* stream);

/* ReLU + BatchNorm adjoint, reduction half: with g = dy * [x*scale+shift > 0] (x = the raw conv output) it reduces sum g and
 * sum g*x per channel and writes dgamma[c], dbeta[c] and the coefficients of dx = g*c1 + x*c2 + c3 (training != 0: batch
 * statistics were used; 0: running statistics, c2 = c3 = 0).  Same workspace as ragmi_bn_train_stats_fwd. */
int ragmi_bn_act_bwd_coeffs(const void* dy, int64_t dy_bstride, int dy_ch0, const void* x, int64_t x_bstride,
                            const void* scale, const void* shift, int relu, const void* mean, const void* invstd, int training,
                            int B, int C, int64_t DHW, void* workspace, void* c1, void* c2, void* c3, void* dgamma, void* dbeta,
                            int accumulate, void* stream);   /* dgamma / dbeta may be NULL; accumulate != 0: += */

/* dx[b,c] = g * c1[c] + x * c2[c] + c3[c]   (train-mode BN backward is linear in g and x per channel; eval BN: c2 = c3 = 0) */
int ragmi_bn_act_bwd_apply(const void* dy, int64_t dy_bstride, int dy_ch0, const void* x, int64_t x_bstride,
                           const void* scale, const void* shift, int relu, const void* c1, const void* c2, const void* c3,
                           void* dx, int64_t dx_bstride, int B, int C, int64_t DHW, void* stream);

/* dw[co][ci][tap] = sum_{b,v} g[b, g_ch0+co, v] * x[b, ci, v + tap]   — weight gradient of the 3x3x3 conv (MFMA; persistent
 * workgroups write partial sums to `workspace`, ragmi_conv3d_k3_wgrad_workspace_elems(...) floats, and a second kernel adds them:
 * deterministic, dw needs no initialisation) */
int64_t ragmi_conv3d_k3_wgrad_workspace_elems(int B, int Cin, int Cout, int D, int H, int W);
int ragmi_conv3d_k3_wgrad(const void* x, int64_t x_bstride, const void* g, int64_t g_bstride, int g_ch0, void* const* dw_list, int n_dw,
                          int accumulate, int planar2d, void* workspace, int B, int Cin, int Cout, int D, int H, int W, void* stream);
/* dw_list: n_dw (1..8) destination tensors, each [Cout/n_dw, Cin, 3,3,3] (planar2d: [.., 3,3], the dz = 1 plane) — stacked
 * sibling convolutions write every unit's gradient in place; accumulate != 0: dw += (gradient accumulation into .grad) */

/* dw[co][ci] (+=) sum_{b,v} g[b, g_ch0+co, v] * x[b, ci, v]   — weight gradient of the 1x1x1 conv */
int ragmi_conv3d_k1_wgrad(const void* x, int64_t x_bstride, const void* g, int64_t g_bstride, int g_ch0, void* dw,
                          int B, int Cin, int Cout, int64_t DHW, void* stream);

/* adjoint of ragmi_trilinear3d_fwd: dx[B,C,Di,Hi,Wi] gathers dy[B,C,Do,Ho,Wo] through the forward's taps (no atomics) */
int ragmi_trilinear3d_bwd(const void* dy, void* dx, int B, int C, int Di, int Hi, int Wi, int Do, int Ho, int Wo,
                          int align_corners, void* stream);

/* adjoint of ragmi_costvol_fwd: dleft/dright [B,C,h,w] from dcost [B,2C,d,h,w] */
int ragmi_costvol_bwd(const void* dcost, void* dleft, void* dright, int B, int C, int d, int h, int w, void* stream);

/* adjoint of ragmi_disp_softargmin_fwd: dcost[B,d,h,w] (+=) from dout[B,Ho,Wo] (recomputes the softmin statistics) */
int ragmi_disp_softargmin_bwd(const void* cost, const void* dout, void* dcost, int B, int d, int h, int w,
                              int maxdisp, int Ho, int Wo, void* stream);

/* backward of ragmi_conv2d_k3_strided_fwd's convolution (Feature-Net stem, operations_2d.py ConvBR_2d stride 3):
 * dx[B,Cin,H,W] from g[B,Cout,Ho,Wo];  dw[Cout,Cin,3,3] (+=) */
int ragmi_conv2d_k3_strided_dgrad(const void* g, const void* weight, void* dx, int B, int Cin, int Cout, int H, int W, int stride,
                                  void* stream);
int ragmi_conv2d_k3_strided_wgrad(const void* x, const void* g, void* dw, int B, int Cin, int Cout, int H, int W, int stride,
                                  void* stream);

/* adjoint of ragmi_disparity_regression_fwd: dprob[B,D,H,W] = dout[B,H,W] * d */
int ragmi_disparity_regression_bwd(const void* dout, void* dprob, int B, int D, int H, int W, void* stream);

/* ------------------------------------------------------------------------------------------------------------------
 * Loss + evaluation metrics in one pass.  mask = 0 < gt < maxdisp (approaches/rag.py:210, 418).
 * out[0] = masked smooth-L1 loss over the batch (rag.py:211, 419); out[1..5] = EPE, D1, Thres1, Thres2, Thres3 of
 * utilstool/metrics.py:45-65, each the mean over the images that pass metrics.py:30's 10 % mask rule; out[6] = masked pixel
 * count; out[7] = images kept.  est, gt: [B,H,W] fp32; acc: B*8 floats of scratch; out: 8 floats. */
int ragmi_stereo_metrics_fwd(const void* disp_est, const void* disp_gt, int B, int H, int W, float maxdisp, void* acc, void* out,
                             void* stream);

/* gradient of out[0] w.r.t. disp_est: ddisp = gout[0] * [mask] * clamp(est - gt, -1, 1) / out[6] */
int ragmi_masked_smooth_l1_bwd(const void* disp_est, const void* disp_gt, const void* out, const void* gout, void* ddisp,
                               int B, int H, int W, float maxdisp, void* stream);

/* ---------------------------------------------------------------------------------------------------------------------
 * clip_grad_norm_(max_norm) + torch.optim.SGD(lr, momentum, weight_decay).step() over flat fp32 buffers of n elements
 * (approaches/rag.py:64-70, 215-216) as two launches: g *= min(1, max_norm/(||g||+1e-6)) (max_norm <= 0: no clipping);
 * d = g + weight_decay*p; buf = first_step ? d : momentum*buf + d; p -= lr*buf.  norm_out (may be NULL) receives ||g|| before
 * clipping; workspace: ragmi_sgd_workspace_bytes() bytes of device scratch. */
int64_t ragmi_sgd_workspace_bytes(void);
int ragmi_sgd_clip_step(void* param, void* grad, void* momentum_buf, int64_t n, float lr, float momentum, float weight_decay,
                        float max_norm, int first_step, void* workspace, void* norm_out, void* stream);

/* ---------------------------------------------------------------------------------------------------------------------
 * Host-side guard for captured steps (no device work): counts the nodes of a captured hipGraph_t by kind.  A captured training
 * step must consist of kernel nodes only: on ROCm 7.2 memset / memcpy NODES of an instantiated graph were corrupted by memcpys
 * issued on the null stream between two replays, so rag_amd.train.GraphedTrainStep refuses a capture for which
 * n_memcpy + n_memset > 0.  `graph`: the hipGraph_t. */
int ragmi_graph_node_census(void* graph, int32_t* n_kernel, int32_t* n_memcpy, int32_t* n_memset, int32_t* n_other);

#ifdef __cplusplus
}
#endif
#endif /* RAG_AMD_H */
