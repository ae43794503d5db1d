"""Thin tensor-level wrappers over the C ABI (include/rag_amd.h).

Every function takes CUDA(=HIP) fp32 tensors, passes raw device pointers, sizes and the
current torch stream to librag_amd.so and returns the output tensor.  No arithmetic
happens in Python/PyTorch here: torch only allocates.
"""
from __future__ import annotations

import ctypes
import os
from typing import Optional, Sequence

import torch

from ._lib import K1RSpec, TailSpec, check, load_library

F32, BF16, F32X3 = 0, 1, 2   # RAGMI_F32 / RAGMI_BF16 / RAGMI_F32X3 of include/rag_amd.h
_DT = {torch.float32: F32, torch.bfloat16: BF16}

# Arithmetic of the fp32 3x3x3 convolutions — an explicit, process-wide host setting that travels to the library as the dtype
# argument of each call (RAGMI_F32 or RAGMI_F32X3); the library itself reads no environment variable.
#   "f16x3":  eligible volumes run as hi/lo-split products on the 16-bit matrix cores, hi*hi + hi*lo + lo*hi, on power-of-two-scaled
#             FP16 halves (fp32-class accuracy; error bound in include/rag_amd.h; the inference default, ABI dtype RAGMI_F32X3).
#             "bf16x3" — round 2's name, when the halves were bf16 — and "split" are accepted as aliases of the same setting;
#   "fp32":   every contraction on the fp32-input MFMA forms (exact fmaf chains).
# The environment variable RAGMI_X3=0 only picks the DEFAULT here, once, at import.
_CONV_PRECISIONS = ("f16x3", "fp32")
_PRECISION_ALIASES = {"bf16x3": "f16x3", "split": "f16x3"}
_conv_precision = "fp32" if os.environ.get("RAGMI_X3", "1").strip() == "0" else "f16x3"


def set_conv_precision(precision: str) -> str:
    """Select the arithmetic of fp32 3x3x3 convolutions ("f16x3" = the split form, aliases "bf16x3" / "split"; or "fp32"); returns
    the previous setting (always a canonical name)."""
    global _conv_precision
    precision = _PRECISION_ALIASES.get(precision, precision)
    if precision not in _CONV_PRECISIONS:
        raise ValueError(f"conv precision must be one of {_CONV_PRECISIONS}, got {precision!r}")
    old, _conv_precision = _conv_precision, precision
    return old


def get_conv_precision() -> str:
    return _conv_precision


class conv_precision:
    """Context manager: `with ops.conv_precision("fp32"): ...`."""

    def __init__(self, precision: str):
        self.precision = precision

    def __enter__(self):
        self.old = set_conv_precision(self.precision)
        return self

    def __exit__(self, *exc):
        set_conv_precision(self.old)
        return False


def _conv_dt(dt: int) -> int:
    """ABI dtype of a 3x3x3 convolution call for activation dtype code `dt` under the current precision setting."""
    return F32X3 if (dt == F32 and _conv_precision == "f16x3") else dt


def _stream() -> int:
    return torch.cuda.current_stream().cuda_stream


def _need_gpu(*ts: torch.Tensor) -> None:
    """Parameters (weights, folded BN) and fp32-only tensors: CUDA + float32."""
    for t in ts:
        if t is None:
            continue
        if not t.is_cuda:
            raise RuntimeError("rag_amd ops run on the MI355X only (got a CPU tensor); there is no CPU fallback")
        if t.dtype != torch.float32:
            raise RuntimeError(f"rag_amd ops: dtype {t.dtype} not built for this argument (fp32 only)")


def _act(*ts: torch.Tensor) -> int:
    """Activations: CUDA, all float32 or all bfloat16 (bf16 storage / fp32 on-chip math). Returns the ABI dtype code."""
    dt = None
    for t in ts:
        if t is None:
            continue
        if not t.is_cuda:
            raise RuntimeError("rag_amd ops run on the MI355X only (got a CPU tensor); there is no CPU fallback")
        if t.dtype not in _DT:
            raise RuntimeError(f"rag_amd ops: activation dtype {t.dtype} not built (float32 or bfloat16)")
        if dt is not None and t.dtype != dt:
            raise RuntimeError("rag_amd ops: activation tensors of one call must share a dtype")
        dt = t.dtype
    return _DT[dt]


def _act_with_tails(x, out, res, tails) -> int:
    """_act over a 3x3x3 call's tensors; a DOWN-SAMPLING tail of a bf16 call may have an fp32 destination (RAGMI_TAIL_F32)."""
    dt = _act(x, out, res, *[t.out for t in (tails or []) if not (t.down and x.dtype == torch.bfloat16 and t.out.dtype == torch.float32)])
    for t in (tails or []):
        if not t.out.is_cuda:
            raise RuntimeError("rag_amd ops run on the MI355X only (got a CPU tensor); there is no CPU fallback")
    return dt


def _planes(t: torch.Tensor) -> int:
    """Check t is [B, C, ...] with dense channel planes (a channel-slice view of a contiguous
    buffer is fine) and return its batch stride in elements."""
    inner = 1
    for i in range(t.dim() - 1, 0, -1):
        if t.shape[i] != 1 and t.stride(i) != inner:
            raise RuntimeError(f"rag_amd ops: planes must be dense N-C-D-H-W, got strides {t.stride()} for {tuple(t.shape)}")
        inner *= t.shape[i]
    return int(t.stride(0))


def _i32_array(vals: Optional[Sequence[int]]):
    if vals is None:
        return None
    return (ctypes.c_int32 * len(vals))(*[int(v) for v in vals])


# bits of the `relu` argument / of ragmi_tail_t.relu (include/rag_amd.h): channel-group-interleaved ("G4") tensors
CONV_X_G4, CONV_Y_G4, TAIL_G4 = 2, 4, 4
TAIL_ROWS = 16                    # ragmi_costvol_stem_conv3d_fwd: the tail rides in rows 12..15 of stem3d1's matrix product (include/rag_amd.h)
_STEM_TAIL_ROWS = [os.environ.get("RAGMI_STEM_TAIL_ROWS", "1") != "0"]      # the DEFAULT, read once at import (A/B tooling)


def set_stem_tail_rows(enabled: bool) -> None:
    """Whether the fused stems compute cell 0's pre_preprocess in the four idle rows of stem3d1's 12-channel matrix product (a
    split-operand product, the RAGMI_F32X3 bound) instead of the exact fp32 chain in the staging thread."""
    _STEM_TAIL_ROWS[0] = bool(enabled)


def stem_tail_rows_enabled() -> bool:
    return _STEM_TAIL_ROWS[0]


TAIL_F32, OUT_F32 = 8, 0x100      # mixed storage (include/rag_amd.h): a bf16 launch's down-sampling tail / resample launch writing fp32
_BF16_DEEP_F32 = [os.environ.get("RAGMI_BF16_DEEP_F32", "1") != "0"]      # the DEFAULT, read once at import (A/B tooling)


def set_bf16_deep_fp32(enabled: bool) -> None:
    """bf16 activation storage (BASELINE configs[2]) for the FULL-RESOLUTION tensors only: the fused executor keeps the cells that work
    below the cost volume's resolution, and everything behind them, in fp32 (tests/analysis_bf16_stage_epe.py: the level-12 cells
    and the head carry most of the bf16 error and ~2 % of the bytes).  False = every stored tensor bf16 (rounds 1-4)."""
    _BF16_DEEP_F32[0] = bool(enabled)


def bf16_deep_fp32_enabled() -> bool:
    return _BF16_DEEP_F32[0]


_BF16_HEAD_F32 = [os.environ.get("RAGMI_BF16_HEAD_F32", "1") != "0"]      # the DEFAULT, read once at import (A/B tooling)


def set_bf16_head_fp32(enabled: bool) -> None:
    """bf16 activation storage with EVERY cell bf16 (set_bf16_deep_fp32(False)): the head still keeps fp32 from last_12_3d on."""
    _BF16_HEAD_F32[0] = bool(enabled)


def bf16_head_fp32_enabled() -> bool:
    return _BF16_HEAD_F32[0]


_BF16_DEEP_RATIO = [int(os.environ.get("RAGMI_BF16_DEEP_RATIO", "8"))]      # (A/B tooling: 64 = the first level-6 cell stays bf16)


def bf16_deep_fp32_ratio() -> int:
    """A cell keeps fp32 under mixed storage when its volume is at most 1/ratio of the cost volume's (8: level 6 and below.  Measured
    with 64 — the first level-6 cell bf16, fp32 from level 12 on: 0.029 / 0.097 px against 0.025 / 0.073 at the two cost scales of
    test_bf16_storage_epe_at_two_cost_scales and no faster, 1 368 vs 1 376 maps/s: its consumers then need two crossing launches)."""
    return _BF16_DEEP_RATIO[0]
_G4 = [os.environ.get("RAGMI_G4", "1") != "0"]      # the DEFAULT, read once at import (A/B tooling); set_g4 chooses afterwards


_FUSE_STEMS = [os.environ.get("RAGMI_FUSE_STEMS", "1") != "0"]      # the DEFAULT, read once at import (A/B tooling)


def set_stem_fusion(enabled: bool) -> None:
    """Whether the fused executor may run stem3d0 + stem3d1 as ops.costvol_stem_conv3d (stem3d0's output never written)."""
    _FUSE_STEMS[0] = bool(enabled)


def stem_fusion_enabled() -> bool:
    return _FUSE_STEMS[0]


def set_g4(enabled: bool) -> None:
    """Whether the fused executor (MatchingNet._run_chain) may keep its private level-3 tensors channel-group-interleaved."""
    _G4[0] = bool(enabled)


def g4_enabled() -> bool:
    return _G4[0]



def to_g4(t: torch.Tensor) -> torch.Tensor:
    """[B, C, D, H, W] channel planes -> the same logical tensor stored group-interleaved [B][C/4][D][H][W][4], returned with the
    plane tensor's SHAPE (a contiguous buffer the G4-aware kernels read; tests and tools only)."""
    B, C = t.shape[:2]
    sp = tuple(t.shape[2:])
    perm = (0, 1) + tuple(range(3, 3 + len(sp))) + (2,)
    return t.reshape((B, C // 4, 4) + sp).permute(perm).contiguous().view(t.shape)


def from_g4(t: torch.Tensor) -> torch.Tensor:
    """inverse of to_g4"""
    B, C = t.shape[:2]
    sp = tuple(t.shape[2:])
    n = len(sp)
    perm = (0, 1, 2 + n) + tuple(range(2, 2 + n))
    return t.reshape((B, C // 4) + sp + (4,)).permute(perm).contiguous().view(t.shape)


def conv3d_k3_g4_caps(cin: int, cout: int, B: int, D: int, H: int, W: int, nset: int = 1, ntail: int = 0, ndown: int = 0,
                      dtype: torch.dtype = torch.float32) -> int:
    """bit 0: this conv3d_k3 / conv3d_k3_dual call accepts a G4 input; bit 1: it can write G4 full-resolution tails (current precision)"""
    return int(load_library().ragmi_conv3d_k3_g4_caps(cin, cout, B, D, H, W, nset, ntail, ndown, _conv_dt(_DT[dtype])))


class Tail:
    """A consumer 1x1x1 ConvBR_3d to be computed in the producing 3x3x3 kernel's epilogue:
    out[:, ch0:ch0+cout] = act(bn(weight2d @ producer_output)) (weight2d [cout <= 4, C_producer])."""

    def __init__(self, weight2d: torch.Tensor, scale: Optional[torch.Tensor], shift: Optional[torch.Tensor], relu: bool,
                 out: torch.Tensor, out_ch0: int, down: bool = False, g4: bool = False):
        """down=True: a DOWN-SAMPLING tail — `out` is at half the producer's resolution and receives
        act(bn(F.interpolate(weight2d @ producer_output, half size, 'trilinear', align_corners=True))) (ragmi_tail_t.relu bit 1).
        g4=True: `out` (fp32, contiguous) is read by its consumer as a channel-group-interleaved tensor [B][C/4][D][H][W][4]
        (RAGMI_TAIL_G4, include/rag_amd.h): this tail writes group out_ch0 / 4 of it (4 output channels, out_ch0 % 4 == 0)."""
        _need_gpu(weight2d, scale, shift)
        _act(out)
        for t in (weight2d, scale, shift):
            if t is not None and not t.is_contiguous():
                raise ValueError("Tail: weight / scale / shift must be contiguous (row slices of contiguous tensors are)")
        self.weight2d, self.scale, self.shift, self.relu, self.out, self.out_ch0 = weight2d, scale, shift, relu, out, out_ch0
        self.down = bool(down)
        self.g4 = bool(g4)
        if self.g4 and (self.down or weight2d.shape[0] != 4 or out_ch0 % 4 or not out.is_contiguous()):
            raise ValueError("Tail: a G4 destination takes a full-resolution tail of 4 channels at a group-aligned channel of a contiguous tensor")

    def spec(self) -> TailSpec:
        p = lambda t: t.data_ptr() if t is not None else None  # noqa: E731
        f32 = TAIL_F32 if (self.down and self.out.dtype == torch.float32) else 0      # (only read by bf16-storage launches)
        return TailSpec(p(self.weight2d), p(self.scale), p(self.shift), int(self.relu) | (2 if self.down else 0) | (TAIL_G4 if self.g4 else 0) | f32, self.out.data_ptr(),
                        _planes(self.out), int(self.out_ch0), int(self.weight2d.shape[0]))


def _tail_array(tails: Optional[Sequence["Tail"]]):
    if not tails:
        return 0, None
    if sum(1 for t in tails if not t.down) > 2 or sum(1 for t in tails if t.down) > 2:
        raise ValueError("at most two full-resolution and two down-sampling tails per conv launch")
    arr = (TailSpec * len(tails))(*[t.spec() for t in tails])
    return len(tails), arr


def costvol(left_fea: torch.Tensor, right_fea: torch.Tensor, maxdisp: int, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """cost[B,2C,maxdisp/3,h,w] of src/models/rag_model.py:375-383."""
    dt = _act(left_fea, right_fea, out)
    if left_fea.shape != right_fea.shape or left_fea.dim() != 4:
        raise ValueError("costvol: left/right features must both be [B, C, h, w]")
    left_fea, right_fea = left_fea.contiguous(), right_fea.contiguous()
    B, C, h, w = left_fea.shape
    d = int(maxdisp / 3)
    if out is None:
        out = torch.empty((B, 2 * C, d, h, w), device=left_fea.device, dtype=left_fea.dtype)
    check(load_library().ragmi_costvol_fwd(left_fea.data_ptr(), right_fea.data_ptr(), out.data_ptr(),
                                           B, C, d, h, w, dt, _stream()), "costvol")
    return out


def costvol_stem_prepare(weight: torch.Tensor) -> torch.Tensor:
    """Pre-summed weight variants of a ConvBR_3d(2C, Cout, 3, 1, 1) that consumes the cost volume (ragmi_costvol_stem_prepare)."""
    _need_gpu(weight)
    Cout, C2 = weight.shape[:2]
    if tuple(weight.shape[2:]) != (3, 3, 3) or C2 % 2:
        raise ValueError("costvol_stem_prepare: weight must be [Cout, 2C, 3, 3, 3]")
    lib = load_library()
    out = torch.empty((lib.ragmi_costvol_stem_weights_elems(C2 // 2, Cout),), device=weight.device, dtype=torch.float32)
    check(lib.ragmi_costvol_stem_prepare(weight.detach().contiguous().data_ptr(), out.data_ptr(), C2 // 2, Cout, _stream()),
          "costvol_stem_prepare")
    return out


def costvol_stem(left_fea: torch.Tensor, right_fea: torch.Tensor, maxdisp: int, variants: torch.Tensor, cout: int,
                 scale: Optional[torch.Tensor], shift: Optional[torch.Tensor], relu: bool, out: Optional[torch.Tensor] = None,
                 tails: Optional[Sequence[Tail]] = None, out_g4: bool = False) -> torch.Tensor:
    """act(bn(conv3x3x3(cost_volume(left_fea, right_fea)))) without building the cost volume: ragmi_costvol_stem_fwd.
    out_g4: `out` is written channel-group-interleaved (RAGMI_CONV_Y_G4)."""
    _need_gpu(variants, scale, shift)
    dt = _act(left_fea, right_fea, out, *[t.out for t in (tails or [])])
    if left_fea.shape != right_fea.shape or left_fea.dim() != 4:
        raise ValueError("costvol_stem: left/right features must both be [B, C, h, w]")
    left_fea, right_fea = left_fea.contiguous(), right_fea.contiguous()
    B, C, h, w = left_fea.shape
    d = int(maxdisp / 3)
    if out is None:
        out = torch.empty((B, cout, d, h, w), device=left_fea.device, dtype=left_fea.dtype)
    if tuple(out.shape[2:]) != (d, h, w) or out.shape[1] < cout or out.shape[0] != B:
        raise ValueError("costvol_stem: out must be [B, >=Cout, maxdisp/3, h, w]")
    lib = load_library()
    ws = torch.empty((lib.ragmi_costvol_stem_workspace_elems(B, C, cout, d, h, w),), device=left_fea.device, dtype=torch.float32)
    ntail, tarr = _tail_array(tails)
    p = lambda t: t.data_ptr() if t is not None else None  # noqa: E731
    if out_g4 and (not out.is_contiguous() or out.shape[1] != cout):
        raise ValueError("costvol_stem: a G4 output is a contiguous [B, Cout, ...] buffer")
    check(lib.ragmi_costvol_stem_fwd(left_fea.data_ptr(), right_fea.data_ptr(), variants.data_ptr(), p(scale), p(shift), int(relu) | (CONV_Y_G4 if out_g4 else 0),
                                     out.data_ptr(), _planes(out), ws.data_ptr(), B, C, cout, d, h, w, ntail, tarr, _conv_dt(dt), _stream()),
          "costvol_stem")
    return out


def costvol_stem_conv3d_supported(C: int, cmid: int, cout: int, B: int, d: int, h: int, w: int, ntail: int = 0,
                                  dtype: torch.dtype = torch.float32) -> bool:
    return bool(load_library().ragmi_costvol_stem_conv3d_supported(C, cmid, cout, B, d, h, w, ntail, _conv_dt(_DT[dtype])))


def costvol_stem_conv3d(left_fea: torch.Tensor, right_fea: torch.Tensor, maxdisp: int, variants: torch.Tensor, cmid: int,
                        scale0: Optional[torch.Tensor], shift0: Optional[torch.Tensor], relu0: bool, tails0: Optional[Sequence[Tail]],
                        packed: torch.Tensor, cout: int, scale: Optional[torch.Tensor], shift: Optional[torch.Tensor], relu: bool,
                        out: Optional[torch.Tensor], out_group_ch: Optional[Sequence[int]] = None,
                        tails: Optional[Sequence[Tail]] = None, store_main: bool = True, tail0_rows: bool = False) -> Optional[torch.Tensor]:
    """stem3d0 (folded with the cost volume) and stem3d1 in one call, stem3d0's output never written: ragmi_costvol_stem_conv3d_fwd.
    `tails0` ride on stem3d0's output, `tails` on stem3d1's; `out` may be None when store_main is False.  tail0_rows: `packed` is the
    pack of stem3d1's weight EXTENDED to 16 output channels whose rows 12..15 hold tails0[0]'s weights at the centre tap
    (stem_tail_rows_weight): that tail falls out of the matrix product (RAGMI_TAIL_ROWS)."""
    _need_gpu(variants, scale0, shift0, packed, scale, shift)
    dt = _act(left_fea, right_fea, out, *[t.out for t in (tails0 or [])], *[t.out for t in (tails or [])])
    if left_fea.shape != right_fea.shape or left_fea.dim() != 4:
        raise ValueError("costvol_stem_conv3d: left/right features must both be [B, C, h, w]")
    if out is None and store_main:
        raise ValueError("costvol_stem_conv3d: store_main needs an output buffer")
    left_fea, right_fea = left_fea.contiguous(), right_fea.contiguous()
    B, C, h, w = left_fea.shape
    d = int(maxdisp / 3)
    lib = load_library()
    ws = torch.empty((lib.ragmi_costvol_stem_workspace_elems(B, C, cmid, d, h, w),), device=left_fea.device, dtype=torch.float32)
    n0, t0 = _tail_array(tails0)
    n1, t1 = _tail_array(tails)
    if tail0_rows:
        if n0 != 1 or cout != 12:
            raise ValueError("costvol_stem_conv3d: tail0_rows takes exactly one tail behind a 12-channel stem3d1")
        t0[0].relu |= TAIL_ROWS
    p = lambda t: t.data_ptr() if t is not None else None  # noqa: E731
    check(lib.ragmi_costvol_stem_conv3d_fwd(left_fea.data_ptr(), right_fea.data_ptr(), variants.data_ptr(), p(scale0), p(shift0), int(relu0),
                                            ws.data_ptr(), n0, t0, packed.data_ptr(), p(scale), p(shift), int(relu),
                                            p(out), _planes(out) if out is not None else 0, _i32_array(out_group_ch), int(store_main), n1, t1,
                                            B, C, cmid, cout, d, h, w, _conv_dt(dt), _stream()), "costvol_stem_conv3d")
    return out


def stem_tail_rows_weight(weight: torch.Tensor, tail_weight2d: torch.Tensor) -> torch.Tensor:
    """[12, Cin, 3, 3, 3] + a [4, Cin] 1x1x1 tail on the conv's INPUT -> the 16-channel weight whose rows 12..15 are the tail at the
    centre tap (what RAGMI_TAIL_ROWS expects packed)."""
    if weight.dim() != 5 or weight.shape[0] != 12 or tuple(tail_weight2d.shape) != (4, weight.shape[1]):
        raise ValueError("stem_tail_rows_weight: a [12, Cin, 3, 3, 3] weight and a [4, Cin] tail")
    rows = torch.zeros((4,) + tuple(weight.shape[1:]), device=weight.device, dtype=weight.dtype)
    rows[:, :, 1, 1, 1] = tail_weight2d
    return torch.cat([weight, rows], dim=0).contiguous()


def conv3d_k3_pack(weight: torch.Tensor, transpose: bool = False, for_current_precision: bool = False) -> torch.Tensor:
    """Pre-pack an nn.Conv3d weight [Cout, Cin, 3, 3, 3] (or an nn.Conv2d weight [Cout, Cin, 3, 3], run as the dz = 1 plane of
    a 3x3x3 on depth-1 volumes) for conv3d_k3.  transpose=True packs the conv that computes the DATA GRADIENT of this weight's
    conv (channels swapped, taps flipped): conv3d_k3(dy, packed, cout=Cin, ...) is then dL/dx.  for_current_precision=True
    (the training step: a pack per call) fills only the sections the CURRENT conv precision reads — under "fp32" the split-operand
    fragments are skipped (ragmi_conv3d_k3_pack_for); such a pack must be consumed under the same precision."""
    _need_gpu(weight)
    planar = weight.dim() == 4
    if tuple(weight.shape[2:]) != ((3, 3) if planar else (3, 3, 3)):
        raise ValueError("conv3d_k3_pack: weight must be [Cout, Cin, 3, 3, 3] (or [Cout, Cin, 3, 3])")
    Cout, Cin = (weight.shape[1], weight.shape[0]) if transpose else (weight.shape[0], weight.shape[1])
    lib = load_library()
    n = lib.ragmi_conv3d_k3_packed_elems(Cout, Cin)
    packed = torch.empty((n,), device=weight.device, dtype=torch.float32)
    w = weight.detach().contiguous()
    if for_current_precision:
        check(lib.ragmi_conv3d_k3_pack_for(w.data_ptr(), packed.data_ptr(), Cout, Cin, int(transpose), int(planar), _conv_dt(F32), _stream()),
              "conv3d_k3_pack_for")
    else:
        check(lib.ragmi_conv3d_k3_pack_ex(w.data_ptr(), packed.data_ptr(), Cout, Cin, int(transpose), int(planar), F32, _stream()),
              "conv3d_k3_pack")
    return packed


def conv3d_k3_plan(cout: int, B: int, D: int, H: int, W: int, nset: int = 1):
    """(log2 x-tile, rows per lane, [G per workgroup]) that conv3d_k3 (nset=1) / conv3d_k3_dual (nset=2)
    will use for this shape; the kernel instantiation is conv3d_k3_kernel<G, log_tx, rows, nset, *>."""
    log_tx, rows = ctypes.c_int32(), ctypes.c_int32()
    groups = (ctypes.c_int32 * 16)()
    n = load_library().ragmi_conv3d_k3_plan(cout, B, D, H, W, nset, ctypes.byref(log_tx), ctypes.byref(rows), groups, 16)
    if n < 0:
        check(n, "conv3d_k3_plan")
    return log_tx.value, rows.value, [groups[i] for i in range(n)]


def packed_groups(cout: int) -> int:
    return (cout + 3) // 4


def conv3d_k3(x: torch.Tensor, packed: torch.Tensor, cout: int, scale: Optional[torch.Tensor], shift: Optional[torch.Tensor],
              relu: bool, out: torch.Tensor, out_group_ch: Optional[Sequence[int]] = None,
              res: Optional[torch.Tensor] = None, res_group_ch: Optional[Sequence[int]] = None,
              tails: Optional[Sequence[Tail]] = None, store_main: bool = True, x_g4: bool = False) -> torch.Tensor:
    """Fused 3x3x3 ConvBR_3d (+ running sum / concat): see ragmi_conv3d_k3_fwd in include/rag_amd.h.
    `out` (and `res`) are [B, C*, D, H, W] buffers; group g of 4 output channels lands at channel
    out_group_ch[g] (default 4g).  x_g4: x is stored channel-group-interleaved (RAGMI_CONV_X_G4)."""
    if x_g4 and not x.is_contiguous():
        raise ValueError("conv3d_k3: a G4 input is a contiguous buffer")
    _need_gpu(packed, scale, shift)
    dt = _act_with_tails(x, out, res, tails)
    B, Cin, D, H, W = x.shape
    xb = _planes(x)
    yb = _planes(out)
    rb = _planes(res) if res is not None else 0
    ng = packed_groups(cout)
    if out_group_ch is not None and len(out_group_ch) != ng:
        raise ValueError("conv3d_k3: out_group_ch needs one entry per group of 4 output channels")
    if res is not None and res_group_ch is None:
        res_group_ch = out_group_ch
    if tuple(out.shape[2:]) != (D, H, W) or out.shape[0] != B:
        raise ValueError("conv3d_k3: out must be [B, C, D, H, W] of the input's spatial size")
    max_ch = max(out_group_ch) + 4 if out_group_ch is not None else cout
    if max_ch > out.shape[1] + (3 if cout % 4 else 0):
        raise ValueError("conv3d_k3: destination channels exceed the output buffer")
    ntail, tarr = _tail_array(tails)
    check(load_library().ragmi_conv3d_k3_fwd_ex(
        x.data_ptr(), xb, packed.data_ptr(),
        scale.data_ptr() if scale is not None else None, shift.data_ptr() if shift is not None else None, int(relu) | (CONV_X_G4 if x_g4 else 0),
        out.data_ptr(), yb, _i32_array(out_group_ch),
        res.data_ptr() if res is not None else None, rb, _i32_array(res_group_ch),
        B, Cin, cout, D, H, W, int(store_main), ntail, tarr, _conv_dt(dt), _stream()), "conv3d_k3")
    return out


def conv3d_k3_small(x: torch.Tensor, weight: torch.Tensor, scale, shift, relu: bool, out: torch.Tensor, out_ch0: int = 0,
                    res: Optional[torch.Tensor] = None, res_ch0: int = 0) -> torch.Tensor:
    """3x3x3 ConvBR_3d with Cout <= 2 on the VALU (raw weight [Cout, Cin, 3, 3, 3]): ragmi_conv3d_k3_small_fwd_ex.  `out` may be
    float32 while x is bfloat16 (the head's `mat` stays fp32 under bf16 activation storage)."""
    _need_gpu(weight, scale, shift)
    dt = _act(x, res)
    ydt = _act(out)
    if ydt != dt and not (dt == BF16 and ydt == F32):
        raise RuntimeError("conv3d_k3_small: out must have x's dtype (or be float32 when x is bfloat16)")
    B, Cin, D, H, W = x.shape
    cout = weight.shape[0]
    w = weight.detach().contiguous()
    if out_ch0 + cout > out.shape[1] or tuple(out.shape[2:]) != (D, H, W):
        raise ValueError("conv3d_k3_small: output buffer too small / wrong spatial size")
    ptr = lambda t: t.data_ptr() if t is not None else None  # noqa: E731
    check(load_library().ragmi_conv3d_k3_small_fwd_ex(
        x.data_ptr(), _planes(x), w.data_ptr(), ptr(scale), ptr(shift), int(relu), out.data_ptr(), _planes(out), out_ch0,
        ptr(res), _planes(res) if res is not None else 0, res_ch0, B, Cin, cout, D, H, W, dt, ydt, _stream()), "conv3d_k3_small")
    return out


def conv3d_k3_dual(x: torch.Tensor, cin_a: int, packed_a: torch.Tensor, scale_a, shift_a,
                   packed_b: torch.Tensor, scale_b, shift_b, cout: int, relu: bool, out: torch.Tensor,
                   out_group_ch: Optional[Sequence[int]] = None, res: Optional[torch.Tensor] = None,
                   res_group_ch: Optional[Sequence[int]] = None, tails: Optional[Sequence[Tail]] = None,
                   store_main: bool = True, x_g4: bool = False) -> torch.Tensor:
    """Two sibling ConvBR_3d groups in one launch: out = act(bnA(convA(x[:, :cin_a]))) + act(bnB(convB(x[:, cin_a:])))
    (+ res): see ragmi_conv3d_k3_dual_fwd in include/rag_amd.h.  x_g4: x is stored channel-group-interleaved (RAGMI_CONV_X_G4)."""
    if x_g4 and not x.is_contiguous():
        raise ValueError("conv3d_k3_dual: a G4 input is a contiguous buffer")
    _need_gpu(packed_a, packed_b, scale_a, shift_a, scale_b, shift_b)
    dt = _act_with_tails(x, out, res, tails)
    B, Cx, D, H, W = x.shape
    ng = packed_groups(cout)
    if out_group_ch is not None and len(out_group_ch) != ng:
        raise ValueError("conv3d_k3_dual: out_group_ch needs one entry per group of 4 output channels")
    if res is not None and res_group_ch is None:
        res_group_ch = out_group_ch
    if tuple(out.shape[2:]) != (D, H, W) or out.shape[0] != B or not 0 < cin_a < Cx:
        raise ValueError("conv3d_k3_dual: bad shapes")
    ptr = lambda t: t.data_ptr() if t is not None else None  # noqa: E731
    ntail, tarr = _tail_array(tails)
    check(load_library().ragmi_conv3d_k3_dual_fwd_ex(
        x.data_ptr(), _planes(x), cin_a, packed_a.data_ptr(), ptr(scale_a), ptr(shift_a),
        Cx - cin_a, packed_b.data_ptr(), ptr(scale_b), ptr(shift_b), int(relu) | (CONV_X_G4 if x_g4 else 0),
        out.data_ptr(), _planes(out), _i32_array(out_group_ch),
        ptr(res), _planes(res) if res is not None else 0, _i32_array(res_group_ch),
        B, cout, D, H, W, int(store_main), ntail, tarr, _conv_dt(dt), _stream()), "conv3d_k3_dual")
    return out


def conv3d_k1(x: torch.Tensor, weight2d: torch.Tensor, scale: Optional[torch.Tensor], shift: Optional[torch.Tensor],
              relu: bool, out: torch.Tensor, out_ch0: int = 0, transposed: bool = False) -> torch.Tensor:
    """Fused 1x1x1 ConvBR_3d writing out[:, out_ch0:out_ch0+Cout].  `transposed`: weight2d is [Cin, Cout] (the forward weight of
    the conv whose data gradient this is), read in place."""
    _need_gpu(weight2d, scale, shift)
    dt = _act(x, out)
    B, Cin = x.shape[:2]
    Cout = weight2d.shape[1 if transposed else 0]
    if not weight2d.is_contiguous() or weight2d.shape[0 if transposed else 1] != Cin:
        raise ValueError("conv3d_k1: weight must be contiguous [Cout, Cin] (or [Cin, Cout] with transposed=True)")
    dhw = 1
    for s in x.shape[2:]:
        dhw *= s
    if out_ch0 + Cout > out.shape[1] or tuple(out.shape[2:]) != tuple(x.shape[2:]):
        raise ValueError("conv3d_k1: output buffer too small / wrong spatial size")
    check(load_library().ragmi_conv3d_k1_fwd_ex(
        x.data_ptr(), _planes(x), weight2d.data_ptr(), int(transposed),
        scale.data_ptr() if scale is not None else None, shift.data_ptr() if shift is not None else None, int(relu),
        out.data_ptr(), _planes(out), out_ch0, B, Cin, Cout, dhw, dt, _stream()), "conv3d_k1")
    return out


_CHAIN_K1 = [os.environ.get("RAGMI_CHAIN_K1", "1") != "0"]      # the DEFAULT, read once at import (A/B tooling)


def set_chain_k1(enabled: bool) -> None:
    """Whether the head runs last_12_3d and last_6_3d's channel mix as one launch (conv3d_k1_chain)."""
    _CHAIN_K1[0] = bool(enabled)


def chain_k1_enabled() -> bool:
    return _CHAIN_K1[0]


def conv3d_k1_chain_supported(cin: int, cmid: int, cout: int) -> bool:
    return bool(load_library().ragmi_conv3d_k1_chain_supported(int(cin), int(cmid), int(cout)))


def conv3d_k1_chain(x: torch.Tensor, w1: torch.Tensor, scale1, shift1, relu1: bool, w2: torch.Tensor, scale2, shift2, relu2: bool,
                    out: torch.Tensor, out_ch0: int = 0) -> torch.Tensor:
    """Two 1x1x1 ConvBR_3d in a row as one launch (ragmi_conv3d_k1_chain_fwd): out[:, ch0:ch0+Cout] = act2(bn2(w2 @ act1(bn1(w1 @ x))));
    the bits of two conv3d_k1 calls."""
    _need_gpu(w1, scale1, shift1, w2, scale2, shift2)
    dt = _act(x, out)
    B, Cin = x.shape[:2]
    Cmid, Cout = w1.shape[0], w2.shape[0]
    if not (w1.is_contiguous() and w2.is_contiguous()) or w1.shape[1] != Cin or w2.shape[1] != Cmid:
        raise ValueError("conv3d_k1_chain: weights must be contiguous [Cmid, Cin] and [Cout, Cmid]")
    dhw = 1
    for sdim in x.shape[2:]:
        dhw *= sdim
    if out_ch0 + Cout > out.shape[1] or tuple(out.shape[2:]) != tuple(x.shape[2:]):
        raise ValueError("conv3d_k1_chain: output buffer too small / wrong spatial size")
    p = lambda t: t.data_ptr() if t is not None else None  # noqa: E731
    check(load_library().ragmi_conv3d_k1_chain_fwd(x.data_ptr(), _planes(x), w1.data_ptr(), p(scale1), p(shift1), int(relu1), Cmid,
                                                   w2.data_ptr(), p(scale2), p(shift2), int(relu2), out.data_ptr(), _planes(out), out_ch0,
                                                   B, Cin, Cout, dhw, dt, _stream()), "conv3d_k1_chain")
    return out


def conv3d_k1_resample(x: torch.Tensor, size: Sequence[int], align_corners: bool, weight2d: torch.Tensor,
                       scale: Optional[torch.Tensor], shift: Optional[torch.Tensor], relu: bool, out: torch.Tensor,
                       out_ch0: int = 0) -> torch.Tensor:
    """act(bn(conv1x1x1(F.interpolate(x, size, 'trilinear', align_corners)))) without materialising the resampled tensor.
    x bf16 with an fp32 `out` is the one mixed-storage form (RAGMI_BF16 | RAGMI_OUT_F32: same fp32 arithmetic on chip)."""
    _need_gpu(weight2d, scale, shift)
    if x.dtype == torch.bfloat16 and out.dtype == torch.float32:
        dt = _act(x) | OUT_F32
        _act(out)
    else:
        dt = _act(x, out)
    B, Cin, Di, Hi, Wi = x.shape
    Do, Ho, Wo = [int(v) for v in size]
    Cout = weight2d.shape[0]
    if out_ch0 + Cout > out.shape[1] or tuple(out.shape[2:]) != (Do, Ho, Wo):
        raise ValueError("conv3d_k1_resample: output buffer too small / wrong spatial size")
    check(load_library().ragmi_conv3d_k1_resample_fwd(
        x.data_ptr(), _planes(x), Di, Hi, Wi, weight2d.data_ptr(),
        scale.data_ptr() if scale is not None else None, shift.data_ptr() if shift is not None else None, int(relu),
        out.data_ptr(), _planes(out), out_ch0, B, Cin, Cout, Do, Ho, Wo, int(bool(align_corners)), dt, _stream()),
        "conv3d_k1_resample")
    return out


def conv3d_k1_resample_pair(specs, size: Sequence[int], out: torch.Tensor) -> torch.Tensor:
    """Two resample(align_corners=True) + 1x1x1 ConvBR_3d writing into `out` at the same output `size`, as one launch.
    specs: two tuples (x, weight2d, scale, shift, relu, out_ch0)."""
    arr = []
    dt = _act(out, *[s[0] for s in specs])
    for (x, w2d, scale, shift, relu, ch0) in specs:
        _need_gpu(w2d, scale, shift)
        p = lambda t: t.data_ptr() if t is not None else None  # noqa: E731
        arr.append(K1RSpec(x.data_ptr(), _planes(x), x.shape[2], x.shape[3], x.shape[4], w2d.data_ptr(), p(scale), p(shift),
                           int(relu), int(ch0), x.shape[1], w2d.shape[0]))
        if ch0 + w2d.shape[0] > out.shape[1]:
            raise ValueError("conv3d_k1_resample_pair: output buffer too small")
    Do, Ho, Wo = [int(v) for v in size]
    if tuple(out.shape[2:]) != (Do, Ho, Wo):
        raise ValueError("conv3d_k1_resample_pair: wrong output spatial size")
    check(load_library().ragmi_conv3d_k1_resample_pair_fwd(ctypes.byref(arr[0]), ctypes.byref(arr[1]), out.data_ptr(), _planes(out),
                                                           out.shape[0], Do, Ho, Wo, 1, dt, _stream()), "conv3d_k1_resample_pair")
    return out


def conv3d_k1_resample_multi(specs, size: Sequence[int]) -> None:
    """Up to three resample(align_corners=True) + 1x1x1 ConvBR_3d at the same output `size` as one launch, each into its own buffer
    (ragmi_conv3d_k1_resample_multi_fwd).  specs: tuples (x, weight2d, scale, shift, relu, out, out_ch0)."""
    if not 1 <= len(specs) <= 3:
        raise ValueError("conv3d_k1_resample_multi: 1..3 specs")
    dt = _act(*[s[0] for s in specs], *[s[5] for s in specs])
    Do, Ho, Wo = [int(v) for v in size]
    arr, ys, bs = [], [], []
    p = lambda t: t.data_ptr() if t is not None else None  # noqa: E731
    for (x, w2d, scale, shift, relu, out, ch0) in specs:
        _need_gpu(w2d, scale, shift)
        if tuple(out.shape[2:]) != (Do, Ho, Wo) or ch0 + w2d.shape[0] > out.shape[1] or out.shape[0] != x.shape[0]:
            raise ValueError("conv3d_k1_resample_multi: output buffer too small / wrong size")
        arr.append(K1RSpec(x.data_ptr(), _planes(x), x.shape[2], x.shape[3], x.shape[4], w2d.data_ptr(), p(scale), p(shift),
                           int(relu), int(ch0), x.shape[1], w2d.shape[0]))
        ys.append(out.data_ptr())
        bs.append(_planes(out))
    n = len(arr)
    from ._lib import c_k1r_p
    parr = (c_k1r_p * n)(*[ctypes.pointer(a) for a in arr])
    check(load_library().ragmi_conv3d_k1_resample_multi_fwd(parr, (ctypes.c_void_p * n)(*ys), (ctypes.c_int64 * n)(*bs), n, specs[0][0].shape[0],
                                                            Do, Ho, Wo, 1, dt, _stream()), "conv3d_k1_resample_multi")


def cell2d_supported(C: int, cin0: int, cin1: int, cout: int, H: int, W: int, dtype=torch.float32) -> bool:
    """True when ragmi_cell2d_fwd is built for this Cell_2d shape under the CURRENT conv precision (f16x3, fp32 storage)."""
    if dtype != torch.float32:
        return False
    return bool(load_library().ragmi_cell2d_supported(int(C), int(cin0), int(cin1), int(cout), int(H), int(W), _conv_dt(_DT[dtype])))


def cell2d(s0: torch.Tensor, pre0, s1: torch.Tensor, pre1, C: int, packed_a: torch.Tensor, scale_a, shift_a,
           packed_b: torch.Tensor, scale_b, shift_b, cout: int, relu: bool, out: torch.Tensor,
           out_group_ch: Optional[Sequence[int]] = None) -> torch.Tensor:
    """A whole Cell_2d (rag_model.py:143-177, every new state = conv(s0) + conv(s1)) as ONE launch: see ragmi_cell2d_fwd in
    include/rag_amd.h.  s0 / s1: [B, Cin, 1, Hi, Wi] (depth-1 volumes) or [B, Cin, Hi, Wi]; pre0 / pre1 = (weight2d [C, Cin], scale,
    shift, relu) of pre_preprocess / preprocess; out: [B, C*, 1, H, W] (or 4-D), group g of 4 output channels at out_group_ch[g]."""
    from ._lib import Cell2dIn
    _need_gpu(packed_a, packed_b, scale_a, shift_a, scale_b, shift_b)
    dt = _act(s0, s1, out)
    H, W = int(out.shape[-2]), int(out.shape[-1])
    B = out.shape[0]
    p = lambda t: t.data_ptr() if t is not None else None  # noqa: E731
    ins = []
    for x, (w2d, sc, sh, rl) in ((s0, pre0), (s1, pre1)):
        _need_gpu(w2d, sc, sh)
        if x.shape[0] != B or w2d.shape != (C, x.shape[1]) or not w2d.is_contiguous():
            raise ValueError("cell2d: input / 1x1 weight shapes do not match")
        ins.append(Cell2dIn(x.data_ptr(), _planes(x), x.shape[1], x.shape[-2], x.shape[-1], w2d.data_ptr(), p(sc), p(sh), int(bool(rl))))
    groups = list(out_group_ch) if out_group_ch is not None else [4 * g for g in range(cout // 4)]
    if len(groups) != cout // 4 or cout % 4 or any(g < 0 or g + 4 > out.shape[1] for g in groups):
        raise ValueError("cell2d: bad destination channel groups")
    garr = (ctypes.c_int32 * len(groups))(*groups)
    check(load_library().ragmi_cell2d_fwd(ctypes.byref(ins[0]), ctypes.byref(ins[1]), int(C), packed_a.data_ptr(), p(scale_a), p(shift_a),
                                          packed_b.data_ptr(), p(scale_b), p(shift_b), int(bool(relu)), out.data_ptr(), _planes(out), garr,
                                          B, int(cout), H, W, _conv_dt(dt), _stream()), "cell2d")
    return out


def conv2d_k3_strided(x: torch.Tensor, weight: torch.Tensor, scale, shift, relu: bool, stride: int) -> torch.Tensor:
    """2-D 3x3 / pad 1 / stride-s ConvBR of the Feature-Net stem: x[B,Cin,H,W] -> [B,Cout,Ho,Wo]."""
    _need_gpu(weight, scale, shift)
    dt = _act(x)
    x = x.contiguous()
    B, Cin, H, W = x.shape
    Cout = weight.shape[0]
    Ho, Wo = (H - 1) // stride + 1, (W - 1) // stride + 1
    out = torch.empty((B, Cout, Ho, Wo), device=x.device, dtype=x.dtype)
    w = weight.detach().contiguous()
    p = lambda t: t.data_ptr() if t is not None else None  # noqa: E731
    check(load_library().ragmi_conv2d_k3_strided_fwd(x.data_ptr(), w.data_ptr(), p(scale), p(shift), int(relu), out.data_ptr(),
                                                     B, Cin, Cout, H, W, int(stride), dt, _stream()), "conv2d_k3_strided")
    return out


def down2_tail_supported(D: int, H: int, W: int) -> bool:
    """True when a x0.5 trilinear down-sampling (align_corners=True) of a [D, H, W] volume reads aligned source pairs (2o, 2o+1) on
    every axis — what a down-sampling tail needs (ragmi_down2_tail_supported)."""
    return bool(load_library().ragmi_down2_tail_supported(int(D), int(H), int(W)))


def upconv3d_c1_supported(cin: int, di: int, hi: int, wi: int) -> bool:
    return bool(load_library().ragmi_upconv3d_c1_supported(int(cin), int(di), int(hi), int(wi)))


def upconv3d_c1(x: torch.Tensor, weight: torch.Tensor, scale: Optional[torch.Tensor], shift: Optional[torch.Tensor], relu: bool,
                out: Optional[torch.Tensor] = None, out_ch0: int = 0, out_dtype: Optional[torch.dtype] = None) -> torch.Tensor:
    """act(bn(conv3x3x3(F.interpolate(x, scale_factor=2, mode='trilinear', align_corners=True)))) for a ONE-output-channel conv
    (raw weight [1, Cin, 3, 3, 3]) without materialising the upsampled tensor: the head's upsample_6 + last_3_3d
    (rag_model.py:357-365), ragmi_upconv3d_c1_fwd.  `out` may be float32 while x is bfloat16."""
    _need_gpu(weight, scale, shift)
    dt = _act(x)
    B, Cin, Di, Hi, Wi = x.shape
    if tuple(weight.shape) != (1, Cin, 3, 3, 3):
        raise ValueError("upconv3d_c1: weight must be [1, Cin, 3, 3, 3]")
    if out is None:
        out = torch.empty((B, 1, 2 * Di, 2 * Hi, 2 * Wi), device=x.device, dtype=out_dtype or x.dtype)
    ydt = _act(out)
    if tuple(out.shape[2:]) != (2 * Di, 2 * Hi, 2 * Wi) or out.shape[0] != B or out_ch0 >= out.shape[1]:
        raise ValueError("upconv3d_c1: out must be [B, >= out_ch0 + 1, 2Di, 2Hi, 2Wi]")
    ptr = lambda t: t.data_ptr() if t is not None else None  # noqa: E731
    check(load_library().ragmi_upconv3d_c1_fwd(x.data_ptr(), _planes(x), weight.detach().contiguous().data_ptr(), ptr(scale), ptr(shift),
                                               int(relu), out.data_ptr(), _planes(out), int(out_ch0), B, Cin, Di, Hi, Wi, dt, ydt, _stream()),
          "upconv3d_c1")
    return out


def trilinear3d(x: torch.Tensor, size: Sequence[int], align_corners: bool) -> torch.Tensor:
    """F.interpolate(x, size, mode='trilinear', align_corners=...) for x[B,C,D,H,W]."""
    dt = _act(x)
    x = x.contiguous()
    B, C, Di, Hi, Wi = x.shape
    Do, Ho, Wo = [int(s) for s in size]
    out = torch.empty((B, C, Do, Ho, Wo), device=x.device, dtype=x.dtype)
    check(load_library().ragmi_trilinear3d_fwd(x.data_ptr(), out.data_ptr(), B, C, Di, Hi, Wi, Do, Ho, Wo,
                                               int(bool(align_corners)), dt, _stream()), "trilinear3d")
    return out


def trilinear3d_act(x: torch.Tensor, size: Sequence[int], align_corners: bool, relu: bool, out: torch.Tensor,
                    out_ch0: int = 0) -> torch.Tensor:
    """out[:, out_ch0:out_ch0+C] = act(F.interpolate(x, size, 'trilinear', align_corners)) — x may be a channel slice."""
    dt = _act(x, out)
    B, C, Di, Hi, Wi = x.shape
    Do, Ho, Wo = [int(s) for s in size]
    if out_ch0 + C > out.shape[1] or tuple(out.shape[2:]) != (Do, Ho, Wo):
        raise ValueError("trilinear3d_act: output buffer too small / wrong spatial size")
    check(load_library().ragmi_trilinear3d_act_fwd(x.data_ptr(), _planes(x), out.data_ptr(), _planes(out), out_ch0, int(relu), B, C,
                                                   Di, Hi, Wi, Do, Ho, Wo, int(bool(align_corners)), dt, _stream()), "trilinear3d_act")
    return out


def add(a: torch.Tensor, a_ch0: int, b: torch.Tensor, b_ch0: int, out: torch.Tensor, out_ch0: int, channels: int) -> torch.Tensor:
    """out[:, out_ch0:+C] = a[:, a_ch0:+C] + b[:, b_ch0:+C]."""
    dt = _act(a, b, out)
    dhw = 1
    for s in a.shape[2:]:
        dhw *= s
    check(load_library().ragmi_add_fwd(a.data_ptr(), _planes(a), a_ch0, b.data_ptr(), _planes(b), b_ch0,
                                       out.data_ptr(), _planes(out), out_ch0, a.shape[0], channels, dhw, dt, _stream()), "add")
    return out


def disp_softargmin(cost: torch.Tensor, maxdisp: int) -> torch.Tensor:
    """Fused Disp.forward: cost[B,1,d,h,w] (or [B,d,h,w]) -> disparity [B,3h,3w]."""
    dt = _act(cost)
    if cost.dim() == 5:
        if cost.shape[1] != 1:
            raise ValueError("disp_softargmin: expected a single-channel cost volume")
        cost = cost[:, 0]
    cost = cost.contiguous()
    B, d, h, w = cost.shape
    out = torch.empty((B, 3 * h, 3 * w), device=cost.device, dtype=torch.float32)
    check(load_library().ragmi_disp_softargmin_fwd(cost.data_ptr(), out.data_ptr(), B, d, h, w, int(maxdisp),
                                                   3 * h, 3 * w, dt, _stream()), "disp_softargmin")
    return out


def disparity_regression(prob: torch.Tensor, maxdisp: int) -> torch.Tensor:
    """DisparityRegression.forward: prob[B,D,H,W] -> [B,H,W]."""
    dt = _act(prob)
    assert prob.is_contiguous()  # the reference asserts this too (rag_model.py:24)
    B, D, H, W = prob.shape
    if D != maxdisp:
        raise ValueError("disparity_regression: prob.shape[1] must equal maxdisp")
    out = torch.empty((B, H, W), device=prob.device, dtype=torch.float32)
    check(load_library().ragmi_disparity_regression_fwd(prob.data_ptr(), out.data_ptr(), B, D, H, W, dt, _stream()),
          "disparity_regression")
    return out


# ------------------------------------------------------------------------------------------------------------------
# training-step pieces (fp32): thin wrappers over the train.hip entry points; rag_amd/autograd.py composes them
def _vol(t: torch.Tensor) -> int:
    n = 1
    for s in t.shape[2:]:
        n *= s
    return n


def bn_train_stats(x: torch.Tensor, gamma: torch.Tensor, beta: torch.Tensor, running_mean: Optional[torch.Tensor],
                   running_var: Optional[torch.Tensor], num_batches_tracked: Optional[torch.Tensor], momentum: float, eps: float):
    """Train-mode BatchNorm statistics of the raw conv output x[B,C,...]: returns the [4, C] tensor (mean, invstd, scale, shift)
    and updates the running statistics in place (ragmi_bn_train_stats_fwd)."""
    _need_gpu(x, gamma, beta, running_mean, running_var)
    B, C = x.shape[:2]
    lib = load_library()
    ws = torch.empty((lib.ragmi_bn_workspace_elems(B, C, _vol(x)),), device=x.device, dtype=torch.float32)
    out = torch.empty((4, C), device=x.device, dtype=torch.float32)
    if num_batches_tracked is not None and num_batches_tracked.dtype != torch.int64:
        raise RuntimeError("bn_train_stats: num_batches_tracked must be int64")
    p = lambda t: t.data_ptr() if t is not None else None  # noqa: E731
    check(lib.ragmi_bn_train_stats_fwd(x.data_ptr(), _planes(x), B, C, _vol(x), gamma.data_ptr(), beta.data_ptr(), p(running_mean),
                                       p(running_var), p(num_batches_tracked), float(momentum), float(eps), ws.data_ptr(),
                                       out[0].data_ptr(), out[1].data_ptr(), out[2].data_ptr(), out[3].data_ptr(), _stream()),
          "bn_train_stats")
    return out


def bn_train_act(x: torch.Tensor, gamma: torch.Tensor, beta: torch.Tensor, running_mean: Optional[torch.Tensor],
                 running_var: Optional[torch.Tensor], num_batches_tracked: Optional[torch.Tensor], momentum: float, eps: float, relu: bool,
                 res: Optional[torch.Tensor] = None):
    """Train-mode BatchNorm + ReLU of the raw conv output x as two launches (ragmi_bn_train_act_fwd): returns (y, stats) with
    stats the [4, C] tensor (mean, invstd, scale, shift); the running statistics are updated in place.  `res` (same shape) is added
    after the activation."""
    _need_gpu(x, gamma, beta, running_mean, running_var, res)
    B, C = x.shape[:2]
    if res is not None and tuple(res.shape) != tuple(x.shape):
        raise ValueError("bn_train_act: res must have the shape of x")
    lib = load_library()
    ws = torch.empty((lib.ragmi_bn_workspace_elems(B, C, _vol(x)),), device=x.device, dtype=torch.float32)
    st = torch.empty((4, C), device=x.device, dtype=torch.float32)
    y = torch.empty(x.shape, device=x.device, dtype=torch.float32)
    if num_batches_tracked is not None and num_batches_tracked.dtype != torch.int64:
        raise RuntimeError("bn_train_act: num_batches_tracked must be int64")
    p = lambda t: t.data_ptr() if t is not None else None  # noqa: E731
    check(lib.ragmi_bn_train_act_fwd(x.data_ptr(), _planes(x), B, C, _vol(x), gamma.data_ptr(), beta.data_ptr(), p(running_mean),
                                     p(running_var), p(num_batches_tracked), float(momentum), float(eps), int(relu), ws.data_ptr(),
                                     st[0].data_ptr(), st[1].data_ptr(), st[2].data_ptr(), st[3].data_ptr(), y.data_ptr(), _planes(y), 0,
                                     p(res), _planes(res) if res is not None else 0, 0, _stream()), "bn_train_act")
    return y, st


def bn_act_bwd(dy: torch.Tensor, x: torch.Tensor, scale, shift, relu: bool, mean, invstd, training: bool,
               out: Optional[torch.Tensor] = None, dgamma_into: Optional[torch.Tensor] = None, dbeta_into: Optional[torch.Tensor] = None):
    """ReLU + BatchNorm adjoint as two launches (ragmi_bn_act_bwd): returns (dx, dgamma, dbeta); with `dgamma_into` / `dbeta_into`
    the parameter gradients are accumulated into those tensors and None is returned for them."""
    _need_gpu(dy, x, scale, shift, mean, invstd, out, dgamma_into, dbeta_into)
    B, C = x.shape[:2]
    lib = load_library()
    ws = torch.empty((lib.ragmi_bn_workspace_elems(B, C, _vol(x)),), device=x.device, dtype=torch.float32)
    dx = out if out is not None else torch.empty(x.shape, device=x.device, dtype=torch.float32)
    direct = dgamma_into is not None and dbeta_into is not None
    gb = None if direct else torch.empty((2, C), device=x.device, dtype=torch.float32)
    dg, db = (dgamma_into, dbeta_into) if direct else (gb[0], gb[1])
    check(lib.ragmi_bn_act_bwd(dy.data_ptr(), _planes(dy), 0, x.data_ptr(), _planes(x), scale.data_ptr(), shift.data_ptr(), int(relu),
                               mean.data_ptr(), invstd.data_ptr(), int(training), B, C, _vol(x), ws.data_ptr(), dx.data_ptr(), _planes(dx),
                               dg.data_ptr(), db.data_ptr(), int(direct), _stream()), "bn_act_bwd")
    return dx, (None if direct else gb[0]), (None if direct else gb[1])


def bn_act(x: torch.Tensor, scale: torch.Tensor, shift: torch.Tensor, relu: bool, out: Optional[torch.Tensor] = None,
           out_ch0: int = 0, res: Optional[torch.Tensor] = None, res_ch0: int = 0) -> torch.Tensor:
    """out[:, ch0:ch0+C] = act(x * scale + shift) (+ res[:, res_ch0:+C])."""
    _need_gpu(x, scale, shift, out, res)
    B, C = x.shape[:2]
    if out is None:
        out = torch.empty_like(x)
    check(load_library().ragmi_bn_act_fwd(x.data_ptr(), _planes(x), scale.data_ptr(), shift.data_ptr(), int(relu),
                                          res.data_ptr() if res is not None else None, _planes(res) if res is not None else 0, res_ch0,
                                          out.data_ptr(), _planes(out), out_ch0, B, C, _vol(x), _stream()), "bn_act")
    return out


def bn_act_bwd_coeffs(dy: torch.Tensor, dy_ch0: int, x: torch.Tensor, scale, shift, relu: bool, mean, invstd, training: bool,
                      dgamma_into: Optional[torch.Tensor] = None, dbeta_into: Optional[torch.Tensor] = None):
    """Reduction half of the ReLU+BN adjoint: returns the [5, C] tensor (c1, c2, c3, dgamma, dbeta) with
    dx = g*c1 + x*c2 + c3, g = dy * [x*scale+shift > 0] (ragmi_bn_act_bwd_coeffs).  `dgamma_into` / `dbeta_into`: accumulate the
    parameter gradients straight into these tensors (a parameter's .grad) instead of rows 3 / 4."""
    _need_gpu(dy, x, scale, shift, mean, invstd, dgamma_into, dbeta_into)
    B, C = x.shape[:2]
    lib = load_library()
    ws = torch.empty((lib.ragmi_bn_workspace_elems(B, C, _vol(x)),), device=x.device, dtype=torch.float32)
    out = torch.empty((5, C), device=x.device, dtype=torch.float32)
    direct = dgamma_into is not None or dbeta_into is not None
    if direct and (dgamma_into is None or dbeta_into is None):
        raise ValueError("bn_act_bwd_coeffs: pass both dgamma_into and dbeta_into, or neither")
    dg, db = (dgamma_into, dbeta_into) if direct else (out[3], out[4])
    check(lib.ragmi_bn_act_bwd_coeffs(dy.data_ptr(), _planes(dy), dy_ch0, x.data_ptr(), _planes(x), scale.data_ptr(), shift.data_ptr(),
                                      int(relu), mean.data_ptr(), invstd.data_ptr(), int(training), B, C, _vol(x), ws.data_ptr(),
                                      out[0].data_ptr(), out[1].data_ptr(), out[2].data_ptr(), dg.data_ptr(), db.data_ptr(), int(direct),
                                      _stream()), "bn_act_bwd_coeffs")
    return out


def bn_act_bwd_apply(dy: torch.Tensor, dy_ch0: int, x: torch.Tensor, scale, shift, relu: bool, c1, c2, c3,
                     out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """dx = g * c1 + x * c2 + c3 (per channel); `out` may be a channel slice of a wider buffer."""
    _need_gpu(dy, x, scale, shift, c1, c2, c3, out)
    B, C = x.shape[:2]
    dx = out if out is not None else torch.empty(x.shape, device=x.device, dtype=torch.float32)
    check(load_library().ragmi_bn_act_bwd_apply(dy.data_ptr(), _planes(dy), dy_ch0, x.data_ptr(), _planes(x), scale.data_ptr(),
                                                shift.data_ptr(), int(relu), c1.data_ptr(), c2.data_ptr(), c3.data_ptr(), dx.data_ptr(),
                                                _planes(dx), B, C, _vol(x), _stream()), "bn_act_bwd_apply")
    return dx


def conv3d_k3_wgrad(x: torch.Tensor, g: torch.Tensor, cout: int, g_ch0: int = 0, into: Optional[Sequence[torch.Tensor]] = None,
                    planar2d: bool = False):
    """dW of a 3x3x3 / pad 1 conv from its input x and output gradient g.  Default: returns a fresh [cout, Cin, 3, 3, 3].
    `into`: 1..8 tensors (each [cout/n, Cin, 3,3,3], or [.., 3,3] with planar2d) the gradient is ACCUMULATED into in place
    (parameters' .grad, one per stacked sibling conv); returns None."""
    _need_gpu(x, g, *(into or []))
    B, Cin, D, H, W = x.shape
    lib = load_library()
    n = lib.ragmi_conv3d_k3_wgrad_workspace_elems(B, Cin, cout, D, H, W)
    if n < 0:
        check(-2, "conv3d_k3_wgrad_workspace_elems")
    ws = torch.empty((n,), device=x.device, dtype=torch.float32)
    if into is None:
        dw = torch.empty((cout, Cin, 3, 3) if planar2d else (cout, Cin, 3, 3, 3), device=x.device, dtype=torch.float32)
        dsts, acc = [dw], 0
    else:
        want = (cout // len(into), Cin, 3, 3) if planar2d else (cout // len(into), Cin, 3, 3, 3)
        if any(tuple(t.shape) != want or not t.is_contiguous() for t in into):
            raise ValueError(f"conv3d_k3_wgrad: every destination must be a contiguous {want} tensor")
        dw, dsts, acc = None, list(into), 1
    arr = (ctypes.c_void_p * len(dsts))(*[t.data_ptr() for t in dsts])
    check(lib.ragmi_conv3d_k3_wgrad(x.data_ptr(), _planes(x), g.data_ptr(), _planes(g), g_ch0, arr, len(dsts), acc, int(planar2d),
                                    ws.data_ptr(), B, Cin, cout, D, H, W, _stream()), "conv3d_k3_wgrad")
    return dw


def conv3d_k1_wgrad(x: torch.Tensor, g: torch.Tensor, cout: int, g_ch0: int = 0, into: Optional[torch.Tensor] = None) -> torch.Tensor:
    """dW[cout, Cin] of a 1x1x1 conv; `into`: accumulate into this contiguous tensor of cout*Cin elements instead."""
    _need_gpu(x, g, into)
    B, Cin = x.shape[:2]
    if into is not None and (into.numel() != cout * Cin or not into.is_contiguous()):
        raise ValueError("conv3d_k1_wgrad: destination must be contiguous with cout*Cin elements")
    dw = into if into is not None else torch.zeros((cout, Cin), device=x.device, dtype=torch.float32)
    check(load_library().ragmi_conv3d_k1_wgrad(x.data_ptr(), _planes(x), g.data_ptr(), _planes(g), g_ch0, dw.data_ptr(), B, Cin, cout,
                                               _vol(x), _stream()), "conv3d_k1_wgrad")
    return dw


def trilinear3d_bwd(dy: torch.Tensor, in_size: Sequence[int], align_corners: bool) -> torch.Tensor:
    """adjoint of trilinear3d: gradient w.r.t. the [B,C,*in_size] input."""
    _need_gpu(dy)
    dy = dy.contiguous()
    B, C, Do, Ho, Wo = dy.shape
    Di, Hi, Wi = [int(v) for v in in_size]
    dx = torch.empty((B, C, Di, Hi, Wi), device=dy.device, dtype=torch.float32)
    check(load_library().ragmi_trilinear3d_bwd(dy.data_ptr(), dx.data_ptr(), B, C, Di, Hi, Wi, Do, Ho, Wo, int(bool(align_corners)),
                                               _stream()), "trilinear3d_bwd")
    return dx


def costvol_bwd(dcost: torch.Tensor):
    """adjoint of costvol: (dleft, dright) [B,C,h,w] from dcost [B,2C,d,h,w]."""
    _need_gpu(dcost)
    dcost = dcost.contiguous()
    B, C2, d, h, w = dcost.shape
    dl = torch.empty((B, C2 // 2, h, w), device=dcost.device, dtype=torch.float32)
    dr = torch.empty_like(dl)
    check(load_library().ragmi_costvol_bwd(dcost.data_ptr(), dl.data_ptr(), dr.data_ptr(), B, C2 // 2, d, h, w, _stream()), "costvol_bwd")
    return dl, dr


def disp_softargmin_bwd(cost: torch.Tensor, dout: torch.Tensor, maxdisp: int) -> torch.Tensor:
    """adjoint of disp_softargmin: dcost with cost's shape."""
    _need_gpu(cost, dout)
    shape = cost.shape
    c4 = (cost[:, 0] if cost.dim() == 5 else cost).contiguous()
    B, d, h, w = c4.shape
    dout = dout.contiguous()
    dc = torch.zeros_like(c4)
    check(load_library().ragmi_disp_softargmin_bwd(c4.data_ptr(), dout.data_ptr(), dc.data_ptr(), B, d, h, w, int(maxdisp), 3 * h, 3 * w,
                                                   _stream()), "disp_softargmin_bwd")
    return dc.reshape(shape)


def conv2d_k3_strided_dgrad(g: torch.Tensor, weight: torch.Tensor, in_hw: Sequence[int], stride: int) -> torch.Tensor:
    """data gradient of the strided 2-D stem conv: g[B,Cout,Ho,Wo] -> dx[B,Cin,H,W]."""
    _need_gpu(g, weight)
    g, w = g.contiguous(), weight.detach().contiguous()
    B, Cout = g.shape[:2]
    Cin, (H, W) = w.shape[1], [int(v) for v in in_hw]
    if tuple(g.shape[2:]) != ((H - 1) // stride + 1, (W - 1) // stride + 1):
        raise ValueError("conv2d_k3_strided_dgrad: gradient size does not match the input size / stride")
    dx = torch.empty((B, Cin, H, W), device=g.device, dtype=torch.float32)
    check(load_library().ragmi_conv2d_k3_strided_dgrad(g.data_ptr(), w.data_ptr(), dx.data_ptr(), B, Cin, Cout, H, W, int(stride),
                                                       _stream()), "conv2d_k3_strided_dgrad")
    return dx


def conv2d_k3_strided_wgrad(x: torch.Tensor, g: torch.Tensor, stride: int, into: Optional[torch.Tensor] = None) -> torch.Tensor:
    """weight gradient [Cout,Cin,3,3] of the strided 2-D stem conv; `into`: accumulate into this tensor instead."""
    _need_gpu(x, g, into)
    x, g = x.contiguous(), g.contiguous()
    B, Cin, H, W = x.shape
    Cout = g.shape[1]
    if into is not None and (tuple(into.shape) != (Cout, Cin, 3, 3) or not into.is_contiguous()):
        raise ValueError("conv2d_k3_strided_wgrad: destination must be a contiguous [Cout, Cin, 3, 3] tensor")
    dw = into if into is not None else torch.zeros((Cout, Cin, 3, 3), device=x.device, dtype=torch.float32)
    check(load_library().ragmi_conv2d_k3_strided_wgrad(x.data_ptr(), g.data_ptr(), dw.data_ptr(), B, Cin, Cout, H, W, int(stride),
                                                       _stream()), "conv2d_k3_strided_wgrad")
    return dw


def disparity_regression_bwd(dout: torch.Tensor, maxdisp: int) -> torch.Tensor:
    """adjoint of disparity_regression: dprob[B,maxdisp,H,W]."""
    _need_gpu(dout)
    dout = dout.contiguous()
    B, H, W = dout.shape
    dp = torch.empty((B, int(maxdisp), H, W), device=dout.device, dtype=torch.float32)
    check(load_library().ragmi_disparity_regression_bwd(dout.data_ptr(), dp.data_ptr(), B, int(maxdisp), H, W, _stream()),
          "disparity_regression_bwd")
    return dp



def conv3d_k3_uses_x3(cin: int, cout: int, B: int, D: int, H: int, W: int, nset: int = 1, has_res: bool = False, ntail: int = 0,
                      dtype: torch.dtype = torch.float32) -> bool:
    """True when conv3d_k3 (nset=1) / conv3d_k3_dual (nset=2, cin = both inputs) runs this shape on a split-operand (16-bit matrix core) kernel under the
    current precision setting."""
    return bool(load_library().ragmi_conv3d_k3_uses_x3(cin, cout, B, D, H, W, nset, int(has_res), ntail, _conv_dt(_DT[dtype])))


def sgd_clip_step(param: torch.Tensor, grad: torch.Tensor, buf: torch.Tensor, lr: float, momentum: float, weight_decay: float,
                  max_norm: float, first_step: bool, workspace: Optional[torch.Tensor] = None,
                  norm_out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """clip_grad_norm_ + SGD step over flat fp32 buffers in place (ragmi_sgd_clip_step); returns the 1-element total-norm tensor."""
    _need_gpu(param, grad, buf, workspace, norm_out)
    for t in (param, grad, buf):
        if t.dim() != 1 or t.dtype != torch.float32 or not t.is_contiguous() or t.numel() != param.numel():
            raise RuntimeError("sgd_clip_step: param / grad / momentum buffers must be flat contiguous fp32 of equal length")
    lib = load_library()
    if workspace is None:
        workspace = torch.empty((lib.ragmi_sgd_workspace_bytes() // 4,), device=param.device, dtype=torch.float32)
    if norm_out is None:
        norm_out = torch.empty((1,), device=param.device, dtype=torch.float32)
    check(lib.ragmi_sgd_clip_step(param.data_ptr(), grad.data_ptr(), buf.data_ptr(), param.numel(), float(lr), float(momentum),
                                  float(weight_decay), float(max_norm), int(first_step), workspace.data_ptr(), norm_out.data_ptr(),
                                  _stream()), "sgd_clip_step")
    return norm_out
