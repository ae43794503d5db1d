"""CPU-side tests (no GPU): the C-ABI library loads and exports every symbol the header
declares, the host mirror keeps the reference's interface / state_dict layout, and the
product path refuses to run without the GPU (no silent fallback)."""
import os
import re
import subprocess
import sys

import numpy as np
import pytest
import torch

from conftest import ROOT, load_golden, split_sd
from oracle import matching_oracle as O


@pytest.fixture(scope="session")
def built_lib():
    import rag_amd
    if not os.path.exists(rag_amd.lib_path()):
        subprocess.run([sys.executable, "-c", "import __graft_entry__ as g; g.build()"], cwd=ROOT, check=True)
    return rag_amd.load_library()


def test_library_exports_every_declared_symbol(built_lib):
    header = open(os.path.join(ROOT, "include", "rag_amd.h")).read()
    declared = set(re.findall(r"\b(ragmi_[a-z0-9_]+)\s*\(", header))
    assert len(declared) >= 11
    from rag_amd._lib import SIGNATURES
    assert declared == set(SIGNATURES), declared ^ set(SIGNATURES)
    for name in declared:
        assert hasattr(built_lib, name), name
    assert built_lib.ragmi_version() >= 100
    # fp32-MFMA section (groups x chunks x 7 VGPRs x 64 lanes) + bf16x3 fragments (1 cog x 21 K-slices x hi/lo x 64 lanes x 4 words)
    # fp32-MFMA section | bf16 fragments | scaled fp16 fragments | per-output-channel multipliers (one 16-channel block)
    assert built_lib.ragmi_conv3d_k3_packed_elems(12, 24) == 3 * 6 * 7 * 64 + 2 * (1 * 21 * 2 * 64 * 4) + 16


def test_abi_rejects_bad_arguments_without_gpu(built_lib):
    # argument validation happens before any launch, so it is checkable on CPU
    assert built_lib.ragmi_costvol_fwd(None, None, None, 1, 12, 8, 4, 4, 0, None) == -1
    assert b"null" in built_lib.ragmi_last_error()
    assert built_lib.ragmi_conv3d_k3_pack(None, None, 4, 4, 0, None) == -1


def test_ops_refuse_cpu_tensors(built_lib):
    import rag_amd
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        rag_amd.ops.costvol(torch.zeros(1, 12, 4, 4), torch.zeros(1, 12, 4, 4), 24)
    net = rag_amd.MatchingNet(rag_amd.ALL_SKIP_GENOTYPE, 24).eval()
    with torch.no_grad(), pytest.raises(RuntimeError, match="no CPU fallback"):
        net(torch.zeros(1, 12, 8, 12), torch.zeros(1, 12, 8, 12))


def test_missing_library_fails_loudly(monkeypatch):
    import rag_amd._lib as L
    monkeypatch.setattr(L, "_LIB", None)
    monkeypatch.setenv("RAG_AMD_LIB", "/nonexistent/librag_amd.so")
    with pytest.raises(RuntimeError, match="HIP library not found"):
        L.load_library()


def test_product_path_never_imports_oracle():
    pkg = os.path.join(ROOT, "rag_amd")
    for dirpath, _dirs, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp")):
                src = open(os.path.join(dirpath, f)).read()
                assert "oracle" not in src.replace("no oracle", ""), f"{f} mentions the oracle"


@pytest.mark.parametrize("fixture", ["g5_forward_conv_48x96_d48", "g5_forward_unsorted_36x60_d24", "g5_forward_skip_48x72_d24"])
def test_state_dict_layout_matches_reference(fixture):
    """Matching-Net keys and shapes of rag_amd.MatchingNet == the reference Network's (checkpoint drop-in)."""
    import rag_amd
    g = load_golden(fixture)
    rows = g["rows"]
    ref = {k: tuple(v.shape) for k, v in split_sd(g).items()
           if k.split(".")[0] in ("stem3d0", "stem3d1", "cells_3d", "last_3_3d", "last_6_3d", "last_12_3d")}
    net = rag_amd.MatchingNet(rag_amd.Genotype(rows, None, rows, None), maxdisp=int(g["maxdisp"]))
    mine = {k: tuple(v.shape) for k, v in net.state_dict().items()}
    assert mine == ref
    net.load_state_dict({k: v for k, v in split_sd(g).items() if k in ref}, strict=True)


@pytest.mark.parametrize("rows", [O.ALL_CONV, O.ALL_SKIP, np.array([[0, 1], [1, 0], [3, 0], [2, 1], [8, 1], [6, 0]]),
                                  np.array([[1, 1], [0, 0], [4, 1], [2, 0], [7, 1], [8, 1]])])
def test_cell_positional_op_pairing_matches_oracle(rows):
    import rag_amd
    cell = rag_amd.Cell_3d(3, 3, 4, 4, rag_amd.Genotype(rows, None, rows, None), 4, 0)
    plan = O.resolve_cell_ops(rows)
    contribs = cell._contributions()
    for step, lst in enumerate(plan):
        mine = contribs[2 + step]
        assert [j for (j, _k, _t) in lst] == [j for (j, _op) in mine]
        for (j, k, op_type), (_j, op) in zip(lst, mine):
            assert op is cell._ops[k]
            assert isinstance(op, rag_amd.ConvBR_3d) == (op_type == 1)


def test_reference_interface_names():
    import rag_amd
    net = rag_amd.MatchingNet(rag_amd.ALL_CONV_GENOTYPE)
    assert net.maxdisp == 192 and isinstance(net.disp, rag_amd.Disp)
    for attr in ("stem3d0", "stem3d1", "cells_3d", "last_3_3d", "last_6_3d", "last_12_3d", "matching", "search_matching"):
        assert hasattr(net, attr)
    assert set(net.arch_init) == {"stem_3d0", "stem_3d1", "last_3_3d", "last_6_3d", "last_12_3d"} | {f"cell_3d{i}" for i in range(8)}
    c = rag_amd.Cell_3d(3, 3, 4, 8, rag_amd.ALL_CONV_GENOTYPE, 16, -1)
    assert (c.C_in, c.C_out, c.C_prev, c.C_prev_prev, c.scale) == (48, 16, 24, 12, 0.5)
    assert c.scale_dimension(64, 0.5) == 32 and c.scale_dimension(7, 0.5) == 4 and c.scale_dimension(3, 2) == 5
    m = rag_amd.ConvBR_3d(12, 1, 3, 1, 1, bn=False, relu=False)
    assert "bn.weight" in m.state_dict()           # bn constructed even when unused (operations_3d.py:38)
    assert list(rag_amd.OPS_3d) == ["skip_connect_3d", "3d_conv_3x3"] == rag_amd.PRIMITIVES_3D


def test_training_mode_runs_on_hip_or_fails_loudly():
    """Train-mode BN / autograd route to the HIP autograd Functions; on a machine without a GPU they raise
    (no PyTorch emulation), and folded BN parameters are never handed out for a train-mode unit."""
    import rag_amd
    from rag_amd import autograd as ag
    m = rag_amd.ConvBR_3d(4, 4, 3, 1, 1)
    m.train()
    assert m.autograd_mode(torch.zeros(1))                      # batch statistics -> training composition
    with pytest.raises(RuntimeError):
        m.prepared()
    m.eval()
    x = torch.zeros(1, 4, 2, 2, 2, requires_grad=True)
    assert m.autograd_mode(x)
    with torch.no_grad():
        assert not m.autograd_mode(x)
    if not torch.cuda.is_available():
        with pytest.raises((RuntimeError, OSError)):            # CPU tensor: "rag_amd ops run on the MI355X only"
            m(x)
    for fn in (ag.ConvBRFn, ag.StridedStemFn, ag.TrilinearFn, ag.CostVolFn, ag.DispFn, ag.DispRegFn, ag.AddFn):
        assert issubclass(fn, torch.autograd.Function)


# ------------------------------------------------------------------ Network (growth loop surface)
def _blob():
    import json
    return json.loads(bytes(load_golden("g8_growth_api")["blob"]).decode())


def test_network_state_dict_matches_reference_network():
    """Full-Network keys and shapes (Feature Net + Matching Net) == the reference Network's; strict load."""
    import rag_amd
    g = load_golden("g5_forward_unsorted_36x60_d24")
    rows = g["rows"]
    ref = split_sd(g)
    net = rag_amd.Network(rag_amd.Genotype(rows, None, rows, None), "cpu", maxdisp=int(g["maxdisp"]))
    assert {k: tuple(v.shape) for k, v in net.state_dict().items()} == {k: tuple(v.shape) for k, v in ref.items()}
    net.load_state_dict(ref, strict=True)
    assert sorted(net.state_dict().keys()) == sorted(ref.keys())


def test_network_growth_api_matches_reference_bookkeeping():
    """expand / get_new_model / select replayed against the reference's own run (tests/golden/make_golden.py G8)."""
    import rag_amd
    blob = _blob()
    mixed = np.array([[0, 1], [1, 0], [3, 0], [2, 1], [8, 1], [6, 0]])
    geno = lambda r: rag_amd.Genotype(r, None, r, None)  # noqa: E731
    net = rag_amd.Network(geno(O.ALL_CONV), "cpu")
    assert sorted(net.state_dict().keys()) == blob["keys_initial"]
    assert {k: list(v) for k, v in net.arch_init.items()} == blob["arch_init"]
    net.expand(1, geno(mixed), "cpu")
    assert sorted(net.state_dict().keys()) == blob["keys_expanded"]
    assert [[round(float(x), 6) for x in p] for p in net.p] == [[round(x, 6) for x in p] for p in blob["p_after_expand"]]
    assert {k: [int(i) for i in v] for k, v in net.new_models.items()} == blob["new_models"]
    for k, p in enumerate(net.p):
        if k in blob["winners"]:
            p[-1] = 0.9
    best = net.select(1)
    as_int = lambda d: {k: [int(i) for i in v] for k, v in d.items()}  # noqa: E731
    assert as_int(best) == blob["best_archi"]
    assert as_int(net.model_to_train) == blob["model_to_train"]
    assert {k: int(v) for k, v in net.length.items()} == blob["length"]
    assert sorted(net.state_dict().keys()) == blob["keys_selected"]
    # get_param / modify_param address exactly the units to train
    net.modify_param({k: list(range(net.length[k])) for k in net.length if k in net.new_models}, False)
    net.modify_param(net.model_to_train, True)
    trainable = {n for n, p in net.named_parameters() if p.requires_grad}
    n_params = sum(len(list(g["params"])) for g in net.get_param(net.model_to_train))
    assert n_params == len(trainable) and any(n.startswith("stem2d1.1.") for n in trainable)
    assert all(not n.startswith("stem2d0.") for n in trainable) and any(n.startswith("last_3_3d.1.") for n in trainable)
    net.expand(2, geno(O.ALL_SKIP), "cpu")
    best2 = net.select(2)
    assert as_int(best2) == blob["best_archi_round2"]
    assert {k: int(v) for k, v in net.length.items()} == blob["length_round2"]
    assert sorted(net.state_dict().keys()) == blob["keys_round2"]


# ------------------------------------------------------------------ edge cases / argument validation
def test_abi_rejects_unbuilt_dtype_and_bad_sizes(built_lib):
    """Validation runs before any launch, so it is checkable without a GPU (pointers are never dereferenced)."""
    import ctypes
    fake = ctypes.c_void_p(0x1000)
    # unknown dtype code -> RAGMI_EUNSUPPORTED (-2)
    assert built_lib.ragmi_costvol_fwd(fake, fake, fake, 1, 12, 8, 4, 4, 7, None) == -2
    assert b"dtype" in built_lib.ragmi_last_error()
    assert built_lib.ragmi_trilinear3d_fwd(fake, fake, 1, 1, 2, 2, 2, 4, 4, 4, 1, 9, None) == -2
    # non-positive sizes -> RAGMI_EINVAL (-1)
    assert built_lib.ragmi_costvol_fwd(fake, fake, fake, 0, 12, 8, 4, 4, 0, None) == -1
    assert built_lib.ragmi_conv3d_k1_fwd(fake, 0, fake, None, None, 0, fake, 0, 0, 1, 0, 4, 8, 0, None) == -1
    # scale without shift
    assert built_lib.ragmi_conv3d_k1_fwd(fake, 0, fake, fake, None, 0, fake, 0, 0, 1, 4, 4, 8, 0, None) == -1
    # too many output channels for one call / dual needs CinA % 4 == 0
    assert built_lib.ragmi_conv3d_k3_fwd(fake, 0, fake, None, None, 0, fake, 0, None, None, 0, None, 1, 4, 68, 2, 2, 2, 0, None) == -2
    assert built_lib.ragmi_conv3d_k3_dual_fwd(fake, 0, 3, fake, None, None, 4, fake, None, None, 0, fake, 0, None, None, 0, None,
                                              1, 4, 2, 2, 2, 0, None) == -1
    # small-Cout form only for Cout <= 2
    assert built_lib.ragmi_conv3d_k3_small_fwd(fake, 0, fake, None, None, 0, fake, 0, 0, None, 0, 0, 1, 4, 3, 2, 2, 2, 0, None) == -2


def test_cost_volume_maxdisp_semantics_match_reference():
    """d = int(maxdisp / 3) like rag_model.py:376-377 (maxdisp need not be a multiple of 3); disparities past the width
    stay zero (the reference's slice assignment is empty there)."""
    L, R = torch.randn(1, 2, 3, 4), torch.randn(1, 2, 3, 4)
    assert O.cost_volume(L, R, 25).shape == (1, 4, 8, 3, 4)
    c = O.cost_volume(L, R, 24)
    assert float(c[:, :, 4:].abs().max()) == 0.0 and torch.equal(c[:, :2, 0], L) and torch.equal(c[:, 2:, 1, :, 1:], R[..., :-1])


def test_oracle_rejects_shapes_the_reference_crashes_on():
    rows = O.ALL_SKIP
    sd = O.random_matching_state_dict(rows)
    with pytest.raises(ValueError):     # h = 10 is not a multiple of 4 -> reference: UnboundLocalError at rag_model.py:360-366
        O.matching(torch.zeros(1, 24, 4, 10, 8), sd, rows)


# ------------------------------------------------------------------ checkpoint round trip
def _grown_network():
    import rag_amd
    from rag_amd.modules import Genotype
    mixed = np.array([[0, 1], [1, 0], [3, 0], [2, 1], [8, 1], [6, 0]])
    torch.manual_seed(5)
    net = rag_amd.Network(rag_amd.ALL_CONV_GENOTYPE, "cpu", maxdisp=48)
    archis = [net.arch_init]
    net.expand(1, Genotype(mixed, None, mixed, None), "cpu")
    for k in (1, 5, 9, 12):                      # the candidate wins in four layers
        net.p[k][-1] = 0.9
    archis.append(net.select(1))
    return net, archis


def test_checkpoint_round_trip_rebuilds_grown_model(tmp_path):
    from rag_amd import checkpoint as ck
    net, archis = _grown_network()
    path = tmp_path / "checkpoint_task1.ckpt"
    ck.save_checkpoint(path, net, archis, task=1)
    raw = torch.load(path, map_location="cpu", weights_only=False)
    assert {"task", "model", "optimizer"} <= set(raw)                      # the reference's keys (run.py:194) are intact
    net2, archis2 = ck.load_checkpoint(str(path), device="cpu")
    sd1, sd2 = net.state_dict(), net2.state_dict()
    assert list(sd1) == list(sd2)
    assert all(torch.equal(sd1[k], sd2[k]) for k in sd1)
    assert archis2 == [{k: [int(v) for v in vs] for k, vs in a.items()} for a in archis]
    assert net2.length == net.length and not net2.training
    # grown cell units were rebuilt from THEIR genotype (identity ops at the same positions)
    for name in ("cell_3d1", "cell_2d2"):
        for u1, u2 in zip(net._units(name), net2._units(name)):
            assert [type(o).__name__ for o in u1._ops] == [type(o).__name__ for o in u2._ops]
    serve = ck.MultiTaskStereo(net2, archis2)
    assert serve.n_tasks == 2
    with pytest.raises(IndexError):
        serve(None, None, 2)


def test_reference_style_checkpoint_needs_genotypes_and_checks_them():
    import rag_amd
    from rag_amd import checkpoint as ck
    net, archis = _grown_network()
    ref_style = {"task": 1, "model": net.state_dict(), "optimizer": None}       # what run.py:194 writes
    with pytest.raises(ValueError, match="no genotypes"):
        ck.load_checkpoint(ref_style, device="cpu")
    with pytest.raises(ValueError, match="wrong genotype"):                      # all-conv for every unit: the mixed units disagree
        ck.load_checkpoint(ref_style, device="cpu", genotypes=rag_amd.ALL_CONV_GENOTYPE)
    net2, archis2 = ck.load_checkpoint(ref_style, device="cpu", genotypes=ck.unit_genotypes(net), archis=archis)
    assert list(net2.state_dict()) == list(net.state_dict()) and len(archis2) == 2
    bad = [dict(archis[0], stem_3d0=[7])]
    with pytest.raises(ValueError, match="archis"):
        ck.load_checkpoint(ref_style, device="cpu", genotypes=ck.unit_genotypes(net), archis=bad)


# ------------------------------------------------------------------ MdeNAS supernet
def test_supernet_state_dict_matches_reference_and_genotype_parse():
    import rag_amd
    g = load_golden("g9_supernet")
    sd = split_sd(g)
    net = rag_amd.BasicNetwork(device="cpu", maxdisp=48)
    assert sorted(net.state_dict().keys()) == sorted(sd.keys())
    net.load_state_dict(sd, strict=True)
    assert sum(1 for _ in net.parameters()) == int(g["n_params"])
    assert net.matching.cells[0]._ops[0] is None and net.matching.cells[0]._ops[1] is not None      # no s0 edge in the first cell
    assert (net.num_edges, net.num_ops) == (9, 2) and net.p["normal"].shape == (9, 2)
    # genotype(): per step the two edges with the largest non-identity probability, each with its argmax op
    net.p["reduce"] = torch.tensor([[0.9, 0.1], [0.2, 0.8], [0.5, 0.5], [0.1, 0.9], [0.3, 0.7], [0.6, 0.4], [0.4, 0.6], [0.45, 0.55],
                                    [0.2, 0.8]]).log()
    rows = net.genotype().reduce.tolist()
    assert rows == [[1, 1], [0, 0], [3, 1], [4, 1], [8, 1], [6, 1]]
    twin = net.new()                             # mdenas_basicmodel.py:70-74: fresh weights, copied probabilities
    assert torch.equal(twin.p["reduce"], net.p["reduce"]) and twin.p["reduce"] is not net.p["reduce"]


def test_conv_precision_names_and_aliases():
    """"f16x3" is the canonical name of the split form (scaled fp16 halves since round 3); "bf16x3" (round 2's name) and "split"
    are aliases of it; anything else is refused; the context manager restores the previous setting."""
    from rag_amd import ops
    old = ops.get_conv_precision()
    try:
        assert ops.set_conv_precision("fp32") == old
        ops.set_conv_precision("bf16x3")
        assert ops.get_conv_precision() == "f16x3"
        with ops.conv_precision("fp32"):
            assert ops.get_conv_precision() == "fp32"
            with ops.conv_precision("split"):
                assert ops.get_conv_precision() == "f16x3"
            assert ops.get_conv_precision() == "fp32"
        assert ops.get_conv_precision() == "f16x3"
        assert ops.set_conv_precision("f16x3") == "f16x3"
        with pytest.raises(ValueError):
            ops.set_conv_precision("fp16")
    finally:
        ops.set_conv_precision(old)


def test_training_defaults_to_strict_fp32():
    """rag.py:204-216 trains in fp32: forward_backward / train_step / GraphedTrainStep default to the strict contract, the split
    form is opt-in by argument."""
    import inspect
    from rag_amd import train
    assert train.TRAIN_PRECISION == "fp32"
    for fn in (train.forward_backward, train.train_step, train.GraphedTrainStep.__init__):
        assert inspect.signature(fn).parameters["precision"].default is None      # None -> TRAIN_PRECISION


def test_g4_layout_helpers_round_trip():
    import rag_amd
    t = torch.arange(2 * 8 * 3 * 4 * 5, dtype=torch.float32).view(2, 8, 3, 4, 5)
    g = rag_amd.ops.to_g4(t)
    assert g.shape == t.shape and torch.equal(rag_amd.ops.from_g4(g), t)
    # group 1, voxel (d, h, w) = (1, 2, 3): its four channels are contiguous in the buffer
    flat = g.reshape(2, -1)
    base = ((1 * 3 + 1) * 4 + 2) * 5 + 3
    assert torch.equal(flat[0, 4 * base:4 * base + 4], t[0, 4:8, 1, 2, 3])
