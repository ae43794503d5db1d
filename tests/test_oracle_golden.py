"""Pin the CPU oracle against fixtures generated from the reference itself
(tests/golden/make_golden.py).  CPU-only; this is what makes parity 'pinned'."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from conftest import load_golden, split_sd
from oracle import matching_oracle as O

TOL = dict(rtol=1e-5, atol=1e-5)


@pytest.mark.parametrize("tag", ["a", "b"])
def test_g1_cost_volume_bit_exact(tag):
    g = load_golden(f"g1_costvol_{tag}")
    cost = O.cost_volume(torch.from_numpy(g["left_fea"]), torch.from_numpy(g["right_fea"]), int(g["maxdisp"]))
    assert cost.shape == g["cost"].shape
    assert np.array_equal(cost.numpy(), g["cost"])          # pure copy: bit exact
    # zero fill for x < i in BOTH halves (rag_model.py:376, 379-380)
    for i in range(1, cost.shape[2]):
        assert float(cost[:, :, i, :, :min(i, cost.shape[4])].abs().max()) == 0.0


def test_g2_disp_and_regression():
    g = load_golden("g2_disp")
    out = O.disp_head(torch.from_numpy(g["x"]), int(g["maxdisp"]))
    np.testing.assert_allclose(out.numpy(), g["out"], **TOL)
    reg = O.disparity_regression(torch.from_numpy(g["prob"]), int(g["maxdisp"]))
    np.testing.assert_allclose(reg.numpy(), g["reg"], **TOL)
    out2 = O.disp_head(torch.from_numpy(g["x2"]), int(g["maxdisp2"]))
    np.testing.assert_allclose(out2.numpy(), g["out2"], **TOL)


@pytest.mark.parametrize("name", ["k3", "k3_wide", "k1", "k3_nobn"])
def test_g3_convbr(name):
    g = load_golden("g3_convbr")
    cin, cout, k, pad, bn, relu = [int(v) for v in g[f"{name}::cfg"]]
    sd = split_sd(g, f"{name}::sd::")
    x = torch.from_numpy(g[f"{name}::x"])
    y = O.conv_br_3d(x, sd, "", padding=pad, bn=bool(bn), relu=bool(relu), training=False)
    np.testing.assert_allclose(y.numpy(), g[f"{name}::y_eval"], **TOL)
    yt = O.conv_br_3d(x, sd, "", padding=pad, bn=bool(bn), relu=bool(relu), training=True)
    np.testing.assert_allclose(yt.numpy(), g[f"{name}::y_train"], rtol=1e-4, atol=1e-5)


@pytest.mark.parametrize("name", ["same_conv", "same_unsorted", "same_deep", "down_even", "down_odd", "up", "skip"])
def test_g4_cell3d(name):
    g = load_golden("g4_cell3d")
    pp, p, fm, du = [int(v) for v in g[f"{name}::cfg"]]
    sd = split_sd(g, f"{name}::sd::")
    prev, cat = O.cell_3d(torch.from_numpy(g[f"{name}::s0"]), torch.from_numpy(g[f"{name}::s1"]), sd, "",
                          g[f"{name}::rows"], fm, du)
    np.testing.assert_allclose(cat.numpy(), g[f"{name}::out"], **TOL)


def test_positional_op_quirk():
    # rows [[0,1],[1,0],[3,0],[2,1],[8,1],[6,0]] -> branch 2 Identity, 3 Conv, 6 Conv, 8 Identity
    plan = O.resolve_cell_ops(np.array([[0, 1], [1, 0], [3, 0], [2, 1], [8, 1], [6, 0]]))
    flat = {(step, j): op for step, s in enumerate(plan) for (j, k, op) in s}
    assert flat == {(0, 0): 1, (0, 1): 0, (1, 0): 0, (1, 1): 1, (2, 1): 1, (2, 3): 0}


@pytest.mark.parametrize("name", ["conv_48x96_d48", "unsorted_36x60_d24", "skip_48x72_d24"])
def test_g5_matching_and_forward(name):
    g = load_golden("g5_forward_" + name)
    sd = split_sd(g)
    lf, rf = torch.from_numpy(g["left_fea"]), torch.from_numpy(g["right_fea"])
    disp, mid = O.matching_net_forward(lf, rf, sd, g["rows"], int(g["maxdisp"]), return_intermediates=True)
    np.testing.assert_allclose(mid["mat"].numpy(), g["mat"], rtol=1e-4, atol=1e-4)
    np.testing.assert_allclose(disp.numpy(), g["disp"], rtol=1e-4, atol=1e-4)
    assert O.epe(disp, torch.from_numpy(g["disp"])) < 1e-5


def test_g7_plumbing_config():
    """BASELINE configs[0]: 256x512 -> padded 264x516, D=48, B=1."""
    g = load_golden("g7_plumbing_264x516_d48")
    sd = split_sd(g)
    torch.set_num_threads(8)
    disp = O.matching_net_forward(torch.from_numpy(g["left_fea"]), torch.from_numpy(g["right_fea"]), sd,
                                  g["rows"], int(g["maxdisp"]))
    y0, y1, x0, x1 = [int(v) for v in g["crop_box"]]
    np.testing.assert_allclose(disp[:, y0:y1, x0:x1].numpy(), g["disp_crop"], rtol=1e-4, atol=1e-4)
    np.testing.assert_allclose(disp.double().mean(dim=2).numpy()[0], g["disp_row_means"], rtol=1e-5, atol=1e-5)
    assert abs(float(disp.double().mean()) - float(g["disp_mean"])) < 1e-5


def test_random_state_dict_matches_reference_layout():
    """Key names/shapes of the oracle's random state_dict == the reference's Matching-Net keys."""
    g = load_golden("g5_forward_unsorted_36x60_d24")
    ref = {k: v.shape for k, v in split_sd(g).items()
           if any(s in k for s in ("stem3d", "cells_3d", "last_3_3d", "last_6_3d", "last_12_3d"))}
    mine = {k: v.shape for k, v in O.random_matching_state_dict(g["rows"]).items()}
    assert mine == ref


# ------------------------------------------------ explicit index-math restatements
@pytest.mark.parametrize("align", [True, False])
@pytest.mark.parametrize("shape,size", [((2, 3, 8, 12, 20), (4, 6, 10)), ((1, 2, 7, 9, 13), (4, 5, 7)),
                                        ((1, 2, 3, 4, 5), (6, 8, 10)), ((1, 1, 8, 4, 8), (24, 12, 24)),
                                        ((1, 1, 5, 3, 4), (13, 9, 12)), ((1, 2, 16, 32, 26), (8, 16, 13))])
def test_trilinear_explicit_matches_aten(align, shape, size):
    x = torch.randn(shape, generator=torch.Generator().manual_seed(5))
    ref = F.interpolate(x, size, mode="trilinear", align_corners=align)
    mine = O.trilinear_explicit(x.numpy(), size, align)
    np.testing.assert_allclose(mine, ref.numpy(), rtol=1e-5, atol=1e-5)


def test_conv3d_explicit_matches_aten():
    g = torch.Generator().manual_seed(6)
    x = torch.randn((2, 5, 4, 6, 7), generator=g)
    w = torch.randn((3, 5, 3, 3, 3), generator=g)
    np.testing.assert_allclose(O.conv3d_explicit(x.numpy(), w.numpy(), 1), F.conv3d(x, w, padding=1).numpy(),
                               rtol=1e-4, atol=1e-4)
    w1 = torch.randn((4, 5, 1, 1, 1), generator=g)
    np.testing.assert_allclose(O.conv3d_explicit(x.numpy(), w1.numpy(), 0), F.conv3d(x, w1).numpy(),
                               rtol=1e-4, atol=1e-4)


def oracle_train_step(g, sd=None):
    """One training step of the Matching Net through the oracle (O.train_step) on a fixture-style dict of numpy arrays
    (stem3d0 is a 'reused' unit: BN in eval).  Returns (disp, loss, grads by parameter name incl. left_fea/right_fea)."""
    sd = split_sd(g) if sd is None else sd
    return O.train_step(torch.from_numpy(g["left_fea"]), torch.from_numpy(g["right_fea"]), torch.from_numpy(g["gt"]), sd,
                        g["rows"], int(g["maxdisp"]))


def test_g6_train_step():
    """fwd+bwd of the reference's training step (rag.py:155-219): disp, loss and parameter/feature gradients."""
    g = load_golden("g6_train_step")
    torch.set_num_threads(8)
    disp, loss, grads = oracle_train_step(g)
    np.testing.assert_allclose(disp.numpy(), g["disp"], rtol=1e-4, atol=1e-4)
    assert abs(loss - float(g["loss"])) < 1e-5
    checked = 0
    for k, ref in g.items():
        if not k.startswith("grad::") or "_2d" in k:      # Feature-Net gradients are outside the oracle
            continue
        got = grads[k[6:]].numpy()
        assert np.abs(got - ref).max() <= 1e-4 * max(1.0, np.abs(ref).max()), k
        checked += 1
    assert checked > 40


def _g10_rows(g):
    return lambda _layer, unit: g["rows_unit0"] if unit == 0 else g["rows_unit1"]


def test_g10_grown_model():
    """A grown model (expand -> select, rag_model.py:391-522, 709-845): the Matching-Net half of task 0 / task 1 on the
    selected model and of search_forward on the expanded supermodel with non-zero unit and head indices (rag_model.py:663-706)."""
    import json
    g = load_golden("g10_grown_model")
    blob = json.loads(bytes(g["blob"]).decode())
    maxdisp = int(g["maxdisp"])
    sel_sd, search_sd = split_sd(g, "selected::"), split_sd(g, "search::")
    for t in (0, 1):
        disp = O.matching_net_forward(torch.from_numpy(g[f"left_fea_t{t}"]), torch.from_numpy(g[f"right_fea_t{t}"]), sel_sd,
                                      _g10_rows(g), maxdisp, task_arch=blob[f"arch_t{t}"])
        np.testing.assert_allclose(disp.numpy(), g[f"disp_t{t}"], rtol=1e-4, atol=1e-4)
    for tag, sel, t in (("a", g["sel_a"], 1), ("b", g["sel_b"], 1), ("c", g["sel_a"], 0)):
        disp = O.matching_net_forward(torch.from_numpy(g[f"search_left_fea_{tag}"]), torch.from_numpy(g[f"search_right_fea_{tag}"]),
                                      search_sd, _g10_rows(g), maxdisp, selected_ops=sel, head_index=t)
        np.testing.assert_allclose(disp.numpy(), g[f"search_disp_{tag}"], rtol=1e-4, atol=1e-4)


@pytest.mark.parametrize("case", ["a", "b", "c", "d"])
def test_stereo_metrics_vs_reference_fixture(case):
    """g11: the reference's own EPE_metric / D1_metric / Thres_metric (utilstool/metrics.py:21-65) and the mask + loss of
    approaches/rag.py:418-430, run by tests/golden/make_golden.py.  a: three ordinary images; b: one image dropped by the 10 %
    rule; c: every image dropped (metrics 0); d: errors exactly on the thresholds (strict comparisons)."""
    g = load_golden("g11_metrics")
    m = O.stereo_metrics(torch.from_numpy(g[f"{case}::est"]), torch.from_numpy(g[f"{case}::gt"]), 192)
    got = np.array([m[k] for k in ("loss", "EPE", "D1", "Thres1", "Thres2", "Thres3")])
    np.testing.assert_allclose(got, g[f"{case}::scalars"], rtol=1e-6, atol=1e-7)


def test_stereo_metrics_known_answer():
    """Hand-computed case for the metrics restatement (kept beside the reference-generated fixture g11)."""
    gt = torch.tensor([[[10.0, 20.0, 0.0, 200.0]], [[300.0, 250.0, 5.0, 0.0]]])          # B=2, H=1, W=4, maxdisp 192
    est = torch.tensor([[[10.5, 24.0, 7.0, 100.0]], [[1.0, 2.0, 5.0, 9.0]]])
    # image 0: mask = [1,1,0,0], gt>0 = [1,1,0,1] -> ratio 0.75 (kept); errors 0.5, 4.0 (4 > 3 px and 20 % > 5 %)
    # image 1: mask = [0,0,1,0], gt>0 = [1,1,1,0] -> ratio 1/3 (kept); error 0
    m = O.stereo_metrics(est, gt, 192)
    assert abs(m["loss"] - (0.5 * 0.25 + 3.5 + 0.0) / 3) < 1e-6
    assert abs(m["EPE"] - ((0.5 + 4.0) / 2 + 0.0) / 2) < 1e-6
    assert abs(m["D1"] - (0.5 + 0.0) / 2) < 1e-6
    assert abs(m["Thres1"] - 0.25) < 1e-6 and abs(m["Thres3"] - 0.25) < 1e-6
    # 10 % rule: image with 1 of 20 positive gt pixels inside the mask is skipped
    gt2 = torch.full((1, 1, 20), 500.0)
    gt2[0, 0, 0] = 4.0
    assert O.stereo_metrics(torch.zeros((1, 1, 20)), gt2, 192)["EPE"] == 0.0
