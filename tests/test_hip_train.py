"""GPU parity tests of the training step (BASELINE config 5): forward AND backward of every autograd Function of
rag_amd/autograd.py against PyTorch-CPU autograd of the oracle on the same inputs, and the whole step against the
reference-generated fixture g6_train_step (approaches/rag.py:155-219: train-mode BN, one reused unit in eval,
smooth-L1 on the 0 < gt < maxdisp mask).

Tolerances: fp32 with a different accumulation order (and fp32 atomics in the reductions): gradients are compared
relative to the largest magnitude of each tensor, 2e-4 unless stated."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from conftest import load_golden, split_sd
from oracle import matching_oracle as O

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def gpu(x):
    return torch.as_tensor(x).to(DEV)


@pytest.fixture(scope="module")
def ra():
    import rag_amd
    rag_amd.load_library()
    return rag_amd


def gen(seed):
    return torch.Generator().manual_seed(seed)


def close(got, ref, tol=2e-4, what=""):
    got, ref = got.detach().cpu().double(), ref.detach().cpu().double()
    assert got.shape == ref.shape, (what, got.shape, ref.shape)
    err = float((got - ref).abs().max())
    bound = tol * max(1.0, float(ref.abs().max()))
    assert err <= bound, (what, err, bound)


# --------------------------------------------------------------------------- kernels
@pytest.mark.parametrize("shape", [(2, 5, 3, 7, 9), (1, 12, 8, 16, 32), (3, 4, 1, 5, 1027), (4, 12, 16, 32, 64)])
def test_bn_train_stats_and_affine(ra, shape):
    C = shape[1]
    x = torch.randn(shape, generator=gen(1)) * 2 + 0.5
    gm, bt = torch.rand(C, generator=gen(2)) + 0.5, torch.randn(C, generator=gen(3))
    rm, rv = torch.randn(C, generator=gen(4)), torch.rand(C, generator=gen(5)) + 0.5
    rm_g, rv_g, nbt = gpu(rm), gpu(rv), torch.zeros((), dtype=torch.int64, device=DEV)
    st = ra.ops.bn_train_stats(gpu(x), gpu(gm), gpu(bt), rm_g, rv_g, nbt, 0.1, 1e-5)
    yr = F.batch_norm(x, rm, rv, gm, bt, training=True, momentum=0.1, eps=1e-5)      # also updates rm / rv
    mean = x.mean(dim=(0, 2, 3, 4))
    var = x.var(dim=(0, 2, 3, 4), unbiased=False)
    close(st[0], mean, 1e-5, "mean")
    close(st[1], torch.rsqrt(var + 1e-5), 1e-5, "invstd")
    close(rm_g, rm, 1e-5, "running_mean")
    close(rv_g, rv, 1e-5, "running_var")
    assert int(nbt) == 1
    y = ra.ops.bn_act(gpu(x), st[2].contiguous(), st[3].contiguous(), False)
    close(y, yr, 1e-5, "y")
    sc, sh = torch.rand(C, generator=gen(2)) + 0.5, torch.randn(C, generator=gen(3))
    y = ra.ops.bn_act(gpu(x), gpu(sc), gpu(sh), True)
    close(y, F.relu(x * sc.view(1, -1, 1, 1, 1) + sh.view(1, -1, 1, 1, 1)), 1e-6)


@pytest.mark.parametrize("shape", [(2, 5, 6, 9, 13), (1, 4, 12, 24, 32), (3, 2, 1, 7, 5)])
@pytest.mark.parametrize("relu", [False, True])
def test_bn_fused_forward_and_adjoint(ra, shape, relu):
    """ragmi_bn_train_act_fwd / ragmi_bn_act_bwd (two launches each) against torch's train-mode batch_norm + relu and its autograd."""
    C = shape[1]
    x = (torch.randn(shape, generator=gen(1)) * 2 + 0.5).requires_grad_(True)
    gm = (torch.rand(C, generator=gen(2)) + 0.5).requires_grad_(True)
    bt = torch.randn(C, generator=gen(3)).requires_grad_(True)
    rm, rv = torch.randn(C, generator=gen(4)), torch.rand(C, generator=gen(5)) + 0.5
    rm_g, rv_g, nbt = gpu(rm), gpu(rv), torch.zeros((), dtype=torch.int64, device=DEV)
    y, st = ra.ops.bn_train_act(gpu(x.detach()), gpu(gm.detach()), gpu(bt.detach()), rm_g, rv_g, nbt, 0.1, 1e-5, relu)
    yr = F.batch_norm(x, rm, rv, gm, bt, training=True, momentum=0.1, eps=1e-5)
    yr = F.relu(yr) if relu else yr
    close(y, yr.detach(), 1e-5, "y")
    close(rm_g, rm, 1e-5, "running_mean")
    close(rv_g, rv, 1e-5, "running_var")
    assert int(nbt) == 1
    dy = torch.randn(shape, generator=gen(6))
    yr.backward(dy)
    dx, dg, db = ra.ops.bn_act_bwd(gpu(dy), gpu(x.detach()), st[2], st[3], relu, st[0], st[1], True)
    close(dx, x.grad, 2e-5, "dx")
    close(dg, gm.grad, 2e-5, "dgamma")
    close(db, bt.grad, 2e-5, "dbeta")
    # accumulate-into form adds to what is there
    tg, tb = torch.ones(C, device=DEV), torch.ones(C, device=DEV)
    ra.ops.bn_act_bwd(gpu(dy), gpu(x.detach()), st[2], st[3], relu, st[0], st[1], True, dgamma_into=tg, dbeta_into=tb)
    close(tg - 1, gm.grad, 2e-5, "dgamma +=")
    close(tb - 1, bt.grad, 2e-5, "dbeta +=")


def _convbr_case(ra, cin, cout, k, bn, relu, bn_training, shape, seed, ndim=3, train_params=True):
    """(module on the GPU, x, torch-CPU reference function)"""
    cls = ra.ConvBR_3d if ndim == 3 else ra.ConvBR_2d
    torch.manual_seed(seed)
    m = cls(cin, cout, k, 1, (k - 1) // 2, bn=bn, relu=relu)
    with torch.no_grad():
        m.bn.weight.copy_(torch.rand(cout) + 0.5)
        m.bn.bias.copy_(torch.randn(cout) * 0.1)
        m.bn.running_mean.copy_(torch.randn(cout) * 0.1)
        m.bn.running_var.copy_(torch.rand(cout) + 0.5)
    m.train(bn_training)
    x = torch.randn(shape, generator=gen(seed + 1))
    return m, x


def _ref_convbr(m, x, ndim=3):
    """PyTorch-CPU autograd of operations_3d.py:40-47 with a private copy of the module's parameters/buffers."""
    w = m.conv.weight.detach().cpu().clone().requires_grad_(True)
    g = m.bn.weight.detach().cpu().clone().requires_grad_(True)
    b = m.bn.bias.detach().cpu().clone().requires_grad_(True)
    rm, rv = m.bn.running_mean.detach().cpu().clone(), m.bn.running_var.detach().cpu().clone()
    xr = x.clone().requires_grad_(True)
    conv = F.conv3d if ndim == 3 else F.conv2d
    y = conv(xr, w, None, stride=m.conv.stride, padding=m.conv.padding)
    if m.use_bn:
        y = F.batch_norm(y, rm, rv, g, b, training=m.bn.training, momentum=0.1, eps=m.bn.eps)
    if m.relu:
        y = F.relu(y)
    return y, xr, w, g, b, rm, rv


@pytest.mark.parametrize("cin,cout,k,bn,relu,bn_training,shape", [
    (12, 12, 3, True, True, True, (2, 12, 5, 9, 33)),
    (24, 12, 3, True, True, True, (2, 24, 4, 8, 20)),
    (4, 4, 3, True, True, False, (1, 4, 6, 10, 40)),
    (16, 16, 3, True, True, True, (2, 16, 3, 7, 13)),
    (12, 1, 3, False, False, True, (2, 12, 5, 12, 40)),      # last_3_3d
    (12, 4, 1, True, True, True, (2, 12, 4, 6, 8)),
    (48, 24, 1, True, True, False, (1, 48, 4, 8, 26)),
    (24, 12, 1, True, False, True, (3, 24, 3, 5, 7)),
])
def test_convbr_fwd_bwd_vs_torch(ra, cin, cout, k, bn, relu, bn_training, shape):
    m, x = _convbr_case(ra, cin, cout, k, bn, relu, bn_training, shape, seed=10 + cin + cout)
    yr, xr, w, g, b, rm, rv = _ref_convbr(m, x)
    dy = torch.randn(yr.shape, generator=gen(7))
    yr.backward(dy)
    m = m.to(DEV)
    xg = gpu(x).requires_grad_(True)
    y = m(xg)
    y.backward(gpu(dy))
    close(y, yr, what="y")
    close(xg.grad, xr.grad, what="dx")
    close(m.conv.weight.grad, w.grad, what="dw")
    if bn:
        close(m.bn.weight.grad, g.grad, what="dgamma")
        close(m.bn.bias.grad, b.grad, what="dbeta")
        close(m.bn.running_mean, rm, 1e-5, "running_mean")
        close(m.bn.running_var, rv, 1e-5, "running_var")
        assert int(m.bn.num_batches_tracked) == (1 if bn_training else 0)


def test_convbr_frozen_unit_passes_gradient_only(ra):
    """A reused unit (rag.py:159-200): eval BN, parameters frozen -> only dx flows, no parameter .grad appears."""
    m, x = _convbr_case(ra, 12, 12, 3, True, True, False, (1, 12, 4, 8, 16), seed=3)
    for p in m.parameters():
        p.requires_grad = False
    yr, xr, *_ = _ref_convbr(m, x)
    dy = torch.randn(yr.shape, generator=gen(8))
    yr.backward(dy)
    m = m.to(DEV)
    xg = gpu(x).requires_grad_(True)
    m(xg).backward(gpu(dy))
    close(xg.grad, xr.grad, what="dx")
    assert all(p.grad is None for p in m.parameters())


def test_train_mode_forward_under_no_grad_uses_batch_statistics(ra):
    m, x = _convbr_case(ra, 12, 12, 3, True, True, True, (2, 12, 4, 6, 10), seed=5)
    yr, *_ = _ref_convbr(m, x)
    with torch.no_grad():
        y = m.to(DEV)(gpu(x))
    close(y, yr)


@pytest.mark.parametrize("cin,cout,k,stride,shape", [(3, 6, 3, 1, (2, 3, 12, 20)), (6, 12, 3, 3, (2, 6, 36, 48)), (12, 12, 3, 1, (2, 12, 9, 13)),
                                                     (12, 12, 1, 1, (2, 12, 8, 8)), (6, 12, 3, 3, (1, 6, 17, 23))])
def test_convbr2d_fwd_bwd_vs_torch(ra, cin, cout, k, stride, shape):
    torch.manual_seed(21)
    m = ra.ConvBR_2d(cin, cout, k, stride, (k - 1) // 2)
    with torch.no_grad():
        m.bn.weight.copy_(torch.rand(cout) + 0.5)
        m.bn.bias.copy_(torch.randn(cout) * 0.1)
    m.train()
    x = torch.randn(shape, generator=gen(22))
    yr, xr, w, g, b, rm, rv = _ref_convbr(m, x, ndim=2)
    dy = torch.randn(yr.shape, generator=gen(23))
    yr.backward(dy)
    m = m.to(DEV)
    xg = gpu(x).requires_grad_(True)
    y = m(xg)
    y.backward(gpu(dy))
    close(y, yr, what="y")
    close(xg.grad, xr.grad, what="dx")
    close(m.conv.weight.grad, w.grad, what="dw")
    close(m.bn.weight.grad, g.grad, what="dgamma")
    close(m.bn.bias.grad, b.grad, what="dbeta")
    close(m.bn.running_var, rv, 1e-5, "running_var")


@pytest.mark.parametrize("align", [True, False])
@pytest.mark.parametrize("shape,size", [((2, 3, 8, 12, 20), (4, 6, 10)), ((1, 2, 7, 9, 13), (4, 5, 7)), ((1, 4, 4, 5, 7), (8, 10, 14)),
                                        ((2, 2, 3, 4, 5), (5, 7, 9)), ((1, 1, 8, 4, 6), (24, 12, 18))])
def test_trilinear_backward_vs_aten(ra, align, shape, size):
    x = torch.randn(shape, generator=gen(31), requires_grad=True)
    yr = F.interpolate(x, size, mode="trilinear", align_corners=align)
    dy = torch.randn(yr.shape, generator=gen(32))
    yr.backward(dy)
    xg = gpu(x.detach()).requires_grad_(True)
    y = ra.autograd.TrilinearFn.apply(xg, size, align)
    y.backward(gpu(dy))
    close(y, yr, 1e-5)
    close(xg.grad, x.grad, 1e-5)


@pytest.mark.parametrize("B,C,h,w,maxdisp", [(2, 12, 6, 20, 24), (1, 3, 5, 7, 30), (1, 12, 4, 9, 48)])
def test_costvol_backward_vs_oracle(ra, B, C, h, w, maxdisp):
    L = torch.randn((B, C, h, w), generator=gen(41), requires_grad=True)
    R = torch.randn((B, C, h, w), generator=gen(42), requires_grad=True)
    cr = O.cost_volume(L, R, maxdisp)
    dc = torch.randn(cr.shape, generator=gen(43))
    cr.backward(dc)
    Lg, Rg = gpu(L.detach()).requires_grad_(True), gpu(R.detach()).requires_grad_(True)
    c = ra.autograd.CostVolFn.apply(Lg, Rg, maxdisp)
    c.backward(gpu(dc))
    assert torch.equal(c.cpu(), cr.detach())
    close(Lg.grad, L.grad, 1e-5)
    close(Rg.grad, R.grad, 1e-5)


@pytest.mark.parametrize("B,d,h,w,maxdisp,scale", [(2, 8, 4, 8, 24, 1.0), (1, 16, 6, 10, 48, 3.0), (1, 7, 5, 3, 21, 0.3), (1, 64, 4, 6, 192, 1.0)])
def test_disp_backward_vs_oracle(ra, B, d, h, w, maxdisp, scale):
    x = (torch.randn((B, 1, d, h, w), generator=gen(51)) * scale).requires_grad_(True)
    outr = O.disp_head(x, maxdisp)
    do = torch.randn(outr.shape, generator=gen(52))
    outr.backward(do)
    xg = gpu(x.detach()).requires_grad_(True)
    out = ra.Disp(maxdisp)(xg)
    out.backward(gpu(do))
    close(out, outr, 2e-4)
    close(xg.grad, x.grad, 2e-4)
    p = torch.rand((B, maxdisp, 3, 5), generator=gen(53)).requires_grad_(True)
    rr = O.disparity_regression(p, maxdisp)
    rr.backward(torch.ones_like(rr))
    pg = gpu(p.detach()).requires_grad_(True)
    ra.DisparityRegression(maxdisp)(pg).sum().backward()
    close(pg.grad, p.grad, 1e-6)


# --------------------------------------------------------------------------- cells
@pytest.mark.parametrize("name", ["same_conv", "same_unsorted", "same_deep", "down_odd", "up", "skip"])
def test_cell3d_train_step_vs_oracle(ra, name):
    """Cell_3d in train mode (batch statistics): concat and all gradients vs the oracle under PyTorch-CPU autograd."""
    g = load_golden("g4_cell3d")
    pp, p, fm, du = [int(v) for v in g[f"{name}::cfg"]]
    rows = g[f"{name}::rows"]
    sd = split_sd(g, f"{name}::sd::")
    params = {k: v.clone().requires_grad_(True) for k, v in sd.items() if v.is_floating_point() and "running_" not in k}
    sd2 = dict(sd)
    sd2.update(params)
    s0 = torch.from_numpy(g[f"{name}::s0"]).requires_grad_(True)
    s1 = torch.from_numpy(g[f"{name}::s1"]).requires_grad_(True)
    _prev, catr = O.cell_3d(s0, s1, sd2, "", rows, fm, du, training=True)
    dy = torch.randn(catr.shape, generator=gen(61))
    catr.backward(dy)

    cell = ra.Cell_3d(3, 3, pp, p, ra.Genotype(rows, None, rows, None), fm, du)
    cell.load_state_dict(sd)
    cell = cell.to(DEV).train()
    s0g, s1g = gpu(s0.detach()).requires_grad_(True), gpu(s1.detach()).requires_grad_(True)
    prev, cat = cell(s0g, s1g)
    assert prev is s1g
    cat.backward(gpu(dy))
    close(cat, catr, what="concat")
    close(s0g.grad, s0.grad, what="ds0")
    close(s1g.grad, s1.grad, what="ds1")
    n = 0
    for k, pr in params.items():
        if pr.grad is None:
            continue
        close(dict(cell.named_parameters())[k].grad, pr.grad, what=k)
        n += 1
    assert n >= 3


# --------------------------------------------------------------------------- the whole step vs the reference fixture
def _smooth_l1_step(disp, gt, maxdisp):
    mask = (gt < maxdisp) & (gt > 0)
    return F.smooth_l1_loss(disp[mask], gt[mask], reduction="mean")


def _check_g6_grads(net, g, tol, skip_2d):
    named = dict(net.named_parameters())
    n = 0
    for k, ref in g.items():
        if not k.startswith("grad::") or k.endswith("_fea") or (skip_2d and "_2d" in k):
            continue
        close(named[k[6:]].grad, torch.from_numpy(ref), tol, k)
        n += 1
    return n


def test_matchingnet_train_step_golden(ra):
    """Features -> disp -> smooth-L1 -> backward, against the reference's own training step (g6)."""
    g = load_golden("g6_train_step")
    maxdisp = int(g["maxdisp"])
    rows = g["rows"]
    net = ra.MatchingNet(ra.Genotype(rows, None, rows, None), maxdisp=maxdisp)
    sd = {k: v for k, v in split_sd(g).items()
          if k.split(".")[0] in ("stem3d0", "stem3d1", "cells_3d", "last_3_3d", "last_6_3d", "last_12_3d")}
    net.load_state_dict(sd, strict=True)
    net = net.to(DEV).train()
    net.stem3d0[0].eval()
    lf, rf = gpu(g["left_fea"]).requires_grad_(True), gpu(g["right_fea"]).requires_grad_(True)
    disp = net(lf, rf)
    loss = _smooth_l1_step(disp, gpu(g["gt"]), maxdisp)
    loss.backward()
    close(disp, torch.from_numpy(g["disp"]), 2e-4, "disp")
    assert abs(loss.item() - float(g["loss"])) < 1e-4
    assert _check_g6_grads(net, g, 5e-4, skip_2d=True) > 40
    close(lf.grad, torch.from_numpy(g["grad::left_fea"]), 5e-4, "dleft_fea")
    close(rf.grad, torch.from_numpy(g["grad::right_fea"]), 5e-4, "dright_fea")
    # running statistics after the step (momentum update of train-mode units; the reused stem is untouched)
    state = net.state_dict()
    n = 0
    for k, ref in g.items():
        if k.startswith("after::") and k[7:] in state:
            close(state[k[7:]].float(), torch.from_numpy(np.asarray(ref)).float(), 1e-4, k)
            n += 1
    assert n > 100
    assert torch.equal(state["stem3d0.0.bn.running_mean"].cpu(), split_sd(g)["stem3d0.0.bn.running_mean"])


def test_network_train_step_from_images_golden(ra):
    """The reference's whole training step from images (Feature Net in train mode too) against g6."""
    g = load_golden("g6_train_step")
    maxdisp = int(g["maxdisp"])
    rows = g["rows"]
    net = ra.Network(ra.Genotype(rows, None, rows, None), DEV, maxdisp=maxdisp)
    net.load_state_dict(split_sd(g), strict=True)
    net = net.to(DEV).train()
    net.stem3d0[0].eval()
    disp = net(gpu(g["left"]), gpu(g["right"]), 0, net.arch_init)
    loss = _smooth_l1_step(disp, gpu(g["gt"]), maxdisp)
    loss.backward()
    close(disp, torch.from_numpy(g["disp"]), 5e-4, "disp")
    assert abs(loss.item() - float(g["loss"])) < 2e-4
    assert _check_g6_grads(net, g, 1e-3, skip_2d=False) > 40


def test_sgd_step_changes_only_trained_units(ra):
    """rag.py:69,101-102,213-216: SGD over requires_grad parameters after freezing reused units; clip_grad_norm_."""
    g = load_golden("g6_train_step")
    maxdisp = int(g["maxdisp"])
    rows = g["rows"]
    net = ra.Network(ra.Genotype(rows, None, rows, None), DEV, maxdisp=maxdisp)
    net.load_state_dict(split_sd(g), strict=True)
    net = net.to(DEV).train()
    frozen = {"stem_3d0": [0], "cell_3d1": [0]}
    net.modify_param(frozen, requires_grad=False)
    net.stem3d0[0].eval()
    net.cells_3d[1][0].eval()
    before = {k: v.clone() for k, v in net.state_dict().items()}
    opt = torch.optim.SGD(filter(lambda p: p.requires_grad, net.parameters()), lr=1e-3, momentum=0.9, weight_decay=3e-3)
    disp = net(gpu(g["left"]), gpu(g["right"]), 0, net.arch_init)
    loss = _smooth_l1_step(disp, gpu(g["gt"]), maxdisp)
    opt.zero_grad()
    loss.backward()
    torch.nn.utils.clip_grad_norm_(net.parameters(), 5)
    opt.step()
    after = net.state_dict()
    for k in before:
        same = torch.equal(before[k], after[k])
        if k.startswith("stem3d0.0.") or k.startswith("cells_3d.1.0."):
            assert same, k
        elif k.startswith("last_3_"):
            assert same == (not k.endswith("conv.weight")), k      # bn=False heads: the unused BatchNorm never moves
        elif k.endswith("conv.weight") or k.endswith("running_mean"):
            assert not same, k


@pytest.mark.parametrize("flat", [False, True])
def test_graphed_train_step_matches_eager(ra, flat):
    """forward+backward replayed as one hipGraph (rag_amd.train.GraphedTrainStep) == the eager step, three steps in a row, with
    torch.optim.SGD and with the fused FlatSGD.  The graphed loop is driven from the default stream with a device sync before
    every step and a 4-byte `.clone()` of the loss (a memcpy on the null stream) between replays — the pattern under which a
    captured step that held a memset / memcpy NODE computed garbage in 6-7 runs of 8.  The step now consists of
    kernel nodes only, and GraphedTrainStep verifies that at capture time (`node_census`); it replays on the caller's stream."""
    from rag_amd.train import GradBucket, GraphedTrainStep, make_optimizer, train_step
    g = load_golden("g6_train_step")
    maxdisp = int(g["maxdisp"])
    rows = g["rows"]
    left, right, gt = gpu(g["left"]), gpu(g["right"]), gpu(g["gt"])
    finals = []
    for graphed in (False, True):
        net = ra.Network(ra.Genotype(rows, None, rows, None), DEV, maxdisp=maxdisp)
        net.load_state_dict(split_sd(g), strict=True)
        net = net.to(DEV).train()
        net.stem3d0[0].eval()
        net.modify_param({"stem_3d0": [0]}, requires_grad=False)
        bucket = GradBucket(net.parameters())
        opt = make_optimizer(net.parameters(), lr=1e-3, bucket=bucket if flat else None)
        losses = []
        if graphed:
            # the capture warm-up runs optimisation steps too: give the eager run the same number of steps
            step = GraphedTrainStep(net, opt, bucket, left, right, gt, warmup=2)
            assert step.node_census["memcpy"] == 0 and step.node_census["memset"] == 0 and step.node_census["kernel"] > 100, step.node_census
            held = []
            for _ in range(3):
                torch.cuda.synchronize()
                held.append(step().clone())
            torch.cuda.synchronize()
            losses = [None, None] + [float(x) for x in held]
        else:
            losses = [float(train_step(net, opt, bucket, left, right, gt)) for _ in range(5)]
        finals.append((losses, {k: v.detach().clone() for k, v in net.state_dict().items()}))
    (l0, s0), (l1, s1) = finals
    # float atomics (1x1x1 weight gradient, soft-argmin adjoint) make two runs differ in the last bits; five SGD steps on
    # randomly initialised weights amplify that to ~1e-3 relative
    for a, b in zip(l0[2:], l1[2:]):
        assert abs(a - b) < 2e-3 * max(1.0, abs(a)), (l0, l1)
    for k in s0:
        close(s1[k].float(), s0[k].float(), 5e-3, k)


@pytest.mark.parametrize("n,clip,wd", [(1000, 5.0, 3e-3), (735024 // 4, 0.05, 3e-3), (77, 0.0, 0.0)])
def test_sgd_clip_step_vs_torch(ra, n, clip, wd):
    """ragmi_sgd_clip_step == clip_grad_norm_ + torch.optim.SGD(momentum 0.9, weight decay) for three steps (rag.py:64-70, 215-216)."""
    p_ref = torch.nn.Parameter(torch.randn(n, generator=gen(81)))
    opt = torch.optim.SGD([p_ref], lr=1e-2, momentum=0.9, weight_decay=wd)
    p, buf = gpu(p_ref.detach().clone()), torch.zeros(n, device=DEV)
    for step in range(3):
        g = torch.randn(n, generator=gen(82 + step)) * (0.3 if step != 1 else 1e-3)       # step 1: norm below the clip threshold
        p_ref.grad = g.clone()
        total_ref = torch.nn.utils.clip_grad_norm_([p_ref], clip) if clip > 0 else torch.linalg.vector_norm(g)
        opt.step()
        g_dev = gpu(g)
        total = ra.ops.sgd_clip_step(p, g_dev, buf, 1e-2, 0.9, wd, clip, step == 0)
        close(total, total_ref.reshape(1), 3e-6, f"norm {step}")   # torch accumulates the norm in fp32
        close(g_dev, p_ref.grad, 1e-6, f"clipped grad {step}")
        close(p, p_ref.detach(), 1e-6, f"param {step}")
        close(buf, opt.state[p_ref]["momentum_buffer"], 1e-6, f"momentum {step}")


def test_flat_sgd_step_matches_torch_sgd(ra):
    """The training step with rag_amd.train.FlatSGD (parameters and gradients in flat buffers, one fused clip + update) ends in
    the same state as with clip_grad_norm_ + torch.optim.SGD; weight caches see the in-place update (loss keeps moving)."""
    from rag_amd.train import FlatSGD, GradBucket, make_optimizer, train_step
    g = load_golden("g6_train_step")
    maxdisp = int(g["maxdisp"])
    rows = g["rows"]
    left, right, gt = gpu(g["left"]), gpu(g["right"]), gpu(g["gt"])
    finals = []
    for flat in (False, True):
        net = ra.Network(ra.Genotype(rows, None, rows, None), DEV, maxdisp=maxdisp)
        net.load_state_dict(split_sd(g), strict=True)
        net = net.to(DEV).train()
        bucket = GradBucket(net.parameters())
        opt = make_optimizer(net.parameters(), lr=1e-3, bucket=bucket if flat else None)
        assert isinstance(opt, FlatSGD) == flat
        losses = [float(train_step(net, opt, bucket, left, right, gt)) for _ in range(3)]
        finals.append((losses, {k: v.detach().clone() for k, v in net.state_dict().items()}))
    (l0, s0), (l1, s1) = finals
    assert l0[0] != l0[1] and l1[0] != l1[1]
    for a, b in zip(l0, l1):
        assert abs(a - b) < 1e-3 * max(1.0, abs(a)), (l0, l1)
    for k in s0:
        close(s1[k].float(), s0[k].float(), 2e-3, k)


def test_flat_sgd_resume_from_checkpoint(ra, tmp_path):
    """Parameters that are views of FlatSGD's flat buffer survive save_checkpoint / load_checkpoint (run.py:194-196 file layout), and
    an optimizer rebuilt from its state_dict continues exactly where the original would have."""
    from rag_amd import checkpoint as ck
    from rag_amd.train import FlatSGD, GradBucket, make_optimizer, train_step
    g = load_golden("g6_train_step")
    rows = g["rows"]
    left, right, gt = gpu(g["left"]), gpu(g["right"]), gpu(g["gt"])
    net = ra.Network(ra.Genotype(rows, None, rows, None), DEV, maxdisp=int(g["maxdisp"]))
    net.load_state_dict(split_sd(g), strict=True)
    net = net.to(DEV).train()
    bucket = GradBucket(net.parameters())
    opt = make_optimizer(net.parameters(), lr=1e-3, bucket=bucket)
    for _ in range(2):
        train_step(net, opt, bucket, left, right, gt)
    path = tmp_path / "resume.ckpt"
    ck.save_checkpoint(path, net, [net.arch_init], task=0, optimizer=opt)
    data = torch.load(path, map_location="cpu", weights_only=False)
    net2, _ = ck.load_checkpoint(data, device=DEV)
    net2 = net2.train()
    for k, v in net.state_dict().items():
        close(net2.state_dict()[k].float(), v.float(), 0.0, k)
    bucket2 = GradBucket(net2.parameters())
    opt2 = make_optimizer(net2.parameters(), lr=1e-3, bucket=bucket2)
    assert isinstance(opt2, FlatSGD)
    opt2.load_state_dict(data["optimizer"])
    l1 = float(train_step(net, opt, bucket, left, right, gt))
    l2 = float(train_step(net2, opt2, bucket2, left, right, gt))
    assert abs(l1 - l2) <= 2e-4 * max(1.0, abs(l1)), (l1, l2)
    for k, v in net.state_dict().items():
        close(net2.state_dict()[k].float(), v.float(), 2e-4, k)


def test_training_step_issues_no_memcpy_or_memset(ra):
    """Nothing in forward + loss + backward may be a device memcpy / memset: captured into a hipGraph those become memcpy /
    memset NODES, which this runtime does not replay safely next to null-stream copies (rag_amd.train.GraphedTrainStep).
    Typical offenders: a contiguous same-dtype copy_/clone, torch.cat of 5-D tensors, the backward of x[:, :, 0]."""
    from rag_amd.train import GradBucket, forward_backward
    try:
        from torch.profiler import ProfilerActivity, profile
    except Exception as exc:  # noqa: BLE001
        pytest.skip(f"torch.profiler unavailable: {exc}")
    g = load_golden("g6_train_step")
    rows = g["rows"]
    net = ra.Network(ra.Genotype(rows, None, rows, None), DEV, maxdisp=int(g["maxdisp"]))
    net.load_state_dict(split_sd(g), strict=True)
    net = net.to(DEV).train()
    bucket = GradBucket(net.parameters())
    left, right, gt = gpu(g["left"]), gpu(g["right"]), gpu(g["gt"])
    forward_backward(net, bucket, left, right, gt)
    torch.cuda.synchronize()
    try:
        with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
            forward_backward(net, bucket, left, right, gt)
            torch.cuda.synchronize()
        names = [e.name for e in prof.events()]
    except RuntimeError as exc:
        pytest.skip(f"profiler could not trace the device: {exc}")
    assert any("ragmi" in n for n in names), "the profile saw no HIP kernels of this library"
    bad = sorted({n for n in names if "memcpy" in n.lower() or "memset" in n.lower() or "copyBuffer" in n})
    assert not bad, bad


# --------------------------------------------------------------------------- fused loss + metrics
@pytest.mark.parametrize("B,H,W,maxdisp,case", [(2, 36, 48, 24, "plain"), (4, 192, 384, 192, "plain"), (3, 33, 47, 48, "skip"),
                                                (1, 24, 24, 192, "sparse")])
def test_stereo_metrics_vs_oracle(ra, B, H, W, maxdisp, case):
    est = torch.rand((B, H, W), generator=gen(71)) * maxdisp
    gt = torch.rand((B, H, W), generator=gen(72)) * maxdisp * 1.1 - 2.0          # some <= 0, some >= maxdisp
    if case == "skip":                       # image 1: almost every positive gt is >= maxdisp -> dropped by the 10 % rule
        gt[1] = maxdisp + 1.0
        gt[1, :2, :5] = 3.0
    if case == "sparse":                     # KITTI-like sparse ground truth
        gt = torch.where(torch.rand((B, H, W), generator=gen(73)) < 0.7, torch.zeros(()), gt)
    est = est + (gt - est) * 0.97            # errors around the 1-3 px thresholds
    ref = O.stereo_metrics(est, gt, maxdisp)
    got = ra.metrics.stereo_metrics(gpu(est), gpu(gt), maxdisp).floats()
    for k, v in ref.items():
        assert abs(got[k] - v) <= 2e-5 * max(1.0, abs(v)), (k, got[k], v)


@pytest.mark.parametrize("case", ["a", "b", "c", "d"])
def test_stereo_metrics_vs_reference_fixture(ra, case):
    """The fused loss + metrics kernel against the REFERENCE's own numbers (g11: utilstool/metrics.py:21-65 + rag.py:418-430)."""
    g = load_golden("g11_metrics")
    got = ra.metrics.stereo_metrics(gpu(g[f"{case}::est"]), gpu(g[f"{case}::gt"]), 192).floats()
    for k, v in zip(("loss", "EPE", "D1", "Thres1", "Thres2", "Thres3"), g[f"{case}::scalars"]):
        assert abs(got[k] - v) <= 2e-5 * max(1.0, abs(v)), (case, k, got[k], v)


def test_masked_smooth_l1_forward_backward(ra):
    est = (torch.rand((2, 36, 48), generator=gen(74)) * 30).requires_grad_(True)
    gt = torch.rand((2, 36, 48), generator=gen(75)) * 30
    mask = (gt < 24) & (gt > 0)
    ref = F.smooth_l1_loss(est[mask], gt[mask], reduction="mean")
    (ref * 1.7).backward()
    eg = gpu(est.detach()).requires_grad_(True)
    loss = ra.metrics.masked_smooth_l1(eg, gpu(gt), 24)
    (loss * 1.7).backward()
    close(loss, ref, 1e-6)
    close(eg.grad, est.grad, 1e-6)


# --------------------------------------------------------------------------- MdeNAS supernet
def _supernet_from_golden(ra, g):
    net = ra.BasicNetwork(device=DEV, maxdisp=int(g["maxdisp"]))
    net.load_state_dict(split_sd(g), strict=True)
    return net.to(DEV)


def test_supernet_eval_forward_golden(ra):
    """BasicNetwork.forward(left, right, fea_ops, mat_ops) in eval mode vs the reference's own supernet (g9)."""
    g = load_golden("g9_supernet")
    net = _supernet_from_golden(ra, g).eval()
    left, right = gpu(g["left"]), gpu(g["right"])
    with torch.no_grad():
        disp = net(left, right, g["fea_ops"], g["mat_ops"])
        disp_conv = net(left, right, np.ones(9, dtype=np.int64), np.ones(9, dtype=np.int64))
    assert O.epe(disp.cpu(), torch.from_numpy(g["disp_eval"])) <= 1e-3
    assert O.epe(disp_conv.cpu(), torch.from_numpy(g["disp_eval_all_conv"])) <= 1e-3


def test_supernet_train_step_golden(ra):
    """One search training step (mdenas_search.py:164-173): disp, loss and gradients vs the reference."""
    g = load_golden("g9_supernet")
    net = _supernet_from_golden(ra, g).train()
    disp = net(gpu(g["left"]), gpu(g["right"]), g["fea_ops"], g["mat_ops"])
    loss = _smooth_l1_step(disp, gpu(g["gt"]), int(g["maxdisp"]))
    loss.backward()
    close(disp, torch.from_numpy(g["disp_train"]), 5e-4, "disp")
    assert abs(loss.item() - float(g["loss"])) < 2e-4 * max(1.0, float(g["loss"]))
    named = dict(net.named_parameters())
    n = 0
    for k, ref in g.items():
        if k.startswith("grad::"):
            close(named[k[6:]].grad, torch.from_numpy(ref), 1e-3, k)
            n += 1
    assert n > 80
    # ops that were not sampled get no gradient, like in the reference
    assert sum(1 for p in net.parameters() if p.grad is not None) == int(g["n_params_with_grad"])


def test_bucket_direct_gradients_match_autograd_accumulation(ra):
    """With a GradBucket the backward kernels accumulate straight into the flat buffer (no AccumulateGrad adds); the
    gradients must equal the ones plain autograd produces, and a second backward must ADD to them."""
    from rag_amd.train import GradBucket
    g = load_golden("g6_train_step")
    maxdisp = int(g["maxdisp"])
    rows = g["rows"]
    left, right, gt = gpu(g["left"]), gpu(g["right"]), gpu(g["gt"])

    def build():
        net = ra.Network(ra.Genotype(rows, None, rows, None), DEV, maxdisp=maxdisp)
        net.load_state_dict(split_sd(g), strict=True)
        net = net.to(DEV).train()
        net.stem3d0[0].eval()
        return net

    plain = build()
    _smooth_l1_step(plain(left, right, 0, plain.arch_init), gt, maxdisp).backward()
    ref = {k: p.grad.clone() for k, p in plain.named_parameters() if p.grad is not None}

    net = build()
    bucket = GradBucket(net.parameters())
    bucket.zero()
    _smooth_l1_step(net(left, right, 0, net.arch_init), gt, maxdisp).backward()
    got = {k: p.grad for k, p in net.named_parameters()}
    for k, r in ref.items():
        close(got[k], r, 1e-4, k)
        assert got[k].data_ptr() >= bucket.flat.data_ptr() and got[k].data_ptr() < bucket.flat.data_ptr() + bucket.flat.numel() * 4
    once = bucket.flat.clone()
    # second backward without zeroing: accumulation (running statistics moved, so compare against a fresh plain run of the same state)
    net2 = build()
    net2.load_state_dict(net.state_dict())
    net2.stem3d0[0].eval()
    _smooth_l1_step(net2(left, right, 0, net2.arch_init), gt, maxdisp).backward()
    second = {k: p.grad.clone() for k, p in net2.named_parameters() if p.grad is not None}
    _smooth_l1_step(net(left, right, 0, net.arch_init), gt, maxdisp).backward()
    views = dict(zip([k for k, p in net.named_parameters() if p.requires_grad], bucket._views()))
    off = 0
    for k, p in net.named_parameters():
        if k in second:
            expect = once[off:off + p.numel()].view_as(p) + second[k]
            close(p.grad, expect, 2e-4, "accumulated " + k)
        off += p.numel()
    assert views


# --------------------------------------------------------------------------- the kernels the TRAINING BENCH runs (configs[4] sizes)
# bench.py --train runs B=4 at 192x384, D=192: the level-3 volumes are 4 x 64 x 64 x 128 = 2^21 voxels, where the data gradient
# (the forward kernel on the weight packed transposed / flipped) dispatches to the f16x3 kernel and the weight gradient to the
# persistent-workgroup kernel with its partial-sum workspace.  The small cases above never reach those paths.
L3_TRAIN = (4, 64, 64, 128)


@pytest.mark.parametrize("cin,cout", [(12, 12), (24, 12), (4, 12)])
def test_dgrad_transposed_pack_on_x3_at_training_size(ra, cin, cout):
    """dL/dx of a 3x3x3 conv with forward weight [cout, cin] = conv3d_k3(dy, pack(w, transpose=True)) at the level-3 training
    shape, against torch.autograd.grad of F.conv3d on the CPU.  (cin, cout): stem3d1 12 -> 12, stem3d0 24 -> 12 (its gradient conv
    maps 12 -> 24 channels, two 16-row output groups), three stacked sibling convs 4 -> 3 x 4 (ConvBRGroupFn)."""
    B, D, H, W = L3_TRAIN
    w = torch.randn((cout, cin, 3, 3, 3), generator=gen(201)) * (2.0 / (27 * cout)) ** 0.5
    dy = torch.randn((B, cout, D, H, W), generator=gen(202))
    x = torch.zeros((B, cin, D, H, W), requires_grad=True)
    torch.set_num_threads(16)
    (ref,) = torch.autograd.grad(F.conv3d(x, w, padding=1), x, dy)
    with ra.ops.conv_precision("f16x3"):
        assert ra.ops.conv3d_k3_uses_x3(cout, cin, B, D, H, W)          # the gradient conv reads cout channels, writes cin
        dx = ra.ops.conv3d_k3(gpu(dy), ra.ops.conv3d_k3_pack(gpu(w), transpose=True), cin, None, None, False,
                              torch.empty((B, cin, D, H, W), device=DEV))
    close(dx, ref, 2e-4, "dx (f16x3, transposed pack)")
    with ra.ops.conv_precision("fp32"):
        dx32 = ra.ops.conv3d_k3(gpu(dy), ra.ops.conv3d_k3_pack(gpu(w), transpose=True), cin, None, None, False,
                                torch.empty((B, cin, D, H, W), device=DEV))
    close(dx32, ref, 2e-5, "dx (fp32 MFMA, transposed pack)")


@pytest.mark.parametrize("cin,cout", [(12, 12), (24, 12), (4, 12)])
def test_wgrad_persistent_kernel_at_training_size(ra, cin, cout):
    """conv3d_k3_wgrad at the level-3 training shape (persistent workgroups + partial-sum workspace + reduce kernel), fresh and
    accumulating into three stacked destinations, against torch.autograd.grad on the CPU."""
    B, D, H, W = L3_TRAIN
    x = torch.randn((B, cin, D, H, W), generator=gen(211))
    dy = torch.randn((B, cout, D, H, W), generator=gen(212)) * 0.01
    w = torch.zeros((cout, cin, 3, 3, 3), requires_grad=True)
    torch.set_num_threads(16)
    (ref,) = torch.autograd.grad(F.conv3d(x, w, padding=1), w, dy)
    dw = ra.ops.conv3d_k3_wgrad(gpu(x), gpu(dy), cout)
    close(dw, ref, 2e-4, "dw")
    if cout % 3 == 0:
        base = [torch.randn((cout // 3, cin, 3, 3, 3), generator=gen(213 + i)).to(DEV) for i in range(3)]
        into = [b.clone() for b in base]
        ra.ops.conv3d_k3_wgrad(gpu(x), gpu(dy), cout, into=into)
        close(torch.cat(into) - torch.cat(base), ref, 2e-4, "dw accumulated into stacked destinations")


def _rel_err(got, ref64):
    got, ref64 = got.detach().cpu().double(), ref64.detach().cpu().double()
    assert got.shape == ref64.shape, (got.shape, ref64.shape)
    return float((got - ref64).abs().max()) / max(1.0, float(ref64.abs().max()))


# Sums over 2^21 voxels (and, end to end, a soft-argmin over costs of 1e3-1e4) are ill-conditioned in fp32: the CPU reference's OWN
# fp32 results differ from an fp64 evaluation by up to 5e-3 of a tensor's largest gradient.  These tests therefore measure every
# path against the fp64 evaluation and bound the GPU's error by a multiple of the error the reference's fp32 arithmetic makes on the
# same inputs (floor: the tolerance of the small-shape tests): strict fp32 (RAGMI_F32) must sit in the same noise class, f16x3
# (RAGMI_F32X3: ~30x the per-product rounding of fp32, include/rag_amd.h) may amplify it by the stated factor.
NOISE_FACTOR = {"fp32": 2.5, "f16x3": 12.0}
# Parameter gradients are sums over every voxel THROUGH the ReLU mask, and the mask is discontinuous: a pre-activation within the
# forward error of zero flips it, which moves the sum by a whole |dy|.  With N voxels per channel, a forward error of eps relative
# to the activation scale flips ~0.4 eps N of them (density of a unit normal at 0): at N = 2^21, fp32 (eps ~ 1e-7) flips < 1
# element, f16x3 (eps ~ 1e-5: 2^-16 per product, include/rag_amd.h) flips ~8, i.e. an absolute error of ~3 |dy| on sums of
# magnitude sqrt(N) ~ 1.4e3 — 2e-3 relative, independent of how well conditioned the sum is in fp32.  Measured 1.4e-3 (dw), 2.5e-3
# (dbeta); elementwise results (y, dx) keep the 2e-4 floor.
REDUCTION_FLOOR = {"fp32": 2e-4, "f16x3": 6e-3}


def test_convbr_group_fn_at_training_size(ra):
    """ConvBRGroupFn (three sibling 4 -> 4 ConvBRs of a level-3 cell state, train-mode BatchNorm, one residual per unit) forward
    and backward at the level-3 training shape: stacked forward conv, per-unit BN + ReLU (+ residual), stacked data gradient
    (transposed pack) and stacked weight gradient (persistent kernel), under both arithmetic contracts, against fp64 autograd."""
    from rag_amd import autograd as ag
    B, D, H, W = L3_TRAIN
    C, n = 4, 3
    mods = []
    for i in range(n):
        m, _x = _convbr_case(ra, C, C, 3, True, True, True, (1, C, 2, 2, 2), seed=300 + i)
        mods.append(m)
    x = torch.randn((B, C, D, H, W), generator=gen(221))
    res = [torch.randn((B, C, D, H, W), generator=gen(222 + i)) for i in range(n)]
    dys = [torch.randn((B, C, D, H, W), generator=gen(230 + i)) for i in range(n)]
    torch.set_num_threads(16)

    def reference(dt):
        xr = x.detach().to(dt).clone().requires_grad_(True)     # a private leaf: x.to(float32) would be x itself
        outs, ps = [], []
        for i, m in enumerate(mods):
            w = m.conv.weight.detach().cpu().to(dt).requires_grad_(True)
            g = m.bn.weight.detach().cpu().to(dt).requires_grad_(True)
            b = m.bn.bias.detach().cpu().to(dt).requires_grad_(True)
            outs.append(F.relu(F.batch_norm(F.conv3d(xr, w, padding=1), None, None, g, b, training=True, eps=m.bn.eps)) + res[i].to(dt))
            ps += [w, g, b]
        torch.autograd.backward(outs, [d.to(dt) for d in dys])
        return [o.detach() for o in outs] + [xr.grad] + [p.grad for p in ps]

    names = [f"y{i}" for i in range(n)] + ["dx"] + [f"{k}{i}" for i in range(n) for k in ("dw", "dgamma", "dbeta")]
    ref64, ref32 = reference(torch.float64), reference(torch.float32)
    noise = {k: _rel_err(r32, r64) for k, r32, r64 in zip(names, ref32, ref64)}
    for prec in ("fp32", "f16x3"):
        gm = [type(m)(C, C, 3, 1, 1) for m in mods]
        for a_, b_ in zip(gm, mods):
            a_.load_state_dict(b_.state_dict())
            a_.train()
        gm = [m.to(DEV) for m in gm]
        xg = gpu(x).requires_grad_(True)
        params = [p for m in gm for p in (m.conv.weight, m.bn.weight, m.bn.bias)]
        with ra.ops.conv_precision(prec):
            assert ra.ops.conv3d_k3_uses_x3(C, n * C, B, D, H, W) == (prec == "f16x3")
            outs = ag.ConvBRGroupFn.apply(xg, tuple(gm), *params, *[gpu(r) for r in res])
            torch.autograd.backward(outs, [gpu(d) for d in dys])
        got = list(outs) + [xg.grad] + [p.grad for p in params]
        errs = {k: _rel_err(t, r64) for k, t, r64 in zip(names, got, ref64)}
        worst = max(errs, key=lambda k: errs[k] / max(noise[k], 2e-5))
        print(f"ConvBRGroupFn at {L3_TRAIN} [{prec}]: worst {worst}: err {errs[worst]:.2e} vs CPU-fp32 noise {noise[worst]:.2e}; "
              + "; ".join(f"{k} {errs[k]:.1e}/{noise[k]:.1e}" for k in names))
        for k in names:
            floor = REDUCTION_FLOOR[prec] if k[0] == "d" and k != "dx" else 2e-4
            assert errs[k] <= max(floor, NOISE_FACTOR[prec] * noise[k]), (prec, k, errs[k], noise[k])


def test_matchingnet_train_step_at_reference_crop(ra):
    """One training step of the Matching Net at the reference's own crop (192x384, stereo_dataset.py:59-62; D = 192; one pair of
    run_rag.sh:17's batch of four): disparity, loss, feature gradients and every parameter gradient — at the sizes at which
    bench.py --train's kernels are selected (level-3 volumes of 2^19 voxels: f16x3 or fp32-MFMA forward and data-gradient
    convolutions, persistent weight-gradient kernel) — against the CPU oracle + PyTorch autograd evaluated in fp64, with the
    oracle's own fp32 evaluation as the noise yardstick.  approaches/rag.py:204-216."""
    from test_oracle_golden import oracle_train_step
    rows = O.ALL_CONV
    maxdisp = 192
    sd = O.random_matching_state_dict(rows, seed=21)
    gq = gen(241)
    g = {"left_fea": torch.randn((1, 12, 64, 128), generator=gq).numpy(), "right_fea": torch.randn((1, 12, 64, 128), generator=gq).numpy(),
         "gt": (torch.rand((1, 192, 384), generator=gq) * 200.0).numpy(), "rows": rows, "maxdisp": maxdisp}
    torch.set_num_threads(16)
    d32, l32, g32 = oracle_train_step(g, sd)
    g64 = {k: (v.astype(np.float64) if isinstance(v, np.ndarray) and v.dtype == np.float32 else v) for k, v in g.items()}
    d64, l64, gr64 = oracle_train_step(g64, {k: (v.double() if v.is_floating_point() else v) for k, v in sd.items()})
    noise16 = {k: _rel_err(g32[k], gr64[k]) for k in gr64}
    # A SECOND fp32 evaluation of the same step on the CPU in another summation order: the problem mirrored along H (features, ground
    # truth and every convolution's kh taps flipped — the cost volume shifts along W only and both trilinear modes are symmetric, so
    # in exact arithmetic the result is the mirror image).  Each fp32 evaluation moves a different handful of near-tie pixels of the
    # soft-argmin and of near-zero pre-activations under the ReLU masks, so its gradient error against fp64 is one draw of a
    # heavy-tailed variable per tensor: two legitimate fp32 evaluations differ from each other by the factors printed below, which is
    # what the strict-fp32 GPU path shows against the unmirrored run (measured r02: GPU 1.05e-2 vs CPU 5.1e-3 on
    # cells_3d.0.0._ops.4.conv.weight).  The yardstick is the larger of the two CPU draws per tensor.
    gm = dict(g, left_fea=np.ascontiguousarray(g["left_fea"][:, :, ::-1]), right_fea=np.ascontiguousarray(g["right_fea"][:, :, ::-1]),
              gt=np.ascontiguousarray(g["gt"][:, ::-1]))
    sdm = {k: (v.flip(3).contiguous() if v.dim() == 5 else v) for k, v in sd.items()}
    _d1, _l1, g1 = oracle_train_step(gm, sdm)
    _d1 = _d1.flip(1)
    g1 = {k: (v.flip(3) if v.dim() == 5 else v.flip(2) if k in ("left_fea", "right_fea") else v) for k, v in g1.items()}
    noise1 = {k: _rel_err(g1[k], gr64[k]) for k in gr64}
    noise = {k: max(noise16[k], noise1[k]) for k in gr64}
    kw = max(noise16, key=lambda k: noise16[k])
    spread = sorted((max(noise16[k], 1e-9) / max(noise1[k], 1e-9) for k in gr64 if max(noise16[k], noise1[k]) > 1e-4))
    print(f"CPU fp32 vs fp64: worst {noise16[kw]:.2e} ({kw}); H-mirrored evaluation on the same tensor {noise1[kw]:.2e}, its worst "
          f"{max(noise1.values()):.2e}; per-tensor ratio of the two CPU evaluations' errors: min {spread[0]:.2f} median {spread[len(spread) // 2]:.2f} "
          f"max {spread[-1]:.2f} over {len(spread)} tensors; EPE between the two CPU fp32 evaluations {O.epe(_d1, d32):.3e} px")
    epe_noise = O.epe(d32, d64)
    for prec in ("fp32", "f16x3"):
        net = ra.MatchingNet(ra.Genotype(rows, None, rows, None), maxdisp=maxdisp)
        net.load_state_dict(sd, strict=True)
        net = net.to(DEV).train()
        net.stem3d0[0].eval()                                 # a reused unit, as in the oracle helper (rag.py:159-200)
        lf, rf = gpu(g["left_fea"]).requires_grad_(True), gpu(g["right_fea"]).requires_grad_(True)
        with ra.ops.conv_precision(prec):
            assert ra.ops.conv3d_k3_uses_x3(12, 12, 1, 64, 64, 128) == (prec == "f16x3")
            disp = net(lf, rf)
            loss = _smooth_l1_step(disp, gpu(g["gt"]), maxdisp)
            loss.backward()
        epe = O.epe(disp.detach().cpu(), d64)
        named = dict(net.named_parameters())
        errs = {}
        for k, ref in gr64.items():
            got = lf.grad if k == "left_fea" else rf.grad if k == "right_fea" else named[k].grad
            errs[k] = _rel_err(got, ref)
        top = sorted(errs.items(), key=lambda kv: -kv[1])[:3]
        print(f"train step at 192x384 [{prec}]: EPE vs fp64 {epe:.3e} px (CPU fp32: {epe_noise:.3e}); loss {loss.item():.6f} vs {l64:.6f}; "
              f"worst gradient errors {top}; CPU-fp32 worst {max(noise.values()):.2e}")
        f = NOISE_FACTOR[prec]
        ratios = {k: errs[k] / max(noise[k], 0.5 * max(noise.values()), 2e-4) for k in errs}
        kr = max(ratios, key=lambda k: ratios[k])
        print(f"  [{prec}] worst error / yardstick = {ratios[kr]:.2f} ({kr}: err {errs[kr]:.2e}, CPU fp32 {noise16[kr]:.2e}, mirrored {noise1[kr]:.2e}); "
              f"allowed {f}")
        assert epe <= max(1e-3, f * epe_noise), (prec, epe, epe_noise)
        assert abs(loss.item() - l64) <= 1e-5 * max(1.0, abs(l64))
        worst_noise = max(noise.values())
        for k, err in errs.items():
            # per tensor against its own noise, with the step's worst noise as the floor (tiny tensors have noisy noise estimates)
            assert err <= f * max(noise[k], 0.5 * worst_noise, 2e-4), (prec, k, err, noise[k])
        assert len(errs) > 200


def test_eval_after_train_forward_without_weight_update_sees_new_running_stats(ra):
    """The eval-mode caches (folded scale / shift, fused sibling weights) key on tensor versions; the train-mode BatchNorm kernels
    update running_mean / running_var through raw pointers.  eval forward -> train forward under no_grad (BN re-estimation: no
    parameter update follows) -> eval forward must use the NEW statistics, for a lone ConvBR and for a cell's fused siblings."""
    m, x = _convbr_case(ra, 12, 12, 3, True, True, False, (2, 12, 5, 9, 33), seed=41)
    ref = ra.ConvBR_3d(12, 12, 3, 1, 1)
    ref.load_state_dict(m.state_dict())
    torch_m = torch.nn.Sequential(ref.conv, ref.bn, torch.nn.ReLU())       # plain PyTorch modules on the CPU
    m = m.to(DEV)
    xg = gpu(x)
    with torch.no_grad():
        for mode in ("eval", "train", "eval"):
            m.train(mode == "train")
            torch_m.train(mode == "train")
            y, yr = m(xg), torch_m(x)
            close(y, yr, 2e-4, f"ConvBR {mode} forward")
    close(m.bn.running_mean, ref.bn.running_mean, 1e-5, "running_mean")
    # a whole cell: the dual launch's stacked scale / shift live in Cell._fused_cache
    g = load_golden("g4_cell3d")
    pp, p, fm, du = [int(v) for v in g["same_conv::cfg"]]
    rows = g["same_conv::rows"]
    cell = ra.Cell_3d(3, 3, pp, p, ra.Genotype(rows, None, rows, None), fm, du)
    cell.load_state_dict(split_sd(g, "same_conv::sd::"))
    cell = cell.to(DEV).eval()
    s0, s1 = gpu(g["same_conv::s0"]), gpu(g["same_conv::s1"])
    with torch.no_grad():
        first = cell(s0, s1)[1].clone()
        cell.train()
        cell(s0 * 2 + 1, s1 * 2 - 1)                       # moves every running statistic
        cell.eval()
        second = cell(s0, s1)[1]
        sd_now = {k: v.detach().cpu() for k, v in cell.state_dict().items()}
    want = O.cell_3d(torch.from_numpy(g["same_conv::s0"]), torch.from_numpy(g["same_conv::s1"]), sd_now, "", rows, fm, du)[1]
    assert not torch.allclose(first, second)
    close(second, want, 2e-4, "cell eval forward after a train-mode pass")
