"""Child process of tests/test_rccl_one_rank.py: the product's only collective (the flat-bucket gradient all-reduce of
rag_amd.train, approaches/rag.py:204-216) on RCCL with the rank count one GPU box has — ONE rank.

Started by tests/conftest.py BEFORE the pytest process touches the GPU (a fresh process: library load, communicator setup, the
collective on the flat bucket, its ordering against FlatSGD on the launch stream, and hipGraph capture with RCCL's watchdog thread
alive).  Runs two eager train_steps and two GraphedTrainStep replays twice — with dist=torch.distributed (backend nccl = RCCL) and
with dist=None — from identical seeds and writes a JSON verdict.  An all-reduce over one rank is the identity, so the two must agree
to the step's own run-to-run noise: the soft-argmin adjoint accumulates tile contributions with float atomics (order not fixed),
which moves parameters by ~1e-8 between two identical runs; a third run (dist=None again) measures exactly that, and
dist-vs-no-dist must stay within 10x of it (floor 1e-6)."""
import json
import os
import socket
import sys
import traceback

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def run(use_dist, dist):
    import numpy as np
    import torch

    import rag_amd
    from rag_amd.train import GradBucket, GraphedTrainStep, make_optimizer, train_step
    dev = torch.device("cuda", 0)
    rows = np.array([[0, 1], [1, 0], [3, 0], [2, 1], [8, 1], [6, 0]])
    geno = rag_amd.Genotype(normal=rows, normal_concat=None, reduce=rows, reduce_concat=None)
    torch.manual_seed(0)
    net = rag_amd.MatchingNet(geno, maxdisp=24).to(dev).train()
    g = torch.Generator().manual_seed(5)
    lf = torch.randn((2, 12, 12, 20), generator=g).to(dev)        # the g6 fixture's size class: ~4 K cost-volume voxels
    rf = torch.randn((2, 12, 12, 20), generator=g).to(dev)
    gt = (torch.rand((2, 36, 60), generator=g) * 30).to(dev)
    bucket = GradBucket(net.parameters())
    opt = make_optimizer(net.parameters(), bucket=bucket)
    d = dist if use_dist else None
    losses = [float(train_step(net, opt, bucket, lf, rf, gt, clip=5.0, dist=d, features=True)) for _ in range(2)]
    graphed = GraphedTrainStep(net, opt, bucket, lf, rf, gt, clip=5.0, dist=d, features=True)     # 2 more (warm-up) steps inside
    for _ in range(2):
        losses.append(float(graphed()))
        float(bucket.flat.sum())            # a null-stream-free D2H between replays, like a training loop's logging
    torch.cuda.synchronize()
    return ({k: v.detach().cpu().clone() for k, v in net.state_dict().items()}, opt.momentum_buffer.cpu().clone(), losses,
            graphed.node_census)


def main(out_path):
    verdict = {"ok": False}
    try:
        import torch
        import torch.distributed as dist
        s = socket.socket()
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
        s.close()
        os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK="0", WORLD_SIZE="1", LOCAL_RANK="0")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        torch.cuda.set_device(0)
        torch.cuda.set_stream(torch.cuda.Stream(0))
        dist.init_process_group("nccl", device_id=torch.device("cuda", 0))
        verdict.update(ranks_seen=dist.get_world_size(), dist_backend=dist.get_backend())
        probe = torch.arange(8, device="cuda", dtype=torch.float32)
        dist.all_reduce(probe)                                   # the collective itself, once, checked
        verdict["probe_ok"] = bool(torch.equal(probe.cpu(), torch.arange(8, dtype=torch.float32)))
        sd_d, mom_d, loss_d, census = run(True, dist)
        sd_n, mom_n, loss_n, _ = run(False, dist)
        sd_n2, _m2, loss_n2, _ = run(False, dist)          # run-to-run determinism of the step itself, without the collective
        diff = [k for k in sd_d if not torch.equal(sd_d[k], sd_n[k])]
        diff_nn = {k: float((sd_n[k].double() - sd_n2[k].double()).abs().max()) for k in sd_n if not torch.equal(sd_n[k], sd_n2[k])}
        dd = {k: float((sd_d[k].double() - sd_n[k].double()).abs().max()) for k in diff}
        noise = max(diff_nn.values(), default=0.0)
        worst = max(dd.values(), default=0.0)
        mom_diff = float((mom_d.double() - mom_n.double()).abs().max())
        verdict.update(run_to_run_maxdiff_nodist=noise, run_to_run_tensors_differing=len(diff_nn), dist_vs_nodist_maxdiff=worst,
                       dist_vs_nodist_tensors_differing=len(dd), losses_nodist2=loss_n2, momentum_maxdiff=mom_diff)
        verdict.update(params_compared=len(sd_d),
                       losses_dist=loss_d, losses_nodist=loss_n, graph_nodes=census,
                       collective="dist.all_reduce on GradBucket.flat (one flat fp32 bucket), then FlatSGD on the same stream")
        tol = max(10.0 * noise, 1e-6)
        loss_close = all(abs(a - b) <= 1e-5 * max(1.0, abs(b)) for a, b in zip(loss_d, loss_n))
        verdict["tolerance"] = tol
        verdict["ok"] = bool(verdict["probe_ok"] and worst <= tol and mom_diff <= 1e-5 and loss_close and all(map(lambda v: v == v, loss_d)))
        dist.barrier()
        dist.destroy_process_group()
    except Exception:  # noqa: BLE001
        verdict["error"] = traceback.format_exc()
    with open(out_path, "w") as f:
        json.dump(verdict, f)
    return 0 if verdict["ok"] else 1


if __name__ == "__main__":
    sys.exit(main(sys.argv[1]))
