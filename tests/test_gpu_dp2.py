"""Two data-parallel ranks through the REAL HIP train step.  One MI355X is all a test box has, and RCCL
refuses two ranks on one device, so the two processes share cuda:0 and exchange over gloo (which stages
device tensors through the host): everything but the transport is the production path -- the recorded
two-lane step cut at the bucket boundaries, begin_step / fire / finish, the 1/world scale inside Adam,
rank 0's broadcast.  Checks (stage 1 and stage 2, per-rank batches differ):
  * after the broadcast both ranks hold rank 0's variables although they were initialised differently;
  * step 1: the weights equal, bit for bit, what one process gets from the SUM of the two ranks' local
    gradients (each reproduced by a single-rank net on that rank's batch) pushed through the same Adam
    with grad_scale = 1/2 -- a two-term f32 sum does not depend on the order;
  * after 2 steps the ranks still hold identical weights and Adam moments (batch-norm moving statistics
    are local by design and differ: SURVEY.md 8e)."""
import os
import socket
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu

B, S = 2, 64


def _make_net(stage, seed, dev):
    from disyolo_amd.net import YOLONet
    import disyolo_oracle as O  # noqa: F401
    net = YOLONet(training=True, device=dev, image_size=S, batch_size=B, stage=stage, seed=seed)
    net.shuffle_seed = 5
    return net


def _seed_heads(net, seed):
    # detections must exist for the mask loss to have RoIs: larger head weights / biases (as test_gpu_configs does)
    g = torch.Generator().manual_seed(seed)
    for idx in (59, 67, 75):
        for leaf, std in (("weights", 0.3), ("biases", 0.5)):
            p = net.params["yolo/convolutional%d/%s" % (idx, leaf)]
            p.copy_((torch.randn(p.shape, generator=g) * std).to(p.device))
    net.refresh_weights()


def _worker(rank, world, port, out, stage):
    ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    for p in (ROOT, os.path.join(ROOT, "oracle")):
        if p not in sys.path:
            sys.path.insert(0, p)
    import disyolo_amd  # noqa: F401
    import disyolo_oracle as O
    from disyolo_amd.dp import enable_data_parallel
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    dev = torch.device("cuda:0")
    net = _make_net(stage, 10 + rank, dev)            # different initialisation per rank
    _seed_heads(net, 50 + rank)
    w_init = net.arena.clone()
    enable_data_parallel(net, bucket_mb=4.0)          # broadcasts rank 0's variables
    gathered = [torch.zeros_like(net.arena) for _ in range(world)]
    dist.all_gather(gathered, net.arena)
    flag = torch.tensor([float((w_init != net.arena).any())], device=dev)
    flags = [torch.zeros_like(flag) for _ in range(world)]
    dist.all_gather(flags, flag)                      # rank 1's variables were replaced, rank 0's were not
    res = {"bcast_equal": all(torch.equal(g, gathered[0]) for g in gathered),
           "differed": [bool(f.item()) for f in flags], "buckets": len(net.dp.buckets)}
    state0 = {k: v.clone() for k, v in net.state_dict().items()}
    batches = [O.synthetic_batch(B, S, seed=100 * (rank + 1) + t) for t in range(2)]
    net.build_program(det_thresh=0.1)
    losses = []
    for t in range(2):
        net.set_batch(batches[t])
        losses.append(float(net.train_step(None).cpu()))
        if t == 0:
            torch.cuda.synchronize()
            res["arena_step1"] = net.arena.cpu().clone()
    torch.cuda.synchronize()
    final = [torch.zeros_like(net.arena) for _ in range(world)]
    dist.all_gather(final, net.arena)
    fm = [torch.zeros_like(net.adam_m) for _ in range(world)]
    dist.all_gather(fm, net.adam_m)
    fv = [torch.zeros_like(net.adam_v) for _ in range(world)]
    dist.all_gather(fv, net.adam_v)
    res["ranks_agree"] = all(torch.equal(a, final[0]) for a in final) and all(torch.equal(a, fm[0]) for a in fm) \
        and all(torch.equal(a, fv[0]) for a in fv)
    res["losses"] = losses
    res["finite"] = bool(torch.isfinite(net.arena).all())
    if rank == 0:
        res["state0"] = {k: v.cpu() for k, v in state0.items()}
        torch.save(res, out)
    dist.barrier()
    dist.destroy_process_group()


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


@pytest.mark.parametrize("stage", [1, 2])
def test_two_ranks_on_one_gpu_over_gloo(dev, tmp_path, stage):
    import disyolo_oracle as O
    out = str(tmp_path / "dp2.pt")
    mp.spawn(_worker, args=(2, _free_port(), out, stage), nprocs=2, join=True)
    res = torch.load(out)
    assert res["bcast_equal"] and res["differed"] == [False, True] and res["finite"]
    assert res["buckets"] >= (3 if stage == 2 else 2)
    assert res["ranks_agree"]
    # single-process reproduction of step 1: local gradients of both ranks' batches from the broadcast state
    grads = []
    for rank in range(2):
        n = _make_net(stage, 0, dev)
        n.load_state_dict({k: v.to(dev) for k, v in res["state0"].items()})
        n.set_batch(O.synthetic_batch(B, S, seed=100 * (rank + 1)))
        n.compute_losses(0.1)
        n.backward()
        torch.cuda.synchronize()
        grads.append(n.grad_arena.clone())
    n.grad_arena.copy_(grads[0] + grads[1])
    n.optimizer_step(0.5)
    torch.cuda.synchronize()
    assert torch.equal(n.arena.cpu(), res["arena_step1"]), \
        "max |dw| %.3g" % float((n.arena.cpu() - res["arena_step1"]).abs().max())
