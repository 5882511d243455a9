"""which in-launch exchanges time out in a training step (GPU box)"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import disyolo_amd  # noqa: F401
from disyolo_amd import lib as L
from disyolo_amd.net import YOLONet
from disyolo_amd.synth import synthetic_batch

B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
dev = torch.device("cuda:0")
net = YOLONet(training=True, device=dev, image_size=576, batch_size=B, stage=1, seed=0)
batch = synthetic_batch(B, 576, seed=1)
if B == 8:
    net.autotune(cache=os.path.join(ROOT, "profiles", "tune_train_B8_576_stage1.json"))
mode = sys.argv[2] if len(sys.argv) > 2 else "eager"
for step in range(2):
    t0 = time.time()
    if mode == "eager":
        net.set_batch(batch)
        net._forward_layers(True)
        torch.cuda.synchronize()
        print("forward %.3f s" % (time.time() - t0)); t0 = time.time()
        net.compute_losses(0.3)
        torch.cuda.synchronize()
        print("losses %.3f s" % (time.time() - t0)); t0 = time.time()
        net.backward()
        torch.cuda.synchronize()
        print("backward %.3f s" % (time.time() - t0))
    elif mode == "joined_nosync":
        if step == 0:
            net.set_batch(batch)
            net.build_program(overlap_tail=False)
        for _ in range(4):
            net.run_program()
        torch.cuda.synchronize()
        print("4 joined replays, no sync in between %.3f s" % (time.time() - t0))
    elif mode == "overlap":
        if step == 0:
            net.set_batch(batch)
            net.build_program(overlap_tail=True)
        for _ in range(4):
            net.run_program()
        torch.cuda.synchronize()
        print("4 overlapped replays %.3f s" % (time.time() - t0))
    else:
        net.train_step(batch)
        torch.cuda.synchronize()
        print("step %.3f s" % (time.time() - t0))
    for l in net.layers:
        for name, buf in (("fwd", l.csync), ("bwd", l.csync_bwd)):
            if buf is not None:
                e = L.cluster_sync_error(buf, l.cout)
                cnt = buf.view(-1, 32)[:, :2].cpu()
                nz = int((cnt != 0).sum())
                if e or nz:
                    print("  layer %d %s: error word %#x, %d non-zero counters %s" % (l.idx, name, e, nz, cnt[(cnt != 0).any(dim=1)][:4].tolist()))
    print("fused fwd:", [l.idx for l in net.layers if l.fused_fwd], "bwd:", [l.idx for l in net.layers if l.fused_bwd])
