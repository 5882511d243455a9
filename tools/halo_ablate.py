"""Ablation timings of the 3x3 patch kernel on its two step-dominating shapes (GPU box; a -DDY_PROBE build:
tools/build_probe.sh).  usage: DISYOLO_LIB=dis-yolo_amd/libdisyolo_<name>.so python tools/halo_ablate.py [label]
flags: 0x10000 no MFMAs, 0x20000 no DMAs inside the loop, 0x80000 no epilogue."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import disyolo_amd
from disyolo_amd import lib as L

dev = torch.device("cuda:0")
bf = torch.bfloat16
label = sys.argv[1] if len(sys.argv) > 1 else os.environ.get("DISYOLO_LIB", "product")
CASES = [(8, 18, 512, 1024, 18), (8, 36, 256, 512, 16), (8, 72, 128, 256, 16), (8, 36, 512, 256, 18)]
if os.environ.get("ABL_CASES"):       # "B,H,Cin,Cout,tile;..."
    CASES = [tuple(int(v, 0) for v in c.split(",")) for c in os.environ["ABL_CASES"].split(";")]
FLAGS = tuple(int(f, 0) for f in os.environ.get("ABL_FLAGS", "0").split(","))
flush = torch.empty(600 << 20, dtype=torch.uint8, device=dev)
for (B, H, Cin, Cout, tile) in CASES:
    x = torch.randn(B, H, H, Cin, device=dev).to(bf)
    w = (torch.randn(Cout, 9 * Cin, device=dev) * 0.02).to(bf)
    y = torch.empty(B, H, H, Cout, dtype=bf, device=dev)
    sc, sh = torch.ones(Cout, device=dev), torch.zeros(Cout, device=dev)
    out = []
    for fl in FLAGS:
        d = L.make_conv_desc(x, w, y, 3, 1, scale=sc, shift=sh, leaky=True, tile=tile)
        d.flags |= fl
        for _ in range(3):
            L.conv2d_fwd(d)
        torch.cuda.synchronize()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(20):
            L.conv2d_fwd(d)
        e.record(); torch.cuda.synchronize()
        hot = s.elapsed_time(e) / 20 * 1e3
        cold = 0.0
        for _ in range(4):
            flush.fill_(1)
            s.record(); L.conv2d_fwd(d); e.record(); torch.cuda.synchronize()
            cold += s.elapsed_time(e) * 1e3 / 4
        out.append("%#x: %.1f/%.1f" % (fl, hot, cold))
    gf = 2.0 * B * H * H * Cout * Cin * 9 / 1e9
    print("[%s] B=%d %d^2 %d->%d tile %d (%.1f GFLOP) hot/cold us | " % (label, B, H, Cin, Cout, tile, gf) + "  ".join(out), flush=True)
