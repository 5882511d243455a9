"""End-to-end box / class / mask agreement of the HIP inference path with the f32 oracle (tests/e2e_parity.py) as a
report: python tools/e2e_parity_report.py [steps] [out.json]   (GPU box; ~1 min per configuration on 128 host cores)"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import torch  # noqa: E402
import disyolo_amd  # noqa: E402,F401
import e2e_parity as E  # noqa: E402

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 1200
out = sys.argv[2] if len(sys.argv) > 2 else None
dev = torch.device("cuda:0")
report = []
for S, B in ((576, 8), (832, 4)):
    batch = E.painted_batch(B, S, seed=5)
    t0 = time.time()
    sd, curve = E.train_overfit(dev, batch, B, S, steps)
    print("S=%d B=%d: %d steps in %.1f s, loss %s" % (S, B, steps, time.time() - t0, [round(c, 2) for c in curve]), flush=True)
    for thr in (0.25, 0.5):
        for Bi in ((B, 1) if thr == 0.25 else (B,)):
            t0 = time.time()
            r = E.compare(dev, sd, batch["images"][:Bi], batch["clip_window"][:Bi], S, thr, with_bf16_oracle=(thr == 0.25))
            r["train_steps"], r["loss_curve"], r["seconds"] = steps, [round(c, 2) for c in curve], round(time.time() - t0, 1)
            print(json.dumps(r), flush=True)
            report.append(r)
if out:
    json.dump(report, open(out, "w"), indent=1)
