"""End-to-end box / class / mask agreement of the HIP inference path with the f32 oracle (tests/e2e_parity.py) as a
report: python tools/e2e_parity_report.py [out.json]   (GPU box; ~1 min per configuration on 128 host cores)"""
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

out = sys.argv[1] if len(sys.argv) > 1 else None
dev = torch.device("cuda:0")
report = []
FIXTURES = [(576, 8, 5, 3000), (576, 8, 7, 2000), (832, 4, 5, 3000), (832, 4, 7, 2000)]     # = tests/test_gpu_e2e_parity.py
for S, B, seed, steps in FIXTURES:
    batch = E.painted_batch(B, S, seed=seed)
    t0 = time.time()
    sd, curve = E.train_overfit(dev, batch, B, S, steps)
    print("S=%d B=%d seed=%d: %d steps in %.1f s, loss %s" % (S, B, seed, steps, time.time() - t0, [round(c, 2) for c in curve]), flush=True)
    for thr in (0.25, 0.5):
        oracle = E.oracle_pair(sd, batch["images"], batch["clip_window"], thr)
        for Bi in ((B, 1) if thr == 0.25 else (B,)):
            t0 = time.time()
            r = E.compare(dev, sd, batch["images"][:Bi], batch["clip_window"][:Bi], S, thr, oracle=oracle)
            r["seed"], r["train_steps"], r["loss_curve"], r["seconds"] = seed, steps, [round(c, 2) for c in curve], round(time.time() - t0, 1)
            print(json.dumps({k: v for k, v in r.items() if k != "pairs"}), flush=True)
            for q in r["pairs"]:
                print("  pair img %d cand %5d score f32 %.3f bf16 %.3f hip %.3f | pair diff %s iou %.3f | d_hip/d_yard txy %.3f/%.3f twh %.3f/%.3f conf %.3f/%.3f cls %.3f/%.3f"
                      % (q["image"], q["candidate"], q["score_f32"], q["score_bf16_same_cell"], q["score_hip_same_cell"],
                         "%.3f" % q["pair_score_diff"] if q["pair_score_diff"] is not None else "  -  ", q["pair_box_iou"],
                         q["d_hip_txy"], q["d_yard_txy"], q["d_hip_twh"], q["d_yard_twh"], q["d_hip_conf"], q["d_yard_conf"],
                         q["d_hip_cls"], q["d_yard_cls"]), flush=True)
            report.append(r)
if out:
    json.dump(report, open(out, "w"), indent=1)
