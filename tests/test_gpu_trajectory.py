"""Multi-step training-mode parity (VERDICT r5 task 8): TEN optimizer steps of the recorded HIP step against the oracle stepping
the same TF-form Adam on the same batch, with the mask-loss shuffles injected -- the one comparison that covers more than one
``sess.run([total_loss, optimizer])`` (train_yolo3_mask.py:216) in a row: forward, both losses, the whole backward pass, Adam,
the re-packed operands and the moving statistics feeding the NEXT step.

Three trajectories from the same initial variables:
    f32  : the oracle in the reference's arithmetic (oracle/disyolo_oracle.py total_loss + adam_tf_step)
    bf16 : the oracle with weights / stored activations rounded to bf16 where the HIP path rounds them (quant=bf16_ste)
    HIP  : YOLONet.train_step on the GPU (recorded program, in-launch batch norm as shipped)
A randomly initialised stack of 30 batch-statistics BN layers amplifies every rounding (the bf16 oracle's own activations sit
tens of percent from the f32 oracle's at the last layers), so the yardstick for the HIP trajectory is the
bf16 oracle's own distance from the f32 oracle, step by step -- the yardstick the end-to-end inference gate uses.

The ten compared steps start from a COMMON state 40 HIP steps away from the random initialisation (variables, Adam moments and
step count, moving statistics handed to both oracles): from zero moments Adam moves every variable by the sign of its gradient,
and at the initialisation most gradients are at rounding-noise level -- ten steps from there were measured first
(profiles/r06_trajectory.json, "from_init"): the bf16 oracle's OWN ten-step update has cosine 0.30 with the f32 oracle's, the HIP
path's 0.32; that comparison says nothing about any of the three.

Bars (set from the measured values in profiles/r06_trajectory.json, with margin; stated in the assertions): first-step loss
within 1 % of the f32 oracle's (measured 0.4 %); rms relative distance of the HIP loss curve from the f32 oracle's <= the bf16
oracle's own + 0.5 % (0.34 % against 1.49 %) and from the bf16 oracle's <= 3 % (1.3 %); cosine of the ten-step update with the
f32 oracle's >= 0.99 (0.998; the bf16 oracle's own: 0.998), update norms within 1 % (0.003 %).
"""
import json
import os

import numpy as np
import pytest
import torch

import disyolo_oracle as O
from disyolo_amd import config as cfg
from test_gpu_net import make_net, oracle_params

pytestmark = pytest.mark.gpu

STEPS = 10
WARM = int(os.environ.get("TRAJ_WARM", "40"))
B, S = 2, 192


def _oracle_run(p0, m0, v0, t0, b, lock, perms, quant):
    p = {k: v.clone() for k, v in p0.items()}
    names = O.trainable_names(lock)
    m = {n: m0[n].clone() for n in names}
    v = {n: v0[n].clone() for n in names}
    losses = []
    for t in range(t0 + 1, t0 + STEPS + 1):
        tr = {n: p[n].clone().requires_grad_(True) for n in names}
        pp = dict(p)
        pp.update(tr)
        upd = {}
        parts, _, _, _ = O.total_loss(pp, b, lock, True, perms, upd, obj_thresh=0.1, quant=quant)
        parts["total"].backward()
        losses.append(float(parts["total"]))
        for n in names:
            p[n], m[n], v[n] = O.adam_tf_step(p[n], tr[n].grad, m[n], v[n], t)
        for n, val in upd.items():
            p[n] = val.detach()
    return np.asarray(losses), p


def test_ten_training_steps_follow_the_oracles_trajectory(dev):
    torch.manual_seed(0)
    net = make_net(dev, True, 1, B=B, S=S, seed=2)
    b = O.synthetic_batch(B, S, seed=21)
    rng = np.random.RandomState(3)
    b["perm_det"] = np.stack([rng.permutation(cfg.MAX_DETECTION) for _ in range(B)]).astype(np.int32)
    b["perm_gt"] = np.stack([rng.permutation(cfg.MAX_BOX_PER_IMAGE) for _ in range(B)]).astype(np.int32)
    perms = [(b["perm_det"][i], b["perm_gt"][i]) for i in range(B)]
    lock = O.default_lock(1)
    names = O.trainable_names(lock)

    net.set_batch(b)
    net.build_program(det_thresh=0.1)
    # warm-up on the HIP path: away from the random initialisation, where the first Adam steps (zero moments) move every
    # variable by the SIGN of a noise-level gradient and three implementations of the same arithmetic go three ways
    warm = [float(net.train_step().cpu()) for _ in range(WARM)]
    torch.cuda.synchronize()
    p0 = oracle_params(net)
    t0 = net.step_count
    assert t0 == WARM
    m0 = {n: net.adam_m[o:o + c].detach().cpu().reshape(p0[n].shape).clone() for n, (o, c) in net.arena_slices.items()}
    v0 = {n: net.adam_v[o:o + c].detach().cpu().reshape(p0[n].shape).clone() for n, (o, c) in net.arena_slices.items()}
    assert set(m0) == set(names)

    hip = np.asarray([float(net.train_step().cpu()) for _ in range(STEPS)])
    torch.cuda.synchronize()
    w_hip = {n: net.params[n].detach().cpu() for n in names}

    f32, w_f = _oracle_run(p0, m0, v0, t0, b, lock, perms, None)
    bf, w_q = _oracle_run(p0, m0, v0, t0, b, lock, perms, O.bf16_ste)

    def delta(w):
        return torch.cat([(w[n] - p0[n]).flatten().double() for n in names])
    dh, dq, df = delta(w_hip), delta(w_q), delta(w_f)
    cos = lambda a, c: float((a @ c) / (a.norm() * c.norm() + 1e-30))
    rms = lambda a: float(np.sqrt(np.mean(np.square(a))))
    report = {"steps": STEPS, "warm_up_steps": WARM, "B": B, "S": S, "loss_warm_first_last": [warm[0], warm[-1]],
              "loss_f32": f32.tolist(), "loss_bf16_oracle": bf.tolist(), "loss_hip": hip.tolist(),
              "abs_hip_minus_f32": np.abs(hip - f32).tolist(), "abs_bf16_minus_f32": np.abs(bf - f32).tolist(),
              "rms_rel_hip_vs_f32": rms((hip - f32) / f32), "rms_rel_bf16_vs_f32": rms((bf - f32) / f32),
              "rms_rel_hip_vs_bf16": rms((hip - bf) / bf),
              "ratio_last_over_first": {"f32": float(f32[-1] / f32[0]), "bf16": float(bf[-1] / bf[0]), "hip": float(hip[-1] / hip[0])},
              "cos_update_hip_vs_bf16": cos(dh, dq), "cos_update_bf16_vs_f32": cos(dq, df), "cos_update_hip_vs_f32": cos(dh, df),
              "update_norm": {"hip": float(dh.norm()), "bf16": float(dq.norm()), "f32": float(df.norm())}}
    out = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out")
    os.makedirs(out, exist_ok=True)
    with open(os.path.join(out, "r06_trajectory.json"), "w") as f:
        json.dump(report, f, indent=1)
    print(json.dumps(report))

    assert np.isfinite(hip).all()
    # the first of the ten steps is ONE forward pass from identical variables: the three losses agree closely
    assert abs(hip[0] - f32[0]) <= BAR_FIRST_STEP * abs(f32[0]), (hip[0], f32[0])
    # the HIP curve is no further from the f32 oracle's than the bf16-emulating oracle's own (rms over the ten steps; measured
    # 0.34 % against the bf16 oracle's 1.49 %) + 0.5 %
    assert report["rms_rel_hip_vs_f32"] <= report["rms_rel_bf16_vs_f32"] + 0.005, report
    assert report["rms_rel_hip_vs_bf16"] <= BAR_CURVE, report
    # the ten updates point the same way and are of the same size
    assert report["cos_update_hip_vs_f32"] >= min(BAR_COS, report["cos_update_bf16_vs_f32"] - 0.05), report
    assert abs(report["update_norm"]["hip"] / report["update_norm"]["f32"] - 1.0) <= 0.01, report


# bars: measured values in profiles/r06_trajectory.json (one MI355X), margin stated there
BAR_FIRST_STEP = 0.01     # measured 0.4 %
BAR_CURVE = 0.03          # rms relative distance from the bf16 oracle's curve: measured 1.3 %
BAR_COS = 0.99            # measured 0.998 (the bf16 oracle's own: 0.998)
