#!/usr/bin/env python3
"""Headline benchmark: DIS-YOLO training images/sec at 576x576, bf16, on N MI355X.

    python bench.py --gpus N --steps K --warmup W            (N = 1)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

A "step" is one pass of the whole hot path over one synthetic batch that is already
resident in HBM: forward (Darknet-53 + 3 heads + mask subnet, training-mode BN on the
unlocked layers), detection filter (decode + per-class NMS + top-30), YOLO loss, mask-RoI
selection + position-sensitive mask loss, backward (dgrad / wgrad / BN), Adam, weight
re-pack -- the work of ``sess.run([total_loss, optimizer])`` (train_yolo3_mask.py:216).
Weak scaling: 8 images per GPU; gradients are all-reduced over RCCL, overlapped with
backward.  One JSON line is printed by rank 0.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
# RCCL creates enough streams to use up ROCm's default 4 hardware queues; the step's side lane
# then shares a queue with the main lane and the two stop overlapping (1416 -> 1215 img/s on
# one MI355X).  Round 5: with the exchange lane and a second communicator, 8 queues map two of the
# step's lanes onto one queue (4.1 -> 10.2 ms per step; 6, 10, 12 ... 32 are fine: profiles/r05_hw_queues.txt).
# Must be set before the HIP runtime initialises.
os.environ.setdefault("GPU_MAX_HW_QUEUES", "12")

import numpy as np
import torch
import torch.distributed as dist

import disyolo_amd  # noqa: F401
from disyolo_amd import lib as L
from disyolo_amd import config as cfg
from disyolo_amd.net import YOLONet
from disyolo_amd.synth import synthetic_batch

# algorithmic work, SURVEY.md 8(d) / BASELINE.md 4 (FLOP = 2*Ho*Wo*Cout*Cin*k^2 over the convs)
FWD_GFLOP_PER_IMG_576 = 132.68
TRAIN_GFLOP_PER_IMG_576 = {1: 210.0, 2: 398.0}
MFMA_PEAK_TFLOPS = 2500.0      # bf16 dense, MI355X_MICROARCH.md "Peak BF16/FP16 MFMA"


HBM_PEAK_TBS = 8.0             # spec; ~6.3 TB/s measured copy
# HBM bytes per launch per kernel instance from the committed PMC passes of this command (tools/profile_round.sh); the newest
# round's file whose kernel names match this library's instances (round 6 added a template argument to conv_igemm_kernel)
PMC_TRAFFIC_FILE = next((f for f in ("r06_pmc_traffic.json", "r05_pmc_traffic.json")
                         if os.path.exists(os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles", f))), "r05_pmc_traffic.json")


def box_fingerprint(dev, local_rank: int = 0):
    """What this box is, so that a slow draw is recognisable from the line itself (boxes of the pool differ by 3-7 % on
    identical code): rocm-smi's clocks / power cap / partition modes, and a fixed micro-benchmark measured BEFORE the
    workload -- one MFMA-bound layer (36^2 256 -> 512 3x3 at B = 8, 24.5 GFLOP, the launcher's own tile) and one
    512 MB device copy -- a few milliseconds of GPU time.  `normalise` in the headline's config uses nothing of this:
    it is a label, not a correction."""
    import subprocess
    fp = {}
    try:
        r = subprocess.run(["rocm-smi", "-d", str(local_rank), "--showclocks", "--showperflevel", "--showmaxpower", "--showpower",
                            "--showcomputepartition", "--showmemorypartition", "--json"], stdout=subprocess.PIPE,
                           stderr=subprocess.DEVNULL, timeout=20)
        js = json.loads(r.stdout.decode() or "{}")
        card = next(iter(js.values())) if js else {}
        keep = {}
        for k, v in card.items():
            kl = k.lower()
            if any(t in kl for t in ("sclk", "mclk", "fclk", "performance level", "max graphics package power", "power (w)",
                                     "socket graphics", "compute partition", "memory partition")):
                keep[k] = v
        fp["rocm_smi"] = keep
    except Exception as e:      # (no rocm-smi, no permission: the micro-benchmark below still labels the box)
        fp["rocm_smi"] = {"error": repr(e)[:120]}
    try:
        bf = torch.bfloat16
        x = torch.randn(8, 36, 36, 256, device=dev).to(bf)
        w = (torch.randn(512, 9 * 256, device=dev) * 0.02).to(bf)
        y = torch.empty(8, 36, 36, 512, dtype=bf, device=dev)
        d = L.make_conv_desc(x, w, y, 3, 1, tile=16)
        src = torch.empty(512 << 20, dtype=torch.uint8, device=dev)     # (beyond the 256 MB Infinity Cache: an HBM copy)
        dst = torch.empty_like(src)

        def timed(fn, n):
            for _ in range(5):
                fn()
            torch.cuda.synchronize()
            ts = []
            for _ in range(7):
                s0, e0 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                s0.record()
                for _ in range(n):
                    fn()
                e0.record()
                torch.cuda.synchronize()
                ts.append(s0.elapsed_time(e0) / n)
            return float(np.median(ts))
        ms_conv = timed(lambda: L.conv2d_fwd(d), 50)
        ms_copy = timed(lambda: dst.copy_(src), 20)
        fp["conv_36x36_256to512_3x3_B8_us"] = round(ms_conv * 1e3, 2)
        fp["conv_tflops"] = round(2.0 * 8 * 36 * 36 * 512 * 2304 / (ms_conv * 1e-3) / 1e12, 1)
        fp["copy_512MB_GBps_read_plus_write"] = round(2 * (512 << 20) / (ms_copy * 1e-3) / 1e9, 1)
        del x, w, y, src, dst
        # (no torch.cuda.empty_cache() here: handing the 1 GB of scratch back to the driver makes the B = 32 inference net
        # that allocates next run 23 % slower -- 6.09 k -> 4.67 k img/s same box, the allocation-history effect of
        # tools/micro/infer_after_alloc.py; left in torch's cache the blocks are reused)
    except Exception as e:
        fp["micro_error"] = repr(e)[:120]
    return fp


def bound_model(B: int, S: int, stage: int):
    """t_bound = sum over the step's kernels of max(F/P_peak, bytes/BW_peak), with the
    algorithmic bf16 bytes of every conv (in + weights + out [+ residual]; its data and weight
    gradients likewise), the BN / activation passes (2 resp. 5 tensor passes per training-mode
    layer) and Adam (28 B per parameter).  Returns seconds per step on one GPU."""
    from disyolo_amd.net import build_topology
    layers = build_topology(3, 3)
    spatial = {0: S}
    t = 0.0
    n_train = 0
    lock_upto = 52 if stage == 1 else 0

    def conv_t(M, N, K, in_b, w_b, out_b):
        return max(2.0 * M * N * K / (MFMA_PEAK_TFLOPS * 1e12), (in_b + w_b + out_b) / (HBM_PEAK_TBS * 1e12))

    for l in layers:
        H = spatial[l.src]
        Ho, _ = L.same_pads(H, l.k, l.stride)
        spatial[l.idx] = Ho
        M, N, K = B * Ho * Ho, l.cout, l.k * l.k * l.cin
        in_b, w_b, out_b = B * H * H * l.cin * 2, K * N * 2, M * N * 2
        res_b = out_b if l.shortcut is not None else 0
        t += conv_t(M, N, K, in_b, w_b, out_b + res_b)                       # forward
        if l.idx > lock_upto:
            n_train += K * N + (N if l.kind == "lin" else 2 * N)
            t += conv_t(M, N, K, in_b + out_b, 0, K * N * 4)                  # weight gradient (f32 out)
            if l.idx > lock_upto + 1:
                t += conv_t(M, N, K, out_b, w_b, in_b)                        # data gradient
            if l.kind != "lin":
                t += (2 + 5) * out_b / (HBM_PEAK_TBS * 1e12)                  # BN fwd (raw->act) + bwd passes
    t += 28.0 * n_train / (HBM_PEAK_TBS * 1e12)                               # Adam
    return t


def cpu_baseline(stage: int, size: int, budget_s: float = 25.0):
    """Reference CPU path stand-in: the oracle (torch-CPU f32 restatement of the reference
    graph; TensorFlow 1.x itself cannot run here) doing the same train step at
    the reference's own batch size (BATCH_SIZE = 2, yolo/config.py:41)."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import disyolo_oracle as O
    B = 2
    lock = O.default_lock(stage)
    params = O.init_params(0, lock, xavier_locked=True)
    batch = O.synthetic_batch(B, size, seed=1234)
    names = O.trainable_names(lock)
    m = {n: torch.zeros_like(params[n]) for n in names}
    v = {n: torch.zeros_like(params[n]) for n in names}

    def step(t):
        tr = {n: params[n].clone().requires_grad_(True) for n in names}
        pp = dict(params)
        pp.update(tr)
        upd = {}
        parts, _, _, _ = O.total_loss(pp, batch, lock, True, None, upd)
        parts["total"].backward()
        with torch.no_grad():
            for n in names:
                params[n], m[n], v[n] = O.adam_tf_step(params[n], tr[n].grad, m[n], v[n], t)
            params.update(upd)

    t0 = time.time()
    step(1)                      # warm-up (oneDNN primitive creation)
    warm = time.time() - t0
    n = max(1, min(5, int(budget_s / max(warm, 1e-3)) - 1))
    t0 = time.time()
    for i in range(n):
        step(2 + i)
    dt = (time.time() - t0) / n
    out = {"value": B / dt, "unit": "images/sec", "cores": torch.get_num_threads(), "kind": "port",
           "sample": "%d train steps of batch %d at %dx%d, stage %d, torch-CPU f32 oracle (not TF1.x); %.2f s/step"
                     % (n, B, size, size, stage, dt)}
    # BASELINE.json configs[0]: one 576x576 image through sess.run(net.evaluation) -- forward + detection filter + mask
    # assembly (calculate_test_map.py:218), the same oracle, inference-mode batch norm
    try:
        img = batch["images"][:1]
        win = np.asarray(batch["clip_window"][:1])

        def fwd():
            with torch.no_grad():
                y, mp = O.build_network(params, img, False, lock)
                pred = O.interpret_output(y)
                det = O.filter_detections(pred[2], pred[3], pred[5], win, 0.25)
                return O.val_test(det, mp)
        fwd()
        t0 = time.time()
        k = 0
        while k < 3 or (time.time() - t0 < 4.0 and k < 20):
            fwd()
            k += 1
        dt1 = (time.time() - t0) / k
        out["config1_forward"] = {"value": round(1.0 / dt1, 3), "unit": "images/sec", "cores": torch.get_num_threads(), "kind": "port",
                                  "sample": "%d x (1x%dx%d forward + filter_detections + val_test), torch-CPU f32 oracle; %.3f s/image"
                                            % (k, size, size, dt1)}
    except Exception as e:
        out["config1_forward"] = {"error": repr(e)[:160]}
    return out


def bench_infer(args, dev, world, rank):
    """Secondary measurement (BASELINE.json config 4): inference-only, batch 32 by default,
    network + decode/NMS + position-sensitive mask assembly, hipGraph-captured."""
    B = args.batch if args.batch != 8 else 32
    S = args.size
    net = YOLONet(training=False, device=dev, image_size=S, batch_size=B, stage=1, seed=0, dtype=args.dtype)
    batch = synthetic_batch(B, S, seed=1234 + rank)
    net._set_inputs(batch["images"], batch["clip_window"])
    if args.dtype == "fp8":
        net.calibrate_fp8()
    cache = tune_cache_path(args, "infer_B%d_%d%s" % (B, S, "" if args.dtype == "bf16" else "_fp8"))
    if args.autotune == "on":
        net.autotune(cache=cache)
    net.build_infer_program(graph=(args.mode in ("auto", "graph")))
    for _ in range(args.warmup):
        net.infer()
    dt, regions = repeated(lambda: net.infer(), args.steps, args.repeats, world, dev)
    if rank == 0:
        value = world * B * args.steps / dt
        gf = FWD_GFLOP_PER_IMG_576 * (S / 576.0) ** 2
        emit(({
            "metric": "inference images/sec @%dx%d bf16 (network + NMS + PS-RoI mask assembly)" % (S, S),
            "value": round(value, 2), "unit": "images/sec", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(dt / args.steps * 1e3, 3), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "bf16" if args.dtype == "bf16" else "fp8 (e4m3 storage + block-scaled MFMA operands, conv10-52) / bf16 (conv1-9 fused launches, heads, mask subnet)",
            "data": "synthetic",
            "config": {"workload": "infer_B%d_%dx%d_3class%s" % (B, S, S, "" if args.dtype == "bf16" else "_fp8"), "images_per_gpu": B,
                       "step_driver": "graph" if net._infer_graph is not None else "program",
                       "conv_tiles": ("table %s" % os.path.relpath(cache, ROOT)) if cache else
                                     ("autotuned in-sequence at setup" if args.autotune == "on" else "launcher heuristic"),
                       "detections_in_batch": int(net.det_count.sum().item())},
            "model_flops": {"fwd_gflop_per_image": round(gf, 2),
                            "achieved_tflops_per_gpu": round(gf * value / world / 1e3, 1),
                            "frac_of_mfma_peak": round(gf * value / world / 1e3 / MFMA_PEAK_TFLOPS, 4)},
            "reference_published": "README.md:23: ~0.1 s/image (10 img/s) on i7-7700 + GTX 1060, incl. host mask crop"}))
    if world > 1:
        torch.cuda.synchronize()
        dist.destroy_process_group()


def tune_cache_path(args, workload: str):
    """the committed tile table of this workload, a user-named file, or None (tune live)"""
    if args.tune_cache == "none" or args.autotune != "on":
        return None
    if args.tune_cache != "auto":
        return args.tune_cache
    p = os.path.join(ROOT, "profiles", "tune_%s.json" % workload)
    return p if os.path.exists(p) else None


def timed_region(step, steps: int, world: int, dev) -> float:
    """EXACTLY `steps` steps between barrier + synchronize on both sides; max over ranks (seconds)"""
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    return dt


def repeated(step, steps: int, repeats: int, world: int, dev):
    """the timed region repeated; returns (median seconds, all regions)"""
    ts = [timed_region(step, steps, world, dev) for _ in range(max(1, repeats))]
    return float(np.median(ts)), ts


def secondary_measurements(args, dev):
    """after the headline (single GPU): the stage-2 train step (all 82 layers trainable) and the B=32
    hipGraph-replayed inference of BASELINE.json configs[3] -- each in a CHILD process running this file.
    A process that has created and freed many GB of device tensors runs later workloads on worse-mapped
    memory (the B=32 inference drops from 5.1 k to 3.9 k img/s after a training net lived in the same
    process; tools/micro/infer_after_alloc.py shows it with allocations alone), so every line is measured
    the way a user would run it: in a fresh process."""
    import subprocess
    out = {}
    S = args.size
    common = ["--steps", "10", "--warmup", "3", "--repeats", "5", "--autotune", args.autotune,
              "--tune-cache", args.tune_cache, "--no-secondary", "--no-cpu-baseline", "--no-kernel-events", "--no-box"]

    def child(extra, size=S):
        r = subprocess.run([sys.executable, os.path.abspath(__file__)] + extra + ["--size", str(size)] + common,
                           stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, timeout=600)
        lines = [ln for ln in r.stdout.decode().splitlines() if ln.startswith("{")]
        if lines and json.loads(lines[-1]).get("error"):
            raise RuntimeError(json.loads(lines[-1])["error"])       # e.g. a non-finite loss: no throughput is taken from it
        if r.returncode != 0 or not lines:
            raise RuntimeError("child bench exited with %d" % r.returncode)
        return json.loads(lines[-1])

    try:
        d = child(["--stage", "2", "--batch", str(args.batch)])
        gf = TRAIN_GFLOP_PER_IMG_576[2] * (S / 576.0) ** 2
        out["train_stage2"] = {"workload": d["config"]["workload"], "value": d["value"], "unit": d["unit"],
                               "ms_per_step": d["ms_per_step"], "steps": d["steps"], "repeats": d.get("repeats"),
                               "frac_of_mfma_peak": round(gf * d["value"] / 1e3 / MFMA_PEAK_TFLOPS, 4),
                               "loss_first": d["config"].get("loss_first"), "loss_last": d["config"].get("loss_last"),
                               "final_total_loss": d["config"].get("final_total_loss"),
                               "conv_tiles": d["config"].get("conv_tiles"), "process": "child"}
    except Exception as e:   # a secondary line must never cost the headline
        out["train_stage2"] = {"error": repr(e)[:200]}
    # the loop a user runs, end to end: polygon records -> the GPU data pipeline (train_data.defect_train) -> Solver.train
    # (tools/solver_rate.py: host clock over the last 150 of 200 steps, device synchronised at the end)
    if args.stage == 1 and args.dtype == "bf16" and args.batch == 8 and S == 576:
        try:
            r = subprocess.run([sys.executable, os.path.join(os.path.dirname(os.path.abspath(__file__)), "tools", "solver_rate.py"),
                                "--steps", "200", "--json"], stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, timeout=600)
            lines = [ln for ln in r.stdout.decode().splitlines() if ln.startswith("{")]
            if r.returncode != 0 or not lines:
                raise RuntimeError("solver_rate exited with %d" % r.returncode)
            out["train_stage1_solver_loop"] = dict(json.loads(lines[-1]), process="child",
                                                   role="not a bench step: the reference's training loop (train_yolo3_mask.py:143-226) "
                                                        "with its data loader, as a user runs it")
        except Exception as e:
            out["train_stage1_solver_loop"] = {"error": repr(e)[:200]}
    # ... and the test loop (evaluate(): one image at a time, paste, mask mAP + mIoU; tools/evaluate_rate.py)
    if args.stage == 1 and args.dtype == "bf16" and S == 576:
        try:
            r = subprocess.run([sys.executable, os.path.join(os.path.dirname(os.path.abspath(__file__)), "tools", "evaluate_rate.py"),
                                "32", "--json"], stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, timeout=600)
            lines = [ln for ln in r.stdout.decode().splitlines() if ln.startswith("{")]
            if r.returncode != 0 or not lines:
                raise RuntimeError("evaluate_rate exited with %d" % r.returncode)
            out["evaluate_loop"] = dict(json.loads(lines[-1]), process="child",
                                        role="not a bench step: the reference's test loop as a user runs it (155 ms per image at the start of round 6)")
        except Exception as e:
            out["evaluate_loop"] = {"error": repr(e)[:200]}
    # the headline workload the way a training loop runs it: a new batch before every step (device-to-device set_batch
    # inside the timed region; the overlapped tail stays open across it -- nothing in the tail reads an input tensor)
    if args.stage == 1 and args.dtype == "bf16":
        try:
            d = child(["--stage", "1", "--batch", str(args.batch), "--feed", "per-step"])
            out["train_stage1_feed_per_step"] = {"workload": d["config"]["workload"], "value": d["value"], "unit": d["unit"],
                                                 "ms_per_step": d["ms_per_step"], "steps": d["steps"], "repeats": d.get("repeats"),
                                                 "what": d["config"].get("feed"), "step_overlap": bool(d["config"].get("step_overlap")),
                                                 "loss_first": d["config"].get("loss_first"), "loss_last": d["config"].get("loss_last"),
                                                 "process": "child"}
        except Exception as e:
            out["train_stage1_feed_per_step"] = {"error": repr(e)[:200]}
    # the same stage-1 workload with the locked backbone batched over two steps (an option of the training loop, not the
    # headline: every step still trains on its own batch and every image passes every layer exactly once)
    if args.stage == 1 and args.dtype == "bf16":
        try:
            d = child(["--stage", "1", "--batch", str(args.batch), "--pair"])
            out["train_stage1_backbone_pair"] = {"workload": d["config"]["workload"], "value": d["value"], "unit": d["unit"],
                                                 "ms_per_step": d["ms_per_step"], "steps": d["steps"], "repeats": d.get("repeats"),
                                                 "what": d["config"].get("backbone_pair"),
                                                 "loss_first": d["config"].get("loss_first"), "loss_last": d["config"].get("loss_last"),
                                                 "conv_tiles": d["config"].get("conv_tiles"), "process": "child"}
        except Exception as e:
            out["train_stage1_backbone_pair"] = {"error": repr(e)[:200]}
    # the step of rounds 1-5 (no backbone pipeline; the side lane's tail overlapped with the next replay's backbone) for continuity
    if args.stage == 1 and args.dtype == "bf16":
        try:
            d = child(["--stage", "1", "--batch", str(args.batch), "--pipeline", "off"])
            out["train_stage1_plain_step"] = {"workload": d["config"]["workload"], "value": d["value"], "unit": d["unit"],
                                              "ms_per_step": d["ms_per_step"], "steps": d["steps"], "repeats": d.get("repeats"),
                                              "what": "the same step without the backbone pipeline (the headline form of rounds 1-5)",
                                              "step_overlap": bool(d["config"].get("step_overlap")), "process": "child"}
        except Exception as e:
            out["train_stage1_plain_step"] = {"error": repr(e)[:200]}
    # the headline workload with the e4m3 backbone (NOT the headline: BASELINE.json's metric is bf16)
    if args.stage == 1 and args.dtype == "bf16":
        try:
            d = child(["--stage", "1", "--batch", str(args.batch), "--dtype", "fp8"])
            out["train_stage1_fp8_backbone"] = {"workload": d["config"]["workload"], "value": d["value"], "unit": d["unit"],
                                                "ms_per_step": d["ms_per_step"], "steps": d["steps"], "repeats": d.get("repeats"),
                                                "dtype": d["dtype"], "loss_first": d["config"].get("loss_first"),
                                                "loss_last": d["config"].get("loss_last"), "process": "child"}
        except Exception as e:
            out["train_stage1_fp8_backbone"] = {"error": repr(e)[:200]}
    # BASELINE.json configs[4] at its per-GPU size: 832x832, 4 images per GPU, stage 1, the locked backbone's conv10-52 in OCP
    # e4m3 on the block-scaled MFMA (round 6; conv1-9 keep their bf16 fused launches), and the same step in bf16 beside it
    for key, dt in (("train_832_fp8", "fp8"), ("train_832_bf16", "bf16")):
        try:
            d = child(["--stage", "1", "--batch", "4", "--dtype", dt], size=832)
            gf = TRAIN_GFLOP_PER_IMG_576[1] * (832 / 576.0) ** 2
            out[key] = {"workload": d["config"]["workload"], "value": d["value"], "unit": d["unit"],
                        "ms_per_step": d["ms_per_step"], "steps": d["steps"], "repeats": d.get("repeats"), "dtype": d["dtype"],
                        "frac_of_mfma_peak": round(gf * d["value"] / 1e3 / MFMA_PEAK_TFLOPS, 4),
                        "loss_first": d["config"].get("loss_first"), "loss_last": d["config"].get("loss_last"),
                        "conv_tiles": d["config"].get("conv_tiles"), "process": "child",
                        "role": ("the same step with a bf16 backbone, for comparison"
                                 if dt == "bf16" else
                                 "BASELINE configs[4] at its per-GPU size: conv10-52 of the locked backbone in e4m3 on "
                                 "v_mfma_scale_f32_16x16x128_f8f6f4 (unit scales), conv1-9 in their bf16 fused launches; the faster "
                                 "path since round 6 (profiles/r06_fp8_mx_layers.txt, r06_backbone_pipeline.txt)")}
        except Exception as e:
            out[key] = {"error": repr(e)[:200]}
    try:
        d = child(["--task", "infer", "--batch", "32"])
        out["infer_b32_graph"] = {"workload": d["config"]["workload"] + "_hipgraph (network + NMS + PS-RoI mask assembly)",
                                  "value": d["value"], "unit": d["unit"], "ms_per_step": d["ms_per_step"],
                                  "steps": d["steps"], "repeats": 5,
                                  "frac_of_mfma_peak": d["model_flops"]["frac_of_mfma_peak"], "process": "child"}
    except Exception as e:
        out["infer_b32_graph"] = {"error": repr(e)[:200]}
    try:     # the same batch with the e4m3 backbone (conv10-52 on the block-scaled MFMA)
        d = child(["--task", "infer", "--batch", "32", "--dtype", "fp8"])
        out["infer_b32_graph_fp8"] = {"workload": d["config"]["workload"] + "_hipgraph (network + NMS + PS-RoI mask assembly)",
                                      "value": d["value"], "unit": d["unit"], "ms_per_step": d["ms_per_step"], "dtype": d["dtype"],
                                      "steps": d["steps"], "repeats": 5, "process": "child"}
    except Exception as e:
        out["infer_b32_graph_fp8"] = {"error": repr(e)[:200]}
    # BASELINE.json configs[0] on the GPU: one 576x576 image through network + detection filter + mask assembly (what
    # calculate_test_map.py:218 runs per test image); the CPU oracle's number for the same call is cpu_baseline.config1_forward
    try:
        d = child(["--task", "infer", "--batch", "1"])
        out["infer_b1_graph"] = {"workload": d["config"]["workload"] + "_hipgraph (BASELINE configs[0] on the GPU)",
                                 "value": d["value"], "unit": d["unit"], "ms_per_step": d["ms_per_step"],
                                 "steps": d["steps"], "repeats": 5, "process": "child"}
    except Exception as e:
        out["infer_b1_graph"] = {"error": repr(e)[:200]}
    return out


def emit(obj) -> None:
    """the ONE JSON line goes to the real stdout; everything else any library prints to fd 1
    (RCCL's version banner, for one) has been re-routed to stderr by main()"""
    os.write(_REAL_STDOUT, (json.dumps(obj) + "\n").encode())


_REAL_STDOUT = 1



def spawn_workers(n: int) -> int:
    """`python bench.py --gpus N` without a launcher: run `python -m torch.distributed.run --nnodes=1 --nproc-per-node N
    --master-addr 127.0.0.1 --master-port P bench.py <same arguments>` as a child process (stdout / stderr inherited: rank 0's
    JSON line is the child's only stdout) and exit with its code.  The parent never initialises the GPU."""
    import socket
    import subprocess
    port = os.environ.get("MASTER_PORT")
    if not port:
        with socket.socket() as so:
            so.bind(("127.0.0.1", 0))
            port = str(so.getsockname()[1])
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n), "--master-addr", "127.0.0.1",
           "--master-port", port, os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")      # dmabuf IPC: what RCCL needs between the ranks on this pool
    print("bench.py: --gpus %d without WORLD_SIZE: spawning %s" % (n, " ".join(cmd[1:9])), file=sys.stderr)
    rc = subprocess.call(cmd, env=env, stdout=_REAL_STDOUT)     # (fd 1 of this process points at stderr: main())
    sys.exit(rc)


def dry_launch(world: int, rank: int, local_rank: int) -> int:
    """what the N > 1 launch path hands every rank, checked without a GPU: a gloo group over the launcher's rendezvous,
    every rank's environment gathered on rank 0, one JSON line"""
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29533")
    dist.init_process_group("gloo", rank=rank, world_size=world)
    mine = {"rank": rank, "local_rank": local_rank, "world_size": world, "pid": os.getpid(),
            "master": "%s:%s" % (os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"]),
            "gpu_max_hw_queues": os.environ.get("GPU_MAX_HW_QUEUES"),
            "hsa_enable_ipc_mode_legacy": os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY")}
    seen = [None] * world
    dist.all_gather_object(seen, mine)
    dist.barrier()
    if rank == 0:
        emit({"dry_launch": True, "n_gpus": world, "ranks": seen})
    dist.destroy_process_group()
    return 0


def main():
    global _REAL_STDOUT
    sys.stdout.flush()
    _REAL_STDOUT = os.dup(1)
    os.dup2(2, 1)
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--batch", type=int, default=8, help="images per GPU")
    ap.add_argument("--size", type=int, default=576)
    ap.add_argument("--stage", type=int, default=1, choices=(1, 2),
                    help="1 = conv1-52 locked (the reference's shipped source), 2 = all layers trainable")
    ap.add_argument("--task", default="train", choices=("train", "infer"),
                    help="train = the headline metric; infer = BASELINE.json config 4 (forward + detection filter + "
                         "PS-RoI mask assembly, hipGraph replay), reported as a secondary line")
    ap.add_argument("--dtype", default="bf16", choices=("bf16", "fp8"),
                    help="fp8: the locked backbone conv1-52 stores activations and feeds the matrix cores in OCP e4m3 "
                         "(per-tensor scales calibrated on the first batch); everything trainable stays bf16 "
                         "(BASELINE.json configs[4]; stage 1 only)")
    ap.add_argument("--repeats", type=int, default=10,
                    help="the timed region of --steps steps is repeated this often; the reported value is the median")
    ap.add_argument("--no-secondary", action="store_true", help="skip the stage-2 / B=32 inference lines of 'secondary'")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-kernel-events", action="store_true")
    ap.add_argument("--no-box", action="store_true", help="skip the box fingerprint (rocm-smi + the 36^2 layer / 512 MB copy micro-benchmark)")
    ap.add_argument("--pipeline", default="auto", choices=("auto", "on", "off"),
                    help="cross-step software pipeline of the locked backbone (stage 1): each step computes the "
                         "backbone forward of the NEXT batch on a third lane while it runs heads/losses/backward/Adam "
                         "of the current one (bit-identical results; +5.9 % since round 6: auto = on in stage 1)")
    ap.add_argument("--autotune", default="on", choices=("on", "off"),
                    help="time the conv tile candidates inside the layer sequence before recording the step "
                         "(setup, outside the timed region)")
    ap.add_argument("--tune-cache", default="auto",
                    help="JSON file to load the autotuned tiles from / save them to.  'auto' (default): the committed "
                         "table profiles/tune_<workload>.json when there is one for this workload (the run is then "
                         "bit-reproducible box to box and its losses are a regression canary), else tune live; "
                         "'none': always tune live")
    ap.add_argument("--overlap-tail", default="auto", choices=("auto", "on", "off"),
                    help="single GPU, list executor: the recorded step leaves its side lane's tail (the optimizer sweeps of the "
                         "slices that became final last, the last weight gradients) running into the next replay's locked-backbone "
                         "forward, with per-tensor dependencies (YOLONet.build_program(overlap_tail=True)); bit-identical results. "
                         "auto = on where it applies")
    ap.add_argument("--pair", action="store_true",
                    help="stage 1: the locked backbone runs once per TWO batches at batch size 2B (YOLONet backbone_pair), "
                         "the trainable part steps through the halves; an even number of steps, each on its own batch")
    ap.add_argument("--poison", action="store_true",
                    help="self-test of the loss canary: a NaN is written into one trainable weight before the timed "
                         "regions; the run must then FAIL (exit code 3, an \"error\" field, no throughput)")
    ap.add_argument("--feed", default="resident", choices=("resident", "per-step"),
                    help="resident (the contract: inputs in HBM when the timed region starts, every step replays them) or "
                         "per-step: what a training loop does -- a NEW batch before every step (two device-resident batches "
                         "alternate; set_batch's device-to-device copies are inside the timed region)")
    ap.add_argument("--force-dp", action="store_true",
                    help="initialise RCCL and run the bucketed gradient all-reduce path even with one rank (self-test)")
    ap.add_argument("--sync-bn", action="store_true",
                    help="data-parallel runs: batch-norm statistics over all ranks (SURVEY.md 8e option; default: per rank)")
    ap.add_argument("--mode", default="auto", choices=("auto", "graph", "program", "eager"),
                    help="how the step is driven: the recorded command list replayed by the native executor "
                         "(default; cut at the all-reduce points when N > 1), its hipGraph capture, or per-launch "
                         "Python calls")
    ap.add_argument("--dry-launch", action="store_true",
                    help="launch self-test (no GPU): every rank reports its RANK / LOCAL_RANK / WORLD_SIZE over a gloo group and "
                         "rank 0 prints them as one JSON line")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        # `python bench.py --gpus N` typed bare: start the N workers ourselves, the way the driver would
        # (torch.distributed.run, one rank per GPU, rendezvous on 127.0.0.1), as a CHILD process -- nothing in this
        # process has touched the GPU yet and nothing is exec'ed -- and hand its output and exit code through
        return spawn_workers(args.gpus)
    if world != args.gpus:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d" % (args.gpus, world))
    if args.dry_launch:
        return dry_launch(world, rank, local_rank)
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    use_dp = world > 1 or args.force_dp
    if use_dp:
        L.reserve_lanes()       # the step's side streams before the process group's (hardware-queue mapping, lib.reserve_lanes)
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        # the gradient all-reduces run on RCCL's own stream: give it high priority so the exchange
        # is not starved by the backward pass it overlaps
        try:
            opts = dist.ProcessGroupNCCL.Options(is_high_priority_stream=True)
            dist.init_process_group("nccl", device_id=dev, rank=rank, world_size=world, pg_options=opts)
        except (AttributeError, TypeError):
            dist.init_process_group("nccl", device_id=dev, rank=rank, world_size=world)

    B, S = args.batch, args.size
    if args.task == "infer":
        return bench_infer(args, dev, world, rank)
    if args.pair:
        args.warmup += args.warmup % 2       # (an even number of untimed steps: the timed regions start on an even step)
        if args.stage != 1 or use_dp or args.steps % 2 or args.mode not in ("auto", "program"):
            raise SystemExit("--pair needs --stage 1, one GPU, the list executor and an even --steps")
    box = box_fingerprint(dev, local_rank) if (rank == 0 and not args.no_box) else None
    net = YOLONet(training=True, device=dev, image_size=S, batch_size=B, stage=args.stage, seed=0, dtype=args.dtype,
                  backbone_pair=args.pair)
    if use_dp:
        from disyolo_amd.dp import enable_data_parallel
        enable_data_parallel(net, sync_bn=args.sync_bn)
    batch = synthetic_batch(B, S, seed=1234 + rank)
    net.set_batch(batch)           # inputs resident in HBM from here on
    if args.pair:
        net.set_batch(synthetic_batch(B, S, seed=4321 + rank), 1)     # the odd steps' batch
    if args.dtype == "fp8":
        net.calibrate_fp8()        # static per-tensor scales (setup, outside the timed region)
    torch.manual_seed(1234 + rank)
    gen = None                     # default CUDA generator
    workload = "train_B%d_%d_stage%d%s%s" % (B, S, args.stage, "" if args.dtype == "bf16" else "_fp8", "_pair" if args.pair else "")
    cache = tune_cache_path(args, workload)
    if args.autotune == "on":
        t_tune = time.perf_counter()
        picks = net.autotune(cache=cache)
        if rank == 0:
            print("autotune %.1f s: %d of %d conv shapes moved off the launcher heuristic"
                  % (time.perf_counter() - t_tune, sum(1 for v in picks.values() if v), len(picks)), file=sys.stderr)

    mode = args.mode
    if mode == "auto":
        # the two-lane command list replayed directly beats its hipGraph capture (ROCm serialises
        # the captured side lane): 1304 vs 1193 img/s on one MI355X
        mode = "program"

    net.shuffle_seed = 1234 + rank   # tf.random_shuffle of the mask-loss RoIs: on the device, every step

    n_trained = [0]

    feed_sets = None
    if args.feed != "resident":
        if args.pair:
            raise SystemExit("--feed per-step is not wired for --pair")
        feed_sets = []
        for q in range(2):
            hb = synthetic_batch(B, S, seed=1234 + rank + 7000 * q)
            feed_sets.append({k: (torch.as_tensor(v).to(dev) if v is not None else None) for k, v in hb.items()})

    pipe_on = [False]          # set below, once the step has been recorded

    def step():
        if feed_sets is not None:
            cur = feed_sets[n_trained[0] & 1]
            if pipe_on[0]:
                # the pipelined step: the labels of THIS step's batch, the images of the NEXT one (whose backbone pass it runs)
                cur = dict(cur)
                cur["images"] = feed_sets[(n_trained[0] + 1) & 1]["images"]
            with net.feed_context():     # (the pipelined step: on the net's feed stream, beside the replay that is running)
                net.set_batch(cur)
        net.train_step(None, want_loss=False)
        n_trained[0] += 1

    # per-kernel durations for the roofline: HIP events around every conv launch over K eager
    # steps of the same workload (events cannot time nodes inside a graph replay; the kernels
    # and their arguments are identical in all three modes)
    timer = None
    if not args.no_kernel_events:      # every rank runs these steps (they contain the collectives)
        for _ in range(2):
            step()
        torch.cuda.synchronize()
        timer = L.KernelTimer()
        L.TIMER = timer
        for _ in range(args.steps):
            step()
        torch.cuda.synchronize()
        L.TIMER = None
    # Rounds 1-3 measured the pipelined step SLOWER (958-1387 vs 1404 img/s) and "auto" left it off.  Round 6 measured it again, with
    # round 5's lanes (streams chosen by measuring how their hardware queues behave beside each other): 2106 vs 1988 img/s, +5.9 %
    # (profiles/r06_backbone_pipeline.txt) -- the third lane's locked-backbone convs fill what the backward pass leaves of the
    # CUs.  Same kernels, same arithmetic, bit-identical variables (tests/test_gpu_net.py::test_pipelined_backbone_step_equals_
    # plain_step); every timed step still passes ONE batch through the backbone and ONE through everything else.  "auto" = on where
    # it applies: stage 1 (a locked prefix), the list executor, no --pair.
    pipe = (args.stage == 1 and mode == "program" and not args.pair and args.pipeline in ("on", "auto")
            and os.environ.get("DISYOLO_SIDE_LANE", "1") != "0")
    if args.pipeline == "on" and not pipe:
        raise SystemExit("--pipeline on needs --stage 1, --mode program and no --pair")
    overlap = (args.overlap_tail != "off" and mode == "program" and (not use_dp or net.dp.inlist) and not args.pair and not pipe
               and os.environ.get("DISYOLO_SIDE_LANE", "1") != "0")
    if args.overlap_tail == "on" and not overlap:
        raise SystemExit("--overlap-tail on needs --mode program, no --pair / --pipeline, and (data parallel) the exchange in the list")
    if mode != "eager":
        net.build_program(graph=(mode == "graph"), pipeline_backbone=pipe, overlap_tail=overlap)
        if pipe:
            if feed_sets is not None:
                net.prime_pipeline(feed_sets[0]["images"], feed_sets[0]["clip_window"])
            else:
                net.prime_pipeline()   # backbone of the first batch, outside the timed region
            pipe_on[0] = True
    if args.poison:
        net.arena[net.n_decay // 2] = float("nan")
        net.refresh_weights()
    # losses are read OUTSIDE the timed regions: after the first recorded step, after the warm-up, after every region
    loss_trace = []
    for i in range(args.warmup):
        step()
        if i == 0:
            loss_trace.append(float(net.total_loss().cpu()))
    if args.warmup > 1:
        loss_trace.append(float(net.total_loss().cpu()))
    regions = []
    for _ in range(max(1, args.repeats)):
        regions.append(timed_region(step, args.steps, world, dev))
        loss_trace.append(float(net.total_loss().cpu()))
    dt = float(np.median(regions))
    loss = loss_trace[-1]
    dp_trace = None
    if net.dp is not None and net.dp.inlist:
        dp_trace = net.dp.describe()     # the collectives are commands of the step: there is no host-side wait to trace
    elif net.dp is not None:
        # after the timed regions: a few steps with the exchange traced (events around every bucket's wait)
        net.dp.trace = []
        for _ in range(5):
            step()
        torch.cuda.synchronize()
        dp_trace = net.dp.trace_summary()
        net.dp.trace = None
    if net.dp is not None and dp_trace is not None:
        # how many ranks the exchange really spans: a vector of ones summed through the step's OWN communicator (the kernel
        # library's RCCL communicator when the exchange is in the list, torch.distributed's otherwise) -- a line whose
        # collectives moved nothing (one rank) says so itself
        ones = torch.ones(16, device=dev)
        if net.dp.comm is not None:
            net.dp.comm.allreduce(ones)
        else:
            dist.all_reduce(ones)
        torch.cuda.synchronize()
        dp_trace["ranks_seen_by_rccl"] = int(round(float(ones[0].cpu())))
    net.check_cluster_sync()       # (an in-launch exchange that timed out anywhere in the run: fail, do not report a throughput)
    finite = bool(np.all(np.isfinite(loss_trace))) and bool(torch.isfinite(net.arena).all())
    if world > 1:
        f = torch.tensor([1.0 if finite else 0.0], device=dev)
        dist.all_reduce(f, op=dist.ReduceOp.MIN)
        finite = bool(f.item() > 0)

    if rank == 0:
        ms = dt / args.steps * 1e3
        value = world * B * args.steps / dt
        train_gflop = TRAIN_GFLOP_PER_IMG_576[args.stage] * (S / 576.0) ** 2
        out = {
            "metric": "train images/sec @%dx%d bf16" % (S, S),
            "value": round(value, 2), "unit": "images/sec", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(ms, 3), "repeats": len(regions),
            "ms_per_step_min_max": [round(min(regions) / args.steps * 1e3, 3), round(max(regions) / args.steps * 1e3, 3)],
            "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None,
            "dtype": "bf16" if args.dtype == "bf16" else "fp8 (e4m3 storage + block-scaled MFMA operands, conv10-52) / bf16 (conv1-9 fused launches, trainable layers)",
            "data": "synthetic",
            "config": {"workload": "train_step_B%d_%dx%d_3class_stage%d%s" % (B, S, S, args.stage, "" if args.dtype == "bf16" else "_fp8"),
                       "images_per_gpu": B, "global_batch": B * world, "image_size": S,
                       "stage": "1: conv1-52 locked (shipped reference source)" if args.stage == 1 else
                                "2: all 82 layers trainable",
                       "parallelism": "dp%d%s" % (world, "+syncbn" if (use_dp and args.sync_bn) else ""), "rccl_buckets": ((len(net.opt_chunks) if net.dp.inlist else len(net.dp.buckets)) + 1) if net.dp else 0, "optimizer": "adam(tf-form) lr=1e-4", "step_driver": mode,
                       "conv_tiles": ("table %s" % os.path.relpath(cache, ROOT)) if cache else
                                     ("autotuned in-sequence at setup" if args.autotune == "on" else "launcher heuristic"),
                       "backbone_pipeline": ("every step runs the locked backbone (conv1-52) of the NEXT batch on a third lane while it runs heads, "
                                             "losses, backward and Adam of the current one (outputs double-buffered; the first batch's "
                                             "backbone pass is primed before the timed regions, the last step's pass is for the batch after "
                                             "them: K steps = K backbone passes + K trainable passes); bit-identical to the plain step")
                                            if pipe else False,
                       "step_overlap": ("the side lane's tail of step t (last optimizer sweeps + re-pack, last weight gradients) runs into "
                                        "the locked-backbone forward of step t+1, per-tensor dependencies; bit-identical to the joined step")
                                       if overlap else False,
                       "backbone_pair": ("locked conv1-52 run once per two batches at 2B; every batch passes every layer once; "
                                         "steps alternate (backbone + trainable part | trainable part), ms_per_step is their mean")
                                        if args.pair else False,
                       "loss_first": round(loss_trace[0], 4) if np.isfinite(loss_trace[0]) else None,
                       "loss_last": round(loss, 4) if np.isfinite(loss) else None,
                       "final_total_loss": round(loss, 4) if np.isfinite(loss) else None,
                       "steps_trained": n_trained[0],
                       # the reference fetches total_loss with every step (train_yolo3_mask.py:216); here the loss stays on
                       # the device inside the timed regions and is read between them (loss_first / loss_last)
                       "loss_fetched_in_timed_region": False,
                       "feed": ("resident inputs replayed every step" if feed_sets is None else
                                "a new batch before every step (two device-resident batches alternate; set_batch inside the timed region)"),
                       "box": box,
                       # how the side lanes were chosen (csrc/runtime.hip pool_lane): a run whose probe fell back to an
                       # unmeasured stream says so here
                       "lanes": L.lanes_report(),
                       "bn_inkernel": {"forward_layers": sum(1 for l in net.layers if l.fused_fwd),
                                       "backward_layers": sum(1 for l in net.layers if l.fused_bwd),
                                       "what": "training-mode batch norm inside the conv launch (in-launch exchange of the statistics "
                                               "rows): layers whose bn_finalize + bn_act_fwd / colreduce + bn_bwd_finalize + bn_bwd_apply "
                                               "launches are gone (profiles/r06_bn_inkernel.txt)"}},
            "parity": {"status": "partial: oracle unpinned against TF1.x (no TF, no reference vectors for the graph)",
                       "end_to_end_tolerance": "HIP inference vs f32 oracle on nets trained with per-step input jitter (2 seeds x 2 lengths) at 576^2 B=8/B=1 and "
                                               "832^2 B=4/B=1: >= 95 % of the oracle's detections found with the same class at box IoU >= 0.75, >= 90 % at "
                                               "IoU >= 0.9; every matched pair |score diff| <= 0.12 unless the bf16-emulating oracle moves THAT detection too "
                                               "(then <= 1.5 x its move + 0.03); at every detected cell the raw t_xy / t_wh / confidence / class logits within "
                                               "1.5 x the bf16 oracle's own deviation + 0.15 / 0.05 / 0.25 / 0.2; mask IoU (> 0.5) >= 0.85 each / >= 0.95 mean",
                       "evidence": "tests/test_gpu_e2e_parity.py, profiles/r05_e2e_parity.json (per-pair table)"},
            "model_flops": {"train_gflop_per_image": round(train_gflop, 1),
                            "achieved_tflops_per_gpu": round(train_gflop * value / world / 1e3, 1),
                            "frac_of_mfma_peak": round(train_gflop * value / world / 1e3 / MFMA_PEAK_TFLOPS, 4)},
        }
        if dp_trace is not None:
            out["dp_exchange"] = dp_trace
        if not finite:
            # a diverged run does less work (NaN scores empty the NMS and the mask-loss chain): its time is not a
            # measurement.  No throughput is reported and the process fails.
            out["error"] = "non-finite loss or weights (loss trace %s)" % [None if not np.isfinite(v) else round(v, 2) for v in loss_trace]
            out["value"] = None
            out["ms_per_step"] = None
            emit(out)
            if use_dp:
                torch.cuda.synchronize()
                dist.destroy_process_group()
            sys.exit(3)
        tb = bound_model(B, S, args.stage)
        out["bound_model"] = {"t_bound_ms": round(tb * 1e3, 3), "achieved_vs_bound": round(tb * 1e3 / ms, 4),
                              "definition": "sum over kernels of max(FLOP/2.5 PF, algorithmic bytes/8 TB/s): convs "
                                            "fwd/dgrad/wgrad, BN passes, Adam (SURVEY.md 8d)"}
        kernels = None
        if timer is not None:
            summ = timer.summary()
            kernels = {}
            for name, r in summ.items():
                avg_ms = r["ms_total"] / r["launches"]
                kernels[name] = {"launches_per_step": r["launches"] / args.steps, "avg_us": round(avg_ms * 1e3, 2),
                                 "ms_per_step": round(r["ms_total"] / args.steps, 3),
                                 "tflops": round(r["flops_total"] / (r["ms_total"] * 1e-3) / 1e12, 1)}
            dom = max(summ, key=lambda k: summ[k]["ms_total"])
            r = summ[dom]
            achieved = r["flops_total"] / (r["ms_total"] * 1e-3) / 1e12
            traffic, traffic_src = None, None
            try:   # HBM bytes per launch from the committed PMC passes (rocprofv3 --pmc, see the file's "method")
                pmc = json.load(open(os.path.join(ROOT, "profiles", PMC_TRAFFIC_FILE)))
                key = dom.replace(",", ", ")
                if key in pmc["kernels"]:
                    traffic = pmc["kernels"][key]["hbm_mb_per_launch_corrected"] * 1e6
                    traffic_src = "profiles/%s (separate rocprofv3 --pmc passes of this command at %s)" % (
                        PMC_TRAFFIC_FILE, pmc.get("measured_at", "?"))
            except Exception:
                pass
            ridge = MFMA_PEAK_TFLOPS * 1e12 / (HBM_PEAK_TBS * 1e12)
            alg_bytes = r.get("bytes_total", 0) / r["launches"]
            intensity = (r["flops_total"] / r["bytes_total"]) if r.get("bytes_total", 0) > 0 else float("inf")
            out["roofline"] = {"bound": "mfma" if intensity >= ridge else "hbm",
                               "bound_from": "algorithmic FLOP per byte of this instance %.0f vs ridge %.0f (2.5 PF / 8 TB/s)" % (intensity, ridge),
                               "kernel": dom, "timed_with": "HIP events, %d eager steps of the same workload" % args.steps, "achieved": round(achieved, 1),
                               "peak": MFMA_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": round(achieved / MFMA_PEAK_TFLOPS, 4),
                               "traffic": traffic, "traffic_unit": "HBM bytes per launch (PMC FETCH_SIZE x2 + WRITE_SIZE)",
                               "algorithmic_bytes": round(alg_bytes),
                               "algorithmic_bytes_definition": "inputs + packed weights + output (+ residual), each once, averaged over this instance's launches",
                               "traffic_source": traffic_src,
                               "flops_per_launch": round(r["flops_total"] / r["launches"] / 1e9, 3),
                               "flops_per_launch_unit": "GFLOP (algorithmic, 2*M*N*K averaged over this kernel's launches)",
                               "avg_launch_us": round(r["ms_total"] / r["launches"] * 1e3, 2),
                               "launches_per_step": r["launches"] / args.steps}
            # the conv instance below the ridge (algorithmic FLOP per byte < 2.5 PF / 8 TB/s) with the largest total time:
            # what it reaches of the HBM roofline, from the same HIP-event timings
            hb = {k: r for k, r in summ.items() if r.get("bytes_total", 0) > 0 and r["flops_total"] / r["bytes_total"] < ridge}
            if hb:
                hk = max(hb, key=lambda k: hb[k]["ms_total"])
                r = hb[hk]
                gbs = r["bytes_total"] / (r["ms_total"] * 1e-3) / 1e9
                out["roofline"]["hbm_bound_instance"] = {
                    "kernel": hk, "achieved": round(gbs, 1), "peak": HBM_PEAK_TBS * 1e3, "unit": "GB/s",
                    "hbm_frac": round(gbs / (HBM_PEAK_TBS * 1e3), 4), "launches_per_step": r["launches"] / args.steps,
                    "avg_launch_us": round(r["ms_total"] / r["launches"] * 1e3, 2),
                    "bytes_per_launch": round(r["bytes_total"] / r["launches"]),
                    "bytes_definition": "algorithmic: inputs + packed weights + output (+ residual), each once"}
        if world == 1 and not args.no_secondary and args.stage == 1 and B == 8 and args.dtype == "bf16":
            del timer
            out["secondary"] = secondary_measurements(args, dev)
            # the loop a user runs -- a new batch before every step -- beside the replay number (VERDICT r5 task 7)
            fps = out["secondary"].get("train_stage1_feed_per_step", {})
            out["config"]["value_per_step_feed"] = fps.get("value")
            out["config"]["ms_per_step_per_step_feed"] = fps.get("ms_per_step")
            # ... and the whole training loop with its data loader (Solver.train over the GPU data pipeline, tools/solver_rate.py)
            out["config"]["value_solver_loop"] = out["secondary"].get("train_stage1_solver_loop", {}).get("value")
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(args.stage, S)
        if kernels is not None:
            # last, and only the heaviest instances: a log tail that is cut still carries secondary / cpu_baseline
            top = sorted(kernels, key=lambda k: -kernels[k]["ms_per_step"])[:12]
            out["kernels"] = {k: kernels[k] for k in top}
            out["kernels_omitted"] = {"instances": len(kernels) - len(top),
                                      "ms_per_step": round(sum(v["ms_per_step"] for k, v in kernels.items() if k not in top), 3)}
        emit(out)
    if use_dp:
        torch.cuda.synchronize()
        if net.dp is not None and net.dp.comm is not None:
            net.dp.comm.close()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
