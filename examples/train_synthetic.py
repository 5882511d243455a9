"""End-to-end example on synthetic data (needs an MI355X): polygon annotations -> GPU data pipeline ->
Solver (recorded pipelined step: the next batch's backbone and its data loading beside the current step) -> TF-format checkpoint -> reload into an
inference net -> detections + instance masks for one image.

    python examples/train_synthetic.py [--steps 40] [--size 192] [--batch 4] [--out /tmp/disyolo_example]

It mirrors what the reference's ``train_yolo3_mask.py:237-248`` (train) and ``calculate_test_map.py`` (test) do
with a real dataset; swap ``synthetic_labels`` for ``disyolo_amd.pre_process.load_verify_contour(path, 'train')``
records (plus the decoded images) to train on one.
"""
import argparse
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import disyolo_amd  # noqa: E402,F401
from disyolo_amd import checkpoint, config as cfg, postprocess  # noqa: E402
from disyolo_amd.net import YOLONet  # noqa: E402
from disyolo_amd.solver import Solver  # noqa: E402
from disyolo_amd.train_data import defect_train  # noqa: E402


def synthetic_labels(rng, n):
    """n records in the layout of the reference's ground-truth cache: an RGB image, one class name and one list of
    polygons ('out' = outline, 'in' = hole) per instance"""
    out = []
    for _ in range(n):
        h, w = rng.randint(200, 320), rng.randint(200, 320)
        image = rng.randint(0, 256, (h, w, 3)).astype(np.uint8)
        polys, names = [], []
        for _j in range(rng.randint(1, 4)):
            cy, cx = rng.uniform(0.3, 0.7) * h, rng.uniform(0.3, 0.7) * w
            ry, rx = rng.uniform(0.1, 0.25) * h, rng.uniform(0.1, 0.25) * w
            t = np.sort(rng.uniform(0, 2 * np.pi, rng.randint(6, 10)))
            polys.append([{"type": "out", "all_points_x": np.clip(cx + rx * np.cos(t), 0, w - 1).astype(int).tolist(),
                           "all_points_y": np.clip(cy + ry * np.sin(t), 0, h - 1).astype(int).tolist()}])
            names.append(cfg.CLASSES[rng.randint(0, len(cfg.CLASSES))])
        out.append({"image": image, "class_names": names, "polygons": polys})
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=40)
    ap.add_argument("--size", type=int, default=192)
    ap.add_argument("--batch", type=int, default=4)
    ap.add_argument("--out", default="/tmp/disyolo_example")
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    rng = np.random.RandomState(0)
    labels = synthetic_labels(rng, 16)

    # --- train: stage 1 (conv1-52 locked), batches built on the GPU from the polygon records
    data = defect_train(labels, batch_size=args.batch, image_size=args.size, device=dev, rng=np.random.RandomState(1))
    net = YOLONet(training=True, device=dev, image_size=args.size, batch_size=args.batch, stage=1, seed=0)
    solver = Solver(net, data, output_dir=args.out, max_iter=args.steps, summary_iter=max(1, args.steps // 4),
                    save_iter=args.steps, log=print)
    hist = solver.train()
    finite = [h for h in hist if np.isfinite(h)]
    print("loss: first %.1f -> last %.1f over %d steps (%d finite)" % (finite[0], finite[-1], len(hist), len(finite)))

    # --- test: reload the checkpoint the Solver wrote into an inference net
    prefix = checkpoint.latest_checkpoint(os.path.join(args.out, "checkpoint"))
    print("restoring", prefix)
    inf = YOLONet(training=False, device=dev, image_size=args.size, batch_size=1, stage=1, seed=123)
    checkpoint.restore_net(inf, prefix)
    rec = labels[0]
    from disyolo_amd.evaluate import image_read
    image, window = image_read(rec["image"], args.size, device=dev)
    window = torch.as_tensor(window, dtype=torch.float32).reshape(1, 4)
    det_boxes, det_masks = inf.evaluation(image[None], window, det_thresh=0.05, masks_on_device=True)
    h, w = rec["image"].shape[:2]
    entries, merged = postprocess.paste_detections(det_boxes[0], det_masks[0], h, w, args.size)
    print("%d detections on a %dx%d image; class map has %d labelled pixels" % (len(entries), h, w, int((merged > 0).sum())))


if __name__ == "__main__":
    main()
