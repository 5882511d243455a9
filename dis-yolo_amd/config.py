"""Hyper-parameters of the DIS-YOLO hot path.

Mirror of the reference's ``yolo/config.py:12-72`` (same names, same values) so
that callers written against ``cfg.*`` keep working.  Filesystem paths are the
only entries that differ (the reference ships a ``*** FILE PATH ***``
placeholder).  ``tests/golden/config.json`` is generated from the reference
module itself and ``tests/test_config.py`` checks this mirror against it.
"""
import os
import numpy as np

MODEL_PATH = os.environ.get("DISYOLO_MODEL_PATH", os.path.join(os.getcwd(), "DIS-YOLO"))
DATASET = os.path.join(MODEL_PATH, "data")
OUTPUT_DIR = os.path.join(MODEL_PATH, "output")
WEIGHTS_FILE = os.path.join(MODEL_PATH, "pretrained_weights", "yolov3_3class_coco.ckpt")

GPU = "0"

# yolo/config.py:21-22
CLASSES = ["crack", "spall", "rebar"]
ANCHORS = np.array([[31, 23], [62, 58], [143, 91], [213, 186], [61, 337], [194, 432],
                    [474, 248], [551, 93], [478, 454]], dtype=np.float32)

FLIPPED = True
BLUR_NOISE_LIGHT = True

MAX_ITER = 10000
SUMMARY_ITER = 50
SAVE_ITER = 500

ALPHA = 0.1

BATCH_SIZE = 2
IMAGE_SIZE = 576
K_MAP = 3

BASE_GRID = int(IMAGE_SIZE / 32)

OBJECT_SCALE = 2.0
NOOBJECT_SCALE = 1.0
CLASS_SCALE = 1.0
COORD_SCALE = 1.0
MASK_SCALE = 5.0
SCORE_SCALE = 2.0

IGNORE_THRESH = 0.5
OBJ_THRESHOLD = 0.25
IOU_THRESHOLD = 0.3
TEST_SIZE = 576
MAX_BOX_PER_IMAGE = 20
MAX_DETECTION = 30

# --- values hard-coded in the reference graph rather than in its config -----
BN_DECAY = 0.997          # yolo/yolo3_net_pos.py:74
BN_EPSILON = 1e-5         # yolo/yolo3_net_pos.py:75
L2_WEIGHT = 1e-4          # yolo/yolo3_net_pos.py:38
LEARNING_RATE = 1e-4      # train_yolo3_mask.py:38 (the later schedule is dead code)
ADAM_BETA1 = 0.9          # tf.train.AdamOptimizer defaults
ADAM_BETA2 = 0.999
ADAM_EPSILON = 1e-8
MASK_ROI_DET = 7          # yolo/yolo3_net_pos.py:783
MASK_ROI_GT = 3           # yolo/yolo3_net_pos.py:783
MASK_ROI_IOU = 0.5        # yolo/yolo3_net_pos.py:60
