"""The config mirror and the oracle's restated constants against golden/config.json,
which tools/make_golden.py dumped from the reference's yolo/config.py."""
import json
import os

import numpy as np

import disyolo_oracle as O
from disyolo_amd import config as cfg

GOLD = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "config.json")))


def test_config_mirror_matches_reference_values():
    for k, v in GOLD.items():
        assert hasattr(cfg, k), "config mirror lacks %s" % k
        got = getattr(cfg, k)
        if isinstance(got, np.ndarray):
            assert got.dtype == np.float32
            np.testing.assert_array_equal(got, np.asarray(v, np.float32))
        else:
            assert got == v, (k, got, v)


def test_oracle_constants_match_reference_values():
    np.testing.assert_array_equal(O.ANCHORS, np.asarray(GOLD["ANCHORS"], np.float32))
    assert O.CLASSES == GOLD["CLASSES"]
    for ok, gk in (("ALPHA", "ALPHA"), ("K_MAP", "K_MAP"), ("OBJECT_SCALE", "OBJECT_SCALE"),
                   ("NOOBJECT_SCALE", "NOOBJECT_SCALE"), ("CLASS_SCALE", "CLASS_SCALE"), ("COORD_SCALE", "COORD_SCALE"),
                   ("MASK_SCALE", "MASK_SCALE"), ("IGNORE_THRESH", "IGNORE_THRESH"), ("OBJ_THRESHOLD", "OBJ_THRESHOLD"),
                   ("IOU_THRESHOLD", "IOU_THRESHOLD"), ("MAX_BOX_PER_IMAGE", "MAX_BOX_PER_IMAGE"),
                   ("MAX_DETECTION", "MAX_DETECTION")):
        assert getattr(O, ok) == GOLD[gk], ok
