"""The committed tile tables bench.py loads by default (profiles/tune_<workload>.json): keys are conv shape keys, values known
tile codes; the training tables never pick the persistent streaming kernel (tile 20: it wins on the tuner's single lane and
loses beside the side lane's resident blocks -- tools/instep_tune.py, profiles/HISTORY.md, round 4)."""
import glob
import json
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
KNOWN_IDS = set(range(0, 22)) | {24, 25}


def test_committed_tile_tables_are_well_formed():
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "tune_train_*.json")) + glob.glob(os.path.join(ROOT, "profiles", "tune_infer_*.json")))
    assert len(files) >= 6
    for f in files:
        t = json.load(open(f))
        assert len(t) >= 20, f
        for k, v in t.items():
            key = json.loads(k)
            assert len(key) == 11 and all(isinstance(x, int) for x in key), (f, k)
            assert isinstance(v, int) and (v & 0xff) in KNOWN_IDS and (v & ~0x3ff) == 0, (f, k, v)
            if "tune_train_" in f:
                assert (v & 0xff) != 20, "streaming tile in a training table: %s %s" % (f, k)
