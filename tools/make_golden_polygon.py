"""Golden vectors for the polygon rasteriser: run with the interpreter that has scikit-image

    /opt/conda/bin/python3.9 tools/make_golden_polygon.py        (skimage 0.18.3 in the build container)

The reference rasterises its annotation polygons with ``skimage.draw.polygon(y_points, x_points)``
(utils/train_data.py:321-338, utils/validation_map.py:70-86).  scikit-image is a third-party dependency of
the reference (unpinned there); this script records what the installed version returns for a set of
polygons so that the oracle's restatement and the HIP kernel can be checked without it."""
import json
import os
import sys

import numpy as np
import skimage
import skimage.draw

rng = np.random.RandomState(7)
cases = []


def add(name, ys, xs):
    rr, cc = skimage.draw.polygon(np.asarray(ys), np.asarray(xs))
    cases.append({"name": name, "y": [int(v) for v in ys], "x": [int(v) for v in xs],
                  "rr": [int(v) for v in rr], "cc": [int(v) for v in cc]})


add("triangle", [2, 2, 12], [3, 17, 9])
add("square_axis_aligned", [5, 5, 15, 15], [4, 14, 14, 4])
add("square_closed_duplicate_last", [5, 5, 15, 15, 5], [4, 14, 14, 4, 4])
add("concave_L", [1, 1, 8, 8, 16, 16], [1, 8, 8, 18, 18, 1])
add("thin_crack", [3, 4, 20, 19], [2, 2, 30, 30])
add("degenerate_line", [4, 4, 4], [2, 9, 15])
add("single_point", [6, 6, 6], [6, 6, 6])
add("two_points", [3, 9], [4, 12])
add("self_intersecting_bowtie", [2, 14, 2, 14], [2, 14, 14, 2])
add("touches_origin", [0, 0, 9], [0, 11, 5])
add("collinear_vertices", [2, 2, 2, 10, 10], [2, 8, 14, 14, 2])
add("clockwise_vs_ccw", [10, 2, 2, 10], [2, 2, 12, 12])
for k in range(12):
    n = rng.randint(3, 9)
    ys = rng.randint(0, 40, size=n)
    xs = rng.randint(0, 50, size=n)
    add("random%d" % k, ys, xs)
out = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "polygon.json")
json.dump({"generator": "tools/make_golden_polygon.py", "skimage": skimage.__version__, "cases": cases}, open(out, "w"))
print(len(cases), "cases ->", out)
