"""CPU checks of the training-data pipeline's oracle: the polygon rasteriser against golden
vectors produced by scikit-image itself (the reference's dependency, utils/train_data.py:331), and the
placing / flipping helpers against hand-derived answers."""
import json
import os

import numpy as np

import disyolo_oracle as O

GOLD = os.path.join(os.path.dirname(__file__), "golden", "polygon.json")


def test_polygon_rasteriser_matches_scikit_image_golden():
    g = json.load(open(GOLD))
    assert g["skimage"].startswith("0.") and len(g["cases"]) >= 20
    for c in g["cases"]:
        rr, cc = O.draw_polygon(c["y"], c["x"])
        assert sorted(zip(rr.tolist(), cc.tolist())) == sorted(zip(c["rr"], c["cc"])), c["name"]


def test_instance_mask_holes_and_vertex_pixels():
    # a 10x10 square with a 4x4 hole: 'in' clears the interior AND the hole's own vertices are set again (:333-336)
    polys = [{"type": "out", "all_points_x": [2, 12, 12, 2], "all_points_y": [2, 2, 12, 12]},
             {"type": "in", "all_points_x": [5, 9, 9, 5], "all_points_y": [5, 5, 9, 9]}]
    m = O.instance_mask(polys, 16, 16)
    assert m[3, 3] and m[2, 2] and m[12, 12] and not m[13, 13]
    assert not m[7, 7] and not m[6, 8]                      # inside the hole
    assert m[5, 5] and m[9, 9] and m[5, 9] and m[9, 5]      # the hole's vertices stay set
    assert not m[5, 7]                                      # the hole's edge between two vertices is cleared


def test_scale_and_crop_places_pads_and_crops():
    im = np.arange(4 * 6 * 1, dtype=np.float32).reshape(4, 6, 1) + 1
    out = O.scale_and_crop(im, 6, 4, 2, 1, 8, 0.0)[..., 0]
    assert out.shape == (8, 8) and out[1, 2] == 1 and out[4, 7] == 24 and out[0].sum() == 0 and out[:, :2].sum() == 0
    out = O.scale_and_crop(im, 6, 4, -3, -2, 8, 127.0)[..., 0]       # negative offsets crop the image
    assert out[0, 0] == im[2, 3, 0] and out[1, 2] == im[3, 5, 0] and out[2, 0] == 127 and out[0, 3] == 127
    big = np.ones((20, 20, 1), np.float32)
    assert O.scale_and_crop(big, 20, 20, -5, -5, 8, 0.0).shape == (8, 8, 1)


def test_resize_u8_identity_and_place_mask_rounding():
    rng = np.random.RandomState(0)
    img = rng.randint(0, 256, (9, 7, 3)).astype(np.uint8)
    assert np.array_equal(O.resize_linear_u8(img, 7, 9), img)         # same size: taps (s, s+1, 0) -> identity
    m = np.zeros((4, 4), np.float32)
    m[:, 2:] = 1
    out = O.place_mask(m, 8, 8, 8, 0, 0, 1)
    # 2x upscaling of a step edge: values 0, .25, .75, 1 across the edge -> np.around -> 0, 0, 1, 1
    assert out[0].tolist() == [False, False, False, False, True, True, True, True]
    assert np.array_equal(O.place_mask(m, 8, 8, 8, 0, 0, 2), out[:, ::-1])


def test_motion_blur_restatement_equals_scipy_convolve2d():
    """pyblur.LinearMotionBlur (utils/train_data.py:490) is scipy.signal.convolve2d(img, LineKernel, mode='same',
    fillvalue=255.0).astype(uint8): scipy is importable here, so the oracle's arithmetic (tap order, f32 rounding, uint8
    truncation, the 255 border) is pinned against it; the kernel table itself is restated from pyblur (not installable)."""
    from scipy.signal import convolve2d
    rng = np.random.RandomState(5)
    img = rng.randint(0, 256, (61, 61, 3)).astype(np.uint8)
    for angle in (0, 45, 90, 135):
        for lt in (0, 1, 2):
            k = O.pyblur_line_kernel3(angle, lt)
            assert np.count_nonzero(k) == (3 if lt == 0 else 2) and k[1, 1] > 0 and abs(float(k.sum()) - 1) < 1e-6
            want = np.stack([convolve2d(img[:, :, c].astype(np.float32), k, mode="same", fillvalue=255.0).astype("uint8")
                             for c in range(3)], -1)
            np.testing.assert_array_equal(O.motion_blur3(img, angle, lt), want, err_msg="%d %d" % (angle, lt))
    # the conventions of pyblur's LineDictionary: 'right' keeps the SECOND anchor -- for 90 degrees that is the row
    # BELOW the centre, for 135 degrees the pixel below-right; a convolution reads the mirrored neighbour
    assert O.pyblur_line_kernel3(90, 1)[2, 1] > 0 and O.pyblur_line_kernel3(135, 1)[2, 2] > 0
    assert O.pyblur_line_kernel3(45, 1)[0, 2] > 0 and O.pyblur_line_kernel3(0, 2)[1, 0] > 0
    one = np.zeros((5, 5, 3), np.uint8)
    one[2, 2] = 90
    out = O.motion_blur3(one, 90, 1).astype(int)[:, :, 0]
    assert out[2, 2] == 45 and out[3, 2] == 45 and out[1, 2] == 0        # the impulse spreads DOWN
    assert out[0, 0] == 127                                              # border: (0 + 255) / 2


def test_sparse_target_rows_flipped_and_normalised_equal_the_dense_grids():
    """round 6: the loader keeps the three target grids as their non-zero rows (synth.assign_target_entries) and applies the flip
    (utils/train_data.py:187-226) and the /net_size (:250-257) to the rows.  Against the dense form of those lines -- mirrored grid
    copies, centres reflected where the object flag is set, the whole grid divided -- on random boxes, all three flips: equal."""
    from disyolo_amd.synth import assign_target_entries, assign_targets
    S, C = 192, 3
    rng = np.random.RandomState(4)
    for trial in range(40):
        n = int(rng.randint(0, 7))
        wh = rng.uniform(6, 150, (n, 2))
        ctr = np.stack([rng.uniform(wh[:, 0] / 2, S - 1 - wh[:, 0] / 2), rng.uniform(wh[:, 1] / 2, S - 1 - wh[:, 1] / 2)], 1)
        bx = np.concatenate([ctr, wh], 1).astype(np.float32)
        cls = [int(v) for v in rng.randint(0, C, n)]
        flip = 1 + trial % 3
        # dense (the reference's form)
        grids = assign_targets(bx, cls, S, C)
        if flip == 2:
            grids = [g[:, ::-1].copy() for g in grids]
            for g in grids:
                obj = g[..., 4] == 1
                g[..., 0][obj] = S - 1 - g[..., 0][obj]
        elif flip == 3:
            grids = [g[::-1].copy() for g in grids]
            for g in grids:
                obj = g[..., 4] == 1
                g[..., 1][obj] = S - 1 - g[..., 1][obj]
        want = []
        for g in grids:
            g = g.copy()
            g[..., 0:4] = g[..., 0:4] / S
            want.append(g)
        # sparse (train_data.defect_train._fill)
        ents = assign_target_entries(bx, cls, S, C)
        got = [np.zeros_like(w) for w in want]
        for k, ent in enumerate(ents):
            gsz = got[k].shape[1]
            if flip == 2:
                for row in ent.values():
                    row[0] = np.float32(S - 1) - row[0]
                ent = {(yi, gsz - 1 - xi, a): row for (yi, xi, a), row in ent.items()}
            elif flip == 3:
                for row in ent.values():
                    row[1] = np.float32(S - 1) - row[1]
                ent = {(gsz - 1 - yi, xi, a): row for (yi, xi, a), row in ent.items()}
            for (yi, xi, a), row in ent.items():
                row[0:4] = row[0:4] / np.float32(S)
                got[k][yi, xi, a] = row
        for g, w in zip(got, want):
            assert np.array_equal(g, w), (trial, flip)
