"""The pinning hook (tests/golden/make_opencv_vectors.py) recovers OpenCV's NORMAL_LUT from the BEHAVIOUR of
cv::linemod::DepthNormal on synthetic planes.  OpenCV is not in this image, so the METHOD is checked here against a
stand-in with the same interface whose table is known: the CPU oracle's DepthNormal.  Every cell the probe reports must
carry the table's byte, and the cells it cannot reach with its safety margins must be a negligible part of the
hemisphere of possible normals."""
import importlib
import os
import sys
import types

import numpy as np

from conftest import GOLDEN


def test_normal_lut_probe_recovers_a_known_table(orc):
    sys.path.insert(0, GOLDEN)
    hook = importlib.import_module("make_opencv_vectors")
    table = orc.normal_lut().copy()
    # a table the probe cannot guess: permute the labels of the built-in one
    perm = np.array([0] + [1 << ((k * 3 + 5) % 8) for k in range(8)], np.uint8)
    idx = np.zeros(256, np.uint8)
    for k in range(8):
        idx[1 << k] = perm[k + 1]
    table = idx[table]

    class QP:
        def __init__(self, q):
            self.q = q

        def quantize(self):
            return self.q

    class Mod:
        def __init__(self, dist, diff):
            self.dist, self.diff = dist, diff

        def process(self, depth, mask):
            return QP(orc.depth_quantize(depth, self.dist, self.diff, lut=table))

    fake = types.SimpleNamespace(linemod=types.SimpleNamespace(DepthNormal_create=lambda a, b, c, d: Mod(a, b)))
    lut, votes, targeted = hook.recover_normal_lut(fake, out_dir=None, every=1)
    seen = votes.sum(1) > 0
    assert seen.sum() > 1300
    assert ((votes > 0).sum(1) <= 1).all()                       # no cell observed with two labels
    assert np.array_equal(lut[seen], table[seen])                # every observed cell carries the table's byte
    # share of the hemisphere of unit normals that falls into cells the probe did not observe
    rng = np.random.default_rng(0)
    u = rng.normal(size=(1000000, 3))
    u /= np.linalg.norm(u, axis=1)[:, None]
    u[:, 2] = -np.abs(u[:, 2])
    c = [np.floor(u[:, 0] * 10 + 10).astype(int), np.floor(u[:, 1] * 10 + 10).astype(int), np.floor(u[:, 2] * 20 + 20).astype(int)]
    ok = (c[0] < 20) & (c[1] < 20) & (c[2] < 20) & (c[2] >= 0)
    flat = c[2][ok] * 400 + c[1][ok] * 20 + c[0][ok]
    assert (~seen[flat]).mean() < 0.005
