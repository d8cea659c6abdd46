"""The viewpoint sampler of the host facade (lmamd::CameraViewPoints, line-mod-pipeline_amd/host/TemplateGenerator.cpp):
its vertex ORDER numbers the templates of a bank (/root/reference/src/TemplateGenerator.cpp:41-62 walks getVertices()),
so the exact sequences are pinned (tests/golden/viewpoints.txt), and their geometry is checked independently."""
import importlib.util
import os

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
spec = importlib.util.spec_from_file_location("make_viewpoints_golden", os.path.join(ROOT, "tests", "golden", "make_viewpoints_golden.py"))
gold = importlib.util.module_from_spec(spec)
spec.loader.exec_module(gold)


def _verts(line):
    name, count, *v = line.split()
    a = np.array([[np.frombuffer(bytes.fromhex(h[8 * k:8 * k + 8]), ">f4")[0] for k in range(3)] for h in v], np.float32)
    return name, int(count), a.reshape(-1, 3)


def test_viewpoint_sequences_are_pinned(lm):
    lines = gold.dump_lines()
    want = open(os.path.join(ROOT, "tests", "golden", "viewpoints.txt")).read().split("\n")[:-1]
    assert gold.summarise(lines) == want
    by = {n: (c, a) for n, c, a in map(_verts, lines)}
    # BASELINE.json's template counts: the shipped model gives 13 viewpoints per radius (x 15 radii x 10 rotations = 1950),
    # a non-symmetric object 162 at subdivision 2 (x 150 = 24 300, config 4) and 642 at the shipped subdivision 3
    assert by["shipped_model_r500_s3"][0] == 13 and by["sphere_r850_s2"][0] == 162 and by["sphere_r500_s3"][0] == 642
    for n, radius, sub in (("sphere_r500_s0", 500, 0), ("sphere_r500_s1", 500, 1), ("sphere_r850_s2", 850, 2), ("sphere_r500_s3", 500, 3)):
        c, a = by[n]
        assert c == 10 * 4 ** sub + 2                                    # V of an icosphere
        assert np.allclose(np.linalg.norm(a.astype(np.float64), axis=1), radius, rtol=1e-5)
        assert len({tuple(v) for v in a.round(2)}) == c                  # no duplicates
        assert np.allclose(a.astype(np.float64).mean(axis=0), 0, atol=radius * 1e-4)   # symmetric about the origin
    # the first 12 vertices of every sphere are the icosahedron itself: (+-a, 0, +-b) and cyclic, a = r / sqrt(phi^2 + 1)
    a = by["sphere_r500_s3"][1][:12].astype(np.float64)
    phi = 1.61803398875
    aa = np.sqrt(500.0 ** 2 / (phi * phi + 1))
    assert sorted(np.unique(np.abs(a).round(2))) == sorted({0.0, round(aa, 2), round(aa * phi, 2)})
    # the meridian arc: y = r sin, z = r cos of whole degrees 0, 7, 14, ... (uint16 truncation of 7.5), pruned to y, z >= 0
    c, a = by["shipped_model_r500_s3"]
    deg = np.degrees(np.arctan2(a[:, 1].astype(np.float64), a[:, 2].astype(np.float64)))
    assert np.allclose(deg, np.arange(13) * 7, atol=1e-3) and np.all(a[:, 0] == 0)
    c, a = by["octant_r600_s2"]
    assert np.all(a >= 0)
    c, a = by["halfspace_r600_s2"]
    assert np.all(a[:, 2] >= 0) and c > 81
