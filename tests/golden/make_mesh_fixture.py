"""Converts the reference's data files models/lagergehaeuse.ply (ASCII PLY, 14136 vertices, 4712 triangles,
BSD-licensed) + models/lagergehaeuse.yml + benchmark/pose0.yml into tests/golden/lagergehaeuse.npz.
Run in the build container only (needs /root/reference)."""
import os
import re
import numpy as np

REF = "/root/reference"
HERE = os.path.dirname(os.path.abspath(__file__))
lines = open(os.path.join(REF, "models", "lagergehaeuse.ply")).read().split("\n")
nv = int([l for l in lines if l.startswith("element vertex")][0].split()[-1])
nf = int([l for l in lines if l.startswith("element face")][0].split()[-1])
h = lines.index("end_header") + 1
verts = np.array([[float(t) for t in l.split()[:3]] for l in lines[h:h + nv]], np.float32)
faces = np.array([[int(t) for t in l.split()[1:4]] for l in lines[h + nv:h + nv + nf]], np.int32)
assert verts.shape == (14136, 3) and faces.shape == (4712, 3)
pose = open(os.path.join(REF, "benchmark", "pose0.yml")).read()
nums = [float(x) for x in re.findall(r"[-+]?\d+\.\d+e[-+]\d+", pose)]
rot, pos = np.array(nums[:9]).reshape(3, 3), np.array(nums[9:12])
np.savez_compressed(os.path.join(HERE, "lagergehaeuse.npz"), vertices=verts, faces=faces, gt_rotation=rot, gt_position=pos,
                    lower_color_range=np.array([0., 0., 0.]), upper_color_range=np.array([255., 150., 255.]),
                    rotationally_symmetrical=np.array(1), planes_of_symmetry=np.array([1., 1., 1.]))
print("bbox", verts.min(0), verts.max(0), "gt position", pos)
