"""Converts the reference's own data files benchmark/img0.png + depth0.png (640x480 RGB8 +
uint16 millimetres; BSD-licensed, /root/reference/LICENSE.md) into tests/golden/frame0.npz.

Run in the build container only (needs /root/reference and PIL).  The colour image is stored in
BGR channel order, which is what cv::VideoCapture hands the reference (detector.cpp:24); depth is
what cv::imread(..., IMREAD_ANYDEPTH) returns (detector.cpp:25-26).
"""
import os
import numpy as np
from PIL import Image

REF = "/root/reference/benchmark"
HERE = os.path.dirname(os.path.abspath(__file__))

rgb = np.array(Image.open(os.path.join(REF, "img0.png")).convert("RGB"), dtype=np.uint8)
depth = np.array(Image.open(os.path.join(REF, "depth0.png")))
assert rgb.shape == (480, 640, 3) and depth.shape == (480, 640), (rgb.shape, depth.shape)
depth = depth.astype(np.uint16)
bgr = np.ascontiguousarray(rgb[:, :, ::-1])
np.savez_compressed(os.path.join(HERE, "frame0.npz"), bgr=bgr, depth=depth)
print("bgr mean", bgr.mean(), "depth min/max", depth.min(), depth.max(), "zeros %", (depth == 0).mean() * 100)
