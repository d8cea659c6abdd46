"""Copies the reference's three cv::FileStorage DATA files (BSD-licensed; settings and ground truth, no code) into
tests/golden/reference_data/, byte for byte: linemod_settings.yml and models/lagergehaeuse.yml (written by the reference's
author in FileStorage syntax, read by the reference with fs["key"] >> x) and benchmark/pose0.yml (WRITTEN by cv::FileStorage:
the benchmark's ground-truth pose).  tests/test_yaml.py parses these very bytes with the library's FileStorage reader.
Run in the build container only (needs /root/reference)."""
import os
import shutil

REF = "/root/reference"
HERE = os.path.dirname(os.path.abspath(__file__))
out = os.path.join(HERE, "reference_data")
os.makedirs(out, exist_ok=True)
for src, dst in (("linemod_settings.yml", "linemod_settings.yml"), ("models/lagergehaeuse.yml", "lagergehaeuse.yml"),
                 ("benchmark/pose0.yml", "pose0.yml")):
    shutil.copyfile(os.path.join(REF, src), os.path.join(out, dst))
    print(dst, os.path.getsize(os.path.join(out, dst)), "bytes")
