"""Writes tests/golden/viewpoints.txt: name, count and SHA-256 of the exact float sequence of every viewpoint
configuration tests/cpp/viewpoints_dump.cpp prints.  The sequences were checked, vertex by vertex, to be those of the
round-1 restatement of /root/reference/src/CameraViewPoints.cpp (whose templates land on benchmark/pose0.yml,
tests/test_facade.py::test_reference_benchmark_pose0) before the sampler was re-derived in round 2.
Run from the repo root:  python tests/golden/make_viewpoints_golden.py"""
import hashlib
import os
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def dump_lines():
    host = os.path.join(ROOT, "line-mod-pipeline_amd", "host")
    libdir = os.path.join(ROOT, "line-mod-pipeline_amd", "lib")
    with tempfile.TemporaryDirectory() as d:
        exe = os.path.join(d, "vpdump")
        subprocess.check_call(["g++", "-std=c++17", "-O1", "-o", exe, os.path.join(ROOT, "tests", "cpp", "viewpoints_dump.cpp")] +
                              [os.path.join(host, f) for f in ("TemplateGenerator.cpp", "HighLevelLinemod.cpp", "PostProcess.cpp")] +
                              ["-L" + libdir, "-llinemod_hip", "-Wl,-rpath," + libdir])
        return subprocess.check_output([exe], text=True).splitlines()


def summarise(lines):
    out = []
    for l in lines:
        name, count, *verts = l.split()
        assert int(count) == len(verts)
        out.append("%s %s %s" % (name, count, hashlib.sha256(" ".join(verts).encode()).hexdigest()))
    return out


if __name__ == "__main__":
    path = os.path.join(ROOT, "tests", "golden", "viewpoints.txt")
    open(path, "w").write("\n".join(summarise(dump_lines())) + "\n")
    print(open(path).read())
