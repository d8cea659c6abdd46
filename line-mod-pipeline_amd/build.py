"""Builds liblinemod_hip.so (hand-written HIP kernels + C ABI) for gfx950 with hipcc, in-tree.

hipcc cross-compiles without a GPU, so this runs in the build container; the resulting .so is
git-ignored but travels to the GPU box with the repo snapshot.
"""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB_DIR = os.path.join(HERE, "lib")
LIB = os.path.join(LIB_DIR, "liblinemod_hip.so")
HIP_SOURCES = ["lm_k_preprocess.hip", "lm_k_scan.hip", "lm_k_refine.hip", "lm_k_post.hip",
               "lm_detector.hip", "lm_detector_post.hip", "lm_detector_gather.hip", "lm_detector_io.hip", "lm_detector_debug.hip", "lm_comm.hip"]
CXX_SOURCES = ["lm_host.cpp", "lm_extract.cpp", "lm_yaml.cpp"]
HEADERS = ["lm_common.h", "lm_kernels.h", "lm_dev.h", "lm_dev_color.h", "lm_dev_depth.h", "lm_dev_memories.h", "lm_detector_impl.h", "lm_host.h", "lm_extract.h", "lm_median25.h", "lm_yaml.h", "lm_comm.h",
           os.path.join("..", "..", "include", "linemod_hip.h")]
# -ffp-contract=off: the two float islands (fastAtan2 polynomial, normal normalisation, raw threshold)
# must round exactly like the oracle, which is built the same way.
FLAGS = ["-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-Wall", "-Wno-unused-function", "-Wno-unused-value"]
ARCH = "gfx950"


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.exists(d) and os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=False):
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    os.makedirs(LIB_DIR, exist_ok=True)
    obj_dir = os.path.join(HERE, "build")
    os.makedirs(obj_dir, exist_ok=True)
    headers = [os.path.join(CSRC, h) for h in HEADERS]
    for h in headers:
        if not os.path.exists(h):
            raise FileNotFoundError("header missing from the checkout: " + h)
    objs, cmds = [], []
    for src in HIP_SOURCES + CXX_SOURCES:
        sp = os.path.join(CSRC, src)
        if not os.path.exists(sp):
            raise FileNotFoundError("source file missing from the checkout: " + sp)
        obj = os.path.join(obj_dir, src + ".o")
        objs.append(obj)
        if force or _stale(obj, [sp] + headers + [os.path.abspath(__file__)]):
            cmd = [hipcc] + FLAGS
            if src.endswith(".hip"):
                cmd += ["--offload-arch=" + ARCH]
            else:
                cmd += ["-x", "c++"]
            cmds.append(cmd + ["-c", sp, "-o", obj])
    # the translation units are independent (r06: the kernel source is split by stage): compile them side by side, bounded by the CPUs
    # this process may use and by LM_BUILD_JOBS
    if cmds:
        from concurrent.futures import ThreadPoolExecutor
        try:
            ncpu = len(os.sched_getaffinity(0))
        except AttributeError:
            ncpu = os.cpu_count() or 1
        jobs = max(1, min(len(cmds), int(os.environ.get("LM_BUILD_JOBS", "0")) or min(ncpu, 8)))

        def run(cmd):
            if verbose:
                print(" ".join(cmd), file=sys.stderr)
            subprocess.check_call(cmd)
        with ThreadPoolExecutor(max_workers=jobs) as ex:
            list(ex.map(run, cmds))
    if force or _stale(LIB, objs):
        cmd = [hipcc, "-shared", "-fPIC", "--offload-arch=" + ARCH, "-o", LIB] + objs + ["-lz", "-ldl"]
        if verbose:
            print(" ".join(cmd), file=sys.stderr)
        subprocess.check_call(cmd)
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
