import numpy as np, subprocess, os, sys, importlib
ROOT=os.path.dirname(os.path.abspath(__file__)); sys.path.insert(0, ROOT)
lm = importlib.import_module("line-mod-pipeline_amd")
d="/tmp/e2e"; os.makedirs(d, exist_ok=True)
g=np.load(os.path.join(ROOT,"tests/golden/lagergehaeuse.npz")); f=np.load(os.path.join(ROOT,"tests/golden/frame0.npz"))
with open(d+"/mesh.bin","wb") as fh:
    fh.write(np.array([len(g["vertices"]), len(g["faces"])], np.uint32).tobytes()); fh.write(g["vertices"].astype(np.float32).tobytes()); fh.write(g["faces"].astype(np.int32).tobytes())
f["bgr"].tofile(d+"/bgr.raw"); f["depth"].tofile(d+"/depth.raw")
libdir=os.path.dirname(lm.LIB_PATH); H=os.path.join(ROOT,"line-mod-pipeline_amd","host")
exe=d+"/pose_e2e"
subprocess.check_call(["g++","-std=c++17","-O2","-o",exe,os.path.join(ROOT,"tests/cpp/pose_e2e.cpp"),H+"/HighLevelLinemod.cpp",H+"/PostProcess.cpp",H+"/TemplateGenerator.cpp","-L"+libdir,"-llinemod_hip","-Wl,-rpath,"+libdir])
print("GT position", g["gt_position"], "GT rot\n", g["gt_rotation"])
for args in (["1","550","700","80"], ["0","550","700","80"], ["1","550","700","70"], ["0","550","700","65"]):
    r=subprocess.run([exe,d+"/mesh.bin",d+"/bgr.raw",d+"/depth.raw"]+args,capture_output=True,text=True,cwd=d)
    print("ARGS",args); print("\n".join(l for l in r.stdout.splitlines() if not l.startswith("ERROR")) [:3000]); print(r.stderr[-500:])
