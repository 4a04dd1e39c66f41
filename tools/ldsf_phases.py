"""Development aid: where the time of an upper-level LDS front goes.  Builds gtsam_personal_amd/liblmgpu_dbg.so (the library with
-DLDSF_STAMPS) if it is missing, runs LM iterations of a SLAM workload without graph replay and prints the per-phase totals of the
first workgroup of every lds_front_kernel launch of at most eight fronts.   python tools/ldsf_phases.py [city10000|sphere2500] [colamd|metis]"""
import ctypes as ct
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ["LMGPU_GRAPH"] = "0"
DBG = os.path.join(ROOT, "gtsam_personal_amd", "liblmgpu_dbg.so")
if not os.path.exists(DBG):
    src = os.path.join(ROOT, "gtsam_personal_amd", "csrc")
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-DLDSF_STAMPS", "-Wno-unused-value",
                           "-o", DBG, os.path.join(src, "lmgpu.hip"), os.path.join(src, "plan.cpp"), "-L/opt/rocm/lib", "-lrccl"])
import numpy as np
from gtsam_personal_amd import _lib
_lib.LIB_PATH = DBG
import bench
from gtsam_personal_amd import LevenbergMarquardtOptimizer, LevenbergMarquardtParams

wl = sys.argv[1] if len(sys.argv) > 1 else "city10000"  # or sphere2500, victoria_park
on = sys.argv[2] if len(sys.argv) > 2 else "colamd"
graph, initial = bench.slam_workload(wl)
fx = np.load(os.path.join(ROOT, "tests", "golden", "slam_orderings.npz"))
keys = np.array(sorted(graph.keys()), dtype=np.uint64)
ordering = [int(k) for k in keys[fx[f"{wl}_{on}"]]]
opt = LevenbergMarquardtOptimizer(graph, initial, ordering, LevenbergMarquardtParams(), device=0)
opt.save_values()
st = opt.copy_state()
lib = _lib.load()
dbg = lib.lmgpu_debug_ldsf
dbg.restype = ct.c_int
dbg.argtypes = [ct.POINTER(ct.c_ulonglong), ct.c_int]
out = (ct.c_ulonglong * 16)()
for it in range(3):
    opt.restore_values(st)
    dbg(out, 1)
    opt.iterate()
    dbg(out, 0)
names = ["descriptors + clear + own factors", "extend-add of the children", "damping", "partial Cholesky", "emit [R S d] + update matrix",
         "(merged launches) wait for the children", "(merged launches) publish"]
n = out[15]
print(f"{wl}/{on}: {n} sampled workgroups (launches of <= 8 fronts; the top quarter of merged launches) in one LM iteration (inner iterations: {opt.timings()['inner_iterations']})")
for i, nm in enumerate(names):
    print(f"  {nm:36s} {out[i] * 0.01:9.1f} us total   {out[i] * 0.01 / max(n, 1):6.2f} us per launch")
