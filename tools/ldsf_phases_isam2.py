"""Development aid: the phases of the LAST quarter of the cliques of every merged elimination launch of VisualISAM2Example (its root clique),
from the library built with -DLDSF_STAMPS (tools/ldsf_phases.py builds it).   python tools/ldsf_phases_isam2.py"""
import ctypes as ct
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
DBG = os.path.join(ROOT, "gtsam_personal_amd", "liblmgpu_dbg.so")
from gtsam_personal_amd import _lib
_lib.LIB_PATH = DBG
import oracle_harness as oh
from gtsam_personal_amd import ISAM2, ISAM2Params
from gtsam_personal_amd.incremental_workloads import visual_steps

lib = _lib.load()
dbg = lib.lmgpu_debug_ldsf
dbg.restype = ct.c_int
dbg.argtypes = [ct.POINTER(ct.c_ulonglong), ct.c_int]
out = (ct.c_ulonglong * 16)()
isam = ISAM2(ISAM2Params(relinearizeThreshold=0.01, relinearizeSkip=1), ccolamd=lambda *a: oh.ccolamd_csc(*a), device=0)
steps = visual_steps()
for i, (g, v) in enumerate(steps):
    if i == len(steps) - 4:
        dbg(out, 1)
    isam.update(g, v)
dbg(out, 0)
names = ["descriptors + clear + own factors", "extend-add of the children", "damping", "partial Cholesky (rest: trailing updates)", "emit [R S d] + update matrix", "  (from the last group to the barrier that ends the Cholesky)", "-", "  eight-pivot groups: diagonal block (wave 0)", "  eight-pivot groups: panel solve", "  eight-pivot groups: rank-8 update inside the panel", "  eight-pivot groups: rank-16 update below the panel"]
n = out[15]
print(f"{n} sampled workgroups over the last 4 updates")
for i, nm in enumerate(names):
    print(f"  {nm:36s} {out[i] * 0.01:9.1f} us total   {out[i] * 0.01 / max(n, 1):6.2f} us per workgroup")
