"""Type-check of the repository's OWN reference-side header, include/lmgpu_gtsam_adapter.h, against the reference's headers.

`g++ -std=c++17 -fsyntax-only` on a translation unit that includes the header and USES every class in it (so that the member
templates are instantiated too).  Nothing of the reference is built, linked or run: the compiler only reads its headers.  GTSAM's
sources include two files its CMake run generates (gtsam/config.h from gtsam/config.h.in, gtsam/dllexport.h from
cmake/dllexport.h.in); the test writes minimal ones into ITS temporary directory (the option set the survey's build used,
SURVEY.md section 8c) -- they exist only for this syntax pass.  Skipped where the reference tree is absent (the GPU box)."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference"

CONFIG_H = """#pragma once
#define GTSAM_VERSION_MAJOR 4
#define GTSAM_VERSION_MINOR 3
#define GTSAM_VERSION_PATCH 0
#define GTSAM_VERSION_NUMERIC 40300
#define GTSAM_VERSION_STRING "4.3a0"
#define GTSAM_SOURCE_TREE_DATASET_DIR ""
#define GTSAM_INSTALLED_DATASET_DIR ""
#define GTSAM_POSE3_EXPMAP
#define GTSAM_ROT3_EXPMAP
#define GTSAM_DT_MERGING
#define GTSAM_HYBRID_TIMING 0
#define GTSAM_ALLOCATOR_STL
#define GTSAM_SUPPORT_NESTED_DISSECTION
#define GTSAM_TANGENT_PREINTEGRATION
#define GTSAM_ENABLE_BOOST_SERIALIZATION 0
#define GTSAM_USE_BOOST_FEATURES 0
"""
DLLEXPORT_H = """#pragma once
#define GTSAM_EXPORT
#define GTSAM_EXTERN_EXPORT extern
"""

# a user of the header: every class constructed, every public member called, the member templates instantiated
TU = r"""
#include "lmgpu_gtsam_adapter.h"
#include <gtsam/inference/Symbol.h>
using namespace gtsam;

double useBatch(const NonlinearFactorGraph& graph, const Values& initial) {
  LevenbergMarquardtParams lp;
  GpuLevenbergMarquardtOptimizer lm(graph, initial, lp, 0, GpuLevenbergMarquardtOptimizer::WholeIterate);
  GaussianFactorGraph::shared_ptr linear = lm.iterate();          // NonlinearOptimizer.h:136: the linearized graph
  double e = linear->error(VectorValues::Zero(linear->optimize()));  // tests/testNonlinearOptimizer.cpp:282 reads it
  GaussianFactorGraph::shared_ptr lin2 = lm.linearize();
  VectorValues d = lm.solve(*lin2, lp);
  e += d.norm() + lm.lambda() + lm.error() + lm.lastLinearGraph()->size();
  const Values& r1 = lm.optimize();
  GpuLevenbergMarquardtOptimizer pw(graph, initial, lp, 0, GpuLevenbergMarquardtOptimizer::Piecewise);
  e += pw.optimize().size() + r1.size();

  GpuGaussNewtonOptimizer gn(graph, initial);
  e += gn.iterate()->size() + gn.optimize().size();
  GpuDoglegOptimizer dl(graph, initial);
  e += dl.iterate()->size() + dl.getDelta() + dl.optimize().size();
  return e;
}

double useIncremental(const NonlinearFactorGraph& newFactors, const Values& newTheta) {
  ISAM2Params ip;
  ip.relinearizeThreshold = 0.01;
  ip.optimizationParams = ISAM2DoglegParams();
  GpuISAM2 isam(ip, 0);
  lmgpu_isam2_result r = isam.update(newFactors, newTheta);
  ISAM2UpdateParams up;
  up.removeFactorIndices = FactorIndices{0};
  r = isam.update(newFactors, newTheta, up);
  r = isam.update(newFactors, newTheta, FactorIndices(), FastMap<Key, int>(), FastList<Key>(), FastList<Key>(), true);
  double s = r.cliques + isam.error() + isam.errors().first + isam.unusedKeys().size();
  s += isam.calculateEstimate().size() + isam.calculateBestEstimate().size() + isam.getLinearizationPoint().size();
  s += isam.calculateEstimate<Pose2>(Symbol('x', 0)).x() + isam.calculateEstimate<Pose3>(Symbol('x', 1)).x();
  s += isam.calculateEstimate<Point3>(Symbol('l', 0)).x() + isam.calculateEstimate<Point2>(Symbol('l', 1)).x();
  s += isam.marginalCovariance(Symbol('x', 0)).trace();
  return s;
}
"""


@pytest.mark.skipif(not os.path.isdir(os.path.join(REF, "gtsam", "nonlinear")), reason="reference headers not present (GPU box)")
@pytest.mark.skipif(shutil.which("g++") is None, reason="no g++")
def test_gtsam_adapter_header_type_checks_against_the_reference_headers(tmp_path):
    (tmp_path / "gtsam").mkdir()
    (tmp_path / "gtsam" / "config.h").write_text(CONFIG_H)
    (tmp_path / "gtsam" / "dllexport.h").write_text(DLLEXPORT_H)
    (tmp_path / "tu.cpp").write_text(TU)
    cmd = ["g++", "-std=c++17", "-fsyntax-only", "-Wall", "-Wextra", "-Wno-unused-parameter", "-Wno-deprecated-copy",
           f"-I{tmp_path}", f"-I{ROOT}/include", f"-I{REF}", f"-I{REF}/gtsam/3rdparty/Eigen",
           f"-I{REF}/gtsam/3rdparty/CCOLAMD/Include", f"-I{REF}/gtsam/3rdparty/SuiteSparse_config", str(tmp_path / "tu.cpp")]
    r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=600)
    ours = [ln for ln in r.stdout.splitlines() if "lmgpu_" in ln and ("error" in ln or "warning" in ln)]
    assert r.returncode == 0, r.stdout[-6000:]
    assert not ours, "\n".join(ours)
