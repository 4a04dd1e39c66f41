"""gtsam_personal_amd — MI355X-native Levenberg-Marquardt inner loop (linearize + multifrontal Cholesky)
behind the reference's optimizer interface.  See DESIGN.md / include/lmgpu.h.

Only the hot path lives here: csrc/ (HIP kernels + C ABI), the host-side mirror of the reference's
graph / optimizer interface (graph.py, optimizer.py) and the data formats either side of it
(datasets.py, synthetic.py).  Importing the package does not load the GPU library; constructing a
LevenbergMarquardtOptimizer does, and fails loudly if liblmgpu.so has not been built."""
from .graph import (CAL3_S2, CAM_BUNDLER, POINT2, POINT3, POSE2, POSE3, C, L, NonlinearFactorGraph, Ordering, P, Values, X, noiseModel, symbol)  # noqa: F401
from .optimizer import (DoglegOptimizer, DoglegParams, GaussNewtonOptimizer, GaussNewtonParams, LevenbergMarquardtOptimizer,  # noqa: F401
                        LevenbergMarquardtParams, JointMarginal, Marginals)
from .isam2 import ISAM2, ISAM2DoglegParams, ISAM2GaussNewtonParams, ISAM2Params, ISAM2Result  # noqa: F401
