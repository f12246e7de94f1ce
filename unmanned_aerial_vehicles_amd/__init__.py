"""MI355X-native Gaussian-Process residual model for the quadrotor GP-MPC workspace.

Host side in Python (the reference's language) over the libgpk C ABI (hand-written gfx950 HIP
kernels).  Public surface mirrors the reference's seams:

    GaussianProcessRegressor, RBF, WhiteKernel, ConstantKernel   (estimator seam, gpr.py / kernels.py)
    SimpleQuadrotorGP, SimpleGPEnhancedMPC                        (model seam, simple_gp.py)
    GaussianProcess                                               (ROS-package GP, package_gp.py)
    GPTrainer, PreTrainedGP                                       (per-output ARD GPs, trainer.py)
    evaluate_gp, ShardedPredictor, sharded_gram
"""
from .kernels import RBF, ConstantKernel, WhiteKernel  # noqa: F401
from .gpr import GaussianProcessRegressor  # noqa: F401
from .simple_gp import SimpleGPEnhancedMPC, SimpleQuadrotorGP  # noqa: F401
from .package_gp import GaussianProcess  # noqa: F401
from .trainer import GPTrainer, PreTrainedGP  # noqa: F401
from .batched import BatchedARDGP  # noqa: F401
from .evaluate import evaluate_gp  # noqa: F401
from .sharded import ShardedPredictor, gram_slab_bounds, shard_bounds, sharded_gram, sharded_predict  # noqa: F401

__all__ = ["GaussianProcessRegressor", "RBF", "WhiteKernel", "ConstantKernel", "SimpleQuadrotorGP",
           "SimpleGPEnhancedMPC", "GaussianProcess", "GPTrainer", "PreTrainedGP", "evaluate_gp",
           "ShardedPredictor", "shard_bounds", "sharded_predict", "sharded_gram", "gram_slab_bounds", "BatchedARDGP"]
