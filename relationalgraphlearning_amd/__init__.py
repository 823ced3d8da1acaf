"""relationalgraphlearning_amd -- MI355X-native RGL relational-graph forward pass and
model-predictive rollout (the hot path of ChanganVR/RelationalGraphLearning), behind the
reference's own Policy / nn.Module surface.  Device code: hand-written HIP for gfx950 in
csrc/, exposed through the C ABI of include/rgl_hip.h (librgl_hip.so)."""
from . import _native
from .actions import ActionXY, ActionRot
from .nets import mlp, RGL, ValueEstimator, StatePredictor, LinearStatePredictor, ValueNetwork, invalidate_packed_weights
from .policy import Policy, ModelPredictiveRL, GCN, register
from .state import FullState, ObservableState, JointState, tensor_to_joint_state
from .rollout import TreeSearch, GcnSearch, ShardedRollout, rotate, shard_bounds
from .vector_explorer import VectorExplorer, ReplayMemory
from .trainer import MPRLTrainer, VNRLTrainer, register_trainers

__all__ = ["ActionXY", "ActionRot", "mlp", "RGL", "ValueEstimator", "StatePredictor", "LinearStatePredictor",
           "ValueNetwork", "invalidate_packed_weights", "Policy", "ModelPredictiveRL", "GCN", "register", "TreeSearch", "GcnSearch",
           "ShardedRollout", "rotate", "shard_bounds", "FullState", "ObservableState", "JointState",
           "tensor_to_joint_state", "VectorExplorer", "ReplayMemory", "MPRLTrainer", "VNRLTrainer", "register_trainers"]
