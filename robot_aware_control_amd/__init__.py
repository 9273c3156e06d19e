"""MI355X-native hot path of robot_aware_control: the conv-SVG video-prediction
model (train step) and the CEM batched-rollout planner, on hand-written gfx950
HIP kernels behind the C ABI of include/rac_hip.h (librac_hip.so)."""
from ._lib import EXPORTS, LIB_PATH, RacError, load  # noqa: F401

__all__ = ["EXPORTS", "LIB_PATH", "RacError", "load"]
