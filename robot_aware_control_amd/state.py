"""State containers exchanged between the MBRL loop and the planner
(same field names as reference src/utils/state.py:4-19)."""
from dataclasses import dataclass
from typing import Any


@dataclass
class State:
    img: Any = None        # (H, W, 3) uint8 frame, or a (n, 3, H, W) float tensor inside the sampler
    state: Any = None      # robot end-effector state
    sim_state: Any = None
    mask: Any = None       # robot mask
    sim: Any = None
    qpos: Any = None       # joint positions for the analytical robot model


@dataclass
class DemoGoalState:
    imgs: Any = None       # list of goal frames (H, W, 3) uint8
    states: Any = None
    sim_states: Any = None
    masks: Any = None      # list of goal robot masks, bool (1, H, W)
    qposes: Any = None
