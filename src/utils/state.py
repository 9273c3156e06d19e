"""`from src.utils.state import State, DemoGoalState` (reference src/utils/state.py)."""
from robot_aware_control_amd.state import DemoGoalState, State  # noqa: F401
