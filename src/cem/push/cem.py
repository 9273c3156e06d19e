"""`from src.cem.push.cem import CEMPolicy` (reference src/cem/push/cem.py:15-104): learned-physics branch."""
from robot_aware_control_amd.cem import SimCEMPolicy as CEMPolicy  # noqa: F401
