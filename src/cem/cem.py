"""`from src.cem.cem import CEMPolicy` (reference src/cem/cem.py:14-111)."""
from robot_aware_control_amd.cem import CEMPolicy  # noqa: F401
