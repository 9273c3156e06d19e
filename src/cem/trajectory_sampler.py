"""`from src.cem.trajectory_sampler import TrajectorySampler` (reference src/cem/trajectory_sampler.py)."""
from robot_aware_control_amd.trajectory_sampler import TrajectorySampler  # noqa: F401
