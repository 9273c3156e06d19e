"""`from src.prediction.models.dynamics import SVGConvModel` (reference dynamics.py:457-644)."""
from robot_aware_control_amd.model import SVGConvModel  # noqa: F401
