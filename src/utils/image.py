"""`from src.utils.image import zero_robot_region` (reference src/utils/image.py)."""
from robot_aware_control_amd.image import zero_robot_region  # noqa: F401
