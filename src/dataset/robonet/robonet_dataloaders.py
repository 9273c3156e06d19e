"""`from src.dataset.robonet.robonet_dataloaders import create_loaders` (reference robonet_dataloaders.py:21-80)."""
from robot_aware_control_amd.data import create_loaders  # noqa: F401
