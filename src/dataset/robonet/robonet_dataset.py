"""`from src.dataset.robonet.robonet_dataset import process_batch, get_batch` (reference robonet_dataset.py:434-467).
The hdf5 dataset classes themselves are the reference's."""
from robot_aware_control_amd.data import get_batch, process_batch  # noqa: F401
