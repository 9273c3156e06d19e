"""`from src.dataset.robonet.robonet_dataset import RoboNetDataset, process_batch, get_batch, normalize, denormalize`
(reference src/dataset/robonet/robonet_dataset.py)."""
from robot_aware_control_amd.data import (RoboNetDataset, denormalize, get_batch, normalize,  # noqa: F401
                                          process_batch)
