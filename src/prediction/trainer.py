"""`from src.prediction.trainer import PredictionTrainer` (reference src/prediction/trainer.py)."""
from robot_aware_control_amd.trainer import PredictionTrainer  # noqa: F401
