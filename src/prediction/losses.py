"""`from src.prediction.losses import ...` (reference src/prediction/losses.py)."""
from robot_aware_control_amd.losses import (Cost, ImgDontcareCost, ImgL2Cost, RobotL2Cost, RobotWorldCost,  # noqa: F401
                                            dontcare_l1_criterion, dontcare_mse_criterion, kl_criterion,
                                            l1_criterion, mse_criterion, robot_mse_criterion, world_mse_criterion)
