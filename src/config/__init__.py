"""`from src.config import argparser, create_parser` (reference src/config/__init__.py)."""
from robot_aware_control_amd.config import argparser, create_parser, str2bool, str2intlist, str2list  # noqa: F401
