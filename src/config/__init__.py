"""`from src.config import argparser, create_parser` (reference src/config/__init__.py)."""
from pkgutil import extend_path

__path__ = extend_path(__path__, __name__)  # fall through to the reference's modules of this package (src/__init__.py)
from robot_aware_control_amd.config import argparser, create_parser, str2bool, str2intlist, str2list  # noqa: F401
