"""torch.hub entry point mirroring the reference's hubconf.py:7."""
dependencies = ["torch", "transformers"]

import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import conette_amd  # noqa: E402,F401
from conette_amd import conette  # noqa: E402

__all__ = ["conette"]
