from .axialnet import *  # noqa: F401,F403   (reference lib/models/__init__.py:2)
from . import axialnet   # noqa: F401
