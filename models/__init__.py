"""Drop-in shim: the reference's drivers do ``from models import *`` (train_mvs4.py:10,
test_mvs4.py:10) and expect MVS4net, MVS4net_loss, Blend_loss (models/__init__.py:2)."""
from mvster_amd import Blend_loss, MVS4net, MVS4net_loss  # noqa: F401
