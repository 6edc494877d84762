"""mvster_amd -- MI355X-native implementation of the MVSTER cost-volume hot path.

Public surface = the reference's ``models`` package surface (models/__init__.py:2):
``MVS4net``, ``MVS4net_loss``, ``Blend_loss``.
"""
from .loss import Blend_loss, MVS4net_loss
from .net import MVS4net

__all__ = ["MVS4net", "MVS4net_loss", "Blend_loss"]
