"""nomad_amd - MI355X-native NOMAD (Non-Matching Audio Distance) scoring engine.

``from nomad_amd import nomad`` gives the module-level singleton the reference exposes as
``from nomad_audio import nomad`` (/root/reference/src/nomad_audio/__init__.py:1-2); it is built on
first use rather than at import time (the reference loads the model - and downloads weights - on
import).
"""
from .nomad import Nomad, TripletModel, LossNetLayers, NomadLoss  # noqa: F401

__all__ = ["Nomad", "TripletModel", "LossNetLayers", "NomadLoss", "nomad"]
_singleton = None


def __getattr__(name):
    global _singleton
    if name == "nomad":
        if _singleton is None:
            _singleton = Nomad()
        return _singleton
    raise AttributeError(name)
