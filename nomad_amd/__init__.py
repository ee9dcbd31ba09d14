"""nomad_amd - MI355X-native NOMAD (Non-Matching Audio Distance) scoring engine.

``from nomad_amd import nomad`` gives the module-level singleton the reference exposes as
``from nomad_audio import nomad`` (/root/reference/src/nomad_audio/__init__.py:1-2: ``from .nomad import Nomad``
then ``nomad = Nomad()``, which rebinds the package attribute ``nomad`` from the submodule to the instance).
Here the instance is built on first use rather than at import time (the reference loads the model - and
downloads weights - on import).  As in the reference, ``from nomad_amd.nomad import Nomad`` still reaches the
implementation module through ``sys.modules``.
"""
from .nomad import Nomad, TripletModel, LossNetLayers, NomadLoss  # noqa: F401

__all__ = ["Nomad", "TripletModel", "LossNetLayers", "NomadLoss", "nomad"]


class _LazyNomad:
    """Stand-in for the reference's ``nomad = Nomad()``: every attribute access goes to ONE ``Nomad()`` that is
    constructed the first time it is needed (``nomad.predict(...)``, ``nomad.forward(...)``, ...)."""

    __slots__ = ()
    _instance = None

    @classmethod
    def _get(cls):
        if cls._instance is None:
            cls._instance = Nomad()
        return cls._instance

    def __getattr__(self, name):
        return getattr(self._get(), name)

    def __setattr__(self, name, value):
        setattr(self._get(), name, value)

    def __dir__(self):
        return sorted(set(dir(Nomad)))

    def __repr__(self):
        return "<nomad_amd.nomad: lazy Nomad() singleton%s>" % ("" if type(self)._instance is None else " (built)")


# the import above bound the submodule as the attribute `nomad`; rebind it to the singleton, like the reference does
nomad = _LazyNomad()
