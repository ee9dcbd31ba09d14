"""CPU: ``from nomad_amd import nomad`` is the lazy counterpart of the reference's module-level singleton
(/root/reference/src/nomad_audio/__init__.py:1-2), not the implementation submodule."""
import types

import pytest


def test_package_attribute_is_the_singleton_not_the_module():
    import nomad_amd
    from nomad_amd import nomad
    assert not isinstance(nomad, types.ModuleType)
    assert nomad is nomad_amd.nomad
    assert "predict" in dir(nomad) and "forward" in dir(nomad) and "get_embeddings" in dir(nomad)
    assert type(nomad)._instance is None            # nothing is constructed (no model load) by importing or dir()
    # the implementation module stays importable the way the reference's is (sys.modules), class included
    from nomad_amd.nomad import Nomad
    assert Nomad is nomad_amd.Nomad


def test_first_use_constructs_one_nomad(monkeypatch):
    import nomad_amd
    built = []

    class Fake:
        def __init__(self):
            built.append(self)

        def predict(self, *a, **k):
            return ("predict", a, k)

    monkeypatch.setattr(nomad_amd, "Nomad", Fake)
    monkeypatch.setattr(type(nomad_amd.nomad), "_instance", None)
    assert nomad_amd.nomad.predict("dir", nmr="a", deg="b") == ("predict", ("dir",), {"nmr": "a", "deg": "b"})
    assert nomad_amd.nomad.predict("csv")[0] == "predict"
    assert len(built) == 1
    with pytest.raises(AttributeError):
        nomad_amd.nomad.no_such_method
