"""The `ops` namespace after the round-6 split (settings / engine / dispatch / functions): a switch has ONE home whichever
spelling sets it, and a re-exported function cannot be replaced on the namespace (the modules that use it would not see it)."""
import pytest


def test_switches_are_one_object_through_both_spellings(monkeypatch):
    from texttoaudiogrounding_amd import dispatch, ops, settings
    assert set(settings.NAMES) >= {"CONV_MATH", "CONV_WINOGRAD", "WINO_MIN_WORK", "WINO_LAUNCHES", "DIRECT_GRADS", "PROFILE"}
    for name in settings.NAMES:
        assert name not in vars(ops), name                    # no stale copy shadows the forwarding
        assert getattr(ops, name) is getattr(settings, name)
    monkeypatch.setattr(ops, "WINO_MIN_WORK", 1)
    assert settings.WINO_MIN_WORK == 1
    monkeypatch.setattr(ops, "CONV_MATH", "x3")
    assert settings.CONV_MATH == "x3" and not dispatch._wino_shape(32, 64, 64)      # the dispatch rule reads the same object
    monkeypatch.setattr(ops, "CONV_MATH", "fp32")
    assert dispatch._wino_shape(32, 64, 64)
    monkeypatch.undo()
    assert settings.WINO_MIN_WORK == ops.WINO_MIN_WORK == 1 << 20
    n = ops.WINO_LAUNCHES
    ops.WINO_LAUNCHES += 1
    assert settings.WINO_LAUNCHES == n + 1
    settings.WINO_LAUNCHES = n


def test_reexports_are_read_only_and_engine_state_is_forwarded():
    from texttoaudiogrounding_amd import engine, functions, ops
    assert ops.new_seed is engine.new_seed and ops.Cnn8RnnFunction is functions.Cnn8RnnFunction
    with pytest.raises(AttributeError, match="re-exported from texttoaudiogrounding_amd.engine"):
        ops.new_seed = lambda: 0
    with pytest.raises(AttributeError, match="no attribute"):
        ops.no_such_name
    prev = engine._RECORDING
    ops._RECORDING = not prev
    assert engine._RECORDING is (not prev)
    ops._RECORDING = prev
