"""torch.autograd bindings over the C ABI of libtag_hip.so -- the namespace callers import (``from texttoaudiogrounding_amd import
ops``).  The code lives in four modules:

* ``settings``  -- every switch of the HIP path (arithmetic, dispatch thresholds, fusion toggles, stream options)
* ``engine``    -- tensor checks, scratch, launch timing, direct-gradient bookkeeping, the autograd base class, the side stream
* ``dispatch``  -- one functional wrapper per kernel family and the rules that choose between kernel forms
* ``functions`` -- the torch.autograd.Function nodes (Cnn8Rnn, Crnn, heads, cross-encoder)

PyTorch is plumbing here: device memory (torch.empty), the current HIP stream and autograd bookkeeping.  All arithmetic of the hot
path runs in the hand-written gfx950 kernels; there is no eager fallback -- a CPU tensor or a missing library raises.

``ops.NAME`` for a switch reads AND assigns ``settings.NAME`` (one object for every consumer; the parity tests flip switches through
this spelling); every other name is a re-export of the module-level object of the module that defines it (read-only here: a
test that replaces a function patches the module that looks it up).
"""
import sys
import types

from . import dispatch, engine, functions, lib, settings
from .lib import call, ptr, query  # noqa: F401

_FORWARDED = {name: settings for name in settings.NAMES}
_FORWARDED["_RECORDING"] = engine              # (rebound per call by TagFunction.apply: a copy here would go stale)

_REEXPORTED = {}
for _mod in (engine, dispatch, functions):
    for _name, _obj in vars(_mod).items():
        if not _name.startswith("__") and _name not in _FORWARDED and not isinstance(_obj, types.ModuleType):
            globals()[_name] = _obj
            _REEXPORTED.setdefault(_name, _mod.__name__)
_env_int = settings._env_int
del _mod, _name, _obj


class _OpsModule(types.ModuleType):
    """Module type of this namespace: the forwarded names live in ONE place (settings / engine), whichever spelling sets them."""

    def __getattr__(self, name):                       # only reached for names that are not module globals
        owner = _FORWARDED.get(name)
        if owner is None:
            raise AttributeError(f"module {self.__name__!r} has no attribute {name!r}")
        return getattr(owner, name)

    def __setattr__(self, name, value):
        owner = _FORWARDED.get(name)
        if owner is not None:
            setattr(owner, name, value)
        elif name in _REEXPORTED:
            # a re-export is a copy of the binding: the modules that USE the object would never see the replacement
            raise AttributeError(f"ops.{name} is re-exported from {_REEXPORTED[name]}; replace it in the module that looks it up "
                                 f"(e.g. functions.{name} for the autograd nodes), not on the ops namespace")
        else:
            super().__setattr__(name, value)

    def __dir__(self):
        return sorted(set(super().__dir__()) | set(_FORWARDED))


sys.modules[__name__].__class__ = _OpsModule
