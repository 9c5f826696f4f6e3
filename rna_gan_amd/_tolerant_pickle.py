"""A ``pickle_module`` for ``torch.load`` that opens checkpoints written by the REFERENCE on a machine where the
reference's own classes cannot be imported.

torchgan's ``Trainer.save_model`` pickles the live plugin objects under ``loss_objects`` / ``metric_objects`` (SURVEY 5;
call sites src/histopathology_gan.py:298-314, src/gan_utils.py:286-297): instances of ``torchgan.losses.*``,
``wgan_loss.Wasserstein*LossVAE`` (each holding a whole ``betaVAE.betaVAE`` module) and, through ``nn.Sequential``
arguments, anything the user's scripts defined.  A plain ``torch.load`` raises ``ModuleNotFoundError`` while it resolves the
first such class -- before the dictionary with the state_dicts exists.

Here every global that cannot be imported resolves to an inert PLACEHOLDER class instead (one per ``module.name``; it accepts
any constructor arguments and any pickled state, keeps both for inspection, and raises when called or used as a module).
Importable globals (torch, collections, numpy, ... and the reference's own modules when they ARE installed) resolve as usual.
The tensors inside a placeholder's state are still materialised by torch's persistent-id machinery, so nothing in the stream
is skipped; ``Trainer.load_model`` then takes ``epoch``, the logs and the model / optimizer state_dicts and keeps its own
live plugin objects.
"""
from __future__ import annotations

import pickle as _pickle
from pickle import *  # noqa: F401,F403  (torch.load looks up load / Unpickler / ... on the pickle_module it is given)

__all__ = list(getattr(_pickle, "__all__", [])) + ["MissingGlobal", "missing_globals", "Unpickler", "load", "loads"]

_PLACEHOLDERS = {}


class MissingGlobal:
    """Base of the placeholder classes.  ``_rg_missing`` = (module, qualified name) the checkpoint referred to."""
    _rg_missing = ("?", "?")

    def __init__(self, *args, **kwargs):
        self._rg_args, self._rg_kwargs = args, kwargs

    def __setstate__(self, state):
        # default object state (a dict, or (dict, slots) pairs) -- kept without interpretation
        if isinstance(state, dict):
            self.__dict__.update(state)
        elif isinstance(state, tuple) and len(state) == 2 and isinstance(state[0], (dict, type(None))):
            for part in state:
                if isinstance(part, dict):
                    self.__dict__.update(part)
        else:
            self.__dict__["_rg_state"] = state

    def __call__(self, *a, **k):
        raise RuntimeError("{}.{} was not importable when this checkpoint was loaded: the object is a placeholder".format(
            *type(self)._rg_missing))

    def __repr__(self):
        return "<placeholder for {}.{}>".format(*type(self)._rg_missing)


def _placeholder(module, name):
    key = (module, name)
    cls = _PLACEHOLDERS.get(key)
    if cls is None:
        cls = type(name.rsplit(".", 1)[-1], (MissingGlobal,), {"_rg_missing": key, "__module__": module})
        _PLACEHOLDERS[key] = cls
    return cls


def missing_globals():
    """(module, name) of every global that has been replaced by a placeholder so far (diagnostics / tests)."""
    return sorted(_PLACEHOLDERS)


def is_placeholder(obj):
    """True for an instance (or class) that stands in for an unimportable global."""
    return isinstance(obj, MissingGlobal) or (isinstance(obj, type) and issubclass(obj, MissingGlobal))


class Unpickler(_pickle.Unpickler):
    def find_class(self, module, name):
        try:
            return super().find_class(module, name)
        except (ImportError, AttributeError):
            # ModuleNotFoundError: torchgan / wgan_loss / betaVAE / dcgan absent; AttributeError: the module exists under
            # that name but is a different one (e.g. another project's ``dcgan``)
            if module.split(".", 1)[0] in ("torch", "builtins", "collections", "numpy", "copyreg", "_codecs", "rna_gan_amd"):
                # a genuinely broken stream must not be papered over -- and neither must THIS package's own classes: a
                # renamed / removed rna_gan_amd global means the checkpoint predates a refactor, which the caller has to see
                raise
            return _placeholder(module, name)


def load(file, **kwargs):
    return Unpickler(file, **kwargs).load()


def loads(data, **kwargs):
    import io
    return Unpickler(io.BytesIO(data), **kwargs).load()
