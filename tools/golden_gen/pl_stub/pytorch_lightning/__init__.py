"""Minimal stand-in for the parts of pytorch_lightning 1.5.8 that the reference's
model classes touch at import and construction time.

Used ONLY by tools/golden_gen/make_golden.py in the build container, so that the
reference's own `vgiwae.models.*` code can be imported and executed to produce the
golden vectors under tests/golden/.  It contains harness plumbing only (a module
base class with an `hparams` attribute dictionary); there is no arithmetic here.
"""
import inspect
import types

import torch.nn as nn


class _AttrDict(dict):
    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError as e:
            raise AttributeError(k) from e

    def __setattr__(self, k, v):
        self[k] = v


class LightningModule(nn.Module):
    CHECKPOINT_HYPER_PARAMS_KEY = "hyper_parameters"
    CHECKPOINT_HYPER_PARAMS_NAME = "hparams_name"
    CHECKPOINT_HYPER_PARAMS_TYPE = "hparams_type"

    def __init__(self, *args, **kwargs):
        super().__init__()
        self.__dict__["_hp"] = _AttrDict()
        self.automatic_optimization = True
        self.current_epoch = 0
        self.trainer = None

    @property
    def hparams(self):
        return self.__dict__["_hp"]

    @property
    def device(self):
        for p in self.parameters():
            return p.device
        import torch
        return torch.device("cpu")

    def save_hyperparameters(self):
        # Collect the calling __init__'s arguments, as Lightning does.
        frame = inspect.currentframe().f_back
        loc = frame.f_locals
        code = frame.f_code
        names = code.co_varnames[: code.co_argcount + code.co_kwonlyargcount]
        for n in names:
            if n == "self":
                continue
            self.hparams[n] = loc[n]
        if code.co_flags & inspect.CO_VARKEYWORDS:
            kwname = code.co_varnames[code.co_argcount + code.co_kwonlyargcount
                                      + (1 if code.co_flags & inspect.CO_VARARGS else 0)]
            for k, v in loc.get(kwname, {}).items():
                self.hparams.setdefault(k, v)

    def log(self, *args, **kwargs):
        pass

    def setup(self, stage=None):
        pass

    def on_train_end(self):
        pass

    def on_train_start(self):
        pass

    def on_train_epoch_start(self):
        pass

    def on_train_epoch_end(self):
        pass

    def on_validation_epoch_end(self):
        pass


class LightningDataModule:
    """hparams attribute dictionary filled from the constructor's arguments, as for LightningModule (plumbing only)"""

    def __init__(self, *args, **kwargs):
        self.__dict__["_hp"] = _AttrDict()

    @property
    def hparams(self):
        return self.__dict__["_hp"]

    save_hyperparameters = LightningModule.save_hyperparameters


class Trainer:
    pass


class Callback:
    pass
