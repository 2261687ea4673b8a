"""What the model classes need from PyTorch Lightning, for machines where it is not installed.

If `pytorch_lightning` imports, its LightningModule is used unchanged.  Otherwise `LightningModule` here
is an nn.Module with the handful of attributes and hooks the vgiwae classes touch (hparams,
save_hyperparameters, log, device, current_epoch, automatic_optimization, optimizers / lr_schedulers /
manual_backward / clip_gradients for the manual-optimisation step); demiss_vae_amd.runner drives it.
"""
import inspect
from typing import Any

import torch
import torch.nn as nn

_UNIT = {}     # per device: the scalar 1 used as the root gradient of backward()

try:  # pragma: no cover - depends on the environment
    import pytorch_lightning as _pl
    LightningModule = _pl.LightningModule
    LightningDataModule = _pl.LightningDataModule
    HAVE_LIGHTNING = True
except Exception:  # ModuleNotFoundError here; any import-time failure means "not usable"
    HAVE_LIGHTNING = False

    class _HParams(dict):
        def __getattr__(self, k):
            try:
                return self[k]
            except KeyError as e:
                raise AttributeError(k) from e

        def __setattr__(self, k, v):
            self[k] = v

    class LightningModule(nn.Module):
        CHECKPOINT_HYPER_PARAMS_KEY = "hyper_parameters"

        def __init__(self, *args, **kwargs):
            super().__init__()
            object.__setattr__(self, "_hparams_store", _HParams())
            self.automatic_optimization = True
            self.current_epoch = 0
            self.global_step = 0
            self.trainer = None
            self._logged = {}
            self._optimizers = None
            self._schedulers = None

        # ---- hyper-parameters
        @property
        def hparams(self):
            return self._hparams_store

        def save_hyperparameters(self):
            """Record the arguments of the __init__ that called this (one class level at a time)."""
            frame = inspect.currentframe().f_back
            code, loc = frame.f_code, frame.f_locals
            n_named = code.co_argcount + code.co_kwonlyargcount
            for name in code.co_varnames[:n_named]:
                if name != "self":
                    self.hparams[name] = loc[name]
            pos = n_named
            if code.co_flags & inspect.CO_VARARGS:
                pos += 1
            if code.co_flags & inspect.CO_VARKEYWORDS:
                for k, v in loc.get(code.co_varnames[pos], {}).items():
                    self.hparams.setdefault(k, v)

        @classmethod
        def load_from_checkpoint(cls, checkpoint_path, **kwargs):
            """Rebuilds a model from a checkpoint written by demiss_vae_amd.run (tensors + the YAML node of the model),
            as this class - the call the reference's refit script makes on Lightning checkpoints
            (refit_encoder_using_iwae_and_estimate_loglik.py:125)."""
            from demiss_vae_amd.run import load_model
            return load_model(checkpoint_path, cls.__name__)

        # ---- plumbing
        @property
        def device(self):
            for p in self.parameters():
                return p.device
            return torch.device("cpu")

        def log(self, name, value, **kwargs):
            self._logged[name] = value.detach() if isinstance(value, torch.Tensor) else value

        def setup(self, stage=None):
            pass

        def on_train_start(self):
            pass

        def on_train_end(self):
            pass

        def on_train_epoch_start(self):
            pass

        def on_train_batch_end(self, outputs, batch, batch_idx):
            pass

        def on_before_optimizer_step(self, optimizer, optimizer_idx):
            pass

        def on_train_epoch_end(self):
            pass

        def on_validation_epoch_end(self):
            pass

        # ---- manual optimisation support
        def optimizers(self, use_pl_optimizer=True):
            return self._optimizers

        def lr_schedulers(self):
            return self._schedulers

        def manual_backward(self, loss, *args, **kwargs):
            if not args and "gradient" not in kwargs and loss.dim() == 0 and loss.dtype == torch.float32:
                # the root gradient of a scalar loss, kept per device instead of filled anew every step
                one = _UNIT.get(loss.device)
                if one is None:
                    one = _UNIT[loss.device] = torch.ones((), device=loss.device)
                kwargs["gradient"] = one
            loss.backward(*args, **kwargs)

        def clip_gradients(self, optimizer, gradient_clip_val=None, gradient_clip_algorithm=None):
            if not gradient_clip_val:
                return
            by_norm = (gradient_clip_algorithm or "norm") == "norm"
            if by_norm and hasattr(optimizer, "clip_on_next_step"):
                optimizer.clip_on_next_step(gradient_clip_val)      # folded into the optimiser's own kernels
                return
            params = [p for g in optimizer.param_groups for p in g["params"] if p.grad is not None]
            if by_norm:
                torch.nn.utils.clip_grad_norm_(params, gradient_clip_val)
            else:
                torch.nn.utils.clip_grad_value_(params, gradient_clip_val)

        def configure_gradient_clipping(self, optimizer, optimizer_idx, gradient_clip_val=None,
                                        gradient_clip_algorithm=None):
            self.clip_gradients(optimizer, gradient_clip_val=gradient_clip_val,
                                gradient_clip_algorithm=gradient_clip_algorithm)

    class LightningDataModule:
        def __init__(self, *args, **kwargs):
            self._hparams_store = _HParams()

        @property
        def hparams(self):
            return self._hparams_store

        save_hyperparameters = LightningModule.save_hyperparameters

STEP_OUTPUT = Any
