"""Runs a reference YAML config without PyTorch Lightning (reference entry point: train.py:11-113).

    python -m demiss_vae_amd.run --config configs/uci_gas/mis50/iwae_i5_encm_stl.yaml [--data.data_root DIR]
                                 [--trainer.max_epochs N] [--trainer.max_steps N] [--any.dotted.key VALUE ...]

The YAML is read as it is: `class_path` / `init_args` nodes are instantiated recursively, strings are turned into
the enum members the constructors annotate (what jsonargparse does for the reference), `data:` builds the
MissingDataModule, `trainer:` supplies max_epochs / max_steps / gradient clipping.  The loop is the part of
Lightning's fit loop these models use: setup hooks, automatic or manual optimisation, per-step schedulers,
gradient clipping through the model's hook, a validation pass per epoch, epoch means of everything passed to
`self.log`, a final checkpoint (state_dict + hyper-parameters).  On one GPU with residual-MLP networks a whole
optimisation step is recorded in a HIP graph and replayed (`--trainer.hip_graph false` turns that off).  One process per GPU; under torchrun the gradients
are all-reduced (demiss_vae_amd.parallel) and each rank trains on its own fixed part of the training split."""
import argparse
import enum
import importlib
import inspect
import json
import os
import random
import time
import types

import numpy as np
import torch
import yaml

from . import parallel as P


# ------------------------------------------------------------------------------------------ instantiation
def _annotations(cls):
    """parameter name -> annotated type (Enum classes, float, int), collected over the constructors of the whole
    MRO (keyword arguments travel up through **kwargs)"""
    out = {}
    for k in inspect.getmro(cls):
        init = k.__dict__.get("__init__")
        if init is None:
            continue
        try:
            sig = inspect.signature(init)
        except (TypeError, ValueError):
            continue
        for name, p in sig.parameters.items():
            ann = p.annotation
            if isinstance(ann, str):
                ann = getattr(importlib.import_module(k.__module__), ann, None)
            if ann in (float, int) or (inspect.isclass(ann) and issubclass(ann, enum.Enum)):
                out.setdefault(name, ann)
    return out


def _coerce(ann, v):
    """what jsonargparse does with the annotation: enum members by name (or value), numbers out of the strings
    YAML 1.1 leaves behind for literals like 1e-4"""
    if v is None or isinstance(v, ann):
        return v
    if issubclass(ann, enum.Enum):
        try:
            return ann[v]
        except KeyError:
            return ann(v)
    if ann is float and isinstance(v, (str, int)) and not isinstance(v, bool):
        return float(v)
    if ann is int and isinstance(v, str):
        return int(v)
    return v


def instantiate(node):
    """Builds the object a `class_path` node describes; other values pass through."""
    if isinstance(node, dict) and "class_path" in node:
        module, _, name = node["class_path"].rpartition(".")
        cls = getattr(importlib.import_module(module), name)
        return construct(cls, node.get("init_args") or {})
    if isinstance(node, dict):
        return {k: instantiate(v) for k, v in node.items()}
    if isinstance(node, list):
        return [instantiate(v) for v in node]
    return node


def construct(cls, kwargs):
    ann = _annotations(cls)
    built = {}
    for k, v in kwargs.items():
        v = instantiate(v)
        built[k] = _coerce(ann[k], v) if k in ann else v
    return cls(**built)


def _set_dotted(cfg, dotted, value):
    keys = dotted.split(".")
    node = cfg
    for k in keys[:-1]:
        node = node.setdefault(k, {})
    node[keys[-1]] = yaml.safe_load(value)


def load_config(path, overrides=()):
    with open(path) as f:
        cfg = yaml.safe_load(f)
    for k, v in overrides:
        _set_dotted(cfg, k, v)
    return cfg


# ------------------------------------------------------------------------------------------ the loop
class _Means:
    """epoch means of the logged values; tensors are summed on the device, so logging never waits for the GPU"""

    def __init__(self):
        self.sum, self.n = {}, {}

    def add(self, logged, weight=1):
        sums, vals = [], []
        for k, v in logged.items():
            if isinstance(v, torch.Tensor):
                v = v.detach()
                v = v if v.dtype == torch.float32 else v.float()
                v = v if weight == 1 else v * weight
                have = self.sum.get(k)
                if isinstance(have, torch.Tensor) and have.device == v.device and have.shape == v.shape:
                    sums.append(have)
                    vals.append(v)
                else:      # first sight: a copy (a replayed step hands out the same tensors every time)
                    self.sum[k] = v.clone() if have is None else have + v
            else:
                self.sum[k] = self.sum.get(k, 0.0) + float(v) * weight
            self.n[k] = self.n.get(k, 0) + weight
        if sums:
            torch._foreach_add_(sums, vals)        # one launch for all metrics of the step

    def result(self, suffix=None):
        return {k: float(self.sum[k]) / self.n[k] for k in sorted(self.sum) if suffix is None or k.endswith(suffix)}


def seed_everything(seed):
    random.seed(seed)
    np.random.seed(seed)
    torch.manual_seed(seed)


def init_distributed():
    """joins the job torchrun started (one process per GPU, RCCL); a plain `python -m` run is a world of one"""
    import torch.distributed as dist
    if int(os.environ.get("WORLD_SIZE", 1)) > 1 and not dist.is_initialized():
        if torch.cuda.is_available():
            torch.cuda.set_device(int(os.environ.get("LOCAL_RANK", 0)) % torch.cuda.device_count())
        # RCCL on the GPU; DMV_DIST_BACKEND=gloo lets several ranks share one GPU in rehearsals and tests
        dist.init_process_group(os.environ.get("DMV_DIST_BACKEND") or ("nccl" if torch.cuda.is_available() else "gloo"))


_UNIT = {}


def _unit_gradient(loss):
    """the root gradient of a scalar loss, kept per device: backward() would otherwise fill a fresh one every step"""
    if loss.dim() != 0 or loss.dtype != torch.float32:
        return None
    one = _UNIT.get(loss.device)
    if one is None:
        one = _UNIT[loss.device] = torch.ones((), device=loss.device)
    return one


class _DeferredScheduler:
    """stands in for the scheduler while a step is recorded: the recording must not bake a learning rate in, so the
    real scheduler is stepped after every replay instead"""

    def step(self, *a, **k):
        pass


class _StepRunner:
    """One optimisation step on a batch: eagerly, or - once a batch shape has been seen twice - as the replay of a HIP
    graph of the whole step (model step, backward, clipping, optimiser).  The learning rates live in device tensors
    the scheduler writes between replays; the library's noise stays fresh through its device counter
    (demiss_vae_amd.rng.CaptureStream).  A recording is keyed by everything the step's host-side control flow
    depends on: batch rows and the epoch-dependent switches of the model."""

    def __init__(self, model, opt, sched, params, clip_val, clip_alg, use_graph, log=print):
        self.model, self.opt, self.sched, self.params, self.log = model, opt, sched, params, log
        self.clip_val, self.clip_alg = clip_val, clip_alg
        self.use_graph = use_graph
        # data parallelism: the recording ends with the backward pass; gradient exchange, clipping and the optimiser
        # follow eagerly after every replay (no collective inside a recording)
        self.split = use_graph and P.world() > 1
        self.graph_grads = {}
        self.seen, self.graphs, self.static, self.eager_keys = {}, {}, {}, set()
        self.replays = 0
        self.side = torch.cuda.Stream() if use_graph else None
        self.own = hasattr(opt, "sync_lr")          # demiss_vae_amd.optim.Adam keeps its rates on the device itself
        if use_graph and not self.own:
            for g in opt.param_groups:
                g["capturable"] = True
                g["lr"] = torch.tensor(float(g["lr"]), device=params[0].device)
            if sched is not None:          # the scheduler remembers the initial rates as floats: give it tensors' values
                sched.base_lrs = [float(b) for b in sched.base_lrs]

    def _key(self, batch):
        m, h = self.model, self.model.hparams
        key = [tuple(batch[0].shape)]
        if getattr(m, "train_imp_distribution", None) is not None:      # imputations start after a delay
            key.append(m.current_epoch >= h.mvb_delay_imputation)
        if hasattr(m, "_log_due") and m._log_due():      # epochs on which mixture logits are copied to the host
            return None
        if h.get("marginal_eval_train"):
            key.append(m.current_epoch % h.marginal_eval_freq == 0 or m.current_epoch == (m.trainer.max_epochs or 0) - 1)
        return tuple(key)

    def _gradients(self, batch):
        m, opt = self.model, self.opt
        loss = m.training_step(batch, 0)
        opt.zero_grad(set_to_none=True)
        loss.backward(_unit_gradient(loss))

    def _update(self):
        m, opt = self.model, self.opt
        P.allreduce_grads(self.params)
        m.on_before_optimizer_step(opt, 0)
        m.configure_gradient_clipping(opt, 0, gradient_clip_val=self.clip_val, gradient_clip_algorithm=self.clip_alg)
        opt.step()

    def _body(self, batch):
        m = self.model
        if m.automatic_optimization:
            self._gradients(batch)
            self._update()
        else:
            m.training_step(batch, 0)          # the model steps its optimiser itself (two-loss DeMissVAE step)

    def _after(self):
        sched = self.sched
        if sched is None or not (self.model.automatic_optimization or self.model.hparams.get("use_lr_scheduler")):
            return
        if self.own:
            sched.step()
            self.opt.sync_lr()
        elif self.use_graph and hasattr(sched, "_get_closed_form_lr"):
            # the same schedule evaluated in closed form on the host: one small write per group instead of the
            # scheduler's chained tensor arithmetic on the device-resident rates
            sched.last_epoch += 1
            values = sched._get_closed_form_lr()
            for g, lr in zip(self.opt.param_groups, values):
                g["lr"].fill_(float(lr))
            sched._last_lr = [float(v) for v in values]
        else:
            sched.step()

    def _eager(self, batch, stream=None):
        m = self.model
        if stream is not None:
            stream.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(stream):
                self._body(batch)
            torch.cuda.current_stream().wait_stream(stream)
        else:
            self._body(batch)
        if m.automatic_optimization or self.use_graph:
            self._after()

    def _flat_gradients(self):
        """the fused networks keep their parameter gradients in one permanent buffer each (what the gradient exchange
        reduces in place); with automatic optimisation (one backward pass a step) the decoder's exchange starts under
        the encoder's backward pass; the library's optimiser writes updated parameters straight into the networks'
        MFMA fragment images"""
        self.flat_ready = True
        m = self.model
        inner = getattr(m, "model", m)
        if not getattr(inner, "fused", False) or not hasattr(m, "engine"):
            return
        nets = tuple(m.engine()[1:])
        for net in nets:
            net.use_flat_grads(True)
        if hasattr(self.opt, "attach_images"):       # the optimiser step keeps the networks' MFMA images current
            self.opt.attach_images(nets)
        if P.world() > 1 and m.automatic_optimization:
            P.overlap_grad_exchange(nets)

    def __call__(self, batch):
        if not getattr(self, "flat_ready", False):
            self._flat_gradients()
        if not self.use_graph:
            return self._eager(batch)
        from .graph import GraphedStep
        if not self.model.automatic_optimization:
            self.model._schedulers = _DeferredScheduler()
        key = self._key(batch)
        if key is None:
            return self._eager(batch)
        n = self.seen.get(key, 0)
        self.seen[key] = n + 1
        if n < 2 or key in self.eager_keys:         # lazily created state (optimiser, workspaces) must exist first
            return self._eager(batch, self.side if n < 2 else None)
        if key not in self.graphs:
            # a loader that reuses its batch tensors hands over the recording's inputs themselves
            static = tuple(t if getattr(t, "_dmv_persistent", False) else t.clone() for t in batch)
            m = self.model
            nets = tuple(m.engine()[1:]) if getattr(getattr(m, "model", m), "fused", False) else ()
            logged = {}

            def body():
                m._logged = logged
                logged.clear()
                if self.split and m.automatic_optimization:
                    self._gradients(static)
                elif self.split:            # the model's own two-loss step, stopped after its backward passes
                    m._defer_update = True
                    try:
                        m.training_step(static, 0)
                    finally:
                        m._defer_update = False
                else:
                    self._body(static)
            try:
                self.graphs[key] = GraphedStep(body, nets=nets, warmup=0)
                if self.split:      # where the recording leaves the gradients (its own memory, same place every replay)
                    self.graph_grads[key] = [p.grad for p in self.params]
            except Exception as e:                  # a step that synchronises with the host cannot be recorded
                self.log(f"[run] step not recordable ({type(e).__name__}: {e}); steps of this kind run eagerly")
                self.eager_keys.add(key)
                torch.cuda.synchronize()
                return self._eager(batch)
            self.static[key] = (static, logged)
        static, logged = self.static[key]
        for dst, src in zip(static, batch):
            if dst.data_ptr() != src.data_ptr():
                dst.copy_(src)
        self.graphs[key]()
        self.replays += 1
        self.model._logged = dict(logged)
        if self.split:
            for p, g in zip(self.params, self.graph_grads[key]):
                p.grad = g          # (the exchange of the previous step re-pointed .grad at its flat buffer)
            if self.model.automatic_optimization:
                self._update()
            else:
                self.model.finish_manual_update()
        self._after()


class Checkpointer:
    """What the reference's checkpoint callback does (train.py:66-79, vgiwae/overrides/model_checkpoint.py:11-66):
    the best model by validation loss (one file, replaced when beaten; the refit configs' `load_best_model` looks for
    exactly one file besides last.ckpt), the last model, and snapshots at chosen epochs / steps."""

    def __init__(self, dirpath, model, model_config, *, keep_best=False, monitor="loss/val", epochs=(), steps=()):
        self.dirpath, self.model, self.model_config = dirpath, model, model_config
        self.keep_best, self.monitor = keep_best, monitor
        self.epochs, self.steps = set(epochs or ()), set(steps or ())
        self.best, self.best_path = float("inf"), None

    def _save(self, name, **extra):
        path = os.path.join(self.dirpath, name)
        save_checkpoint(self.model, path, self.model_config, extra)
        return path

    def on_step(self, step):
        if step in self.steps:
            self._save(f"custom_ckpt_by_step_{step}.ckpt", global_step=step)

    def on_validation_end(self, epoch, step, row):
        value = row.get(self.monitor)
        if self.keep_best and value is not None and value < self.best:
            if self.best_path is not None and os.path.exists(self.best_path):
                os.remove(self.best_path)
            self.best = value
            self.best_path = self._save(f"epoch={epoch}-step={step}.ckpt", epoch=epoch + 1, global_step=step,
                                        monitor=self.monitor, value=value)
        if epoch in self.epochs:
            self._save(f"custom_ckpt_by_epoch_{epoch}.ckpt", epoch=epoch + 1, global_step=step)

    def on_train_end(self, epochs, step):
        self._save("last.ckpt", epoch=epochs, global_step=step)


class _RunLogger:
    """the slice of Lightning's logger the models touch: the directory of this run (imputation / mixture-weight dumps)"""

    def __init__(self, log_dir):
        self.log_dir = log_dir
        self.experiment = self

    def get_logdir(self):
        os.makedirs(self.log_dir, exist_ok=True)
        return self.log_dir


class _ValRunner:
    """Validation steps replayed from recordings like training steps, one recording per batch shape, for the model
    classes whose validation step is just the bound of the batch (the step functions named in _PLAIN_VALIDATION: a
    subclass that overrides validation_step - grid marginals, FID, imputation updates - runs eagerly as before).  Noise
    stays fresh through the recording's device-resident Philox counter; the logged values are the recording's static
    outputs, summed into the epoch means right after each replay."""

    def __init__(self, model, use_graph, log=print):
        self.model, self.log = model, log
        self.ok = bool(use_graph) and getattr(type(model), "validation_step", None) in _plain_validation_steps()
        self.seen, self.graphs, self.static, self.eager_keys = {}, {}, {}, set()
        self.replays = 0

    def __call__(self, batch, i):
        m = self.model
        if not self.ok:
            return m.validation_step(batch, i)
        key = tuple(batch[0].shape)
        n = self.seen.get(key, 0)
        self.seen[key] = n + 1
        if n < 1 or key in self.eager_keys:
            return m.validation_step(batch, i)
        if key not in self.graphs:
            from .graph import GraphedStep
            static = tuple(t if getattr(t, "_dmv_persistent", False) else t.clone() for t in batch)
            nets = tuple(m.engine()[1:])
            logged = {}

            def body():
                m._logged = logged
                logged.clear()
                with torch.no_grad():
                    m.validation_step(static, 0)
            try:
                self.graphs[key] = GraphedStep(body, nets=nets, warmup=0)
            except Exception as e:
                self.log(f"[run] validation step not recordable ({type(e).__name__}: {e}); runs eagerly")
                self.eager_keys.add(key)
                torch.cuda.synchronize()
                return m.validation_step(batch, i)
            self.static[key] = (static, logged)
        static, logged = self.static[key]
        for dst, src in zip(static, batch):
            if dst.data_ptr() != src.data_ptr():
                dst.copy_(src)
        self.graphs[key]()
        self.replays += 1
        m._logged = dict(logged)


def _plain_validation_steps():
    from vgiwae.models.vae import VAE
    from vgiwae.models.iwae import IWAE
    from vgiwae.models.multiple_vae import MultipleVAE
    from vgiwae.models.multiple_iwae import MultipleIWAE
    return {VAE.validation_step, IWAE.validation_step, MultipleVAE.validation_step, MultipleIWAE.validation_step}


def fit(model, datamodule, trainer_cfg, log=print, checkpointer=None, log_dir=None):
    from . import rng
    dev = torch.device("cuda", int(os.environ.get("LOCAL_RANK", 0)) % max(torch.cuda.device_count(), 1))
    max_epochs = trainer_cfg.get("max_epochs")
    max_steps = trainer_cfg.get("max_steps", -1)
    max_steps = -1 if max_steps is None else max_steps
    clip_val, clip_alg = trainer_cfg.get("gradient_clip_val"), trainer_cfg.get("gradient_clip_algorithm")
    datamodule.setup("fit")
    train_loader = datamodule.train_dataloader()
    trainer = types.SimpleNamespace(max_epochs=max_epochs, max_steps=max_steps, datamodule=datamodule, global_step=0,
                                    num_training_batches=len(train_loader), accumulate_grad_batches=1)
    model.trainer = trainer
    if log_dir is not None and getattr(model, "logger", None) is None:
        model.logger = _RunLogger(log_dir)
    model.set_datamodule(datamodule)
    model.to(dev)
    rng.set_seed(torch.initial_seed() + P.rank())
    model.setup("fit")
    conf = model.configure_optimizers()
    opt = conf["optimizer"] if isinstance(conf, dict) else conf
    sched = conf.get("lr_scheduler", {}).get("scheduler") if isinstance(conf, dict) else None
    model._optimizers, model._schedulers = opt, sched
    params = [p for g in opt.param_groups for p in g["params"]]
    inner = getattr(model, "model", model)
    use_graph = bool(trainer_cfg.get("hip_graph", True)) and getattr(inner, "fused", False) \
        and (P.world() == 1 or model.automatic_optimization or hasattr(model, "finish_manual_update"))
    run_step = _StepRunner(model, opt, sched, params, clip_val, clip_alg, use_graph, log=log)
    run_val = _ValRunner(model, use_graph and getattr(inner, "fused", False), log=log)
    if use_graph and hasattr(train_loader, "reuse_buffers"):
        train_loader.reuse_buffers = True          # each batch is consumed by its step before the next is drawn
    model.on_train_start()
    val_loader = None
    step, epoch, history, t0 = 0, 0, [], time.time()
    done = False
    while not done and (max_epochs in (None, -1) or epoch < max_epochs):
        model.current_epoch = epoch
        model.train()
        model.on_train_epoch_start()
        means = _Means()
        for i, batch in enumerate(train_loader):
            model._logged = {}
            run_step(batch)
            step += 1
            model.global_step = trainer.global_step = step
            model.on_train_batch_end(None, batch, i)
            means.add(model._logged)
            if checkpointer is not None:
                checkpointer.on_step(step)
            if max_steps != -1 and step >= max_steps:
                done = True
                break
        model._schedulers = sched
        model._logged = {}
        model.on_train_epoch_end()
        means.add(model._logged)            # what the hook logs once per epoch
        model.eval()
        model._logged = {}
        vmeans = _Means()
        if val_loader is None or not run_val.ok:
            val_loader = datamodule.val_dataloader()
            if run_val.ok and hasattr(val_loader, "reuse_buffers"):
                val_loader.reuse_buffers = True      # one loader for the whole run: its batch tensors are the recordings' inputs
        with torch.no_grad():
            for i, batch in enumerate(val_loader):
                model._logged = {}
                run_val(batch, i)
                vmeans.add(model._logged)
        model.on_validation_epoch_end()
        row = {"epoch": epoch, "step": step, "time_s": round(time.time() - t0, 2), **means.result(), **vmeans.result()}
        history.append(row)
        if P.rank() == 0:
            log(json.dumps(row))
        if checkpointer is not None:
            checkpointer.on_validation_end(epoch, step, row)
        epoch += 1
    model.on_train_end()
    if checkpointer is not None:
        checkpointer.on_train_end(epoch, step)
    trainer.hip_graph_replays = run_step.replays
    summary = {"steps": step, "hip_graph_replays": run_step.replays, "validation_replays": run_val.replays}
    if P.world() > 1:
        # replicas must hold identical parameters: largest difference between any two ranks
        import torch.distributed as dist
        flat = torch.cat([p.detach().flatten() for p in params])
        hi, lo = P.all_reduce(flat.clone(), dist.ReduceOp.MAX), P.all_reduce(flat.clone(), dist.ReduceOp.MIN)
        summary.update(world=P.world(), replica_spread=float((hi - lo).abs().max()))
    if P.rank() == 0:
        log(json.dumps(summary))
    return history


def test(model, datamodule, log=print, shard="auto"):
    """One pass over the test split through `test_step` (what trainer.test does for these models); returns the means
    of everything logged, weighted by batch size.

    Under torch.distributed the work is split over the ranks (one process per GPU):
      points   batch i goes to rank i mod world; no exchange until one final SUM all-reduce of (sum, count) per metric;
      samples  every rank sees every batch and folds its share of the K importance samples; the (max, sum-exp) states
               are combined per batch with a logsumexp all-reduce of 2 x B floats (models with `can_shard_samples`);
      auto     samples when the test split has fewer batches than ranks (small test sets), else points.
    Device noise of batch i is drawn from counter range i << 40 of one Philox stream shared by all ranks, so for
    single-Gaussian encoders the result does not depend on the number of ranks or on the mode."""
    from . import rng
    datamodule.setup("test")
    model.eval()
    loader = datamodule.test_dataloader()
    r, w = P.rank(), P.world()
    if shard == "auto":
        shard = "samples" if (len(loader) < w and getattr(model, "can_shard_samples", False)) else "points"
    assert shard in ("points", "samples"), shard
    if shard == "samples" and w > 1:
        assert getattr(model, "can_shard_samples", False), "this model's test step cannot split importance samples"
        model.loglik_shard = (r, w)
    seed = torch.initial_seed() & 0xFFFFFFFFFFFFFFFF
    means = _Means()
    device = None
    with torch.no_grad():
        for i, batch in enumerate(loader):
            device = batch[0].device
            if shard == "points" and i % w != r:
                continue
            rng.set_state(seed, i << 40)
            model._logged = {}
            model.test_step(batch, i)
            means.add(model._logged, weight=batch[0].shape[0])
    if hasattr(model, "loglik_shard"):
        del model.loglik_shard
    if shard == "points" and w > 1:
        out = P.allreduce_weighted_means(means.sum, means.n, device=device)
    else:
        out = means.result()
    if P.rank() == 0:
        log(json.dumps(out))
    return out


def _plain(v):
    if isinstance(v, enum.Enum):
        return v.name
    if isinstance(v, (list, tuple)):
        return [_plain(x) for x in v]
    return v


def next_version_dir(exp_dir):
    """<exp_dir>/lightning_logs/version_<n>, n = the first unused one (the layout the refit configs' model_path names)"""
    base = os.path.join(exp_dir, "lightning_logs")
    n = 0
    while os.path.exists(os.path.join(base, f"version_{n}")):
        n += 1
    return os.path.join(base, f"version_{n}")


def run_version_dir(exp_dir):
    """next_version_dir, chosen once per job: rank 0 picks and creates the directory, the other ranks are told its
    name (a rank that looked later would find rank 0's directory in place and move on to the next number)"""
    if P.world() == 1:
        return next_version_dir(exp_dir)
    import torch.distributed as dist
    box = [None]
    if P.rank() == 0:
        box[0] = next_version_dir(exp_dir)
        os.makedirs(box[0], exist_ok=True)
    dist.broadcast_object_list(box, src=0)
    return box[0]


def save_checkpoint(model, path, model_config=None, extra=None):
    """state_dict + the primitive hyper-parameters + the YAML node that rebuilds the model: tensors and plain
    containers only, so `torch.load(weights_only=True)` reads it back."""
    os.makedirs(os.path.dirname(path), exist_ok=True)
    hp = {k: _plain(v) for k, v in model.hparams.items()
          if isinstance(v, (int, float, str, bool, type(None), enum.Enum, list, tuple))}
    torch.save({"state_dict": model.state_dict(), "hyper_parameters": hp, "model_config": model_config, **(extra or {})}, path)


def load_model(path, model_class=None):
    """Rebuilds a model from a checkpoint of save_checkpoint; `model_class` (a name in vgiwae.models) replaces the
    class the checkpoint was trained as, the way the reference's refit script picks the class to load with
    (refit_encoder_using_iwae_and_estimate_loglik.py:100-125)."""
    ckpt = torch.load(path, map_location="cpu", weights_only=True)
    model = instantiate(retarget(ckpt["model_config"], model_class))
    model.load_state_dict(ckpt["state_dict"])
    return model


def retarget(node, model_class=None):
    """the YAML node of a model, rewritten for another class of vgiwae.models: keyword arguments of a diagnostic
    wrapper the model was trained under mean nothing to a class without that wrapper"""
    node = dict(node)
    if model_class is not None:
        from vgiwae.models import _DIAGNOSTIC_KWARGS
        keep = next((set(kw) for suffix, kw in _DIAGNOSTIC_KWARGS.items() if model_class.endswith(suffix)), set())
        drop = set().union(*_DIAGNOSTIC_KWARGS.values()) - keep
        node["class_path"] = "vgiwae.models." + model_class
        node["init_args"] = {k: v for k, v in (node.get("init_args") or {}).items() if k not in drop}
    return node


def resolve_model_path(template, exp_dir, load_best=False):
    """fills `seed_{}` with this run's seed part and `version_{}` with the newest version on disk
    (refit_encoder_using_iwae_and_estimate_loglik.py:103-122)"""
    seed_part = exp_dir.split("seed_")[1].split("/")[0]
    path = template.format(seed_part, "{}")
    if "version_{}" in path:
        root = path.split("version_")[0]
        versions = sorted((int(v.split("version_")[1]) for v in os.listdir(root) if v.startswith("version_")), reverse=True)
        if len(versions) > 1:
            print(f"Multiple versions in {path}")
        path = path.format(versions[0])
    if load_best:
        folder = path.split("last.ckpt")[0]
        best = [m for m in os.listdir(folder) if not m.startswith("last.ckpt")]
        assert len(best) == 1, f"Found multiple models in {folder}"
        path = os.path.join(folder, best[0])
    return path


def _parse_overrides(rest):
    overrides, it = [], iter(rest)
    for tok in it:
        assert tok.startswith("--"), f"unexpected argument {tok}"
        if "=" in tok:
            k, v = tok[2:].split("=", 1)
        else:
            k, v = tok[2:], next(it)
        overrides.append((k, v))
    return overrides


def main(argv=None, log=print):
    ap = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    ap.add_argument("--config", required=True)
    args, rest = ap.parse_known_args(argv)
    cfg = load_config(args.config, _parse_overrides(rest))
    init_distributed()
    seed_everything(cfg["seed_everything"])
    from vgiwae.data import MissingDataModule
    trainer_cfg = dict(cfg.get("trainer") or {})
    exp_dir = "{}/seed_m{}_d{}".format(cfg["experiment_subdir_base"], cfg["seed_everything"], cfg["data"].get("setup_seed"))
    exp_dir = os.path.join(trainer_cfg.get("default_root_dir") or ".", exp_dir)
    log("Experiment directory: " + exp_dir)
    datamodule = construct(MissingDataModule, cfg["data"])
    refit = "refit_model" in cfg
    if refit:
        # the reference's second entry point: refit the encoder of a trained model, then the K-sample test estimate
        from vgiwae.test.refit_encoder_using_iwae_and_estimate_loglik import RefitEncoderUsingIWAE
        model = construct(RefitEncoderUsingIWAE, cfg["refit_model"])
        path = resolve_model_path(cfg["model_path"], exp_dir, cfg.get("load_best_model", False))
        log("Using pre-trained model from: " + path)
        model.set_model(load_model(path, cfg.get("model_class")))
        model_config = retarget(torch.load(path, map_location="cpu", weights_only=True)["model_config"], cfg.get("model_class"))
    else:
        model = instantiate(cfg["model"])
        model_config = cfg["model"]
    seed_everything(cfg["seed_everything"])      # same generator state at the start of fitting whatever was built
    out = run_version_dir(exp_dir)
    ckpt = None
    if P.rank() == 0:
        # snapshots at chosen epochs / steps belong to the checkpoint callback (train.py:66-79): none without it
        keep = bool(cfg.get("add_checkpoint_callback", False))
        ckpt = Checkpointer(os.path.join(out, "checkpoints"), model, model_config, keep_best=keep,
                            epochs=cfg.get("save_custom_epochs") if keep else (),
                            steps=cfg.get("save_custom_steps") if keep else ())
    history = fit(model, datamodule, trainer_cfg, log=log, checkpointer=ckpt, log_dir=out)
    results = {}
    if refit and cfg.get("run_test", True):
        log("Estimating test-log-likelihood.")
        results = test(model, datamodule, log=log, shard=trainer_cfg.get("test_shard", "auto"))
    if P.rank() == 0:
        with open(os.path.join(out, "metrics.jsonl"), "w") as f:
            for row in history:
                f.write(json.dumps(row) + "\n")
            if results:
                f.write(json.dumps(results) + "\n")
    if P.world() > 1:
        torch.distributed.destroy_process_group()
    return history, results


if __name__ == "__main__":
    main()
