"""Adam / AMSGrad with global gradient-norm clipping in two kernel launches (dmv_opt_adam_step in include/dmv.h).

`Adam` takes what `torch.optim.Adam` takes in the reference's `configure_optimizers` (vgiwae/models/vae.py:349-374:
parameter groups with their own `lr` and `amsgrad`) and applies the same update rule; `state_dict()` has torch's
layout (step, exp_avg, exp_avg_sq, max_exp_avg_sq per parameter).  Differences in mechanics only:
  * clipping can ride along: `clip_on_next_step(max_norm)` makes the next `step()` scale the gradients by torch's
    clip_grad_norm_ coefficient inside the update (the gradients in memory stay unscaled);
  * learning rates and step counts live on the device, so a step recorded in a HIP graph replays correctly;
    `group["lr"]` stays a float for schedulers, `step()` mirrors it to the device when it changed and `sync_lr()` does
    so between replays."""
import ctypes as C

import torch

from . import _lib as L
from . import rng


class Adam(torch.optim.Optimizer):
    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0, amsgrad=False):
        if weight_decay != 0:
            raise NotImplementedError("weight decay is not used by the reference and has no kernel")
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay, amsgrad=amsgrad))
        if len(self.param_groups) > 4:
            raise NotImplementedError("at most 4 parameter groups")
        self._clip = 0.0
        self._built = None
        self.grad_norm = None          # device scalar: norm of the gradients seen by the last clipped step
        self._image_nets = []
        self._images_written = ()

    def attach_images(self, nets):
        """The PackedNets among `nets` all of whose parameters this optimiser updates: step() then also writes every
        new parameter value into their MFMA fragment images, and the next forward pass launches no packing kernel."""
        mine = {id(p) for g in self.param_groups for p in g["params"]}
        self._image_nets = [n for n in nets if all(id(p) in mine for p in n.params())]
        for n in self._image_nets:
            n.image_kept_by_optimizer = True
        self._built = None
        return self

    def clip_on_next_step(self, max_norm: float):
        self._clip = float(max_norm or 0.0)

    # ---- device-side mirrors
    def _build(self):
        ps = [p for g in self.param_groups for p in g["params"]]
        dev = ps[0].device
        for p in ps:
            if not (p.is_cuda and p.dtype == torch.float32 and p.is_contiguous() and p.device == dev):
                raise RuntimeError("demiss_vae_amd.optim.Adam needs contiguous fp32 CUDA parameters on one device")
        if len(ps) > 72:
            raise NotImplementedError("more than 72 parameter tensors")
        group_of = []
        # one device scalar per group, kept for the optimiser's lifetime: a step recorded in a HIP graph holds these
        # addresses, so they are only ever written in place
        self._lr_dev = [torch.tensor(float(g["lr"]), device=dev) for g in self.param_groups]
        self._step_dev = [torch.zeros((), device=dev) for _ in self.param_groups]
        self._lr_written = [float(g["lr"]) for g in self.param_groups]
        for gi, g in enumerate(self.param_groups):
            group_of += [gi] * len(g["params"])
            for p in g["params"]:
                st = self.state[p]
                if "exp_avg" not in st:
                    st["exp_avg"] = torch.zeros_like(p)
                    st["exp_avg_sq"] = torch.zeros_like(p)
                    if g["amsgrad"]:
                        st["max_exp_avg_sq"] = torch.zeros_like(p)
                else:                                   # state loaded from a checkpoint
                    self._step_dev[gi].fill_(float(st["step"]))
                st["step"] = self._step_dev[gi]          # one counter per group, shared by its parameters
        n = len(ps)
        numel = (C.c_int64 * n)(*[p.numel() for p in ps])
        groups = (L.DmvAdamGroup * len(self.param_groups))()
        for gi, g in enumerate(self.param_groups):
            b1, b2 = g["betas"]
            groups[gi] = L.DmvAdamGroup(self._lr_dev[gi].data_ptr(), self._step_dev[gi].data_ptr(), b1, b2, g["eps"])
        ws_floats = L.lib().dmv_opt_ws_floats(n, numel)
        b = dict(ps=ps, n=n, numel=numel, group_of=(C.c_int32 * n)(*group_of), groups=groups,
                 ws=torch.empty(ws_floats, dtype=torch.float32, device=dev), ws_floats=ws_floats,
                 p=L.ptr_array([p.detach() for p in ps]), m=L.ptr_array([self.state[p]["exp_avg"] for p in ps]),
                 v=L.ptr_array([self.state[p]["exp_avg_sq"] for p in ps]),
                 vmax=(L._P * n)(*[self.state[p]["max_exp_avg_sq"].data_ptr() if "max_exp_avg_sq" in self.state[p] else None
                                  for p in ps]))
        # where every parameter lives in its network's fragment image (entries of the others stay empty)
        b["pack_map"] = None
        if self._image_nets:
            entries = (L.DmvPackMap * n)()
            index = {id(p): i for i, p in enumerate(ps)}
            for net in self._image_nets:
                for p, e in zip(net.params(), net.pack_maps()):
                    entries[index[id(p)]] = e
            b["pack_map"] = torch.frombuffer(bytearray(bytes(entries)), dtype=torch.uint8).to(dev)
            b["images"] = [net._packed.data_ptr() for net in self._image_nets]
        if self.grad_norm is None or self.grad_norm.device != dev:
            self.grad_norm = torch.zeros((), device=dev)
        self._built = b
        return b

    def sync_lr(self):
        """write the groups' (host) learning rates to their device mirrors where they changed"""
        if self._built is None:
            return
        for gi, g in enumerate(self.param_groups):
            lr = float(g["lr"])
            if lr != self._lr_written[gi]:
                self._lr_dev[gi].fill_(lr)
                self._lr_written[gi] = lr

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        b = self._built or self._build()
        if self._image_nets:
            for net in self._image_nets:
                net.packed()           # entries the step does not write (padding) must be there; normally a no-op
            if [net._packed.data_ptr() for net in self._image_nets] != b["images"]:
                b = self._build()      # an image moved (parameters went to another device)
        self.sync_lr()
        live = [i for i, p in enumerate(b["ps"]) if p.grad is not None]
        if not live:
            return loss
        grads = [b["ps"][i].grad if b["ps"][i].grad.is_contiguous() else b["ps"][i].grad.contiguous() for i in live]
        pack_map = None
        self._images_written = ()
        if len(live) == b["n"]:
            n, P_, M_, V_, X_, numel, group_of = b["n"], b["p"], b["m"], b["v"], b["vmax"], b["numel"], b["group_of"]
            if b["pack_map"] is not None:
                pack_map, self._images_written = b["pack_map"], tuple(self._image_nets)
        else:       # parameters without a gradient are left alone (torch skips them too): tables of the others
            n = len(live)
            pick = lambda arr, ctype: (ctype * n)(*[arr[i] for i in live])
            P_, M_, V_, X_ = (pick(b[k], L._P) for k in ("p", "m", "v", "vmax"))
            numel, group_of = pick(b["numel"], C.c_int64), pick(b["group_of"], C.c_int32)
        # inside a recording the update kernel also advances the device-resident Philox counter by the step's draws
        bump = rng.advance_in_kernel()
        L.check(L.lib().dmv_opt_adam_step(n, P_, L.ptr_array(grads), M_, V_, X_, numel, group_of, len(self.param_groups),
                                          b["groups"], self._clip, L.ptr(b["ws"]), b["ws_floats"], L.ptr(self.grad_norm),
                                          L.ptr(pack_map), C.c_void_p(bump[0]) if bump else None, bump[1] if bump else 0,
                                          L.stream_ptr()))
        self._clip = 0.0
        return loss
