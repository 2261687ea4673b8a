"""Gradient variance and signal-to-noise over an epoch's optimisation steps (reference:
vgiwae/shared/vae_grad_var.py:5-45; the `*GradVar` model classes).  The reference keeps every step's flattened
gradient and takes std / mean at the end of the epoch; here running sums (count, sum, sum of squares, in double
precision, in buffers that live for the whole run so a recorded step keeps writing to them) give the same statistics
in constant memory."""
import torch


def _flat_grads(net):
    return torch.cat([p.grad.detach().flatten() for p in net.parameters()]).double()


def gradvar_on_train_epoch_start(self):
    if "_gradvar" not in self.__dict__:
        self.__dict__["_gradvar"] = None           # created at the first step, when the gradients' device is known
    elif self._gradvar is not None:
        for t in self._gradvar.values():
            t.zero_()


def gradvar_on_before_optimizer_step(self, optimizer, optimizer_idx):
    g = {"generator": _flat_grads(self.generator_network), "encoder": _flat_grads(self.var_latent_network)}
    if self.__dict__.get("_gradvar") is None:
        acc = {"n": torch.zeros((), dtype=torch.float64, device=g["generator"].device)}
        for k, v in g.items():
            acc[k + "_sum"], acc[k + "_sq"] = torch.zeros_like(v), torch.zeros_like(v)
        self.__dict__["_gradvar"] = acc
    acc = self._gradvar
    acc["n"] += 1
    for k, v in g.items():
        acc[k + "_sum"] += v
        acc[k + "_sq"] += v * v


def gradvar_training_epoch_end(self):
    acc = self.__dict__.get("_gradvar")
    if acc is None:
        return
    n = acc["n"]
    for k, name in (("generator", "generator"), ("encoder", "encoder")):
        mean = acc[k + "_sum"] / n
        var = torch.clamp(acc[k + "_sq"] - n * mean * mean, min=0) / (n - 1)        # unbiased, per coordinate
        self.log(f"{name}_grad_var/train", var.sum().float(), on_epoch=True, prog_bar=False, logger=True)
        snr = torch.abs(mean / var.sqrt())
        # coordinates whose gradient is always zero (dead units) give 0 / 0 and are left out, as in the reference
        self.log(f"{name}_grad_snr/train", snr[torch.isfinite(snr)].sum().float(), on_epoch=True, prog_bar=False, logger=True)
