import torch

from demiss_vae_amd import parallel as P


_MINUS_ONE = {}


def _descent_direction(t, maximise):
    """root gradient of the backward pass: None (plain loss) or a cached -1 for an objective to be maximised (the
    gradient of its negation, without negating it first)"""
    if not maximise:
        return {}
    m = _MINUS_ONE.get(t.device)
    if m is None:
        m = _MINUS_ONE[t.device] = -torch.ones((), device=t.device)
    return {"gradient": m}


def mvb2_manual_optimisation_step(self, enc_loss, gen_loss, maximise=False):
    """One optimiser step driven by two losses: the generator loss updates the decoder parameters, the encoder
    loss the encoder parameters; then (under data parallelism) average the gradients over ranks, clip, step and
    (per-step) schedule
    (reference: vgiwae/shared/mvb2_manual_optimisation.py:1-22).

    maximise: the two arguments are objectives to ascend (scalars); same gradients as passing their negations.

    With `self._defer_update` set the step stops after the two backward passes and the caller completes it with
    `finish_manual_update` (the runner does so under data parallelism, where the backward passes replay from a
    recording and the gradient exchange must stay outside it)."""
    opt = self.optimizers(use_pl_optimizer=True)
    opt.zero_grad()
    self.manual_backward(gen_loss, retain_graph=True, inputs=list(self.generator_network.parameters()),
                         **_descent_direction(gen_loss, maximise))
    self.manual_backward(enc_loss, inputs=list(self.var_latent_network.parameters()), **_descent_direction(enc_loss, maximise))
    if getattr(self, "_defer_update", False):
        return
    finish_manual_update(self)


def finish_manual_update(self):
    opt = self.optimizers(use_pl_optimizer=True)
    P.allreduce_grads([p for g in opt.param_groups for p in g["params"]])
    self.configure_gradient_clipping(opt, -1, gradient_clip_val=self.hparams.mvb2_optim_gradclip_val,
                                     gradient_clip_algorithm=self.hparams.mvb2_optim_gradclip_alg)
    self.on_before_optimizer_step(opt, 0)       # Lightning calls the hook from inside the wrapped optimizer.step()
    opt.step()
    if self.hparams.use_lr_scheduler:
        self.lr_schedulers().step()
