"""Importance-weighted DeMissVAE (reference: vgiwae/models/mvbiwae.py:14-789): K persistent imputations per
datapoint and I importance samples per imputation.  The networks, the sampling and the five per-sample terms run in
the HIP pipeline (demiss_vae_amd.objectives.demiss_terms); the dozen objective variants differ only in how they pool
those [nz, I, B, K] terms, which is done here with a few small tensor operations whose cotangents the pipeline's
backward turns into network gradients."""
import math
from enum import Enum, auto
from typing import Optional

import torch

import demiss_vae_amd as dmv
from demiss_vae_amd.objectives import demiss_terms
from demiss_vae_amd.pl_compat import STEP_OUTPUT
from vgiwae.models.iwae import IWAE
from vgiwae.shared.mvb2_manual_optimisation import finish_manual_update, mvb2_manual_optimisation_step


class MVB_IWAE_OBJ(Enum):
    mvb_indep = auto()                                    # MVB objective, every imputation particle on its own
    mvb_all = auto()                                      # placeholder in the reference: particles pooled (no valid bound)
    cvi_indep = auto()                                    # CVI objective, every particle on its own
    cvi_all = auto()                                      # CVI objective, particles pooled
    mvb2_encindep_decall = auto()                         # encoder: mvb_indep, decoder: cvi_all
    mvb2_encindep_decindep = auto()                       # encoder: mvb_indep, decoder: cvi_indep
    mvb2_encall_decall = auto()                           # encoder: mvb_all, decoder: cvi_all
    mvb2_encallweightedsmis_decall = auto()               # encoder gradient re-weighted by SMIS weights, decoder: cvi_all
    mvb2_encallweighteddmmis_decall = auto()              # encoder gradient re-weighted by DMMIS weights, decoder: cvi_all
    mvb2_all_encweighteddmmis_decweighteddmmis = auto()   # both networks on the DMMIS re-weighted objective
    mvb2_all_encweightedsmis_decweightedsmis = auto()     # placeholder in the reference (not implemented there)
    mvb2_encallweighteddmmis_decallresampl = auto()       # as ..dmmis_decall with an importance-resampled decoder objective


class IS_WEIGHTS(Enum):
    smis = auto()     # standard importance weights: q of the particle's own imputation
    dmmis = auto()    # deterministic-mixture weights: uniform mixture over the K imputations' posteriors


_SINGLE = {"mvb_indep": "indep", "mvb_all": "all", "cvi_indep": "indep", "cvi_all": "all"}
# objective -> (pooling of the decoder objective, pooling of the encoder objective)
_TWO_PLAIN = {"mvb2_encindep_decall": ("all", "indep"), "mvb2_encindep_decindep": ("indep", "indep"),
              "mvb2_encall_decall": ("all", "all")}
# objective -> importance weights that re-weight the encoder gradient
_TWO_WEIGHTED = {"mvb2_encallweightedsmis_decall": "smis", "mvb2_encallweighteddmmis_decall": "dmmis",
                 "mvb2_encallweighteddmmis_decallresampl": "dmmis", "mvb2_all_encweighteddmmis_decweighteddmmis": "dmmis"}
# metric the encoder objective is logged under (mvbiwae.py:674-699); the all-weighted variants log no cvi/<stage>
_ENC_METRIC = {**{k: "mvb" for k in _TWO_PLAIN}, **{k: "mvb_gradientweightedobj" for k in _TWO_WEIGHTED}}
_NO_CVI_METRIC = ("mvb2_all_encweighteddmmis_decweighteddmmis", "mvb2_encallweighteddmmis_decallresampl")


def _pool(lw):
    """[nz, I, B, K] -> [nz, I*K, B]: the K imputations' samples of a datapoint side by side"""
    nz, I, B, K = lw.shape
    return lw.permute(0, 1, 3, 2).reshape(nz, I * K, B)


def _bound(lw, how):
    """log-mean-exp of the log-weights over I*K pooled samples ("all") or over I per imputation, then averaged over
    K ("indep"); outer samples are averaged (mvbiwae.py:97-122) -> [B]"""
    if how == "all":
        pooled = _pool(lw)
        return (torch.logsumexp(pooled, dim=1) - math.log(pooled.shape[1])).mean(0)
    return (torch.logsumexp(lw, dim=1) - math.log(lw.shape[1])).mean((0, 2))


class _Terms:
    """names for the rows of the terms tensor"""

    def __init__(self, t):
        self.gen_marg, self.gen_full, self.prior = t[0], t[1], t[2]
        self.lq = {IS_WEIGHTS.smis: t[3], IS_WEIGHTS.dmmis: t[4]}


class MVBIWAE(IWAE):
    def __init__(self, *args, mvb_sample_num_imputations: int = None, mvb_update_val_imps: bool = True,
                 mvb_eval_val_loss: bool = False, enc_isweights: IS_WEIGHTS = IS_WEIGHTS["dmmis"],
                 dec_isweights: IS_WEIGHTS = IS_WEIGHTS["dmmis"],
                 mvb_objective: MVB_IWAE_OBJ = MVB_IWAE_OBJ["mvb2_encindep_decall"], mvb2_optim_gradclip_val: float = 0,
                 mvb2_optim_gradclip_alg: str = "norm", **kwargs):
        super().__init__(*args, **kwargs)
        self.save_hyperparameters()
        name = self.hparams.mvb_objective.name
        if name.startswith("mvb2"):
            self.automatic_optimization = False          # two losses, one manual step
        if name in _SINGLE or name.startswith("mvb2_all"):
            assert self.hparams.enc_isweights is self.hparams.dec_isweights, \
                "Encoder and decoder should use the same importance weights."

    mvb2_manual_optimisation_step = mvb2_manual_optimisation_step
    finish_manual_update = finish_manual_update

    def on_train_epoch_end(self) -> None:
        super().on_train_epoch_end()
        self.resample_imputations(stage="train")

    def on_validation_epoch_end(self) -> None:
        super().on_validation_epoch_end()
        self.resample_imputations(stage="val")

    # ---- per-sample terms of an imputed batch through the fused pipeline
    def _terms(self, Xt, M, *, split, eps=None):
        if not self.fused:
            raise NotImplementedError("MVBIWAE runs on residual-MLP networks (no shipped config pairs it with the "
                                      "convolutional trunks)")
        h = self.hparams
        dims, enc, dec = self.engine()
        B, K = Xt.shape[:2]
        ek, dk = self.dropout_masks(B * K, h.num_latent_samples * h.num_importance_samples * B * K)
        return demiss_terms(dims, enc, dec, Xt, M, num_latent_samples=h.num_latent_samples,
                            num_importance_samples=h.num_importance_samples,
                            q_live=not (h.var_latent_STL or h.var_latent_DREG), split=split, eps=eps, enc_keep=ek, dec_keep=dk)

    def _single_objective(self, Xt, M, complete, eps=None):
        """one bound for both networks: CVI scores the observed entries, MVB all entries of the imputed datapoint
        (mvbiwae.py:124-161, 614-650)"""
        h = self.hparams
        t = _Terms(self._terms(Xt, M, split=False, eps=eps)[0])
        gen = t.gen_full if complete else t.gen_marg
        return _bound(gen + t.prior - t.lq[h.enc_isweights], _SINGLE[h.mvb_objective.name])

    def estimate_cvi_objective(self, Xt, M, eps=None):
        return self._single_objective(Xt, M, complete=False, eps=eps)

    def estimate_mvb_objective(self, Xt, M, eps=None):
        return self._single_objective(Xt, M, complete=True, eps=eps)

    def estimate_mvb2_objectives(self, Xt, M, eps=None, resample_idx=None):
        """(generator objective [B], encoder objective [B]) (mvbiwae.py:163-612).  The first carries gradient to the
        decoder parameters only, the second to the encoder parameters only."""
        h = self.hparams
        name = h.mvb_objective.name
        dec_terms, enc_terms = self._terms(Xt, M, split=True, eps=eps)
        td, te = _Terms(dec_terms), _Terms(enc_terms)
        marg_d = td.gen_marg + td.prior - td.lq[h.dec_isweights]          # log-weights of the observed entries
        full_e = te.gen_full + te.prior - te.lq[h.enc_isweights]          # log-weights of the imputed datapoints
        if name in _TWO_PLAIN:
            dec_how, enc_how = _TWO_PLAIN[name]
            return _bound(marg_d, dec_how), _bound(full_e, enc_how)
        if name not in _TWO_WEIGHTED:
            raise NotImplementedError(name)
        kind = IS_WEIGHTS[_TWO_WEIGHTED[name]]

        def reweighted(t, full):
            """value: bound of the observed-entry weights a; gradient: that of the complete-data weights under
            softmax(a) (mvbiwae.py:361-380)"""
            a = t.gen_marg + t.prior - t.lq[kind]
            return a, _bound(a.detach() - full.detach() + full, "all")

        a, enc_obj = reweighted(te, full_e)
        if h.var_latent_DREG:
            # score-function part of the re-weighted gradient, reparameterised (mvbiwae.py:382-392)
            pooled = _pool(a)
            w = torch.softmax(pooled, dim=1).detach()
            v = ((1 - w) * w * pooled).sum(1).mean(0)
            enc_obj = enc_obj - v + v.detach()
        if name == "mvb2_all_encweighteddmmis_decweighteddmmis":
            full_d = td.gen_full + td.prior - td.lq[h.enc_isweights]
            return reweighted(td, full_d)[1], enc_obj
        if name.endswith("resampl"):
            # importance-resampled stand-in for the pooled bound: the mean log-weight of I*K samples drawn in
            # proportion to the weights (mvbiwae.py:301-311)
            nz, I, B, K = marg_d.shape
            assert nz == 1, "the reference's gather only broadcasts for one outer sample"
            flat = marg_d.permute(0, 2, 1, 3).reshape(nz, B, I * K)
            if resample_idx is None:
                probs = torch.softmax(flat.detach(), -1).reshape(nz * B, I * K)
                resample_idx = torch.multinomial(probs, I * K, replacement=True).t().reshape(I * K, nz, B)
            picked = torch.gather(flat.expand(resample_idx.shape[0], nz, B, I * K), -1, resample_idx.unsqueeze(-1)).squeeze(-1)
            return picked.mean((0, 1)), enc_obj
        return _bound(marg_d, "all"), enc_obj

    # ---- steps
    def _objective_step(self, Xt, M, stage, optimise):
        name = self.hparams.mvb_objective.name
        if name.startswith("mvb2"):
            gen_obj, enc_obj = self.estimate_mvb2_objectives(Xt, M)
            gen_obj, enc_obj = gen_obj.mean(), enc_obj.mean()
            if optimise:
                self.mvb2_manual_optimisation_step(-enc_obj, -gen_obj)
            self.log(f"{_ENC_METRIC[name]}/{stage}", enc_obj, on_step=True, on_epoch=True, prog_bar=False, logger=True)
            if name not in _NO_CVI_METRIC:
                self.log(f"cvi/{stage}", gen_obj, on_step=True, on_epoch=True, prog_bar=False, logger=True)
            return -gen_obj
        if name in ("mvb_indep", "mvb_all"):
            obj = self.estimate_mvb_objective(Xt, M).mean()
            self.log(f"mvb/{stage}", obj, on_step=True, on_epoch=True, prog_bar=False, logger=True)
            return -obj
        if name in ("cvi_indep", "cvi_all"):
            obj = self.estimate_cvi_objective(Xt, M).mean()
            self.log(f"cvi/{stage}", obj, on_step=True, on_epoch=True, prog_bar=False, logger=True)
            return -obj
        raise ValueError(f"Unknown MVB objective: {self.hparams.mvb_objective}")

    def training_step(self, batch, batch_idx) -> STEP_OUTPUT:
        dist = self.train_imp_distribution
        imputed = self.update_imputations(dist.get_imputed_datapoints(batch), stage="train")
        n_sub = self.hparams.mvb_sample_num_imputations
        if n_sub is not None:
            imputed = dist.get_imputed_datapoints(batch, num_imputations=n_sub)
        loss = self._objective_step(imputed[0], imputed[1], "train", optimise=True)
        self.log("loss/train", loss, on_step=True, on_epoch=True, prog_bar=True, logger=True)
        return loss

    def validation_step(self, batch, batch_idx) -> Optional[STEP_OUTPUT]:
        imputed = None
        if self.hparams.mvb_update_val_imps:
            imputed = self.update_imputations(self.val_imp_distribution.get_imputed_datapoints(batch), stage="val")
        if self.hparams.mvb_eval_val_loss:
            if imputed is None:
                imputed = self.val_imp_distribution.get_imputed_datapoints(batch)
            with torch.no_grad():
                loss = self._objective_step(imputed[0], imputed[1], "val", optimise=False)
            self.log("loss/val", loss, on_step=True, on_epoch=True, prog_bar=True, logger=True)
