"""Worker of the sharded log-likelihood tests: one process per rank (plain `python` = a world of one, or torchrun with
several ranks sharing the GPU over gloo).  Loads a checkpoint of the runner, wraps it in RefitEncoderUsingIWAE and
runs `demiss_vae_amd.run.test` with test points and with importance samples split over the ranks; then the
K-sharded estimator on the golden cases with the reference's recorded noise.  Rank 0 prints one JSON line."""
import json
import os
import sys

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path[:0] = [os.path.dirname(HERE), HERE]


def main(ckpt, data_root, K):
    from demiss_vae_amd import run, parallel as P, imputation as imp
    from vgiwae.data import MissingDataModule
    from vgiwae.test.refit_encoder_using_iwae_and_estimate_loglik import RefitEncoderUsingIWAE, RefitEncoderClass
    import golden_util as gu
    import gpu_util as G

    run.init_distributed()
    run.seed_everything(20220118)
    out = {"world": P.world()}
    dm = run.construct(MissingDataModule, dict(setup_seed=20220118, dataset="toy_mog2_large", use_test_instead_train=True,
                                               batch_size=128, test_batch_size=50, missingness="uniform", total_miss_train=0.5,
                                               total_miss_val=0.5, total_miss_test=0.0, pre_imputation="oracle",
                                               pre_impute_val=False, data_root=data_root))
    model = RefitEncoderUsingIWAE(RefitEncoderClass.standard, num_test_importance_samples=K)
    model.set_model(run.load_model(ckpt, "IWAE"))
    model.set_datamodule(dm)
    model.cuda()
    for mode in ("points", "samples", "auto"):
        run.seed_everything(20220118)
        out[mode] = run.test(model, dm, log=lambda *_: None, shard=mode)["est_loglik/test"]

    # the reference's own numbers: golden cases with recorded noise, importance samples split over the ranks
    for name in ("loglik_standard_K64", "loglik_stratified_K64", "loglik_standard_K4096"):
        g = gu.load(name)
        spec, enc_p, dec_p = gu.model_case(g)
        dims = G.model_dims(spec)
        enc, dec = G.packed_net(spec.enc, enc_p, requires_grad=False), G.packed_net(spec.dec, dec_p, requires_grad=False)
        X, M = torch.from_numpy(g["X"]).cuda(), torch.from_numpy(g["M"]).cuda()
        B, Kg = int(g["B"]), int(g["K"])
        if str(g["kind"]) == "standard":
            eps, comp = gu.eps_of(g, 1, Kg, B, spec.Zd)[0].cuda(), None
        else:
            eps, comp = torch.from_numpy(g["eps"])[0].cuda(), torch.from_numpy(g["comp_idx"])[0].cuda()
        val, _ = imp.estimate_loglik(dims, enc, dec, X, M, num_importance_samples=Kg, chunk=1000, eps=eps, comp_idx=comp,
                                     k_shard=(P.rank(), P.world()))
        out[name] = [val.item(), float(g["est_loglik"])]
    if P.rank() == 0:
        print(json.dumps(out), flush=True)
    if P.world() > 1:
        torch.distributed.barrier()
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2], int(sys.argv[3]))
