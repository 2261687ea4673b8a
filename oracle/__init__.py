"""CPU restatement of the vgiwae training / evaluation hot path.

THIS PACKAGE IS TEST INFRASTRUCTURE.  It is the checker for the HIP path, never the
thing measured or shipped: only `tests/`, `__graft_entry__.smoke()` and the
`cpu_baseline` leg of `bench.py` may import it.  Nothing under `demiss_vae_amd/`
or `vgiwae/` imports it.

It restates, in stock PyTorch CPU ops (`F.linear`, `softplus`, `logsumexp`, autograd)
and without `torch.distributions`, the arithmetic of these reference functions
(paths relative to the reference checkout):

  vgiwae/shared/neural_nets.py:49-133      ResidualBlock / ResidualFCNetwork.forward
  vgiwae/models/vae.py:80-91,101-260       predict_var_latent_params, create_distribution
  vgiwae/models/vae.py:393-473             ELBO with analytic / Monte-Carlo KL
  vgiwae/shared/iwae.py:9-132              SMIS / DMIS log-weights, latent_mixture_logprob
  vgiwae/models/iwae.py:70-154             IWAE bound (plain, STL, DREG)
  vgiwae/models/multiple_vae.py:109-195    MissSVAE stratified ELBO
  vgiwae/models/multiple_iwae.py:124-237   MissSIWAE bound (tight / looser)
  vgiwae/models/mvbvae.py:67-172           DeMissVAE mvb2 / cvi objectives
  vgiwae/shared/vae_imputation.py:26-82    pseudo-Gibbs iteration
  vgiwae/shared/vae_imputation.py:177-201,390-696   LAIR iteration + sampler
  vgiwae/utils/imputation_distribution.py:139-187   imputation buffer gather / scatter
  vgiwae/test/refit_encoder_using_iwae_and_estimate_loglik.py:189-222   K-sample log-lik

Pinning: every function here is checked in `tests/test_oracle_golden.py` against the
golden vectors in `tests/golden/`, which were produced by importing and running the
reference's own code (`tools/golden_gen/make_golden.py`, run in the build container
where the reference checkout is mounted).  Noise (normal, uniform, categorical indices,
dropout masks) is always an explicit input, so the comparison is deterministic.
"""
