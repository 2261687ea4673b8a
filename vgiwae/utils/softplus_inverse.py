"""vgiwae/utils/softplus_inverse.py:3-7"""
from demiss_vae_amd.imputation import softplus_inverse as torch_softplus_inverse  # noqa: F401
