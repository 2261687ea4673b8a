"""MI355X-native hot path of vgiwae (VAE / IWAE / MissSVAE / MissSIWAE / DeMissVAE estimation from
incomplete data): hand-written gfx950 HIP kernels in libdmv.so behind a C ABI (include/dmv.h), and
the host-side Python that mirrors the reference's estimator functions and model classes."""
from ._lib import lib, lib_path  # noqa: F401
from .mlp import NetDims, PackedNet, GeneratedDropout, ResidualMLPFunction, mlp_forward, mlp_backward, make_keep_masks  # noqa: F401
from .objectives import (ModelDims, estimate_elbo, estimate_iwelbo, estimate_multiple_elbo,  # noqa: F401
                         estimate_multiple_iwelbo, mvb2_objectives, cvi_objective, mvb_objective)
from .graph import GraphedStep  # noqa: F401
