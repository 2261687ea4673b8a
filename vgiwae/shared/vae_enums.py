"""Names the YAML configs use for distributions (reference: vgiwae/shared/vae_enums.py:1-31).

The member names and their numeric order are part of the config / checkpoint surface, so they match the
reference exactly; the table below is built from the family structure rather than written out."""
from enum import Enum

EPSILON = 1e-5

_ORDER = (
    ["normal", "bern", "cont_bern", "cont_bern_prob", "cont_bern_orig", "cont_bern_orig_prob", "normal_with_eps"]
    + [f"stratified_mixture{c}_normal_with_eps" for c in (5, 15, 25)]
    + [f"stratified_mixture{c}_normal" for c in (5, 15, 25)]
    + ["bern_m1_p1", "stratified_mixture1_normal_with_eps", "stratified_mixture1_normal"]
    + [f"reparametrised_mixture{c}_normal_with_eps" for c in (5, 15, 25)]
    + [f"reparametrised_mixture{c}_normal" for c in (5, 15, 25)]
)
DISTRIBUTION = Enum("DISTRIBUTION", _ORDER, module=__name__)
PRIOR_DISTRIBUTION = Enum("PRIOR_DISTRIBUTION", ["std_normal"], module=__name__)


def mixture_components(distribution) -> int:
    """Number of mixture components a stratified / reparametrised member names (0 for the others)."""
    name = distribution.name
    for fam in ("stratified_mixture", "reparametrised_mixture"):
        if name.startswith(fam):
            return int(name[len(fam):].split("_")[0])
    return 0


def engine_name(distribution) -> str:
    """Name used by demiss_vae_amd.ModelDims for this member."""
    name = distribution.name
    if name.startswith("stratified_mixture"):
        return "stratified_with_eps" if name.endswith("with_eps") else "stratified"
    if name.startswith("reparametrised_mixture"):
        return "reparam_with_eps" if name.endswith("with_eps") else "reparam"
    return name
