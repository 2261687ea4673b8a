"""Load a golden case (tests/golden/*.npz) back into specs, parameters and inputs.

Small cases carry their inputs; full-width cases carry only a seed and are rebuilt with the
recipes in oracle/recipes.py (the same recipes tools/golden_gen/make_golden.py used)."""
import os
from collections import OrderedDict

import numpy as np
import torch

from oracle.recipes import NetSpec, make_params, make_batch, make_noise
from oracle.core import ModelSpec

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load(name):
    return dict(np.load(os.path.join(GOLDEN, name + ".npz"), allow_pickle=False))


def names(prefix):
    return sorted(f[:-4] for f in os.listdir(GOLDEN) if f.startswith(prefix) and f.endswith(".npz"))


def net_spec(arr, dropout=0.0):
    i, o, nb, h = (int(v) for v in arr)
    return NetSpec(i, o, nb, h, dropout=float(dropout))


def stored_params(g, prefix, spec, seed):
    key = f"{prefix}.initial_layer.weight"
    if key in g:
        out = OrderedDict()
        for n in spec.layer_names():
            out[n + ".weight"] = torch.from_numpy(g[f"{prefix}.{n}.weight"])
            out[n + ".bias"] = torch.from_numpy(g[f"{prefix}.{n}.bias"])
        return out
    return make_params(spec, seed)


def model_case(g):
    """(ModelSpec, enc_params, dec_params) of an objective case."""
    seed = int(g["seed"])
    enc = net_spec(g["enc_spec"], g["enc_dropout"] if "enc_dropout" in g else 0.0)
    dec = net_spec(g["dec_spec"], g["dec_dropout"] if "dec_dropout" in g else 0.0)
    C = int(g["C"]) if "C" in g else 1
    q = str(g["q_dist"]) if "q_dist" in g else ("stratified_with_eps" if "C" in g and "kind" not in g else "normal_with_eps")
    if "kind" in g:
        q = "stratified_with_eps" if str(g["kind"]) == "stratified" else "normal_with_eps"
    spec = ModelSpec(D=int(g["D"]), Zd=int(g["Zd"]), enc=enc, dec=dec,
                     gen_dist=str(g["gen_dist"]) if "gen_dist" in g else "normal_with_eps",
                     q_dist=q, C=C,
                     encoder_use_mis_mask=bool(g["mask_concat"]) if "mask_concat" in g else (enc.in_dim == 2 * int(g["D"])))
    return spec, stored_params(g, "enc", enc, seed), stored_params(g, "dec", dec, seed + 1)


def mlp_io(g, spec):
    """(x, upstream gradient, rows) of an MLP case: stored, or rebuilt from the seed - for long batches the drawn rows
    less those the generator left out (rows within a margin of a relu kink; `row_ok` holds the kept ones bit-packed)"""
    seed, rows = int(g["seed"]), int(g["rows"])
    if "x" in g:
        return torch.from_numpy(g["x"]), torch.from_numpy(g["gy"]), rows
    drawn = int(g["rows_drawn"]) if "rows_drawn" in g else rows
    x, gy = make_noise(seed + 1, drawn, spec.in_dim), make_noise(seed + 2, drawn, spec.out_dim)
    if "row_ok" in g:
        ok = torch.from_numpy(np.unpackbits(g["row_ok"])[:drawn].astype(bool))
        x, gy = x[ok].contiguous(), gy[ok].contiguous()
    assert x.shape[0] == rows
    return x, gy, rows


def batch_of(g, binary=None):
    seed = int(g["seed"])
    if "X" in g:
        return torch.from_numpy(g["X"]), torch.from_numpy(g["M"])
    if binary is None:
        binary = bool(g["binary"]) if "binary" in g else False
    return make_batch(seed + 2, int(g["B"]), int(g["D"]), binary=binary)


def eps_of(g, *shape):
    if "eps" in g and g["eps"].size:
        return torch.from_numpy(g["eps"])
    return make_noise(int(g["seed"]) + 3, *shape)


def dropout_masks(seed, spec: NetSpec, rows_shape):
    if spec.dropout <= 0:
        return None
    rng = np.random.Generator(np.random.PCG64(seed))
    return [torch.from_numpy((rng.random(tuple(rows_shape) + (spec.hidden,)) >= spec.dropout))
            for _ in range(spec.nb)]


def imputed_batch(g, binary=None):
    """Xt [B,K,D], Mk [B,K,D] of a DeMissVAE case."""
    if "Xt" in g:
        Xt = torch.from_numpy(g["Xt"])
        M = torch.from_numpy(g["M"])
        if M.dim() == 2:
            M = M[:, None, :].expand_as(Xt).contiguous()
        return Xt, M
    seed, B, K, D = int(g["seed"]), int(g["B"]), int(g["K"]), int(g["D"])
    if binary is None:
        binary = bool(g["binary"]) if "binary" in g else False
    X, M = make_batch(seed + 2, B, D, binary=binary)
    fill, _ = make_batch(seed + 6, B * K, D, binary=binary)
    Mk = M[:, None, :].expand(B, K, D).contiguous()
    Xt = torch.where(Mk, X[:, None, :].expand(B, K, D), fill.reshape(B, K, D)).contiguous()
    return Xt, Mk


def check_tensor(g, key, t, rtol, atol):
    """Compare a 2-d tensor with what the fixture holds under `key`: the full array, or (long batches) its digest of
    row sums, column sums and a strided sub-sample.  Sums are compared to the tolerance of their terms times the
    number of terms' typical growth (sqrt of the count)."""
    t = t.detach().cpu().to(torch.float64)
    if key in g:
        torch.testing.assert_close(t, torch.from_numpy(g[key]).double(), rtol=rtol, atol=atol)
        return "full"
    torch.testing.assert_close(t[::7, ::5], torch.from_numpy(g[f"{key}.sub"]).double(), rtol=rtol, atol=atol)
    scale = float(t.abs().mean())
    for name, dim in (("rowsum", 1), ("colsum", 0)):
        ref = torch.from_numpy(g[f"{key}.{name}"]).double()
        grow = float(t.shape[dim]) ** 0.5
        torch.testing.assert_close(t.sum(dim), ref, rtol=rtol, atol=(atol + rtol * scale) * grow,
                                   msg=lambda m: f"{key}.{name}: {m}")
    return "digest"


# How far the gradient comparisons are from a purely relative tolerance: per test, the number of entries compared, how
# many of them needed the absolute (floor) term, the worst relative error among the entries that carry a tensor's norm
# (|ref| >= 1e-3 max|ref|) and the worst norm-wise error |got - ref|_2 / |ref|_2.  tests/conftest.py writes the table
# to gpurun_out/parity_margins.json at the end of a session; tools/parity_margins.py turns it into a note.
MARGINS = {}
import os as _os
FLOOR_FRACTION_MAX = float(_os.environ.get("DMV_FLOOR_FRACTION_MAX", "0.05"))     # at most this share of a tensor's entries may lean on the floor term


def _note_margin(what, t, ref, rtol):
    import os
    test = os.environ.get("PYTEST_CURRENT_TEST", "?").split(" ")[0]
    err, mag = (t - ref).abs(), ref.abs()
    beyond = int((err > rtol * mag).sum())
    sig = mag >= 1e-3 * float(mag.max()) if mag.numel() else mag > 0
    worst_sig = float((err[sig] / mag[sig]).max()) if bool(sig.any()) and float(mag.max()) > 0 else 0.0
    normwise = float(err.norm() / ref.norm()) if float(ref.norm()) > 0 else 0.0
    m = MARGINS.setdefault(test, dict(tensors=0, entries=0, beyond_rel=0, worst_rel_significant=0.0, worst_normwise=0.0,
                                      worst_floor_fraction=0.0))
    m["tensors"] += 1
    m["entries"] += int(ref.numel())
    m["beyond_rel"] += beyond
    m["worst_rel_significant"] = max(m["worst_rel_significant"], worst_sig)
    m["worst_normwise"] = max(m["worst_normwise"], normwise)
    frac = beyond / max(1, ref.numel())
    m["worst_floor_fraction"] = max(m["worst_floor_fraction"], frac)
    return frac, worst_sig


def check_grads(g, prefix, grads, rtol, atol, floor=2e-5, floor_fraction_max=FLOOR_FRACTION_MAX):
    """Compare a dict name -> gradient tensor with the digests stored under `prefix`.

    Elementwise |got - ref| <= rtol*|ref| + max(atol, floor*max|ref|): entries that are sums of
    cancelling fp32 terms cannot agree to a relative tolerance of their own (tiny) magnitude, so
    the absolute part scales with the tensor's largest entry.  The share of a tensor's entries that needed that
    absolute part is printed and bounded (floor_fraction_max), see MARGINS."""
    checked = 0

    def close(t, ref, what, scale=1.0):
        ref = torch.from_numpy(ref).double()
        a = max(atol, floor * float(ref.abs().max())) * scale
        torch.testing.assert_close(t, ref, rtol=rtol, atol=a, msg=lambda m: f"{what}: {m}")
        frac, worst = _note_margin(what, t, ref, rtol)
        if frac > 0:
            print(f"{what}: {100 * frac:.2f} % of {ref.numel()} entries beyond rel {rtol:g} (floor term used); "
                  f"worst relative error among significant entries {worst:.2e}")
        # (four entries of a short tensor are always allowed: a 32-entry bias has a 3 % grain)
        assert frac * ref.numel() <= max(4.0, floor_fraction_max * ref.numel()), \
            f"{what}: {100 * frac:.1f} % of the entries lean on the absolute floor"

    for n, t in grads.items():
        t = t.detach().cpu().to(torch.float64)
        if f"{prefix}.{n}.full" in g:
            close(t, g[f"{prefix}.{n}.full"], f"{prefix}.{n}")
            checked += 1
        elif f"{prefix}.{n}.sub" in g:
            close(t[::7, ::5], g[f"{prefix}.{n}.sub"], f"{prefix}.{n}.sub")
            close(t.sum(1), g[f"{prefix}.{n}.rowsum"], f"{prefix}.{n}.rowsum")
            close(t.sum(0), g[f"{prefix}.{n}.colsum"], f"{prefix}.{n}.colsum")
            checked += 1
    return checked
