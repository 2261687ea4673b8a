"""Host side of the fused residual-MLP kernels (dmv_mlp_* in include/dmv.h).

`PackedNet` owns the fragment-ordered copy of a network's weights and re-packs it when the
parameters may have changed: after any optimiser step anywhere in the process (a global post-step hook;
torch's fused optimisers update parameters without touching their version counters) and whenever a
parameter's version counter or address moved (load_state_dict, manual in-place edits).  `mlp_forward` / `mlp_backward` are thin wrappers over the C ABI working on padded
row-major buffers; `ResidualMLPFunction` exposes one network call to autograd.
"""
import ctypes as C
import os
import weakref
from dataclasses import dataclass
from typing import List, Optional, Sequence

import torch
from torch.optim.optimizer import register_optimizer_step_post_hook

from . import _lib as L


_optimizer_steps = 0


def _count_optimizer_step(opt=None, *_args, **_kwargs):
    global _optimizer_steps
    _optimizer_steps += 1
    # an optimiser that wrote the new parameters into its networks' fragment images itself (demiss_vae_amd.optim.Adam
    # with attach_images) leaves those images current
    for net in getattr(opt, "_images_written", ()):
        net._stamp = net._current_stamp()


register_optimizer_step_post_hook(_count_optimizer_step)


_SLAB_OWNERS = {}       # storage address of a gradient slab -> weak reference to its PackedNet


def slab_owner(grad: torch.Tensor):
    """the PackedNet whose gradient slab `grad` is a view of, or None"""
    ref = _SLAB_OWNERS.get(grad.untyped_storage().data_ptr())
    net = ref() if ref is not None else None
    if net is None or net._slab is None or net._slab.untyped_storage().data_ptr() != grad.untyped_storage().data_ptr():
        return None
    return net


def pad32(n: int) -> int:
    return 32 * ((n + 31) // 32)


@dataclass(frozen=True)
class NetDims:
    """Constructor arguments of ResidualFCNetwork (vgiwae/shared/neural_nets.py:92-99)."""
    in_dim: int
    out_dim: int
    nb: int
    hidden: int
    act: str = "relu"
    dropout: float = 0.0

    def c_struct(self):
        if self.act not in L.ACT:
            raise NotImplementedError(f"activation {self.act!r} has no HIP kernel")
        return L.DmvNet(self.in_dim, self.out_dim, self.hidden, self.nb, L.ACT[self.act], float(self.dropout))

    @property
    def num_layers(self):
        return 2 * self.nb + 2


class PackedNet:
    """A network's parameters (2*nb+2 weight and bias tensors, nn.Linear layout, in layer order)
    plus their MFMA-fragment image on the device."""

    def __init__(self, dims: NetDims, weights: Sequence[torch.Tensor], biases: Sequence[torch.Tensor]):
        assert len(weights) == dims.num_layers and len(biases) == dims.num_layers
        self.dims = dims
        self.weights = list(weights)
        self.biases = list(biases)
        self.c = dims.c_struct()
        self._packed = None
        self._stamp = None
        n = L.lib().dmv_mlp_packed_floats(C.byref(self.c))
        if n == 0:
            raise RuntimeError(f"unsupported network shape for the HIP kernels: {dims}")
        self._n_packed = n
        self.image_kept_by_optimizer = False     # set by demiss_vae_amd.optim.Adam.attach_images
        self.flat_grads = False
        self._slab = None
        self._slab_offsets = None
        self.on_grads_ready = None      # callable(net): the slab holds this step's complete gradients (see parallel.py)
        self.pending_exchange = None

    # ---- parameter gradients in one permanent buffer
    def use_flat_grads(self, on: bool = True):
        """Backward passes write this network's parameter gradients into one buffer that is allocated once (each
        tensor's gradient a view at a fixed, 256-byte aligned offset), so a data-parallel exchange reduces the buffer
        as it stands: no gather copy per step.  The gradients of a step are overwritten by the next backward pass
        (after zero_grad(set_to_none=True), as the training loops do): clone what must outlive the step."""
        self.flat_grads = bool(on)
        return self

    def grad_slab(self) -> torch.Tensor:
        if self._slab is None or self._slab.device != self.weights[0].device:
            offs, at = [], 0
            for t in self.params():
                offs.append(at)
                at += 64 * ((t.numel() + 63) // 64)
            self._slab = torch.zeros(at, dtype=torch.float32, device=self.weights[0].device)
            self._slab_offsets = offs
            _SLAB_OWNERS[self._slab.untyped_storage().data_ptr()] = weakref.ref(self)
        return self._slab

    def grad_views(self):
        """Fresh views of the slab, one per parameter in params() order - or None while a parameter's .grad still
        lives in the slab (a second backward pass before zero_grad: autograd must then add to it, not alias it)."""
        slab = self.grad_slab()
        home = slab.untyped_storage().data_ptr()
        ps = self.params()
        for t in ps:
            if t.grad is not None and t.grad.untyped_storage().data_ptr() == home:
                return None
        return [slab[o:o + t.numel()].view(t.shape) for o, t in zip(self._slab_offsets, ps)]

    def refuse_pass_during_exchange(self):
        """A backward pass that produces parameter gradients while this network's overlapped exchange is in flight
        would accumulate into the buffer the collective is reducing, and its contribution would never be exchanged:
        refuse it (overlap_grad_exchange allows one backward pass per step; accumulate without the overlap instead)."""
        if self.pending_exchange is not None:
            raise RuntimeError("a backward pass reached this network's parameters while the exchange of its gradients "
                               "from an earlier pass is in flight: with overlap_grad_exchange a step has one backward "
                               "pass per network (call allreduce_grads first, or do not overlap)")

    def grads_live_in_slab(self) -> bool:
        """every parameter's .grad is the slab's view of it (autograd kept the tensors the backward pass returned)"""
        if self._slab is None:
            return False
        home = self._slab.untyped_storage().data_ptr()
        for o, t in zip(self._slab_offsets, self.params()):
            g = t.grad
            if g is None or g.untyped_storage().data_ptr() != home or g.storage_offset() != o or not g.is_contiguous():
                return False
        return True

    def params(self) -> List[torch.Tensor]:
        out = []
        for w, b in zip(self.weights, self.biases):
            out += [w, b]
        return out

    def invalidate(self):
        """Forget the fragment image: the next use re-packs.  Needed when the parameters changed without torch
        noticing (an optimiser step replayed inside a captured graph does not bump tensor versions)."""
        self._stamp = None

    def pack_maps(self):
        """DmvPackMap entries of the weight and bias tensors in params() order: where each lives in the fragment
        image (which is allocated here if it is not yet and then stays where it is)."""
        self.packed()
        nl = self.dims.num_layers
        ws, bs = (L.DmvPackMap * nl)(), (L.DmvPackMap * nl)()
        L.check(L.lib().dmv_mlp_pack_map(C.byref(self.c), L.ptr(self._packed), ws, bs))
        out = []
        for w, b in zip(ws, bs):
            out += [w, b]
        return out

    def _current_stamp(self):
        return (_optimizer_steps,) + tuple((t.data_ptr(), t._version) for t in self.params())

    def packed(self) -> torch.Tensor:
        """Fragment image of the current weights (re-packed only when a parameter changed)."""
        stamp = self._current_stamp()
        if self._packed is None or stamp != self._stamp or self._packed.device != self.weights[0].device:
            dev = self.weights[0].device
            for t in self.params():
                if t.dtype != torch.float32 or not t.is_contiguous() or t.device != dev or not t.is_cuda:
                    raise RuntimeError("network parameters must be contiguous fp32 CUDA tensors on one device")
            if self._packed is None or self._packed.device != dev:
                self._packed = _image_storage(self._n_packed, dev)
            L.check(L.lib().dmv_mlp_pack(C.byref(self.c), L.ptr_array([w.detach() for w in self.weights]),
                                         L.ptr_array([b.detach() for b in self.biases]),
                                         L.ptr(self._packed), L.stream_ptr()))
            self._stamp = stamp
        return self._packed


# Weight images come out of one slab per device (bump allocation, 4 KiB granules; images that no longer fit get a tensor
# of their own).  Every workgroup of the small-batch kernels streams both networks' images from L2 over and over, and where
# two separately allocated 2 MB images fall in memory decides whether they stay there: measured in the one-launch sampler,
# the decoder's chain took 59 k cycles with the caching allocator's placement and 52.6 k - the encoder's time - with both
# images in one large allocation, at any offsets within it.
IMAGE_SLAB_BYTES = int(os.environ.get("DMV_IMAGE_SLAB_MB", "64")) << 20
_image_slabs = {}


def _image_storage(n_floats: int, dev) -> torch.Tensor:
    nbytes = (4 * n_floats + 4095) & ~4095
    key = (dev.type, dev.index if dev.index is not None else torch.cuda.current_device())
    slab = _image_slabs.get(key)
    if slab is None and IMAGE_SLAB_BYTES > 0:
        slab = _image_slabs[key] = [torch.empty(IMAGE_SLAB_BYTES, dtype=torch.uint8, device=dev), 0]
    if slab is None or slab[1] + nbytes > slab[0].numel():
        return torch.empty(n_floats, dtype=torch.float32, device=dev)
    off = slab[1]
    slab[1] = off + nbytes
    return slab[0][off: off + 4 * n_floats].view(torch.float32)


class GeneratedDropout:
    """Stands where a keep-mask tensor would: the kernels draw the keep decisions of this call themselves from the
    library's Philox stream (relu networks), so no [nb, rows, hidden] mask array is written or read.  reserve() takes
    the counter range of one forward call; keep_bytes() writes the same decisions out as a mask tensor."""

    def __init__(self, net: PackedNet, rows: int):
        from . import rng
        self.net, self.rows = net, rows
        n = L.lib().dmv_mlp_dropout_rng_elements(C.byref(net.c), rows)
        seed, off = rng.next_offset(n)
        self.c = L.DmvDropRng(seed, off, rng.base_ptr())

    @staticmethod
    def supported(net: PackedNet) -> bool:
        return net.dims.act == "relu"

    def keep_bytes(self) -> torch.Tensor:
        d = self.net.dims
        keep = torch.empty((d.nb, self.rows, d.hidden), dtype=torch.uint8, device=self.net.weights[0].device)
        L.check(L.lib().dmv_mlp_dropout_keep(C.byref(self.net.c), self.rows, C.byref(self.c), L.ptr(keep), L.stream_ptr()))
        return keep


def _drop_args(drop_keep):
    """(drop_keep pointer, drop_rng pointer) of dmv_mlp_forward"""
    if isinstance(drop_keep, GeneratedDropout):
        return None, C.byref(drop_keep.c)
    return L.ptr(drop_keep), None


def mlp_forward(net: PackedNet, x: torch.Tensor, ldx: int, rows: int, *, in_mode=L.IN_PLAIN, D: Optional[int] = None,
                m: Optional[torch.Tensor] = None, ldm: int = 0, save: bool = False, drop_keep=None):
    """y [rows, pad32(out_dim)] (columns past out_dim are zero) and the saved-activation buffer (or None).
    drop_keep: None, a keep-mask tensor [nb, rows, hidden] uint8, or a GeneratedDropout."""
    d = net.dims
    if D is None:
        D = d.in_dim
    dev = x.device
    y = torch.empty((rows, pad32(d.out_dim)), dtype=torch.float32, device=dev)
    saved = None
    if save:
        saved = torch.empty(L.lib().dmv_mlp_saved_floats(C.byref(net.c), rows), dtype=torch.float32, device=dev)
    if rows:
        if isinstance(drop_keep, GeneratedDropout):
            assert drop_keep.net is net and drop_keep.rows == rows
        kp, kr = _drop_args(drop_keep)
        L.check(L.lib().dmv_mlp_forward(C.byref(net.c), L.ptr(net.packed()), in_mode, D, L.ptr(x), ldx, L.ptr(m), ldm,
                                        rows, L.ptr(y), y.shape[1], L.ptr(saved), kp, kr, L.stream_ptr()))
    return y, saved


_SIDE_STREAMS = {}


def side_stream(device) -> torch.cuda.Stream:
    """one extra stream per device for work that may overlap the main stream's (weight-gradient GEMMs of small batches)"""
    key = torch.device(device).index
    if key not in _SIDE_STREAMS:
        _SIDE_STREAMS[key] = torch.cuda.Stream(device=device)
    return _SIDE_STREAMS[key]


def overlap_pays(rows: int) -> bool:
    """The weight-gradient GEMMs of a batch this small leave part of the GPU idle, and so do the latent-side kernel and
    the encoder's backward pass that follow them: worth a second stream.  Measured gain at 20 480 decoder rows 2 %, at
    40 960 rows 1 % of the step; beyond that both sides fill the chip on their own."""
    return 0 < rows <= 49152


def mlp_backward(net: PackedNet, saved: torch.Tensor, gy: torch.Tensor, rows: int, *, need_dx: bool,
                 need_param_grads: bool, drop_keep=None,
                 wgrad_stream: Optional[torch.cuda.Stream] = None, keepalive: Optional[list] = None,
                 complete: bool = False):
    """gy [rows, pad32(out_dim)] -> (dx [rows, pad32(in_dim)] or None, [gW0, gb0, gW1, ...] or None).

    complete: these are the network's whole gradients for this step; with use_flat_grads() and a registered
    on_grads_ready the hook is called on the stream that carries the weight-gradient kernels (a data-parallel exchange
    of this network's gradients can start while the caller carries on upstream).

    wgrad_stream: run the weight-gradient GEMMs there, after the input-gradient chain on the current stream; the
    caller makes the current stream wait for it before the gradients are used and keeps `keepalive` (filled with the
    buffers the side stream reads) alive until then."""
    d = net.dims
    dev = gy.device
    assert gy.shape == (rows, pad32(d.out_dim)) and gy.is_contiguous()
    keep_ptr = L.DROP_GENERATED if isinstance(drop_keep, GeneratedDropout) else L.ptr(drop_keep)
    dx = torch.empty((rows, pad32(d.in_dim)), dtype=torch.float32, device=dev) if need_dx else None
    gws = gbs = None
    gw_arr = gb_arr = None
    in_slab = False
    if need_param_grads:
        net.refuse_pass_during_exchange()
        views = net.grad_views() if net.flat_grads else None
        if views is not None:
            gws, gbs, in_slab = views[0::2], views[1::2], True
        else:
            gws = [torch.empty_like(w) for w in net.weights]
            gbs = [torch.empty_like(b) for b in net.biases]
        gw_arr, gb_arr = L.ptr_array(gws), L.ptr_array(gbs)
    nbytes = L.lib().dmv_mlp_backward_ws_bytes(C.byref(net.c), rows)
    ws = torch.empty((nbytes + 3) // 4, dtype=torch.float32, device=dev)
    if rows and wgrad_stream is not None and need_param_grads:
        lib = L.lib()
        L.check(lib.dmv_mlp_backward(C.byref(net.c), L.ptr(net.packed()), L.ptr(saved), keep_ptr,
                                     L.ptr(gy), gy.shape[1], rows, L.ptr(dx), dx.shape[1] if need_dx else 0,
                                     None, None, 0.0, L.ptr(ws), nbytes, L.stream_ptr()))
        wgrad_stream.wait_stream(torch.cuda.current_stream())
        L.check(lib.dmv_mlp_backward_wgrad(C.byref(net.c), L.ptr(saved), L.ptr(gy), gy.shape[1], rows, gw_arr, gb_arr, 0.0,
                                           L.ptr(ws), nbytes, C.c_void_p(wgrad_stream.cuda_stream)))
        keepalive += [ws, gy, saved]
        if complete and in_slab and net.on_grads_ready is not None:
            with torch.cuda.stream(wgrad_stream):
                net.on_grads_ready(net)
            complete = False
    elif rows:
        L.check(L.lib().dmv_mlp_backward(C.byref(net.c), L.ptr(net.packed()), L.ptr(saved), keep_ptr,
                                         L.ptr(gy), gy.shape[1], rows, L.ptr(dx), dx.shape[1] if need_dx else 0,
                                         gw_arr, gb_arr, 0.0, L.ptr(ws), nbytes, L.stream_ptr()))
    elif need_param_grads:
        for t in gws + gbs:
            t.zero_()
    grads = None
    if need_param_grads:
        grads = []
        for w, b in zip(gws, gbs):
            grads += [w, b]
        if complete and in_slab and net.on_grads_ready is not None:
            net.on_grads_ready(net)
    return dx, grads


def mlp_backward_chain(net: PackedNet, saved: torch.Tensor, gy: torch.Tensor, rows: int, *, need_dx: bool, drop_keep=None):
    """The input-gradient half of a backward pass on its own (dmv_mlp_backward with grad_w = NULL): (dx or None, ws,
    nbytes); ws holds the per-layer gradients that mlp_wgrad / mlp_wgrad_pair turn into weight gradients."""
    d = net.dims
    dev = gy.device
    assert rows > 0 and gy.shape == (rows, pad32(d.out_dim)) and gy.is_contiguous()
    keep_ptr = L.DROP_GENERATED if isinstance(drop_keep, GeneratedDropout) else L.ptr(drop_keep)
    dx = torch.empty((rows, pad32(d.in_dim)), dtype=torch.float32, device=dev) if need_dx else None
    nbytes = L.lib().dmv_mlp_backward_ws_bytes(C.byref(net.c), rows)
    ws = torch.empty((nbytes + 3) // 4, dtype=torch.float32, device=dev)
    L.check(L.lib().dmv_mlp_backward(C.byref(net.c), L.ptr(net.packed()), L.ptr(saved), keep_ptr, L.ptr(gy), gy.shape[1], rows,
                                     L.ptr(dx), dx.shape[1] if need_dx else 0, None, None, 0.0, L.ptr(ws), nbytes,
                                     L.stream_ptr()))
    return dx, ws, nbytes


def mlp_wgrad(net: PackedNet, saved: torch.Tensor, gy: torch.Tensor, rows: int, ws: torch.Tensor, nbytes: int, *,
              stream: Optional[torch.cuda.Stream] = None, complete: bool = False):
    """The weight-gradient half of a backward pass whose input-gradient chain already left the per-layer gradients in
    `ws` (dmv_mlp_backward with grad_w = NULL, or dmv_vae_backward): [gW0, gb0, gW1, ...], into the network's permanent
    gradient buffer when it has one.  stream: run there (the caller has made it wait for the chain and joins it before
    the gradients are used); complete: as in mlp_backward."""
    net.refuse_pass_during_exchange()
    views = net.grad_views() if net.flat_grads else None
    if views is not None:
        gws, gbs, in_slab = views[0::2], views[1::2], True
    else:
        gws, gbs, in_slab = [torch.empty_like(w) for w in net.weights], [torch.empty_like(b) for b in net.biases], False
    st = C.c_void_p(stream.cuda_stream) if stream is not None else L.stream_ptr()
    L.check(L.lib().dmv_mlp_backward_wgrad(C.byref(net.c), L.ptr(saved), L.ptr(gy), gy.shape[1], rows, L.ptr_array(gws),
                                           L.ptr_array(gbs), 0.0, L.ptr(ws), nbytes, st))
    if complete and in_slab and net.on_grads_ready is not None:
        if stream is not None:
            with torch.cuda.stream(stream):
                net.on_grads_ready(net)
        else:
            net.on_grads_ready(net)
    grads = []
    for w, b in zip(gws, gbs):
        grads += [w, b]
    return grads


def mlp_wgrad_pair(net_a: PackedNet, saved_a, gy_a, rows_a: int, ws_a, nbytes_a: int, net_b: PackedNet, saved_b, gy_b,
                   rows_b: int, ws_b, nbytes_b: int, *, complete_a: bool = False, complete_b: bool = False):
    """mlp_wgrad for two networks whose input-gradient chains are done, as one pair of launches on the current stream
    (dmv_mlp_backward_wgrad_pair): (gradients of a, gradients of b), or None when the library does not take the pair."""
    out = []
    for net in (net_a, net_b):
        net.refuse_pass_during_exchange()
        views = net.grad_views() if net.flat_grads else None
        if views is not None:
            out.append((views[0::2], views[1::2], True))
        else:
            out.append(([torch.empty_like(w) for w in net.weights], [torch.empty_like(b) for b in net.biases], False))
    (gwa, gba, slab_a), (gwb, gbb, slab_b) = out
    rc = L.lib().dmv_mlp_backward_wgrad_pair(
        C.byref(net_a.c), L.ptr(saved_a), L.ptr(gy_a), gy_a.shape[1], rows_a, L.ptr_array(gwa), L.ptr_array(gba), L.ptr(ws_a),
        nbytes_a, C.byref(net_b.c), L.ptr(saved_b), L.ptr(gy_b), gy_b.shape[1], rows_b, L.ptr_array(gwb), L.ptr_array(gbb),
        L.ptr(ws_b), nbytes_b, 0.0, L.stream_ptr())
    if rc == L.E_UNSUPPORTED:
        return None
    L.check(rc)
    for net, complete, in_slab in ((net_a, complete_a, slab_a), (net_b, complete_b, slab_b)):
        if complete and in_slab and net.on_grads_ready is not None:
            net.on_grads_ready(net)
    flat = lambda ws_, bs_: [t for pair in zip(ws_, bs_) for t in pair]
    return flat(gwa, gba), flat(gwb, gbb)


def make_keep_masks(dims: NetDims, rows: int, device, generator: Optional[torch.Generator] = None):
    """Bernoulli(1-p) keep masks [nb, rows, hidden] uint8 for the blocks' dropout layers."""
    if dims.dropout <= 0 or dims.nb == 0:
        return None
    u = torch.rand((dims.nb, rows, dims.hidden), device=device, generator=generator)
    return (u >= dims.dropout).to(torch.uint8)


class ResidualMLPFunction(torch.autograd.Function):
    """y = net(x) for x [rows, in_dim] through the HIP kernels, differentiable w.r.t. x and the parameters.

    The general-purpose entry used wherever the reference calls a network directly
    (e.g. model.generator_network(Z), vgiwae/models/vae.py:279,309)."""

    @staticmethod
    def forward(ctx, net: PackedNet, grad: bool, drop_keep, x, *params):
        d = net.dims
        x2 = x.reshape(-1, d.in_dim).contiguous().float()
        rows = x2.shape[0]
        # grad: gradient mode at call time (needs_input_grad mirrors requires_grad even when gradients are switched
        # off, and inside forward the mode always reads "off")
        need = grad and any(ctx.needs_input_grad[3:])
        y, saved = mlp_forward(net, x2, d.in_dim, rows, save=need, drop_keep=drop_keep)
        ctx.net, ctx.saved, ctx.rows, ctx.drop_keep = net, saved, rows, drop_keep
        ctx.lead = x.shape[:-1]
        return y[:, :d.out_dim].reshape(*x.shape[:-1], d.out_dim)

    @staticmethod
    def backward(ctx, gy):
        net, d = ctx.net, ctx.net.dims
        g = torch.zeros((ctx.rows, pad32(d.out_dim)), dtype=torch.float32, device=gy.device)
        g[:, :d.out_dim] = gy.reshape(ctx.rows, d.out_dim)
        need_dx = ctx.needs_input_grad[3]
        need_p = any(ctx.needs_input_grad[4:])
        dx, grads = mlp_backward(net, ctx.saved, g, ctx.rows, need_dx=need_dx, need_param_grads=need_p,
                                 drop_keep=ctx.drop_keep)
        gx = dx[:, :d.in_dim].reshape(*ctx.lead, d.in_dim) if need_dx else None
        if grads is None:
            grads = [None] * (2 * d.num_layers)
        return (None, None, None, gx, *grads)
