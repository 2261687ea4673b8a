"""HIP residual-MLP kernels vs the golden vectors (reference outputs) and the CPU oracle.  Through the C ABI."""
import pytest
import torch

import golden_util as gu
from oracle import core
from oracle.recipes import NetSpec, make_noise, make_params

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("name", gu.names("mlp_"))
def test_mlp_golden(name):
    import demiss_vae_amd as dmv
    import gpu_util as G
    g = gu.load(name)
    spec = gu.net_spec(g["spec"], g["dropout"])
    seed = int(g["seed"])
    params = gu.stored_params(g, "p", spec, seed)
    net = G.packed_net(spec, params)
    x, gy, rows = gu.mlp_io(g, spec)
    x, gy = x.cuda().requires_grad_(True), gy.cuda()
    keep = G.keep_to_dev(gu.dropout_masks(seed + 3, spec, (rows,)))
    y = dmv.ResidualMLPFunction.apply(net, True, keep, x, *net.params())
    y.backward(gy)
    # long batches (rows above every small-batch kernel family: the one-wave kernels the headline benchmark times)
    # hold digests of the reference's outputs instead of the full arrays, and leave out rows at a relu kink
    gu.check_tensor(g, "y", y, 1e-5, 2e-6)
    gu.check_tensor(g, "dx", x.grad, 1e-4, 2e-6)
    assert gu.check_grads(g, "g", G.grads_of(net, spec), 1e-4, 2e-6) == 2 * (2 * spec.nb + 2)


@pytest.mark.parametrize("shape", [(2, 10, 3, 200, 1000), (16, 16, 2, 256, 4096 + 17), (86, 325, 5, 256, 333),
                                   (7, 3, 0, 17, 65), (16, 32, 1, 96, 31)])
@pytest.mark.parametrize("act", ["relu", "tanh"])
def test_mlp_vs_oracle(shape, act):
    """Shapes of the toy / gas / miniboone networks plus edge cases (no residual block, ragged row counts)."""
    import demiss_vae_amd as dmv
    import gpu_util as G
    i, o, nb, h, rows = shape
    spec = NetSpec(i, o, nb, h, act=act)
    params = make_params(spec, 77)
    net = G.packed_net(spec, params)
    x_cpu = make_noise(78, rows, i)
    gy_cpu = make_noise(79, rows, o)
    x = x_cpu.cuda().requires_grad_(True)
    y = dmv.ResidualMLPFunction.apply(net, True, None, x, *net.params())
    y.backward(gy_cpu.cuda())
    p64 = {k: v.double().requires_grad_(True) for k, v in params.items()}
    x64 = x_cpu.double().requires_grad_(True)
    y64 = core.mlp_forward(p64, x64, spec)
    y64.backward(gy_cpu.double())
    torch.testing.assert_close(y.detach().cpu().double(), y64.detach(), rtol=1e-5, atol=1e-5)
    torch.testing.assert_close(x.grad.cpu().double(), x64.grad, rtol=1e-4, atol=1e-5)
    got = G.grads_of(net, spec)
    for k, v in p64.items():
        ref = v.grad
        tol = 2e-5 * float(ref.abs().max()) + 1e-6
        torch.testing.assert_close(got[k].cpu().double(), ref, rtol=1e-4, atol=tol, msg=lambda m: f"{k}: {m}")


def test_mlp_empty_and_errors():
    import demiss_vae_amd as dmv
    import gpu_util as G
    spec = NetSpec(4, 4, 1, 32)
    net = G.packed_net(spec, make_params(spec, 1))
    y = dmv.ResidualMLPFunction.apply(net, True, None, torch.zeros(0, 4, device="cuda"), *net.params())
    assert y.shape == (0, 4)
    with pytest.raises(RuntimeError):
        dmv.PackedNet(dmv.NetDims(4, 4, 1, 520), net.weights, net.biases)   # hidden width the kernels do not cover
    with pytest.raises(RuntimeError):
        dmv.PackedNet(dmv.NetDims(4, 4, 1, 512, act="tanh"), net.weights, net.biases)   # above 256: relu only


def test_small_batch_and_large_batch_kernels_agree_bitwise():
    """A whole round of 1024 row tiles runs with one wave per tile for the whole network; a quarter of it (256 tiles)
    with a tile shared by the four waves of a workgroup.  Both accumulate every output in the same order: identical
    bits for outputs and input gradients."""
    import demiss_vae_amd as dmv
    from demiss_vae_amd.mlp import ResidualMLPFunction
    from oracle.recipes import NetSpec, make_params
    import gpu_util as G
    torch.manual_seed(0)
    for spec in (NetSpec(16, 16, 2, 256), NetSpec(5, 4, 3, 200)):
        p = make_params(spec, 11)
        rows = 1024 * 32                          # one full round of one-wave tiles; a quarter of it: four-wave tiles
        x = torch.randn(rows, spec.in_dim, device="cuda")
        gy = torch.randn(rows, spec.out_dim, device="cuda")
        outs = []
        for chunks in (1, 4):
            net = G.packed_net(spec, p)
            ys, dxs = [], []
            for xc, gc in zip(x.chunk(chunks), gy.chunk(chunks)):
                xc = xc.clone().requires_grad_(True)
                y = ResidualMLPFunction.apply(net, True, None, xc, *net.params())
                y.backward(gc)
                ys.append(y.detach())
                dxs.append(xc.grad)
            outs.append((torch.cat(ys), torch.cat(dxs), [w.grad.clone() for w in net.weights]))
        (y1, dx1, gw1), (y3, dx3, gw3) = outs
        assert torch.equal(y1, y3) and torch.equal(dx1, dx3)
        for a, b in zip(gw1, gw3):                 # weight gradients: different row slicing, same values to rounding
            torch.testing.assert_close(a, b, rtol=2e-4, atol=2e-4 * float(a.abs().max()))


def _rows_clear_of_relu_kinks(spec, params, x, margin=2e-6):
    """Rows none of whose hidden pre-activations lies within `margin` of zero (float64).  Where one does, fp32 and
    fp64 can disagree on the sign, i.e. on the relu derivative of a whole row: such rows (a handful per million
    pre-activations) say nothing about the kernels and are left out of oracle comparisons."""
    p = {k: v.double() for k, v in params.items()}
    a = x.double() @ p["initial_layer.weight"].T + p["initial_layer.bias"]
    closest = a.abs().min(dim=1).values
    a = torch.relu(a)
    for b in range(spec.nb):
        t = a @ p[f"blocks.{b}.linear_layers.0.weight"].T + p[f"blocks.{b}.linear_layers.0.bias"]
        closest = torch.minimum(closest, t.abs().min(dim=1).values)
        u = torch.relu(t) @ p[f"blocks.{b}.linear_layers.1.weight"].T + p[f"blocks.{b}.linear_layers.1.bias"] + a
        closest = torch.minimum(closest, u.abs().min(dim=1).values)
        a = torch.relu(u)
    return closest > margin


@pytest.mark.parametrize("hidden", [32, 64, 100, 128, 160, 192, 224, 256])
@pytest.mark.parametrize("rows", [45, 1024])
def test_sixteen_row_kernels_all_widths(hidden, rows):
    """Batches of at most 4096 rows run on 16-row tiles (another MFMA shape, another summation order, quarter
    fragments of the same packed images).  Every hidden tile count, odd fragment-group counts of the input and the
    output width (20 -> 3 groups, 40 -> 5), ragged row counts: against the float64 oracle, and against the same rows
    inside a batch large enough for the 32-row kernels."""
    import demiss_vae_amd as dmv
    import gpu_util as G
    spec = NetSpec(20, 40, 2, hidden)
    params = make_params(spec, 5)
    x_cpu, gy_cpu = make_noise(6, rows, 20), make_noise(7, rows, 40)
    keep = _rows_clear_of_relu_kinks(spec, params, x_cpu)
    assert int(keep.sum()) >= 0.9 * len(keep)
    x_cpu, gy_cpu, rows = x_cpu[keep], gy_cpu[keep], int(keep.sum())

    def run(x_in, gy_in):
        net = G.packed_net(spec, params)
        x = x_in.cuda().requires_grad_(True)
        y = dmv.ResidualMLPFunction.apply(net, True, None, x, *net.params())
        y.backward(gy_in.cuda())
        return y.detach().cpu(), x.grad.cpu(), G.grads_of(net, spec)

    y, dx, got = run(x_cpu, gy_cpu)
    p64 = {k: v.double().requires_grad_(True) for k, v in params.items()}
    x64 = x_cpu.double().requires_grad_(True)
    y64 = core.mlp_forward(p64, x64, spec)
    y64.backward(gy_cpu.double())
    torch.testing.assert_close(y.double(), y64.detach(), rtol=1e-5, atol=1e-5)
    torch.testing.assert_close(dx.double(), x64.grad, rtol=1e-4, atol=1e-5)
    for k, v in p64.items():
        tol = 2e-5 * float(v.grad.abs().max()) + 1e-6
        torch.testing.assert_close(got[k].cpu().double(), v.grad, rtol=1e-4, atol=tol, msg=lambda m: f"{k}: {m}")
    # the same rows at the head of a 6000-row batch (32-row tiles); the padding rows carry no output gradient
    pad = 6000 - rows
    yb, dxb, gotb = run(torch.cat([x_cpu, make_noise(8, pad, 20)]), torch.cat([gy_cpu, torch.zeros(pad, 40)]))
    torch.testing.assert_close(y, yb[:rows], rtol=1e-5, atol=1e-5)
    torch.testing.assert_close(dx, dxb[:rows], rtol=1e-4, atol=1e-5)
    for k in got:
        tol = 2e-5 * float(gotb[k].abs().max()) + 1e-6
        torch.testing.assert_close(got[k].cpu(), gotb[k].cpu(), rtol=1e-4, atol=tol, msg=lambda m: f"{k}: {m}")


@pytest.mark.parametrize("shape", [(50, 784, 2, 512, 300), (784, 100, 2, 512, 300), (20, 40, 1, 288, 45), (20, 40, 1, 416, 45),
                                   (20, 40, 1, 288, 4400), (24, 36, 2, 512, 4700), (20, 40, 1, 416, 9000)])
def test_hidden_widths_above_256(shape):
    """Hidden widths 257..512 (9..16 register tiles): the 16-row kernels up to 4096 rows, beyond that rounds of four-wave
    32-row tiles with a last partial round of 16-row tiles (4400 and 4700 rows: four-wave tiles only; 9000 rows: one
    four-wave round and 808 rows of 16-row tiles).  The synthetic 784-dimensional MLP pair (not a reference
    configuration) and intermediate widths, against the float64 oracle."""
    import demiss_vae_amd as dmv
    import gpu_util as G
    i, o, nb, h, rows = shape
    spec = NetSpec(i, o, nb, h)
    params = make_params(spec, 21)
    net = G.packed_net(spec, params)
    x_cpu, gy_cpu = make_noise(22, rows, i), make_noise(23, rows, o)
    keep = _rows_clear_of_relu_kinks(spec, params, x_cpu)
    assert int(keep.sum()) >= 0.9 * len(keep)
    x_cpu, gy_cpu = x_cpu[keep], gy_cpu[keep]
    x = x_cpu.cuda().requires_grad_(True)
    y = dmv.ResidualMLPFunction.apply(net, True, None, x, *net.params())
    y.backward(gy_cpu.cuda())
    p64 = {k: v.double().requires_grad_(True) for k, v in params.items()}
    x64 = x_cpu.double().requires_grad_(True)
    y64 = core.mlp_forward(p64, x64, spec)
    y64.backward(gy_cpu.double())
    torch.testing.assert_close(y.detach().cpu().double(), y64.detach(), rtol=1e-5, atol=1e-5)
    torch.testing.assert_close(x.grad.cpu().double(), x64.grad, rtol=1e-4, atol=1e-5)
    got = G.grads_of(net, spec)
    for k, v in p64.items():
        tol = 2e-5 * float(v.grad.abs().max()) + 1e-6
        torch.testing.assert_close(got[k].cpu().double(), v.grad, rtol=1e-4, atol=tol, msg=lambda m: f"{k}: {m}")


def test_hidden_512_many_workgroups_bitwise():
    """Hidden 512 at 4000 rows (sixteen-row tiles, a full round of workgroups): outputs and input gradients equal those of
    the same rows run in three calls, bit for bit; and at 4500 rows (four-wave 32-row tiles) the same pass twice gives
    the same bits and agrees with the three 1500-row calls (sixteen-row tiles, another summation order inside the
    matrix instructions) to rounding.  Not compared with the float64 oracle at this size: among 11 million
    pre-activations one or two lie within rounding of zero, where fp32 and fp64 disagree on the relu derivative of a
    whole row."""
    import demiss_vae_amd as dmv
    import gpu_util as G
    spec = NetSpec(784, 100, 2, 512)
    params = make_params(spec, 21)

    def run(x, gy, chunks):
        net = G.packed_net(spec, params)
        ys, dxs = [], []
        for xc, gc in zip(x.chunk(chunks), gy.chunk(chunks)):
            xc = xc.clone().requires_grad_(True)
            y = dmv.ResidualMLPFunction.apply(net, True, None, xc, *net.params())
            y.backward(gc)
            ys.append(y.detach())
            dxs.append(xc.grad)
        return torch.cat(ys), torch.cat(dxs), [w.grad.clone() for w in net.weights]

    x5, gy5 = make_noise(24, 4500, 784).cuda(), make_noise(25, 4500, 100).cuda()
    a, b, c = run(x5, gy5, 1), run(x5, gy5, 1), run(x5, gy5, 3)
    assert torch.equal(a[0], b[0]) and torch.equal(a[1], b[1])
    # a row with a pre-activation within rounding of zero may take the other relu branch under the other summation
    # order: compare all but a handful of rows
    bad_y = ((a[0] - c[0]).abs() > 1e-4 * (1 + c[0].abs())).any(1)
    bad_dx = ((a[1] - c[1]).abs() > 1e-3 * (c[1].abs().max())).any(1)
    assert int(bad_y.sum()) <= 4 and int(bad_dx.sum()) <= 8, (int(bad_y.sum()), int(bad_dx.sum()))
    for ga, gc in zip(a[2], c[2]):
        # (a flipped row changes the weight-gradient entries of its units by that row's contribution)
        far = (ga - gc).abs() > 1e-3 * float(ga.abs().max())
        assert float(far.float().mean()) < 5e-3
        torch.testing.assert_close(ga, gc, rtol=0, atol=2e-2 * float(ga.abs().max()))

    rows = 4000
    x, gy = make_noise(22, rows, 784).cuda(), make_noise(23, rows, 100).cuda()
    outs = []
    for chunks in (1, 3):
        net = G.packed_net(spec, params)
        ys, dxs = [], []
        for xc, gc in zip(x.chunk(chunks), gy.chunk(chunks)):
            xc = xc.clone().requires_grad_(True)
            y = dmv.ResidualMLPFunction.apply(net, True, None, xc, *net.params())
            y.backward(gc)
            ys.append(y.detach())
            dxs.append(xc.grad)
        outs.append((torch.cat(ys), torch.cat(dxs), [w.grad.clone() for w in net.weights]))
    (y1, dx1, gw1), (y3, dx3, gw3) = outs
    assert torch.equal(y1, y3) and torch.equal(dx1, dx3)
    for a, b in zip(gw1, gw3):
        torch.testing.assert_close(a, b, rtol=2e-4, atol=2e-4 * float(a.abs().max()))


@pytest.mark.gpu
@pytest.mark.parametrize("fused", [False, True])
def test_packed_weights_follow_optimizer_steps(fused):
    """torch's fused optimisers update parameters without bumping their version counters: the fragment image must
    still be rebuilt after a step (the forward after the step = a forward through a freshly packed copy)."""
    import demiss_vae_amd as dmv
    from oracle.recipes import NetSpec, make_params
    import gpu_util as G
    spec = NetSpec(16, 16, 2, 64)
    net = G.packed_net(spec, make_params(spec, 5))
    x = torch.randn(96, 16, device="cuda")
    y0, _ = dmv.mlp_forward(net, x, 16, 96)
    opt = torch.optim.Adam(net.params(), lr=1e-2, fused=fused)
    for p in net.params():
        p.grad = torch.ones_like(p)
    opt.step()
    y1, _ = dmv.mlp_forward(net, x, 16, 96)
    fresh = dmv.PackedNet(net.dims, [w.detach().clone() for w in net.weights], [b.detach().clone() for b in net.biases])
    y2, _ = dmv.mlp_forward(fresh, x, 16, 96)
    assert not torch.equal(y0, y1) and torch.equal(y1, y2)


@pytest.mark.gpu
@pytest.mark.parametrize("clip", [0.0, 0.05])
def test_library_adam_matches_torch_adam(clip):
    """dmv_opt_adam_step vs torch.nn.utils.clip_grad_norm_ + torch.optim.Adam (two groups, AMSGrad on one of them,
    odd tensor sizes, a changing learning rate) over several steps; also the state_dict layout."""
    from demiss_vae_amd.optim import Adam
    torch.manual_seed(0)
    shapes = [(256, 16), (256,), (5000,), (3, 7), (1,), (4097,)]
    mine = [torch.randn(s, device="cuda").requires_grad_() for s in shapes]
    ref = [p.detach().clone().requires_grad_() for p in mine]

    def groups(ps):
        return [{"params": ps[:3], "lr": 1e-2, "amsgrad": True}, {"params": ps[3:], "lr": 3e-3, "amsgrad": False}]
    a, b = Adam(groups(mine)), torch.optim.Adam(groups(ref))
    for step in range(6):
        gs = [torch.randn(s, device="cuda") * (10.0 if step == 2 else 0.1) for s in shapes]
        for p, q, g in zip(mine, ref, gs):
            p.grad, q.grad = g.clone(), g.clone()
        if clip:
            a.clip_on_next_step(clip)
            norm = torch.nn.utils.clip_grad_norm_(ref, clip)
        if step == 3:
            for o in (a, b):
                o.param_groups[0]["lr"] = 5e-3
        a.step(), b.step()
        if clip:
            torch.testing.assert_close(a.grad_norm, norm, rtol=1e-5, atol=0)
        for p, q in zip(mine, ref):
            torch.testing.assert_close(p, q, rtol=1e-5, atol=5e-7)
    sa, sb = a.state_dict()["state"], b.state_dict()["state"]
    assert sa.keys() == sb.keys()
    for k in sa:
        assert sa[k].keys() == sb[k].keys() and float(sa[k]["step"]) == float(sb[k]["step"]) == 6
        torch.testing.assert_close(sa[k]["exp_avg_sq"], sb[k]["exp_avg_sq"], rtol=1e-5, atol=1e-9)
    assert torch.equal(mine[0].grad, gs[0])                                # gradients are not rescaled in memory


@pytest.mark.parametrize("shape", [(16, 16, 2, 256), (5, 4, 3, 200), (43, 86, 1, 64), (7, 3, 0, 17), (12, 20, 1, 320)])
def test_optimiser_keeps_the_fragment_images_current(shape):
    """Adam.attach_images: the step writes every new parameter value into its network's MFMA fragment image, so no
    packing kernel runs afterwards.  After several steps the image must equal, bit for bit, a fresh packing of the
    updated parameters (forward and transposed fragments, bias fragments, odd widths), the next forward pass must
    reuse it, and a parameter change from elsewhere must still trigger a re-pack - eagerly and in a replayed step."""
    import ctypes as C
    import demiss_vae_amd as dmv
    import gpu_util as G
    from demiss_vae_amd import _lib as L
    from demiss_vae_amd.optim import Adam
    spec = NetSpec(*shape)
    net = G.packed_net(spec, make_params(spec, 61))
    other = G.packed_net(NetSpec(4, 4, 1, 32), make_params(NetSpec(4, 4, 1, 32), 62))
    opt = Adam([{"params": net.params(), "lr": 1e-2, "amsgrad": True}, {"params": other.params(), "lr": 1e-3}])
    opt.attach_images((net, other))
    assert net.image_kept_by_optimizer and other.image_kept_by_optimizer
    x = make_noise(63, 100, shape[0]).cuda()

    def fresh_image():
        ref = dmv.PackedNet(net.dims, [w.detach().clone() for w in net.weights], [b.detach().clone() for b in net.biases])
        return ref.packed().clone()

    def step():
        opt.zero_grad(set_to_none=True)
        for n, xin in ((net, x), (other, x[:, :4].contiguous())):
            y = dmv.ResidualMLPFunction.apply(n, True, None, xin, *n.params())
            y.square().sum().backward()
        opt.clip_on_next_step(1.0)
        opt.step()

    for _ in range(3):
        step()
    image = net._packed
    assert net._stamp == net._current_stamp()                  # current without a packing launch ...
    assert torch.equal(image, fresh_image())                   # ... and equal to what packing would write
    assert net.packed().data_ptr() == image.data_ptr()
    # a change from outside the optimiser is still noticed
    with torch.no_grad():
        net.weights[0].mul_(1.5)
    assert net._stamp != net._current_stamp()
    assert torch.equal(net.packed(), fresh_image())
    # the same inside a replayed step
    g = dmv.GraphedStep(step, nets=(net, other), warmup=1)
    for _ in range(3):
        g()
    assert torch.equal(net._packed, fresh_image())
    with torch.no_grad():
        net.biases[-1].add_(1.0)
    g()
    assert torch.equal(net._packed, fresh_image())


@pytest.mark.parametrize("rows,hidden,nb", [(300, 64, 2), (5000, 256, 2), (20000, 256, 1), (777, 200, 3), (45, 96, 1), (200, 320, 1)])
@pytest.mark.parametrize("p", [0.5, 0.25, 0.1])
def test_generated_dropout_equals_its_own_masks(rows, hidden, nb, p):
    """Dropout decisions drawn inside the kernels (no mask array): in every kernel family (16-row tiles, four-wave
    tiles, one wave per tile, by batch size; hidden widths above 256) the outputs, input gradients and weight
    gradients are bit-identical to a run that is handed the same decisions as explicit keep bytes
    (dmv_mlp_dropout_keep), for dropout probabilities that draw 1, 8 and 16 bits per unit."""
    import demiss_vae_amd as dmv
    import gpu_util as G
    from demiss_vae_amd import rng
    spec = NetSpec(12, 20, nb, hidden, dropout=p)
    params = make_params(spec, 31)
    x_cpu, gy_cpu = make_noise(32, rows, 12), make_noise(33, rows, 20)

    def run(keep_of):
        net = G.packed_net(spec, params)
        x = x_cpu.cuda().requires_grad_(True)
        keep = keep_of(net)
        y = dmv.ResidualMLPFunction.apply(net, True, keep, x, *net.params())
        y.backward(gy_cpu.cuda())
        return y.detach(), x.grad, [t.grad for t in net.params()], keep

    rng.set_seed(99)
    rng.next_offset(40)
    y1, dx1, g1, gen = run(lambda net: dmv.GeneratedDropout(net, rows))
    masks = gen.keep_bytes()
    assert masks.shape == (nb, rows, hidden)
    y2, dx2, g2, _ = run(lambda net: masks)
    assert torch.equal(y1, y2) and torch.equal(dx1, dx2)
    for a, b in zip(g1, g2):
        assert torch.equal(a, b)
    # the decisions themselves: keep rate 1 - p, no dependence between neighbours, blocks or rows
    k = masks.float()
    n = k.numel()
    assert abs(k.mean().item() - (1 - p)) < 5 * (p * (1 - p) / n) ** 0.5 + 1e-5
    if n > 100000:
        c = k - k.mean()
        for a, b in ((c[:, :, :-1], c[:, :, 1:]), (c[:, :-1], c[:, 1:])) + (((c[:-1], c[1:]),) if nb > 1 else ()):
            assert abs((a * b).mean().item()) / (p * (1 - p)) < 6 / a.numel() ** 0.5
    # a second reservation draws other decisions
    assert not torch.equal(dmv.GeneratedDropout(G.packed_net(spec, params), rows).keep_bytes(), masks)


def test_explicit_masks_in_every_kernel_family_match_the_oracle():
    """recorded keep masks through the 16-row, four-wave and one-wave kernels against the float64 oracle"""
    import demiss_vae_amd as dmv
    import gpu_util as G
    spec = NetSpec(10, 6, 2, 128, dropout=0.5)
    params = make_params(spec, 41)
    for rows in (100, 6000, 40000):
        x_cpu, gy_cpu = make_noise(42, rows, 10), make_noise(43, rows, 6)
        ok = _rows_clear_of_relu_kinks(spec, params, x_cpu)
        masks = gu.dropout_masks(44, spec, (rows,))
        net = G.packed_net(spec, params)
        x = x_cpu.cuda().requires_grad_(True)
        y = dmv.ResidualMLPFunction.apply(net, True, G.keep_to_dev(masks), x, *net.params())
        y.backward(gy_cpu.cuda())
        p64 = {k: v.double() for k, v in params.items()}
        x64 = x_cpu.double().requires_grad_(True)
        y64 = core.mlp_forward(p64, x64, spec, masks)
        y64.backward(gy_cpu.double())
        torch.testing.assert_close(y.detach().cpu().double()[ok], y64.detach()[ok], rtol=1e-5, atol=1e-5)
        torch.testing.assert_close(x.grad.cpu().double()[ok], x64.grad[ok], rtol=1e-4, atol=1e-5)


@pytest.mark.parametrize("rows", [8192 + 4096 - 5, 32768 + 300, 32768 + 8192 + 1000, 2 * 32768 + 20000])
def test_batches_cut_into_launches_of_different_kernel_families(rows):
    """Whole rounds of 1024 row tiles run one wave per tile, the remainder on four-wave tiles, a last partial round on
    16-row tiles: every row of such a batch equals the same row computed alone in a small batch (bitwise for the
    first two families, to rounding on the 16-row tiles), and the weight gradients equal the float64 oracle's."""
    import demiss_vae_amd as dmv
    import gpu_util as G
    spec = NetSpec(16, 16, 2, 256)
    params = make_params(spec, 51)
    x_cpu, gy_cpu = make_noise(52, rows, 16), make_noise(53, rows, 16)
    net = G.packed_net(spec, params)
    x = x_cpu.cuda().requires_grad_(True)
    y = dmv.ResidualMLPFunction.apply(net, True, None, x, *net.params())
    y.backward(gy_cpu.cuda())
    # the same rows in chunks of one four-wave round (256 tiles)
    ys, dxs = [], []
    net2 = G.packed_net(spec, params)
    for lo in range(0, rows, 8192):
        xc = x_cpu[lo:lo + 8192].cuda().requires_grad_(True)
        yc = dmv.ResidualMLPFunction.apply(net2, True, None, xc, *net2.params())
        yc.backward(gy_cpu[lo:lo + 8192].cuda())
        ys.append(yc.detach())
        dxs.append(xc.grad)
    y2, dx2 = torch.cat(ys), torch.cat(dxs)
    ok = _rows_clear_of_relu_kinks(spec, params, x_cpu).cuda()
    torch.testing.assert_close(y.detach()[ok], y2[ok], rtol=1e-5, atol=1e-5)
    torch.testing.assert_close(x.grad[ok], dx2[ok], rtol=1e-4, atol=1e-5)
    whole = (rows // 32768) * 32768
    if whole:
        assert torch.equal(y.detach()[:whole], y2[:whole]) and torch.equal(x.grad[:whole], dx2[:whole])
    for a, b in zip(net.params(), net2.params()):
        torch.testing.assert_close(a.grad, b.grad, rtol=2e-4, atol=2e-4 * float(b.grad.abs().max()))


def test_groups_keep_their_device_scalars_when_their_rates_part():
    """two parameter groups that start with the same learning rate: every group has a device scalar of its own whose
    address never changes (a step recorded in a HIP graph holds it), whether a schedule moves the rates together or a
    group's rate is changed on its own; the update follows torch's"""
    from demiss_vae_amd.optim import Adam
    torch.manual_seed(3)
    mine = [torch.randn(300, device="cuda").requires_grad_(), torch.randn(70, 9, device="cuda").requires_grad_()]
    ref = [p.detach().clone().requires_grad_() for p in mine]
    groups = lambda ps: [{"params": [ps[0]], "lr": 1e-2}, {"params": [ps[1]], "lr": 1e-2}]
    a, b = Adam(groups(mine)), torch.optim.Adam(groups(ref))
    for step in range(6):
        for p, q in zip(mine, ref):
            g = torch.randn_like(p)
            p.grad, q.grad = g.clone(), g.clone()
        if step == 2:                       # a schedule moves both rates together
            for o in (a, b):
                for grp in o.param_groups:
                    grp["lr"] = 5e-3
        if step == 4:                       # one group on its own
            for o in (a, b):
                o.param_groups[1]["lr"] = 1e-3
        a.step(), b.step()
        if step == 0:
            homes = [t.data_ptr() for t in a._lr_dev + a._step_dev]
        assert a._lr_dev[0] is not a._lr_dev[1]
        assert [t.data_ptr() for t in a._lr_dev + a._step_dev] == homes
        for p, q in zip(mine, ref):
            torch.testing.assert_close(p, q, rtol=1e-5, atol=5e-7)
    assert float(a.state_dict()["state"][0]["step"]) == 6


@pytest.mark.gpu
def test_weight_images_share_a_slab_and_fall_back_when_it_is_full():
    """The first weight images of a process are views of one allocation (4 KiB granules); one that no longer fits gets a
    tensor of its own; either way the forward pass is the same."""
    import demiss_vae_amd as dmv
    from demiss_vae_amd import mlp, synth
    dev = torch.device("cuda", torch.cuda.current_device())
    key = (dev.type, dev.index)
    saved, saved_bytes = mlp._image_slabs.pop(key, None), mlp.IMAGE_SLAB_BYTES
    try:
        mlp.IMAGE_SLAB_BYTES = 6 << 20
        shape = synth.NetShape(16, 16, 2, 256)
        nets = [synth.packed_net(shape, 3, "cuda", requires_grad=False) for _ in range(4)]
        imgs = [n.packed() for n in nets]
        slab = mlp._image_slabs[key][0]
        lo, hi = slab.data_ptr(), slab.data_ptr() + slab.numel()
        inside = [lo <= t.data_ptr() < hi for t in imgs]
        assert inside[:2] == [True, True] and not inside[3], inside            # 2.2 MB each: two fit in 6 MB
        assert all(t.data_ptr() % 4096 == 0 for t, ok in zip(imgs, inside) if ok)
        x = torch.randn(100, 16, device="cuda")
        ys = [dmv.mlp_forward(n, x, 16, 100)[0] for n in nets]
        assert all(torch.equal(ys[0], y) for y in ys[1:])
    finally:
        mlp.IMAGE_SLAB_BYTES = saved_bytes
        mlp._image_slabs.pop(key, None)
        if saved is not None:
            mlp._image_slabs[key] = saved


@pytest.mark.gpu
@pytest.mark.parametrize("shapes", [((16, 32, 2, 256), 512, (16, 16, 2, 256), 2560),
                                    ((86, 325, 5, 256), 1024, (32, 86, 2, 256), 5120),
                                    ((10, 4, 3, 200), 320, (2, 10, 3, 200), 64),
                                    ((16, 32, 8, 64), 100, (16, 16, 3, 64), 500)])
def test_weight_gradients_of_two_networks_in_one_launch(shapes):
    """dmv_mlp_backward_wgrad_pair forms both networks' weight gradients as two dmv_mlp_backward_wgrad calls do (to
    rounding: the pair cuts the rows into fewer slices), the same bits call after call; networks with more than 24 layers
    between them are declined (the last case) and the caller makes the two calls."""
    import ctypes as C
    import demiss_vae_amd as dmv
    import gpu_util as G
    from demiss_vae_amd import _lib as L
    from demiss_vae_amd.mlp import mlp_forward, mlp_wgrad, mlp_wgrad_pair, pad32
    sa, ra, sb, rb = shapes
    done = []
    for k, (shape, rows) in enumerate(((sa, ra), (sb, rb))):
        spec = NetSpec(*shape)
        net = G.packed_net(spec, make_params(spec, 70 + k))
        x = make_noise(80 + k, rows, pad32(spec.in_dim)).cuda()
        _, saved = mlp_forward(net, x, x.shape[1], rows, save=True)
        gy = make_noise(90 + k, rows, pad32(spec.out_dim)).cuda()
        nbytes = L.lib().dmv_mlp_backward_ws_bytes(C.byref(net.c), rows)
        ws = torch.empty((nbytes + 3) // 4, device="cuda")
        L.check(L.lib().dmv_mlp_backward(C.byref(net.c), L.ptr(net.packed()), L.ptr(saved), None, L.ptr(gy), gy.shape[1], rows,
                                         None, 0, None, None, 0.0, L.ptr(ws), nbytes, L.stream_ptr()))
        done.append((net, saved, gy, rows, ws, nbytes))
    one = [mlp_wgrad(*d) for d in done]
    pair = mlp_wgrad_pair(*done[0], *done[1])
    if 2 * (sa[2] + sb[2]) + 4 > 24:
        assert pair is None
        return
    again = mlp_wgrad_pair(*done[0], *done[1])
    for a, b, c in zip(one[0] + one[1], pair[0] + pair[1], again[0] + again[1]):
        assert a.shape == b.shape and torch.equal(b, c) and float(a.abs().max()) > 0
        assert float((a - b).abs().max()) <= 1e-5 * float(a.abs().max())
