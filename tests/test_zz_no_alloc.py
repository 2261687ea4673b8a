"""The optimiser step that allocates nothing (dmv_opt_adam_step_static: workspace in a file-scope device array)
against the workspace-taking entry point it is a variant of (dmv_opt_adam_step), through the C ABI.  Same kernels, so
the results must agree bit for bit; two calls in a row, the second on a smaller problem, so that whatever the first
call left in the shared array cannot leak into the second."""
import ctypes as C

import pytest
import torch

pytestmark = pytest.mark.gpu


def _problem(sizes, seed, amsgrad):
    g = torch.Generator().manual_seed(seed)
    ps = [torch.randn(n, generator=g).cuda() for n in sizes]
    gs = [(3.0 * torch.randn(n, generator=g)).cuda() for n in sizes]
    ms = [(0.1 * torch.randn(n, generator=g)).cuda() for n in sizes]
    vs = [(0.1 * torch.rand(n, generator=g)).cuda() for n in sizes]
    xs = [(0.1 * torch.rand(n, generator=g)).cuda() for n in sizes] if amsgrad else None
    return ps, gs, ms, vs, xs


def _step(static, sizes, tensors, max_norm, step0):
    from demiss_vae_amd import _lib as L
    lib = L.lib()
    ps, gs, ms, vs, xs = ([t.clone() for t in ts] if ts is not None else None for ts in tensors)
    n = len(sizes)
    numel = (C.c_int64 * n)(*sizes)
    group_of = (C.c_int32 * n)(*[i % 2 for i in range(n)])
    lr = [torch.tensor(1e-3, device="cuda"), torch.tensor(3e-4, device="cuda")]
    step = [torch.tensor(float(step0), device="cuda"), torch.tensor(float(step0), device="cuda")]
    groups = (L.DmvAdamGroup * 2)()
    for i in range(2):
        groups[i] = L.DmvAdamGroup(lr[i].data_ptr(), step[i].data_ptr(), 0.9, 0.999, 1e-8)
    norm = torch.zeros((), device="cuda")
    X_ = L.ptr_array(xs) if xs is not None else None
    if static:
        rc = lib.dmv_opt_adam_step_static(n, L.ptr_array(ps), L.ptr_array(gs), L.ptr_array(ms), L.ptr_array(vs), X_, numel,
                                          group_of, 2, groups, max_norm, L.ptr(norm), None, None, 0, L.stream_ptr())
    else:
        wsf = lib.dmv_opt_ws_floats(n, numel)
        ws = torch.empty(wsf, dtype=torch.float32, device="cuda")
        rc = lib.dmv_opt_adam_step(n, L.ptr_array(ps), L.ptr_array(gs), L.ptr_array(ms), L.ptr_array(vs), X_, numel,
                                   group_of, 2, groups, max_norm, L.ptr(ws), wsf, L.ptr(norm), None, None, 0, L.stream_ptr())
    assert rc == 0, rc
    torch.cuda.synchronize()
    return ps, ms, vs, xs, norm, step


def _same(a, b):
    pa, ma, va, xa, na, sa = a
    pb, mb, vb, xb, nb_, sb = b
    for ta, tb in zip(pa + ma + va + (xa or []) + [na] + sa, pb + mb + vb + (xb or []) + [nb_] + sb):
        assert torch.equal(ta, tb)


def test_static_workspace_adam_step_matches_workspace_entry_point():
    # first call: 7 tensors incl. ragged sizes (a tensor shorter than a chunk quarter, one that ends mid-quarter), AMSGrad
    big = [65536, 4096 * 3 + 17, 1, 300, 256 * 256, 1023, 5000]
    t1 = _problem(big, 1, amsgrad=True)
    _same(_step(True, big, t1, 5.0, 0), _step(False, big, t1, 5.0, 0))
    # second call, smaller: fewer tensors, fewer chunks, plain Adam, no clipping (the sums of squares of the first call
    # are still in the array and must not be read)
    small = [100, 4097]
    t2 = _problem(small, 2, amsgrad=False)
    _same(_step(True, small, t2, 0.0, 10), _step(False, small, t2, 0.0, 10))
    # and a clipped small step right after (partial sums beyond its own chunks are stale values of the first call)
    _same(_step(True, small, t2, 0.5, 3), _step(False, small, t2, 0.5, 3))


def test_static_workspace_rejects_sizes_beyond_its_array():
    from demiss_vae_amd import _lib as L
    lib = L.lib()
    n = 40
    sizes = [2_000_000] * n                      # 40 x 489 chunks x 4 quarters > 65536 floats
    numel = (C.c_int64 * n)(*sizes)
    assert lib.dmv_opt_ws_floats(n, numel) > 65536
    fake = (L._P * n)(*[256] * n)                # never dereferenced: the size check comes first
    group_of = (C.c_int32 * n)(*[0] * n)
    groups = (L.DmvAdamGroup * 1)()
    groups[0] = L.DmvAdamGroup(256, 256, 0.9, 0.999, 1e-8)
    rc = lib.dmv_opt_adam_step_static(n, fake, fake, fake, fake, None, numel, group_of, 1, groups, 1.0, None, None, None, 0, None)
    assert rc == -3                              # DMV_E_WORKSPACE
    assert b"workspace" in lib.dmv_strerror(rc).lower()


def test_static_entry_allocates_nothing_and_records_into_a_graph():
    """One call outside capture (it looks the array's address up), then the same call recorded with torch.cuda.graph and
    replayed twice: three steps in all, equal to three eager steps of the workspace entry point.  Device memory in
    use (torch's allocator and the driver's own count) is the same before and after the calls."""
    from demiss_vae_amd import _lib as L
    lib = L.lib()
    sizes = [4096 * 2 + 5, 777, 256 * 32]
    n = len(sizes)
    base = _problem(sizes, 5, amsgrad=True)
    numel = (C.c_int64 * n)(*sizes)
    group_of = (C.c_int32 * n)(*[i % 2 for i in range(n)])

    def state():
        ps, gs, ms, vs, xs = ([t.clone() for t in ts] for ts in base)
        lr = [torch.tensor(1e-3, device="cuda"), torch.tensor(3e-4, device="cuda")]
        step = [torch.zeros((), device="cuda"), torch.zeros((), device="cuda")]
        groups = (L.DmvAdamGroup * 2)()
        for i in range(2):
            groups[i] = L.DmvAdamGroup(lr[i].data_ptr(), step[i].data_ptr(), 0.9, 0.999, 1e-8)
        norm = torch.zeros((), device="cuda")
        tables = [L.ptr_array(ts) for ts in (ps, gs, ms, vs, xs)]
        return dict(ps=ps, gs=gs, ms=ms, vs=vs, xs=xs, lr=lr, step=step, groups=groups, norm=norm, tables=tables)

    def call_static(s, stream):
        P_, G_, M_, V_, X_ = s["tables"]
        return lib.dmv_opt_adam_step_static(n, P_, G_, M_, V_, X_, numel, group_of, 2, s["groups"], 2.0, L.ptr(s["norm"]),
                                            None, None, 0, stream)

    a = state()
    torch.cuda.synchronize()
    used_torch, free_drv = torch.cuda.memory_allocated(), torch.cuda.mem_get_info()[0]
    assert call_static(a, L.stream_ptr()) == 0                       # first call: outside any capture
    torch.cuda.synchronize()
    assert torch.cuda.memory_allocated() == used_torch and torch.cuda.mem_get_info()[0] == free_drv
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        assert call_static(a, L.stream_ptr()) == 0                   # recorded, not run
    graph.replay()
    graph.replay()
    torch.cuda.synchronize()

    b = state()
    wsf = lib.dmv_opt_ws_floats(n, numel)
    ws = torch.empty(wsf, dtype=torch.float32, device="cuda")
    P_, G_, M_, V_, X_ = b["tables"]
    for _ in range(3):
        assert lib.dmv_opt_adam_step(n, P_, G_, M_, V_, X_, numel, group_of, 2, b["groups"], 2.0, L.ptr(ws), wsf,
                                     L.ptr(b["norm"]), None, None, 0, L.stream_ptr()) == 0
    torch.cuda.synchronize()
    for k in ("ps", "ms", "vs", "xs", "step"):
        for ta, tb in zip(a[k], b[k]):
            assert torch.equal(ta, tb), k
    assert torch.equal(a["norm"], b["norm"]) and float(a["step"][0]) == 3.0


def test_static_entry_with_fragment_images_and_counter():
    """pack_map (the step also writes the new values into a network's MFMA fragment image) and counter (a device
    counter advanced by the update kernel) through the static entry point, against the workspace entry point."""
    import demiss_vae_amd as dmv
    from demiss_vae_amd import _lib as L
    from oracle.recipes import NetSpec, make_params
    lib = L.lib()
    ns = NetSpec(6, 5, 1, 40)
    pr = make_params(ns, 9)
    outs = []
    for static in (True, False):
        ws_ = [pr[k + ".weight"].clone().cuda() for k in ns.layer_names()]
        bs_ = [pr[k + ".bias"].clone().cuda() for k in ns.layer_names()]
        net = dmv.PackedNet(dmv.NetDims(6, 5, 1, 40), ws_, bs_)
        ps = net.params()
        n = len(ps)
        g = torch.Generator().manual_seed(3)
        gs = [torch.randn(p.shape, generator=g).cuda() for p in ps]
        ms, vs = [torch.zeros_like(p) for p in ps], [torch.zeros_like(p) for p in ps]
        entries = (L.DmvPackMap * n)(*net.pack_maps())
        pack_map = torch.frombuffer(bytearray(bytes(entries)), dtype=torch.uint8).cuda()
        numel = (C.c_int64 * n)(*[p.numel() for p in ps])
        group_of = (C.c_int32 * n)(*[0] * n)
        lr, step = torch.tensor(1e-2, device="cuda"), torch.zeros((), device="cuda")
        groups = (L.DmvAdamGroup * 1)()
        groups[0] = L.DmvAdamGroup(lr.data_ptr(), step.data_ptr(), 0.9, 0.999, 1e-8)
        counter = torch.tensor([41], dtype=torch.int64, device="cuda")
        args = (n, L.ptr_array(ps), L.ptr_array(gs), L.ptr_array(ms), L.ptr_array(vs), None, numel, group_of, 1, groups, 1.0)
        if static:
            rc = lib.dmv_opt_adam_step_static(*args, None, L.ptr(pack_map), L.ptr(counter), 7, L.stream_ptr())
        else:
            wsf = lib.dmv_opt_ws_floats(n, numel)
            ws = torch.empty(wsf, dtype=torch.float32, device="cuda")
            rc = lib.dmv_opt_adam_step(*args, L.ptr(ws), wsf, None, L.ptr(pack_map), L.ptr(counter), 7, L.stream_ptr())
        assert rc == 0, rc
        torch.cuda.synchronize()
        assert int(counter) == 48
        image = net._packed.clone()
        # the image the step maintained = the image a fresh packing of the new parameters gives
        net.invalidate()
        assert torch.equal(net.packed(), image)
        outs.append((ps, image))
    for ta, tb in zip(outs[0][0] + [outs[0][1]], outs[1][0] + [outs[1][1]]):
        assert torch.equal(ta, tb)
