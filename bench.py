#!/usr/bin/env python
"""Benchmark of the vgiwae hot path on MI355X.

  python bench.py --gpus N --steps K --warmup W

N > 1 runs one rank per GPU: either a launcher started the ranks (torch.distributed.run sets WORLD_SIZE; it must
equal N), or this process starts them itself as a child `python -m torch.distributed.run --nproc-per-node N bench.py ...`
before it touches the GPU, relays the child's one JSON line and exits with the child's code.

A step = one data-parallel training step of the UCI-gas MissIWAE model (K=5 importance samples, STL,
mask-concat encoder (16,32,2,256), decoder (16,16,2,256)): fused objective forward + backward over one
synthetic batch already resident in HBM, gradient all-reduce over RCCL when N > 1, clip-norm 5 and one
Adam(amsgrad) update (so the weight re-pack that a real training step needs is inside the timed region).
`value` = importance-weighted samples (batch x K) per second summed over all ranks; per-GPU batch is
fixed as N grows (weak scaling).

The JSON line also carries
  roofline      fp32-MFMA roofline of the dominant kernel (decoder MLP kernels), timed live with events
  cpu_baseline  the CPU oracle (stock PyTorch ops) timed on this host on a bounded sample of the workload
  extra         fwd+bwd-only rate, the yaml batch size (512), the DeMissVAE step and the LL-eval rate
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_FP32_MFMA_TFLOPS = 157.3   # MI355X fp32 matrix peak (dense)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--batch", type=int, default=32768, help="datapoints per GPU per step")
    ap.add_argument("--with-conv", action="store_true",
                    help="also time the image DeMissVAE step with convolutional trunks (MIOpen warm-up takes a while)")
    ap.add_argument("--no-extra", action="store_true", help="skip the secondary workloads")
    ap.add_argument("--torch-optim", action="store_true",
                    help="clip_grad_norm_ + torch.optim.Adam(fused) instead of the library's clip + Adam step")
    ap.add_argument("--no-cpu", action="store_true", help="skip the CPU baseline")
    ap.add_argument("--cpu-budget", type=float, default=12.0, help="seconds of CPU work for the main CPU baseline")
    ap.add_argument("--kernels-only", action="store_true",
                    help="only launch the decoder MLP kernels of the roofline record (for rocprofv3 runs)")
    return ap.parse_args()


def mlp_flops_per_row(spec):
    return 2 * (spec.in_dim * spec.hidden + spec.nb * 2 * spec.hidden * spec.hidden + spec.hidden * spec.out_dim)


def make_gas(device, B, seed=20220118):
    """UCI-gas MissIWAE shapes (configs/uci_gas/compare/mis50/iwae_i5_encm_stl.yaml) with synthetic data and
    nn.Linear-style random weights; product-side generators only (the oracle is used by the CPU baseline alone)."""
    import demiss_vae_amd as dmv
    from demiss_vae_amd import synth
    enc_s, dec_s = synth.NetShape(16, 32, 2, 256), synth.NetShape(16, 16, 2, 256)
    enc, dec = synth.packed_net(enc_s, seed, device), synth.packed_net(dec_s, seed + 1, device)
    dims = dmv.ModelDims(D=8, Zd=16, gen_dist="normal_with_eps", q_dist="normal_with_eps", encoder_use_mis_mask=True)
    X, M = synth.batch(seed + 2, B, 8)
    return dims, enc, dec, enc_s, dec_s, X.to(device), M.to(device)


def timed(fn, steps, warmup, barrier):
    for _ in range(warmup):
        fn()
    barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        fn()
    torch.cuda.synchronize()
    barrier()
    return (time.perf_counter() - t0) / steps


def event_time(fn, reps, warm=1):
    """Average device time of fn() in seconds, bracketed by events on the current stream, after `warm` untimed calls."""
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e-3 / reps


def shader_clock_ghz(fn, launches, watch_ms=20.0):
    """Shader clock held while fn() runs back to back: a one-wave watcher on a side stream stamps the shader-cycle
    counter and the constant-rate counter `watch_ms` apart (dmv_clock_watch) while `launches` calls of fn keep the
    chip busy on the current stream (not a timed pass: the watcher takes one wave slot)."""
    import ctypes as C
    from demiss_vae_amd import _lib as L
    lib = L.lib()
    khz = C.c_int32(0)
    L.check(lib.dmv_wall_clock_khz(C.byref(khz)))
    out = torch.zeros(4, dtype=torch.int64, device="cuda")
    side = torch.cuda.Stream()
    torch.cuda.synchronize()
    for _ in range(3):
        fn()
    L.check(lib.dmv_clock_watch(int(watch_ms * 1e-3 * khz.value * 1e3), L.ptr(out), C.c_void_p(side.cuda_stream)))
    for _ in range(launches):
        fn()
    torch.cuda.synchronize()
    c0, r0, c1, r1 = out.tolist()
    return (c1 - c0) / max(1, r1 - r0) * khz.value * 1e3 / 1e9


PROFILE_MD = "profiles/r3_decoder_kernels.md"   # written by tools/prof.sh + tools/summarize_prof.py from `bench.py --kernels-only`


def profiled_traffic(rows, kernel):
    """HBM bytes per launch of the named kernel, from the committed rocprofv3 PMC passes of `bench.py --kernels-only`
    (FETCH_SIZE / WRITE_SIZE in runs of their own, gfx950 unit correction applied by tools/summarize_prof.py).  Counters
    cannot be collected from inside this process, so the record is null unless the committed summary is of this very
    workload: the same row count (stated in its title) and a section for the kernel that is timed here."""
    import re
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), PROFILE_MD)
    try:
        text = open(path).read()
        if f"{rows} rows" not in text.splitlines()[0]:
            return {"traffic": None}
        pat = {"k_mlp_fwd": r"^### .*k_mlp_fwd<", "k_mlp_bwd": r"^### .*k_mlp_bwd<"}.get(kernel, r"^### .*k_mlp_wgrad")
        m0 = re.search(pat, text, re.M)
        sec = text[m0.start():]
        sec = sec[:sec.index("\n### ", 4)] if "\n### " in sec[4:] else sec
        m = re.search(r"HBM read bytes[^=]*=[^=]*= ([\d.,]+) MB; written = ([\d.,]+) MB", sec)
        trace = re.search(r"^\| [^|]*" + {"k_mlp_fwd": "k_mlp_fwd<", "k_mlp_bwd": "k_mlp_bwd<"}.get(kernel, "k_mlp_wgrad") +
                          r"[^|]*\| \d+ \| ([\d.]+) \| ([\d.]+) \|", text, re.M)
        made = re.search(r"^Collected: (.*)$", text, re.M)
        return {"traffic": sum(float(m.group(i).replace(",", "")) for i in (1, 2)) * 1e6, "traffic_unit": "bytes per launch",
                "traffic_source": f"{PROFILE_MD} (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes, {rows} rows)",
                "profile_avg_us": float(trace.group(1)) if trace else None,
                "profile_steady_avg_us": float(trace.group(2)) if trace else None,   # the timed launches (last 20)
                "profile_collected": made.group(1) if made else None}
    except Exception:
        return {"traffic": None}


def kernel_roofline(dec, dec_s, rows, device, watch_clock=True):
    """Times the three decoder MLP kernels on their own (forward, input-gradient chain, weight-gradient
    GEMM + split fold) and returns the roofline record of the slowest, plus all three."""
    import ctypes as C
    import demiss_vae_amd as dmv
    from demiss_vae_amd import _lib as L
    from demiss_vae_amd.mlp import mlp_forward, pad32
    Z = torch.randn(rows, dec_s.in_dim, device=device)
    y, saved = mlp_forward(dec, Z, dec_s.in_dim, rows, save=True)
    gy = torch.randn_like(y)
    gy[:, dec_s.out_dim:] = 0
    lib, st = L.lib(), L.stream_ptr()
    nbytes = lib.dmv_mlp_backward_ws_bytes(C.byref(dec.c), rows)
    ws = torch.empty((nbytes + 3) // 4, dtype=torch.float32, device=device)
    dx = torch.empty(rows, pad32(dec_s.in_dim), device=device)
    gws = [torch.empty_like(w) for w in dec.weights]
    gbs = [torch.empty_like(b) for b in dec.biases]
    packed = dec.packed()

    def fwd():
        L.check(lib.dmv_mlp_forward(C.byref(dec.c), L.ptr(packed), L.IN_PLAIN, dec_s.in_dim, L.ptr(Z), dec_s.in_dim, None, 0,
                                    rows, L.ptr(y), y.shape[1], L.ptr(saved), None, None, st))

    def chain():
        L.check(lib.dmv_mlp_backward(C.byref(dec.c), L.ptr(packed), L.ptr(saved), None, L.ptr(gy), gy.shape[1], rows,
                                     L.ptr(dx), dx.shape[1], None, None, 0.0, L.ptr(ws), nbytes, st))

    def chain_wgrad():
        L.check(lib.dmv_mlp_backward(C.byref(dec.c), L.ptr(packed), L.ptr(saved), None, L.ptr(gy), gy.shape[1], rows,
                                     L.ptr(dx), dx.shape[1], L.ptr_array(gws), L.ptr_array(gbs), 0.0, L.ptr(ws), nbytes, st))

    # 30 untimed launches first (about 25 ms): after the allocations above the chip needs some tens of milliseconds of
    # load before its clock settles (the first launches of a burst run 10 - 15 % longer than the later ones)
    reps, warm = 20, 30
    t_f, t_c, t_cw = event_time(fwd, reps, warm), event_time(chain, reps, warm), event_time(chain_wgrad, reps, warm)
    flops = rows * mlp_flops_per_row(dec_s)
    parts = {
        "k_mlp_fwd": {"ms": t_f * 1e3, "tflops": flops / t_f / 1e12},
        "k_mlp_bwd": {"ms": t_c * 1e3, "tflops": flops / t_c / 1e12},
        "k_mlp_wgrad+k_wgrad_reduce": {"ms": (t_cw - t_c) * 1e3, "tflops": flops / max(t_cw - t_c, 1e-9) / 1e12},
    }
    name = max(parts, key=lambda k: parts[k]["ms"])
    ach = parts[name]["tflops"]
    prof = profiled_traffic(rows, name)
    # the same fraction from the committed rocprofv3 kernel trace of this command (another box, another day: the
    # spread between the two is box to box and profiled against unprofiled)
    frac_prof = (round(flops / (prof["profile_steady_avg_us"] * 1e-6) / 1e12 / PEAK_FP32_MFMA_TFLOPS, 4)
                 if prof.get("profile_steady_avg_us") else None)
    frac_prof_all = (round(flops / (prof["profile_avg_us"] * 1e-6) / 1e12 / PEAK_FP32_MFMA_TFLOPS, 4)
                     if prof.get("profile_avg_us") else None)
    launches = max(30, int(0.03 / max(t_f, 1e-5)))      # >= 30 ms of back-to-back launches around a 20 ms watch
    # (not under the profiler: the watcher's wave takes a SIMD from the one-wave-per-SIMD kernels and stretches the
    # launches it overlaps, which would end up in the profile's averages)
    ghz = shader_clock_ghz({"k_mlp_fwd": fwd, "k_mlp_bwd": chain}.get(name, chain_wgrad), launches) if watch_clock else 0.0
    roof = {"bound": "mfma", "kernel": name, "achieved": round(ach, 2), "peak": PEAK_FP32_MFMA_TFLOPS,
            "unit": "TFLOP/s", "frac": round(ach / PEAK_FP32_MFMA_TFLOPS, 4), "frac_in_run": round(ach / PEAK_FP32_MFMA_TFLOPS, 4),
            "frac_from_profile": frac_prof, "frac_from_profile_all_launches": frac_prof_all, **prof,
            "flops_per_launch": flops, "ms_per_launch": round(parts[name]["ms"], 4),
            "shader_clock_ghz": round(ghz, 3), "peak_clock_ghz": 2.4,
            "frac_of_peak_at_observed_clock": round(ach / (PEAK_FP32_MFMA_TFLOPS * ghz / 2.4), 4) if ghz > 0.5 else None,
            "all_kernels": {k: {kk: round(vv, 3) for kk, vv in v.items()} for k, v in parts.items()}}
    return roof


def cpu_baseline(budget_s=15.0, threads=None, B=512):
    """The CPU oracle (stock PyTorch ops + autograd, the op sequence the reference issues) on the same workload
    (B datapoints, K=5), repeated for about `budget_s` seconds on `threads` of this host's cores (default: the box's
    CPU share, 16)."""
    from oracle import core
    from oracle.recipes import NetSpec, make_params, make_batch, make_noise
    threads = min(16, os.cpu_count() or 1) if threads is None else threads
    torch.set_num_threads(threads)
    enc_s, dec_s = NetSpec(16, 32, 2, 256), NetSpec(16, 16, 2, 256)
    spec = core.ModelSpec(D=8, Zd=16, enc=enc_s, dec=dec_s, encoder_use_mis_mask=True)
    ep = {k: v.requires_grad_(True) for k, v in make_params(enc_s, 1).items()}
    dp = {k: v.requires_grad_(True) for k, v in make_params(dec_s, 2).items()}
    K = 5
    X, M = make_batch(3, B, 8)

    def step():
        eps = torch.randn(1, K, B, 16)
        bound, _ = core.estimate_iwelbo(spec, ep, dp, X, M, eps, STL=True)
        (-bound).backward()
        for p in list(ep.values()) + list(dp.values()):
            p.grad = None

    for _ in range(3 if B <= 4096 else 1):
        step()
    n, t0 = 0, time.perf_counter()
    while time.perf_counter() - t0 < budget_s:
        step()
        n += 1
    dt = (time.perf_counter() - t0) / n
    return {"value": round(B * K / dt, 1), "unit": "samples/s", "cores": threads, "kind": "port",
            "sample": f"{n} fwd+bwd steps of B={B}, K=5 in {time.perf_counter() - t0:.0f} s; CPU oracle = stock PyTorch ops"}


def dbg(msg):
    if os.environ.get("DMV_BENCH_DEBUG"):
        print(f"[bench rank {os.environ.get('RANK', 0)}] {msg}", file=sys.stderr, flush=True)


def _claim_stdout():
    """The contract is ONE JSON line on stdout.  Libraries write there too (RCCL prints a version banner at
    communicator creation): from here on file descriptor 1 is stderr for everything, and the returned handle is the
    only way to the real stdout."""
    sys.stdout.flush()
    real = os.fdopen(os.dup(1), "w")
    os.dup2(2, 1)
    return real


def launch_ranks(args):
    """`--gpus N` without a launcher: N ranks as a child process (torch.distributed.run, one rank per GPU), started
    before this process has touched the GPU.  The child's rank 0 prints the JSON line on the inherited stdout; the
    child's exit code is returned."""
    import socket
    import subprocess
    with socket.socket() as sock:
        sock.bind(("127.0.0.1", 0))
        port = sock.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__), *sys.argv[1:]]
    return subprocess.call(cmd, env=env)


def main():
    args = parse()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(launch_ranks(args))
    out_stream = _claim_stdout()
    rank = int(os.environ.get("RANK", 0))
    world = int(os.environ.get("WORLD_SIZE", 1))
    local = int(os.environ.get("LOCAL_RANK", 0))
    if world != args.gpus:
        print(f"bench.py: --gpus {args.gpus} but the launcher started {world} rank(s)", file=sys.stderr, flush=True)
        sys.exit(2)
    dist = world > 1 or bool(os.environ.get("DMV_BENCH_FORCE_DIST"))   # the latter: one-rank rehearsal of the RCCL calls
    backend = None
    if dist:
        import torch.distributed as td
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        ndev = torch.cuda.device_count()
        backend = os.environ.get("DMV_BENCH_BACKEND", "nccl")   # "gloo" lets two ranks share one GPU in rehearsals
        if backend == "nccl" and world > ndev:
            print(f"bench.py: {world} ranks need {world} GPUs, this node shows {ndev} (RCCL wants one GPU per rank)",
                  file=sys.stderr, flush=True)
            sys.exit(2)
        torch.cuda.set_device(local % ndev)
        if backend == "nccl":
            td.init_process_group("nccl", device_id=torch.device("cuda", local % ndev))
        else:
            td.init_process_group(backend)
        local = local % ndev
    device = torch.device("cuda", local)
    torch.cuda.set_device(device)

    import demiss_vae_amd as dmv
    from demiss_vae_amd import rng
    from demiss_vae_amd.parallel import all_reduce, allreduce_grads, overlap_grad_exchange
    from demiss_vae_amd.run import _unit_gradient as unit_gradient
    rng.set_seed(20220118 + rank)

    B, K = args.batch, 5
    dims, enc, dec, enc_s, dec_s, X, M = make_gas(device, B, seed=20220118 + 1000 * rank)
    params = enc.params() + dec.params()
    # as the trainer sets the networks up: gradients in one permanent buffer per network; under torch.distributed the
    # decoder's exchange starts under the encoder's backward pass
    overlap_grad_exchange((enc, dec), force=dist)
    if args.kernels_only:
        print(json.dumps({"roofline": kernel_roofline(dec, dec_s, B * K, device, watch_clock=False)}), file=out_stream, flush=True)
        return
    groups = [{"params": dec.params(), "amsgrad": True, "lr": 1e-3}, {"params": enc.params(), "amsgrad": True, "lr": 1e-3}]
    if args.torch_optim:        # stock torch: clip_grad_norm_ + fused Adam (about a dozen launches)
        opt = torch.optim.Adam(groups, fused=True)
    else:                       # the library's clip + AMSGrad step (three launches; what the model classes use)
        from demiss_vae_amd.optim import Adam
        opt = Adam(groups).attach_images((enc, dec))        # as the trainer sets it up: the step keeps the MFMA images current

    def barrier():
        if dist:
            td.barrier()

    def max_over_ranks(seconds):
        if not dist:
            return seconds
        t = torch.tensor([seconds], device=device, dtype=torch.float64)
        all_reduce(t, td.ReduceOp.MAX)
        return t.item()

    def fwd_bwd():
        # the loss (minus the bound) comes out of the library with the bound, as in the model classes' training_step
        _, _, loss = dmv.estimate_iwelbo(dims, enc, dec, X, M, num_latent_samples=1, num_importance_samples=K, STL=True,
                                         with_loss=True)
        loss.backward(unit_gradient(loss))      # the trainer's cached root gradient: no ones-fill per step

    def step():
        opt.zero_grad(set_to_none=True)
        fwd_bwd()
        if dist:
            allreduce_grads(params, force=dist)
        if args.torch_optim:
            torch.nn.utils.clip_grad_norm_(params, 5.0, foreach=True)
        else:
            opt.clip_on_next_step(5.0)
        opt.step()

    dbg("model built")
    dt = max_over_ranks(timed(step, args.steps, args.warmup, barrier))
    dbg(f"timed {dt}")

    # K = 50 000 log-likelihood estimation: (a) test points dealt to the ranks, every rank all K samples of its points,
    # rates summed; (b) N > 1: the K samples of the SAME points split over the ranks, (max, sum-exp) states combined
    # with the logsumexp all-reduce, whole-job rate = points x K / slowest rank's time
    ll, ll_k = None, None
    if not args.no_extra:
        from demiss_vae_amd.bench_extra import loglik_eval_rate
        barrier()
        dbg("loglik start")
        ll = loglik_eval_rate(device)
        dbg("loglik done")
        if dist:
            t = torch.tensor([ll["samples_per_s"]], device=device, dtype=torch.float64)
            all_reduce(t, td.ReduceOp.SUM)
            ll["samples_per_s"] = round(t.item(), 1)
            ll["workload"] += f", {world} ranks each on its own test points"
        if world > 1:
            barrier()
            ll_k = loglik_eval_rate(device, k_shard=(rank, world), barrier=barrier)
            ll_k["ms_per_batch"] = round(max_over_ranks(ll_k["ms_per_batch"]), 3)
            ll_k["samples_per_s"] = round(50 * 50000 / (ll_k["ms_per_batch"] * 1e-3), 1)

    out = None
    if rank == 0:
        value = world * B * K / dt
        flops_per_sample = (mlp_flops_per_row(enc_s) + K * mlp_flops_per_row(dec_s)) * 3 / K
        out = {
            "metric": "IWAE fwd+bwd samples/sec (batch x K)", "value": round(value, 1), "unit": "samples/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(dt * 1e3, 4),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32",
            "data": "synthetic (seeded normal data, uniform 50 % masks, nn.Linear-style random weights; latent noise and "
                    "dropout from the library's device Philox streams: checked statistically, not against torch's streams)",
            "config": {"workload": f"UCI-gas MissIWAE K=5 STL training step (fwd+bwd+allreduce+clip+Adam), "
                                   f"B={B} per GPU, enc (16,32,2,256) mask-concat, dec (16,16,2,256), D=8, Zd=16",
                       "global_batch": world * B, "K": K, "parallelism": f"dp{world}"},
            "step_tflops": round(value * flops_per_sample / 1e12, 2),
            "step_frac_of_fp32_mfma_peak": round(value * flops_per_sample / 1e12 / (PEAK_FP32_MFMA_TFLOPS * world), 4),
        }
        if dist:
            out["rccl_ranks"] = td.get_world_size()
            out["collective_backend"] = "rccl" if backend == "nccl" else backend
        # rank 0's GPU, after the timed region (the other ranks wait at the closing barrier)
        out["roofline"] = kernel_roofline(dec, dec_s, B * K, device)
    if rank == 0 and world == 1 and not args.no_extra:
        extra = {}
        t_fb = timed(lambda: (opt.zero_grad(set_to_none=True), fwd_bwd()), max(5, args.steps // 2), 2, barrier)
        extra["fwd_bwd_only_samples_per_s"] = round(B * K / t_fb, 1)
        # the yaml batch size (configs/uci_gas/compare/mis50/iwae_i5_encm_stl.yaml: batch_size 512)
        d2 = make_gas(device, 512)
        dims2, enc2, dec2, X2, M2 = d2[0], d2[1], d2[2], d2[5], d2[6]
        p2 = enc2.params() + dec2.params()
        from demiss_vae_amd.optim import Adam as LibAdam
        opt2 = LibAdam(p2, lr=1e-3, amsgrad=True).attach_images((enc2, dec2))

        def step512():
            opt2.zero_grad(set_to_none=True)
            _, _, loss = dmv.estimate_iwelbo(dims2, enc2, dec2, X2, M2, num_latent_samples=1, num_importance_samples=K,
                                             STL=True, with_loss=True)
            loss.backward(unit_gradient(loss))      # the trainer's cached root gradient: no ones-fill per step
            opt2.clip_on_next_step(5.0)
            opt2.step()

        t512 = timed(step512, 50, 10, barrier)
        extra["yaml_batch_512"] = {"ms_per_step": round(t512 * 1e3, 4), "samples_per_s": round(512 * K / t512, 1)}
        # the same step recorded once in a HIP graph (library noise: fresh on every replay)
        g512 = dmv.GraphedStep(step512, nets=(enc2, dec2))
        tg = timed(g512, 200, 20, barrier)
        extra["yaml_batch_512_hip_graph"] = {"ms_per_step": round(tg * 1e3, 4), "samples_per_s": round(512 * K / tg, 1)}
        from demiss_vae_amd.bench_extra import demiss_toy_step, missiwae_step
        extra["demiss_toy_mvb2"] = demiss_toy_step(device)
        extra["demiss_uci_gas_lair"] = demiss_toy_step(device, B=512, K=5, uci=True)
        extra["hepmass_missiwae"] = missiwae_step(device, "UCI-hepmass", 4096, 21, 16, (42, 165, 2, 256), (16, 42, 2, 256))
        extra["miniboone_missiwae"] = missiwae_step(device, "UCI-miniboone", 8192, 43, 32, (86, 325, 5, 256),
                                                    (32, 86, 2, 256), dropout=0.5)
        # hidden 512 with 784-dimensional Bernoulli data: NOT a reference configuration (its image models are
        # convolutional); the shape the wide four-wave kernels are measured on
        from demiss_vae_amd.bench_extra import iwae_step
        wide = iwae_step(device, 4096, name="synthetic 784-d pair, hidden 512 (not a reference config)", D=784, Zd=50,
                         enc_shape=(784, 100, 2, 512), dec_shape=(50, 784, 2, 512), gen_dist="bern", mask_concat=False)
        wide["step_frac_of_fp32_mfma_peak"] = round(wide["step_tflops"] / PEAK_FP32_MFMA_TFLOPS, 4)
        extra["synthetic_784_h512"] = wide
        if args.with_conv:
            from demiss_vae_amd.bench_extra import mnist_demiss_step
            extra["mnist_demiss_mvb2"] = mnist_demiss_step(device)
        out["extra"] = extra
    if rank == 0:
        if ll is not None:
            out.setdefault("extra", {})["loglik_eval_K50000"] = ll
        if ll_k is not None:
            out["extra"]["loglik_eval_K50000_samples_sharded"] = ll_k
        if not args.no_cpu:
            # the headline workload itself (same B, a few steps), then the yaml batch on all cores and on one
            out["cpu_baseline"] = cpu_baseline(budget_s=args.cpu_budget, B=B)
            out.setdefault("extra", {})["cpu_baseline_yaml_batch"] = cpu_baseline(budget_s=args.cpu_budget * 0.6, B=512)
            out["extra"]["cpu_baseline_single_thread"] = cpu_baseline(budget_s=args.cpu_budget * 0.5, threads=1, B=512)
        print(json.dumps(out), file=out_stream, flush=True)
    if dist:
        td.barrier()
        td.destroy_process_group()


if __name__ == "__main__":
    main()
