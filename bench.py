#!/usr/bin/env python3
"""Headline benchmark: UNet denoise img-steps/sec at 64x64 latents on MI355X (BASELINE.json metric).

Default mode -- one "step" = one pass of the hot path over one batch: ``UNet.forward`` (SD-1.5-shaped, bf16 storage / fp32
accumulate) on 32 latents of 4x64x64 with a 77x768 text context, followed by ``DDPMScheduler.step`` on the result --
BASELINE.json configs[1], the configuration the metric is quoted on.  Inputs are resident in HBM before the timed region.
With --gpus N every rank runs the same per-GPU workload on its own batch shard (independent latents, no data-path
collective: weak scaling); value = images x steps over all ranks / max-over-ranks time.

Other modes (same launch contract, same JSON schema):
  --strong   BASELINE.json configs[2]: 256 latents of 4x64x64 sharded over the N ranks (256/N each, run in micro-batches
             of 32), K reverse steps of the sampling loop timed; "scaling": "strong" (total work fixed as N grows).
  --train    BASELINE.json configs[3]: the reference's train step (add_noise -> bf16 forward -> MSE -> backward with the
             bucketed RCCL gradient all-reduce -> clip -> AdamW), 32 latents per GPU; value in img/s; "scaling": "weak".

Passes (default and --strong modes): `value` / `ms_per_step` come from K steps with NO per-launch instrumentation; the
`roofline` object comes from a SECOND pass over the same K steps with two HIP events around every GEMM launch (that pass's
own ms_per_step is printed as config.ms_per_step_instrumented_pass); the LayerNorm-fold A/B (N=1) repeats both passes with
the fold switched the other way, so the two variants are timed with identical instrumentation.

Printed JSON (rank 0, one line) also carries
  roofline      the dominant kernel (bf16 implicit-GEMM conv/linear; in --train possibly the weight-gradient kernel):
                algorithmic FLOPs per launch / its average launch duration, measured with HIP events on the launch
                stream over the instrumented pass, against the dense bf16 MFMA peak; `traffic` = HBM bytes per launch from the
                rocprofv3 --pmc passes recorded in bench_pmc_traffic.json (tools/profile_bench.sh writes it), reported only
                while the kernel sources still hash to what was profiled -- else null;
  cpu_baseline  the CPU oracle (oracle/unet_ref.py + scheduler_ref.py: a port, not the product) timed on this
                host's cores on a bounded sample of the same workload (N=1, default mode only).
"""
from __future__ import annotations

import argparse
import hashlib
import json
import math
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(ROOT, "latent-diffusion_amd"))
sys.path.insert(0, ROOT)

MFMA_BF16_DENSE_PEAK = 2.5e15       # FLOP/s, /opt/skills/guides/MI355X_MICROARCH.md (dense; 2:1-sparse figure is 5e15)
FLOPS_PER_IMAGE_64 = 803273441280.0  # algorithmic forward FLOPs per latent at 64x64 (tests/golden/meta.json)
BATCH, LATENT, CTX_LEN, CTX_DIM = 32, 64, 77, 768
STRONG_LATENTS, STRONG_MICRO_BATCH = 256, 32


def init_weights_(net: torch.nn.Module, seed: int):
    """Random-init weights of the real architecture, drawn on the device (no checkpoints offline)."""
    g = torch.Generator(device="cuda").manual_seed(seed)
    with torch.no_grad():
        for name, p in net.named_parameters():
            if p.dim() >= 2:
                fan_in = p[0].numel()
                p.normal_(0.0, 1.0 / math.sqrt(fan_in), generator=g)
            elif name.endswith("weight"):
                p.fill_(1.0)
            else:
                p.normal_(0.0, 0.02, generator=g)


def cpu_baseline(sample_forwards: int = 12):
    """Oracle on the host cores: default UNet fp32 forward at B=1, 64x64 + one scheduler step (bounded sample)."""
    from oracle import scheduler_ref as sref
    from oracle import unet_ref as uref
    cfg = uref.UNetConfig()
    shapes = uref.param_shapes(cfg)
    g = torch.Generator().manual_seed(0)
    sd = {}
    for k, shp in shapes.items():  # cheap fill (values do not matter for timing)
        sd[k] = torch.empty(shp).normal_(0, 0.02, generator=g) if len(shp) > 1 else torch.ones(shp)
    x = torch.randn(1, 4, LATENT, LATENT, generator=g)
    ctx = torch.randn(1, CTX_LEN, CTX_DIM, generator=g)
    tabs = sref.build_tables()
    noise = torch.randn(1, 4, LATENT, LATENT, generator=g)
    with torch.no_grad():
        uref.unet_forward(sd, cfg, x, 500, ctx)  # warm-up
        t0 = time.perf_counter()
        for _ in range(sample_forwards):
            eps = uref.unet_forward(sd, cfg, x, 500, ctx)
            sref.step(tabs, eps, 500, 480, x, noise)
        dt = time.perf_counter() - t0
    return {"value": sample_forwards / dt, "unit": "img-steps/s", "cores": torch.get_num_threads(), "kind": "port",
            "sample": f"{sample_forwards} forwards of the default UNet (fp32, B=1, 4x64x64, ctx 77x768) + scheduler step "
                      f"after 1 warm-up, {dt:.1f} s; os.cpu_count()={os.cpu_count()}"}


def kernel_source_hash() -> str:
    """Hash of the kernel sources a PMC figure is valid for (bench_pmc_traffic.json carries the hash it was taken on)."""
    h = hashlib.sha256()
    d = os.path.join(ROOT, "latent-diffusion_amd", "csrc")
    for name in sorted(os.listdir(d)):
        if name.endswith((".hip", ".h", ".cpp")):
            h.update(open(os.path.join(d, name), "rb").read())
    return h.hexdigest()[:16]


def pmc_traffic(kernel: str, workload_key: str):
    """(bytes per launch, note) from the committed rocprofv3 --pmc passes, or (None, why)."""
    path = os.path.join(ROOT, "bench_pmc_traffic.json")
    try:
        rec = json.load(open(path))
    except (OSError, ValueError):
        return None, "no bench_pmc_traffic.json (run tools/profile_bench.sh on the GPU box)"
    if rec.get("source_hash") != kernel_source_hash():
        return None, f"kernel sources changed since the PMC passes of {rec.get('profile', '?')}"
    ent = rec.get("workloads", {}).get(workload_key, {}).get(kernel)
    if ent is None:
        return None, f"kernel not in the PMC passes of {rec.get('profile', '?')} for workload {workload_key}"
    return float(ent["hbm_bytes_per_launch"]), (f"FETCH_SIZE (doubled: gfx950 correction) + WRITE_SIZE per launch, separate rocprofv3 --pmc "
                                                f"passes, {rec.get('profile', '?')}; Infinity-Cache hits included (upper bound of DRAM bytes)")


def roofline_from_log(log, elapsed, workload_key):
    per = {}
    for name, flops, e0, e1, shape in log:
        d = per.setdefault(name, [0, 0.0, 0.0, 0.0])
        d[0] += 1
        d[1] += flops
        d[2] += e0.elapsed_time(e1) * 1e-3
        # algorithmic HBM bytes of the launch: the input activation once (not once per tap or per tile column), the filter,
        # the output, and the residual operand when the epilogue adds one (bf16; the GLU output is half as wide)
        m, n, k, ksize, stride, up, _c1, epi = shape
        cin = k // (ksize * ksize)
        in_pix = m * stride * stride / (4 if up else 1)
        nout = n // 2 if epi & 48 else n
        d[3] += 2.0 * (in_pix * cin + n * k + m * nout + (m * n if epi & 4 else 0))
    gemm_time = sum(d[2] for d in per.values())
    dom = max(per, key=lambda k: per[k][2])
    n_l, fl, tt, ab = per[dom]
    achieved = fl / tt
    traffic, note = pmc_traffic(dom, workload_key)
    roof = {"bound": "mfma", "kernel": dom, "achieved": round(achieved / 1e12, 1), "peak": MFMA_BF16_DENSE_PEAK / 1e12,
            "unit": "TFLOP/s", "frac": round(achieved / MFMA_BF16_DENSE_PEAK, 4), "traffic": traffic, "traffic_note": note,
            "launches": n_l, "avg_launch_us": round(1e6 * tt / n_l, 1),
            "algorithmic_gflop_per_launch": round(fl / n_l / 1e9, 2),
            "algorithmic_MB_per_launch": round(ab / n_l / 1e6, 1),
            "traffic_over_algorithmic_bytes": None if traffic is None else round(traffic / (ab / n_l), 2),
            "all_gemm_kernels": {k: {"launches": v[0], "tflops": round(v[1] / v[2] / 1e12, 1),
                                     "avg_us": round(1e6 * v[2] / v[0], 1)} for k, v in sorted(per.items())}}
    return roof, gemm_time / elapsed


def hbm_kernel_rows(dev, sched, ops, iters=24, warm=3):
    """Achieved HBM rate of the memory-bound kernels of the path at sizes where a rate means something (the step itself runs
    them at launch-bound sizes): scheduler step and add_noise on (256, 4, 64, 64) float32 latents -- BASELINE.json configs[2]'s
    whole job in one launch -- GroupNorm + SiLU and LayerNorm on the 64x64 level's activation (32 x 4096 x 320, bf16).  Every
    launch works on its own set of operand buffers and the sets of one row add up to more than 512 MiB, twice the 256 MiB
    Infinity Cache, so no launch finds its operands (or the lines its results go to) in a cache: these are DRAM rates.  HIP
    events on torch's current stream, the stream these kernels are launched on; algorithmic bytes = every operand read once +
    every result written once.  Outside every timed region."""
    def timeit(fns):
        for f in fns[:warm]:
            f()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        # These launches last 10-50 us, about what the host needs to issue one through Python: behind a ~2 ms spin kernel the
        # host runs ahead, the queue is full when the first event fires, and the launches run back to back (what is timed
        # is the GPU, not the issue rate of this loop).
        torch.cuda.synchronize(dev)
        try:
            torch.cuda._sleep(4_000_000)
        except Exception:
            pass
        e0.record()
        for i in range(iters):
            fns[i % len(fns)]()
        e1.record()
        torch.cuda.synchronize(dev)
        return e0.elapsed_time(e1) * 1e3 / iters       # microseconds per launch

    rows = []
    g = torch.Generator(device=dev).manual_seed(7)
    shape = (256, 4, 64, 64)
    nset = 12                                          # 12 sets x 3 inputs x 16.8 MB = 604 MB read per sweep
    sets = [tuple(torch.randn(shape, device=dev, generator=g) for _ in range(3)) for _ in range(nset)]
    n = sets[0][0].numel()
    t_mid = sched.timesteps[len(sched.timesteps) // 2]
    us = timeit([(lambda s=s: sched.step(s[0], t_mid, s[1], noise=s[2])) for s in sets])
    rows.append(("ddpm_step_kernel (eps, x_t, noise -> prev, x0)", list(shape), 5 * 4 * n, us, nset * 3 * 4 * n))
    tt = torch.randint(0, 1000, (shape[0],), device=dev, generator=g)
    us = timeit([(lambda s=s: sched.add_noise(s[1], s[2], tt)) for s in sets])
    rows.append(("ddpm_add_noise_kernel (x0, noise -> noisy)", list(shape), 3 * 4 * n, us, nset * 2 * 4 * n))
    del sets
    nact = 8                                           # 8 x 83.9 MB = 671 MB of distinct bf16 activations
    acts = [torch.randn(32, 64, 64, 320, device=dev, generator=g).bfloat16() for _ in range(nact)]
    gam, bet = torch.ones(320, device=dev), torch.zeros(320, device=dev)
    us = timeit([(lambda a=a: ops.groupnorm(a, gam, bet, 32, silu=True)) for a in acts])
    rows.append(("ldm_groupnorm + SiLU (one-kernel form at this shape: one read + one write)", list(acts[0].shape),
                 2 * 2 * acts[0].numel(), us, nact * 2 * acts[0].numel()))
    us = timeit([(lambda a=a: ops.layernorm(a.view(32, 4096, 320), gam, bet)) for a in acts])
    rows.append(("ldm_layernorm", [32, 4096, 320], 2 * 2 * acts[0].numel(), us, nact * 2 * acts[0].numel()))
    return [{"kernel": k, "shape": sh, "algorithmic_bytes": b, "us_per_launch": round(u, 2), "GB_per_s": round(b / u / 1e3, 1),
             "frac_of_hbm_peak": round(b / u / 1e3 / 8000.0, 3), "distinct_input_MB_rotated": round(rot / 1e6)}
            for k, sh, b, u, rot in rows]


def _unet_mod():
    from ldm import unet
    return unet


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=None, help="timed steps (default 10; 5 with --train)")
    ap.add_argument("--warmup", type=int, default=None, help="untimed warm-up steps (default 3; 2 with --train)")
    ap.add_argument("--batch", type=int, default=BATCH, help="latents per GPU (default: the metric's 32)")
    ap.add_argument("--latent", type=int, default=LATENT, help="latent extent (default 64: the metric's; 96 = BASELINE configs[4])")
    ap.add_argument("--strong", action="store_true", help="configs[2]: 256 latents sharded over the ranks, strong scaling")
    ap.add_argument("--train", action="store_true", help="configs[3]: the training step, 32 latents per GPU")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--plain-only", action="store_true",
                    help="profiler runs (tools/profile_bench.sh): only the un-instrumented pass -- no event-log pass, no A/B legs, no roofline")
    args = ap.parse_args()
    if args.strong and args.train:
        raise SystemExit("--strong and --train are exclusive")
    if args.steps is None:
        args.steps = 5 if args.train else 10
    if args.warmup is None:
        args.warmup = 2 if args.train else 3

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch with: python -m torch.distributed.run --nnodes=1 --nproc-per-node N "
                             "--master-addr 127.0.0.1 --master-port P bench.py --gpus N ...")
        raise SystemExit(f"--gpus {args.gpus} does not match WORLD_SIZE {world}")
    if os.environ.get("LDM_BENCH_SHARE_GPU") == "1":   # rehearsal only: every rank on device 0
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        # RCCL ("nccl" on ROCm) is the backend of record; LDM_BENCH_BACKEND=gloo with LDM_BENCH_SHARE_GPU=1 only
        # rehearses the launch contract with several ranks on a one-GPU box (RCCL refuses two ranks on one device).
        backend = os.environ.get("LDM_BENCH_BACKEND", "nccl")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(backend)

    from ldm import ops
    from ldm.scheduling_ddpm import DDPMScheduler
    from ldm.unet import UNet

    with torch.device("meta"):
        net = UNet()
    net = net.to_empty(device=dev)
    init_weights_(net, seed=1234)
    sched = DDPMScheduler(device=dev)

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize(dev)

    def max_over_ranks(elapsed):
        if dist is None:
            return elapsed
        tmax = torch.tensor([elapsed], device=dev if dist.get_backend() == "nccl" else "cpu", dtype=torch.float64)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        return float(tmax)

    LAT = args.latent
    if LAT == 64:
        flops_per_image = FLOPS_PER_IMAGE_64
    else:   # algorithmic FLOPs of another extent: the oracle's shape walk (host arithmetic only, nothing is executed)
        from oracle import unet_ref as uref
        flops_per_image = uref.flops_per_image(uref.UNetConfig(), LAT, LAT)
    gen = torch.Generator(device=dev).manual_seed(100 + rank)  # every rank owns different latents (batch shard)
    log = []
    out = None

    if args.train:
        # ------------------------------------------------------------------ configs[3]: the training step
        from ldm.optim import FusedAdamW, mse_loss
        from ldm.parallel import DataParallel
        net.train()
        B = args.batch
        model = DataParallel(net)
        opt = FusedAdamW(net.parameters(), lr=1e-5, betas=(0.9, 0.999), weight_decay=1e-2, eps=1e-8, max_grad_norm=1.0)
        lat = torch.randn(B, 4, LAT, LAT, device=dev, generator=gen)
        ctx = torch.randn(B, CTX_LEN, CTX_DIM, device=dev, generator=gen)

        def train_step():
            noise = torch.randn(lat.shape, device=dev, generator=gen)
            t = torch.randint(0, 1000, (B,), device=dev, generator=gen)
            noisy = sched.add_noise(lat, noise, t)                       # train_latent_diffusion.py:1062
            with torch.autocast("cuda", dtype=torch.bfloat16):
                pred = model(noisy, t, ctx)                              # :1105
            loss, dpred = mse_loss(pred.detach(), noise)                # :1108 and its gradient, one pass
            pred.backward(dpred)                                         # :1119 (+ bucketed all-reduce when N > 1)
            opt.step()                                                   # :1123-1124, clip folded into AdamW
            opt.zero_grad(set_to_none=True)
            return loss

        def timed_pass(instrument, warm):
            for _ in range(warm):
                loss = train_step()
            ops.set_launch_log(instrument)
            try:
                barrier()
                t0 = time.perf_counter()
                for _ in range(args.steps):
                    loss = train_step()
                barrier()
                dt = time.perf_counter() - t0
            finally:
                ops.set_launch_log(None)
            return dt, loss

        elapsed, loss = timed_pass(None, args.warmup)     # `value`: no per-launch instrumentation
        elapsed = max_over_ranks(elapsed)
        elapsed_log, _ = timed_pass(log, 1)               # `roofline`: the same steps with HIP events around every GEMM / wgrad launch
        assert torch.isfinite(loss), "non-finite loss"
        if rank == 0:
            roof, share = roofline_from_log(log, elapsed_log, f"train_b{B}_l{LAT}")
            roof["measured_in"] = ("a second pass over the same steps with two HIP events around every GEMM / weight-gradient "
                                   "launch; `value` is from the un-instrumented pass")
            flops = 3.0 * flops_per_image * B
            out = {
                "metric": f"UNet training img/s at {LAT}x{LAT} latents (noise-prediction MSE, bf16 compute / fp32 master, AdamW)",
                "value": round(world * B * args.steps / elapsed, 2), "unit": "img/s", "n_gpus": world, "steps": args.steps,
                "warmup": args.warmup, "ms_per_step": round(1e3 * elapsed / args.steps, 3), "higher_is_better": True,
                "scaling": "weak", "vs_baseline": None, "dtype": "bf16", "data": "synthetic",
                "config": {"workload": "SD-1.5-shaped UNet (859.5M params, random init) training step: add_noise, bf16 forward, MSE, "
                                       f"backward, gradient all-reduce (RCCL, 64 MiB buckets), clip 1.0, AdamW; batch {B} of 4x{LAT}x{LAT} "
                                       "per GPU (BASELINE.json configs[3])",
                           "per_gpu_batch": B, "global_batch": B * world, "latent": f"4x{LAT}x{LAT}",
                           "parallelism": f"data parallel x{world}, bucketed gradient all-reduce overlapped with the backward",
                           "collectives_per_backward": model.reducer.launched,
                           "frac_of_mfma_peak_at_3x_forward_flops": round(flops / (elapsed / args.steps) / MFMA_BF16_DENSE_PEAK, 4),
                           "ms_per_step_instrumented_pass": round(1e3 * elapsed_log / args.steps, 3),
                           "gemm_kernel_share_of_instrumented_step": round(share, 3),
                           "peak_device_memory_gb": round(torch.cuda.max_memory_allocated(dev) / 2 ** 30, 1),
                           "loss": round(float(loss), 5)},
                "roofline": roof}
    else:
        net.eval()
        sched.set_timesteps(50, device=dev)
        ts = sched.timesteps
        if args.strong:
            # ------------------------------------------ configs[2]: 256 latents sharded over the ranks, micro-batches of 32
            from ldm.sampling import shard_range
            lo, hi = shard_range(STRONG_LATENTS, rank, world)
            chunks = [(c, min(c + STRONG_MICRO_BATCH, hi)) for c in range(lo, hi, STRONG_MICRO_BATCH)]
            B = hi - lo
        else:
            B = args.batch
            chunks = [(0, B)]
        lat0 = torch.randn(B, 4, LAT, LAT, device=dev, generator=gen)
        ctx = torch.randn(B, CTX_LEN, CTX_DIM, device=dev, generator=gen).bfloat16()
        noise = torch.randn(B, 4, LAT, LAT, device=dev, generator=gen)
        base = chunks[0][0]

        def one_step(i):
            t = ts[i % len(ts)]
            last = None
            for c0, c1 in chunks:
                s = slice(c0 - base, c1 - base)
                eps = net(lat0[s], t, ctx[s])                 # bf16 UNet forward (autocast; conv_in reads the fp32 latents directly)
                last = sched.step(eps, t, lat0[s], noise=noise[s]).prev_sample  # fp32 scheduler update, fused kernel
            return last

        def timed_pass(instrument):
            """W warm-up steps, then K steps bracketed by barrier + synchronize.  instrument: a list that receives two HIP
            events per GEMM launch (ops.set_launch_log) -- the roofline pass; None -- the pass `value` is taken from."""
            nonlocal lat
            with torch.inference_mode(), torch.autocast("cuda", dtype=torch.bfloat16):
                for i in range(args.warmup):
                    lat = one_step(i)
                ops.set_launch_log(instrument)
                try:
                    barrier()
                    t0 = time.perf_counter()
                    for i in range(args.steps):
                        lat = one_step(i)
                    barrier()
                    dt = time.perf_counter() - t0
                finally:
                    ops.set_launch_log(None)
            return dt

        lat = None
        elapsed = timed_pass(None)                   # `value`: no per-launch instrumentation inside the timed region
        assert torch.isfinite(lat).all(), "non-finite latents"
        elapsed = max_over_ranks(elapsed)
        if args.plain_only:
            os.environ["LDM_BENCH_AB"] = "0"
            elapsed_log = float("nan")
        else:
            elapsed_log = timed_pass(log)            # `roofline`: the same steps once more with the event log on
        fold_on = bool(ops._ln_fold_checked) and all(ops._ln_fold_checked.values()) and _unet_mod()._LN_FOLD
        if rank == 0:
            ms = 1e3 * elapsed / args.steps
            total = STRONG_LATENTS if args.strong else world * B
            headline = LAT == 64 and ((not args.strong and B == BATCH) or args.strong)
            wkey = f"{'strong' if args.strong else 'fwd'}_b{min(B, STRONG_MICRO_BATCH) if args.strong else B}_l{LAT}"
            if args.plain_only:
                roof, share = None, float("nan")
            else:
                roof, share = roofline_from_log(log, elapsed_log, wkey)
                roof["measured_in"] = ("a second pass over the same steps with two HIP events around every GEMM launch; "
                                       "`value` is from the un-instrumented pass")
            if args.strong:
                workload = (f"SD-1.5-shaped UNet (859.5M params, random init) bf16 denoise + DDPM scheduler step over {STRONG_LATENTS} latents of "
                            f"4x{LAT}x{LAT} + 77x768 context, sharded {STRONG_LATENTS}/{world} per GPU in micro-batches of "
                            f"{STRONG_MICRO_BATCH} (BASELINE.json configs[2])")
            else:
                workload = ("SD-1.5-shaped UNet (859.5M params, random init) bf16 forward + DDPM scheduler step, "
                            f"batch {B} of 4x{LAT}x{LAT} latents + 77x768 context per GPU "
                            + ("(BASELINE.json configs[1])" if headline else "(non-headline shape)"))
            out = {
                "metric": f"UNet denoise img-steps/sec at {LAT}x{LAT} latents",
                "value": round(total * args.steps / elapsed, 2), "unit": "img-steps/s", "n_gpus": world, "steps": args.steps,
                "warmup": args.warmup, "ms_per_step": round(ms, 3), "higher_is_better": True,
                "scaling": "strong" if args.strong else "weak", "vs_baseline": None, "dtype": "bf16", "data": "synthetic",
                "config": {"workload": workload, "per_gpu_batch": B, "global_batch": total, "latent": f"4x{LAT}x{LAT}",
                           "parallelism": f"batch-sharded x{world}, no data-path collective",
                           "end_to_end_tflops_per_gpu": round(flops_per_image * total / world / (elapsed / args.steps) / 1e12, 1),
                           "end_to_end_frac_of_mfma_peak": round(flops_per_image * total / world / (elapsed / args.steps) / MFMA_BF16_DENSE_PEAK, 4),
                           "ms_per_step_instrumented_pass": round(1e3 * elapsed_log / args.steps, 3),
                           "gemm_kernel_share_of_instrumented_step": round(share, 3),
                           "layernorm_fold_active": fold_on},
                "roofline": roof}
            if args.plain_only:
                out["config"]["ms_per_step_instrumented_pass"] = out["config"]["gemm_kernel_share_of_instrumented_step"] = None
            if world == 1 and not args.no_cpu_baseline and LAT == 64 and not args.strong:
                out["cpu_baseline"] = cpu_baseline()
        side = world == 1 and not args.strong and out is not None and os.environ.get("LDM_BENCH_AB", "1") != "0"
        if side and LAT == 64:
            try:
                with torch.no_grad():
                    out["config"]["hbm_kernels"] = hbm_kernel_rows(dev, sched, ops)
            except Exception as exc:   # a side measurement never costs the headline line
                out["config"]["hbm_kernels"] = f"failed: {type(exc).__name__}: {exc}"
        # A/B inside the same process and on the same device (devices of the pool differ by up to 10 %): both passes once
        # more with the LayerNorm fold switched the other way -- same warm-up, same steps, same instrumentation per pass.
        if side and ops.ln_fold_supported(320, torch.bfloat16):
            um = _unet_mod()
            was = um._LN_FOLD
            ab = {}
            try:
                um._LN_FOLD = not fold_on
                net.invalidate_packed_weights()
                dt_plain = timed_pass(None)
                other_log = []
                dt_log = timed_pass(other_log)
                other_on = bool(ops._ln_fold_checked) and all(ops._ln_fold_checked.values()) and um._LN_FOLD
                roof_o, _ = roofline_from_log(other_log, dt_log, "ab")
                ab = {"fold": other_on, "ms_per_step": round(1e3 * dt_plain / args.steps, 3),
                      "ms_per_step_instrumented_pass": round(1e3 * dt_log / args.steps, 3),
                      "gemm_kernels": roof_o["all_gemm_kernels"]}
            except Exception as exc:
                ab = {"failed": f"{type(exc).__name__}: {exc}"}
            finally:
                um._LN_FOLD = was
                net.invalidate_packed_weights()
            out["config"]["layernorm_fold_ab"] = {
                "this_line": {"fold": fold_on, "ms_per_step": out["ms_per_step"],
                              "ms_per_step_instrumented_pass": out["config"]["ms_per_step_instrumented_pass"]},
                "other": ab}
    if rank == 0:
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
