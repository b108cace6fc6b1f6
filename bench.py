#!/usr/bin/env python
"""bench.py — denoised frames/sec (UNet sampling loop + VAE decode), BASELINE.json's metric.

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

A "step" is ONE whole video through the hot path behind the reference's entry points:
`TextToVideoSynthesis.infer_conditioned` = Txt2VideoSampler.sample_loop (50 DDIM_Gaussian steps,
classifier-free guidance 9, eta 0; cond+uncond batched => 50 b=2 UNet forwards + 50 fused
update kernels) + batched VAE decode of all frames + uint8 conversion, all on device
(inputs — noise, conditioning, weights — are resident in HBM before the timed region).
Workload at N=1 (and N=2, one video per CFG pair): BASELINE.json configs[1] — ModelScope t2v fp16, 24 frames @256x256.
N>=4 (even): configs[2] — ONE 125-frame video, frames sharded along T over N/2 GPUs x the CFG pair (exchanges inside the
library over RCCL); the replicas layout (every GPU its own 24-frame video) is timed beside it and reported as `replicas`.
Weights are random-init of the exact ModelScope architecture (no checkpoints offline).

Prints ONE JSON line (rank 0) with the driver's fields plus
  roofline     — dominant kernel = the MFMA implicit-GEMM family (conv3x3 / temporal conv /
                 linear): algorithmic FLOPs of its launches in one UNet step / their summed
                 durations, measured live with HIP events on the launch stream
  cpu_baseline — the oracle port (oracle/torch_port.py, fp32, torch CPU) timed on the host
                 cores on a bounded sample of the same workload, extrapolated to frames/s
"""
import argparse
import json
import os
import sys
import time

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

MFMA_PEAK_TFLOPS = 2500.0      # dense fp16/bf16 MFMA peak, /opt/skills/guides/MI355X_MICROARCH.md
UNET_TFLOP_PER_FRAME = 0.3057  # algorithmic 2*MAC per frame per forward @256x256
VAE_TFLOP_PER_FRAME = 0.622


def random_weights_(module, seed):
    g = torch.Generator(device="cuda").manual_seed(seed)
    with torch.no_grad():
        for n, p in module.named_parameters():
            if p.ndim >= 2:
                p.normal_(0, 1.0 / p[0].numel() ** 0.5, generator=g)
            elif n.endswith("weight"):
                p.normal_(1.0, 0.1, generator=g)
            else:
                p.normal_(0.0, 0.05, generator=g)


def _usable_cores():
    """Host cores this process may really use: affinity mask, capped by the cgroup CPU quota."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        with open("/sys/fs/cgroup/cpu.max") as f:
            quota, period = f.read().split()
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except Exception:
        pass
    return n


def cpu_baseline_worker(frames, ddim_steps):
    """Runs in a child process (bounded by a timeout in the parent): the oracle port
    (oracle/torch_port.py, fp32) on the host cores — ONE UNet forward (b=1, 8 frames @256x256 = the C1 geometry,
    2.44 TFLOP: long enough that streaming the 5.6 GB of fp32 weights no longer dominates) and ONE VAE frame decode
    (0.62 TFLOP).  frames/s for the whole workload is extrapolated as 1 / (2*steps*t_unet_per_frame + t_vae_frame): UNet
    cost is linear in F.  Weights are seeded synthetic (timing does not depend on their values)."""
    from oracle import configs, synth, torch_port as tp
    from sd_webui_text2video_amd import unet as U, vae as V
    cores = _usable_cores()
    threads = min(cores, 64)          # torch CPU ops stop scaling (and can thrash) far beyond this
    torch.set_num_threads(threads)
    cfg, ddcfg = configs.MODELSCOPE_UNET, configs.VAE_DDCONFIG
    t0 = time.time()
    spec = synth.param_spec(U.UNetSD(**cfg, init_weights=False))
    g = torch.Generator().manual_seed(0)
    sd = {}
    for n, shp in spec:
        if len(shp) > 1:
            fan_in = 1
            for d in shp[1:]:
                fan_in *= d
            sd[n] = torch.empty(shp).normal_(0, fan_in ** -0.5)
        else:
            sd[n] = torch.ones(shp) if n.endswith("weight") else torch.zeros(shp)
    t_w = time.time() - t0
    fs = 8
    x = torch.randn(1, 4, fs, 32, 32, generator=g)
    y = torch.randn(1, 77, 1024, generator=g)
    with torch.no_grad():
        t0 = time.time()
        tp.unet_forward(sd, cfg, x, torch.tensor([500]), y)
        t_unet = time.time() - t0
    del sd
    vspec = synth.param_spec(V.AutoencoderKL(ddcfg, 4, init_weights=False))
    vsd = synth.synth_state_dict(vspec, seed=3)
    z = torch.randn(1, 4, 32, 32, generator=g)
    with torch.no_grad():
        t0 = time.time()
        tp.vae_decode(vsd, ddcfg, z)
        t_vae = time.time() - t0
    per_frame = 2 * ddim_steps * (t_unet / fs) + t_vae
    print(json.dumps({"value": round(1.0 / per_frame, 5), "unit": "frames/s", "cores": threads, "kind": "port",
                      "sample": f"oracle/torch_port.py fp32, {threads} torch threads ({cores} usable host cores): 1 UNet forward "
                                f"b=1 {fs}f@256x256 = {t_unet:.2f}s + 1 VAE frame decode = {t_vae:.2f}s (weights built in "
                                f"{t_w:.0f}s, untimed); extrapolated to {frames}f x {ddim_steps} steps x 2 (CFG) + decode"}))


def cpu_baseline(frames, ddim_steps, timeout_s=240):
    import subprocess
    try:
        out = subprocess.run([sys.executable, os.path.abspath(__file__), "--cpu-baseline-worker", "--frames", str(frames),
                              "--ddim-steps", str(ddim_steps)], capture_output=True, text=True, timeout=timeout_s,
                             env={**os.environ, "HIP_VISIBLE_DEVICES": "", "WORLD_SIZE": "1"})
        for line in reversed(out.stdout.strip().splitlines()):
            if line.startswith("{"):
                return json.loads(line)
        return {"value": None, "unit": "frames/s", "cores": _usable_cores(), "kind": "port",
                "sample": "cpu baseline worker failed: " + out.stderr[-300:]}
    except subprocess.TimeoutExpired:
        return {"value": None, "unit": "frames/s", "cores": _usable_cores(), "kind": "port",
                "sample": f"cpu baseline worker exceeded {timeout_s}s"}


def pmc_traffic():
    """HBM bytes per launch of the GEMM family from the committed PMC passes (collected with
    tools/gpu_profile.sh -> tools/pmc_post.py; counters cannot be read from inside this process)."""
    here = os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles")
    path = os.path.join(here, "r02_pmc_traffic.json")
    if not os.path.exists(path):
        path = os.path.join(here, "r01_pmc_traffic.json")
    try:
        with open(path) as fh:
            return round(json.load(fh)["hbm_bytes_per_launch"])
    except (OSError, KeyError, ValueError):
        return None


def gemm_algorithmic_bytes(op) -> int:
    """Unique operand + result bytes of one GEMM op: activations once (the 9 / 3 conv taps re-read the same
    rows), weights once, fp32 residual once, output once; split-K adds its fp32 slabs (written + read)."""
    i = op.i
    M, N, K, gather, epi, split = i[0], i[1], i[2], i[7], i[16], max(i[19], 1)
    if gather == 0:
        a = M * K * 2
    elif gather == 2:
        a = M * i[10] * 2
    else:
        a = (M * max(i[11], 1) ** 2 // (4 if i[12] else 1)) * i[10] * 2
    n_out = N // 2 if epi == 1 else N
    out = M * n_out * (4 if i[17] == 1 else 2)
    res = M * n_out * 4 if op.p[4].space != "null" else 0
    slabs = 2 * split * M * N * 4 if split > 1 else 0
    return a + N * K * 2 + out + res + slabs


def gemm_strict_bytes(op) -> int:
    """The floor any fp16 design pays: fp16 activations once + weights once + an fp16 result (no fp32 residual stream,
    no split-K slabs) — what VERDICT r01 asks the PMC traffic to be compared with."""
    i = op.i
    M, N, K, gather, epi = i[0], i[1], i[2], i[7], i[16]
    if gather == 0:
        a = M * K * 2
    elif gather == 2:
        a = M * i[10] * 2
    else:
        a = (M * max(i[11], 1) ** 2 // (4 if i[12] else 1)) * i[10] * 2
    return a + N * K * 2 + M * (N // 2 if epi == 1 else N) * 2


def choose_layout(world: int, requested: str, frames_arg: int = 0):
    """-> (layout, frames per video).  auto: one GPU = configs[1] (24 frames, cond + uncond batched); 2 GPUs = one 24-frame
    video per CFG pair; an even world >= 4 = configs[2], ONE 125-frame video, frames sharded along T over world / 2 GPUs x the
    CFG pair (north_star's layout); an odd world > 1 = one video per GPU.  `--frames` overrides the frame count only."""
    mode = requested
    if mode == "auto":
        mode = "single" if world == 1 else ("pairs" if world == 2 else ("tshard" if world % 2 == 0 else "replicas"))
    if world == 1:
        mode = "single"
    return mode, (frames_arg or (125 if mode == "tshard" else 24))


BASELINE_CONFIGS = {(24, 256, 256): "BASELINE.json configs[1]", (125, 256, 256): "BASELINE.json configs[2]",
                    (24, 576, 1024): "BASELINE.json configs[3]", (8, 256, 256): "BASELINE.json configs[0] geometry"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2, help="timed videos")
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--frames", type=int, default=0,
                    help="frames per video; default 24 (configs[1]) for N <= 2 and the replicas layout, 125 (configs[2]) for the "
                         "T-sharded layout of N >= 4")
    ap.add_argument("--height", type=int, default=256)
    ap.add_argument("--width", type=int, default=256)
    ap.add_argument("--ddim-steps", type=int, default=50)
    ap.add_argument("--videos", type=int, default=1,
                    help="independent videos per batch and GPU (default 1 = the reference's one-video pipeline; the UNet step "
                         "becomes one b=2*videos forward)")
    ap.add_argument("--also-batched", type=int, default=4,
                    help="at N=1 with --videos 1: also time one pass with this many videos per batch and report it as "
                         "`batched` beside the headline (0 / 1 = skip)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--parallel", default="auto", choices=["auto", "replicas", "pairs", "tshard"],
                    help="N>1 layout; auto = one video per CFG pair at N=2, ONE T-sharded video (frames over N/2 GPUs x CFG pair) "
                         "for even N>=4, one video per GPU otherwise")
    ap.add_argument("--cpu-baseline-worker", action="store_true", help=argparse.SUPPRESS)
    args = ap.parse_args()
    if args.cpu_baseline_worker:
        cpu_baseline_worker(args.frames, args.ddim_steps)
        return

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    ctl = None
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        # T2V_BENCH_ONE_DEVICE=1 (rehearsal of the N > 1 code path on a 1-GPU box): every rank on cuda:0, gloo instead of
        # RCCL (device buffers staged through the host, parallel.all_gather_into) — never a measurement configuration
        one_device = os.environ.get("T2V_BENCH_ONE_DEVICE") == "1"
        dist.init_process_group(backend="gloo" if one_device else "nccl")      # "nccl" is RCCL on ROCm
        ctl = dist.new_group(backend="gloo")         # control plane (layout agreement), never on the data path
        if one_device:
            local_rank = 0
    assert world == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE {world}"
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)

    from sd_webui_text2video_amd import configs
    from sd_webui_text2video_amd import parallel, pipeline, unet as U, vae as V

    cfg, ddcfg = configs.MODELSCOPE_UNET, configs.VAE_DDCONFIG
    net = U.UNetSD(**cfg, init_weights=False).half().to(dev).eval()
    random_weights_(net, 0)
    ae = V.AutoencoderKL(ddcfg, 4, init_weights=False).half().to(dev).eval()
    random_weights_(ae, 3)
    pipe = pipeline.TextToVideoSynthesis(sd_model=net, autoencoder=ae, device=dev)
    pipe.diffusion.progress = False
    g = torch.Generator().manual_seed(1)
    cond = torch.randn(1, 77, 1024, generator=g).half().to(dev)
    uncond = torch.randn(1, 77, 1024, generator=g).half().to(dev)

    requested = args.parallel
    mode, frames = choose_layout(world, requested, args.frames)

    def build(mode_, frames_, videos=args.videos):
        return parallel.make_runner(pipe, world, rank, frames=frames_, height=args.height, width=args.width,
                                    ddim_steps=args.ddim_steps, guidance=9.0, mode="pairs" if mode_ == "single" else mode_,
                                    videos=videos)

    def sync():
        torch.cuda.synchronize(dev)
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize(dev)

    def all_ok(ok: bool) -> bool:
        """Every rank learns whether ANY rank failed (gloo control group): the layout decision is collective."""
        if world == 1:
            return ok
        flag = torch.tensor([0 if ok else 1], dtype=torch.int32)
        dist.all_reduce(flag, group=ctl)
        return int(flag.item()) == 0

    def timed(runner_, n_warm, n_steps):
        for i in range(n_warm):
            runner_(cond, uncond, 1234 + i)
        sync()
        t0 = time.perf_counter()
        out_ = None
        for i in range(n_steps):
            out_ = runner_(cond, uncond, 1234 + i)
        sync()
        el = time.perf_counter() - t0
        if world > 1:
            tmax = torch.tensor([el], device=dev, dtype=torch.float64)
            dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
            el = float(tmax.item())
        assert out_ is not None and out_.dtype == torch.uint8
        return el

    fallback = None
    runner = None
    err = ""
    try:
        runner = build(mode, frames)
        if os.environ.get("T2V_BENCH_INJECT_FAILURE") in ("all", str(rank)):   # rehearsal hook: exercise the collective fallback
            raise RuntimeError("injected failure (T2V_BENCH_INJECT_FAILURE)")
        runner(cond, uncond, 999)                  # first pass: lowering, weight packing, communicator set-up
        sync()
        ok = True
    except Exception as exc:                       # noqa: BLE001 — reported, never silent (below)
        ok, err = False, f"{type(exc).__name__}: {exc}"
    if not all_ok(ok):
        # An explicitly requested layout that breaks fails the run.  Under `auto` every rank switches TOGETHER to the
        # collective-free layout and the JSON line says so (requested vs. actual layout + the first error seen here).
        # (This covers failures every rank sees at the same point — a missing library, an unsupported shape.  A failure
        # on SOME ranks while the others already wait inside a collective cannot be agreed on: the job then ends with
        # the process-group timeout, an explicit failure, never a silent change of what is measured.)
        if requested != "auto" or world == 1:
            raise RuntimeError(f"layout {mode!r} failed on rank {rank}: {err or 'another rank failed'}")
        fallback = {"requested_layout": mode, "reason": err or "failure on another rank"}
        print(f"[bench] rank {rank}: layout {mode!r} failed ({fallback['reason']}); all ranks switch to replicas", file=sys.stderr, flush=True)
        net.t_shard = None
        mode, frames = "replicas", args.frames or 24
        runner = build(mode, frames)
    elapsed = timed(runner, args.warmup, args.steps)

    total_frames = runner.frames_per_video_all_ranks * args.steps
    value = total_frames / elapsed
    ms_per_step = elapsed / args.steps * 1e3
    geom = (frames, args.height, args.width)
    named = BASELINE_CONFIGS.get(geom, "not a BASELINE.json configuration") if args.ddim_steps == 50 else \
        f"{BASELINE_CONFIGS.get(geom, 'custom geometry')} with {args.ddim_steps} instead of 50 steps"

    result = {
        "metric": f"denoised frames/sec (UNet+VAE), ModelScope {frames}f@{args.width}x{args.height}",
        "value": round(value, 4), "unit": "frames/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(ms_per_step, 2), "higher_is_better": True,
        "scaling": "weak" if mode in ("single", "replicas") else "strong", "vs_baseline": None,
        "dtype": "f16", "data": "synthetic",
        "config": {"workload": f"ModelScope t2v fp16 (random-init 1.41B UNetSD + VAE decoder), {frames} frames @ "
                               f"{args.width}x{args.height}, {args.ddim_steps} DDIM_Gaussian steps, CFG 9.0 "
                               f"({named}); one step = one whole video" + ("" if args.videos == 1 else f" x {args.videos} per batch"),
                   "frames_per_video": runner.frames_per_video_all_ranks, "videos_per_batch": args.videos,
                   "layout": mode, "layout_requested": requested, "parallelism": runner.describe},
    }
    if fallback is not None:
        result["config"]["layout_fallback"] = fallback
    if world > 1 and os.environ.get("T2V_BENCH_ONE_DEVICE") == "1":
        result["data"] = "synthetic; REHEARSAL: all ranks on one GPU over gloo — not a measurement"
    if world > 1 and mode != "replicas":
        # the collective-free layout beside the headline: every GPU its own 24-frame video (configs[1] per GPU)
        try:
            rep = build("replicas", 24, videos=1)
            el = timed(rep, 1, 1)
            result["replicas"] = {"value": round(rep.frames_per_video_all_ranks / el, 4), "unit": "frames/s", "ms_per_step": round(el * 1e3, 2),
                                  "note": "one independent 24-frame video per GPU (no data-path collective); not the headline"}
        except Exception as exc:                   # noqa: BLE001
            result["replicas"] = {"value": None, "note": f"failed: {type(exc).__name__}: {exc}"}

    if rank == 0:
        # ---- live roofline of the dominant kernel (HIP events on the launch stream) -------------
        F_loc = runner.unet_frames
        net.t_shard = None
        x = torch.randn(runner.unet_batch, 4, F_loc, args.height // 8, args.width // 8, device=dev)
        nv = max(1, runner.unet_batch // 2)
        y = torch.cat([cond.expand(nv, -1, -1), uncond.expand(nv, -1, -1)], 0)[: runner.unet_batch].contiguous()
        t = torch.full((runner.unet_batch,), 500, device=dev)
        net.forward_timed(x, t, y)
        _, ms, prog = net.forward_timed(x, t, y)
        gemm_ms = sum(m for op, m in zip(prog.ops, ms) if op.kind == 1)
        gemm_fl = sum(op.flops for op in prog.ops if op.kind == 1)
        n_gemm = sum(1 for op in prog.ops if op.kind == 1)
        alg_bytes = sum(gemm_algorithmic_bytes(op) for op in prog.ops if op.kind == 1)
        strict_bytes = sum(gemm_strict_bytes(op) for op in prog.ops if op.kind == 1)
        px = (args.height // 8) * (args.width // 8) / 1024.0
        video_tflop = (2 * args.ddim_steps * UNET_TFLOP_PER_FRAME + VAE_TFLOP_PER_FRAME) * px * runner.frames_per_video_all_ranks
        traffic = pmc_traffic() if (args.videos, frames, args.height, args.width, world) == (1, 24, 256, 256, 1) else None
        step_ms = sum(ms)
        achieved = gemm_fl / (gemm_ms * 1e-3) / 1e12
        result["roofline"] = {
            "bound": "mfma", "kernel": "gemm2_kernel<WM,WN,TM,TN,BK,STAGES,MINW,GATHER,PP> + gemm_kernel<BM,BN,WM,WN,GATHER> (one implicit-GEMM family: conv3x3 / temporal conv / linear)",
            "achieved": round(achieved, 1), "peak": MFMA_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": round(achieved / MFMA_PEAK_TFLOPS, 4),
            # the committed PMC passes are of the b=2, 24-frame, 256x256 step
            "traffic": traffic,
            "launches_per_unet_step": n_gemm, "avg_launch_us": round(gemm_ms / n_gemm * 1e3, 2),
            "flops_per_unet_step_T": round(gemm_fl / 1e12, 3),
            "unet_step_ms_events": round(step_ms, 3),
            "unet_step_tflops_all_kernels": round(prog.total_flops() / (step_ms * 1e-3) / 1e12, 1),
            "unet_step_frac_of_peak": round(prog.total_flops() / (step_ms * 1e-3) / 1e12 / MFMA_PEAK_TFLOPS, 4),
            "traffic_unit": "HBM bytes per launch (FETCH_SIZE x2 + WRITE_SIZE, rocprofv3 --pmc, profiles/r0N_pmc_traffic.json)",
            "algorithmic_bytes_per_launch": round(alg_bytes / n_gemm),
            "strict_bytes_per_launch": round(strict_bytes / n_gemm),
            "traffic_over_strict": round(traffic / (strict_bytes / n_gemm), 3) if traffic else None,
            "bytes_note": "algorithmic = this design (fp32 residual reads / fp32 stream outputs / split-K slabs included); strict = fp16 "
                          "activations once + weights once + fp16 result",
            # the WHOLE job against the matrix peak: SURVEY §8(d) FLOPs of every UNet forward and VAE frame / wall time / N GPUs
            "whole_video": {"tflop": round(video_tflop, 1), "tflops_per_gpu": round(video_tflop / (ms_per_step * 1e-3) / world, 1),
                            "frac": round(video_tflop / (ms_per_step * 1e-3) / world / MFMA_PEAK_TFLOPS, 4)},
        }
        if world == 1 and args.videos == 1 and args.also_batched > 1:
            # Reported beside the headline, never as `value`: the same workload with several videos per batch (the
            # reference's batch_count loop in one pass) — one warm-up pass, one timed pass.
            nv = args.also_batched
            r2 = parallel.make_runner(pipe, 1, 0, frames=frames, height=args.height, width=args.width,
                                      ddim_steps=args.ddim_steps, guidance=9.0, videos=nv)
            r2(cond, uncond, 4321)
            torch.cuda.synchronize(dev)
            tb = time.perf_counter()
            r2(cond, uncond, 4322)
            torch.cuda.synchronize(dev)
            tb = time.perf_counter() - tb
            result["batched"] = {"videos_per_batch": nv, "value": round(nv * frames / tb, 4), "unit": "frames/s",
                                 "ms_per_step": round(tb * 1e3, 2), "note": "same workload, several videos per batch; not the headline"}
        if not args.no_cpu_baseline and world == 1:       # the CPU leg is reported at N=1 only
            result["cpu_baseline"] = cpu_baseline(frames, args.ddim_steps)
        print(json.dumps(result), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
