"""Headline benchmark: images/sec of the ViT-B/16 224 bf16 classification forward on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--batch B] [--config vit_b16]

A "step" is one forward of the hot path (patch embedding -> 12 blocks -> final norm -> head) over one batch
of synthetic images already resident in HBM.  With N > 1 every rank (one per GPU) runs its own replica on its
own batch shard -- inference needs no collective -- and the reported value is all images of all ranks divided
by the slowest rank's time (weak scaling).  ``python bench.py --gpus N`` outside a torch.distributed.run launch
starts the N ranks itself (a child ``python -m torch.distributed.run``, before this process touches a GPU) and
relays rank 0's line and the exit code; a launch whose rank count differs from --gpus exits non-zero.

Timing: W warm-up steps (continued until the clocks have had >= 2 s of back-to-back forwards), then rounds of
EXACTLY K steps, each bracketed by barrier + synchronize, until >= 1 s has been timed; the line reports the
median round (``rounds`` lists how many, with the fastest and slowest).

Rank 0 prints ONE JSON line.  ``roofline`` prices the dominant kernel, the LayerNorm-folded fc1 GEMM with the
GELU epilogue (M = B*197, N = 3072, K = 768), by its algorithmic FLOPs per launch over its average launch
duration, measured with events on the launch stream in an instrumented pass after the timed region.
``cpu_baseline`` times the fp32 CPU oracle (a restatement of the reference's PyTorch forward) on a bounded
sample of the same workload.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for _p in (ROOT, os.path.join(ROOT, "vision-transformer_amd"), os.path.join(ROOT, "tests", "golden")):
    if _p not in sys.path:
        sys.path.insert(0, _p)

import torch  # noqa: E402
import torch.nn as nn  # noqa: E402

import weights as W  # noqa: E402

PEAK_BF16_TFLOPS = 2500.0   # dense bf16 MFMA, MI355X
PEAK_HBM_GBS = 8000.0
ALGO_BYTES_PER_IMAGE = {"vit_b16": 79.9e6, "vit_s16": 39.9e6, "vit_b16_384": 234e6}   # per-op-fused bf16 intermediates


def csrc_digest():
    """sha256 over the kernel sources the loaded library is built from (file names and contents, sorted): the identity
    the committed counter files are checked against.  Same function in profiles/summarize.py."""
    import glob
    import hashlib
    h = hashlib.sha256()
    d = os.path.join(ROOT, "vision-transformer_amd", "csrc")
    for path in sorted(glob.glob(os.path.join(d, "*.hip")) + glob.glob(os.path.join(d, "*.h")) + glob.glob(os.path.join(d, "*.cpp"))):
        h.update(os.path.basename(path).encode())
        with open(path, "rb") as f:
            h.update(f.read())
    return h.hexdigest()[:16]


def committed_counters(kind, config, batch):
    """(value, note) from the newest profiles/r*_<kind>_traffic.json that was collected for this workload: the PMC
    figure when that file was collected on these very kernel sources, else (None, why) -- a counter reading of other
    kernels is not a measurement of these."""
    import glob
    key = {"fc1": "hbm_bytes_per_launch", "forward": "hbm_bytes_per_image"}[kind]
    note = f"no committed rocprofv3 --pmc collection for {config} at batch {batch}"
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", f"r*_{kind}_traffic.json")), reverse=True):
        try:
            with open(path) as f:
                t = json.load(f)
        except (OSError, ValueError):
            continue
        if t.get("config") != config or t.get("batch_per_gpu") != batch:
            continue
        name = os.path.basename(path)
        if t.get("csrc_sha256") != csrc_digest():
            return None, (f"{name} was collected on other kernel sources (csrc {t.get('csrc_sha256')}, commit "
                          f"{t.get('commit')}; this tree: csrc {csrc_digest()}): stale, not reported")
        return t.get(key), f"{name}, commit {t.get('commit')}, csrc {t.get('csrc_sha256')}"
    return None, note


def build_model(cfg, device, seed=0):
    from vitm_amd.models import VisionTransformer, VITC
    LN = nn.LayerNorm
    vt = VisionTransformer(cfg["image_size"], patch_size=cfg["patch_size"], num_classes=cfg["num_classes"],
                           num_heads=cfg["num_heads"], mlp_ratio=cfg["mlp_ratio"], depth=cfg["depth"],
                           embed_dim=cfg["embed_dim"], norm_layer=LN,
                           embed_norm_layer=LN if cfg["embed_norm"] else None,
                           final_norm_layer=LN if cfg["final_norm"] else None)
    m = VITC(vt).eval()
    sd = {k: torch.from_numpy(v) for k, v in W.make_state_dict(cfg, seed).items()}
    m.load_state_dict(sd, strict=True)
    return m.to(device).to(torch.bfloat16), sd



def flops_per_image(shp: dict) -> float:
    """Algorithmic FLOPs (MAC = 2) of one classification forward; LayerNorm / softmax / GELU excluded:
    2 (N-1) K_pe D + L (2 N D 3D + 4 N^2 D + 2 N D^2 + 4 N D hid) + 2 D C."""
    N, D, L = shp["tokens"], shp["embed_dim"], shp["depth"]
    hid, C, kpe = shp["hidden"], shp.get("num_classes", 0), shp["k_pe"]
    return (2.0 * (N - 1) * kpe * D + L * (2.0 * N * D * 3 * D + 4.0 * N * N * D + 2.0 * N * D * D + 4.0 * N * D * hid)
            + 2.0 * D * C)

def cpu_baseline(cfg, sd, budget_s=15.0, batch=8):
    """fp32 CPU oracle on `batch` images per pass, as many passes as fit the time budget (>= 2).  The reference's own
    CPU-runnable case (ViT-tiny on CIFAR-sized images) is timed at its batch of 64."""
    from oracle import vit_oracle as O
    x = torch.from_numpy(W.make_input(cfg, batch, 7))
    ocfg = W.oracle_cfg(cfg)
    with torch.no_grad():
        O.vitc_forward(x, sd, ocfg)  # warm-up
        t0 = time.perf_counter()
        n = 0
        while True:
            O.vitc_forward(x, sd, ocfg)
            n += 1
            dt = time.perf_counter() - t0
            if (dt >= budget_s and n >= 2) or n >= 50:
                break
    return dict(value=round(n * batch / dt, 2), unit="images/sec", cores=torch.get_num_threads(), kind="port", batch=batch,
                sample=f"{n} passes of batch {batch}, fp32 CPU oracle (torch {torch.__version__}), "
                       f"{dt:.1f} s wall; same model and image size as the GPU workload")


def gemm_kernel_time(model, x, steps, which="fc1"):
    """Average duration (ms) of one GEMM of the block over `steps` forwards -- "fc1": the LN + GELU launches, "fc2":
    the residual launches whose K is the hidden width -- by events on the launch stream around each such launch.  An
    event pair with nothing between is recorded right behind each launch; its average interval (what two records cost
    by themselves) is returned beside the raw average."""
    from vitm_amd import _ops
    orig = _ops.gemm
    pairs, nulls = [], []
    hidden = model.vit_model.blocks[0].mlp.hidden_features

    def timed_gemm(A, pk, out, epilogue, M, **kw):
        hit = epilogue == _ops.EPI_LN_GELU if which == "fc1" else (epilogue == _ops.EPI_RESIDUAL and pk.K == hidden)
        if not hit:
            return orig(A, pk, out, epilogue, M, **kw)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        r = orig(A, pk, out, epilogue, M, **kw)
        e1.record()
        pairs.append((e0, e1))
        n0, n1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        n0.record()
        n1.record()
        nulls.append((n0, n1))
        return r

    _ops.gemm = timed_gemm
    try:
        with torch.no_grad():
            for _ in range(steps):
                model(x)
        torch.cuda.synchronize()
    finally:
        _ops.gemm = orig
    ms = [a.elapsed_time(b) for a, b in pairs]
    null_ms = [a.elapsed_time(b) for a, b in nulls]
    return sum(ms) / len(ms), len(ms), sum(null_ms) / len(null_ms)


def train_bench(args, world, rank, dist, dev):
    """Informational: images/sec of the training step -- fp32 (the reference trains in fp32) or, with --dtype bf16,
    mixed precision -- data-parallel over the ranks with the gradient all-reduce on RCCL.  Not the headline metric."""
    from vitm_amd import training
    from vitm_amd.sharding import max_over_ranks
    cfg = W.CONFIGS[args.config]
    shp = W.shapes(cfg)
    model, _ = build_model(cfg, dev)
    model = model.float().train()
    opt, sched = training.reference_optimizer_and_scheduler(model, 100000)
    bf16 = args.dtype == "bf16"
    tr = training.Trainer(model, opt, sched, max_norm=1.0, compute_dtype=torch.bfloat16 if bf16 else torch.float32)
    g = torch.Generator(device="cpu").manual_seed(4321 + rank)
    x = (torch.rand((args.batch, 3, cfg["image_size"], cfg["image_size"]), generator=g) * 2 - 1).to(dev)
    y = torch.randint(0, cfg["num_classes"], (args.batch,), generator=g).to(dev)

    def barrier():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    step = tr.step
    if args.graph:       # the step replayed from a HIP graph (single process): two eager steps, one capture, then replays
        from vitm_amd import GraphedTrainStep
        if world > 1:
            raise SystemExit("--graph with --mode train is single-process (the gradient all-reduce is not captured)")
        step = GraphedTrainStep(tr, warmup=2).step
        args.warmup = max(args.warmup, 4)
    for _ in range(args.warmup):
        step(x, y)
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        loss = step(x, y)
    barrier()
    elapsed = max_over_ranks(time.perf_counter() - t0, dist, dev)
    assert torch.isfinite(loss).all()
    if rank == 0:
        value = world * args.batch * args.steps / elapsed
        tf = value / world * 3.0 * flops_per_image(shp) / 1e12      # backward = 2x forward
        peak = 2500.0 if bf16 else 157.0
        print(json.dumps({
            "metric": f"images/sec, {args.config} {'bf16 mixed-precision' if bf16 else 'fp32'} training step", "value": round(value, 1), "unit": "images/sec",
            "n_gpus": world, "ranks_seen": dist.get_world_size() if dist is not None else 1,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(elapsed / args.steps * 1e3, 3),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "bf16" if bf16 else "f32", "data": "synthetic",
            "config": {"workload": f"{args.config}: forward + cross-entropy + backward + gradient all-reduce + clip + AdamW",
                       "batch_per_gpu": args.batch, "global_batch": args.batch * world,
                       "parallelism": f"data parallel x{world}, one flat-gradient all-reduce (RCCL) per step",
                       "launch": "hip-graph replay" if args.graph else "eager (one host call per kernel)"},
            "roofline": {"bound": "mfma", "achieved": round(tf, 1), "peak": peak, "unit": "TFLOP/s",
                         "frac": round(tf / peak, 4), "traffic": None,
                         "kernel": f"whole step against the dense {args.dtype} MFMA peak (3 x forward FLOPs)"},
            "final_loss": round(float(loss), 4)}), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


def self_launch(n: int, argv) -> int:
    """Start n ranks of this script under torch.distributed.run (child process; nothing here has touched a GPU),
    relay rank 0's output, return the child's exit code.  The reference's own multi-process entry spawns its
    ranks the same way (vitm/models/sv22b.py:260-315)."""
    import socket
    import subprocess
    with socket.socket() as so:
        so.bind(("127.0.0.1", 0))
        port = so.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + list(argv)
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    return subprocess.call(cmd, env=env)


def gather_floats(value: float, dist, dev):
    """Every rank's scalar, on every rank (a list of world-size floats)."""
    if dist is None:
        return [float(value)]
    t = torch.tensor([value], dtype=torch.float64, device=dev)
    out = [torch.zeros_like(t) for _ in range(dist.get_world_size())]
    dist.all_gather(out, t)
    return [float(o.item()) for o in out]


def hipblaslt_calibration(dev, n=4096, iters=20):
    """TFLOP/s of torch.matmul (hipBLASLt) on an n^3 bf16 problem, random data, on this box: the practical MFMA
    ceiling next to which the roofline fraction is also reported.  Outside the timed region."""
    a = torch.randn(n, n, device=dev).to(torch.bfloat16)
    b = torch.randn(n, n, device=dev).to(torch.bfloat16)
    for _ in range(5):
        a @ b
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        a @ b
    e1.record()
    torch.cuda.synchronize()
    return 2.0 * n ** 3 / (e0.elapsed_time(e1) / iters * 1e-3) / 1e12


def host_fed_rate(model, x, dev, args):
    """Images/s when the images start in HOST memory (never the headline ``value``): 4 micro-batches of the benchmark
    batch per pass through vitm_amd.sharding.HostFedInference (upload of micro-batch i + 1 under the compute of i), for
    pinned fp32, pinned bf16 and pageable fp32 sources, next to the one-shot cost of an un-overlapped upload."""
    import torch
    from vitm_amd.sharding import HostFedInference
    feeder = HostFedInference(model, dev, micro_batch=args.batch)
    n_mb = 4
    host32 = x.float().cpu().repeat(n_mb, 1, 1, 1)
    res = {"micro_batch": args.batch, "images_per_pass": host32.shape[0],
           "note": "uploads overlapped with compute on a second stream; not the headline value"}
    for tag, src in (("pinned_f32", host32.pin_memory()), ("pinned_bf16", host32.to(torch.bfloat16).pin_memory()),
                     ("pageable_f32", host32)):
        feeder.run(src)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        passes = 3
        for _ in range(passes):
            out = feeder.run(src)
        torch.cuda.synchronize()
        res[tag + "_images_per_sec"] = round(passes * src.shape[0] / (time.perf_counter() - t0), 1)
    # un-overlapped: upload everything, then compute
    src = host32.to(torch.bfloat16).pin_memory()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    xd = src.to(dev, non_blocking=True)
    torch.cuda.synchronize()
    t_up = time.perf_counter() - t0
    res["upload_GBps_pinned"] = round(src.numel() * 2 / t_up / 1e9, 1)
    with torch.no_grad():
        for i in range(n_mb):
            out = model(xd[i * args.batch:(i + 1) * args.batch])
    torch.cuda.synchronize()
    res["serial_pinned_bf16_images_per_sec"] = round(src.shape[0] / (time.perf_counter() - t0), 1)
    return res


def selftest_launch(args, world, rank, dist):
    """Launcher rehearsal without a GPU (gloo): the same rank start-up, barriers, max-over-ranks and JSON path
    around a no-op step.  Used by the CPU tests."""
    def barrier():
        if dist is not None:
            dist.barrier()
    for _ in range(args.warmup):
        time.sleep(0.001)
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        time.sleep(0.001)
    barrier()
    el = time.perf_counter() - t0
    per_rank = gather_floats(args.batch * args.steps / el, dist, None)
    from vitm_amd.sharding import max_over_ranks
    el = max_over_ranks(el, dist, None)
    ranks_seen = dist.get_world_size() if dist is not None else 1
    if rank == 0:
        print(json.dumps({"metric": "launcher self-test (no GPU work)", "value": round(world * args.batch * args.steps / el, 1),
                          "unit": "items/sec", "n_gpus": world, "ranks_seen": ranks_seen, "steps": args.steps,
                          "warmup": args.warmup, "per_rank": [round(v, 1) for v in per_rank]}), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    return 0 if ranks_seen == args.gpus else 3


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=30)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--batch", type=int, default=256, help="images per GPU per step")
    ap.add_argument("--config", default="vit_b16")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--warm-seconds", type=float, default=2.0, help="minimum warm-up time before the timed region")
    ap.add_argument("--min-timed-seconds", type=float, default=1.0, help="repeat the K-step round until this much is timed")
    ap.add_argument("--graph", action="store_true", help="replay the forward from a HIP graph (same kernels, one host call)")
    ap.add_argument("--mode", default="forward", choices=["forward", "train"],
                    help="train: one training step per step (forward, loss, backward, all-reduce, clip, AdamW)")
    ap.add_argument("--dtype", default=None, choices=["bf16", "f32"],
                    help="training arithmetic (default f32, the reference's); the forward benchmark is bf16")
    ap.add_argument("--force-dist", action="store_true",
                    help="create the nccl process group even for one rank: rehearses the barrier / gather / max-over-ranks "
                         "path on RCCL with the one GPU at hand")
    ap.add_argument("--from-host", action="store_true",
                    help="also report the rate with the images starting in host memory (line['host_fed'])")
    ap.add_argument("--selftest-launch", action="store_true",
                    help="rehearse the multi-rank launch on CPU (gloo) around a no-op step; no GPU is touched")
    args = ap.parse_args()
    if os.environ.get("VITM_BENCH_WATCHDOG"):   # seconds; a run still alive then prints every thread's stack and exits
        import faulthandler
        faulthandler.dump_traceback_later(float(os.environ["VITM_BENCH_WATCHDOG"]), exit=True)

    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        # not under a launcher: start the ranks ourselves, before anything here initialises a GPU
        sys.exit(self_launch(args.gpus, sys.argv[1:]))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: refusing to report an {args.gpus}-GPU number "
                         f"from {world} rank(s)")
    dist = None
    if args.selftest_launch:
        if world > 1:
            import torch.distributed as dist
            dist.init_process_group(backend="gloo")
        sys.exit(selftest_launch(args, world, rank, dist))
    if world > 1 or args.force_dist:
        import torch.distributed as dist
        torch.cuda.set_device(local_rank)
        if "MASTER_ADDR" not in os.environ:      # --force-dist outside a launcher
            os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=os.environ.get("MASTER_PORT", "29531"), RANK="0",
                              WORLD_SIZE="1")
        dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))
        if dist.get_world_size() != args.gpus:
            raise SystemExit(f"process group has {dist.get_world_size()} ranks, --gpus {args.gpus}")
    dev = torch.device("cuda", local_rank)
    torch.cuda.set_device(dev)

    if args.mode == "train":
        args.dtype = args.dtype or "f32"
        return train_bench(args, world, rank, dist, dev)
    cfg = W.CONFIGS[args.config]
    shp = W.shapes(cfg)
    gflop_img = flops_per_image(shp) / 1e9

    model, sd = build_model(cfg, dev)
    g = torch.Generator(device="cpu").manual_seed(1234 + rank)
    x = (torch.rand((args.batch, 3, cfg["image_size"], cfg["image_size"]), generator=g) * 2 - 1).to(dev).to(torch.bfloat16)

    def barrier():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    fwd = model
    if args.graph:
        from vitm_amd import GraphedForward
        fwd = GraphedForward(model, x)
    from vitm_amd.sharding import max_over_ranks
    with torch.no_grad():
        for _ in range(args.warmup):
            fwd(x)
        torch.cuda.synchronize()
        # keep warming until the chip has seen WARM_S seconds of back-to-back forwards (its clock settles over
        # seconds under an MFMA-dense load); the count is agreed across ranks
        t0 = time.perf_counter()
        for _ in range(3):
            fwd(x)
        torch.cuda.synchronize()
        per_fwd = max_over_ranks((time.perf_counter() - t0) / 3, dist, dev)
        extra_warm = max(0, min(2000, int(args.warm_seconds / per_fwd) - 3))
        for _ in range(extra_warm):
            fwd(x)
        # rounds of exactly --steps forwards, each bracketed by barrier + synchronize, until >= --min-timed-seconds
        n_rounds = max(1, min(200, int(args.min_timed_seconds / (per_fwd * args.steps) + 0.999)))
        rounds = []
        for _ in range(n_rounds):
            barrier()
            t0 = time.perf_counter()
            for _ in range(args.steps):
                out = fwd(x)
            barrier()
            rounds.append(time.perf_counter() - t0)
    assert torch.isfinite(out.float()).all()

    # per round the slowest rank's time; the line reports the median round
    own_rounds = rounds
    rounds = [max_over_ranks(r, dist, dev) for r in own_rounds]
    order = sorted(range(len(rounds)), key=lambda i: rounds[i])
    med = order[len(order) // 2]
    elapsed = rounds[med]
    ranks_seen = dist.get_world_size() if dist is not None else 1
    per_rank = gather_floats(args.batch * args.steps / own_rounds[med], dist, dev)   # each rank's own clock, median round

    line = None
    if rank == 0:
        ms_step = elapsed / args.steps * 1e3
        value = world * args.batch * args.steps / elapsed
        fc1_ms, n_launch, null_ms = gemm_kernel_time(model, x, max(2, min(args.steps, 10)))
        M, N, K = args.batch * shp["tokens"], shp["hidden"], shp["embed_dim"]
        fc1_tflops = 2.0 * M * N * K / (fc1_ms * 1e-3) / 1e12
        model_tflops = value / world * gflop_img / 1e3
        fc1_traffic, fc1_note = committed_counters("fc1", args.config, args.batch)
        from vitm_amd import _lib
        big_tile = bool(_lib.load().vitm_gemm_uses_big_tile(M, (N + 7) // 8 * 8, (K + 63) // 64 * 64))
        line = {
            "metric": "images/sec, ViT-B/16 224px bf16 classification forward" if args.config == "vit_b16"
                      else f"images/sec, {args.config} bf16 classification forward",
            "value": round(value, 1), "unit": "images/sec", "n_gpus": world, "ranks_seen": ranks_seen,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms_step, 3), "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "bf16", "data": "synthetic",
            "config": {"workload": f"{args.config}: image {cfg['image_size']}, patch {cfg['patch_size']}, "
                                   f"{shp['tokens']} tokens, dim {shp['embed_dim']}, depth {shp['depth']}, "
                                   f"{shp['num_classes']} classes; classification forward (VITC)",
                       "batch_per_gpu": args.batch, "global_batch": args.batch * world,
                       "parallelism": f"batch-sharded replicas x{world}, no collective",
                       "launch": "hip-graph replay" if args.graph else "eager (one host call per kernel)",
                       "gflop_per_image": round(gflop_img, 3)},
            "roofline": {"bound": "mfma", "achieved": round(fc1_tflops, 1), "peak": PEAK_BF16_TFLOPS,
                         "unit": "TFLOP/s", "frac": round(fc1_tflops / PEAK_BF16_TFLOPS, 4),
                         "traffic": fc1_traffic, "traffic_source": fc1_note,
                         "kernel": ("gemm_pp[_shared]_kernel<EPI_LN_GELU> (fc1: LayerNorm-folded 256x256 ping-pong GEMM + GELU; whole tiles and any leftover-tile slices in one launch)"
                                    if big_tile else "gemm_kernel<EPI_LN_GELU> (fc1: LayerNorm-folded 128x128-tile GEMM + GELU; this shape is too small for the 256x256 kernel)"),
                         "flops_per_launch": 2.0 * M * N * K, "avg_launch_ms": round(fc1_ms, 4),
                         "launches_timed": n_launch,
                         # what two event records cost with nothing between them, measured right behind each launch:
                         # the part of avg_launch_ms that is not the kernel (achieved / frac do NOT subtract it)
                         "event_pair_overhead_ms": round(null_ms, 4),
                         "avg_launch_ms_net": round(fc1_ms - null_ms, 4)},
            "whole_forward": {"tflops_per_gpu": round(model_tflops, 1),
                              "frac_of_bf16_peak": round(model_tflops / PEAK_BF16_TFLOPS, 4)},
            "rounds": {"timed": len(rounds), "steps_each": args.steps, "reported": "median",
                       "ms_per_step_min": round(min(rounds) / args.steps * 1e3, 3),
                       "ms_per_step_max": round(max(rounds) / args.steps * 1e3, 3),
                       "warmup_forwards": args.warmup + 3 + extra_warm},
            "per_rank_images_per_sec": [round(v, 1) for v in per_rank],
        }
        if args.config == "vit_s16":
            # embed dim 384: 6 K-tiles per output tile and 230 FLOP per algorithmic byte over the forward -- under the
            # machine balance of ~310, so this workload is priced against HBM.  Its largest kernel is the fc2 GEMM (residual
            # epilogue, K = hidden): algorithmic bytes = A (M x hidden) + W + residual in + output, all bf16.
            fc2_ms, n2, null2 = gemm_kernel_time(model, x, max(2, min(args.steps, 10)), which="fc2")
            D, hid = shp["embed_dim"], shp["hidden"]
            fc2_bytes = 2.0 * (M * hid + D * hid + 2 * M * D)
            gbs = fc2_bytes / (fc2_ms * 1e-3) / 1e9
            line["roofline_mfma_fc1"] = line["roofline"]
            line["roofline"] = {"bound": "hbm", "achieved": round(gbs, 1), "peak": PEAK_HBM_GBS, "unit": "GB/s",
                                "frac": round(gbs / PEAK_HBM_GBS, 4), "traffic": None,
                                "kernel": "gemm_pp_kernel<EPI_RESIDUAL, moments, K deep> (fc2: 256x256 ping-pong GEMM, bias + LayerScale + residual, row moments)",
                                "bytes_per_launch": fc2_bytes, "avg_launch_ms": round(fc2_ms, 4), "launches_timed": n2,
                                "event_pair_overhead_ms": round(null2, 4),
                                "mfma_tflops_same_kernel": round(2.0 * M * D * hid / (fc2_ms * 1e-3) / 1e12, 1)}
        blaslt = hipblaslt_calibration(dev)
        mfma_roof = line.get("roofline_mfma_fc1", line["roofline"])
        mfma_roof["hipblaslt_4096_tflops"] = round(blaslt, 1)
        mfma_roof["frac_of_hipblaslt_4096"] = round(fc1_tflops / blaslt, 4)
        line["whole_forward"]["frac_of_hipblaslt_4096"] = round(model_tflops / blaslt, 4)
        wf, wf_note = committed_counters("forward", args.config, args.batch)
        line["whole_forward"]["hbm_bytes_per_image_measured"] = wf
        line["whole_forward"]["hbm_bytes_per_image_source"] = wf_note
        line["whole_forward"]["hbm_bytes_per_image_algorithmic"] = ALGO_BYTES_PER_IMAGE.get(args.config)
        if args.from_host:
            line["host_fed"] = host_fed_rate(model, x, dev, args)
        if world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(cfg, sd, batch=64 if args.config == "vit_tiny_cifar" else 8)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    if line is not None:
        print(json.dumps(line), flush=True)


if __name__ == "__main__":
    main()
