"""world_size-2 gloo rehearsal of the multi-GPU path on the CPU: each rank takes its contiguous batch shard and runs
the forward on it with no data-path collective; the concatenation of the shards' outputs equals the unsharded
forward bit for bit, and the slowest rank's time is what every rank reports.  (The CPU oracle stands in for the
per-rank compute here; on the GPU box the same invariant is checked on the HIP path in
test_gpu_model.py::test_batch_independence_and_determinism.)"""
import os
import socket
import sys

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, q):
    for p in (ROOT, os.path.join(ROOT, "vision-transformer_amd"), os.path.join(ROOT, "tests", "golden")):
        sys.path.insert(0, p)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    torch.set_num_threads(2)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import weights as W
        from oracle import vit_oracle as O
        from vitm_amd.sharding import shard_bounds, max_over_ranks
        cfg = W.CONFIGS["tiny2"]
        sd = {k: torch.from_numpy(v) for k, v in W.make_state_dict(cfg, 0).items()}
        x = torch.from_numpy(W.make_input(cfg, 5, 77))      # 5 images over 2 ranks: shards of 3 and 2
        s, e = shard_bounds(x.shape[0], world, rank)
        mine = O.vitc_forward(x[s:e], sd, W.oracle_cfg(cfg))
        # outputs stay per rank; gathering them is NOT part of the data path -- done here only to check
        parts = [None] * world
        dist.all_gather_object(parts, (s, e, mine))
        slow = max_over_ranks(1.0 + rank, dist)             # rank r "took" 1 + r seconds
        if rank == 0:
            full = O.vitc_forward(x, sd, W.oracle_cfg(cfg))
            got = torch.cat([p[2] for p in sorted(parts, key=lambda t: t[0])])
            q.put((bool(torch.equal(got, full)), [(p[0], p[1]) for p in parts], slow))
        else:
            assert slow == float(world)
    finally:
        dist.destroy_process_group()


def test_two_rank_batch_sharding_equals_unsharded_forward():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    same, bounds, slow = q.get(timeout=240)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert same, "sharded forward differs from the unsharded one"
    assert sorted(bounds) == [(0, 3), (3, 5)]
    assert slow == 2.0


def _allreduce_worker(rank, world, port, q):
    import os
    import torch
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from vitm_amd.training import FlatGradients
        torch.manual_seed(0)
        lin = torch.nn.Sequential(torch.nn.Linear(5, 3), torch.nn.LayerNorm(3))
        fg = FlatGradients(lin.parameters())
        # every parameter's .grad is a view into the flat buffer, in parameter order, each on a 256-byte boundary
        for p, off in zip(lin.parameters(), fg.offsets):
            assert p.grad.data_ptr() == fg.flat.data_ptr() + 4 * off and p.grad.shape == p.shape and off % 64 == 0
        assert fg.offsets == sorted(fg.offsets) and fg.flat.numel() >= sum(p.numel() for p in lin.parameters())
        for i, p in enumerate(lin.parameters()):
            p.grad.fill_(float(rank + 1) * (i + 1))
        fg.all_reduce()
        want = sum(r + 1 for r in range(world)) / world
        ok = all(torch.allclose(p.grad, torch.full_like(p, want * (i + 1))) for i, p in enumerate(lin.parameters()))
        lin.zero_grad(set_to_none=True)
        fg.zero()                      # views are re-attached and zeroed
        ok = ok and all(p.grad is not None and float(p.grad.abs().sum()) == 0.0 for p in lin.parameters())
        q.put((rank, bool(ok)))
    finally:
        dist.destroy_process_group()


def test_flat_gradient_allreduce_two_ranks():
    """The data-parallel gradient exchange of the training step (one flat buffer, averaged), rehearsed over gloo."""
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_allreduce_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in procs)
    for p in procs:
        p.join(timeout=60)
    assert res == [(0, True), (1, True)]


def _reducer_worker(rank, world, port, q):
    import os
    import torch
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from vitm_amd.training import FlatGradients, GradientReducer
        torch.manual_seed(0)
        layers = [torch.nn.Linear(40, 40) for _ in range(5)]
        net = torch.nn.Sequential(*layers)
        fg = FlatGradients(net.parameters())
        red = GradientReducer(fg, None, bucket_bytes=2 * 40 * 40 * 4)     # about two layers per bucket
        assert red.active and not red.avg
        for i, p in enumerate(net.parameters()):
            p.grad.fill_(float(rank + 1) * (i + 1))
        fired = []
        for layer in reversed(layers):                # the order a backward pass finishes them in
            red.ready(list(layer.parameters()))
            fired.append(len(red.works))
        nworks = len(red.works)
        red.finish()
        want = sum(r + 1 for r in range(world)) / world
        ok = all(torch.allclose(p.grad, torch.full_like(p, want * (i + 1))) for i, p in enumerate(net.parameters()))
        # buckets went out while "the backward" was still running, and the alignment gaps stayed zero
        mask = torch.ones_like(fg.flat, dtype=torch.bool)
        for p, off in zip(fg.params, fg.offsets):
            mask[off:off + p.numel()] = False
        ok = ok and 1 <= nworks < len(layers) and fired[0] == 0 and float(fg.flat[mask].abs().sum()) == 0.0
        ok = ok and len(red.works) == 0               # ready for the next step
        q.put((rank, bool(ok)))
    finally:
        dist.destroy_process_group()


def test_bucketed_gradient_allreduce_two_ranks():
    """The overlapped data-parallel exchange: contiguous slices of the flat gradient buffer go out in buckets as the
    backward reaches them (last layer first); rehearsed over gloo."""
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_reducer_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in procs)
    for p in procs:
        p.join(timeout=60)
    assert res == [(0, True), (1, True)]


def _runner_worker(rank, world, port, q):
    for p in (ROOT, os.path.join(ROOT, "vision-transformer_amd"), os.path.join(ROOT, "tests", "golden")):
        sys.path.insert(0, p)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    torch.set_num_threads(1)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from vitm_amd.sharding import ShardedInference
        torch.manual_seed(0)
        model = torch.nn.Sequential(torch.nn.Flatten(), torch.nn.Linear(12, 5))     # a stand-in replica
        x = torch.randn(7, 3, 2, 2)
        runner = ShardedInference(model, dist, micro_batch=2)
        local, (s, e) = runner(x)
        assert local.shape[0] == e - s
        via_fn, b2 = ShardedInference(model, dist)(lambda a, b: x[a:b], n_items=7)
        assert b2 == (s, e) and torch.allclose(via_fn, local, atol=1e-6)   # CPU GEMM blocking differs with the row count
        full = runner.gather(local)
        if rank == 0:
            q.put(bool(full.shape[0] == 7 and torch.allclose(full, model(x), atol=1e-6)))
        else:
            assert full is None
    finally:
        dist.destroy_process_group()


def test_sharded_inference_runner_two_ranks():
    """ShardedInference: contiguous chunks, micro-batching, callable loaders and the optional gather."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_runner_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    ok = q.get(timeout=120)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert ok


def test_bench_self_launches_its_ranks():
    """`python bench.py --gpus 2` outside a launcher starts two ranks itself and reports them; a rank count that
    differs from --gpus is refused with a non-zero exit.  (--selftest-launch: gloo, no GPU work.)"""
    import json
    import subprocess
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--selftest-launch", "--steps", "3",
                        "--warmup", "1"], capture_output=True, text=True, timeout=240, env=env)
    assert r.returncode == 0, r.stderr[-2000:]
    line = [ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1]
    out = json.loads(line)
    assert out["n_gpus"] == 2 and out["ranks_seen"] == 2 and len(out["per_rank"]) == 2
    env["WORLD_SIZE"] = "1"
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--selftest-launch"],
                       capture_output=True, text=True, timeout=120, env=env)
    assert r.returncode != 0 and "refusing" in r.stderr
