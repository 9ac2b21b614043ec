"""CPU tier: the N > 1 path — shard bookkeeping, per-sample noise streams and the gather — with world_size 2 over gloo.
The denoiser is a CPU stand-in (the HIP kernels need a GPU); what is checked is that a sharded run reproduces the
single-process result sample for sample."""
import os

import torch
import torch.multiprocessing as mp

from stedm_amd import parallel as par

GLOBAL_B, SHAPE, STEPS = 7, (4, 8, 8), 3


def _denoise(x, ids):
    """Stand-in sampling loop: per-sample arithmetic only, with per-sample noise each step."""
    for i in range(STEPS):
        x = torch.tanh(x * 0.9) + 0.1 * par.per_sample_normal(1234, ids, SHAPE, stream=1 + i)
    return x


def _worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import torch.distributed as dist
    dist.init_process_group("gloo", rank=rank, world_size=world)
    lo, hi = par.shard_range(GLOBAL_B, rank, world)
    ids = list(range(lo, hi))
    x = _denoise(par.per_sample_normal(1234, ids, SHAPE), ids)
    full = par.all_gather_samples(x, GLOBAL_B)
    if rank == 0:
        q.put(full.numpy())
    dist.barrier()
    dist.destroy_process_group()


def test_shard_range_partitions():
    for gb in (1, 7, 64, 512):
        for w in (1, 2, 3, 8):
            r = [par.shard_range(gb, k, w) for k in range(w)]
            assert r[0][0] == 0 and r[-1][1] == gb
            assert all(r[i][1] == r[i + 1][0] for i in range(w - 1))
            assert max(h - l for l, h in r) - min(h - l for l, h in r) <= 1


def test_two_rank_gloo_matches_single_process():
    ids = list(range(GLOBAL_B))
    ref = _denoise(par.per_sample_normal(1234, ids, SHAPE), ids)
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = torch.from_numpy(q.get(timeout=120))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert torch.equal(got, ref)   # bit-identical: sharding does not change any sample


def _grad_worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import torch.distributed as dist
    dist.init_process_group("gloo", rank=rank, world_size=world)
    g = torch.Generator().manual_seed(100 + rank)
    flat = torch.randn(10_000, generator=g)
    w = par.all_reduce_buckets(flat, 3_000)          # 4 buckets, the last one short
    if rank == 0:
        q.put((w, flat.numpy()))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_gloo_gradient_buckets_sum():
    """training path: the gradient arena is summed over ranks bucket by bucket, in place (the 1/world average is applied by the
    optimizer kernel)"""
    ref = sum(torch.randn(10_000, generator=torch.Generator().manual_seed(100 + r)) for r in range(2))
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 31500 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_grad_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    world, got = q.get(timeout=120)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert world == 2
    assert torch.allclose(torch.from_numpy(got), ref, rtol=0, atol=1e-6)
    assert par.all_reduce_buckets(torch.ones(8), 4) == 1     # not initialised: identity


def _bench(args, env_extra):
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT")}
    env.update(env_extra)
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py")] + args, env=env, capture_output=True, text=True, timeout=300)
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    return r, (json.loads(lines[-1]) if lines else None)


def test_bench_gpus_flag_launches_the_ranks_itself():
    """`python bench.py --gpus 2` with no launcher around it must start two ranks (rehearsed on CPU tensors over gloo: the launcher, the
    process group, the barriers, the max-over-ranks timing, the sample all-gather and the rank-0 line) and say n_gpus 2."""
    r, line = _bench(["--gpus", "2", "--steps", "2", "--warmup", "1", "--batch", "3"], {"STEDM_BENCH_DRY": "1"})
    assert r.returncode == 0, r.stderr[-2000:]
    assert line["n_gpus"] == 2 and line["rccl_ranks"] == 2 and line["dry_run"] is True and line["value"] is None
    assert line["config"]["global_batch"] == 6
    assert sum(ln.startswith("{") for ln in r.stdout.splitlines()) == 1          # ONE line, from rank 0


def test_bench_eight_ranks_uneven_shards_rehearsal():
    """the N = 8 line the driver will ask for, rehearsed on CPU tensors over gloo: eight ranks started by bench.py
    itself, a global batch the ranks do not divide (510 = 6 x 64 + 2 x 63), shard_range / all_gather of unequal shards / max-over-ranks at
    world 8, ONE line from rank 0 that says why it carries no cpu_baseline."""
    r, line = _bench(["--gpus", "8", "--steps", "2", "--warmup", "1", "--batch", "64"], {"STEDM_BENCH_DRY": "1", "STEDM_BENCH_DRY_GLOBAL": "510"})
    assert r.returncode == 0, r.stderr[-2000:]
    assert line["n_gpus"] == 8 and line["rccl_ranks"] == 8 and line["dry_run"] is True
    assert line["config"]["global_batch"] == 510 and line["config"]["shard_sizes"] == [64] * 6 + [63] * 2
    assert isinstance(line["cpu_baseline"], str) and "world > 1" in line["cpu_baseline"]
    assert sum(ln.startswith("{") for ln in r.stdout.splitlines()) == 1


def test_bench_under_torch_distributed_run_as_the_driver_launches_it():
    """The driver's N > 1 command line, verbatim: `python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1
    --master-port P bench.py --gpus N --steps K --warmup W`. bench.py must take RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* from the launcher
    (not start ranks of its own), and rank 0 alone prints the line (rehearsed on CPU tensors over gloo)."""
    import json
    import socket
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = str(sk.getsockname()[1])
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT", "MASTER_ADDR")}
    env["STEDM_BENCH_DRY"] = "1"
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                        "--master-port", port, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--batch", "3"],
                       env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1                                                        # ONE line, from rank 0
    line = json.loads(lines[0])
    assert line["n_gpus"] == 2 and line["rccl_ranks"] == 2 and line["config"]["global_batch"] == 6 and line["steps"] == 2 and line["warmup"] == 1


def test_bench_refuses_a_world_size_that_contradicts_gpus():
    r, line = _bench(["--gpus", "1"], {"STEDM_BENCH_DRY": "1", "WORLD_SIZE": "2", "RANK": "0", "LOCAL_RANK": "0"})
    assert r.returncode != 0 and line is None and "WORLD_SIZE=2" in r.stderr


def test_bucket_schedule_cuts_and_firing():
    """buckets are cut at parameter boundaries on vector-aligned offsets, cover the padded arena exactly once, the tail parameters
    (gradients that arrive after the backward) get a bucket of their own, and a bucket fires exactly once, when its last parameter completes"""
    import random
    rnd = random.Random(3)
    sizes = [rnd.choice([4, 8, 12, 36, 100, 128, 1152, 6]) for _ in range(60)] + [6, 2]
    offs, o = [], 0
    for i, sz in enumerate(sizes):
        if i == 60:
            o = (o + 3) // 4 * 4
        offs.append(o); o += sz
    total = (o + 3) // 4 * 4
    s = par.BucketSchedule(offs, sizes, 2000, tail_from=60, align=4, total=total)
    assert s.bounds[0][0] == 0 and s.bounds[-1][1] == total and s.bounds[-1][0] == offs[60]
    for (a, b), (c, d) in zip(s.bounds, s.bounds[1:]):
        assert b == c and c % 4 == 0 and a < b
    assert all(s.bounds[s.param_bucket[i]][0] <= offs[i] and offs[i] + sizes[i] <= s.bounds[s.param_bucket[i]][1] for i in range(len(sizes)))
    order = list(range(len(sizes)))[::-1]
    rnd.shuffle(order)
    fired = [b for i in order if (b := s.done(i)) is not None]
    assert sorted(fired) == list(range(len(s.bounds))) and s.pending() == [] and s.done(order[0]) is None


def _overlap_worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import torch.distributed as dist
    dist.init_process_group("gloo", rank=rank, world_size=world)
    g = torch.Generator().manual_seed(100 + rank)
    sizes = [4 * k for k in (3, 50, 7, 200, 1, 90, 33, 64, 5, 120)]
    offs = [sum(sizes[:i]) for i in range(len(sizes))]
    sched = par.BucketSchedule(offs, sizes, 600, align=4, total=sum(sizes))
    grads = torch.randn(sum(sizes), generator=g)
    plain = grads.clone()
    par.all_reduce_bounds(plain, sched.bounds)                    # after the "backward", bucket by bucket
    over = grads.clone()
    works = []
    for i in reversed(range(len(sizes))):                         # the backward walks the parameters back to front
        b = sched.done(i)
        if b is not None:
            lo, hi = sched.bounds[b]
            works.append(dist.all_reduce(over[lo:hi], op=dist.ReduceOp.SUM, async_op=True))
    for w in works:
        w.wait()
    if rank == 0:
        q.put((plain.numpy(), over.numpy(), len(works), len(sched.bounds)))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_overlapped_buckets_equal_the_plain_bucketed_all_reduce():
    """world_size 2 over gloo: collectives issued per bucket as soon as the bucket is complete (overlapping the rest of the backward)
    give bitwise the sums of the same buckets reduced after the backward"""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29800 + (os.getpid() % 1000)
    procs = [ctx.Process(target=_overlap_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    plain, over, nworks, nb = q.get(timeout=120)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert nworks == nb and nb >= 3
    assert (plain == over).all()


def test_sharded_noise_streams_do_not_depend_on_the_world_size():
    """x_T / per-step noise of sample i under 1, 2 and 3 ranks (the host logic of latent_diffusion.predict_latents_sharded)"""
    from stedm_amd import parallel as par
    full = par.per_sample_normal(1234, list(range(GLOBAL_B)), (4, 32, 32), stream=3)
    for world in (2, 3):
        parts = [par.per_sample_normal(1234, list(range(*par.shard_range(GLOBAL_B, r, world))), (4, 32, 32), stream=3) for r in range(world)]
        assert torch.equal(torch.cat(parts), full)
