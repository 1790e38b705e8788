"""N>1 path on CPU: two gloo processes exercise the bucketed gradient all-reduce, the clip sharding and the
folded meter all-reduce of grove_amd.train exactly as the RCCL path uses them."""
import os
import socket
import pytest

import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from grove_amd.train import allreduce_buckets, shard_clips
    g = torch.arange(1000, dtype=torch.float32) * (rank + 1)
    allreduce_buckets(g, 96)  # 11 buckets, last one ragged
    ok = torch.equal(g, torch.arange(1000, dtype=torch.float32) * 3)
    # bf16 wire format (DeepSpeed's default communication dtype under bf16): rounded, summed, widened back in place
    gen = torch.Generator().manual_seed(7)
    base = torch.randn(1000, generator=gen)
    h = base * (rank + 1)
    allreduce_buckets(h, 96, comm_buf=torch.empty(1000, dtype=torch.bfloat16))
    want = (base.to(torch.bfloat16) + (base * 2).to(torch.bfloat16)).float()  # what a bf16 sum of the two rounded ranks holds
    ok = ok and torch.equal(h, want) and (h - base * 3).abs().max().item() <= 3 * base.abs().max().item() * 2 ** -7
    # the overlapped exchange: groups handed over out of order as the backward finishes them, the rest at finish(); both modes
    from grove_amd.train import GradExchange
    for mode in ("allreduce", "rs_ag", "a2a_f32"):
        for wire in (torch.bfloat16, torch.float32):
            base = torch.randn(1003, generator=torch.Generator().manual_seed(11))
            flat = base * (rank + 1)
            ex = GradExchange(flat, world, 96, comm_dtype=wire, mode=mode)
            ex.ready(800, 1003)   # "decoder": ragged tail (1003 is odd: falls back to all-reduce for the last bucket)
            ex.ready(200, 500)    # "lm_head"
            done = ex.finish()    # everything else
            assert sorted(done) == [(0, 200), (200, 500), (500, 800), (800, 1003)], done
            if wire == torch.float32:
                ok = ok and torch.allclose(flat, base * 3, rtol=1e-6, atol=0)
            else:
                want = (base.to(torch.bfloat16).float() + (base * 2).to(torch.bfloat16).float())
                ok = ok and (flat - want).abs().max().item() <= want.abs().max().item() * 2 ** -7
    # fp32 accumulation of a bf16 wire (mode a2a_f32): the sum of the two ROUNDED contributions is exact in fp32 and rounded once —
    # the ring forms round the running sum instead; here (two ranks) both give round(bf16(a) + bf16(2a))
    base = torch.randn(960, generator=torch.Generator().manual_seed(13))
    flat = base * (rank + 1)
    ex = GradExchange(flat, world, 96, comm_dtype=torch.bfloat16, mode="a2a_f32")
    ex.ready(0, 960)
    ex.finish()
    want = (base.to(torch.bfloat16).float() + (base * 2).to(torch.bfloat16).float()).to(torch.bfloat16).float()
    ok = ok and torch.equal(flat, want)
    # sparse embedding-row exchange: each rank touches a few rows of a [50, 8] table; padded to the max count over the ranks;
    # the dense slice ends up holding the fp32 SUM of both ranks' rows (rows touched by both are added), everything else zero
    H, V = 8, 50
    for wire in (torch.bfloat16, torch.float32):
        flat = torch.zeros(16 + V * H + 24)
        flat[:16] = rank + 1.0          # a dense group before the table
        flat[16 + V * H:] = 2.0 * (rank + 1)
        ex = GradExchange(flat, world, 64, comm_dtype=wire, mode="allreduce")
        my_ids = torch.tensor([3, 7, 20] if rank == 0 else [7, 41], dtype=torch.int32)
        ex.sparse_begin(my_ids.numel())
        K = ex.sparse_kmax()
        assert K == 3
        rows = torch.zeros(K, H)
        rows[:my_ids.numel()] = (my_ids.float()[:, None] + 0.5) * (rank + 1)
        ids = torch.full((K,), -1, dtype=torch.int32)
        ids[:my_ids.numel()] = my_ids
        ex.ready(0, 16)
        ex.sparse_rows(ids, rows, 16, 16 + V * H, H)
        done = ex.finish()
        assert sorted(done) == [(0, 16), (16, 16 + V * H), (16 + V * H, flat.numel())], done
        table = flat[16:16 + V * H].view(V, H)
        want = torch.zeros(V, H)
        want[3], want[20], want[41] = 3.5, 20.5, 2 * 41.5
        want[7] = 7.5 + 2 * 7.5
        ok = ok and torch.equal(table, want) and torch.equal(flat[:16], torch.full((16,), 3.0)) and \
            torch.equal(flat[16 + V * H:], torch.full((24,), 6.0))
    mine = shard_clips(7, rank, world)
    gathered = [None] * world
    dist.all_gather_object(gathered, mine)
    t = torch.tensor([1.0 + rank, 1.0, 2.0 * rank, 1.0])  # (sum, count) x 2 meters in ONE collective
    dist.all_reduce(t)
    out[rank] = (ok, gathered, t.tolist())
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_gradient_exchange_and_sharding():
    world, port = 2, _free_port()
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_worker, args=(world, port, out), nprocs=world, join=True)
    for r in range(world):
        ok, gathered, meters = out[r]
        assert ok
        assert sorted(i for part in gathered for i in part) == [0, 0, 1, 2, 3, 4, 5, 6]  # padded by wrap-around
        assert len(gathered[0]) == len(gathered[1]) == 4
        assert meters == [3.0, 2.0, 2.0, 2.0]


def _worker3(rank, world, port, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from grove_amd.train import GradExchange
    # rows touched by all three ranks with values whose fp32 sum depends on the order: (1e8 + 1) - 1e8 = 0 in fp32, 1 + (1e8 - 1e8) = 1
    H, V = 4, 10
    vals = {0: 1.0e8, 1: 1.0, 2: -1.0e8}
    res = {}
    for wire in (torch.float32, torch.bfloat16):
        flat = torch.zeros(V * H + 8)
        flat[V * H:] = float(rank)
        ex = GradExchange(flat, world, 64, comm_dtype=wire, mode="allreduce")
        my_ids = torch.tensor([[5, 2], [2, 5, 9], [5]][rank], dtype=torch.int32)
        ex.sparse_begin(my_ids.numel())
        K = ex.sparse_kmax()
        rows = torch.zeros(K, H)
        rows[:my_ids.numel()] = vals[rank]
        ids = torch.full((K,), -1, dtype=torch.int32)
        ids[:my_ids.numel()] = my_ids
        ex.sparse_rows(ids, rows, 0, V * H, H)
        ex.finish()
        res[str(wire)] = flat.clone()
    # a2a_f32 at world 3 with an odd shard (bucket 99 / 3 = 33: all-reduce fallback) and an even one
    base = torch.randn(198 + 99, generator=torch.Generator().manual_seed(3))
    flat = base * (rank + 1)
    ex = GradExchange(flat, world, 198, comm_dtype=torch.bfloat16, mode="a2a_f32")
    ex.finish()
    res["a2a"] = flat.clone()
    res["a2a_want"] = sum((base * (r + 1)).to(torch.bfloat16).float() for r in range(world))
    out[rank] = res
    dist.barrier()
    dist.destroy_process_group()


def test_three_rank_sparse_rows_sum_in_rank_order_on_every_replica():
    """a token id touched by three or more ranks must be summed in the SAME order on every rank, or the replicas'
    embed_tokens gradients differ bitwise and the master weights drift apart. Rank order: (1e8 + 1) + (-1e8) = 0 in fp32 for row 5
    on every rank (any other order gives 1)."""
    world, port = 3, _free_port()
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_worker3, args=(world, port, out), nprocs=world, join=True)
    for key in ("torch.float32", "torch.bfloat16"):
        for r in (1, 2):
            assert torch.equal(out[0][key], out[r][key]), (key, r)
    t = out[0]["torch.float32"][:40].view(10, 4)
    assert t[5].tolist() == [0.0] * 4          # ((1e8) + 1) - 1e8 in rank order; 1.0 in any other
    assert t[2].tolist() == [1.0e8] * 4 and t[9].tolist() == [1.0] * 4 and float(out[0]["torch.float32"][40]) == 3.0
    for r in range(3):
        assert (out[r]["a2a"] - out[r]["a2a_want"]).abs().max().item() <= out[r]["a2a_want"].abs().max().item() * 2 ** -7
        assert torch.equal(out[r]["a2a"], out[0]["a2a"])


def _worker8(rank, world, port, out):
    """World 8 = the size of the one run that matters (train_scripts/train_howtoground.sh:20-28: 8 ranks per node; train.py:453,
    466-486). Every exchange arm on both wires over a flat buffer whose length is NOT a multiple of world x bucket, groups handed
    over out of order, the touched-row path with row counts uneven over the ranks (two ranks touch nothing), shard_clips."""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from grove_amd.train import GradExchange, allreduce_buckets, shard_clips
    res = {}
    n = 3 * 8 * 96 + 517  # three whole rounds of 8-rank buckets + a ragged remainder that is odd (all-reduce fallback of the tail)
    mult = sum(r + 1 for r in range(world))
    g = torch.arange(n, dtype=torch.float32) * (rank + 1)
    allreduce_buckets(g, 96)
    res["plain"] = bool(torch.equal(g, torch.arange(n, dtype=torch.float32) * mult))
    for mode in ("allreduce", "rs_ag", "a2a_f32"):
        for wire in (torch.bfloat16, torch.float32):
            base = torch.randn(n, generator=torch.Generator().manual_seed(23))
            flat = base * (rank + 1)
            ex = GradExchange(flat, world, 96 * 8, comm_dtype=wire, mode=mode)
            assert ex.bucket % world == 0
            ex.ready(2000, n)     # ragged tail first (the "decoder" group)
            ex.ready(301, 1203)   # neither end on a bucket boundary
            done = ex.finish()
            assert sorted(done) == [(0, 301), (301, 1203), (1203, 2000), (2000, n)], done
            if wire == torch.float32:
                ok = torch.allclose(flat, base * mult, rtol=2e-6, atol=0)
            else:  # a sum of 8 bf16-rounded contributions, running sums rounded to bf16 by the ring forms: <= 8 half-ulps of the result
                want = sum((base * (r + 1)).to(torch.bfloat16).float() for r in range(world))
                ok = (flat - want).abs().max().item() <= want.abs().max().item() * 8 * 2 ** -8
            res[f"{mode}/{wire}"] = (bool(ok), flat.clone())
    # fp32 accumulation of the bf16 wire is EXACT: the sum of the 8 rounded contributions in fp32, rounded once
    base = torch.randn(8 * 96 * 2, generator=torch.Generator().manual_seed(29))
    flat = base * (rank + 1)
    ex = GradExchange(flat, world, 96 * 8, comm_dtype=torch.bfloat16, mode="a2a_f32")
    ex.finish()
    want = sum((base * (r + 1)).to(torch.bfloat16).float() for r in range(world)).to(torch.bfloat16).float()
    res["a2a_exact"] = bool(torch.equal(flat, want))
    # touched rows: rank r touches r % 4 rows (ranks 0 and 4: none) of a [40, 4] table, ids overlapping between ranks
    H, V = 4, 40
    for wire in (torch.float32, torch.bfloat16):
        flat = torch.zeros(8 + V * H + 8)
        flat[:8] = 1.0
        flat[8 + V * H:] = float(rank)
        ex = GradExchange(flat, world, 64, comm_dtype=wire, mode="allreduce")
        my_ids = torch.tensor([(5 * rank + 3 * j) % V for j in range(rank % 4)], dtype=torch.int32)
        ex.sparse_begin(my_ids.numel())
        K = ex.sparse_kmax()
        assert K == 3
        rows = torch.zeros(K, H)
        rows[:my_ids.numel()] = float(rank + 1)
        ids = torch.full((K,), -1, dtype=torch.int32)
        ids[:my_ids.numel()] = my_ids
        ex.sparse_rows(ids, rows, 8, 8 + V * H, H)
        ex.finish()
        want = torch.zeros(V, H)
        for r in range(world):
            for j in range(r % 4):
                want[(5 * r + 3 * j) % V] += float(r + 1)
        res[f"rows/{wire}"] = (bool(torch.equal(flat[8:8 + V * H].view(V, H), want) and torch.equal(flat[:8], torch.full((8,), float(world)))
                                    and torch.equal(flat[8 + V * H:], torch.full((8,), float(sum(range(world)))))), flat.clone())
    res["shards"] = {n_: shard_clips(n_, rank, world) for n_ in (7, 8, 9, 64)}
    out[rank] = res
    dist.barrier()
    dist.destroy_process_group()


def test_eight_rank_gradient_exchange_rows_and_sharding():
    world, port = 8, _free_port()
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_worker8, args=(world, port, out), nprocs=world, join=True)
    for r in range(world):
        res = out[r]
        assert res["plain"] and res["a2a_exact"], r
        for k, v in res.items():
            if isinstance(v, tuple):
                assert v[0], (r, k)
                assert torch.equal(v[1], out[0][k][1]), (r, k)  # every replica holds the same bits
    for n_ in (7, 8, 9, 64):
        parts = [out[r]["shards"][n_] for r in range(world)]
        per = (n_ + world - 1) // world
        assert all(len(p) == per for p in parts)
        flat = sorted(i for p in parts for i in p)
        assert set(flat) == set(range(n_)) and len(flat) == per * world  # every clip at least once, padding by wrap-around


def test_warmup_decay_lr():
    from grove_amd.train import WarmupDecayLR
    s = WarmupDecayLR(3e-4, 1000, 100)
    assert s.get(0) == 0.0 and abs(s.get(50) - 1.5e-4) < 1e-12 and abs(s.get(100) - 3e-4) < 1e-12
    assert abs(s.get(550) - 1.5e-4) < 1e-12 and s.get(1000) == 0.0
    # DeepSpeed's calling order (scheduler stepped after the optimizer, starting from warmup_min_lr): updates 1 and 2 run at 0,
    # update k at gamma(k - 2)
    assert s.for_update(1) == 0.0 and s.for_update(2) == 0.0 and abs(s.for_update(3) - 3e-6) < 1e-15
    assert abs(s.for_update(102) - 3e-4) < 1e-12


@pytest.mark.parametrize("total,warm", [(250, 100), (5000, 100), (60, 100), (7, 1)])
def test_lr_of_every_update_follows_deepspeeds_calling_order(total, warm):
    """`for_update` (which lr the k-th optimizer update runs with) against a step-by-step simulation of DeepSpeed's
    schedule classes and engine order restated in oracle/ds_lr_schedule.py (deepspeed==0.15.1 is not installed offline: restated from the
    published source, flagged there) — warm-up, decay, past the end, a total shorter than the warm-up, and warm-up < 2 (clamped to 2)."""
    from grove_amd.train import WarmupDecayLR
    from oracle.ds_lr_schedule import lr_of_updates
    ref = lr_of_updates(3e-4, total, total + 30, warmup_num_steps=warm)
    s = WarmupDecayLR(3e-4, total, warm)
    mine = [s.for_update(k) for k in range(1, total + 31)]
    assert max(abs(a - b) for a, b in zip(ref, mine)) <= 1e-18, [(k + 1, a, b) for k, (a, b) in enumerate(zip(ref, mine)) if abs(a - b) > 1e-18][:5]


def _infer_worker(rank, world, port, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from types import SimpleNamespace
    from grove_amd import infer

    calls = []

    def fake_infer_clip(model, g_all, s_all, prompt_ids, size, **kw):  # the per-clip driver needs a GPU; the job logic does not
        calls.append(int(g_all.flatten()[0]))
        return {"pred_bboxes": [g_all.flatten()[:4] + rank * 0.0], "frame_indices": [0], "by_rank": rank}

    infer.infer_clip = fake_infer_clip
    clips = [(f"clip{i}", torch.full((1, 3, 8, 2, 2), float(i)), torch.zeros(1, 3, 8, 2, 2), (640, 360)) for i in range(5)]
    model = SimpleNamespace(dev=torch.device("cpu"))
    merged = infer.infer_dataset(model, clips, torch.tensor([1, -200, 5]))
    out[rank] = (calls, sorted(merged), {k: v["by_rank"] for k, v in merged.items()}, float(merged["clip3"]["pred_bboxes"][0][0]))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_inference_job_shards_clips_and_gathers():
    """infer_iground.py:290-293, 538-551 on two gloo ranks: the un-shuffled DistributedSampler partition (wrap-around padded), one
    barrier + all_gather_object at the end, first-wins merge — every rank ends with every clip exactly once."""
    world, port = 2, _free_port()
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_infer_worker, args=(world, port, out), nprocs=world, join=True)
    assert out[0][0] == [0, 2, 4] and out[1][0] == [1, 3, 0]          # 5 clips on 2 ranks: rank 1's last round wraps to clip 0
    for r in range(world):
        calls, keys, by_rank, v3 = out[r]
        assert keys == [f"clip{i}" for i in range(5)]
        assert by_rank == {"clip0": 0, "clip1": 1, "clip2": 0, "clip3": 1, "clip4": 0}   # duplicate clip0: rank 0's copy wins
        assert v3 == 3.0


def test_inference_job_single_process():
    from types import SimpleNamespace
    from grove_amd import infer
    orig = infer.infer_clip
    infer.infer_clip = lambda model, g, s, p, size, **kw: {"n": int(g.flatten()[0])}
    try:
        clips = [(f"c{i}", torch.full((1, 1), float(i)), torch.zeros(1, 1), (1, 1)) for i in range(3)]
        res = infer.infer_dataset(SimpleNamespace(dev=torch.device("cpu")), clips, None)
    finally:
        infer.infer_clip = orig
    assert res == {"c0": {"n": 0}, "c1": {"n": 1}, "c2": {"n": 2}}
