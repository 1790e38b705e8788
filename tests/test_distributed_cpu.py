"""N>1 path on CPU: two gloo processes exercise the bucketed gradient all-reduce, the clip sharding and the
folded meter all-reduce of grove_amd.train exactly as the RCCL path uses them."""
import os
import socket

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
    for mode in ("allreduce", "rs_ag"):
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


def test_warmup_decay_lr():
    from grove_amd.train import WarmupDecayLR
    s = WarmupDecayLR(3e-4, 1000, 100)
    assert s.get(0) == 0.0 and abs(s.get(50) - 1.5e-4) < 1e-12 and abs(s.get(100) - 3e-4) < 1e-12
    assert abs(s.get(550) - 1.5e-4) < 1e-12 and s.get(1000) == 0.0
    # DeepSpeed's calling order (scheduler stepped after the optimizer, starting from warmup_min_lr): updates 1 and 2 run at 0,
    # update k at gamma(k - 2)
    assert s.for_update(1) == 0.0 and s.for_update(2) == 0.0 and abs(s.for_update(3) - 3e-6) < 1e-15
    assert abs(s.for_update(102) - 3e-4) < 1e-12
