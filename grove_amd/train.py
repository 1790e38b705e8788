"""Data-parallel training entry points — the mirror of the reference's train.py for the hot path.

The reference drives the model through a DeepSpeed ZeRO-2 engine (train.py:466-486, 761-782). ZeRO is a
memory trick for 40-80 GB GPUs; at 288 GB per MI355X the build keeps a full replica per GPU and
exchanges gradients with plain RCCL all-reduces over xGMI (one process per GPU, torch.distributed
backend "nccl" == RCCL). GroveEngine exposes the engine methods train.py uses: __call__, backward(loss),
step(), train()/eval(), save_checkpoint(dir), load_checkpoint(dir).

Gradient exchange: the trainable set (~482 M parameters that really receive gradients)
lives in ONE flat fp32 buffer; it is all-reduced in a few large buckets on a side stream (xGMI is
point-to-point, large messages amortise the per-link latency), then a fused AdamW kernel updates the
fp32 master copy and the bf16 model weights in one pass.
"""
import argparse
import math
import os
import time

import torch
import torch.distributed as dist

from . import ops
from .model.GROVE import GROVEForCausalLM, trainable_names
from .synthetic import GroveDims


def parse_args(argv=None):
    """Subset of train.py:40-112 that concerns the hot path (same names and defaults)."""
    p = argparse.ArgumentParser(description="GROVE Model Training (MI355X)")
    p.add_argument("--precision", default="bf16", type=str)
    p.add_argument("--num_frames", default=8, type=int)
    p.add_argument("--out_dim", default=256, type=int)
    p.add_argument("--lr", default=0.0003, type=float)
    p.add_argument("--wd", default=0.0, type=float)
    p.add_argument("--beta1", default=0.9, type=float)
    p.add_argument("--beta2", default=0.95, type=float)
    p.add_argument("--epochs", default=10, type=int)
    p.add_argument("--steps_per_epoch", default=500, type=int)
    p.add_argument("--batch_size", default=1, type=int)
    p.add_argument("--grad_accumulation_steps", default=1, type=int)
    p.add_argument("--ce_loss_weight", default=1.0, type=float)
    p.add_argument("--giou_loss_weight", default=1.0, type=float)
    p.add_argument("--temp_objectness_loss_weight", default=1.0, type=float)
    p.add_argument("--train_mask_decoder", action="store_true", default=True)
    p.add_argument("--print_freq", default=1, type=int)
    p.add_argument("--local_rank", default=int(os.environ.get("LOCAL_RANK", 0)), type=int)
    p.add_argument("--log_dir", default="./output", type=str)
    p.add_argument("--exp_name", default="grove", type=str)
    p.add_argument("--start_epoch", default=0, type=int)
    p.add_argument("--eval_only", action="store_true", default=False)
    p.add_argument("--auto_resume", action="store_true", default=False)
    p.add_argument("--resume", default="", type=str)
    p.add_argument("--grove_weights", default="", type=str, help="consolidated pytorch_model.bin / HF directory to start from")
    p.add_argument("--val_batches", default=2, type=int, help="validation batches per epoch (synthetic loader)")
    # synthetic-data stand-ins for the dataset arguments (datasets / tokenizer are out of scope)
    p.add_argument("--dims", default="full", choices=["full", "tiny"], help="architecture size when no checkpoint gives it")
    p.add_argument("--text_len", default=128, type=int)
    p.add_argument("--n_det", default=3, type=int)
    return p.parse_args(argv)


def initialize_model(args, dims: GroveDims, state_dict=None, device=None):
    """train.py:197-218: build GROVEForCausalLM in bf16 with the loss weights / token ids of `args`."""
    device = device or torch.device("cuda", args.local_rank)
    return GROVEForCausalLM(dims=dims, device=device, state_dict=state_dict, train=True,
                            det_token_idx=getattr(args, "det_token_idx", dims.det_token_idx), num_frames=args.num_frames,
                            out_dim=args.out_dim, ce_loss_weight=args.ce_loss_weight, giou_loss_weight=args.giou_loss_weight,
                            temp_objectness_loss_weight=args.temp_objectness_loss_weight,
                            train_mask_decoder=args.train_mask_decoder, use_temp_objectness=True)


def prepare_model_for_training(model):
    """train.py:234-333 freeze policy: returns the names that train (the rest is frozen by construction)."""
    return trainable_names(model.dims)


class WarmupDecayLR:
    """DeepSpeed WarmupDecayLR (train.py:471-474): linear warm-up 0 -> lr over warmup steps, then linear decay to 0."""

    def __init__(self, lr, total_steps, warmup_steps=100):
        self.lr, self.total, self.warm = lr, max(total_steps, 1), warmup_steps
        self.last = 0.0

    def for_update(self, k):
        """Learning rate of the k-th optimizer update (k = 1, 2, ...) under DeepSpeed's calling order: the scheduler is built with
        last_batch_iteration = -1, which writes warmup_min_lr (0, train.py:472) into the optimizer — the value update 1 uses —
        and the engine steps the scheduler AFTER every optimizer update (iteration 0 after update 1, gamma(0) = 0 for update 2,
        gamma(1) for update 3, ...). So update k runs with gamma(k - 2): the first two updates have lr = 0.
        UNPINNED: this rests on deepspeed==0.15.1's `WarmupLR.__init__` writing `warmup_min_lr` into the optimizer's
        param groups when `last_batch_iteration == -1` (deepspeed/runtime/lr_schedules.py, `_format_param` / `update_lr` called from
        `__init__`) and on `DeepSpeedEngine._take_model_step` calling `lr_scheduler.step()` after `optimizer.step()`. DeepSpeed is not
        installed offline, so no lr trace of the reference stack could be captured into tests/golden; older DeepSpeed releases left
        the optimizer's base lr for update 1. `get(step)` is the closed form either way; only the index shift is version dependent."""
        return self.get(max(k - 2, 0))

    def get(self, step):
        if step < self.warm:
            g = step / max(1, self.warm)
        else:
            g = max(0.0, (self.total - step) / max(1.0, self.total - self.warm))
        self.last = self.lr * g
        return self.last

    def get_last_lr(self):
        return [self.last]


def allreduce_buckets(flat, bucket_elems, comm_stream=None, comm_buf=None):
    """SUM all-reduce of one flat gradient buffer in buckets of `bucket_elems` elements. On GPU the
    collectives are issued on `comm_stream` (RCCL over xGMI) after the producing stream's work, and the
    compute stream waits for them; on CPU (gloo, used by the tests) the same bucketing runs inline.
    `comm_buf` (bf16, same length): exchange in bf16 — DeepSpeed's `communication_data_type` default when bf16 is enabled
    (the reference's config sets none, train.py:466-478), half the xGMI bytes of the fp32 buffer: the gradients are rounded
    into comm_buf, summed there, and widened back into `flat`."""
    n = flat.numel()
    wire = flat
    if comm_buf is not None:
        assert comm_buf.numel() == n and comm_buf.dtype == torch.bfloat16
        if flat.is_cuda:
            ops.to_bf16(flat, out=comm_buf)
        else:
            comm_buf.copy_(flat)  # host tensors exist only in the gloo tests
        wire = comm_buf
    if comm_stream is None:
        handles = [dist.all_reduce(wire[s0:s0 + bucket_elems], op=dist.ReduceOp.SUM, async_op=True)
                   for s0 in range(0, n, bucket_elems)]
        for h in handles:
            h.wait()
    else:
        ev = torch.cuda.Event()
        ev.record()
        with torch.cuda.stream(comm_stream):
            comm_stream.wait_event(ev)
            handles = [dist.all_reduce(wire[s0:s0 + bucket_elems], op=dist.ReduceOp.SUM, async_op=True)
                       for s0 in range(0, n, bucket_elems)]
            for h in handles:
                h.wait()
        torch.cuda.current_stream().wait_stream(comm_stream)
    if comm_buf is not None:
        if flat.is_cuda:
            ops.to_f32(comm_buf, out=flat)
        else:
            flat.copy_(comm_buf)


class GradExchange:
    """Gradient exchange of the flat fp32 gradient buffer, overlapped with the backward.

    The trainable parameters lie in the flat buffer in GROUPS whose gradients complete at different points of the backward
    (reverse-autograd order: box decoder -> lm_head -> text_hidden_fcs -> SAM adapters 3..0 on the SAM stream -> embed_tokens +
    mm_projector at the very end). `ready(lo, hi, stream)` is called by the model as soon as a group's slice [lo, hi) is final:
    the slice is rounded into the bf16 wire buffer (DeepSpeed's communication dtype under bf16; fp32 optional) and exchanged on the
    communication stream in buckets, while the rest of the backward keeps the CUs busy. `finish()` (engine.step) makes the compute
    stream wait for the collectives and widens the wire buffer back into the fp32 buffer the optimizer reads.
      mode "allreduce": one SUM all-reduce per bucket (RCCL accumulates in the wire dtype).
      mode "rs_ag":     reduce-scatter + all-gather per bucket (the two halves of a ring all-reduce as separate RCCL calls: the form
                        a sharded optimizer update slots between; here the full replica updates everything, so the result is the same).
      mode "a2a_f32":   FP32 ACCUMULATION of a bf16 wire: per bucket an all-to-all hands rank r chunk r of
                        every rank's bucket (xGMI is point-to-point and fully connected: each chunk crosses ONE link, no ring hops),
                        the rank sums its `world` copies in fp32 (grove_colsum_f32), rounds ONCE, and an all-gather distributes the
                        shard sums — the same bytes per rank as reduce-scatter + all-gather, one rounding instead of world - 1.
    `sparse_rows(...)`: the embedding table's gradient has at most B * L non-zero rows per rank (the text tokens of the batch) out of
    32 K — it travels as an all-gather of (row ids, rows) and is summed in fp32 into the dense slice by every rank
    (<= 512 rows x 8 KB per rank instead of 262 MB; SURVEY.md section 8(a) a7 "sparse rows"). The optimizer stays dense.
    On CPU tensors (gloo: the tests) the same code runs inline without streams."""

    def __init__(self, flat, world, bucket_elems, comm_dtype=torch.bfloat16, mode="allreduce", comm_stream=None):
        assert mode in ("allreduce", "rs_ag", "a2a_f32")
        self.flat, self.world, self.mode = flat, world, mode
        self.wire = torch.empty(flat.numel(), dtype=comm_dtype, device=flat.device) if comm_dtype != torch.float32 else None
        # buckets are multiples of the world size so that reduce-scatter shards are equal
        self.bucket = max(world, bucket_elems // world * world)
        self.stream = comm_stream
        self.pending = []     # [lo, hi) ranges already handed to the communication stream this step
        self.sparse_done = [] # [lo, hi) ranges whose SUM already sits in `flat` (sparse_rows): not widened from the wire
        self.handles = []
        self._keep = []       # staging tensors of collectives in flight (freed at finish)
        self._kmax = None

    def _sum_copies(self, recv, out):
        """out (wire dtype) [k] = round(sum over the `world` rows of recv [world, k]) with the sum in fp32."""
        if recv.is_cuda:
            acc = ops.colsum(recv)                       # fp32 [k]
            if out.dtype == torch.float32:
                out.copy_(acc)
            else:
                ops.to_bf16(acc, out=out)
        else:
            out.copy_(recv.float().sum(0).to(out.dtype))

    def _exchange(self, buf):
        n = buf.numel()
        for s0 in range(0, n, self.bucket):
            b = buf[s0:s0 + self.bucket]
            if self.mode in ("rs_ag", "a2a_f32") and b.numel() % self.world == 0:
                k = b.numel() // self.world
                r = dist.get_rank()
                shard = b[r * k:(r + 1) * k]  # in place: RCCL reduces into / gathers from the rank's own slice of the bucket
                if self.mode == "a2a_f32":
                    recv = torch.empty((self.world, k), dtype=b.dtype, device=b.device)
                    dist.all_to_all_single(recv.view(-1), b)  # (stream-ordered on the communication stream; blocking on gloo)
                    self._sum_copies(recv, shard)
                    if b.is_cuda:
                        self._keep.append(recv)
                        self.handles.append(dist.all_gather_into_tensor(b, shard, async_op=True))
                    else:
                        dist.all_gather_into_tensor(b, shard.clone())
                elif b.is_cuda:  # stream-ordered on the communication stream
                    self.handles.append(dist.reduce_scatter_tensor(shard, b, op=dist.ReduceOp.SUM, async_op=True))
                    self.handles.append(dist.all_gather_into_tensor(b, shard, async_op=True))
                else:          # gloo (tests): no in-place aliasing, no ordering between queued operations
                    tmp = torch.empty_like(shard)
                    dist.reduce_scatter_tensor(tmp, b.clone(), op=dist.ReduceOp.SUM)
                    dist.all_gather_into_tensor(b, tmp)
            else:  # ragged tail of a group (not a multiple of the world size): plain all-reduce
                self.handles.append(dist.all_reduce(b, op=dist.ReduceOp.SUM, async_op=True))

    def ready(self, lo, hi, producer_stream=None, event=None):
        """Gradients of flat[lo:hi] are final once `producer_stream` reaches this point (or once `event`, recorded earlier by the
        caller, has passed): start their exchange. Collectives are issued in call order — identical on every rank."""
        if hi <= lo:
            return
        src = self.flat[lo:hi]
        if not self.flat.is_cuda:
            buf = src
            if self.wire is not None:
                self.wire[lo:hi].copy_(src)
                buf = self.wire[lo:hi]
            self._exchange(buf)
            self.pending.append((lo, hi))
            return
        ev = event
        if ev is None:
            ev = torch.cuda.Event()
            ev.record(producer_stream if producer_stream is not None else torch.cuda.current_stream(self.flat.device))
        with torch.cuda.stream(self.stream):
            self.stream.wait_event(ev)
            buf = src
            if self.wire is not None:
                buf = self.wire[lo:hi]
                ops.to_bf16(src, out=buf)
            self._exchange(buf)
        self.pending.append((lo, hi))

    # ---- sparse rows (embed_tokens)
    def sparse_begin(self, count):
        """Forward time: this rank will contribute `count` distinct rows. The padded row count every rank uses is the MAX over
        ranks; its tiny all-reduce is issued now (first in the communication queue of the step) and read at the end of the
        backward, when it has long completed — no host stall in the step."""
        t = torch.tensor([int(count)], dtype=torch.int32)
        if not self.flat.is_cuda:
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            self._kmax = (t, None)
            return
        if getattr(self, "_kmax_host", None) is None:  # one pinned word and one device word for the life of the exchange
            self._kmax_host = torch.empty(1, dtype=torch.int32, pin_memory=True)
            self._kmax_dev = torch.empty(1, dtype=torch.int32, device=self.flat.device)
        host, td = self._kmax_host, self._kmax_dev
        with torch.cuda.stream(self.stream):
            td.fill_(int(count))
            dist.all_reduce(td, op=dist.ReduceOp.MAX)
            host.copy_(td, non_blocking=True)
            ev = torch.cuda.Event()
            ev.record(self.stream)
        self._kmax = (host, ev, td)

    def sparse_kmax(self):
        assert self._kmax is not None, "sparse_begin() was not called in this step's forward"
        if self._kmax[1] is not None:
            self._kmax[1].synchronize()
        return int(self._kmax[0][0])

    def sparse_rows(self, ids, rows, lo, hi, ld, producer_stream=None, event=None):
        """ids int32 [K] (distinct row numbers of this rank, -1 = padding), rows [K, ld] fp32 (this rank's gradient rows in the
        order of ids; padding rows ignored), K = the same on every rank (sparse_kmax()). The dense slice flat[lo:hi] viewed as
        [*, ld] must be ZERO on entry; on return (stream-ordered) it holds the sum over all ranks. All-gather of ids and of the rows
        in the wire dtype; the sum itself is fp32 (scatter-add of every rank's block)."""
        K = ids.numel()
        dense = self.flat[lo:hi].view(-1, ld)
        wdt = self.wire.dtype if self.wire is not None else torch.float32
        if not self.flat.is_cuda:
            wr = rows.to(wdt)
            all_ids = torch.empty(self.world * K, dtype=torch.int32)
            all_rows = torch.empty((self.world * K, ld), dtype=wdt)
            dist.all_gather_into_tensor(all_ids, ids)
            dist.all_gather_into_tensor(all_rows.view(-1), wr.reshape(-1))
            keep = all_ids >= 0
            dense.index_add_(0, all_ids[keep].long(), all_rows[keep].float())
        else:
            ev = event
            if ev is None:
                ev = torch.cuda.Event()
                ev.record(producer_stream if producer_stream is not None else torch.cuda.current_stream(self.flat.device))
            with torch.cuda.stream(self.stream):
                self.stream.wait_event(ev)
                wr = rows if wdt == torch.float32 else ops.to_bf16(rows)
                all_ids = torch.empty(self.world * K, dtype=torch.int32, device=ids.device)
                all_rows = torch.empty((self.world * K, ld), dtype=wdt, device=ids.device)
                dist.all_gather_into_tensor(all_ids, ids)
                dist.all_gather_into_tensor(all_rows.view(-1), wr.view(-1))
                if wdt == torch.float32:
                    ops.scatter_add_rows_f32(all_rows, dense, all_ids)
                else:
                    ops.scatter_add_f32(all_rows, dense, all_ids, self.world * K, ld)
                for t in (ids, rows, wr, all_ids, all_rows):
                    t.record_stream(self.stream)
        self.pending.append((lo, hi))
        self.sparse_done.append((lo, hi))
        self._kmax = None

    def finish(self):
        """Everything not handed over by ready() is exchanged now; then wait and widen. Returns the exchanged ranges."""
        covered = sorted(self.pending)
        pos, gaps = 0, []
        for lo, hi in covered:
            if lo > pos:
                gaps.append((pos, lo))
            pos = max(pos, hi)
        if pos < self.flat.numel():
            gaps.append((pos, self.flat.numel()))
        for lo, hi in gaps:
            self.ready(lo, hi)
        for h in self.handles:
            h.wait()
        self.handles = []
        if self.flat.is_cuda:
            torch.cuda.current_stream(self.flat.device).wait_stream(self.stream)
        self._keep = []
        if self.wire is not None:
            dense = [r for r in self.pending if r not in self.sparse_done]
            if not self.sparse_done:
                dense = [(0, self.flat.numel())]
            for lo, hi in dense:  # (a sparse slice already holds its fp32 sum: the wire never carried it)
                if self.flat.is_cuda:
                    ops.to_f32(self.wire[lo:hi], out=self.flat[lo:hi])
                else:
                    self.flat[lo:hi].copy_(self.wire[lo:hi])
        done, self.pending, self.sparse_done = self.pending, [], []
        return done


def shard_clips(n_clips, rank, world):
    """DistributedSampler partition of train.py:453 (no shuffle, padded by wrap-around): clip indices of `rank`."""
    per = (n_clips + world - 1) // world
    idx = list(range(n_clips)) + list(range(per * world - n_clips))
    return idx[rank:per * world:world]


class GroveEngine:
    """Replica-per-GPU data-parallel engine with the DeepSpeed-engine surface train.py relies on."""

    def __init__(self, model: GROVEForCausalLM, args, total_steps=None, bucket_bytes=128 << 20, comm_dtype=torch.bfloat16,
                 exchange="allreduce", overlap=True, sparse_embed=True):
        self.module = model
        self.args = args
        self.dev = model.dev
        torch.cuda.set_device(self.dev)  # what deepspeed.initialize() did for the reference: this rank's launches go to ITS GPU
        self.world = dist.get_world_size() if dist.is_initialized() else 1
        self.rank = dist.get_rank() if dist.is_initialized() else 0
        self.names = model.trainable
        g = model._flat_grad
        self.master = torch.zeros_like(g)
        self.m = torch.zeros_like(g)
        self.v = torch.zeros_like(g)
        self.slices = []
        for n in self.names:
            off = model._grad_off[n]
            k = model._sd[n].numel()
            w = model._sd[n]
            if n.endswith("conv3d.weight"):
                w = w.permute(0, 2, 3, 4, 1)  # the contiguous tap-major storage behind the canonical view
            assert w.is_contiguous(), n
            ops.to_f32(w.reshape(-1), out=self.master[off:off + k])
            self.slices.append((n, off, k, w))
        # device tables of the multi-tensor optimizer step (the slices keep their tensors alive, so the pointers stay valid)
        self._seg_off = torch.tensor([off for _, off, _, _ in self.slices], dtype=torch.int64).to(g.device)
        self._seg_len = torch.tensor([k for _, _, k, _ in self.slices], dtype=torch.int64).to(g.device)
        self._seg_ptr = torch.tensor([w.data_ptr() for _, _, _, w in self.slices], dtype=torch.int64).to(g.device)
        total = total_steps if total_steps is not None else args.epochs * args.steps_per_epoch
        self.scheduler = WarmupDecayLR(args.lr, total, 100)
        self.global_step = 0
        self.micro = 0
        self.clip = 1.0  # "gradient_clipping": 1.0 (train.py:475)
        self._sumsq = torch.zeros(1, dtype=torch.float32, device=g.device)
        self._norm = torch.zeros(1, dtype=torch.float32, device=g.device)
        # gradient exchange: bf16 on the wire like DeepSpeed under bf16 (engine.communication_data_type) or torch.float32; bucketed,
        # launched from inside the backward as each parameter group's gradients complete (GradExchange)
        self.comm_stream = torch.cuda.Stream(device=self.dev) if self.world > 1 else None
        self.exchange = None
        if self.world > 1:
            self.exchange = GradExchange(g, self.world, bucket_bytes // (2 if comm_dtype == torch.bfloat16 else 4), comm_dtype, exchange,
                                         self.comm_stream)
        self.overlap = overlap
        # embed_tokens' gradient as touched rows (all-gather of (ids, rows), fp32 sum) instead of the dense 131 M-element slice
        self.sparse_embed = sparse_embed
        self.exposed_comm_events = None  # (start, end) events around the wait for the collectives in the last step (N > 1)
        self.opt_stream = torch.cuda.Stream(device=self.dev)
        self.overlap_optimizer = True  # False: the compute stream waits for the update inside step() (A/B arm)
        self._grads_cleared = False
        self.training = True
        self.broadcast_parameters()

    def broadcast_parameters(self):
        """Replicas start from rank 0's values (DeepSpeed broadcasts the module's parameters at initialize() and after a
        checkpoint load): the fp32 master copy and the bf16 trainable tensors; frozen tensors come from the same checkpoint /
        initialiser on every rank and are not sent."""
        if self.world <= 1:
            return
        dist.broadcast(self.master, src=0)
        for _, _, _, w in self.slices:
            dist.broadcast(w, src=0)
        # values derived from trainable tensors at build time (the SAM adapters' fp32 alpha scalars) follow the new weights;
        # every other trainable tensor is read in place
        self.module.sam.refresh_adapter_scalars()

    # ---- DeepSpeed-engine surface
    def __call__(self, **batch):
        if self.micro == 0 and not self._grads_cleared:
            self.module.zero_grad()
        self._grads_cleared = False
        last_micro = self.micro + 1 >= self.args.grad_accumulation_steps
        # (with gradient accumulation the touched rows are a union over micro-steps: the dense exchange serves that case)
        self.module._sparse_embed = self.exchange if (self.exchange is not None and self.sparse_embed and last_micro and
                                                      self.args.grad_accumulation_steps == 1 and self.module._train_mode) else None
        return self.module(**batch)

    def train(self):
        self.training = True
        return self

    def eval(self):
        self.training = False
        return self

    def backward(self, loss):
        last_micro = self.micro + 1 >= self.args.grad_accumulation_steps
        # on the micro-step that completes the accumulation the model reports every finished gradient group to the exchange
        self.module._grad_ready_cb = self.exchange.ready if (self.exchange is not None and self.overlap and last_micro) else None
        try:
            self.module.backward(loss)
        finally:
            self.module._grad_ready_cb = None
        self.micro += 1

    def _allreduce(self):
        """Wait for (and, for groups not handed over from inside the backward, start) the gradient collectives. The two events
        bracket what the compute stream WAITS here = the exposed communication of the step (bench.py prints it for N > 1)."""
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        self.exchanged_ranges = self.exchange.finish()
        e1.record()
        self.exposed_comm_events = (e0, e1)

    def exposed_comm_ms(self):
        """Device time the compute stream spent waiting for gradient collectives (+ widening the wire buffer) in the last step."""
        if self.exposed_comm_events is None:
            return None
        self.exposed_comm_events[1].synchronize()
        return self.exposed_comm_events[0].elapsed_time(self.exposed_comm_events[1])

    def step(self):
        a = self.args
        if self.micro < a.grad_accumulation_steps:
            return
        self.micro = 0
        if self.world > 1:
            self._allreduce()
        g = self.module._flat_grad
        scale = 1.0 / (self.world * a.grad_accumulation_steps)
        self.global_step += 1
        lr = self.scheduler.for_update(self.global_step)
        # The update runs on its own stream. It is HBM-bound (5.8 GB of optimizer state, ~3 ms) and touches only the trainable
        # tensors, while the next step starts with the frozen CLIP tower and SAM blocks 0-7: the model's forward waits for
        # `weights_ready` exactly where it first reads a trainable tensor (projector after the CLIP tower, first SAM adapter), so
        # the update hides under the next step's first GEMMs. The gradient buffer is cleared on the same stream, behind the update.
        main = torch.cuda.current_stream(self.dev)
        self.opt_stream.wait_stream(main)
        with torch.cuda.stream(self.opt_stream):
            # global-norm clipping at 1.0: the sum of squares stays on the device and the AdamW kernel derives the clip factor from
            # it (no host read-back in the step: the host keeps queueing the next step's launches while this one runs)
            self._sumsq.zero_()
            ops.sumsq(g, out=self._sumsq)
            # one multi-tensor launch (DeepSpeed's FusedAdam does the same): 112 per-tensor launches left 0.9 ms of gaps per step
            ops.adamw_step_multi(self.master, g, self.m, self.v, self._seg_off, self._seg_len, self._seg_ptr, lr, a.beta1, a.beta2, 1e-8,
                                 a.wd, scale, self.global_step, sumsq=self._sumsq, clip=self.clip, norm_out=self._norm)
            self.module.sam.refresh_adapter_scalars()
            g.zero_()
            self._grads_cleared = True
        ev = torch.cuda.Event()
        ev.record(self.opt_stream)
        self.module.set_weights_event(ev)
        if not self.overlap_optimizer:
            main.wait_event(ev)

    @property
    def last_grad_norm(self):
        """Pre-clip global gradient norm of the last step (a device scalar; reading it synchronises)."""
        self.opt_stream.synchronize()
        return float(self._norm[0])

    def save_checkpoint(self, save_dir, tag=None, consolidated=True):
        """Engine state for resume (`<tag>.pt` + `latest`, like DeepSpeed's tag directory) AND, beside it, the consolidated fp32
        `pytorch_model.bin` under the reference's key names — what `zero_to_fp32.py ./ pytorch_model.bin` makes out of a DeepSpeed
        checkpoint (infer_eval_iground.sh:11-15), so the reference's inference scripts read this directory without that step."""
        self.module.wait_weights()  # the last update may still be running on the optimizer stream
        if self.rank == 0:
            os.makedirs(save_dir, exist_ok=True)
            tag = tag or f"global_step{self.global_step}"
            torch.save({"module": {k: v.cpu() for k, v in self.module.state_dict().items()}, "master": self.master.cpu(),
                        "exp_avg": self.m.cpu(), "exp_avg_sq": self.v.cpu(), "global_step": self.global_step,
                        "trainable": list(self.names), "world": self.world},
                       os.path.join(save_dir, tag + ".pt"))
            if consolidated:
                from .checkpoint import save_grove_weights
                save_grove_weights(self.module, os.path.join(save_dir, "pytorch_model.bin"), engine=self)
            with open(os.path.join(save_dir, "latest"), "w") as f:
                f.write(tag)
        if dist.is_initialized():
            dist.barrier()

    def load_checkpoint(self, load_dir):
        """Resume: every rank reads the file (one node, page cache), validates it against THIS engine's layout, and the replicas
        are re-synchronised from rank 0 afterwards."""
        with open(os.path.join(load_dir, "latest")) as f:
            tag = f.readlines()[0].strip()
        ck = torch.load(os.path.join(load_dir, tag + ".pt"), map_location="cpu", weights_only=True)
        if ck["master"].numel() != self.master.numel() or list(ck.get("trainable", self.names)) != list(self.names):
            raise RuntimeError(f"{load_dir}/{tag}.pt was written for another trainable set / architecture "
                               f"({ck['master'].numel()} vs {self.master.numel()} optimizer elements)")
        self.module.wait_weights()
        self.module.load_state_dict(ck["module"])
        self.master.copy_(ck["master"])
        self.m.copy_(ck["exp_avg"])
        self.v.copy_(ck["exp_avg_sq"])
        self.global_step = int(ck["global_step"])
        self.broadcast_parameters()
        if dist.is_initialized():
            dist.barrier()
        return load_dir, {"global_step": self.global_step}


class AverageMeter:
    """utils/utils.py:22-77 (sum/count meter; all_reduce folds every meter of a log step into ONE collective
    in train() below instead of one blocking all-reduce per meter)."""

    def __init__(self, name, fmt=":f"):
        self.name, self.fmt = name, fmt
        self.reset()

    def reset(self):
        self.val = self.avg = self.sum = self.count = 0.0

    def update(self, val, n=1):
        self.val = val
        self.sum += val * n
        self.count += n
        self.avg = self.sum / self.count


def train(data_iter, engine: GroveEngine, epoch, args, log=print):
    """train.py:704-793: the hot loop. data_iter yields the collate dict (dataset/dataset.py:64-70)."""
    trackers = {k: AverageMeter(k) for k in ("loss", "ce_loss", "giou_loss", "l1_loss", "temp_objectness_loss")}
    batch_time = AverageMeter("Time")
    engine.train()
    end = time.time()
    for global_step in range(args.steps_per_epoch):
        for _ in range(args.grad_accumulation_steps):
            batch = next(data_iter)
            out = engine(**batch)
            vals = torch.stack([out[k].float() for k in trackers if k in out]).cpu()  # one D2H copy per micro-step
            for (k, tr), v in zip([(k, t) for k, t in trackers.items() if k in out], vals.tolist()):
                tr.update(v, 1)
            engine.backward(out["loss"])
            engine.step()
        batch_time.update(time.time() - end)
        end = time.time()
        if global_step % args.print_freq == 0:
            if engine.world > 1:
                t = torch.tensor([x for tr in trackers.values() for x in (tr.sum, tr.count)], dtype=torch.float32, device=engine.dev)
                dist.all_reduce(t)
                t = t.tolist()
                for i, tr in enumerate(trackers.values()):
                    tr.sum, tr.count = t[2 * i], t[2 * i + 1]
                    tr.avg = tr.sum / (tr.count + 1e-5)
            if engine.rank == 0:
                log(f"Epoch: [{epoch}][{global_step + 1}/{args.steps_per_epoch}] time {batch_time.avg:.3f} " +
                    " ".join(f"{k} {tr.avg:.4f}" for k, tr in trackers.items()))
            for tr in trackers.values():
                tr.reset()
    return data_iter


@torch.no_grad()
def validate_model_performance(val_iter, engine: GroveEngine, n_batches, args):
    """train.py:876-916 (the live loss-validation branch; the bbox branch cannot run in the reference, quirk Q8)."""
    meters = {k: AverageMeter(k) for k in ("loss", "ce_loss", "giou_loss", "l1_loss", "temp_objectness_loss")}
    engine.eval()
    was = engine.module._train_mode
    engine.module._train_mode = False  # loss only: no tape, no saved activations, no dlogits
    try:
        for _ in range(n_batches):
            out = engine.module(**next(val_iter))
            for k, m in meters.items():
                if k in out:
                    m.update(float(out[k]), 1)
    finally:
        engine.module._ctx = None
        engine.module._train_mode = was
    if engine.world > 1:  # the reference all-reduces every meter (utils/utils.py:72); here all of them in ONE collective
        t = torch.tensor([x for m in meters.values() for x in (m.sum, m.count)], dtype=torch.float32, device=engine.dev)
        dist.all_reduce(t)
        t = t.tolist()
        for i, m in enumerate(meters.values()):
            m.sum, m.count = t[2 * i], t[2 * i + 1]
            m.avg = m.sum / max(m.count, 1e-5)
    return {k: m.avg for k, m in meters.items()}


def save_checkpoint(engine: GroveEngine, args, epoch, metric_name, metric_value, is_best):
    """train.py:685-701: only improving checkpoints are kept."""
    if is_best:
        save_dir = os.path.join(args.log_dir, "ckpt_model_best")
        engine.save_checkpoint(save_dir)


def synthetic_loader(dims, args, device, rank, world, seed0=0):
    """Stand-in for the DataLoader over HowToGround / iGround with a DistributedSampler (train.py:440-463): an endless iterator of
    collate dicts (dataset/dataset.py:64-70) whose clip indices are sharded over the ranks like `shard_clips`. Datasets, video
    decoding and the tokenizer are out of scope; the content is synthetic, the shapes are the real ones."""
    from .synthetic import synthetic_batch
    step = 0
    while True:
        clip0 = (step * world + rank) * args.batch_size
        b = synthetic_batch(dims, B=args.batch_size, T=args.num_frames, L=args.text_len, n_det=args.n_det, seed=seed0 + clip0,
                            device=device, dtype=torch.bfloat16)
        yield b.as_kwargs(inference=False)
        step += 1


def resume_training_from_checkpoint(engine, args, log=print):
    """train.py:489-500: --auto_resume picks `<log_dir>/ckpt_model_last_epoch` / `ckpt_model_best` when present, --resume names a
    directory; the epoch to continue from is derived from the restored step count."""
    path = args.resume
    if not path and args.auto_resume:
        for name in ("ckpt_model_last_epoch", "ckpt_model_best"):
            cand = os.path.join(args.log_dir, name)
            if os.path.exists(os.path.join(cand, "latest")):
                path = cand
                break
    if path:
        engine.load_checkpoint(path)
        args.start_epoch = engine.global_step // max(args.steps_per_epoch, 1)
        log(f"Resume training from {path}, start from epoch {args.start_epoch}")


def main(args, dims=None, log=print):
    """train.py:609-680 for the hot path: model -> (optional fine-tune weights) -> engine -> resume -> epochs of
    train() + loss validation + keep-the-best checkpointing, one process per GPU (launch with
    `python -m torch.distributed.run --nproc-per-node N -m grove_amd.train ...`, or one plain process for N = 1)."""
    from .synthetic import FULL, TINY, synthetic_state_dict
    world = int(os.environ.get("WORLD_SIZE", "1"))
    device = torch.device("cuda", args.local_rank)
    torch.cuda.set_device(device)
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        backend = os.environ.get("GROVE_BACKEND", "nccl")  # "nccl" IS RCCL on ROCm; gloo only for one-GPU rehearsals
        dist.init_process_group(backend, **({"device_id": device} if backend == "nccl" else {}))
    rank = dist.get_rank() if dist.is_initialized() else 0
    if dims is None:
        dims = FULL if args.dims == "full" else TINY
    if args.grove_weights:
        log(f"Fine-tuning using GROVE weights from {args.grove_weights}.")
        from .checkpoint import dims_from_checkpoint, read_state_dict
        sd = read_state_dict(args.grove_weights)
        dims = dims_from_checkpoint(args.grove_weights, sd, base=dims)
        model = initialize_model(args, dims, state_dict=None, device=device)
        from .checkpoint import load_grove_weights
        rep = load_grove_weights(model, args.grove_weights, sd=sd, log=log if rank == 0 else (lambda m: None))
        if rank == 0:
            log(f"missing keys: {len(rep.missing_keys)} ({len(rep.initialised)} trainable ones initialised like the reference's "
                f"constructors, {len(rep.missing_frozen)} frozen left zero), unexpected keys: {len(rep.unexpected_keys)}")
        del sd
    else:
        sd = synthetic_state_dict(dims, device=device, dtype=torch.bfloat16)  # no checkpoints offline: deterministic random init
        model = initialize_model(args, dims, state_dict=sd, device=device)
        del sd
    prepare_model_for_training(model)
    engine = GroveEngine(model, args)
    resume_training_from_checkpoint(engine, args, log)
    train_iter = synthetic_loader(dims, args, device, rank, world, seed0=0)
    val_iter = synthetic_loader(dims, args, device, rank, world, seed0=10 ** 6)
    if args.eval_only:
        val = validate_model_performance(val_iter, engine, args.val_batches, args)
        if rank == 0:
            log(f"Validation: {val}")
        return val
    best_val_loss = float("inf")
    val = None
    for epoch in range(args.start_epoch, args.epochs):
        train_iter = train(train_iter, engine, epoch, args, log)
        if dist.is_initialized():
            dist.barrier()
        val = validate_model_performance(val_iter, engine, args.val_batches, args)
        cur = val["loss"]
        is_best = cur < best_val_loss
        best_val_loss = min(cur, best_val_loss)
        if rank == 0:
            log(f"Epoch: {epoch}, Current Validation Loss: {cur:.4f}, Best Validation Loss: {best_val_loss:}")
        save_checkpoint(engine, args, epoch, "loss", f"{cur:.4f}", is_best)
    return {"best_val_loss": best_val_loss, "last_val": val, "global_step": engine.global_step}


if __name__ == "__main__":
    import sys
    _args = parse_args(sys.argv[1:])
    _args.local_rank = int(os.environ.get("LOCAL_RANK", _args.local_rank))
    torch.manual_seed(42)
    main(_args)
    if dist.is_initialized():
        dist.destroy_process_group()
