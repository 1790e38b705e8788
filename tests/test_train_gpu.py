"""The training loop of grove_amd.train on the GPU at tiny dimensions: train() (train.py:704-793), the live loss-validation
branch of validate_model_performance (876-916), checkpoint save / resume (685-701) and one optimizer step against the
restated DeepSpeed AdamW + global-norm clipping on the oracle's gradients."""
import itertools
import math

import pytest
import torch

pytestmark = pytest.mark.gpu
bf = torch.bfloat16


def _engine(dev, lr=1e-3):
    from grove_amd import train as T
    from grove_amd.synthetic import TINY, synthetic_state_dict
    args = T.shipped_args()
    args.lr, args.steps_per_epoch, args.print_freq = lr, 4, 2
    model = T.initialize_model(args, dims=TINY, state_dict=synthetic_state_dict(TINY), device=dev)
    return T, args, TINY, T.GroveEngine(model, args, total_steps=1000)


def _batch(d, dev, seed):
    from grove_amd.synthetic import synthetic_batch
    kw = synthetic_batch(d, B=2, T=8, L=48, n_det=2, seed=seed, ragged=True).as_kwargs()
    for k in ("global_enc_images", "grounding_enc_images"):
        kw[k] = kw[k].to(dev).to(bf)
    for k in ("input_ids", "labels", "attention_masks", "offset"):
        kw[k] = kw[k].to(dev)
    return kw


def test_train_validate_checkpoint(dev, tmp_path):
    T, args, d, engine = _engine(dev)
    engine.scheduler.warm = 2  # (DeepSpeed's order gives updates 1 and 2 lr = 0; ramp over two steps so that 3 and 4 move the weights)
    args.log_dir = str(tmp_path)
    batch = _batch(d, dev, 1)
    logs = []
    before = T.validate_model_performance(itertools.repeat(batch), engine, 1, args)
    T.train(itertools.repeat(batch), engine, 0, args, log=logs.append)
    after = T.validate_model_performance(itertools.repeat(batch), engine, 1, args)
    assert engine.global_step == 4 and len(logs) == 2 and "ce_loss" in logs[0]
    assert all(math.isfinite(v) for v in after.values())
    assert after["loss"] < before["loss"], (before, after)  # four steps on one repeated batch must fit it better
    # checkpoint round trip: a fresh engine resumes with identical weights, optimizer state and step count
    T.save_checkpoint(engine, args, 0, "loss", after["loss"], True)
    T2, args2, _, fresh = _engine(dev)
    fresh.load_checkpoint(str(tmp_path / "ckpt_model_best"))
    assert fresh.global_step == 4
    # the directory also holds what zero_to_fp32.py would have produced (infer_eval_iground.sh:13): the reference's loader reads it
    cons = torch.load(str(tmp_path / "ckpt_model_best" / "pytorch_model.bin"), map_location="cpu", weights_only=True)
    assert set(cons) == set(engine.module.state_dict()) and all(v.dtype == torch.float32 for v in cons.values())
    assert torch.equal(fresh.master, engine.master) and torch.equal(fresh.m, engine.m) and torch.equal(fresh.v, engine.v)
    resumed = T.validate_model_performance(itertools.repeat(batch), fresh, 1, args)
    assert abs(resumed["loss"] - after["loss"]) < 1e-3 * max(1.0, abs(after["loss"]))


def test_optimizer_step_matches_restated_adamw(dev):
    """One engine step = global-norm clip (1.0) + AdamW (betas 0.9/0.95, eps 1e-8, wd 0, bias-corrected, lr from WarmupDecayLR) on
    fp32 master weights, checked parameter by parameter on the gradients the engine itself produced."""
    T, args, d, engine = _engine(dev, lr=3e-4)
    engine.scheduler.warm = 0  # no warm-up: update 1 runs at the full rate (under the shipped warm-up its lr is 0, see for_update)
    batch = _batch(d, dev, 2)
    out = engine(**batch)
    engine.backward(out["loss"])
    g = engine.module._flat_grad.clone()
    w0 = engine.master.clone()
    engine.step()
    torch.cuda.synchronize()  # the update runs on the engine's optimizer stream (it overlaps the next forward's frozen layers)
    norm = float(g.double().pow(2).sum().sqrt())
    scale = 1.0 if norm <= 1.0 else 1.0 / (norm + 1e-6)
    lr = engine.scheduler.for_update(1)
    assert lr == 3e-4
    gs = g.double() * scale
    m = (1 - args.beta1) * gs
    v = (1 - args.beta2) * gs * gs
    upd = (m / (1 - args.beta1)) / ((v / (1 - args.beta2)).sqrt() + 1e-8)
    ref = w0.double() - lr * upd
    touched = torch.zeros_like(g, dtype=torch.bool)
    for n, off, k, w in engine.slices:
        touched[off:off + k] = True
    err = (engine.master.double() - ref)[touched].abs().max().item()
    assert err <= 1e-6 + 1e-3 * lr, err
    assert abs(engine.last_grad_norm - norm) <= 1e-3 * norm
    # the bf16 working copies are the rounded masters
    n, off, k, w = engine.slices[0]
    assert torch.equal(w.reshape(-1), engine.master[off:off + k].to(bf))


@pytest.fixture(scope="module")
def rccl_world1(dev):
    """A ONE-rank process group on the real RCCL backend ("nccl" on ROCm), bound to the device like the N > 1 job's."""
    import os
    import socket

    import torch.distributed as dist
    if dist.is_initialized():
        pytest.skip("a process group already exists in this process")
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    yield dist
    dist.destroy_process_group()


def test_rccl_bucketed_allreduce_single_rank(dev, rccl_world1):
    """The GPU form of the gradient exchange (RCCL on a side stream, bucketed, compute stream waits) on a one-rank group:
    the collective is the identity, but every call of the N > 1 path — init with device_id, async all_reduce under the comm
    stream, event / stream waits — runs against the real RCCL."""
    dist = rccl_world1
    from grove_amd.train import allreduce_buckets
    g = torch.arange(1 << 20, dtype=torch.float32, device=dev)
    ref = g.clone()
    g.mul_(2.0)  # work queued on the compute stream that the comm stream must wait for
    allreduce_buckets(g, 300_000, torch.cuda.Stream(device=dev))
    g.add_(1.0)  # ... and compute that must wait for the collective
    torch.cuda.synchronize()
    assert torch.equal(g, ref * 2 + 1)
    # the bf16 wire format (the engine's default for N > 1): rounded into the comm buffer, reduced there by RCCL, widened back
    h = torch.randn(1 << 20, device=dev)
    want = h.to(bf).float()
    allreduce_buckets(h, 300_000, torch.cuda.Stream(device=dev), torch.empty(1 << 20, dtype=bf, device=dev))
    torch.cuda.synchronize()
    assert torch.equal(h, want)
    t = torch.tensor([1.0, 2.0], device=dev)
    dist.all_reduce(t)
    assert t.tolist() == [1.0, 2.0]


@pytest.mark.parametrize("overlap", [True, False])
@pytest.mark.parametrize("sparse", [True, False])
@pytest.mark.parametrize("mode,wire", [("allreduce", bf), ("allreduce", torch.float32), ("rs_ag", bf), ("rs_ag", torch.float32), ("a2a_f32", bf)])
def test_engine_step_through_real_rccl_world1(dev, rccl_world1, mode, wire, sparse, overlap):
    """the ENGINE's N > 1 step on backend nccl (= RCCL) with one rank — every collective the multi-GPU step
    issues goes through real RCCL once: broadcast_parameters (master + every trainable tensor), the one-word MAX all-reduce of the
    touched-row count, the two all_gather_into_tensor of (ids, rows), per-bucket all_reduce / reduce_scatter + all_gather /
    all_to_all_single + all_gather, from inside the backward (overlap) or after it. With one rank every collective is the identity,
    so the flat gradient after the exchange must equal, BIT FOR BIT, the gradient the backward left (fp32 wire) or its bf16 rounding
    (bf16 wire), and embed_tokens' dense slice must equal the scatter of the touched rows — which also proves that no group was
    handed to the exchange before its gradient was final. Small buckets so that every group spans several of them."""
    from grove_amd import train as T
    from grove_amd.synthetic import TINY, synthetic_state_dict
    args = T.shipped_args()
    args.lr = 1e-3
    model = T.initialize_model(args, dims=TINY, state_dict=synthetic_state_dict(TINY), device=dev)
    engine = T.GroveEngine(model, args, total_steps=1000, bucket_bytes=96 << 10, comm_dtype=wire, exchange=mode, overlap=overlap,
                           sparse_embed=sparse, force_exchange=True)
    assert engine.exchange is not None and engine.comm_stream is not None
    sent = []
    orig_sparse = engine.exchange.sparse_rows

    def spy(ids, rows, lo, hi, ld, **k):
        sent.append((ids.clone(), rows.clone(), lo, hi, ld))
        return orig_sparse(ids, rows, lo, hi, ld, **k)
    engine.exchange.sparse_rows = spy
    batch = _batch(TINY, dev, 21)
    out = engine(**batch)
    engine.backward(out["loss"])
    torch.cuda.synchronize()
    g_raw = model._flat_grad.clone()           # ready() only reads the flat buffer: still what the backward left
    engine._allreduce()                        # what step() does first: finish(), wait, widen
    torch.cuda.synchronize()
    got = model._flat_grad
    want = g_raw if wire == torch.float32 else g_raw.to(bf).float()
    if sparse:
        assert len(sent) == 1, "embed_tokens' gradient did not travel as touched rows"
        ids, rows, lo, hi, ld = sent[0]
        dense = torch.zeros((hi - lo) // ld, ld, device=dev)
        keep = ids >= 0
        dense.index_add_(0, ids[keep].long(), rows[keep] if wire == torch.float32 else rows[keep].to(bf).float())
        want = want.clone()
        want[lo:hi] = dense.view(-1)
        assert bool((rows[keep] != 0).any())
    else:
        assert not sent
    assert torch.equal(got, want), (mode, wire, sparse, overlap, float((got - want).abs().max()))
    assert sorted(engine.exchanged_ranges)[0][0] == 0 and max(hi for _, hi in engine.exchanged_ranges) == got.numel()
    assert engine.exposed_comm_ms() is not None
    if overlap:
        assert len(engine.exchanged_ranges) >= 4   # groups were handed over from inside the backward, not in one piece at the end
    # the CU reservation of the in-flight buckets is lifted again, and the update runs
    from grove_amd import _lib, ops
    assert _lib.lib().grove_gemm_persistent_blocks() == 0 and ops._pre_gemm_hook is None
    engine.exchange.sparse_rows = orig_sparse
    # ... and the whole step() runs, twice (second step: zero_grad, a new sparse_begin, the CU reservation taken and lifted again)
    engine.scheduler.warm = 0
    w0 = engine.master.clone()
    engine.step()
    out = engine(**batch)
    engine.backward(out["loss"])
    engine.step()
    torch.cuda.synchronize()
    assert engine.global_step == 2 and not torch.equal(engine.master, w0) and bool(torch.isfinite(engine.master).all())


def test_exchange_calibration_table_through_real_rccl_world1(dev, rccl_world1):
    """Round 5: the pass `bench.py --gpus N` runs before its timed region — every exchange arm {allreduce, rs_ag,
    a2a_f32} x {0, 16 CUs reserved for RCCL} driven for a few steps of the real step function through real RCCL (one rank: the
    collectives are identities), one table row per arm with ms/step, the exposed-communication time and the number of GEMM launches
    that ran under the CU cap; the engine is left on the fastest arm, with the cap and the pre-launch poll hook released."""
    from grove_amd import _lib, ops, train as T
    from grove_amd.synthetic import TINY, synthetic_state_dict
    args = T.shipped_args()
    args.lr = 1e-3
    model = T.initialize_model(args, dims=TINY, state_dict=synthetic_state_dict(TINY), device=dev)
    engine = T.GroveEngine(model, args, total_steps=1000, bucket_bytes=96 << 10, force_exchange=True)
    assert engine.exchange is not None and engine.exchange.reserve_cus == 0   # the reservation is opt-in since round 5
    batch = _batch(TINY, dev, 21)

    def step():
        out = engine(**batch)
        engine.backward(out["loss"])
        engine.step()
    table, best = T.calibrate_exchange(engine, step, steps=2)
    assert [(a["exchange"], a["reserved_cus"]) for a in table] == T.EXCHANGE_ARMS and len(table) == 6
    for a in table:
        assert a["ms_per_step"] > 0 and a["exposed_comm_ms"] is not None and a["exposed_comm_ms"] >= 0
        assert (a["gemm_launches_under_cap"] > 0) == (a["reserved_cus"] > 0), a   # the cap is active exactly in the arms that ask for it
    assert best in table and (engine.exchange.mode, engine.exchange.reserve_cus) == (best["exchange"], best["reserved_cus"])
    torch.cuda.synchronize()
    assert _lib.lib().grove_gemm_persistent_blocks() == 0 and ops._pre_gemm_hook is None
    assert bool(torch.isfinite(engine.master).all())
    assert _lib.lib().grove_gemm_persistent_blocks() == 0


def test_a2a_f32_refuses_an_fp32_wire_on_the_gpu(dev):
    from grove_amd.train import GradExchange
    with pytest.raises(ValueError, match="a2a_f32"):
        GradExchange(torch.zeros(64, device=dev), 2, 32, comm_dtype=torch.float32, mode="a2a_f32")


def test_consolidated_checkpoint_export_import(dev, tmp_path):
    """the trained model leaves as the flat fp32 `pytorch_model.bin` the reference's inference scripts read
    (zero_to_fp32's output: reference key names, canonical Conv3d layout, trainable weights from the fp32 master copy) and
    comes back through load_grove_weights — including a pre-trained-style SAM whose position tables are at twice the grid."""
    from grove_amd import checkpoint as ck
    from grove_amd.model.GROVE import trainable_names
    from grove_amd.synthetic import param_shapes, synthetic_state_dict
    T, args, d, engine = _engine(dev)
    batch = _batch(d, dev, 3)
    T.train(itertools.repeat(batch), engine, 0, args, log=lambda *_: None)
    model = engine.module
    path = str(tmp_path / "pytorch_model.bin")
    sd = ck.save_grove_weights(model, path, engine=engine)
    shapes = param_shapes(d)
    assert set(sd) == set(shapes) and all(tuple(sd[k].shape) == tuple(shapes[k]) and sd[k].dtype == torch.float32 for k in sd)
    # trainable tensors are the master copy (more bits than the bf16 working weights), and round to exactly those weights
    name = next(n for n in trainable_names(d) if n.endswith("conv3d.weight"))
    assert torch.equal(sd[name].to(bf), model.state_dict()[name].cpu())
    assert (sd[name] != sd[name].to(bf).float()).any()
    frozen = "model.layers.0.mlp.up_proj.weight"
    assert torch.equal(sd[frozen], model.state_dict()[frozen].float().cpu())
    # import into a model that starts from different weights: same validation loss as the exporter
    want = T.validate_model_performance(itertools.repeat(batch), engine, 1, args)
    other = T.initialize_model(args, d, state_dict={k: v * 0.5 for k, v in synthetic_state_dict(d).items()}, device=dev)
    rep = ck.load_grove_weights(other, path)
    assert rep.missing_keys == [] and rep.unexpected_keys == [] and rep.resized == []
    got = T.validate_model_performance(itertools.repeat(batch), T.GroveEngine(other, args, total_steps=1000), 1, args)
    assert abs(got["loss"] - want["loss"]) < 1e-3 * max(1.0, abs(want["loss"]))
    # a checkpoint saved for twice the SAM grid is resized on the way in (train.py:561-576), wrapper prefixes are dropped
    p = "model.grounding_encoder.image_encoder."
    big = dict(sd)
    g = d.sam_grid
    big[p + "pos_embed"] = torch.randn(1, 2 * g, 2 * g, d.sam_dim)
    for i in d.sam_global:
        for ax in ("h", "w"):
            k = p + f"blocks.{i}.attn.rel_pos_{ax}"
            big[k] = torch.randn(4 * g - 1, sd[k].shape[1])
    torch.save({"base_model.model." + k: v for k, v in big.items()}, str(tmp_path / "pretrained.bin"))
    rep = ck.load_grove_weights(other, str(tmp_path / "pretrained.bin"))
    assert len(rep.resized) == 1 + 2 * len(d.sam_global) and rep.missing_keys == []
    ref = ck.resize_abs_pos_embedding(big[p + "pos_embed"], d.sam_image, d.sam_patch)
    assert torch.equal(other.state_dict()[p + "pos_embed"].cpu(), ref.to(bf))
    # a tensor of the wrong shape is an error, not a silent skip
    bad = dict(sd)
    bad[frozen] = torch.zeros(3, 3)
    torch.save(bad, str(tmp_path / "bad.bin"))
    with pytest.raises(RuntimeError):
        ck.load_grove_weights(other, str(tmp_path / "bad.bin"))


@pytest.fixture
def det(dev):
    """Deterministic mode (include/grove_hip.h: grove_set_deterministic): fixed-order sums instead of fp32 atomics."""
    from grove_amd import ops
    prev = ops.set_deterministic(True)
    yield
    ops.set_deterministic(prev)


def test_tower_overlap_changes_nothing_but_time(dev, det):
    """The SAM tower on its own stream (forward and backward) and the host planning on a side stream are scheduling only: in
    deterministic mode the losses and the gradients are BIT-equal to the serial order's (a race between the streams would show as a
    difference; in the default mode the CE sum and split-K use fp32 atomics and the comparison would be to accumulation noise)."""
    T, args, d, engine = _engine(dev)
    model = engine.module
    batch = _batch(d, dev, 4)
    res = {}
    for overlap in (True, False, True):
        model.tower_overlap = overlap
        model.zero_grad()
        out = engine(**batch)
        engine.backward(out["loss"])
        torch.cuda.synchronize()
        res.setdefault(overlap, []).append(({k: float(v) for k, v in out.items() if k.endswith("loss")}, model._flat_grad.clone()))
    (l_on, g_on), (l_on2, g_on2) = res[True]
    (l_off, g_off), = res[False]
    assert l_on == l_off and l_on2 == l_off, (l_on, l_on2, l_off)
    assert g_off.abs().max().item() > 0 and torch.equal(g_on, g_off) and torch.equal(g_on2, g_off), (g_on - g_off).abs().max().item()


def test_decoder_wgrad_side_stream_changes_nothing_but_time(dev, det):
    """Round 6b: the box decoder's weight / bias gradients ride a side stream (tape.py, `decoder.wgrad_stream`), gated on events of the
    dgrad chain and joined before the group is final. Scheduling only: in deterministic mode losses and the whole flat gradient are
    BIT-equal to the in-line order's, with another stream keeping the memory system busy so that the two orders really interleave."""
    T, args, d, engine = _engine(dev)
    model = engine.module
    side = model.decoder.wgrad_stream
    assert side is not None, "the training model must own the decoder's side stream by default"
    names = [n for n in model.trainable if "mask_decoder.transformer" in n and n.endswith("weight")]
    assert names, "the decoder's transformer must be trainable here (its linears are what moves to the side stream)"
    batch = _batch(d, dev, 4)
    noise = torch.randn(16 << 20, device=dev)
    churn = torch.cuda.Stream()
    res = {}
    for on in (True, False, True, True):
        model.decoder.wgrad_stream = side if on else None
        model.zero_grad()
        out = engine(**batch)
        with torch.cuda.stream(churn):
            noise.mul_(1.0001)
        engine.backward(out["loss"])
        torch.cuda.synchronize()
        res.setdefault(on, []).append(model._flat_grad.clone())
    model.decoder.wgrad_stream = side
    g_off = res[False][0]
    lo = min(model._grad_off[n] for n in names)
    assert g_off[lo:lo + 64].abs().max().item() > 0
    for g_on in res[True]:
        assert torch.equal(g_on, g_off), (g_on - g_off).abs().max().item()


def test_tower_overlap_default_mode_agrees_to_accumulation_noise(dev):
    """The same comparison with the atomics on (the mode training runs in): agreement to accumulation-order noise."""
    T, args, d, engine = _engine(dev)
    model = engine.module
    batch = _batch(d, dev, 4)
    res = {}
    for overlap in (True, False):
        model.tower_overlap = overlap
        model.zero_grad()
        out = engine(**batch)
        engine.backward(out["loss"])
        torch.cuda.synchronize()
        res[overlap] = ({k: float(v) for k, v in out.items() if k.endswith("loss")}, model._flat_grad.clone())
    (l_on, g_on), (l_off, g_off) = res[True], res[False]
    for k in l_off:
        assert abs(l_on[k] - l_off[k]) <= 1e-5 * max(1.0, abs(l_off[k])), (k, l_on, l_off)
    scale = g_off.abs().max().item()
    assert scale > 0 and (g_on - g_off).abs().max().item() <= 1e-3 * scale


def test_train_main_entry_point(dev, tmp_path):
    """train.py:609-680 / :929-937: `main(args)` = model -> engine -> epochs of train() + loss validation + keep-the-best
    checkpoint, on the synthetic loader; then --auto_resume continues from the saved step and --eval_only validates."""
    from grove_amd import train as T
    argv = ["--lora_r", "0", "--pretrained", "--train_mask_decoder", "--dims", "tiny", "--epochs", "2", "--steps_per_epoch", "3", "--batch_size", "1", "--text_len", "40", "--n_det", "2",
            "--val_batches", "1", "--lr", "1e-3", "--log_dir", str(tmp_path), "--print_freq", "1"]
    args = T.parse_args(argv)
    logs = []
    res = T.main(args, log=logs.append)
    assert res["global_step"] == 6 and math.isfinite(res["best_val_loss"])
    assert any(s.startswith("Epoch: [0][1/3]") for s in logs) and any("Current Validation Loss" in s for s in logs)
    best = tmp_path / "ckpt_model_best"
    assert (best / "latest").exists() and (best / "pytorch_model.bin").exists()
    args2 = T.parse_args(argv + ["--auto_resume", "--eval_only"])
    logs2 = []
    val = T.main(args2, log=logs2.append)
    assert any("Resume training from" in s for s in logs2) and math.isfinite(val["loss"])
    # fine-tune start from the consolidated file the run wrote (train.py:621-624)
    args3 = T.parse_args(argv[:-4] + ["--log_dir", str(tmp_path / "ft"), "--grove_weights", str(best), "--epochs", "1"])
    res3 = T.main(args3, log=lambda *_: None)
    assert res3["global_step"] == 3


def _run_rehearsal(cmd, env):
    """Runs a two-rank one-GPU rehearsal command. These rehearsals drive gloo collectives on CUDA tensors from two processes that share
    ONE GPU (not the product path, which is RCCL with a GPU per rank). Round 3 saw one launch in a dozen suite runs never print its line
    and retried it blindly. Round 4: 50 of 50 stand-alone launches pass without a retry (tools/rehearsal_loop.py,
    profiles/r04_rehearsal_loop.json), so the stall needs the suite around it to show; every launch therefore runs under bench.py's own
    watchdogs now — a rank that sits in one stage for 60 s dumps the Python stack of every thread and exits, the self-launching parent
    prints each rank's last stage and kills exactly its own process group after 240 s — and a failed first attempt is retried ONCE
    with its whole diagnostic output written to gpurun_out/rehearsal_stall_*.txt and attached to a warning, so the next occurrence
    names the call each rank sat in instead of disappearing."""
    import os
    import signal
    import subprocess
    import time
    import types
    import warnings
    cmd = list(cmd) + ["--stage_timeout", "60", "--launch_timeout", "240"]
    for attempt in (0, 1):
        proc = subprocess.Popen(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, start_new_session=True)
        try:
            out, err = proc.communicate(timeout=330)
        except subprocess.TimeoutExpired:  # (the parent's own 240 s limit should have fired first)
            os.killpg(proc.pid, signal.SIGKILL)
            out, err = proc.communicate()
            proc.returncode = 124
        if proc.returncode == 0 or attempt:
            return types.SimpleNamespace(returncode=proc.returncode, stdout=out, stderr=err)
        root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
        os.makedirs(os.path.join(root, "gpurun_out"), exist_ok=True)
        path = os.path.join(root, "gpurun_out", f"rehearsal_stall_{int(time.time())}.txt")
        with open(path, "w") as fh:
            fh.write("CMD: " + " ".join(cmd) + f"\nRC: {proc.returncode}\n--- STDERR ---\n" + err[-100000:] + "\n--- STDOUT ---\n" + out[-5000:])
        warnings.warn(f"two-rank rehearsal failed once (rc {proc.returncode}) and is retried; diagnostics in {path}: {err[-1500:]!r}")


def test_bench_self_launches_eight_ranks(tmp_path):
    """the driver's one shot at `bench.py --gpus 8` is the first time eight ranks ever run this code — so they run
    here first: eight processes on this box's one GPU (tiny dims; gloo, because RCCL refuses two ranks on one device), the self-launch,
    per-rank batches, the world-8 bucket / chunk arithmetic of every exchange arm in the calibration table, the barrier + MAX timing and
    the single JSON line (train_scripts/train_howtoground.sh:20-28: 8 ranks per node)."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, GROVE_BENCH_BACKEND="gloo", GROVE_BENCH_ONE_GPU="1", GLOO_SOCKET_IFNAME="lo", OMP_NUM_THREADS="2")
    env.pop("WORLD_SIZE", None)
    p = _run_rehearsal([sys.executable, os.path.join(root, "bench.py"), "--gpus", "8", "--dims", "tiny", "--steps", "2", "--warmup", "1", "--frames", "8",
                        "--text_len", "48", "--no_cpu_baseline", "--calibration_steps", "1"], env)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, p.stdout
    res = json.loads(lines[0])
    assert res["n_gpus"] == 8 and res["config"]["ranks"] == 8 and res["config"]["global_batch_clips"] == 16 and res["config"]["collective_backend"] == "gloo"
    assert res["value"] > 0 and res["scaling"] == "weak" and res["config"]["exposed_comm_ms"] is not None
    cal = res["config"]["exchange_calibration"]
    assert cal is not None and len(cal["arms"]) == 6 and all(a["ms_per_step"] > 0 for a in cal["arms"])
    print(json.dumps({"ms_per_step": res["ms_per_step"], "arms": [(a["exchange"], a["reserved_cus"], a["ms_per_step"]) for a in cal["arms"]]}))


def test_bench_self_launches_ranks(tmp_path):
    """`python bench.py --gpus 2` with no launcher around it (the shape of the driver's command) spawns its own two ranks before
    touching the GPU and prints ONE JSON line with n_gpus = 2. On this one-GPU box both ranks share cuda:0 and the collectives go
    through gloo (RCCL refuses two ranks on one device); on an N-GPU node the same path runs RCCL."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, GROVE_BENCH_BACKEND="gloo", GROVE_BENCH_ONE_GPU="1", GLOO_SOCKET_IFNAME="lo")
    env.pop("WORLD_SIZE", None)
    base = [sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--dims", "tiny", "--steps", "2", "--warmup", "1",
            "--frames", "8", "--text_len", "48", "--no_cpu_baseline"]

    def run(extra):
        return _run_rehearsal(base + extra, env)

    p = run([])
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, p.stdout
    res = json.loads(lines[0])
    assert res["n_gpus"] == 2 and res["config"]["ranks"] == 2 and res["config"]["collective_backend"] == "gloo"
    assert res["value"] > 0 and res["steps"] == 2 and res["scaling"] == "weak"
    # the exchange variants reach the same loss: group-by-group from inside the backward (default), after the backward, and the
    # reduce-scatter + all-gather form
    losses = []  # (the calibrated run has taken 24 more optimizer steps by its last loss: it is not part of this comparison)
    assert res["config"]["exposed_comm_ms"] is not None and "touched rows" in res["config"]["gradient_exchange"]
    # round 5: the one-shot run explains itself — the calibration pass timed every exchange arm before the timed region
    cal = res["config"]["exchange_calibration"]
    assert cal is not None and len(cal["arms"]) == 6 and cal["fastest"] in cal["arms"]
    assert {(a["exchange"], a["reserved_cus"]) for a in cal["arms"]} == {(m, r) for m in ("allreduce", "rs_ag", "a2a_f32") for r in (0, 16)}
    assert all(a["ms_per_step"] > 0 and a["exposed_comm_ms"] is not None for a in cal["arms"])
    # the line was timed BEFORE the calibration on all-reduce, or re-timed on an arm that beat it by more than 1 %
    assert res["config"]["gradient_exchange"].startswith(cal["line_timed_on"]["exchange"])
    assert cal["line_timed_on"]["exchange"] == "allreduce" or "start_arm_line" in cal
    # a wedged collective inside the calibration cannot lose the line: arm 1 never returns, nothing moves for 4 s -> rank 0 prints the
    # line it measured before the calibration, marked, and every rank leaves with exit code 0
    st = _run_rehearsal(base, dict(env, GROVE_BENCH_STALL_S="4", GROVE_BENCH_TEST_STALL_ARM="1"))
    assert st.returncode == 0, st.stderr[-2000:]
    sl = [ln for ln in st.stdout.splitlines() if ln.strip()]
    assert len(sl) == 1, st.stdout
    sres = json.loads(sl[0])
    assert sres["value"] > 0 and sres["n_gpus"] == 2
    assert sres["config"]["exchange_calibration"]["status"].startswith("STALLED in 'allreduce, 16 CUs reserved'"), sres["config"]["exchange_calibration"]
    for extra in ([], ["--no_comm_overlap"], ["--exchange", "rs_ag"], ["--exchange", "a2a_f32"], ["--dense_embed"]):
        q = run(extra + ["--no_calibration"])
        assert q.returncode == 0, q.stderr[-2000:]
        losses.append(json.loads([ln for ln in q.stdout.splitlines() if ln.strip()][0])["config"]["last_loss"])
    assert max(losses) - min(losses) <= 2e-3 * max(1.0, abs(losses[0])), losses


def test_bench_infer_mode_runs_on_two_ranks(tmp_path):
    """BASELINE config 5 is an 8-GPU inference job: `bench.py --mode infer --gpus 2` launches its own ranks (replicas: clips sharded
    by rank, no data-path collective), times under barrier + max over ranks and ends with the reference's all_gather_object of the
    per-clip results (infer_iground.py:290-293). One-GPU rehearsal over gloo, fp8 and bf16."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, GROVE_BENCH_BACKEND="gloo", GROVE_BENCH_ONE_GPU="1", GLOO_SOCKET_IFNAME="lo")
    env.pop("WORLD_SIZE", None)
    for dtype in ("bf16",):
        p = _run_rehearsal([sys.executable, os.path.join(root, "bench.py"), "--mode", "infer", "--dtype", dtype, "--gpus", "2", "--dims", "tiny",
                            "--steps", "2", "--warmup", "1", "--frames", "16", "--batch", "1", "--text_len", "48", "--no_cpu_baseline"], env)
        assert p.returncode == 0, p.stderr[-2000:]
        lines = [ln for ln in p.stdout.splitlines() if ln.strip()]
        assert len(lines) == 1, p.stdout
        res = json.loads(lines[0])
        assert res["n_gpus"] == 2 and res["config"]["ranks"] == 2 and res["config"]["ranks_gathered"] == 2
        assert res["config"]["collective_backend"] == "gloo" and res["value"] > 0 and res["dtype"] == dtype
    # config 2 (round 5): the infer_iground flow — clip-batched centre `evaluate` + the other windows — as replicas on two ranks
    p = _run_rehearsal([sys.executable, os.path.join(root, "bench.py"), "--mode", "infer_iground", "--gpus", "2", "--dims", "tiny", "--steps", "1", "--warmup", "1",
                        "--frames", "24", "--clips", "2", "--clips_per_batch", "2", "--max_new_tokens", "4", "--no_cpu_baseline"], env)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, p.stdout
    res = json.loads(lines[0])
    assert res["n_gpus"] == 2 and res["value"] > 0 and res["config"]["clips_per_batch"] == 2 and res["config"]["speedup_over_batch1"] > 0


def _three_steps(dev, overlap=True):
    T, args, d, engine = _engine(dev)
    engine.scheduler.warm = 0
    engine.overlap_optimizer = overlap
    losses = []
    for s in range(3):
        out = engine(**_batch(d, dev, 10 + s))
        losses.append(out["loss"])
        engine.backward(out["loss"])
        engine.step()
    torch.cuda.synchronize()
    return torch.stack(losses).cpu(), engine.master.clone().cpu()


def test_optimizer_stream_overlap_is_race_free(dev, det):
    """The update runs on its own stream and the next forward waits for it only where it first reads a trainable tensor. In
    deterministic mode three steps with the overlap give BIT-identical weights and losses to three steps with the compute stream
    waiting inside step() (rounds 2-3 compared to run-to-run atomics noise, 3e-3 on the weights, which a small race hides in)."""
    l_a, w_a = _three_steps(dev, True)
    l_b, w_b = _three_steps(dev, False)
    l_c, w_c = _three_steps(dev, True)
    assert torch.equal(l_a, l_b) and torch.equal(l_a, l_c), (l_a, l_b, l_c)
    assert torch.equal(w_a, w_b) and torch.equal(w_a, w_c), ((w_a - w_b).abs().max().item(), (w_a - w_c).abs().max().item())


def test_training_steps_repeat_bit_for_bit(dev, det):
    """Two engines from the same seed, three steps each with every overlap on: identical losses and master weights."""
    (l0, w0), (l1, w1) = _three_steps(dev), _three_steps(dev)
    assert torch.equal(l0, l1), (l0, l1)
    assert torch.equal(w0, w1), (w0 - w1).abs().max().item()


def test_optimizer_stream_overlap_default_mode(dev):
    """The default (atomics) mode of the same comparison: run-to-run noise of one configuration is the yardstick (tools/overlap_stress.py,
    12 runs per arm: the third loss spreads over 9.1538 .. 9.1579 in BOTH arms, final weights of two runs differ by 2.9-3.3e-3)."""
    l_a, w_a = _three_steps(dev, True)
    l_b, w_b = _three_steps(dev, False)
    l_c, w_c = _three_steps(dev, True)
    noise = max((w_a - w_c).abs().max().item(), 1e-7)
    assert (w_a - w_b).abs().max().item() <= 4 * noise + 1e-6, ((w_a - w_b).abs().max().item(), noise)
    assert (l_a - l_b).abs().max().item() <= 1e-3 * l_b.abs().max().item() + 4 * (l_a - l_c).abs().max().item(), (l_a, l_b, l_c)