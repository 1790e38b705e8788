"""Reference-checkpoint import / export (grove_amd/checkpoint.py): host logic on the CPU.

The position-table resizing is checked against outputs of the reference's own functions (tests/golden/posembed_resize_seed5.npz,
made by oracle/refgen/make_posembed_golden.py); the readers against files written here in each format the reference's
scripts produce (flat .bin, DeepSpeed-style {"module": ...}, LoRA-prefixed keys, HF sharded directory, safetensors)."""
import json
import os

import numpy as np
import pytest
import torch

from grove_amd import checkpoint as ck

GOLD = os.path.join(os.path.dirname(__file__), "golden", "posembed_resize_seed5.npz")


@pytest.mark.parametrize("tag", ["sam", "small", "up"])
def test_position_table_resize_matches_reference(tag):
    z = np.load(GOLD)
    gs, target, patch, c, hd = (int(v) for v in z[tag + "_cfg"])
    out = ck.resize_abs_pos_embedding(torch.from_numpy(z[tag + "_pos_in"]), target, patch)
    assert tuple(out.shape) == (1, target // patch, target // patch, c)
    assert torch.equal(out, torch.from_numpy(z[tag + "_pos_out"]))  # the same torch op sequence: bit-exact
    oh, ow = ck.resize_rel_pos_embedding(torch.from_numpy(z[tag + "_relh_in"]), torch.from_numpy(z[tag + "_relw_in"]), target, patch)
    assert torch.equal(oh, torch.from_numpy(z[tag + "_relh_out"])) and torch.equal(ow, torch.from_numpy(z[tag + "_relw_out"]))


def _toy_sd(seed=0):
    g = torch.Generator().manual_seed(seed)
    p = "model.grounding_encoder.image_encoder."
    return {"model.layers.0.mlp.up_proj.weight": torch.randn(6, 4, generator=g), "lm_head.weight": torch.randn(5, 4, generator=g),
            p + "pos_embed": torch.randn(1, 8, 8, 4, generator=g), p + "blocks.1.attn.rel_pos_h": torch.randn(15, 2, generator=g),
            p + "blocks.1.attn.rel_pos_w": torch.randn(15, 2, generator=g), p + "blocks.0.attn.rel_pos_h": torch.randn(5, 2, generator=g)}


def _same(a, b):
    assert set(a) == set(b)
    for k in a:
        assert torch.equal(a[k], b[k]), k


def test_read_flat_bin_deepspeed_and_lora_prefix(tmp_path):
    sd = _toy_sd()
    torch.save(sd, tmp_path / "pytorch_model.bin")
    _same(ck.read_state_dict(str(tmp_path / "pytorch_model.bin")), sd)
    torch.save({"module": {"module." + k: v for k, v in sd.items()}, "global_step": 3}, tmp_path / "mp_rank_00_model_states.pt")
    _same(ck.read_state_dict(str(tmp_path / "mp_rank_00_model_states.pt")), sd)
    torch.save({"base_model.model." + k: v for k, v in sd.items()}, tmp_path / "lora.bin")  # infer_iground.py:530-535
    _same(ck.read_state_dict(str(tmp_path / "lora.bin")), sd)
    with pytest.raises(ValueError):
        torch.save({"a": 1}, tmp_path / "bad.bin")
        ck.read_state_dict(str(tmp_path / "bad.bin"))


def test_read_hf_sharded_directory_and_safetensors(tmp_path):
    from safetensors.torch import save_file
    sd = _toy_sd(1)
    keys = sorted(sd)
    d1 = tmp_path / "bin_shards"
    d1.mkdir()
    shards = {"pytorch_model-00001-of-00002.bin": keys[:3], "pytorch_model-00002-of-00002.bin": keys[3:]}
    for f, ks in shards.items():
        torch.save({k: sd[k] for k in ks}, d1 / f)
    json.dump({"metadata": {}, "weight_map": {k: f for f, ks in shards.items() for k in ks}}, open(d1 / "pytorch_model.bin.index.json", "w"))
    _same(ck.read_state_dict(str(d1)), sd)
    d2 = tmp_path / "st"
    d2.mkdir()
    save_file({k: v.contiguous() for k, v in sd.items()}, str(d2 / "model.safetensors"))
    _same(ck.read_state_dict(str(d2)), sd)
    with pytest.raises(FileNotFoundError):
        (tmp_path / "empty").mkdir()
        ck.read_state_dict(str(tmp_path / "empty"))


def test_interpolate_only_what_does_not_fit():
    sd = _toy_sd(2)
    before = {k: v.clone() for k, v in sd.items()}
    p = "model.grounding_encoder.image_encoder."
    changed = ck.interpolate_positional_embeddings(sd, img_size=64, patch_size=16, global_blocks=(1,))  # grid 8 -> 4
    assert set(changed) == {p + "pos_embed", p + "blocks.1.attn.rel_pos_h", p + "blocks.1.attn.rel_pos_w"}
    assert tuple(sd[p + "pos_embed"].shape) == (1, 4, 4, 4) and tuple(sd[p + "blocks.1.attn.rel_pos_h"].shape) == (7, 2)
    assert torch.equal(sd[p + "blocks.0.attn.rel_pos_h"], before[p + "blocks.0.attn.rel_pos_h"])  # a window block: untouched
    assert ck.interpolate_positional_embeddings(sd, img_size=64, patch_size=16, global_blocks=(1,)) == []  # idempotent


def test_missing_trainable_keys_get_constructor_init_not_zeros():
    """a base LLaVA checkpoint has no text_hidden_fcs / SAM adapters / box heads; they must come out like the
    reference's freshly constructed modules (nn.Linear default U(+-1/sqrt(fan_in)), alpha 0), never all-zero, and be reported."""
    import math
    from types import SimpleNamespace
    from grove_amd.model.GROVE import trainable_names
    from grove_amd.synthetic import TINY, param_shapes, synthetic_state_dict
    d = TINY
    shapes = param_shapes(d)
    full = synthetic_state_dict(d)
    drop = [n for n in shapes if n.startswith("model.text_hidden_fcs.") or "adapters." in n or "bbox_prediction_head" in n
            or n.endswith("layers.0.input_layernorm.weight")]
    ckpt = {k: v for k, v in full.items() if k not in drop}
    loaded = {}

    class FakeModel:  # the host-side surface load_grove_weights uses (the real model needs a GPU)
        dims = d
        trainable = trainable_names(d)

        def state_dict(self):
            return {n: torch.zeros(s) for n, s in shapes.items()}

        def load_state_dict(self, sd, strict=False):
            loaded.update(sd)
            return SimpleNamespace(missing_keys=[n for n in shapes if n not in sd], unexpected_keys=[n for n in sd if n not in shapes])

    msgs = []
    rep = ck.load_grove_weights(FakeModel(), "<memory>", sd=ckpt, log=msgs.append)
    train = set(trainable_names(d))
    assert set(rep.initialised) == {n for n in drop if n in train}
    assert rep.missing_frozen == [n for n in shapes if n in drop and n not in train]  # the frozen norm weight + CLIP adapters: reported
    assert len(msgs) == 2 and "constructor" in msgs[0] and "FROZEN" in msgs[1]
    w = loaded["model.text_hidden_fcs.0.0.weight"]
    bound = 1.0 / math.sqrt(w.shape[1])
    assert w.abs().max() <= bound and w.std() > 0.5 * bound / math.sqrt(3) and abs(w.mean()) < 0.1 * bound
    b = loaded["model.text_hidden_fcs.0.0.bias"]
    assert b.abs().max() <= bound and b.abs().max() > 0
    a = loaded["model.grounding_encoder.image_encoder.adapters.0.alpha"]
    assert torch.equal(a, torch.zeros_like(a))                                   # image_encoder.py:45
    cw = loaded["model.grounding_encoder.image_encoder.adapters.0.conv3d.weight"]
    assert cw.abs().max() <= 1.0 / math.sqrt(cw.shape[1] * 27) and cw.abs().max() > 0
    # deterministic per name: every rank builds the same tensors
    rep2 = ck.init_missing_trainable(FakeModel(), drop)
    assert all(torch.equal(rep2[n], loaded[n]) for n in rep2)


@pytest.mark.parametrize("use_obj", [True, False])
def test_consolidated_key_set_equals_the_reference_models(use_obj):
    """a model built with use_temp_objectness=False (ANet / VidSTG, train.py:203) has NO temporal_objectness_head
    in the reference (mask_decoder.py:83-87), and infer_anet.py:556 loads `pytorch_model.bin` with strict=True — so the consolidated
    checkpoint written here must carry exactly the reference model's key set in either mode: without the head when it is off, and
    WITH the tensors of modules that are not on this path (region encoder, the prompt encoder's point / mask tables) passed through
    from the checkpoint the run started from. Key sets: tests/golden/state_dict_keys.json, made from the reference's own
    `state_dict()` by oracle/refgen/make_keys_golden.py."""
    from types import SimpleNamespace
    from grove_amd.model.GROVE import OBJ_HEAD, trainable_names
    from grove_amd.synthetic import TINY, param_shapes
    gold = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "state_dict_keys.json")))
    ref_keys = set(gold["with_objectness" if use_obj else "without_objectness"])
    d = TINY
    shapes = param_shapes(d)
    visible = {n: s for n, s in shapes.items() if use_obj or not n.startswith(OBJ_HEAD)}
    names = trainable_names(d, True, use_obj)
    assert all(not n.startswith(OBJ_HEAD) for n in names) or use_obj
    assert set(names) <= set(visible)

    class FakeModel:  # the host-side surface of GROVEForCausalLM that checkpoint.py uses (the real model needs a GPU)
        dims = d
        trainable = names
        config = SimpleNamespace(use_temp_objectness=use_obj)

        def state_dict(self):
            return {n: torch.zeros(s) for n, s in visible.items()}

        def load_state_dict(self, sd, strict=False):
            return SimpleNamespace(missing_keys=[], unexpected_keys=[n for n in sd if n not in visible])

    # the checkpoint a run starts from: the reference model's full key set of the OTHER run type too (a HowToGround pre-train carries the head)
    src = {k: torch.zeros(shapes[k]) if k in shapes else torch.zeros(3) for k in gold["with_objectness"]}
    m = FakeModel()
    ck.load_grove_weights(m, "<memory>", sd=src, log=lambda _m: None)
    out = ck.consolidated_state_dict(m)
    assert set(out) == ref_keys, (sorted(set(out) - ref_keys)[:3], sorted(ref_keys - set(out))[:3])
