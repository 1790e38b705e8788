"""The Python half of the drop-in boundary (SURVEY.md section 8(b) "train.py entry points"), host side, no GPU:
grove_amd.train.parse_args against the reference's own argparse surface (tests/golden/train_cli.json, extracted from
train.py:40-112 and train_scripts/*.sh by oracle/refgen/make_cli_golden.py), the names infer_iground.py:26-27 imports from
`train`, the tokenizer stand-in's special-token bookkeeping (train.py:124-159) and the loud rejections."""
import inspect
import json
import os

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
CLI = json.load(open(os.path.join(HERE, "golden", "train_cli.json")))


def test_every_reference_flag_is_accepted_with_its_default():
    from grove_amd import train as T
    args = T.parse_args([])
    assert len(CLI["flags"]) >= 56
    for flag, spec in CLI["flags"].items():
        dest = flag.lstrip("-").replace("-", "_")
        assert hasattr(args, dest), f"{flag} missing from grove_amd.train.parse_args"
        if flag == "--local_rank":  # default follows LOCAL_RANK here (the reference overwrites it from the env right after parsing, train.py:933)
            continue
        want = spec.get("default", False if spec.get("action") == "store_true" else None)
        assert getattr(args, dest) == want, (flag, getattr(args, dest), want)
    # every ignored flag is documented, and is a reference flag
    ref = {f.lstrip("-").replace("-", "_") for f in CLI["flags"]}
    assert set(T.IGNORED_REFERENCE_FLAGS) <= ref


@pytest.mark.parametrize("script", sorted(CLI["launch_lines"]))
def test_shipped_launch_lines_parse_and_are_supported(script):
    """train_scripts/train_howtoground.sh:35 (and the ANet / VidSTG lines) verbatim."""
    from grove_amd import train as T
    argv = CLI["launch_lines"][script]
    args = T.parse_args(argv)
    T.check_supported(args)  # --lora_r 0 --pretrained: the implemented configuration
    assert args.lora_r == 0 and args.pretrained and args.train_mask_decoder and args.lr == 5e-5
    assert args.giou_loss_weight == 2 and args.temp_objectness_loss_weight == 2 and args.epochs == 20 and args.steps_per_epoch == 350
    assert args.version == "MBZUAI/GLaMM-GranD-Pretrained" and args.grove_weights.endswith("grove_pt_howtoground1m_ckpt.bin")
    assert args.log_dir == os.path.join("/home/grove_checkpoints", args.exp_name)  # initialize_environment, train.py:116
    if script == "train_howtoground.sh":
        assert args.dataset == "HowToGround" and args.train_keys == "/home/train_keys_deduplicated.pkl"


def test_names_the_inference_scripts_import_exist_with_reference_signatures():
    import grove_amd.train as T
    for name in CLI["infer_imports"]:
        assert callable(getattr(T, name)), name
    sig = lambda f: list(inspect.signature(f).parameters)
    assert sig(T.setup_tokenizer_and_special_tokens)[0] == "args"
    assert sig(T.initialize_custom_layers_in_model) == ["model"]
    assert sig(T.initialize_custom_layers_in_global_encoder) == ["vision_tower"]
    assert sig(T.interpolate_positional_embeddings)[0] == "ds_model"
    assert sig(T.setup_lora_config) == ["model", "args"]
    assert sig(T.initialize_model)[:2] == ["args", "tokenizer"]
    assert sig(T.prepare_model_for_training)[:3] == ["model", "tokenizer", "args"]
    for name in ("train", "validate_model_performance", "save_checkpoint", "main", "parse_args", "initialize_environment",
                 "resume_training_from_checkpoint", "set_seed"):
        assert callable(getattr(T, name)), name


def test_unsupported_configurations_are_rejected_loudly():
    from grove_amd import train as T
    with pytest.raises(NotImplementedError, match="lora_r"):
        T.check_supported(T.parse_args([]))                      # the reference's DEFAULT is --lora_r 8 (peft)
    with pytest.raises(NotImplementedError, match="pretrained"):
        T.check_supported(T.parse_args(["--lora_r", "0"]))
    with pytest.raises(NotImplementedError, match="precision"):
        T.check_supported(T.parse_args(["--lora_r", "0", "--pretrained", "--precision", "fp16"]))
    with pytest.raises(NotImplementedError, match="bbox_validation"):
        T.check_supported(T.parse_args(["--lora_r", "0", "--pretrained", "--bbox_validation"]))
    with pytest.raises(NotImplementedError):
        T.setup_lora_config(None, T.shipped_args())
    T.check_supported(T.shipped_args())


def test_tokenizer_stand_in_special_tokens():
    """train.py:124-159 on the stand-in: pad = unk, <vid_start>/<vid_end> and [DET] appended, the four ids picked with the
    reference's index choices ([1] behind the "▁" piece for the pre-existing non-special tokens, [0] for [DET])."""
    from grove_amd import train as T
    args = T.shipped_args()
    tok = T.setup_tokenizer_and_special_tokens(args)
    assert tok.pad_token == tok.unk_token and tok.pad_token_id == tok.unk_token_id == 0
    assert len(tok) == 32000 + 6 + 3
    assert args.det_token_idx == len(tok) - 1 and tok("[DET]", add_special_tokens=False).input_ids == [args.det_token_idx]
    assert tok("<bbox>", add_special_tokens=False).input_ids == [29871, args.bbox_token_idx]
    assert len({args.bbox_token_idx, args.det_token_idx, args.bop_token_idx, args.eop_token_idx}) == 4
    # not --pretrained: the region / phrase tokens are added here as special tokens (one piece each)
    a2 = T.parse_args(["--lora_r", "0"])
    t2 = T.setup_tokenizer_and_special_tokens(a2, T.SyntheticTokenizer(pretrained=False))
    assert len(t2) == 32000 + 2 + 5 and a2.det_token_idx == 32000 + 4


def test_trainable_set_follows_train_mask_decoder():
    """train.py:279-288: without --train_mask_decoder only the box head and the temporal-objectness head of the decoder train."""
    from grove_amd.model.GROVE import trainable_names
    from grove_amd.model.decoder import M_
    from grove_amd.synthetic import TINY
    full, heads = trainable_names(TINY, True), trainable_names(TINY, False)
    assert set(heads) < set(full)
    dec = [n for n in heads if n.startswith(M_)]
    assert dec and all(n.startswith((M_ + "bbox_prediction_head.", M_ + "temporal_objectness_head.")) for n in dec)
    assert any(n.startswith(M_ + "transformer.") for n in full)
    assert [n for n in full if not n.startswith(M_)] == [n for n in heads if not n.startswith(M_)]


def test_bench_and_train_help_render():
    """argparse expands `%` in help strings: a literal per-cent sign in one of them breaks `--help` (round 5 had one)."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    for cmd in ([sys.executable, os.path.join(root, "bench.py"), "--help"], [sys.executable, "-m", "grove_amd.train", "--help"]):
        p = subprocess.run(cmd, cwd=root, capture_output=True, text=True)
        assert p.returncode == 0 and "usage:" in p.stdout, p.stderr[-500:]
