"""Container-only harness that IMPORTS the reference (/root/reference) to generate golden vectors.

TEST INFRASTRUCTURE — never imported by grove_amd, bench.py's timed path or the GPU tests; it
cannot run on the GPU box (/root/reference does not exist there). It follows SURVEY.md Appendix A:
stub packages for modules the reference only constructs (mmdet/mmcv/mmengine/bleach/ffmpeg),
a restated torchvision GIoU, offline construction of the CLIP tower, and `.cuda()` neutralised
for CPU. Nothing from the reference is copied: the modules are imported where they lie.
"""
import importlib.util
import os
import sys
import types

sys.dont_write_bytecode = True
HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
REF = os.environ.get("GROVE_REFERENCE", "/root/reference")

import torch  # noqa: E402
import transformers  # noqa: E402,F401  (must be imported BEFORE a fake torchvision appears)
from transformers import CLIPVisionConfig  # noqa: E402

if REPO not in sys.path:
    sys.path.insert(0, REPO)
from grove_amd.synthetic import GroveDims, param_shapes, synthetic_state_dict  # noqa: E402


def _install():
    stubs = os.path.join(HERE, "stubs")
    for p in (REF, stubs):
        if p not in sys.path:
            sys.path.insert(0, p)
    if "torchvision" not in sys.modules:
        tv = types.ModuleType("torchvision")
        spec = importlib.util.spec_from_file_location("torchvision.ops", os.path.join(stubs, "tv_stub", "ops.py"))
        ops = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(ops)
        tr = types.ModuleType("torchvision.transforms")
        trf = types.ModuleType("torchvision.transforms.functional")
        trf.resize = lambda *a, **k: None
        trf.to_pil_image = lambda *a, **k: None
        tr.functional = trf
        tv.ops, tv.transforms = ops, tr
        sys.modules.update({"torchvision": tv, "torchvision.ops": ops, "torchvision.transforms": tr,
                            "torchvision.transforms.functional": trf})
    torch.Tensor.cuda = lambda self, *a, **k: self  # model/GROVE.py:203,260 hard-code .cuda()


def build_reference_model(d: GroveDims, dtype=torch.float32, use_temp_objectness=True, loss_weights=(1.0, 1.0, 1.0)):
    """Builds model/GROVE.py::GROVEForCausalLM at dims `d`, SAM at 512 px (train.py:561-576
    semantics), and loads the deterministic synthetic state dict. Returns (model, state_dict)."""
    _install()
    import model.GROVE as G
    import importlib
    BS = importlib.import_module("model.SAM.build_sam")
    import model.llava.model.multimodal_encoder.clip_encoder as CE
    from model.llava.model.language_model.llava_llama import Llava1Config

    clip_cfg = CLIPVisionConfig(hidden_size=d.clip_dim, intermediate_size=d.clip_mlp, num_hidden_layers=d.clip_layers,
                                num_attention_heads=d.clip_heads, image_size=d.clip_image, patch_size=d.clip_patch,
                                hidden_act="quick_gelu", layer_norm_eps=d.clip_eps)
    CE.CLIPVisionConfig.from_pretrained = classmethod(lambda cls, *a, **k: clip_cfg)
    CE.CLIPImageProcessor.from_pretrained = classmethod(lambda cls, *a, **k: None)
    CE.CLIPVisionModel.from_pretrained = classmethod(lambda cls, *a, **k: CE.CLIPVisionModel(clip_cfg))

    def build_sam(checkpoint=None, use_temp_objectness=True):
        return BS._build_sam(encoder_embed_dim=d.sam_dim, encoder_depth=d.sam_depth, encoder_num_heads=d.sam_heads,
                             encoder_global_attn_indexes=list(d.sam_global), checkpoint=None,
                             use_temp_objectness=use_temp_objectness)
    G.build_sam_vit_h = build_sam

    cfg = Llava1Config(hidden_size=d.hidden, intermediate_size=d.mlp, num_hidden_layers=d.n_layers,
                       num_attention_heads=d.n_heads, num_key_value_heads=d.n_heads, vocab_size=d.vocab,
                       rms_norm_eps=d.rms_eps, rope_theta=d.rope_theta, max_position_embeddings=2048,
                       attn_implementation="eager", pad_token_id=d.pad_token_id, bos_token_id=d.bos_token_id,
                       eos_token_id=d.eos_token_id, tie_word_embeddings=False)
    cfg.mm_vision_tower = "openai/clip-vit-large-patch14-336"
    cfg.vision_tower = cfg.mm_vision_tower
    cfg.mm_hidden_size = d.clip_dim
    cfg.mm_vision_select_layer = -2
    cfg.mm_vision_select_feature = "patch"
    cfg.with_region = True
    cfg.num_level_reg_features = 4
    cfg.pretrain_mm_mlp_adapter = None
    cfg.mm_use_im_start_end = True
    cfg.use_temp_objectness = use_temp_objectness  # (GROVEBaseModel reads it before _set_model_configurations' value is used, GROVE.py:53)
    model = G.GROVEForCausalLM(cfg, det_token_idx=d.det_token_idx, ce_loss_weight=loss_weights[0], giou_loss_weight=loss_weights[1],
                               temp_objectness_loss_weight=loss_weights[2], out_dim=d.out_dim, num_frames=d.num_frames,
                               use_temp_objectness=use_temp_objectness, train_mask_decoder=True, with_region=True)
    model.get_model().initialize_vision_modules(model.get_model().config)
    # SAM at 512 px: absolute and global-block relative position tables at their post-interpolation size
    enc = model.get_model().grounding_encoder.image_encoder
    g = d.sam_grid
    hd = d.sam_dim // d.sam_heads
    enc.pos_embed = torch.nn.Parameter(torch.zeros(1, g, g, d.sam_dim))
    for i in d.sam_global:
        enc.blocks[i].attn.rel_pos_h = torch.nn.Parameter(torch.zeros(2 * g - 1, hd))
        enc.blocks[i].attn.rel_pos_w = torch.nn.Parameter(torch.zeros(2 * g - 1, hd))
    enc.img_size = d.sam_image
    sd = synthetic_state_dict(d)
    if not use_temp_objectness:  # the reference's decoder then has no objectness head (mask_decoder.py:83-87)
        sd = {k: v for k, v in sd.items() if "temporal_objectness_head" not in k}
    own = dict(model.state_dict())
    missing_in_model = [k for k in sd if k not in own]
    assert not missing_in_model, f"names not in the reference state dict: {missing_in_model[:5]}"
    for k, v in sd.items():
        assert tuple(own[k].shape) == tuple(v.shape), (k, tuple(own[k].shape), tuple(v.shape))
    model.load_state_dict(sd, strict=False)
    model = model.to(dtype)
    model.eval()
    return model, sd


def hot_path_names_check(d: GroveDims):
    """Every reference parameter outside the dead region encoder / dormant mask branch must be in
    param_shapes() — guards the state-dict compatibility surface."""
    model, sd = build_reference_model(d)
    ours = set(param_shapes(d))
    dead = ("region_encoder.", "point_embeddings", "not_a_point_embed", "mask_downscaling")
    theirs = [k for k in model.state_dict() if not any(s in k for s in dead)]
    return sorted(set(theirs) - ours), sorted(ours - set(theirs))
