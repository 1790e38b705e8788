"""which launch aborts after the deep-narrow generate()? (debug helper)"""
import faulthandler, os, sys
faulthandler.enable()
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from grove_amd import ops
dev = torch.device("cuda:0")
bf = torch.bfloat16
what = sys.argv[1] if len(sys.argv) > 1 else "probe"
def say(*a):
    print(*a, flush=True)
if what == "probe":
    import test_full_depth_gpu as T
    T.decode_precision_probe(dev)
    say("probe done")
elif what == "gen":
    from grove_amd import GROVEForCausalLM
    from grove_amd.synthetic import synthetic_batch, synthetic_state_dict
    import test_full_depth_gpu as T
    d = T.deep_narrow_dims()
    sd_dev = synthetic_state_dict(d, device=dev, dtype=bf)
    model = GROVEForCausalLM(dims=d, device=dev, state_dict=sd_dev, det_token_idx=d.det_token_idx, num_frames=8, pe_dtype=torch.float32)
    batch = synthetic_batch(d, B=1, T=8, L=64, n_det=1, seed=5)
    feats, _ = model(mode="encode_images", images=batch.global_enc_images.to(bf).to(dev))
    out = model.generate(input_ids=batch.input_ids[:, :40].contiguous().to(dev), image_features=feats, max_new_tokens=9, eos_token_id=-1,
                         output_hidden_states=True, return_dict_in_generate=True)
    torch.cuda.synchronize(); say("generate done")
    if len(sys.argv) > 2 and sys.argv[2] == "del":
        del model, out, feats, sd_dev
        import gc; gc.collect(); torch.cuda.empty_cache(); say("freed")
for (M, N, K) in [(1, 300, 512), (2, 4096, 4096), (3, 1000, 1024), (4, 1000, 1096), (3, 1000, 1096)]:
    x = (torch.randn(M, K) * 1.0).to(bf).to(dev); w = (torch.randn(N, K) * 0.05).to(bf).to(dev)
    torch.cuda.synchronize(); say("alloc ok", M, N, K)
    y = ops.gemv(x, w)
    torch.cuda.synchronize(); say("gemv ok", M, N, K, float(y.float().abs().max()))
say("all ok")
