import os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from grove_amd import GROVEForCausalLM
from grove_amd.synthetic import TINY, synthetic_batch, synthetic_state_dict
bf = torch.bfloat16
dev = torch.device("cuda:0")
d = TINY
sd = synthetic_state_dict(d)
model = GROVEForCausalLM(dims=d, device=dev, state_dict=sd, det_token_idx=d.det_token_idx, num_frames=8, pe_dtype=torch.float32)
g = np.load("tests/golden/tiny_evaluate_B2_T8_seed3.npz")
batch = synthetic_batch(d, B=2, T=8, L=24, n_det=1, seed=3)
prompt = batch.input_ids[:, :int(g["prompt_len"])].clone()
feats, outs = model(mode="encode_images", images=batch.global_enc_images.to(bf).to(dev))
print("golden", g["greedy_ids"][:, -12:])
for i in range(3):
    ids_u, hid_u = model.generate_greedy(feats, prompt.to(dev), 12, use_cache=False)
    ids_c, hid_c = model.generate_greedy(feats, prompt.to(dev), 12, use_cache=True)
    ids_n, hid_n = model.generate_greedy(feats, prompt.to(dev), 12, use_cache=True, use_graph=False)
    print(i, "uncached", ids_u[:, -12:].tolist(), "\n  graph   ", ids_c[:, -12:].tolist(), "\n  nograph ", ids_n[:, -12:].tolist(), ids_u.shape, ids_c.shape, hid_u.shape, hid_c.shape)
