"""Config 2's batched form alone (no batch-1 reference pass), for a kernel trace: python3 tools/dev/infer_batched_only.py [passes]"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from dataclasses import replace  # noqa: E402

from grove_amd import GROVEForCausalLM  # noqa: E402
from grove_amd.infer import infer_clips_batched  # noqa: E402
from grove_amd.synthetic import FULL, synthetic_batch, synthetic_state_dict  # noqa: E402

dev = torch.device("cuda:0")
bf = torch.bfloat16
dims = FULL
sd = synthetic_state_dict(dims, device=dev, dtype=bf)
model = GROVEForCausalLM(dims=dims, device=dev, state_dict=sd, det_token_idx=dims.det_token_idx, num_frames=8)
del sd
model.dims = replace(model.dims, eos_token_id=-1)
if os.environ.get("SERIAL_TOWERS") == "1":
    model.tower_overlap = False
clips = []
for c in range(8):
    b = synthetic_batch(dims, B=1, T=48, L=24, n_det=2, seed=11 + c, device=dev, dtype=bf)
    clips.append((b.global_enc_images, b.grounding_enc_images, b.original_size_list[0]))
prompt = synthetic_batch(dims, B=1, T=48, L=24, n_det=2, seed=11, device="cpu").input_ids[0, :20].clone()
for i in range(int(sys.argv[1]) if len(sys.argv) > 1 else 2):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    infer_clips_batched(model, clips, prompt, max_tokens_new=64)
    torch.cuda.synchronize()
    print(f"pass {i}: {time.perf_counter() - t0:.4f} s", flush=True)
