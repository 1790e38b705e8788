"""Profiling target: the fused CLIP-ViT + projector + LLaMA forward of the bench batch (run under rocprofv3 --kernel-trace --stats)."""
import argparse
import sys
import torch
sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
import bench
from grove_amd.synthetic import FULL
dev = torch.device("cuda:0")
args = argparse.Namespace(frames=16, batch=2, text_len=128)
model, engine = bench.build(FULL, dev, args)
batch = bench.make_batch(FULL, dev, args, 0)
print(bench.vit_llama_forward(model, batch, FULL, reps=5))
