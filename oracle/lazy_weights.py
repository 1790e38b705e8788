"""Weights for the CPU oracle at FULL dimensions without holding the 30 GB fp32 state dict on the host.

TEST INFRASTRUCTURE ONLY (tests/, bench.py's cpu_baseline leg). The oracle only indexes / .get()s its state dict, so a mapping
that regenerates one tensor per access from the name-keyed generator of grove_amd/synthetic.py (bit-identical on CPU and GPU) is
enough; `gen_device` may be the GPU, which makes the 7.6e9 values in seconds. `fetch_seconds` accumulates the time spent
generating / copying, so that a timed oracle run can report its own compute time.
"""
import time

import torch


class LazyRoundedWeights(dict):
    """{name: fp32 tensor of the bf16-rounded synthetic weight}, generated on access."""

    def __init__(self, d, gen_device="cpu", outliers=0.0, scale=None, keep_prefix=None, keep_fp32=False):
        """scale: {name: factor} applied to the generated tensor before its bf16 rounding (a test that wants, say, louder token embeddings
        applies the same factor to the device model's copy). keep_prefix: names starting with it are kept on the host as bf16 after their
        first use (a decode loop touches every LLaMA weight once per token: 13.5 GB held instead of 7.6e9 values regenerated per step);
        keep_fp32 holds them widened (27 GB for LLaMA-7B: no per-access conversion, a cached decode step then costs one fp32 mat-vec sweep)."""
        super().__init__()
        self.outliers = outliers
        self.scale = scale or {}
        self.keep_prefix, self._kept, self.keep_fp32 = keep_prefix, {}, keep_fp32
        from grove_amd.synthetic import param_shapes
        self.d, self.shapes = d, param_shapes(d)
        self._last = (None, None)
        self.gen_device = gen_device
        self.fetch_seconds = 0.0

    def __contains__(self, k):
        return k in self.shapes

    def __getitem__(self, k):
        from grove_amd.synthetic import synthetic_param
        if self._last[0] == k:
            return self._last[1]
        t0 = time.perf_counter()
        if k in self._kept:
            t = self._kept[k].float()  # (a no-op for fp32)
        else:
            t = synthetic_param(k, self.shapes[k], self.d, self.gen_device, self.outliers)
            if k in self.scale:
                t = t * self.scale[k]
            t16 = t.to(torch.bfloat16).cpu()
            t = t16.float()
            if self.keep_prefix is not None and k.startswith(self.keep_prefix):
                self._kept[k] = t if self.keep_fp32 else t16
        self._last = (k, t)
        self.fetch_seconds += time.perf_counter() - t0
        return t

    def get(self, k, default=None):
        return self[k] if k in self.shapes else default
