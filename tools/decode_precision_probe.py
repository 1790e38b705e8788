"""How accurate are the hidden states of the CACHED DECODE steps next to the prefill rows? (the probe behind
tests/test_full_depth_gpu.py::test_generated_rows_hidden_precision_full_depth; prints its JSON)"""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tests.test_full_depth_gpu import decode_precision_probe  # noqa: E402

if __name__ == "__main__":
    print(json.dumps(decode_precision_probe(torch.device("cuda:0"))))
