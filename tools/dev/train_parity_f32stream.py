import json, sys, torch
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import test_full_depth_gpu as F
r = F.run_training_parity(torch.device("cuda:0"), "full", stream_dtype=torch.float32)
print("F32STREAM", json.dumps({"box_l1": r["box_l1_train_mode_vs_oracle"], "loss_rel": r["loss_terms_rel_err"], "whole": r["whole_gradient"]}))
