import importlib.util, json, os, sys, torch
ROOT=os.environ.get("GRAFT_REPO_ROOT","/root/repo"); sys.path.insert(0,ROOT)
spec=importlib.util.spec_from_file_location("fd",os.path.join(ROOT,"tests","test_full_depth_gpu.py")); FD=importlib.util.module_from_spec(spec); spec.loader.exec_module(FD)
dev=torch.device("cuda:0")
out={}
for F,sdt in ((1000.0,torch.float32),(100.0,None),(100.0,torch.float32),(0.0,torch.float32)):
    r=FD.run_training_parity(dev,"deep_narrow",outliers=F,stream_dtype=sdt)
    out[f"F{int(F)}_{'f32' if sdt else 'bf16'}"]={"groups":{g:(round(v['cos'],4),round(v['norm_ratio'],4)) for g,v in r["gradient_groups"].items()},"whole":r["whole_gradient"],"loss":r["loss_terms_rel_err"],"box":r["box_l1_train_mode_vs_oracle"]}
    json.dump(out,open(os.path.join(ROOT,"gpurun_out","j4_outlier_probe.json"),"w"),indent=1)
print(json.dumps(out))
