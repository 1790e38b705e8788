cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r03
GROVE_FULL_TRAIN_PARITY=1 timeout 2400 python -m pytest tests/test_full_depth_gpu.py -x -q -m gpu -k "training_vs_oracle_autograd" -s > gpurun_out/r03/job8_full_train.log 2>&1
echo "rc=$?"; tail -2 gpurun_out/r03/job8_full_train.log
python - <<'PY'
import json
for w in ("deep_narrow", "full"):
    d = json.load(open(f"gpurun_out/full_depth_training_parity_{w}.json"))
    print(w)
    print("  ", d["box_loss_gradient_at_hip_boxes_vs_oracle_boxes"], d["gradient_groups"]["box_decoder"])
PY
