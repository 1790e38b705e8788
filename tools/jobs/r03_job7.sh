cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r03
free -g | head -2; nproc
avail=$(free -g | awk '/Mem:/ {print $7}')
echo "available GiB: $avail"
if [ "$avail" -ge 160 ]; then
  GROVE_FULL_TRAIN_PARITY=1 timeout 2400 python -m pytest tests/test_full_depth_gpu.py -x -q -m gpu -k "training_vs_oracle_autograd" -s > gpurun_out/r03/job7_full_train.log 2>&1
  echo "full-width training parity rc=$?"; tail -3 gpurun_out/r03/job7_full_train.log
else
  echo "not enough host memory for the full-width autograd oracle: skipped"
fi
