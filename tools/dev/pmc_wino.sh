#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/pmc_wino; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
timeout 600 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/f -o f -- python3 $R/tools/dev/pmc_wino.py > /dev/null 2> $O/err; echo f_rc=$?
timeout 600 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/w -o w -- python3 $R/tools/dev/pmc_wino.py > /dev/null 2>> $O/err; echo w_rc=$?
python3 $R/tools/dev/pmc_wino_summary.py $O/f $O/w > $O/pmc_winograd_traffic.json; cat $O/pmc_winograd_traffic.json
