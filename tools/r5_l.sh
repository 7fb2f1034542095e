cd $GRAFT_REPO_ROOT
O=gpurun_out/r5_l; mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_savp.py -x -q -m gpu > $O/tests_savp.log 2>&1; tail -30 $O/tests_savp.log
timeout 300 python tools/fingerprint.py arch2tree > $O/fingerprint.txt 2>&1; tail -10 $O/fingerprint.txt
