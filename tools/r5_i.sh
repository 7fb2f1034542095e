cd $GRAFT_REPO_ROOT
O=gpurun_out/r5_i; mkdir -p $O
VF_LIBRARY=build/ab/knobs.so timeout 600 python tools/exp_two_cohorts.py 25 2>&1 | tee $O/two_cohorts_25.txt
VF_LIBRARY=build/ab/knobs.so timeout 600 python tools/exp_two_cohorts.py 50 2>&1 | tee $O/two_cohorts_50.txt
