cd $GRAFT_REPO_ROOT
O=gpurun_out/r5_f; mkdir -p $O
python tools/host_profile.py --samples 25 --steps 10 > $O/host_profile_25.txt 2>&1; head -70 $O/host_profile_25.txt
VF_LIBRARY=build/ab/trace.so timeout 300 python tools/trace_chain.py 25 > $O/chain_25.txt 2>&1; tail -36 $O/chain_25.txt
