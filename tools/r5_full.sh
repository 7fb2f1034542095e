cd $GRAFT_REPO_ROOT
O=gpurun_out/r5_full; mkdir -p $O
timeout 1700 python -m pytest tests -q -m gpu > $O/gputests.log 2>&1; echo "tests rc $?"; tail -3 $O/gputests.log
timeout 300 python tools/fingerprint.py r5 > $O/fingerprint.txt 2>&1
VF_LIBRARY=build/ab/gatediv.so timeout 300 python tools/fingerprint.py gatediv > $O/fingerprint_gatediv.txt 2>&1
bash tools/run_profiles.sh r05
