cd $GRAFT_REPO_ROOT
timeout 900 python tools/debug_savp2.py 2>&1 | grep -v amdgpu | tee gpurun_out/debug_savp2.txt
