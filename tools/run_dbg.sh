cd $GRAFT_REPO_ROOT
for m in bd wps3; do
 echo "== VF_ROLE_DEBUG=$m"
 VF_ROLE_DEBUG=$m VF_PROBE_STATS=1 timeout 200 python tools/role_probe.py 96 2>&1 | grep -v amdgpu.ids | head -8
done
