cd $GRAFT_REPO_ROOT
for dbg in 0 256 512 768; do
 echo "== VF_FUSE_DEBUG=$dbg"
 VF_FUSE_DEBUG=$dbg VF_FUSE_TOP=1 python tools/persist_stats.py 200 2>&1 | grep -E "TOP_FUSED|per slot" | tail -3
done
