cd $GRAFT_REPO_ROOT
VF_ROLE_MODE=1 VF_ROLE_DEBUG=bd timeout 300 python bench.py --no-alt --no-cpu-baseline --steps 8 --warmup 2 2>/dev/null | tail -1 | python tools/bench_line.py - dbg-bd
VF_ROLE_MODE=1 VF_ROLE_DEBUG=wps3 timeout 300 python bench.py --no-alt --no-cpu-baseline --steps 8 --warmup 2 2>/dev/null | tail -1 | python tools/bench_line.py - dbg-wps3
VF_ROLE_MODE=1 timeout 300 python tools/persist_stats.py 200 2>&1 | tail -42
