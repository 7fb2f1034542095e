#!/bin/bash
# Per-kernel register / LDS / spill figures of the gfx950 code object (device-only compile of the engine,
# then the AMDGPU metadata notes).  Usage: tools/kernel_resources.sh [extra hipcc flags]
set -e
here=$(cd "$(dirname "$0")/.." && pwd)
out=${TMPDIR:-/tmp}/vf_engine_gfx950.co
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off --cuda-device-only --no-gpu-bundle-output -c "$@" \
    -o "$out" "$here/visual_foresight_amd/csrc/vf_engine.hip"
/opt/rocm/lib/llvm/bin/llvm-readelf --notes "$out" | python3 -c "
import re, sys
txt = sys.stdin.read()
for blk in txt.split('- .agpr_count')[1:]:
    get = lambda k: (re.search(r'\.%s:\s+(\S+)' % k, blk) or [None, '?'])[1]
    name = get('name')
    import subprocess
    print('%-90s vgpr %s agpr %s sgpr %s vspill %s sspill %s scratch %s lds %s' % (
        subprocess.run(['c++filt', name], capture_output=True, text=True).stdout.strip()[:90],
        get('vgpr_count'), blk.split()[0], get('sgpr_count'), get('vgpr_spill_count'), get('sgpr_spill_count'),
        get('private_segment_fixed_size'), get('group_segment_fixed_size')))
"
