#!/bin/bash
# Build the library variants the diagnostic tools use, in parallel: the default library (in place), the event-log
# build (build/ab/trace.so, tools/trace_cu.py) and the variant with the IEEE-division gate math of rounds 1-2
# (build/ab/gatediv.so: the fingerprints of tools/fingerprint.py then equal the round-2 ones bit for bit).
cd "$(dirname "$0")/.."
mkdir -p build/ab
F="--offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -shared -fPIC"
S=visual_foresight_amd/csrc/vf_engine.hip
(hipcc $F -DVF_GATE_DIV -o build/ab/gatediv.so $S 2>&1 | grep -E 'error' -A5) &
(hipcc $F -DVF_TRACE -o build/ab/trace.so $S 2>&1 | grep -E 'error' -A5) &
(hipcc $F -o visual_foresight_amd/libvf_hip.so.tmp $S 2>&1 | grep -E 'error' -A5; mv visual_foresight_amd/libvf_hip.so.tmp visual_foresight_amd/libvf_hip.so) &
wait
ls -la build/ab/gatediv.so build/ab/trace.so visual_foresight_amd/libvf_hip.so
