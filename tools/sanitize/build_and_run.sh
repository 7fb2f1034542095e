#!/bin/bash
# ASan + UBSan CPU build of the engine's host side (never a GPU build; see host_selftest.cc).
set -e
cd "$(dirname "$0")/../.."
mkdir -p build
HIPCC=${HIPCC:-/opt/rocm/bin/hipcc}
$HIPCC -x hip --offload-host-only -DVF_HOST_SELFTEST -std=c++17 -O1 -g -fno-omit-frame-pointer \
    -fsanitize=address,undefined -fno-sanitize-recover=undefined \
    visual_foresight_amd/csrc/vf_engine.hip tools/sanitize/host_selftest.cc -o build/vf_host_selftest
ASAN_OPTIONS=detect_leaks=1:abort_on_error=0 UBSAN_OPTIONS=print_stacktrace=1 ./build/vf_host_selftest
