#!/bin/bash
# scratch (spill) instructions per device function of a gfx950 code object: tools/function_spills.sh [file.co]
co=${1:-${TMPDIR:-/tmp}/vf_engine_gfx950.co}
/opt/rocm/lib/llvm/bin/llvm-objdump -d --no-show-raw-insn "$co" | awk '
/^[0-9a-f]+ <.*>:$/ { name=$2; next }
/scratch_(load|store)/ { n[name]++ }
/v_mfma/ { m[name]++ }
END { for (f in m) printf "%6d mfma %5d scratch  %s\n", m[f], n[f]+0, f }' | sort -k4 | while read a b c d e; do echo "$a $b $c $d $(echo $e | tr -d '<>:' | c++filt | cut -c1-120)"; done
