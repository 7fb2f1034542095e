L=build/ab/knobs.so
run() { local M=$1; shift; local extra=$1; shift; echo "# samples $M $extra"; local v=("auto=$L"); for pl in "$@"; do v+=("p$pl=$L:VF_LSTM_MREP=$pl"); done; bash tools/ab_bench.sh "${v[@]}" -- --samples $M $extra; }
run 25 "" hhhhqhh hhhhhhh qqhhqhh hhqqqqh
run 50 "" hhhhhhh 11hhhh1 hhhhqhh 1hhhhh1
run 100 "" 1111h11 11hhhh1 hhhhhhh 1hhhhh1
run 200 "" 1111h11 11hhhh1 1111111
