# usage: ab.sh <outdir> <lib> <tag>
O=$1; L=$2; T=$3
mkdir -p $O
export VF_LIBRARY=$L
python bench.py --no-alt --no-cpu-baseline --steps 10 2>/dev/null | tail -1 > $O/bench_${T}_200.json; python tools/bench_line.py $O/bench_${T}_200.json ${T}-200
python bench.py --samples 25 --no-alt --no-cpu-baseline --steps 10 --warmup 2 2>/dev/null | tail -1 > $O/bench_${T}_25.json; python tools/bench_line.py $O/bench_${T}_25.json ${T}-25
python bench.py --workload c4 --samples 125 --no-alt --no-cpu-baseline --steps 6 --warmup 2 2>/dev/null | tail -1 > $O/bench_${T}_125.json; python tools/bench_line.py $O/bench_${T}_125.json ${T}-125
