cd $GRAFT_REPO_ROOT
O=gpurun_out/r5_e; mkdir -p $O
bash tools/sweep_plans.sh 25 c2 auto hhqqqqh qqqqqqq hqqqqqh qhqqqqq hhqqqqq qqqqqqh hhhqqqh hhqqhqh 2>&1 | tee $O/sweep_25.txt
bash tools/sweep_plans.sh 50 c2 auto hhqqqhh hhhhhhh hhqqqqh 11qqqh1 2>&1 | tee $O/sweep_50.txt
bash tools/sweep_plans.sh 125 c4 auto 111hhh1 11hhhh1 1111h11 hhhhhhh 2>&1 | tee $O/sweep_125.txt
for y in 0 60 120 250; do VF_YIELD=$y bash tools/sweep_plans.sh 125 c4 auto; done 2>&1 | tee $O/yield_125.txt
for y in 0 120; do VF_YIELD=$y bash tools/sweep_plans.sh 100 c2 auto; VF_YIELD=$y bash tools/sweep_plans.sh 160 c2 auto; done 2>&1 | tee $O/yield_100_160.txt
