#!/bin/bash
# round 6, session i: F (u64/32768/2) -- the register-resident composed plan against the one-launch plan with its block products on
# incomplete transforms, at F's bench batch; then the round-end evidence set
cd "$(dirname "$0")/../.."
here=$(pwd)
mkdir -p gpurun_out
{
for rep in 1 2; do
  for x in 0 1; do for lv in 0 2; do
    echo -n "F batch 2048 NFLHIP_XCD=$x level $lv: "; NFLHIP_XCD=$x NFL_POLYMUL_LEVEL=$lv PYTHONPATH=$here timeout 200 python tools/probes/hold_polymul.py 32768 2 2048 2 2>/dev/null
  done; done
done
} > gpurun_out/r06_F_plans.txt 2>&1
cat gpurun_out/r06_F_plans.txt
bash tools/final_profiles.sh r06_final > gpurun_out/r06_final_log.txt 2>&1
tail -5 gpurun_out/r06_final_log.txt
python bench.py --steps 20 --warmup 5 > gpurun_out/r06_bench_B_driver_command.json 2> gpurun_out/r06_bench_B_driver_command.err
tail -c 300 gpurun_out/r06_bench_B_driver_command.json
