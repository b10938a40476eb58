#!/bin/bash
# round 5, session b: per-launch durations of the n = 65536 pipeline (batch 64 / 128) and of the 32768-word plan
cd "$(dirname "$0")/../.."
export TMPDIR=/tmp
here=$(pwd)
mkdir -p gpurun_out
{
for b in 64 128; do
  rm -rf /tmp/prof_e
  (cd /tmp && NFLHIP_XCD=0 PYTHONPATH=$here rocprofv3 --kernel-trace --output-format csv -d /tmp/prof_e -- python $here/tools/probes/e_alias_probe.py --child $b 4 2>&1 | grep '^{')
  echo "== E batch $b"; python tools/probes/launch_durations.py /tmp/prof_e nflhip_polymul 6
done
} > gpurun_out/r05_E_launches.txt 2>&1
tail -80 gpurun_out/r05_E_launches.txt
