#!/bin/bash
# round 6, session n: why is a queue run on the queue's own thread slower?  per-run times, worker pinned next to / far from the recorder
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out
make -s -C tests/cpp resident_test 2>&1 | tail -3
{
echo
for cfg in "0 -" "1 -" "1 1" "1 9" "1 70"; do
  set -- $cfg
  echo "## thread $1 worker cpu $2 (recorder pinned to cpu 0)"
  if [ "$2" = "-" ]; then unset NFL_HIP_QUEUE_CPU; else export NFL_HIP_QUEUE_CPU=$2; fi
  NFL_HIP_QUEUE_STATS=1 NFL_HIP_QUEUE_THREAD=$1 NFL_HIP_QUEUE_MIN=100000 NFL_LWE_REPS=16384 taskset -c 0 tests/cpp/resident_test 2>&1 >/dev/null | grep "queue run" | awk '$4 > 1000' | sed -n 12,17p
done
} > gpurun_out/r06_queue_thread_why.txt 2>&1
cat gpurun_out/r06_queue_thread_why.txt
