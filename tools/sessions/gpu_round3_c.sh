#!/bin/bash
# GPU session C of round 3: (1) same-box A/B of the operand change (VGPR vs SGPR operands of the plain VOP2 ops): bench value,
# and GRBM_GUI_ACTIVE cycles per launch; (2) power and clock under the product kernels vs the pure-VALU micro-benchmark
# (rocm-smi sampled while they run); (3) small batches at n = 32768: register-resident rows vs round 2's one-launch plan
set -u
out=gpurun_out
mkdir -p $out
here=$(pwd)
export TMPDIR=/tmp
cp nfllib_amd/libnflhip.so /tmp/lib_new.so
cp build/alt/nfllib_amd/libnflhip.so /tmp/lib_old.so
: > $out/r03c_ab.txt
for rep in 1 2; do
  for v in new old; do
    cp /tmp/lib_$v.so nfllib_amd/libnflhip.so
    for wl in A B; do
      r=$(timeout 300 python bench.py --workload $wl --steps 100 --warmup 10 --no-cpu-baseline --no-traffic --no-extras 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print(d['value'], d['roofline']['kernel_ms'])")
      echo "$v $wl rep$rep value kernel_ms: $r" >> $out/r03c_ab.txt
    done
  done
done
for v in new old; do
  cp /tmp/lib_$v.so nfllib_amd/libnflhip.so
  for wl in A B; do
    rm -rf /tmp/pmc_$v$wl
    (cd /tmp && timeout 300 rocprofv3 --pmc GRBM_GUI_ACTIVE --output-format csv -d /tmp/pmc_$v$wl -- python $here/bench.py --workload $wl --steps 8 --warmup 2 --no-extras --no-cpu-baseline --no-traffic --no-rccl > /dev/null 2>&1)
    f=$(find /tmp/pmc_$v$wl -name "*counter_collection.csv" | head -1)
    [ -n "$f" ] && cp "$f" $out/r03c_pmc_GRBM_${v}_$wl.csv
  done
done
cp /tmp/lib_new.so nfllib_amd/libnflhip.so
cat $out/r03c_ab.txt
# (2) power / clocks
smi() { for i in $(seq 1 $2); do rocm-smi -P -c --json 2>/dev/null | tr -d '\n'; echo; sleep 0.25; done > $out/r03c_smi_$1.jsonl; }
rocm-smi -P -c -M --json > $out/r03c_smi_idle.json 2>&1
(smi B 40 &) ; timeout 120 python bench.py --workload B --steps 3000 --warmup 10 --no-cpu-baseline --no-traffic --no-extras > $out/r03c_bench_B_long.json 2>/dev/null; sleep 1
(smi A 40 &) ; timeout 120 python bench.py --workload A --steps 4000 --warmup 10 --no-cpu-baseline --no-traffic --no-extras > $out/r03c_bench_A_long.json 2>/dev/null; sleep 1
(smi ubench_mad 40 &) ; for i in 1 2 3 4 5 6 7 8; do timeout 60 ./build/ubench_issue op_mad64_4 > /dev/null 2>&1; done; sleep 1
(smi ubench_bfly 40 &) ; for i in 1 2 3 4 5 6 7 8; do timeout 60 ./build/ubench_issue bfly64iv_4 > /dev/null 2>&1; done; sleep 1
cut -c1-200 $out/r03c_bench_B_long.json
# (3) small batches at n = 32768
: > $out/r03c_small_F.txt
for b in 8 32 128 512; do
  for v in 1 0; do
    r=$(NFLHIP_ROW32K=$v timeout 300 python bench.py --workload F --batch $b --steps 200 --warmup 20 --no-cpu-baseline --no-traffic --no-extras 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print(d['value'], d['roofline']['kernel_ms'])")
    echo "F batch $b ROW32K=$v: $r" >> $out/r03c_small_F.txt
  done
done
cat $out/r03c_small_F.txt
