#!/bin/bash
# sweep of the one-launch kernel's ring size 2^RLOG (rows in flight per XCD) and grid; prints polymul/s per setting
# usage: xcd_sweep.sh "E 32 3,768 4,768" "R32 512 3,768"
run() { env NFLHIP_VARIANT=52 "$@" timeout 120 python tools/quick_bench.py --child $B 5 $W 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('%10.0f /s  digest %x' % (d['polymul_per_s'], d['digest']))"; }
for spec in "$@"; do
  set -- $spec; W=$1; B=$2; shift 2
  echo "== $W batch $B"; echo -n "pipeline      : "; run NFLHIP_XCD=0
  for cfg in "$@"; do
    IFS=, read r g d <<< "$cfg"
    echo -n "R=2^$r WGS=$g D=2^$d : "; run NFLHIP_XCD=1 NFLHIP_XCD_RLOG=$r NFLHIP_XCD_WGS=$g NFLHIP_XCD_DLOG=$d
  done
done
