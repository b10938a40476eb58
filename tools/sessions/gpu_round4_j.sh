#!/bin/bash
# round 4, session j: the CRT lift on the matrix cores (kernels_crt_mfma.hip): parity, then rate against the VALU kernels' 52 k polys/s at E
export TMPDIR=/tmp
O=gpurun_out/r4j
mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "crt" > $O/pytest.log 2>&1; echo "pytest rc $?" >> $O/pytest.log
tail -15 $O/pytest.log
timeout 300 python tools/crt_bench.py > $O/crt_bench.txt 2>&1
cat $O/crt_bench.txt
cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$O/prof -- python $GRAFT_REPO_ROOT/tools/crt_bench.py > /dev/null 2>&1
cd $GRAFT_REPO_ROOT
f=$(ls $O/prof/*/*kernel_stats.csv 2>/dev/null | head -1); [ -n "$f" ] && head -8 "$f" | cut -c1-200
