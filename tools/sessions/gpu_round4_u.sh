#!/bin/bash
# round 4, session u: the whole GPU suite on the library with the matrix-core CRT kernels
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r4u
mkdir -p $O
( time timeout 2400 python -m pytest tests -x -q -m gpu > $O/pytest_all.log 2>&1 ) 2> $O/pytest_all.time; echo "pytest rc $?" >> $O/pytest_all.log
tail -6 $O/pytest_all.log; tail -3 $O/pytest_all.time
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
