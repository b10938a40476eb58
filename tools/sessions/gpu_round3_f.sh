#!/bin/bash
# GPU session F of round 3: the host-pointer pipeline with a second host thread for the results
set -u
out=gpurun_out
mkdir -p $out
export TMPDIR=/tmp
(timeout 900 python -m pytest tests -m gpu -x -q -k "host_pointer or pinned_pipeline or surface" 2>&1 | tail -5) > $out/r03f_pytest.txt
tail -2 $out/r03f_pytest.txt
python - <<'PY' > $out/r03f_host_pipe.txt 2>&1
import ctypes as C, time, numpy as np
from nfllib_amd import Engine, _lib
lib=_lib.lib
lib.nflhip_debug_host_pipe_seconds.argtypes=[C.c_void_p, C.POINTER(C.c_double)]
lib.nflhip_debug_host_pipe_seconds.restype=None
e=Engine(64,4096,4)
for batch in (512, 2048, 8192):
    a=e.to_host(e.fill_uniform(e.empty(batch),1,0)); b=e.to_host(e.fill_uniform(e.empty(batch),1,1))
    out=e.h_polymul(a,b)
    for rep in range(2):
        s0=(C.c_double*4)(); lib.nflhip_debug_host_pipe_seconds(e.ctx,s0)
        t=time.perf_counter(); e.h_polymul(a,b,out=out); dt=time.perf_counter()-t
        s1=(C.c_double*4)(); lib.nflhip_debug_host_pipe_seconds(e.ctx,s1)
        d=[s1[i]-s0[i] for i in range(4)]
        print("batch %d: %.1f ms = %.0f polymul/s = %.1f GB/s moved; copy-in %.1f ms, copy-out %.1f ms (second thread), that thread waiting for the device %.1f ms, in the call %.1f ms" % (batch, dt*1e3, batch/dt, 3*batch*131072/dt/1e9, d[0]*1e3, d[1]*1e3, d[2]*1e3, d[3]*1e3))
PY
cat $out/r03f_host_pipe.txt
timeout 600 python bench.py --no-cpu-baseline --no-traffic > $out/r03f_bench_B.json 2>/dev/null
python -c "
import json; d=json.loads(open('$out/r03f_bench_B.json').readline()); print(d['value'], d['extras']['host_pointer_polymul_per_s'])"
