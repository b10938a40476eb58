#!/bin/bash
# GPU session E of round 3: the pruned library -- tests, bench lines of the main workloads, where the pipelined host-pointer
# path spends its time
set -u
out=gpurun_out
mkdir -p $out
export TMPDIR=/tmp
(timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -30) > $out/r03e_pytest.txt
tail -3 $out/r03e_pytest.txt
for wl in B F A H T; do
  timeout 600 python bench.py --workload $wl --no-cpu-baseline > $out/r03e_bench_$wl.json 2> $out/r03e_bench_$wl.err
  python -c "
import json; d=json.loads(open('$out/r03e_bench_$wl.json').readline()); e=d['extras']; print('$wl', d['value'], d['roofline']['frac'], d['roofline'].get('traffic'), {k:e[k] for k in ('ntt_fwd_GBs','ntt_inv_GBs','polymul_b_pretransformed_per_s','crt_lift_GBs','crt_project_GBs','host_pointer_polymul_per_s')})"
done
python - <<'PY' > $out/r03e_host_pipe.txt 2>&1
import ctypes as C, time, numpy as np
from nfllib_amd import Engine, _lib
lib=_lib.lib
lib.nflhip_debug_host_pipe_seconds.argtypes=[C.c_void_p, C.POINTER(C.c_double)]
lib.nflhip_debug_host_pipe_seconds.restype=None
e=Engine(64,4096,4)
for batch in (512, 2048, 8192):
    a=e.to_host(e.fill_uniform(e.empty(batch),1,0)); b=e.to_host(e.fill_uniform(e.empty(batch),1,1))
    out=e.h_polymul(a,b)
    s0=(C.c_double*4)(); lib.nflhip_debug_host_pipe_seconds(e.ctx,s0)
    t=time.perf_counter(); e.h_polymul(a,b,out=out); dt=time.perf_counter()-t
    s1=(C.c_double*4)(); lib.nflhip_debug_host_pipe_seconds(e.ctx,s1)
    d=[s1[i]-s0[i] for i in range(4)]
    print("batch %d: %.1f ms = %.0f polymul/s = %.1f GB/s moved; copy-in %.1f ms, copy-out %.1f ms, waiting for the device %.1f ms, in the call %.1f ms" % (batch, dt*1e3, batch/dt, 3*batch*131072/dt/1e9, d[0]*1e3, d[1]*1e3, d[2]*1e3, d[3]*1e3))
    t=time.perf_counter(); c2=np.empty_like(a); np.copyto(c2,a); dt2=time.perf_counter()-t
    print("   numpy copy of one operand into fresh pages: %.1f ms (%.1f GB/s)" % (dt2*1e3, a.nbytes/dt2/1e9))
    t=time.perf_counter(); np.copyto(c2,a); dt2=time.perf_counter()-t
    print("   numpy copy again (touched pages): %.1f ms (%.1f GB/s)" % (dt2*1e3, a.nbytes/dt2/1e9))
PY
cat $out/r03e_host_pipe.txt
