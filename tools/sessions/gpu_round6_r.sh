#!/bin/bash
# round 6, session r: the whole GPU suite with the queue's thread as the default, then the driver's bench line
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out
timeout 2400 python -m pytest tests -m gpu -x -q 2>&1 | tail -15 > gpurun_out/r06_gputests_r.txt
cat gpurun_out/r06_gputests_r.txt
timeout 900 python bench.py --steps 20 --warmup 5 > gpurun_out/r06_bench_r.json 2> gpurun_out/r06_bench_r.err
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r06_bench_r.json").read().strip().splitlines()[-1])
print("B", d["value"], d["roofline"]["frac"])
print(json.dumps(d["extras"]["lwe"]["cpp_header"], indent=0))
print({k: (v.get("frac"), v.get("parity_sample_ok")) for k, v in d["extras"]["configs"].items()})
PY
