#!/bin/bash
# round 6, session ac: the round-end evidence set on the final tree of the third session (tools/final_profiles.sh r06_final) + the driver's
# command + the GPU suite
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out
timeout 2400 python -m pytest tests -m gpu -x -q 2>&1 | grep -E "passed|failed|rror" | tail -4 > gpurun_out/r06_gputests_final.txt
timeout 900 python bench.py --steps 20 --warmup 5 > gpurun_out/r06_bench_final.json 2> gpurun_out/r06_bench_final.err
bash tools/final_profiles.sh r06_final > gpurun_out/r06_final_log.txt 2>&1
cat gpurun_out/r06_gputests_final.txt
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r06_bench_final.json").read().strip().splitlines()[-1])
print("B", d["value"], d["roofline"]["frac"], d["roofline"].get("traffic"), d["roofline"].get("frac_of_ceiling"))
print(json.dumps(d["extras"]["lwe"]["cpp_header"], indent=0)[:700])
print({k: (v.get("frac"), v.get("parity_sample_ok")) for k, v in d["extras"]["configs"].items()})
print({k: d["extras"]["lwe"][k].get("encryptions_per_s") for k in ("unfused", "fused")})
PY
