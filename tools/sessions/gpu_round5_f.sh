#!/bin/bash
# round 5, session f: the narrow draws -- exact tests, rates per limb width, SQ counters of the sampler kernels
cd "$(dirname "$0")/../.."
here=$(pwd)
export PYTHONPATH=$here TMPDIR=/tmp
mkdir -p gpurun_out
python -m pytest tests/test_gpu_samplers.py -m gpu -q -x 2>&1 | cut -c1-400 | tail -40 > gpurun_out/r05_samplers_tests.txt
tail -25 gpurun_out/r05_samplers_tests.txt
{
echo "== rates"; python tools/probes/sampler_rates.py 2>/dev/null | grep '^{'
echo "== SQ counters, mean per launch (rocprofv3 --pmc, separate passes; tools/probes/sampler_rates.py quick)"
dirs=""
for c in "SQ_INSTS_VALU SQ_WAVES" "SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES" "SQ_INST_CYCLES_VMEM SQ_WAVE_CYCLES" "SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "GRBM_GUI_ACTIVE SQ_INSTS_LDS" "SQ_INSTS_SALU SQ_ACTIVE_INST_LDS" "SQ_WAIT_INST_LDS SQ_INSTS_VMEM_WR"; do
  d=/tmp/pmcs_$(echo $c | tr ' ' '_'); rm -rf $d
  (cd /tmp && rocprofv3 --pmc $c --output-format csv -d $d -- python $here/tools/probes/sampler_rates.py quick > /dev/null 2>&1) || echo "pass $c failed"
  dirs="$dirs $d"
done
for k in k_gauss_small8 k_gauss_small16 k_sample_uniform8 k_sample_uniform_narrow k_sample_gauss8 k_sample_gauss16; do echo "-- $k"; python tools/pmc_sq.py "$k" $dirs; done
} > gpurun_out/r05_sampler_rates.txt 2>&1
cat gpurun_out/r05_sampler_rates.txt
