#!/bin/bash
# round 4, session l: matrix-core CRT lift -- the modulus count where it overtakes the VALU kernels; SQ counters of the kernel at E
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r4l
mkdir -p $O
cp nfllib_amd/libnflhip.so /tmp/libnflhip_default.so
for v in default nomfma; do
  if [ $v = default ]; then cp /tmp/libnflhip_default.so nfllib_amd/libnflhip.so; else cp build/ab/libnflhip_$v.so nfllib_amd/libnflhip.so; fi
  timeout 300 python tools/probes/crt_lift_sweep.py $v >> $O/sweep.txt 2>&1
done
cp /tmp/libnflhip_default.so nfllib_amd/libnflhip.so
cat $O/sweep.txt
cd /tmp
dirs=""
for c in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES" "SQ_INSTS_VALU SQ_INSTS_MFMA" "SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU" "SQ_WAIT_ANY SQ_WAIT_INST_ANY" "SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "SQ_INST_CYCLES_VMEM SQ_INSTS_SALU" "SQ_WAVES GRBM_GUI_ACTIVE"; do
  d=/tmp/pmcc_$(echo $c | tr ' ' '_'); rm -rf $d
  timeout 200 rocprofv3 --pmc $c --output-format csv -d $d -- python $GRAFT_REPO_ROOT/tools/probes/crt_lift_time.py pmc > /dev/null 2>&1 || echo "pass $c failed"
  dirs="$dirs $d"
done
python $GRAFT_REPO_ROOT/tools/pmc_sq.py "k_crt_lift_mfma" $dirs | tee $O/pmc.txt
