#!/bin/bash
# Builds the ablated libraries of tools/sessions/gpu_round3_g.sh into build/abl_<name>/nfllib_amd/libnflhip.so (run from the
# repository root, on the CPU container; the objects of the current build are reused, only the code object is regenerated).
set -eu
for v in ${ABLATIONS:-tw0 nolds row0 nobfly tw0,nolds,row0}; do
  d=build/abl_$(echo $v | tr , _)
  rm -rf $d
  mkdir -p $d/nfllib_amd $d/tools
  cp -r nfllib_amd/csrc $d/nfllib_amd/
  cp -r include $d/
  cp tools/gen_*.py $d/tools/
  cp -r tools/asmgen $d/tools/
  touch $d/tools/gen_polymul_asm.py
  (cd $d/nfllib_amd/csrc && NFL_GEN_ABLATE=$v make -s -j4 > make.log 2>&1 && echo "$v built" || echo "$v FAILED") &
done
wait
for d in build/abl_*; do rm -rf $d/nfllib_amd/csrc $d/include $d/tools; done   # only the library travels to the GPU box
ls -la build/abl_*/nfllib_amd/libnflhip.so
