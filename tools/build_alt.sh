#!/bin/bash
# Build an alternate libnflhip.so under build/NAME (a copy of nfllib_amd/, tools/, include/ with its own generated listings)
# with extra environment for the generator / compiler:  tools/build_alt.sh NAME [VAR=value ...]
# Used for same-box A/Bs on the GPU box: PYTHONPATH=build/NAME python ...  or  tools/ab_probe.py build/NAME/nfllib_amd/libnflhip.so
set -e
cd "$(dirname "$0")/.."
name=$1; shift
dst=build/$name
rm -rf "$dst"; mkdir -p "$dst"
for d in nfllib_amd tools include; do
  mkdir -p "$dst/$d"
  (cd $d && tar cf - --exclude='*.o' --exclude='*.so' --exclude='*.inc' --exclude='*.hsaco' --exclude='*_gfx950.s' --exclude='__pycache__' --exclude='_build' .) | (cd "$dst/$d" && tar xf -)
done
cp nfllib_amd/csrc/ablation_knobs.inc "$dst/nfllib_amd/csrc/" 2>/dev/null || true
env "$@" make -s -j8 -C "$dst/nfllib_amd/csrc" >/dev/null
rm -f "$dst"/nfllib_amd/csrc/*.o "$dst"/nfllib_amd/csrc/*.inc "$dst"/nfllib_amd/csrc/*.hsaco
ls -la "$dst/nfllib_amd/libnflhip.so"
