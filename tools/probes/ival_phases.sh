#!/bin/bash
# ival_phases.sh -- builds the interval variants of the one-wave kernels (gen_sweep_ma.py SMCX_GEN_IVAL=a:b): libsmcx_iv_<a>_<b>.so.
# They need s100, s101 among the kernels' clobbers: the list in smcx_sweep_ma.hip is extended for these builds only and restored
# afterwards (the file's hash is part of the product kernels' source ids).  Run here (CPU), then tools/probes/ival_phases.py on a GPU box.
set -e
cd "$(dirname "$0")/../../montecarlo-surfacer_amd/csrc"
cp smcx_sweep_ma.hip /tmp/smcx_sweep_ma.hip.orig
trap 'cp /tmp/smcx_sweep_ma.hip.orig smcx_sweep_ma.hip' EXIT
sed -i 's/"s94", "s95"$/"s94", "s95", "s100", "s101"/' smcx_sweep_ma.hip
grep -q '"s95", "s100", "s101"' smcx_sweep_ma.hip
for iv in ${@:-0:31 31:32 32:21 21:22 22:23 23:24 24:8 8:10 0:10}; do
    tag=iv_${iv/:/_}
    make VARIANT=$tag GENENV="SMCX_GEN_IVAL=$iv" -j8 > /tmp/build_$tag.log 2>&1 || { tail -5 /tmp/build_$tag.log; exit 1; }
    rm -rf build_$tag
    echo built $tag
done
