#!/bin/bash
# run tools/geom_sweep.py against an alternative build of the library: geom_sweep_lib.sh <lib.so> <args>
cp montecarlo-surfacer_amd/libsmcx.so /tmp/libsmcx_keep.so
cp "$1" montecarlo-surfacer_amd/libsmcx.so
shift
python tools/geom_sweep.py "$@" 2>&1 | grep -v amdgpu.ids
cp /tmp/libsmcx_keep.so montecarlo-surfacer_amd/libsmcx.so
