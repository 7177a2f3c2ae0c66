#!/bin/bash
# geom_sweep_lib.sh LIB [geom_sweep.py args] -- geometry sweep through a variant build, loaded via SMCX_LIB
# (the product library montecarlo-surfacer_amd/libsmcx.so is never overwritten)
lib=$1; shift
SMCX_LIB="$lib" python3 "$(dirname "$0")/geom_sweep.py" "$@"
