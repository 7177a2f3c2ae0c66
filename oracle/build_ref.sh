#!/bin/bash
# oracle/build_ref.sh -- builds oracle/_ref/ from the REAL reference sources where they lie
# under $REF_DIR (default /root/reference).  TEST INFRASTRUCTURE; runs in the build container
# only (the GPU box has no reference tree and uses the prebuilt .so files that travel with the
# snapshot).  Nothing of the reference is written to disk: the line ranges named in
# ref_smc_prelude.c / ref_nw_wrap.c are streamed from the read-only tree straight into gcc's
# stdin, and only the shared objects land in oracle/_ref/ (git-ignored).
#
# The reference fixes the particle number at compile time (SMC.h:29, SMC_noMPI_noWall.c:18),
# so there is one library per N.  Flags: -O2 -ffp-contract=off, as the oracle.
set -euo pipefail
cd "$(dirname "$0")"
REF_DIR="${REF_DIR:-/root/reference}"
CC="${CC:-gcc}"
SMC_NS="${SMC_NS:-32 108 256 500 1024 4000 4096 16384}"
NW_NS="${NW_NS:-32 108 256}"
FLAGS="-O2 -ffp-contract=off -fPIC -w -std=gnu11 -shared -Wl,-Bsymbolic"

if [ ! -f "$REF_DIR/SMC.c" ]; then
    echo "reference tree absent: keeping prebuilt oracle/_ref (if any)"
    exit 0
fi
mkdir -p _ref

# the lines the recipe rewrites must be what it expects (the tree is read-only, but be loud)
[ "$(sed -n '29p' "$REF_DIR/SMC.h")" = "#define N 108" ] || { echo "SMC.h:29 is not '#define N 108'"; exit 1; }
[ "$(sed -n '18p' "$REF_DIR/SMC_noMPI_noWall.c")" = "#define N 32" ] || { echo "SMC_noMPI_noWall.c:18 is not '#define N 32'"; exit 1; }

$CC -O2 -ffp-contract=off -fPIC -w -std=gnu11 -shared \
    -DREF_MATEMATICOSE_C="\"$REF_DIR/matematicose.c\"" \
    -o _ref/libmatematicose_ref.so ref_matematicose_wrap.c -lm

for n in $SMC_NS; do
    out=_ref/libref_smc_N$n.so
    if [ -f "$out" ] && [ "$out" -nt ref_smc_wrap.c ] && [ "$out" -nt ref_smc_prelude.c ] && [ "$out" -nt build_ref.sh ]; then continue; fi
    {
        cat ref_smc_prelude.c
        echo "#line 26 \"$REF_DIR/SMC.h\""
        sed -n '26,121p' "$REF_DIR/SMC.h" | sed "s/^#define N 108\$/#define N $n/"
        echo "#line 269 \"$REF_DIR/SMC.c\""
        sed -n '269,1049p' "$REF_DIR/SMC.c"
        echo "#line 1094 \"$REF_DIR/SMC.c\""
        sed -n '1094,1169p' "$REF_DIR/SMC.c"
        echo "#line 1 \"ref_smc_wrap.c\""
        cat ref_smc_wrap.c
    } | $CC $FLAGS -DREF_MATEMATICOSE_C="\"$REF_DIR/matematicose.c\"" -x c - -o "$out" -lm
done

# timing builds of the same translation unit (bench.py's cpu_baseline leg, kind "reference"): -O3 for a portable
# AVX2/FMA x86-64 level -- the GPU box's host CPU differs from this container's, so no -march=native here
TIME_NS="${TIME_NS:-4096}"
for n in $TIME_NS; do
    out=_ref/libref_smc_N${n}_O3.so
    if [ -f "$out" ] && [ "$out" -nt ref_smc_wrap.c ] && [ "$out" -nt ref_smc_prelude.c ] && [ "$out" -nt build_ref.sh ]; then continue; fi
    {
        cat ref_smc_prelude.c
        echo "#line 26 \"$REF_DIR/SMC.h\""
        sed -n '26,121p' "$REF_DIR/SMC.h" | sed "s/^#define N 108\$/#define N $n/"
        echo "#line 269 \"$REF_DIR/SMC.c\""
        sed -n '269,1049p' "$REF_DIR/SMC.c"
        echo "#line 1094 \"$REF_DIR/SMC.c\""
        sed -n '1094,1169p' "$REF_DIR/SMC.c"
        echo "#line 1 \"ref_smc_wrap.c\""
        cat ref_smc_wrap.c
    } | $CC -O3 -march=x86-64-v3 -fPIC -w -std=gnu11 -shared -Wl,-Bsymbolic -DREF_MATEMATICOSE_C="\"$REF_DIR/matematicose.c\"" -x c - -o "$out" -lm
done

for n in $NW_NS; do
    out=_ref/libref_nw_N$n.so
    if [ -f "$out" ] && [ "$out" -nt ref_nw_wrap.c ] && [ "$out" -nt build_ref.sh ]; then continue; fi
    {
        echo "#line 1 \"$REF_DIR/SMC_noMPI_noWall.c\""
        sed -n '1,10p' "$REF_DIR/SMC_noMPI_noWall.c"
        echo "#line 12 \"$REF_DIR/SMC_noMPI_noWall.c\""
        sed -n '12,72p' "$REF_DIR/SMC_noMPI_noWall.c" | sed "s/^#define N 32\$/#define N $n/"
        echo "#line 258 \"$REF_DIR/SMC_noMPI_noWall.c\""
        sed -n '258,723p' "$REF_DIR/SMC_noMPI_noWall.c"
        echo "#line 787 \"$REF_DIR/SMC_noMPI_noWall.c\""
        sed -n '787,896p' "$REF_DIR/SMC_noMPI_noWall.c"
        echo "#line 1 \"ref_nw_wrap.c\""
        cat ref_nw_wrap.c
    } | $CC $FLAGS -x c - -o "$out" -lm
done
echo "built oracle/_ref: $(ls _ref | tr '\n' ' ')"
