#!/bin/bash
# isa_summary.sh FILE.hip KERNEL_SUBSTRING [extra hipcc flags] -- registers, spills and the instruction mix of
# one kernel's innermost move loop (the deepest "Depth=3" loop), from hipcc -S
src=$1; pat=$2; shift 2
out=/tmp/isa_$$.s
/opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -fno-slp-vectorize "$@" -S --cuda-device-only "$src" -o $out 2>/dev/null || exit 1
name=$(grep -E "^_Z.*$pat.*:" $out | head -1 | cut -d: -f1)
echo "kernel: $name"
awk -v n="$name:" '$1==n{f=1} f{print} f&&/s_endpgm/{exit}' $out > /tmp/isa_kernel.s
grep -A40 "\.name: *$name" $out | grep -E "vgpr_count|sgpr_count|spill|private_segment_fixed|group_segment_fixed" | sed 's/^ */  /'
echo "total lines: $(wc -l < /tmp/isa_kernel.s)"
