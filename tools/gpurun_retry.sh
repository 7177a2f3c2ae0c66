#!/bin/bash
# gpurun_retry.sh TIMEOUT 'command' -- retries ONLY when gpurun answers 3 (no box or slot free, nothing charged)
t=$1; shift
for i in 1 2 3 4 5 6 7 8 9 10 11 12; do
    /usr/local/graft/bin/gpurun --timeout "$t" -- "$@"
    rc=$?
    [ $rc -ne 3 ] && exit $rc
    sleep 120
done
exit 3
