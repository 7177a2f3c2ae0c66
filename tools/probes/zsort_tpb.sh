#!/bin/bash
# zsort_kernel with 128 / 256 / 512 threads per replica: helper time per sweep from the default bench
for t in 256 512 1024; do
  SMCX_ZSORT_TPB=$t python3 bench.py --no-cpu 2>/dev/null | python3 -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        j = json.loads(l); d = j['device_ms']
        print('tpb $t: %.3f ms/step  helpers %.3f ms per sweep' % (j['ms_per_step'], d['helpers'] / j['steps']))
"
done
