#!/bin/bash
# tools/bench_stage_ops.py under several builds of the library (PRHF_LIB), interleaved twice: op, kernel ms, GB/s, share of 8 TB/s
LIBS=${@:-pyrayhf_amd/libprhf.so build/ab/libprhf_r05f.so}
for rnd in 1 2; do for l in $LIBS; do
  echo "== $l"
  PRHF_LIB=$PWD/$l python tools/bench_stage_ops.py 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        r = json.loads(l); print('%-62s %.4f ms %5.0f GB/s %.3f' % (r['op'][:62], r['kernel_ms'], r['roofline']['achieved'], r['roofline']['frac']))"
done; done
