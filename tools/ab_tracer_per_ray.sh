#!/bin/bash
# Per-ray tracer workload (200 000 random rays, flat then spherical) under several builds of the library, interleaved.
# Usage: tools/ab_tracer_per_ray.sh lib [lib ...]
for rnd in 1 2 3; do
for lib in "$@"; do
  PRHF_LIB=$PWD/$lib python tools/tracer_workload.py 2>/dev/null | grep "^{" | python -c "
import sys,json
r=[json.loads(l) for l in sys.stdin]
print('%-34s' % '$lib', ' '.join('%s %.4e' % (x['tracer'], x['rays_per_s_kernel']) for x in r))"
done; done
