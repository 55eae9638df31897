#!/bin/bash
# Tracer workloads (200 000 random rays per-ray; fans of 204 800) under several builds of the library, interleaved twice.
# Usage: tools/ab_tracer_builds.sh [lib ...]     (default: the in-tree library and build/ab/libprhf_r05f.so)
LIBS=${@:-pyrayhf_amd/libprhf.so build/ab/libprhf_r05f.so}
for rnd in 1 2; do
for lib in $LIBS; do
  echo "== $lib"
  PRHF_LIB=$PWD/$lib python tools/tracer_workload.py 2>/dev/null | grep "^{" | cut -c1-200
  PRHF_LIB=$PWD/$lib python tools/tracer_fan_workload.py 2>/dev/null | grep "^{" | python -c "
import sys,json
for l in sys.stdin:
    r=json.loads(l); print(r['tracer'],'fan %.4e per-ray %.4e same_rays %s worst %.1e'%(r['rays_per_s_fan'],r['rays_per_s_per_ray'],r['same_rays_turn'],r['worst_path_difference']))"
done; done
