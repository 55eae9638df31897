#!/bin/bash
# tracer workloads under three builds, interleaved twice
for rnd in 1 2; do
for lib in pyrayhf_amd/libprhf.so build/ab/libprhf_r05f.so build/ab/libprhf_sw5.so; do
  echo "== $lib"
  PRHF_LIB=$PWD/$lib python tools/tracer_workload.py 2>/dev/null | grep "^{" | cut -c1-200
  PRHF_LIB=$PWD/$lib python tools/tracer_fan_workload.py 2>/dev/null | grep "^{" | python -c "
import sys,json
for l in sys.stdin:
    r=json.loads(l); print(r['tracer'],'fan %.4e per-ray %.4e same_rays %s worst %.1e'%(r['rays_per_s_fan'],r['rays_per_s_per_ray'],r['same_rays_turn'],r['worst_path_difference']))"
done; done
