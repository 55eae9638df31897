#!/bin/bash
# VALU instructions and time per reflecting pair for a few workload shapes (kernel trace + one PMC pass each).
# Usage: tools/valu_split.sh <tag>
set -u
TAG=${1:-vs}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/valu_$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
cd /tmp
i=0
while read -r ARGS; do
  i=$((i+1))
  B="python3 $ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-single-profile --no-legs --no-traffic $ARGS"
  rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/w$i/trace" -- $B > "$OUT/w$i.json" 2> "$OUT/w$i.trace.log"
  rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVES SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SALU GRBM_GUI_ACTIVE --output-format csv -d "$OUT/w$i/pmc" -- $B > /dev/null 2> "$OUT/w$i.pmc.log"
  echo "$ARGS" > "$OUT/w$i.args"
done <<'LIST'
--profiles 10000 --freqs 174 --n-points 1000 --mode X
--profiles 10000 --freqs 174 --n-points 1000 --mode X --option shortx_kernel=0
--profiles 10000 --freqs 174 --n-points 200 --mode X
--profiles 10000 --freqs 174 --n-points 200 --mode X --option shortx_kernel=0
LIST
python3 - "$OUT" <<'PY'
import csv, glob, json, sys, os, collections
out = sys.argv[1]
for d in sorted(glob.glob(os.path.join(out, "w*.args"))):
    w = d[:-5]
    args = open(d).read().strip()
    line = [l for l in open(w + ".json") if l.startswith("{")]
    b = json.loads(line[-1]) if line else {}
    acc = collections.defaultdict(list)
    for p in glob.glob(w + "/pmc/*/*_counter_collection.csv"):
        for r in csv.DictReader(open(p)):
            if "vfo_" in r["Kernel_Name"] and "finalize" not in r["Kernel_Name"] and float(r["Counter_Value"]) > 0:
                acc[(r["Kernel_Name"][:30], r["Counter_Name"])].append(float(r["Counter_Value"]))
    ks = collections.defaultdict(dict)
    for (k, c), v in acc.items():
        ks[k][c] = sum(v) / len(v)
    c = b.get("config", {}); pairs = (c.get("profiles_per_gpu") or 0) * (c.get("n_freq") or 0)
    refl = b.get("reflecting_fraction") or b.get("finite_fraction") or 0
    print(json.dumps({"args": args, "kernel_ms": b.get("kernel_ms"), "pairs": pairs, "reflecting": refl,
                      "kernels": {k: {"valu": v.get("SQ_INSTS_VALU"), "valu_per_reflecting_pair": (v.get("SQ_INSTS_VALU", 0) / (pairs * refl)) if pairs and refl else None,
                                      "valu_busy": (v.get("SQ_ACTIVE_INST_VALU", 0) * 4 / 1024) / (v.get("GRBM_GUI_ACTIVE", 1) / 8),
                                      "salu": v.get("SQ_INSTS_SALU"), "lds_insts": v.get("SQ_INSTS_LDS"), "lds_active": v.get("SQ_LDS_IDX_ACTIVE"),
                                      "lds_conflict": v.get("SQ_LDS_BANK_CONFLICT")} for k, v in ks.items()}}))
PY
