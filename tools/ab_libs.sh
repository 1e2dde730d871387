#!/bin/bash
# same-box A/B of several builds of one source tree (make VARIANT=_x EXTRA=... in homography.js_amd), one process per library, alternating:
#   tools/ab_libs.sh "cur _x _y" CONFIGS SOURCES [REPS] [sweep args...]      ("cur" = lib/libhgwarp.so, "_x" = lib/libhgwarp_x.so through HGWARP_LIB)
# prints, sorted by config: library, config, sources, options, kernel ms, step ms
export TMPDIR=/tmp
libs=${1:-cur}; cfgs=${2:-C3,C4}; srcs=${3:-shared}; reps=${4:-2}; shift 4
o=$PWD/gpurun_out/ab_libs; mkdir -p $o; log=$o/ab_$(date +%s).log; : > $log
for rep in $(seq 1 $reps); do
for lib in $libs; do
  if [ $lib = cur ]; then unset HGWARP_LIB; else export HGWARP_LIB=$PWD/homography.js_amd/lib/libhgwarp$lib.so; fi
  for c in ${cfgs//,/ }; do
    python tools/sweep.py $c ${@:-phase=-1} --sources $srcs 2>&1 | grep "config\|rror" | sed "s/^/$lib /" | cut -c1-220 >> $log
  done
done; done
unset HGWARP_LIB
python - $log <<'PY'
import json, sys, collections
rows = collections.OrderedDict()
for line in open(sys.argv[1]):
    lib, _, js = line.partition(" ")
    try: d = json.loads(js)
    except Exception: print(line.rstrip()); continue
    key = (d["config"], d["sources"], " ".join(f"{k}={v}" for k, v in d.items() if k not in ("config", "F", "sources", "kernel", "kernel_ms", "step_ms", "same_bytes", "redone")), str(d["kernel"])[:40])
    rows.setdefault(key, collections.OrderedDict()).setdefault(lib, []).append((d["kernel_ms"], d["step_ms"], d["same_bytes"], d["redone"]))
for key, libs in rows.items():
    print(*key)
    for lib, v in libs.items():
        print(f"    {lib:8s} kernel " + " ".join(f"{a:.4f}" for a, _, _, _ in v) + "   step " + " ".join(f"{b:.4f}" for _, b, _, _ in v) + ("" if all(x[2] for x in v) else "  SAME_BYTES FALSE") + ("" if not any(x[3] for x in v) else f"  redone {[x[3] for x in v]}"))
PY
