#!/bin/bash
export TMPDIR=/tmp
o=$PWD/gpurun_out/node_host_path; rm -rf $o; mkdir -p $o
timeout 600 node tools/bench_node.mjs > $o/node_host_path.json 2> $o/node.err; echo "rc=$?"; tail -3 $o/node.err
timeout 600 node --expose-gc tools/bench_node.mjs > $o/node_host_path_exposegc.json 2> $o/node2.err; echo "rc=$?"; tail -3 $o/node2.err
python - <<'PY'
import json
for f in ("node_host_path.json","node_host_path_exposegc.json"):
    try:
        d=json.loads(open("gpurun_out/node_host_path/"+f).read().strip().splitlines()[-1])
        print(f, {k:(v.get("ms_per_frame") if isinstance(v,dict) else v) for k,v in d.items()})
    except Exception as e: print(f,"ERR",e)
PY
