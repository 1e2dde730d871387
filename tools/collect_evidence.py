#!/usr/bin/env python3
"""After `tools/evidence.sh rNN` (+ tools/node_host_path.sh) came back through gpurun: copy what is to be judged from gpurun_out/ into profiles/.
python tools/collect_evidence.py ROUND   -- runs tools/update_traffic.py ROUND first (the per-config summaries + profiles/hbm_traffic.json)."""
import json, os, shutil, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
rnd = int(sys.argv[1]); tag = f"r{rnd:02d}"
G, P = os.path.join(ROOT, "gpurun_out"), os.path.join(ROOT, "profiles")
subprocess.check_call([sys.executable, os.path.join(ROOT, "tools", "update_traffic.py"), str(rnd)])
ev = os.path.join(G, tag + "ev")

def copy(src, dst):
    if os.path.exists(src) and os.path.getsize(src) > 0:
        shutil.copyfile(src, os.path.join(P, dst)); print("copied", dst)
    else: print("MISSING", src)

for c in ("default", "C5", "C4", "C2"):                     # bench lines: the last line of each file is the JSON line
    src = os.path.join(ev, f"bench_{c}.json")
    if os.path.exists(src):
        line = open(src).read().strip().splitlines()[-1]
        json.loads(line)
        open(os.path.join(P, f"{tag}_bench_{c}.json"), "w").write(line + "\n"); print("copied", f"{tag}_bench_{c}.json")
copy(os.path.join(ev, "default_trace", "summary.txt"), f"{tag}_rocprofv3_summary_default.txt")
copy(os.path.join(ev, "forward_tiles.txt"), f"{tag}_forward_tiles.txt")
copy(os.path.join(G, "latency", "device.log"), f"{tag}_latency_F1_8.jsonl")
host = [l for l in open(os.path.join(ev, "latency.log")) if " host {" in l] if os.path.exists(os.path.join(ev, "latency.log")) else []
if host: open(os.path.join(P, f"{tag}_latency_host.jsonl"), "w").writelines(host); print("copied", f"{tag}_latency_host.jsonl")
nh = os.path.join(G, "node_host_path")
try:
    d = {"plain_node": json.loads(open(os.path.join(nh, "node_host_path.json")).read().strip().splitlines()[-1]),
         "node_expose_gc": json.loads(open(os.path.join(nh, "node_host_path_exposegc.json")).read().strip().splitlines()[-1])}
    json.dump(d, open(os.path.join(P, f"{tag}_node_host_path.json"), "w"), indent=1); print("copied", f"{tag}_node_host_path.json")
except Exception as e: print("node host path:", e)
