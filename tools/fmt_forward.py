#!/usr/bin/env python3
"""tools/bench_forward.py JSON lines -> the table kept as profiles/rNN_forward_tiles.txt"""
import json, sys
for ln in open(sys.argv[1]):
    d = json.loads(ln)
    red = f"  redone={d['tiles_redone']}" if "tiles_redone" in d else ""
    print(f" {d['source']:>9} F={d['frames']}  {d['case']:<30} out {d['out_Mpx_per_frame']:6.2f} Mpx/frame  scatter+gather {d['scatter_gather_us_per_frame']:7.1f} us/frame"
          f"  tiles {d['tiles_us_per_frame']:7.1f} us/frame  same_bytes={d.get('same_bytes')}{red}")
