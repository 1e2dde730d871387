#!/usr/bin/env python3
"""Summarises rocprofv3 sqlite outputs (kernel trace stats + PMC passes) as plain text.

Refuses (exit status 2, nothing on stdout) when a bench line found in the run's logs says "verified": false: a summary whose own
bench line is unverified is not evidence and must not reach profiles/."""
import builtins
import glob
import io
import json
import os
import sqlite3
import sys

out = sys.argv[1]
_buf, _unverified = io.StringIO(), []


def print(*a, **k):                                         # collected; written at the end only if every bench line is verified
    builtins.print(*a, file=_buf, **k)


for db in sorted(glob.glob(os.path.join(out, "trace", "*.db"))):
    con = sqlite3.connect(db)
    print("== rocprofv3 --kernel-trace --stats  (top kernels; durations in us)")
    print(f"{'calls':>7} {'total_us':>12} {'avg_us':>10} {'pct':>6}  kernel")
    for name, calls, total, avg, pct in con.execute("select name,total_calls,total_duration,average,percentage from top_kernels order by total_duration desc limit 12"):
        print(f"{calls:7d} {total/1e3 if total > 1e6 else total:12.1f} {avg/1e3 if avg > 1e5 else avg:10.2f} {pct:6.2f}  {name[:110]}")
    print("   (units as reported by rocprofv3's top_kernels view)")
    try:
        rows = con.execute("select name, count(*), avg(end-start), min(end-start), max(end-start) from kernels where name like '%hg::%' group by name").fetchall()
        print("== per-dispatch durations of the hg kernels (ns)")
        for r in rows:
            print(f"  {r[0][:70]:70s} n={r[1]:4d} avg={r[2]:12.0f} min={r[3]:12.0f} max={r[4]:12.0f}")
    except Exception as e:  # schema differs between rocprof versions
        print("  (kernels view unavailable:", e, ")")
    # the plain `python bench.py` measures several configs in one process: the same kernel name then covers launches of different
    # workloads -- told apart by their grid (workgroups = grid_x / workgroup_x; C3 x 64 frames: 35 840 row groups, C4: 19 968, ...)
    for col in ("name", "kernel_name"):
        try:
            rows = con.execute(f"select {col}, grid_x / workgroup_x, count(*), avg(end-start), min(end-start), max(end-start) from kernels "
                               f"where {col} like '%hg::%' group by {col}, grid_x / workgroup_x having count(*) >= 5 order by sum(end-start) desc limit 24").fetchall()
        except Exception:
            continue
        print("== the same by launch shape (kernel, workgroups per launch; ns)")
        for r in rows:
            print(f"  {r[0][:70]:70s} wgs={r[1]:7d} n={r[2]:4d} avg={r[3]:12.0f} min={r[4]:12.0f} max={r[5]:12.0f}")
        break
for d in sorted(glob.glob(os.path.join(out, "pmc_*"))):
    if not os.path.isdir(d):
        continue
    for db in glob.glob(os.path.join(d, "*.db")):
        con = sqlite3.connect(db)
        print(f"== PMC pass {os.path.basename(d)}  (average per dispatch)")
        q = ("select kernel_name, counter_name, count(*), avg(value) from counters_collection "
             "where kernel_name like '%hg::%' group by kernel_name, counter_name")
        for kn, cn, n, v in con.execute(q):
            print(f"  {kn[:44]:44s} {cn:24s} n={n:3d} avg={v:.6g}")
for log in sorted(glob.glob(os.path.join(out, "*.log"))):
    lines = open(log, errors="replace").read().strip().splitlines()
    js = [ln for ln in lines if ln.startswith("{")]
    if js:
        try:
            if json.loads(js[-1]).get("verified") is False:
                _unverified.append(os.path.basename(log))
        except ValueError:
            pass
        print(f"-- {os.path.basename(log)}: bench line printed by this run: {js[-1]}")
    else:
        print(f"-- {os.path.basename(log)}: {lines[-1][:300] if lines else ''}")

if _unverified:
    builtins.print(f"profile_summary: REFUSED -- bench line(s) with \"verified\": false in {', '.join(_unverified)}", file=sys.stderr)
    sys.exit(2)
sys.stdout.write(_buf.getvalue())
