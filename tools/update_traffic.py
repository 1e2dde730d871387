#!/usr/bin/env python3
"""Refresh one round's entries of profiles/hbm_traffic.json from the PMC passes of tools/profile_round.sh:
python tools/update_traffic.py ROUND  (reads gpurun_out/profile_rNN_<config>[_distinct]/summary.txt, copies each summary to
profiles/rNN_rocprofv3_summary_<config>[_distinct].txt).  HBM bytes = 2 x FETCH_SIZE + WRITE_SIZE (KB), see profiles/r02_pmc_calibration.json."""
import json, os, re, shutil, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
rnd = int(sys.argv[1])
tag = f"r{rnd:02d}"
path = os.path.join(ROOT, "profiles", "hbm_traffic.json")
entries = json.load(open(path))
old = {(e["config"], e.get("sources", "shared")): e for e in entries if e.get("round") == rnd}
anyround = {}                                             # algorithmic bytes / frames per launch do not change with the round: the latest earlier entry carries them
for e in sorted(entries, key=lambda e: e.get("round", 0)):
    anyround[(e["config"], e.get("sources", "shared"))] = e
entries = [e for e in entries if e.get("round") != rnd]
for config in ("C3", "C4", "C5", "C2"):
    for sources in ("shared", "distinct"):
        name = f"{tag}_{config}" + ("_distinct" if sources == "distinct" else "")
        src = os.path.join(ROOT, "gpurun_out", f"profile_{name}", "summary.txt")
        if not os.path.exists(src):
            if (config, sources) in old: entries.append(old[(config, sources)])
            continue
        dst = os.path.join(ROOT, "profiles", f"{tag}_rocprofv3_summary_{name[len(tag) + 1:]}.txt")
        shutil.copyfile(src, dst)
        text = open(src).read()
        # dominant kernel = the warp kernel with the most dispatches in the fetch pass
        best = None
        for m in re.finditer(r"^\s+(void hg::)?(k_pw_rows_s80|k_pw_rows8|k_pw_rows|k_pw_patch|k_pw_tile|k_geo_fast)\S*.*?FETCH_SIZE\s+n=\s*(\d+) avg=(\S+)", text, re.M):
            if best is None or int(m.group(3)) > best[1]: best = (m.group(2), int(m.group(3)), float(m.group(4)))
        kern, _, fetch = best
        wr = None
        for m in re.finditer(r"^\s+(void hg::)?(k_pw_rows_s80|k_pw_rows8|k_pw_rows|k_pw_patch|k_pw_tile|k_geo_fast)\S*.*?WRITE_SIZE\s+n=\s*(\d+) avg=(\S+)", text, re.M):
            if m.group(2) == kern and (wr is None or int(m.group(3)) > wr[0]): wr = (int(m.group(3)), float(m.group(4)))
        prev = old.get((config, sources)) or anyround.get((config, sources), {})
        alg = prev.get("algorithmic_bytes_per_launch")
        hbm = int(round((2 * fetch + wr[1]) * 1024))
        e = {"round": rnd, "config": config, "frames_per_launch": prev.get("frames_per_launch", 8 if config == "C5" else 64), "sources": sources,
             "kernel": kern, "source": os.path.relpath(dst, ROOT), "FETCH_SIZE_KB": int(round(fetch)), "WRITE_SIZE_KB": int(round(wr[1])),
             "correction": "bytes = 2 x FETCH_SIZE + WRITE_SIZE (profiles/r02_pmc_calibration.json: every 128-byte line the L2 requests counts 64 B; nt stores count exactly)",
             "hbm_bytes_per_launch": hbm}
        if alg: e["algorithmic_bytes_per_launch"] = alg; e["ratio_to_algorithmic"] = round(hbm / alg, 3)
        entries.append(e)
        print(config, sources, kern, "fetch KB", int(fetch), "write KB", int(wr[1]), "hbm", hbm, "ratio", e.get("ratio_to_algorithmic"))
json.dump(entries, open(path, "w"), indent=1)
