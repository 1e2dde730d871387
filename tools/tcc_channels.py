#!/usr/bin/env python3
"""Per-channel L2 <-> fabric requests of one kernel from a `rocprofv3 --pmc TCC_EA0_RDREQ TCC_EA0_WRREQ` run (the sqlite output keeps one row
per counter INSTANCE: 16 TCC channels x 8 XCDs on an MI355X):   python tools/tcc_channels.py <output dir> <kernel substring>
Prints one JSON line: mean dispatch duration under the counters, and per counter the total, the coefficient of variation over the 128
instances, over the 16 channels (summed over XCDs) and over the 8 XCDs, and the busiest / idlest channel relative to the mean."""
import glob, json, sqlite3, sys
import numpy as np

out, pat = sys.argv[1], sys.argv[2]
for db in glob.glob(out + "/**/*.db", recursive=True):
    con = sqlite3.connect(db)
    tab = lambda stem: [r[0] for r in con.execute("select name from sqlite_master where type='table' and name like ?", (stem + "%",))][0]
    ev, kd, ks, pm = tab("rocpd_pmc_event"), tab("rocpd_kernel_dispatch"), tab("rocpd_info_kernel_symbol"), tab("rocpd_info_pmc")
    names = dict(con.execute(f"select id, name from {pm}"))
    rows = con.execute(f"select K.event_id, K.end - K.start from {kd} K join {ks} S on S.id = K.kernel_id where S.display_name like ? order by K.id", (f"%{pat}%",)).fetchall()
    if not rows:
        continue
    rows = rows[len(rows) // 2:]                              # the second half of the run (warm)
    res = {"kernel": pat, "dispatches": len(rows), "mean_us_under_pmc": round(sum(r[1] for r in rows) / len(rows) / 1e3, 2)}
    for pid, nm in names.items():
        acc = None
        for e, _ in rows[:40]:
            v = np.array([r[0] for r in con.execute(f"select value from {ev} where event_id = ? and pmc_id = ? order by id", (e, pid))], dtype=np.float64)
            if v.size != 128:
                continue
            acc = v if acc is None else acc + v
        if acc is None:
            continue
        a = acc.reshape(8, 16)
        ch, xc = a.sum(axis=0), a.sum(axis=1)
        cv = lambda x: round(float(x.std() / x.mean()), 4)
        res[nm] = {"per_dispatch": round(float(acc.sum() / min(len(rows), 40))), "cv_instances": cv(acc), "cv_channels": cv(ch), "cv_xcds": cv(xc),
                   "max_channel": round(float(ch.max() / ch.mean()), 4), "min_channel": round(float(ch.min() / ch.mean()), 4)}
    print(json.dumps(res))
