#!/usr/bin/env python
"""Per-kernel averages of the PMC counters in a rocprofv3 rocpd database."""
import sqlite3
import sys
from collections import defaultdict


def main(path, filt=""):
    db = sqlite3.connect(path)
    t = {r[0].split("_0000")[0]: r[0] for r in db.execute("select name from sqlite_master where type='table'")}
    q = ("select s.kernel_name, p.symbol, e.value, d.id from %s e join %s p on e.pmc_id = p.id "
         "join %s d on e.event_id = d.event_id join %s s on d.kernel_id = s.id"
         % (t["rocpd_pmc_event"], t["rocpd_info_pmc"], t["rocpd_kernel_dispatch"], t["rocpd_info_kernel_symbol"]))
    acc = defaultdict(lambda: defaultdict(list))
    for name, sym, val, did in db.execute(q):
        if filt in name:
            acc[name.split("(")[0]][sym].append(val)
    for name, d in acc.items():
        print(name)
        for sym, vals in sorted(d.items()):
            print("   %-32s n=%-3d avg=%.4g" % (sym, len(vals), sum(vals) / len(vals)))


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2] if len(sys.argv) > 2 else "")
