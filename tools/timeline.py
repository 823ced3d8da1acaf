#!/usr/bin/env python
"""Timeline of ONE step out of a rocprofv3 kernel-trace database: every dispatch of the step in launch order with its start
offset, duration, gap to the previous kernel's end and grid size -- where a short step's time goes (launch gaps vs kernels).

    python tools/timeline.py <results.db> <first kernel of a step, substring> [<step index>]
"""
import sqlite3
import sys


def main(path, first, which=-3):
    db = sqlite3.connect(path)
    tables = [r[0] for r in db.execute("select name from sqlite_master where type='table'")]
    disp = [t for t in tables if t.startswith("rocpd_kernel_dispatch")][0]
    sym = [t for t in tables if t.startswith("rocpd_info_kernel_symbol")][0]
    scols = [r[1] for r in db.execute("pragma table_info(%s)" % sym)]
    name_col = "kernel_name" if "kernel_name" in scols else ("display_name" if "display_name" in scols else "name")
    dcols = [r[1] for r in db.execute("pragma table_info(%s)" % disp)]
    gx = "d.grid_size_x" if "grid_size_x" in dcols else "0"
    wx = "d.workgroup_size_x" if "workgroup_size_x" in dcols else "0"
    rows = list(db.execute("select s.%s, d.start, d.end, %s, %s from %s d join %s s on d.kernel_id = s.id order by d.start"
                           % (name_col, gx, wx, disp, sym)))
    starts = [i for i, r in enumerate(rows) if first in r[0]]
    # a step begins at every occurrence of `first` that follows a different kernel
    begins = [i for i in starts if i == 0 or first not in rows[i - 1][0]]
    b = begins[which]
    e = begins[which + 1] if which + 1 < 0 or which + 1 < len(begins) else len(rows)
    if which + 1 == 0:
        e = len(rows)
    t0, prev_end = rows[b][1], rows[b][1]
    print("| # | kernel | start us | dur us | gap us | grid | wg |")
    print("|---|---|---|---|---|---|---|")
    for k, (name, st, en, g, w) in enumerate(rows[b:e]):
        print("| %d | %s | %.2f | %.2f | %.2f | %s | %s |" % (k, name.split("(")[0][:60], (st - t0) / 1e3, (en - st) / 1e3,
                                                           (st - prev_end) / 1e3, g, w))
        prev_end = en
    print("step span: %.2f us (first start to last end), kernels busy %.2f us" % (
        (rows[e - 1][2] - t0) / 1e3, sum(r[2] - r[1] for r in rows[b:e]) / 1e3))


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2], int(sys.argv[3]) if len(sys.argv) > 3 else -3)
