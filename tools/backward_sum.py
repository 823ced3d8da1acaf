#!/usr/bin/env python
"""Device time of ONE backward call from a rocprofv3 kernel trace of tools/backward_time.py: the sum of the backward pass's kernels
divided by the number of calls (= launches of its reduction kernel).  usage: backward_sum.py <results.db>"""
import sqlite3
import sys

NAMES = ("rgl_scene_backward_kernel", "reduce_slabs_kernel", "mlp_rows_kernel", "mlp2_rows_kernel", "graph_kernel", "reduce_ranges_kernel")


def main(path):
    db = sqlite3.connect(path)
    tables = [r[0] for r in db.execute("select name from sqlite_master where type='table'")]
    disp = [t for t in tables if t.startswith("rocpd_kernel_dispatch")][0]
    sym = [t for t in tables if t.startswith("rocpd_info_kernel_symbol")][0]
    scols = [r[1] for r in db.execute("pragma table_info(%s)" % sym)]
    name_col = "kernel_name" if "kernel_name" in scols else ("display_name" if "display_name" in scols else "name")
    rows = list(db.execute("select s.%s, count(*), sum(d.end - d.start) from %s d join %s s on d.kernel_id = s.id group by s.%s"
                           % (name_col, disp, sym, name_col)))
    tot, calls = 0.0, 0
    for name, n, t in rows:
        if any(k in name for k in NAMES):
            tot += t
        if "reduce_slabs_kernel" in name or "reduce_ranges_kernel" in name:
            calls += n
    print("%.1f us per backward call (%d calls)" % (tot / 1e3 / max(1, calls), calls))


if __name__ == "__main__":
    main(sys.argv[1])
