#!/usr/bin/env python
"""Timeline of ONE step out of a rocprofv3 kernel-trace database of `bench.py --graph off` (RGL_BENCH_NO_F32_LINE=1): every dispatch
of the step in launch order with its start offset, duration, gap to the previous kernel's end and grid size -- where a short
step's time goes.  The step is found as the period of the kernel-name sequence in the middle of the trace; it starts at the
state-predictor kernel with the smallest grid (level 0).

    python tools/timeline.py <results.db>
"""
import sqlite3
import sys


def main(path):
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
    names = [(r[0].split("(")[0], r[3]) for r in rows]
    lo, hi = len(rows) // 4, (len(rows) * 6) // 10
    period = None
    for p in range(2, 64):
        if all(names[i] == names[i + p] for i in range(lo, hi - p)):
            period = p
            break
    if period is None:
        print("no periodic step found in the trace")
        return
    cands = [i for i in range(lo, lo + period) if "row_mlp2" in names[i][0] or "scene_graph" in names[i][0]]
    b = min(cands, key=lambda i: (names[i][1], i)) if cands else lo
    # the level-0 embedding kernel (if any) precedes the level-0 scene kernel
    if b > 0 and "scene_graph" in names[b][0] and "row_mlp2" in names[b - 1][0]:
        b -= 1
    b += period * 3
    e = b + period
    t0, prev_end = rows[b][1], rows[b][1]
    print("| # | kernel | start us | dur us | gap us | grid | wg |")
    print("|---|---|---|---|---|---|---|")
    for k, (name, st, en, g, w) in enumerate(rows[b:e]):
        short = name.split("(")[0]
        short = short[short.find("N_1") + 5:] if "N_1" in short else short
        print("| %d | %s | %.2f | %.2f | %.2f | %s | %s |" % (k, short[:58], (st - t0) / 1e3, (en - st) / 1e3,
                                                           (st - prev_end) / 1e3, g, w))
        prev_end = en
    print()
    print("%d launches per step; span %.2f us (first start to last end), kernels busy %.2f us" % (
        period, (rows[e - 1][2] - t0) / 1e3, sum(r[2] - r[1] for r in rows[b:e]) / 1e3))


if __name__ == "__main__":
    main(sys.argv[1])
