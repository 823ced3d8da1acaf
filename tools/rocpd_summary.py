#!/usr/bin/env python
"""Summarise a rocprofv3 rocpd database (`*_results.db`) into a per-kernel table
(calls, total/avg/min/max duration) -- the same content as `--stats`' kernel_stats.csv.

    python tools/rocpd_summary.py gpurun_out/prof/bench_results.db > profiles/<name>.kernel_stats.md
"""
import sqlite3
import sys


def main(path):
    db = sqlite3.connect(path)
    tables = [r[0] for r in db.execute("select name from sqlite_master where type='table'")]
    disp = [t for t in tables if t.startswith("rocpd_kernel_dispatch")][0]
    sym = [t for t in tables if t.startswith("rocpd_info_kernel_symbol")][0]
    cols = [r[1] for r in db.execute("pragma table_info(%s)" % disp)]
    scols = [r[1] for r in db.execute("pragma table_info(%s)" % sym)]
    name_col = "kernel_name" if "kernel_name" in scols else ("display_name" if "display_name" in scols else "name")
    q = ("select s.%s, count(*), sum(d.end - d.start), avg(d.end - d.start), min(d.end - d.start), "
         "max(d.end - d.start) from %s d join %s s on d.kernel_id = s.id group by s.%s order by 3 desc"
         % (name_col, disp, sym, name_col))
    rows = list(db.execute(q))
    total = sum(r[2] for r in rows) or 1
    print("| kernel | calls | total ms | avg us | min us | max us | % |")
    print("|---|---|---|---|---|---|---|")
    for name, n, tot, avg, mn, mx in rows:
        short = name.split("(")[0]
        print("| %s | %d | %.3f | %.2f | %.2f | %.2f | %.1f |" % (short, n, tot / 1e6, avg / 1e3, mn / 1e3, mx / 1e3,
                                                                  100.0 * tot / total))
    extra = [c for c in ("vgpr_count", "arch_vgpr_count", "accum_vgpr_count", "sgpr_count", "lds_block_size",
                         "group_segment_size", "scratch_size") if c in scols]
    if extra:
        print()
        print("| kernel | " + " | ".join(extra) + " |")
        print("|---|" + "---|" * len(extra))
        for r in db.execute("select %s, %s from %s" % (name_col, ", ".join(extra), sym)):
            if any(r[0] == x[0] for x in rows):
                print("| %s | %s |" % (r[0].split("(")[0], " | ".join(str(v) for v in r[1:])))


if __name__ == "__main__":
    main(sys.argv[1])
