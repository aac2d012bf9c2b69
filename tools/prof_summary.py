#!/usr/bin/env python3
"""Summarise a rocprofv3 rocpd sqlite database (kernel-trace) into a per-kernel stats table.
Usage: python tools/prof_summary.py gpurun_out/prof/x_results.db > profiles/x_kernel_stats.txt"""
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
rows = db.execute(
    "select name, count(*), avg(end-start), min(end-start), max(end-start), sum(end-start), "
    "max(vgpr_count), max(accum_vgpr_count), max(sgpr_count), max(lds_size), max(grid_x), max(workgroup_x) "
    "from kernels group by name order by sum(end-start) desc").fetchall()
tot = sum(r[5] for r in rows)
print("# rocprofv3 --kernel-trace summary of %s" % sys.argv[1])
print("%-72s %7s %10s %10s %10s %10s %6s %5s %5s %6s %8s %5s" % (
    "kernel", "calls", "avg_us", "min_us", "max_us", "total_ms", "pct", "vgpr", "agpr", "lds", "grid", "wg"))
for r in rows[: int(sys.argv[2]) if len(sys.argv) > 2 else 40]:
    print("%-72s %7d %10.2f %10.2f %10.2f %10.3f %6.2f %5d %5d %6d %8d %5d" % (
        r[0][:72], r[1], r[2] / 1e3, r[3] / 1e3, r[4] / 1e3, r[5] / 1e6, 100.0 * r[5] / tot,
        r[6] or 0, r[7] or 0, r[9] or 0, r[10] or 0, r[11] or 0))
