#!/usr/bin/env python
"""Summarise a rocprofv3 --kernel-trace --stats results .db (rocpd sqlite) as a small CSV: per-kernel calls, total/avg/min/max ms,
percentage, launch geometry and register/LDS use.  Usage: rocprof_summary.py <results.db> [out.csv]"""
import sqlite3
import sys


def main():
    db = sqlite3.connect(sys.argv[1])
    rows = list(db.execute(
        "select name, count(*), sum(end-start)/1e6, avg(end-start)/1e6, min(end-start)/1e6, max(end-start)/1e6, "
        "max(grid_x), max(workgroup_x), max(lds_size), max(vgpr_count), max(accum_vgpr_count), max(sgpr_count), max(scratch_size) "
        "from kernels group by name order by 3 desc"))
    tot = sum(r[2] for r in rows) or 1.0
    out = open(sys.argv[2], "w") if len(sys.argv) > 2 else sys.stdout
    out.write("kernel,calls,total_ms,avg_ms,min_ms,max_ms,pct,grid_x,workgroup_x,lds_bytes,vgpr,agpr,sgpr,scratch\n")
    for r in rows:
        out.write('"%s",%d,%.3f,%.4f,%.4f,%.4f,%.2f,%d,%d,%d,%d,%d,%d,%d\n' % (r[0], r[1], r[2], r[3], r[4], r[5], 100.0 * r[2] / tot, *r[6:]))


if __name__ == "__main__":
    main()
