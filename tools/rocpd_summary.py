#!/usr/bin/env python3
"""Dump the per-kernel stats of a rocprofv3 (rocpd sqlite) result into a text table for profiles/."""
import sqlite3
import sys


def main(db, out=None):
    c = sqlite3.connect(db)
    rows = list(c.execute("select name, total_calls, total_duration, average, percentage from top_kernels"))
    regs = {}
    for name, v, s, l in c.execute("select name, max(vgpr_count), max(sgpr_count), max(lds_size) from kernels group by name"):
        regs[name] = (v, s, l)
    lines = ["%-90s %8s %14s %14s %8s %6s %6s %8s" % ("kernel", "calls", "total_us", "avg_us", "pct", "vgpr", "sgpr", "lds_B")]
    for name, calls, tot, avg, pct in rows:
        v, s, l = regs.get(name, ("", "", ""))
        lines.append("%-90s %8d %14.1f %14.2f %8.3f %6s %6s %8s" % (name[:90], calls, tot / 1e3, avg / 1e3, pct, v, s, l))
    txt = "\n".join(lines) + "\n"
    if out:
        open(out, "w").write(txt)
    sys.stdout.write(txt)


if __name__ == "__main__":
    main(*sys.argv[1:3])
