#!/usr/bin/env python3
"""Per-kernel summary of a rocprofv3 rocpd database (the default output of rocprofv3 7.x when no --output-format is given).
    python tools/rocpd_kernels.py <results.db> [calls_per_unit]   -> name, launches / unit, average us, ms / unit"""
import re
import sqlite3
import sys


def main():
    db = sqlite3.connect(sys.argv[1])
    per = float(sys.argv[2]) if len(sys.argv) > 2 else 1.0
    rows = db.execute("select name, count(*), sum(end - start) / 1e3 from kernels group by name order by 3 desc").fetchall()
    tot, n = sum(r[2] for r in rows), sum(r[1] for r in rows)
    print(f"total kernel time {tot / per / 1e3:.3f} ms / unit, {n / per:.1f} launches / unit")
    for name, cnt, us in rows[:30]:
        short = re.sub(r"\(anonymous namespace\)::", "", name).replace("void ", "").split("(")[0][:64]
        print(f"  {short:64s} n={cnt / per:7.1f} avg_us={us / cnt:8.2f} ms={us / per / 1e3:7.3f}")


if __name__ == "__main__":
    main()
