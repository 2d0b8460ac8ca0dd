#!/usr/bin/env python
"""Summarise rocprofv3 (ROCm 7.2, rocpd sqlite output) runs into the small text files committed
under profiles/.  Usage: summarize_rocpd.py kernel <db> | pmc <db> <COUNTER>"""
import sqlite3
import sys


def short(n, w=96):
    n = n.replace("void ", "")
    return n if len(n) <= w else n[: w - 3] + "..."


def kernel(db):
    c = sqlite3.connect(db)
    rows = list(c.execute("select name, count(*), sum(duration), avg(duration), min(duration), max(duration) from kernels group by name order by 3 desc"))
    tot = sum(r[2] for r in rows)
    print(f"# rocprofv3 --kernel-trace --stats  ({db});  total GPU kernel time {tot / 1e6:.1f} ms over {sum(r[1] for r in rows)} dispatches")
    print(f"{'kernel':96s} {'calls':>7s} {'total_ms':>10s} {'avg_us':>9s} {'min_us':>9s} {'max_us':>9s} {'pct':>6s}")
    for n, cnt, s, a, mn, mx in rows[:25]:
        print(f"{short(n):96s} {cnt:7d} {s / 1e6:10.2f} {a / 1e3:9.1f} {mn / 1e3:9.1f} {mx / 1e3:9.1f} {100 * s / tot:6.2f}")


def pmc(db, counter):
    c = sqlite3.connect(db)
    rows = list(c.execute("select kernel_name, count(*), avg(value), sum(value) from counters_collection where counter_name=? group by kernel_name order by 4 desc", (counter,)))
    print(f"# rocprofv3 --pmc {counter}  ({db});  value unit = KiB per dispatch as reported (FETCH_SIZE on gfx950 reads 1/2 of a wide coalesced stream: x2 for bytes)")
    print(f"{'kernel':96s} {'calls':>7s} {'avg_KiB':>12s} {'sum_MiB':>12s}")
    for n, cnt, a, s in rows[:20]:
        print(f"{short(n):96s} {cnt:7d} {a:12.1f} {s / 1024:12.1f}")


if __name__ == "__main__":
    if sys.argv[1] == "kernel":
        kernel(sys.argv[2])
    else:
        pmc(sys.argv[2], sys.argv[3])
