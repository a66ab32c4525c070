#!/usr/bin/env python3
"""Summarise a rocprofv3 rocpd SQLite database (kernel-trace) into a per-kernel table: calls, total / average
/ min / max duration.  Usage: rocpd_summary.py results.db [out.md]
With ROCPD_TAIL=K only the last K launches of every kernel are counted (the steady-state window of bench.py)."""
import re
import sqlite3
import sys


def main():
    db = sqlite3.connect(sys.argv[1])
    cols = [r[1] for r in db.execute("pragma table_info(kernels)")]
    name_col = "name" if "name" in cols else [c for c in cols if "name" in c][0]
    import os
    rows = db.execute("select %s, start, end from kernels order by start" % name_col).fetchall()
    rows = [(re.sub(r"^void ", "", re.sub(r"\(.*$", "", nm)), s, e) for nm, s, e in rows]
    tail = int(os.environ.get("ROCPD_TAIL", "0"))
    if tail:
        seen, keep = {}, []
        for r in reversed(rows):
            if seen.get(r[0], 0) < tail:
                seen[r[0]] = seen.get(r[0], 0) + 1
                keep.append(r)
        rows = keep
    agg = {}
    for name, s, e in rows:
        a = agg.setdefault(name, [0, 0, 1 << 62, 0])
        d = e - s
        a[0] += 1; a[1] += d; a[2] = min(a[2], d); a[3] = max(a[3], d)
    tot = sum(a[1] for a in agg.values())
    lines = ["| kernel | calls | total ms | avg us | min us | max us | % |", "|---|---|---|---|---|---|---|"]
    for name, a in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        lines.append("| %s | %d | %.3f | %.1f | %.1f | %.1f | %.1f |" % (name, a[0], a[1] / 1e6, a[1] / a[0] / 1e3, a[2] / 1e3, a[3] / 1e3, 100.0 * a[1] / tot))
    out = "\n".join(lines)
    print(out)
    if len(sys.argv) > 2:
        open(sys.argv[2], "a").write(out + "\n")


if __name__ == "__main__":
    main()
