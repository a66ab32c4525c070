#!/usr/bin/env python3
"""Per-stream timeline of one steady-state frame from a rocprofv3 --kernel-trace database of a multi-slice run: for every
queue, each kernel's start (relative to the frame's first kernel on that queue), duration and the gap to the previous
kernel of the same queue.  Usage: timeline.py results.db [frames_from_end]"""
import re, sqlite3, sys
from collections import defaultdict
db = sqlite3.connect(sys.argv[1])
back = int(sys.argv[2]) if len(sys.argv) > 2 else 3
cols = [r[1] for r in db.execute("pragma table_info(kernels)")]
print("columns:", cols)
qcol = "queue_id" if "queue_id" in cols else ("stream_id" if "stream_id" in cols else None)
rows = db.execute("select name, start, end, %s from kernels order by start" % (qcol or "0")).fetchall()
rows = [(re.sub(r"^void ", "", re.sub(r"\(.*$", "", n)).replace("msckf::", ""), s, e, q) for n, s, e, q in rows]
byq = defaultdict(list)
for r in rows:
    byq[r[3]].append(r)
print("queues:", {q: len(v) for q, v in byq.items()})
for q, v in sorted(byq.items(), key=lambda kv: -len(kv[1]))[:5]:
    # frames are delimited by k_propagate
    idx = [i for i, r in enumerate(v) if r[0].startswith("k_propagate")]
    if len(idx) < back + 2:
        continue
    a, b = idx[-back - 1], idx[-back]
    t0 = v[a][1]
    print("queue", q, "frame of", b - a, "kernels, span %.1f us" % ((v[b][1] - t0) / 1e3))
    prev_end = None
    for r in v[a:b]:
        gap = 0.0 if prev_end is None else (r[1] - prev_end) / 1e3
        print("   %-44s start %7.1f  dur %6.1f  gap %6.1f" % (r[0][:44], (r[1] - t0) / 1e3, (r[2] - r[1]) / 1e3, gap))
        prev_end = r[2]
