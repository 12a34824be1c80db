#!/usr/bin/env python3
"""Timeline analysis of a rocprofv3 rocpd db: per queue, busy time and idle gaps inside the steady-state region.
    python tools/rocpd_timeline.py x_results.db [--skip 0.5]"""
import sqlite3, sys, collections
db = sqlite3.connect(sys.argv[1])
cols = [r[1] for r in db.execute("pragma table_info('kernels')")]
qcol = "queue_id" if "queue_id" in cols else ("stream_id" if "stream_id" in cols else None)
rows = db.execute(f"select name, start, end, {qcol or '0'} from kernels order by start").fetchall()
skip = float(sys.argv[sys.argv.index("--skip") + 1]) if "--skip" in sys.argv else 0.5
t0, t1 = rows[0][1], rows[-1][2]
cut = t0 + (t1 - t0) * skip
rows = [r for r in rows if r[1] >= cut]
span = rows[-1][2] - rows[0][1]
print(f"columns: {cols}")
print(f"{len(rows)} kernels over {span/1e6:.2f} ms (after skipping the first {skip:.0%})")
# union of busy intervals (any queue)
busy, cur_s, cur_e = 0, None, None
for _, s, e, _ in rows:
    if cur_e is None or s > cur_e:
        if cur_e is not None:
            busy += cur_e - cur_s
        cur_s, cur_e = s, e
    else:
        cur_e = max(cur_e, e)
busy += cur_e - cur_s
print(f"GPU busy (union of kernels) {busy/1e6:.2f} ms = {busy/span*100:.1f} %; idle {(span-busy)/1e6:.2f} ms")
perq = collections.defaultdict(lambda: [0, 0])
for n, s, e, q in rows:
    perq[q][0] += 1; perq[q][1] += e - s
for q, (n, t) in perq.items():
    print(f"queue {q}: {n} kernels, {t/1e6:.2f} ms")
# gaps (no kernel running) histogram + which kernel follows the biggest ones
gaps = collections.defaultdict(lambda: [0, 0])
cur_e = None
for n, s, e, q in rows:
    if cur_e is not None and s > cur_e:
        short = n.replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0][:50]
        gaps[short][0] += 1; gaps[short][1] += s - cur_e
    cur_e = e if cur_e is None else max(cur_e, e)
print("idle time by the kernel that FOLLOWS the gap:")
for k, (n, t) in sorted(gaps.items(), key=lambda kv: -kv[1][1])[:15]:
    print(f"  {k:52s} {n:6d} gaps  {t/1e6:8.3f} ms  avg {t/n/1e3:6.1f} us")

# exclusive / overlapped time per queue (sweep line over kernel start/end events)
ev = []
for n, s_, e, q in rows:
    ev.append((s_, 1, q)); ev.append((e, -1, q))
ev.sort()
active = collections.Counter()
last = ev[0][0]
excl = collections.Counter(); both = 0
for t, d, q in ev:
    qs = [k for k, v in active.items() if v > 0]
    if len(qs) == 1:
        excl[qs[0]] += t - last
    elif len(qs) > 1:
        both += t - last
    active[q] += d
    last = t
print("time with exactly one queue active:", {k: f"{v/1e6:.2f} ms" for k, v in excl.items()}, f"; several queues active: {both/1e6:.2f} ms")
