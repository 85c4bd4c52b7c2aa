"""Intervals of the step_kernel launches in a rocprofv3 kernel trace (csv) of tools/chains_trace.py or bench.py: per launch
size (grid), the average duration of one launch, the time from the first start to the last end of every burst of launches
(a graph replay), the step period that follows from it, and how much of a burst two launches were in flight at once.

    python tools/chains_timeline.py <rocprofv3 output dir> [--dump N]
"""
import csv
import glob
import os
import sys
from collections import defaultdict

d = sys.argv[1]
dump = int(sys.argv[sys.argv.index("--dump") + 1]) if "--dump" in sys.argv else 0
rows = []
for f in sorted(glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True)):
    for row in csv.DictReader(open(f)):
        if "step_kernel" not in row.get("Kernel_Name", ""):
            continue
        grid = int(row.get("Grid_Size") or row.get("Grid_Size_X") or 0)
        rows.append((int(row["Start_Timestamp"]), int(row["End_Timestamp"]), grid, row.get("Queue_Id", "?"), row["Kernel_Name"]))
rows.sort()
if not rows:
    sys.exit("no step_kernel launches in " + d)
# bursts: launches separated by less than 30 us from the previous end belong to one replay
bursts, cur, last_end = [], [], None
for r in rows:
    if last_end is not None and r[0] - last_end > 30000:
        bursts.append(cur)
        cur = []
    cur.append(r)
    last_end = max(last_end or 0, r[1])
bursts.append(cur)
by = defaultdict(list)
for b in bursts:
    grids = sorted(set(r[2] for r in b))
    queues = sorted(set(r[3] for r in b))
    span = max(r[1] for r in b) - min(r[0] for r in b)
    # sweep: time with >= 1 and >= 2 launches in flight
    ev = sorted([(r[0], 1) for r in b] + [(r[1], -1) for r in b])
    depth, t_prev, busy1, busy2 = 0, ev[0][0], 0, 0
    for t, dlt in ev:
        if depth >= 1:
            busy1 += t - t_prev
        if depth >= 2:
            busy2 += t - t_prev
        depth += dlt
        t_prev = t
    boards = sum(r[2] for r in b)          # one lane per board: grid size (threads) = boards of the launch
    by[(tuple(grids), len(queues), len(b))].append((span, busy1, busy2, boards, sum(r[1] - r[0] for r in b) / len(b)))
print("%d step_kernel launches in %d bursts (%s)" % (len(rows), len(bursts), d))
for (grids, nq, nl), v in sorted(by.items()):
    if nl < 4:
        continue
    v.sort()
    span, busy1, busy2, boards, avg = v[len(v) // 2]
    steps = boards / (1 << 20)
    print("launch sizes %s on %d queue(s), %d launches per burst, %d bursts: median burst first start -> last end %.1f us = %.2f us per Mi boards "
          "(%.3f of 8 TB/s at 46 B per board); a launch lasts %.2f us on average; >= 1 launch in flight %.1f %% of the burst, >= 2 in flight %.1f %%"
          % (list(grids), nq, nl, len(v), span / 1e3, span / 1e3 / steps, 46 * (1 << 20) / (span / steps) / 8000.0,
             avg / 1e3, 100.0 * busy1 / span, 100.0 * busy2 / span))
if dump:
    for b in bursts:
        if len(b) >= 4 and len(set(r[3] for r in b)) > 1:
            t0 = b[0][0]
            print("first %d launches of a two-queue burst (start / end in us from the burst's first start, grid, queue):" % dump)
            for r in b[:dump]:
                print("   %8.2f %8.2f  grid %8d  queue %s" % ((r[0] - t0) / 1e3, (r[1] - t0) / 1e3, r[2], r[3]))
            break
