"""Intervals of the step_kernel launches in a rocprofv3 kernel trace (csv) of tools/chains_trace.py or bench.py: per launch
size (grid), the average duration of one launch, the time from the first start to the last end of every burst of launches
(a graph replay), the step period that follows from it, and how much of a burst two launches were in flight at once.

    python tools/chains_timeline.py <rocprofv3 output dir> [--dump N]
"""
import csv
import re
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
        # boards of the launch = threads x boards per lane (the kernel's third template argument: step_kernel<F64, RESET, B, BLOCK, ..>)
        m = re.search(r"step_kernel<[^,]+,[^,]+,\s*(\d+)\s*,", row["Kernel_Name"])
        boards = grid * (int(m.group(1)) if m else 1)
        if boards < (1 << 18):          # only the headline-sized launches (sub-batch chains and whole batches)
            continue
        rows.append((int(row["Start_Timestamp"]), int(row["End_Timestamp"]), boards, row.get("Queue_Id", "?"), row["Kernel_Name"]))
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
    if len(b) < 4:
        continue
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
    steps = sum(r[2] for r in b) / float(1 << 20)          # r[2] = boards of the launch
    by[(tuple(grids), len(queues))].append((span / 1e3 / steps, sum(r[1] - r[0] for r in b) / len(b) / 1e3, 100.0 * busy1 / span,
                                             100.0 * busy2 / span, len(b)))
print("%d step_kernel launches of >= 262,144 boards in %d bursts (%s)" % (len(rows), len(bursts), d))
print("per launch form: bursts, launches per burst (min..max), then MEDIANS over the bursts of: first start -> last end per Mi boards, the "
      "duration of one launch, share of the burst with >= 1 / >= 2 launches in flight")
for (grids, nq), v in sorted(by.items()):
    per_mi = sorted(x[0] for x in v)[len(v) // 2]
    print("  launches of %s boards on %d queue(s): %3d bursts of %d..%d launches: %6.2f us per Mi boards (%.3f of 8 TB/s at 46 B per board); "
          "a launch lasts %5.2f us; >= 1 in flight %5.1f %%, >= 2 in flight %5.1f %%"
          % (list(grids), nq, len(v), min(x[4] for x in v), max(x[4] for x in v), per_mi, 46 * (1 << 20) / (per_mi * 1e3) / 8000.0,
             sorted(x[1] for x in v)[len(v) // 2], sorted(x[2] for x in v)[len(v) // 2], sorted(x[3] for x in v)[len(v) // 2]))
if dump:
    def overlap(b):
        ev = sorted([(r[0], 1) for r in b] + [(r[1], -1) for r in b])
        depth, t_prev, busy2 = 0, ev[0][0], 0
        for t, dlt in ev:
            if depth >= 2:
                busy2 += t - t_prev
            depth += dlt
            t_prev = t
        return busy2 / float(max(r[1] for r in b) - min(r[0] for r in b))
    two = [b for b in bursts if len(b) >= 4 and len(set(r[3] for r in b)) > 1]
    if two:
        b = max(two, key=overlap)
        t0 = b[0][0]
        print("first %d launches of the two-queue burst with the largest overlap (start / end in us from the burst's first start, boards, queue):" % dump)
        for r in b[:dump]:
            print("   %8.2f %8.2f   %8d  queue %s" % ((r[0] - t0) / 1e3, (r[1] - t0) / 1e3, r[2], r[3]))
