"""Reads a rocprofv3 kernel trace of `bench.py --steps 20 --warmup 5` and prints, for the launches around the timed region, every
skin launch (start, duration, idle time on the skin stream in front of it) and the move stage beside it: where a 20-step region
loses against the steady state (fill, drain, the transient of the first launches after the synchronisation)."""
import csv, sys
rows = []
for r in csv.DictReader(open(sys.argv[1])):
    n = r["Kernel_Name"]
    k = "skin" if ("skin_ticket" in n or "skin_kernel" in n) else "group" if "move_group" in n else "heavy" if "move_kernel<1" in n else "move0" if "move_kernel<0" in n else "pose" if "pose_kernel" in n else None
    if k: rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), k))
rows.sort()
skins = [r for r in rows if r[2] == "skin"]
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 20
# the timed launches: the last `steps` multi-character launches (the lbs_alone launches behind them are skin_kernel too: drop the last 8)
alone = 8
timed = skins[-(steps + alone + 6):-alone]
t0 = timed[6][0]
prev_end = None
for i, (a, b, _) in enumerate(timed):
    mv = [r for r in rows if r[2] == "move0" and r[0] <= a + 200000 and r[0] >= a - 900000]
    gr = [r for r in rows if r[2] == "group" and r[0] >= a - 200000 and r[0] < b]
    ps = [r for r in rows if r[2] == "pose" and r[1] <= a + 1000 and r[1] >= a - 1200000]
    print("%s skin %2d  start %+9.3f ms  length %.3f  idle before %.3f | pose that fed it: %s | grouped launches beside it: %s" % (
        "warm" if i < 6 else "TIME", i - 6, (a - t0) / 1e6, (b - a) / 1e6, 0 if prev_end is None else (a - prev_end) / 1e6,
        ", ".join("%.3f" % ((y - x) / 1e6) for x, y, _ in ps[-1:]), ", ".join("%.3f" % ((y - x) / 1e6) for x, y, _ in gr)))
    prev_end = b
first_move = [r for r in rows if r[2] == "move0" and r[0] < t0 - 300000][-1]  # the move stage that leads to the first timed skin launch (the next step's move0 starts just before that launch)
print("first move0 of the timed region starts %.3f ms before its skin launch; region = %.3f ms from that move0 to the end of the last skin launch = %.4f ms per step" % (
    (t0 - first_move[0]) / 1e6, (timed[-1][1] - first_move[0]) / 1e6, (timed[-1][1] - first_move[0]) / 1e6 / steps))
