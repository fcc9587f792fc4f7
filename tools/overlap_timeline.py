"""Reads a rocprofv3 kernel trace (csv) of the default bench.py run and reports, for a window of steady-state steps, when each
kernel of the character path ran: per step the start / end of skin_kernel, move kernels and pose_kernel relative to the step's
skin start, and the fraction of the skin kernel's duration during which a collision or pose kernel was running too."""
import csv, sys
rows = []
for r in csv.DictReader(open(sys.argv[1])):
    name = r["Kernel_Name"]
    key = "skin" if ("skin_kernel" in name or "skin_ticket" in name) else "group" if "move_group_kernel" in name else "heavy" if "move_kernel<1" in name else \
        "move0" if "move_kernel<0" in name else "pose" if "pose_kernel" in name else None
    if key:
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), key))
rows.sort()
skins = [r for r in rows if r[2] == "skin"]
print("kernels in trace:", {k: sum(1 for r in rows if r[2] == k) for k in ("skin", "move0", "group", "heavy", "pose")})
mid = skins[len(skins) // 2: len(skins) // 2 + 6]
tot_skin = tot_cov = 0
for s0, s1, _ in mid:
    others = [(a, b, k) for a, b, k in rows if k != "skin" and b > s0 and a < s1]
    cov = 0
    cur = s0
    for a, b, k in sorted(others):
        a, b = max(a, cur), min(b, s1)
        if b > a:
            cov += b - a
            cur = b
    tot_skin += s1 - s0
    tot_cov += cov
    print("skin %.3f ms; concurrent: %s" % ((s1 - s0) / 1e6, ", ".join("%s[%+.3f..%+.3f]" % (k, (a - s0) / 1e6, (b - s0) / 1e6) for a, b, k in sorted(others))))
print("fraction of skin_kernel time with a collision / pose kernel running beside it: %.2f" % (tot_cov / max(tot_skin, 1)))
if len(skins) > 20:
    a, b = skins[10], skins[-10]
    print("mean step period (skin start to skin start): %.4f ms" % ((b[0] - a[0]) / 1e6 / (len(skins) - 20)))
