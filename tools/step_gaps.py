"""Reads a rocprofv3 kernel trace (csv) of the default bench.py run and prints, for a few steady-state steps, EVERY dispatch of the
step (the small list / fill / copy kernels too) with its queue and its start / end relative to the step's skin start, and then the
mean idle gaps along the dependent chain  move0 -> heavy / group -> pose -> next skin  over the middle of the run.

    rocprofv3 --kernel-trace -d <dir> -o run --output-format csv -- python3 bench.py --steps 200 --warmup 20
    python3 tools/step_gaps.py <dir>/.../run_kernel_trace.csv
"""
import csv, sys

def short(name):
    for pat, key in (("skin_ticket", "skin"), ("skin_kernel", "skin"), ("move_group_kernel", "group"), ("move_kernel<1", "heavy"),
                     ("move_kernel<0", "move0"), ("pose_kernel", "pose"), ("classify_kernel", "classify"), ("order_scan", "scan"),
                     ("order_scatter", "scatter"), ("fillBuffer", "fill"), ("copyBuffer", "copy")):
        if pat in name:
            return key
    return name[:24]

rows = []
for r in csv.DictReader(open(sys.argv[1])):
    rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), short(r["Kernel_Name"]), r.get("Queue_Id", "?")))
rows.sort()
skins = [i for i, r in enumerate(rows) if r[2] == "skin"]
if len(skins) < 40:
    sys.exit("trace too short")
mid = skins[len(skins) // 2: len(skins) // 2 + 4]
for n, i in enumerate(mid[:-1]):
    s0 = rows[i][0]
    nxt = rows[mid[n + 1]][0]
    print("step (period %.3f ms)" % ((nxt - s0) / 1e6))
    for a, b, k, q in rows:
        if a >= s0 - 40000 and a < nxt - 40000:
            print("   q%-3s %-9s %+8.3f .. %+8.3f  (%.3f)" % (q, k, (a - s0) / 1e6, (b - s0) / 1e6, (b - a) / 1e6))

# chain gaps over the middle half of the run
def nextOf(key, t, strict=True):
    for a, b, k, q in rows:
        if k == key and a >= t:
            return a, b
    return None
acc = {}
lo, hi = skins[len(skins) // 4], skins[3 * len(skins) // 4]
for i in skins[len(skins) // 4: 3 * len(skins) // 4]:
    s0, s1 = rows[i][0], rows[i][1]
    m0 = nextOf("move0", s0 - 60000)
    if not m0: continue
    hv = nextOf("heavy", m0[0]); gr = nextOf("group", m0[0])
    if not hv or not gr: continue
    ps = nextOf("pose", gr[0])
    if not ps: continue
    sk = nextOf("skin", s0 + 1)
    mv = nextOf("move0", m0[0] + 1)
    if not sk or not mv: continue
    vals = {"move0 length": m0[1] - m0[0], "move0 end -> heavy start": hv[0] - m0[1], "move0 end -> group start": gr[0] - m0[1],
            "group length": gr[1] - gr[0], "heavy length": hv[1] - hv[0], "max(group, heavy) end -> pose start": ps[0] - max(gr[1], hv[1]),
            "pose length": ps[1] - ps[0], "pose end -> next skin start": sk[0] - ps[1], "skin end -> next skin start": sk[0] - s1,
            "pose end -> next move0 start": mv[0] - ps[1], "skin length": s1 - s0, "period": sk[0] - s0,
            "move0 start -> pose end (chain)": ps[1] - m0[0], "max(group, heavy) end -> next move0 start": mv[0] - max(gr[1], hv[1]),
            "move0 start -> next move0 start": mv[0] - m0[0], "pose end -> the skin launch that reads it": (nextOf("skin", ps[1]) or (ps[1], 0))[0] - ps[1]}
    for k, v in vals.items():
        acc.setdefault(k, []).append(v)
print("means over %d steps (us):" % len(next(iter(acc.values()))))
for k, v in acc.items():
    v = sorted(v)
    print("   %-40s mean %8.1f   median %8.1f   p10 %8.1f   p90 %8.1f" % (k, sum(v) / len(v) / 1e3, v[len(v) // 2] / 1e3, v[len(v) // 10] / 1e3, v[9 * len(v) // 10] / 1e3))
