#!/usr/bin/env python3
"""Summary of a tools/probes/wg_stamps timeline (gpurun_out/wg_stamps.csv): durations by strip kind, chunk, CU pairing."""
import collections
import statistics as st
import sys

path = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/wg_stamps.csv"
rows = [l.strip().split(",") for l in open(path) if not l.startswith("#") and "," in l]
print(open(path).readline().strip())
R = [(int(r[0]), float(r[1]), float(r[2]), int(r[3]), int(r[4]), int(r[5]), int(r[6]), int(r[7])) for r in rows]
nstrips = max(r[6] for r in R) + 1
rim = [r for r in R if r[6] in (0, nstrips - 1)]
inner = [r for r in R if r[6] not in (0, nstrips - 1)]
dur = lambda r: r[2] - r[1]
print(f"{len(R)} workgroups, start spread {max(r[1] for r in R):.1f} us, last end {max(r[2] for r in R):.1f}")
for name, grp in (("rim", rim), ("inner", inner)):
    if grp:
        d = sorted(dur(r) for r in grp)
        print(f"{name:6s} n={len(d)} dur min {d[0]:.0f} med {st.median(d):.0f} max {d[-1]:.0f}; end med {st.median(r[2] for r in grp):.0f} max {max(r[2] for r in grp):.0f}")
bychunk = collections.defaultdict(list)
for r in inner:
    bychunk[r[7]].append(dur(r))
print("inner by chunk (min/med/max):", {k: (round(min(v)), round(st.median(v)), round(max(v))) for k, v in sorted(bychunk.items())})
cu = collections.defaultdict(list)
for r in R:
    cu[(r[5], r[3], r[4])].append(r)
pairs = [sorted(v, key=lambda r: r[0]) for v in cu.values() if len(v) == 2]
first = [dur(v[0]) for v in pairs]
second = [dur(v[1]) for v in pairs]
print(f"CUs {len(cu)}, with two workgroups {len(pairs)}: lower block id dur med {st.median(first):.0f}, higher {st.median(second):.0f}")
alone = [dur(v[0]) for v in cu.values() if len(v) == 1]
if alone:
    print(f"alone on a CU: n={len(alone)} med {st.median(alone):.0f}")
hist = collections.Counter(int(dur(r) // 100) * 100 for r in R)
print("duration histogram (us):", sorted(hist.items()))
