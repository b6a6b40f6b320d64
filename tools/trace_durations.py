"""Per-launch durations and gaps of one kernel from a rocprofv3 --kernel-trace CSV, in launch order (a developer tool).   python tools/trace_durations.py <kernel_trace.csv> <name substring> [group]"""
import csv
import sys

rows = [r for r in csv.DictReader(open(sys.argv[1])) if sys.argv[2] in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
group = int(sys.argv[3]) if len(sys.argv) > 3 else 10
dur = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in rows]
gap = [0.0] + [(int(b["Start_Timestamp"]) - int(a["End_Timestamp"])) / 1e3 for a, b in zip(rows, rows[1:])]
grid = [r.get("Grid_Size", "?") for r in rows]
print(f"{len(rows)} launches of *{sys.argv[2]}*; means per {group} launches: duration us | gap to the previous launch's end us | grid")
for i in range(0, len(rows), group):
    d, g = dur[i:i + group], gap[i:i + group]
    print(f"  launches {i:4d}-{i + len(d) - 1:4d}: {sum(d) / len(d):9.1f} | {sum(g) / len(g):9.1f} (max {max(g):9.1f}) | {grid[i]}")
