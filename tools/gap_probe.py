"""Developer tool: idle gaps of the GPU inside the bench step, from a rocprofv3 kernel trace.
    rocprofv3 --kernel-trace --output-format csv -d gpurun_out/gap -- python3 bench.py --no-cpu-baseline --no-extras --steps 5 --warmup 2
    python tools/gap_probe.py gpurun_out/gap
Prints, for the last steps, every kernel with the idle time before it."""
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
names = [r["Kernel_Name"] for r in rows]
# last occurrence of the sweep kernel marks a step; print from the cell-list pack before the previous sweep
idx = [i for i, n in enumerate(names) if "k_knn_pair" in n]
lo = idx[-3] if len(idx) >= 3 else 0
prev_end = None
tot_gap = tot_busy = 0
for r in rows[lo:]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    gap = (s - prev_end) / 1e3 if prev_end is not None else 0.0
    name = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "")[:40]
    print(f"{gap:8.1f} us idle | {(e - s) / 1e3:8.1f} us  {name}")
    if prev_end is not None:
        tot_gap += max(gap, 0)
    tot_busy += (e - s) / 1e3
    prev_end = max(e, prev_end or e)
print(f"busy {tot_busy:.1f} us, idle {tot_gap:.1f} us")
