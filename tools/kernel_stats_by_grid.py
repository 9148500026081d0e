#!/usr/bin/env python3
"""rocprofv3 --kernel-trace CSV -> per (kernel, grid size) launch statistics (the --stats table averages over ALL launches of a kernel name -- the 64-env
lone-wave probes and the parity launches of bench.py with the timed ones).  usage: kernel_stats_by_grid.py <dir with *kernel_trace.csv> [min calls]  -> CSV on stdout.
A 4096-env humanoid run as two groups launches k_env_step_duo with 1024 workgroups x 64 threads = grid 65 536 (one launch of 4096 envs: 131 072; the 64-env probe: 2 048)."""
import collections, csv, glob, os, sys
import numpy as np
src = sys.argv[1]; min_calls = int(sys.argv[2]) if len(sys.argv) > 2 else 1
files = [src] if os.path.isfile(src) else glob.glob(os.path.join(src, "**", "*kernel_trace.csv"), recursive=True)
acc = collections.defaultdict(list)
for f in files:
    for r in csv.DictReader(open(f)):
        g = int(r.get("Grid_Size", r.get("Grid_Size_X", 0)) or 0)
        acc[(r["Kernel_Name"].split("(")[0][:100], g, int(r.get("Workgroup_Size", r.get("Workgroup_Size_X", 0)) or 0))].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
w = csv.writer(sys.stdout)
w.writerow(["Kernel_Name", "Grid_Size", "Workgroup_Size", "Calls", "AverageNs", "MedianNs", "MinNs", "MaxNs", "StdDevNs", "TotalNs"])
for (k, g, wg), d in sorted(acc.items(), key=lambda kv: -sum(kv[1])):
    if len(d) >= min_calls:
        a = np.array(d, dtype=np.float64)
        w.writerow([k, g, wg, len(d), "%.1f" % a.mean(), "%.1f" % np.median(a), int(a.min()), int(a.max()), "%.1f" % a.std(), int(a.sum())])
