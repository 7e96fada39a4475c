"""Timeline of one CG iteration from a rocprofv3 --kernel-trace CSV: kernel, stream/queue, start offset, duration, gap to the
previous kernel's end.  usage: python tools/trace_gaps.py <kernel_trace.csv> [first_kernel_substring] [n_rows]"""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
key = sys.argv[2] if len(sys.argv) > 2 else "cgm_update_kernel"
n = int(sys.argv[3]) if len(sys.argv) > 3 else 40
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
starts = [i for i, r in enumerate(rows) if key in r["Kernel_Name"]]
i0 = starts[len(starts) // 2]           # an iteration from the middle of the run
t0, prev_end = int(rows[i0]["Start_Timestamp"]), None
for r in rows[i0:i0 + n]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    gap = "" if prev_end is None else f"{(s - prev_end) / 1e3:8.1f}"
    try:
        wgs = f"{int(r['Grid_Size_X']) // max(int(r['Workgroup_Size_X']), 1):5d} wg"
    except (KeyError, ValueError):
        wgs = "    ? wg"
    print(f"{(s - t0) / 1e3:9.1f} us  dur {(e - s) / 1e3:8.1f} us  gap {gap:>8}  q{r.get('Queue_Id', '?')} {wgs}  {r['Kernel_Name'][:90]}")
    prev_end = max(e, prev_end or e)
