"""Kernel timeline of the last N dispatches of a rocprofv3 --kernel-trace run: start, duration, gap to the previous kernel."""
import csv, sys, glob, re
f = glob.glob(sys.argv[1] + '/**/*kernel_trace.csv', recursive=True)[0]
n = int(sys.argv[2]) if len(sys.argv) > 2 else 36
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
tail = rows[-n:]
t0 = int(tail[0]['Start_Timestamp'])
prev_end = None
for r in tail:
    s, e = int(r['Start_Timestamp']), int(r['End_Timestamp'])
    m = re.search(r'(\w+)<', r['Kernel_Name'])
    name = (m.group(1) if m else r['Kernel_Name'])[:28]
    targs = re.search(r'<([^>]*)>', r['Kernel_Name'])
    print(f"{(s - t0) / 1e3:9.2f} dur {(e - s) / 1e3:8.2f} gap {((s - prev_end) / 1e3 if prev_end else 0):7.2f} {name:28s} <{targs.group(1) if targs else ''}> grid={r.get('Grid_Size_X', '')}x{r.get('Grid_Size_Y', '')} wg={r.get('Workgroup_Size_X', '')}")
    prev_end = e
