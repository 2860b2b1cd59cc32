"""Print the tail of a rocprofv3 --kernel-trace CSV as a timeline: start (us, relative), duration, gap to the previous
kernel's end, stream/queue, grid, name.  python tools/trace_tail.py <kernel_trace.csv> [n_last]"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
n = int(sys.argv[2]) if len(sys.argv) > 2 else 60
rows.sort(key=lambda r: int(r['Start_Timestamp']))
rows = rows[-n:]
t0 = int(rows[0]['Start_Timestamp'])
prev_end = t0
for r in rows:
    s, e = int(r['Start_Timestamp']), int(r['End_Timestamp'])
    name = r['Kernel_Name']
    name = name[:name.index('(')] if '(' in name else name
    print('%9.1f us  dur %8.1f  gap %7.1f  q%-3s grid %-8s %s' % ((s - t0) / 1e3, (e - s) / 1e3, (s - prev_end) / 1e3, r.get('Queue_Id', '?'),
                                                                  r.get('Grid_Size_X', r.get('Grid_Size', '?')), name[-70:]))
    prev_end = max(prev_end, e)
