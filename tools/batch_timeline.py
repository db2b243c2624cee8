"""Development aid: device timeline of the steps of a `bench.py --batch N` run from a rocprofv3 kernel trace
(rocprofv3 --kernel-trace --output-format csv -d DIR -o t -- python bench.py --batch N --no-hamming --no-cpu --no-extras).
usage: python tools/batch_timeline.py DIR [first_event] [events]"""
import csv
import glob
import sys

d = sys.argv[1]
first = int(sys.argv[2]) if len(sys.argv) > 2 else -1
count = int(sys.argv[3]) if len(sys.argv) > 3 else 60
ev = []
for path in glob.glob(d + '/**/t_kernel_trace.csv', recursive=True):
    for r in csv.DictReader(open(path)):
        ev.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name'][:40], r.get('Queue_Id', '?'), int(r.get('Grid_Size', 0) or 0)))
ev.sort()
print(len(ev), "kernel launches")
if first < 0:
    first = len(ev) // 2
t0 = ev[first][0]
busy_end = t0
for e in ev[first:first + count]:
    gap = (e[0] - busy_end) / 1e3
    busy_end = max(busy_end, e[1])
    print("%9.1f -> %9.1f us (%7.1f) gap %6.1f q%-3s grid %9d %s" % ((e[0] - t0) / 1e3, (e[1] - t0) / 1e3, (e[1] - e[0]) / 1e3, gap, e[3], e[4], e[2]))
