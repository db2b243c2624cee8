"""Development aid: timeline of one single-frame call of examples/orb_demo from a rocprofv3 run
(rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d DIR -o one -- examples/orb_demo --time IN 50)."""
import csv
import sys

d = sys.argv[1]
ev = []
for r in csv.DictReader(open(d + '/one_kernel_trace.csv')):
    ev.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name'][:34]))
try:
    for r in csv.DictReader(open(d + '/one_memory_copy_trace.csv')):
        ev.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), 'COPY ' + r['Direction']))
except FileNotFoundError:
    pass
ev.sort()
idx = [i for i, e in enumerate(ev) if 'pyramid_few' in e[2]]
i0 = idx[20]
t0 = ev[i0 - 1][0]
for e in ev[i0 - 1:i0 + 8]:
    print("%8.1f -> %8.1f us  (%6.1f)  %s" % ((e[0] - t0) / 1e3, (e[1] - t0) / 1e3, (e[1] - e[0]) / 1e3, e[2]))
