#!/bin/bash
# kernel trace of a few bench steps (extra arguments go to bench.py, e.g. --force-dp); prints the timeline of the last step (start, duration, stream/queue, kernel)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -rf /tmp/tr1
rocprofv3 --kernel-trace --output-format csv -d /tmp/tr1 -- python3 $R/bench.py --steps 6 --warmup 3 --no-cpu-baseline --no-profile "$@" > $R/gpurun_out/trace_run.log 2>&1
python3 - <<'PY'
import csv, glob, os
R = os.environ['GRAFT_REPO_ROOT']
f = glob.glob('/tmp/tr1/**/*kernel_trace.csv', recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
# last step: find the last 'adam_table' kernel and walk back to the previous one
idx = [i for i, r in enumerate(rows) if 'adam_table' in r['Kernel_Name']]
a, b = idx[-2] + 1, idx[-1] + 1
t0 = int(rows[a]['Start_Timestamp'])
with open(R + '/gpurun_out/trace_step.txt', 'w') as o:
    for r in rows[a:b]:
        s, e = int(r['Start_Timestamp']) - t0, int(r['End_Timestamp']) - t0
        o.write('%8.1f %8.1f  q%-3s %s\n' % (s / 1e3, (e - s) / 1e3, r.get('Queue_Id', '?'), r['Kernel_Name'][:70]))
print(open(R + '/gpurun_out/trace_step.txt').read())
PY
