#!/bin/bash
# FETCH_SIZE of the conv forward kernels in the micro-benchmark: conv3_fwd2 against the tap-sharing conv3_fwd3 (LSTUR_CONV_FWD3=1)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for v in 0 1; do
  rm -rf /tmp/pf$v
  if [ $v = 1 ]; then export LSTUR_CONV_FWD3=1; fi
  rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d /tmp/pf$v -- python3 $R/tools/bench_conv3f.py 3520 2 > $R/gpurun_out/pmc_fetch_run$v.log 2>&1
done
python3 - <<'PY'
import csv, glob, collections
for v in (0, 1):
    f = glob.glob('/tmp/pf%d/**/*counter_collection.csv' % v, recursive=True)[0]
    tot, cnt = collections.Counter(), collections.Counter()
    for r in csv.DictReader(open(f)):
        if r['Counter_Name'] == 'FETCH_SIZE' and 'conv3_fwd' in r['Kernel_Name'] and 'fix' not in r['Kernel_Name']:
            k = r['Kernel_Name'][:40]
            tot[k] += float(r['Counter_Value']); cnt[k] += 1
    for k in tot:
        print('run %d  %-42s fetch %.1f MB per launch (%d launches; FETCH_SIZE KB x 2: gfx950)' % (v, k, 2 * 1024 * tot[k] / cnt[k] / 1e6, cnt[k]))
PY
