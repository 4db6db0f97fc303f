#!/bin/bash
# SQ counter pass over the conv forward A/B (tools/bench_conv3f.py); summaries -> gpurun_out/pmc_conv_*.txt
set +e
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -rf /tmp/pmc1
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --kernel-trace --output-format csv -d /tmp/pmc1 -- python3 $R/tools/bench_conv3f.py 3520 2 > $R/gpurun_out/pmc_conv_run1.log 2>&1
rm -rf /tmp/pmc2
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_VMEM --kernel-trace --output-format csv -d /tmp/pmc2 -- python3 $R/tools/bench_conv3f.py 3520 2 > $R/gpurun_out/pmc_conv_run2.log 2>&1
python3 - <<'PY'
import csv, glob, collections, os
R = os.environ['GRAFT_REPO_ROOT']
for d in ('/tmp/pmc1', '/tmp/pmc2'):
    f = glob.glob(d + '/**/*counter_collection.csv', recursive=True)
    if not f:
        print('no csv in', d, glob.glob(d + '/**', recursive=True)[:10]); continue
    tot = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
    for r in csv.DictReader(open(f[0])):
        k = r['Kernel_Name'][:60]
        tot[k][r['Counter_Name']] += float(r['Counter_Value'])
        cnt[(k, r['Counter_Name'])] += 1
    with open(R + '/gpurun_out/pmc_conv_' + os.path.basename(d) + '.txt', 'w') as o:
        for k in tot:
            if 'conv3' not in k and 'gather' not in k:
                continue
            o.write(k + '\n')
            for c, v in sorted(tot[k].items()):
                o.write('   %-28s %16.0f per launch (%d launches)\n' % (c, v / cnt[(k, c)], cnt[(k, c)]))
PY
cat $R/gpurun_out/pmc_conv_pmc1.txt $R/gpurun_out/pmc_conv_pmc2.txt
