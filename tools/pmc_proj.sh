#!/bin/bash
# counter passes over the mhsa projection GEMMs (tools/bench_proj.py); summary -> gpurun_out/pmc_proj.txt
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
i=0
for set in "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_ACTIVE_INST_VALU" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1)); rm -rf /tmp/pp$i
  rocprofv3 --pmc $set --kernel-trace --output-format csv -d /tmp/pp$i -- python3 $R/tools/bench_proj.py > $R/gpurun_out/pmc_proj_run$i.log 2>&1
done
python3 - <<'PY'
import csv, glob, collections, os
R = os.environ['GRAFT_REPO_ROOT']
with open(R + '/gpurun_out/pmc_proj.txt', 'w') as o:
    for i in (1, 2, 3, 4):
        f = glob.glob('/tmp/pp%d/**/*counter_collection.csv' % i, recursive=True)
        if not f:
            o.write('no csv for pass %d\n' % i); continue
        tot = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
        for r in csv.DictReader(open(f[0])):
            if 'gemm_bf16' not in r['Kernel_Name']:
                continue
            k = r['Kernel_Name'][:70]
            tot[k][r['Counter_Name']] += float(r['Counter_Value']); cnt[(k, r['Counter_Name'])] += 1
        for k in tot:
            o.write(k + '\n')
            for c, v in sorted(tot[k].items()):
                o.write('   %-28s %16.0f per launch (%d launches)\n' % (c, v / cnt[(k, c)], cnt[(k, c)]))
print(open(R + '/gpurun_out/pmc_proj.txt').read())
PY
