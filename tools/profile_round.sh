#!/bin/bash
# All profile passes of a round for the default bench workload; summaries -> gpurun_out/<tag>_*.  usage: profile_round.sh TAG
TAG=${1:-r03}
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
CMD="python3 $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-profile"
rm -rf /tmp/p_stats /tmp/p_fetch /tmp/p_write /tmp/p_mfma
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p_stats -- $CMD > $R/gpurun_out/${TAG}_stats_run.log 2>&1
cp "$(find /tmp/p_stats -name '*kernel_stats.csv' | head -1)" $R/gpurun_out/${TAG}_kernel_stats.csv
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d /tmp/p_fetch -- $CMD > $R/gpurun_out/${TAG}_fetch_run.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d /tmp/p_write -- $CMD > $R/gpurun_out/${TAG}_write_run.log 2>&1
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d /tmp/p_mfma -- $CMD > $R/gpurun_out/${TAG}_mfma_run.log 2>&1
FD=$(dirname "$(find /tmp/p_fetch -name '*counter_collection.csv' | head -1)")
WD=$(dirname "$(find /tmp/p_write -name '*counter_collection.csv' | head -1)")
python3 $R/tools/pmc_traffic.py "$FD" "$WD" $R/gpurun_out/${TAG}_pmc_traffic.json
python3 - <<PY
import csv, glob, json, collections
f = glob.glob('/tmp/p_mfma/**/*counter_collection.csv', recursive=True)[0]
kt = glob.glob('/tmp/p_mfma/**/*kernel_trace.csv', recursive=True)
dur = collections.defaultdict(list)
if kt:
    for r in csv.DictReader(open(kt[0])):
        dur[r['Kernel_Name']].append((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3)
tot = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
for r in csv.DictReader(open(f)):
    tot[r['Kernel_Name']][r['Counter_Name']] += float(r['Counter_Value']); cnt[(r['Kernel_Name'], r['Counter_Name'])] += 1
out = {}
for k, v in tot.items():
    if 'SQ_VALU_MFMA_BUSY_CYCLES' not in v or v['SQ_VALU_MFMA_BUSY_CYCLES'] == 0:
        continue
    busy = v['SQ_VALU_MFMA_BUSY_CYCLES'] / cnt[(k, 'SQ_VALU_MFMA_BUSY_CYCLES')]
    gui = v['GRBM_GUI_ACTIVE'] / cnt[(k, 'GRBM_GUI_ACTIVE')]
    us = sum(dur[k]) / max(1, len(dur[k])) if k in dur else None
    out[k[:80]] = dict(mfma_busy_cycles=busy, grbm_gui_active=gui, avg_us_under_pmc=us, mfma_util=round(busy / (gui / 8 * 1024), 4),
                       clock_ghz=round(gui / 8 / (us * 1e3), 3) if us else None)
json.dump(dict(command='rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace -- python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-profile',
               method='per-launch means; SQ_VALU_MFMA_BUSY_CYCLES sums over the 1024 SIMDs (32 cycles per v_mfma_f32_32x32x16_bf16); GRBM_GUI_ACTIVE sums the 8 XCDs; mfma_util = busy / (GRBM_GUI_ACTIVE / 8 * 1024)',
               kernels=out), open('$R/gpurun_out/${TAG}_pmc_mfma.json', 'w'), indent=1)
for k, v in out.items():
    print('%-70s util %.3f clock %s' % (k[:70], v['mfma_util'], v['clock_ghz']))
PY
echo done
