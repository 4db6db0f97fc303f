set -x
python bench.py --steps 200 --warmup 20 > gpurun_out/r03_bench.json 2> gpurun_out/r03_bench.err
python bench.py --workload lstur_con_b64 --steps 200 --warmup 20 --no-cpu-baseline > gpurun_out/r03_bench_lstur_con_b64.json 2> gpurun_out/r03_con.err
python bench.py --workload mhsa_b64 --steps 200 --warmup 20 --no-cpu-baseline > gpurun_out/r03_bench_mhsa_b64.json 2> gpurun_out/r03_mhsa.err
python bench.py --workload lstur_ini_b512_1Musers --steps 60 --warmup 10 --no-cpu-baseline --e2e-steps 0 > gpurun_out/r03_bench_lstur_ini_b512_1Musers.json 2> gpurun_out/r03_b512.err
python bench.py --workload lstur_ini_b512_1Musers --steps 60 --warmup 10 --no-cpu-baseline --e2e-steps 0 --fresh-users > gpurun_out/r03_bench_lstur_ini_b512_1Musers_fresh_users.json 2> gpurun_out/r03_b512f.err
python bench.py --workload mhsa_b512_1Musers --steps 60 --warmup 10 --no-cpu-baseline --e2e-steps 0 > gpurun_out/r03_bench_mhsa_b512_1Musers.json 2> gpurun_out/r03_mhsa512.err
for f in gpurun_out/r03_bench*.json; do python -c "
import json,sys;d=json.load(open('$f'));print('$f',d['value'],d['ms_per_step'],d['roofline']['kernel'],d['roofline']['frac'],d['roofline_gather']['frac'],d.get('traffic_stale'))"; done
