set -e
run() { tag=$1; shift; env "$@" python bench.py --steps 200 --warmup 20 --no-cpu-baseline --e2e-steps 0 > gpurun_out/pp_$tag.json 2> gpurun_out/pp_$tag.err || { tail -5 gpurun_out/pp_$tag.err; exit 1; }; python -c "
import json,sys;d=json.load(open('gpurun_out/pp_$tag.json'));print('$tag',d['value'],d['ms_per_step'], {k:v['avg_us'] for k,v in d.get('kernels',{}).items()} if 'kernels' in d else '')"; }
run base240 LSTUR_LIB=mnexp_amd/liblstur_hip_ablate.so LSTUR_PROBE_NCU=240
run fwd240 LSTUR_LIB=mnexp_amd/liblstur_hip_ablate.so LSTUR_PIPE_PROBE=1 LSTUR_PROBE_NCU=240
run bwd240 LSTUR_LIB=mnexp_amd/liblstur_hip_ablate.so LSTUR_PIPE_PROBE=2 LSTUR_PROBE_NCU=240
run both240 LSTUR_LIB=mnexp_amd/liblstur_hip_ablate.so LSTUR_PIPE_PROBE=3 LSTUR_PROBE_NCU=240
run both232 LSTUR_LIB=mnexp_amd/liblstur_hip_ablate.so LSTUR_PIPE_PROBE=3 LSTUR_PROBE_NCU=232
