#!/bin/bash
# data-parallel call sequence with ONE rank (RCCL in the loop): dense table sweep against the lazy update, beside the fused step
pj() { python -c "import sys,json; [print(sys.argv[1], json.loads(l)['value'], json.loads(l)['ms_per_step']) for l in sys.stdin if l.startswith('{')]" "$1"; }
for i in 1 2; do
  python bench.py --force-dp --no-cpu-baseline --e2e-steps 0 2>/dev/null | pj "dp dense"
  LSTUR_LAZY_ADAM=1 python bench.py --force-dp --no-cpu-baseline --e2e-steps 0 2>/dev/null | pj "dp lazy "
  LSTUR_LAZY_ADAM=1 python bench.py --force-dp --fresh-users --warmup 400 --no-cpu-baseline --e2e-steps 0 2>/dev/null | pj "dp lazy fresh users"
  python bench.py --no-cpu-baseline --e2e-steps 0 2>/dev/null | pj "fused   "
done
