#!/bin/bash
# data-parallel call sequence with ONE rank (RCCL in the loop) beside the fused step.  LSTUR_DP_FOLD=0: stand-alone staging /
# unpacking launches; LSTUR_DP_NO_COLLECTIVE=1: without the RCCL call (the library's own share of the split sequence);
# LSTUR_DP_DEFER=0: the table sweep inside lstur_adam_step instead of under the next forward's recurrence
pj() { python -c "import sys,json; [print(sys.argv[1], json.loads(l)['value'], json.loads(l)['ms_per_step']) for l in sys.stdin if l.startswith('{')]" "$1"; }
for i in 1 2; do
  python bench.py --no-cpu-baseline --e2e-steps 0 2>/dev/null | pj "fused                     "
  python bench.py --force-dp --no-cpu-baseline --e2e-steps 0 2>/dev/null | pj "dp                        "
  LSTUR_DP_NO_COLLECTIVE=1 python bench.py --force-dp --no-cpu-baseline --e2e-steps 0 2>/dev/null | pj "dp without the RCCL call   "
  LSTUR_DP_FOLD=0 python bench.py --force-dp --no-cpu-baseline --e2e-steps 0 2>/dev/null | pj "dp, stage / unpack launches"
  LSTUR_DP_DEFER=0 python bench.py --force-dp --no-cpu-baseline --e2e-steps 0 2>/dev/null | pj "dp, sweep in adam          "
done
