# same-box A/B of every scheduling / kernel-choice switch of the training step; each line: one 200-step bench run -> gpurun_out/ab_<tag>.log
run() { tag=$1; shift; env "$@" python bench.py --steps 200 --warmup 20 --no-cpu-baseline --e2e-steps 0 > gpurun_out/ab_$tag.log 2>&1 || exit 1; }
run base A=1
run nofused LSTUR_FUSED_ADAM=0
run sw32 LSTUR_SWEEP_BLOCKS=32
run sw128 LSTUR_SWEEP_BLOCKS=128
run nowg LSTUR_WG_STREAM=0
run nocand LSTUR_CAND_SPLIT=0
run noap LSTUR_ADAM_PACK=0
run gf32 LSTUR_GEMM_F32=1
run gruf32 LSTUR_GRU_F32=1
run bf16x3 LSTUR_CONV_SCHEME=bf16x3
run convf32 LSTUR_CONV_F32=1
run base2 A=1
