import os, sys, subprocess
for v in ['0', '1', '3', '7', '15', '2', '4', '8']:
    env = dict(os.environ, LSTUR_GRU_DEBUG=v)
    out = subprocess.run([sys.executable, 'tools/bench_gru.py', 'one'], env=env, capture_output=True, text=True).stdout
    print('bits', v, [l for l in out.splitlines() if 'n=64 W=50 H=200' in l])
