"""The torch-CPU (autograd) re-derivation lives with the oracle (oracle/torch_ref.py: test infrastructure and the torch
leg of bench.py's cpu_baseline); tests keep importing it under this name."""
from oracle.torch_ref import *  # noqa: F401,F403
