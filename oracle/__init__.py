"""CPU oracle (test infrastructure only; see oracle/lstur_oracle.py header)."""
