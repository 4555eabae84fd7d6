"""CPU oracle (test infrastructure only) -- see pp_oracle.py / pp_oracle.c headers."""
