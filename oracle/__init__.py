"""CPU oracle for the accelerated path -- test infrastructure only (see gcn_oracle.py header)."""
