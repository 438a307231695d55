"""CPU oracle (test infrastructure only).  See oracle/rva_oracle.c for the pinning status."""
