"""CPU oracle (test infrastructure only). See g2048_oracle.h."""
