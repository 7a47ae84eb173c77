"""CPU oracle for the FA-2 forward hot path -- TEST INFRASTRUCTURE ONLY (see fa2_oracle.c)."""
