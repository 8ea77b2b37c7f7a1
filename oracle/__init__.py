"""CPU oracle: test infrastructure only (see oracle/vo_oracle.h).  Never imported by the shipped package."""
