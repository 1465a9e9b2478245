"""CPU oracle of the Panacea denoising hot path — test infrastructure only (see panacea_oracle.py)."""
