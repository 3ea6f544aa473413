"""Host-side mirror of the halo2_proofs modules on the hot path (names follow upstream)."""
