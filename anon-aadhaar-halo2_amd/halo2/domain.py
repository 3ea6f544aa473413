"""Mirror of halo2_proofs::poly::domain::EvaluationDomain (v2023_01_20 [UP]) — filled in as the
device side of each method lands. Host constants only here; O(n) work is always on the device."""
