"""MI355X-native backend for the Halo2/KZG prover hot path of anon-aadhaar-halo2.

The product is `libamdzk.so` (HIP kernels for gfx950 behind the C ABI in include/amdzk.h). This
package is the thin host-side mirror of the upstream interface for that path
(halo2_proofs::arithmetic / poly::domain / poly::kzg::commitment — SURVEY.md §8(b)), over ctypes.

There is no CPU fallback: importing works anywhere, but every operation needs the built library
and a gfx950 device and raises otherwise. The directory name contains a hyphen (it is the
repository's name); import it with `__graft_entry__.load_package()` which registers it as
`anon_aadhaar_halo2_amd`.
"""
from .ffi import AmdzkError, Context, build_info, lib, lib_path  # noqa: F401
from . import batch  # noqa: F401
from .halo2 import arithmetic, domain, kzg, plonk  # noqa: F401
from . import feeder, workloads  # noqa: F401
