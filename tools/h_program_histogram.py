import sys, os, ctypes as C, collections
import numpy as np
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
import __graft_entry__ as ge, zkutil as zu, circuits, limb_program_check as LC
pkg = ge.load_package(); O = zu.Oracle(); ctx = pkg.Context(0)
shape = dict(circuits.SHAPES["full"]); shape.pop("composite"); shape["k"] = 10
c = circuits.full_aadhaar_shape(pkg.plonk, **shape)
params = pkg.kzg.ParamsKZG.setup(ctx, c.k, zu.fr_from_int(12345))
fixed = np.stack([zu.ints_to_fr(O, col) for col in c.fixed])
pk = pkg.plonk.ProvingKey(ctx, params, c.desc, fixed, c.assembly.mapping, zu.fr_from_int(99))
n = C.c_size_t(0); pkg.lib().amdzk_pk_h_program(pk.h, None, 0, C.byref(n))
w = np.zeros(n.value, np.uint32); pkg.lib().amdzk_pk_h_program(pk.h, w.ctypes.data, n.value, C.byref(n))
h = collections.Counter(LC.NAME[int(x) >> 24] for x in w)
print(len(w), dict(h))
