"""Synthetic workloads with the reference circuits' shape: what bench.py measures and the parity tests prove.

The reference's witness synthesis is Rust (`Circuit::synthesize`, /root/reference/src/lib.rs:328-397 with the
big_uint / RSA / SHA-256 chips) and out of reach here; these builders produce SATISFYING witnesses for constraint
systems with the same column / gate / lookup / permutation budget, written against the product's ConstraintSystem
mirror the way the reference's circuits are written against halo2's (`configure` + `synthesize`):

  rsa_sha256_shape   — TestRSASignatureWithHashCircuit1 (/root/reference/src/lib.rs:263-274, 295-326):
                       NUM_ADVICE vertical-gate columns (halo2-base FlexGate q*(a + b*c - d) over 4 rotations),
                       range-lookup advice columns (one table of 2^lookup_bits), two-column "spread" lookups (SHA),
                       one constants column, two instance columns, every advice column in the permutation.
  full_aadhaar_shape — the composite AadhaarQRVerifierCircuit (/root/reference/src/aadhaar_verifier_circuit.rs:49-56):
                       the above plus IdentityCircuit, TimestampCircuit and SquareCircuit columns and gates.

The exact column counts of halo2-base / halo2-dynamic-sha256 at the reference's pins are restated from the
configuration constants at src/lib.rs:263-274 (unverifiable here, no Rust
toolchain); this is a shape proxy of the "full Aadhaar circuit", not its synthesized witness.

One circuit = one layout (fixed columns, copy constraints; `layout_seed`) and any number of witnesses
(`Circuit.witness(seed)`): a batch of independent proofs shares one proving key, as BASELINE config 4 asks.
"""
import numpy as np

from .halo2 import plonk

R = 21888242871839275222246405745257275088548364400416034343698204186575808495617

# bench.py / test shapes (BASELINE.json configs[1] and configs[2])
SHAPES = {
    # the reference's own configuration (src/lib.rs:263-274, k = 15 at src/lib.rs:444)
    "k15": dict(k=15, num_advice=80, num_lookup_advice=16, lookup_bits=12, num_spread=8, spread_bits=8),
    # BASELINE.json configs[1] "k~18": same area, 8x fewer gate columns
    "k18": dict(k=18, num_advice=10, num_lookup_advice=2, lookup_bits=12, num_spread=1, spread_bits=8),
    # BASELINE.json configs[2]: the composite AadhaarQRVerifierCircuit budget: k15 + Identity / Timestamp / Square
    "full": dict(k=15, num_advice=80, num_lookup_advice=16, lookup_bits=12, num_spread=8, spread_bits=8, composite=True),
}


class Circuit:
    """A configured constraint system with its fixed columns, copy constraints and one assigned witness."""

    def __init__(self, cs, k):
        self.cs, self.k, self.n = cs, k, 1 << k
        self.desc = cs.describe(k)
        self.usable = self.n - (self.desc["blinding_factors"] + 1)
        self.fixed = [[0] * self.n for _ in range(cs.num_fixed)]
        self.advice = [[0] * self.n for _ in range(cs.num_advice)]
        self.instances = [[] for _ in range(cs.num_instance)]
        self.assembly = None
        self.copies = []  # (perm column, row, perm column, row) in the order they were made
        self._witness_fn = None

    def perm_index(self, col):
        return self.cs.permutation_columns.index(col)

    def copy(self, c1, r1, c2, r2):
        self.copies.append((self.perm_index(c1), r1, self.perm_index(c2), r2))
        self.assembly.copy(*self.copies[-1])

    def value(self, col, row):
        if col.kind == 0:
            return self.advice[col.index][row]
        if col.kind == 1:
            return self.fixed[col.index][row]
        v = self.instances[col.index]
        return v[row] if row < len(v) else 0

    def witness(self, seed):
        """(advice, instances) of another satisfying witness of this same layout (same fixed columns, same copy
        constraints, so the same proving key): what a second proof of a batch proves."""
        if self._witness_fn is None:
            raise ValueError("this circuit has a single hand-assigned witness")
        return self._witness_fn(seed)


def _spread(v, bits):
    out = 0
    for bit in range(bits):
        out |= ((v >> bit) & 1) << (2 * bit)
    return out


def rsa_sha256_shape(k=15, seed=7, num_advice=80, num_lookup_advice=16, lookup_bits=12, num_spread=8, spread_bits=8,
                     configure_extra=None, layout_seed=7):
    """`seed` picks the witness values, `layout_seed` the positions of the copy constraints and the constants.
    `configure_extra(cs)` may add further sub-circuit configurations after this one (as AadhaarQRVerifierCircuit::configure
    does) and returns the function that assigns their cells."""
    cs = plonk.ConstraintSystem()
    gate_cols = [cs.advice_column() for _ in range(num_advice)]
    sels = [cs.selector() for _ in range(num_advice)]
    lk_cols = [cs.advice_column() for _ in range(num_lookup_advice)]
    sp_dense = [cs.advice_column() for _ in range(num_spread)]
    sp_spread = [cs.advice_column() for _ in range(num_spread)]
    t_rng, t_dense, t_spread, konst = cs.fixed_column(), cs.fixed_column(), cs.fixed_column(), cs.fixed_column()
    inst = [cs.instance_column(), cs.instance_column()]
    for col in gate_cols + lk_cols + sp_dense + sp_spread + [konst] + inst:
        cs.enable_equality(col)
    for col, s in zip(gate_cols, sels):
        cs.create_gate(lambda m, col=col, s=s: [m.query_selector(s) * (m.query_advice(col, 0) + m.query_advice(col, 1) * m.query_advice(col, 2)
                                                                     - m.query_advice(col, 3))])
    for col in lk_cols:
        cs.lookup(lambda m, col=col: [(m.query_advice(col, 0), m.query_fixed(t_rng, 0))])
    for dcol, scol in zip(sp_dense, sp_spread):
        cs.lookup(lambda m, dcol=dcol, scol=scol: [(m.query_advice(dcol, 0), m.query_fixed(t_dense, 0)),
                                                   (m.query_advice(scol, 0), m.query_fixed(t_spread, 0))])
    synthesize_extra = configure_extra(cs) if configure_extra else None
    c = Circuit(cs, k)
    n, u = c.n, c.usable
    c.assembly = plonk.Assembly(n, len(cs.permutation_columns))
    tr = min(1 << lookup_bits, u)
    ts = min(1 << spread_bits, u)
    spread_tab = [_spread(i, spread_bits) for i in range(ts)]

    # ---- layout: tables, selectors, copy constraints (positions only), constants
    for i in range(u):
        c.fixed[t_rng.index][i] = i if i < tr else 0
        c.fixed[t_dense.index][i] = i if i < ts else 0
        c.fixed[t_spread.index][i] = spread_tab[i] if i < ts else 0
    ngates = (u - 3) // 4  # every 4th row starts a gate (a, b, c, d) with d = a + b*c
    for s in sels:
        col_s = c.fixed[s.index]
        for g in range(ngates):
            col_s[4 * g] = 1
    lay = np.random.RandomState(layout_seed)
    used = set()  # every cell takes part in at most one copy, so fixing up a copied-to cell never disturbs another

    used_in = {}  # gate column -> cells of it in `used`

    def free_gate(ci):
        if ngates <= 0 or used_in.get(ci, 0) >= 4 * ngates:
            raise ValueError("rsa_sha256_shape: k = %d with %d gate columns has too few gate rows for the layout's copy constraints "
                             "(it needs 72 free gates, column %d is full): use a larger k or more columns" % (k, num_advice, ci))
        while True:
            g = int(lay.randint(ngates))
            if all((ci, 4 * g + o) not in used for o in range(4)):
                for o in range(4):
                    used.add((ci, 4 * g + o))
                used_in[ci] = used_in.get(ci, 0) + 4
                return 4 * g

    # assignments applied to every witness, in order: ("gate", ci, r0, off, source) sets gate input `off` of the gate
    # at r0 of gate column ci from `source` and recomputes that gate's output
    fixups = []
    for t in range(max(8, u // 8)):  # advice <-> advice
        i1, i2 = int(lay.randint(num_advice)), int(lay.randint(num_advice))
        r1, r2 = free_gate(i1), free_gate(i2)
        fixups.append((i2, r2, 0, ("gate", i1, r1)))
        c.copy(gate_cols[i1], r1, gate_cols[i2], r2)
    for t in range(16):  # range-checked cells feeding gates
        li, gi = t % num_lookup_advice, t % num_advice
        r0 = free_gate(gi)
        fixups.append((gi, r0, 1, ("lookup", li, t)))
        c.copy(lk_cols[li], t, gate_cols[gi], r0 + 1)
    for t in range(8):  # constants
        gi = t % num_advice
        r0 = free_gate(gi)
        kv = int(lay.randint(0, 1 << 62, dtype=np.int64))
        c.fixed[konst.index][t] = kv
        fixups.append((gi, r0, 2, ("const", kv)))
        c.copy(konst, t, gate_cols[gi], r0 + 2)
    publics = []
    for i in range(32):  # public inputs: 32 modulus limbs / 32 hash bytes in the reference (src/lib.rs:389-394)
        gi = i % num_advice
        r0 = free_gate(gi)
        li = i % num_lookup_advice
        publics.append((gi, r0 + 1, li, 20 + i))
        c.copy(inst[0], i, gate_cols[gi], r0 + 1)
        c.copy(inst[1], i, lk_cols[li], 20 + i)

    # ---- witness: values for one seed (canonical Python ints; products need more than 64 bits)
    def make_witness(wseed):
        rnd = np.random.RandomState(wseed)
        advice = [[0] * n for _ in range(cs.num_advice)]
        gvals = []
        for ci in range(num_advice):
            vals = rnd.randint(0, 1 << 62, size=u, dtype=np.int64).tolist()
            for r0 in range(0, 4 * ngates, 4):
                vals[r0 + 3] = (vals[r0] + vals[r0 + 1] * vals[r0 + 2]) % R
            gvals.append(vals)
        lvals = [rnd.randint(0, tr, size=u).tolist() for _ in lk_cols]
        for (ci, r0, off, src) in fixups:
            v = gvals[src[1]][src[2]] if src[0] == "gate" else lvals[src[1]][src[2]] if src[0] == "lookup" else src[1]
            vals = gvals[ci]
            vals[r0 + off] = v % R
            vals[r0 + 3] = (vals[r0] + vals[r0 + 1] * vals[r0 + 2]) % R
        for ci, col in enumerate(gate_cols):
            advice[col.index][:u] = gvals[ci]
        for li, col in enumerate(lk_cols):
            advice[col.index][:u] = lvals[li]
        for dcol, scol in zip(sp_dense, sp_spread):
            dv = rnd.randint(0, ts, size=u).tolist()
            advice[dcol.index][:u] = dv
            advice[scol.index][:u] = [spread_tab[v] for v in dv]
        instances = [[] for _ in range(cs.num_instance)]
        instances[inst[0].index] = [gvals[gi][r] for (gi, r, _, _) in publics]
        instances[inst[1].index] = [lvals[li][r] for (_, _, li, r) in publics]
        if synthesize_extra:
            synthesize_extra(c, advice, instances)
        return advice, instances

    c._witness_fn = make_witness
    c.advice, c.instances = make_witness(seed)
    return c


def full_aadhaar_shape(k=15, seed=7, signal=5, **kw):
    """Column/gate budget of the composite AadhaarQRVerifierCircuit
    (/root/reference/src/aadhaar_verifier_circuit.rs:49-56): the RSA-SHA256 shape, then
      IdentityCircuit  (/root/reference/src/conditional_secrets.rs:81-190) 20 advice, 1 selector, 8 gates
                       (12 polynomials: 4 booleanity, age/gender/pincode reveals, 5 state bytes),
      TimestampCircuit (/root/reference/src/timestamp.rs:58-138) 7 advice columns no gate queries (its
                       range gates are commented out there; its selector is never used in a gate, so
                       halo2's selector compression gives it no fixed column and neither do we),
      SquareCircuit    (/root/reference/src/signal.rs:27-76) 2 equality-enabled advice, 1 instance,
                       1 selector, gate s*(a1 - a0^2).
    Each sub-circuit assigns one row (row 0 of its own columns), as the reference's regions do."""

    def configure_extra(cs):
        # IdentityCircuit
        names = ["reveal_age", "age", "qr_age", "reveal_gender", "gender", "qr_gender", "reveal_pincode", "pincode", "qr_pincode",
                 "reveal_state"] + ["state%d" % i for i in range(5)] + ["qr_state%d" % i for i in range(5)]
        idc = {nm: cs.advice_column() for nm in names}
        s_id = cs.selector()
        for nm in ("reveal_age", "reveal_gender", "reveal_pincode", "reveal_state"):
            cs.create_gate(lambda m, nm=nm: [m.query_selector(s_id) * m.query_advice(idc[nm], 0)
                                             * (m.query_advice(idc[nm], 0) - plonk.Expression.constant(1))])
        cs.create_gate(lambda m: [m.query_selector(s_id) * (m.query_advice(idc["age"], 0)
                                                            - m.query_advice(idc["reveal_age"], 0) * m.query_advice(idc["qr_age"], 0))])
        cs.create_gate(lambda m: [m.query_selector(s_id) * (m.query_advice(idc["gender"], 0) - m.query_advice(idc["qr_gender"], 0))])
        cs.create_gate(lambda m: [m.query_selector(s_id) * (m.query_advice(idc["pincode"], 0) - m.query_advice(idc["qr_pincode"], 0))])
        cs.create_gate(lambda m: [m.query_selector(s_id) * (m.query_advice(idc["state%d" % i], 0) - m.query_advice(idc["qr_state%d" % i], 0))
                                  for i in range(5)])
        # TimestampCircuit: year, month, day, hour, minute, second, timestamp
        ts = [cs.advice_column() for _ in range(7)]
        # SquareCircuit
        sq = [cs.advice_column(), cs.advice_column()]
        sq_inst = cs.instance_column()
        s_sq = cs.selector()
        for col in sq + [sq_inst]:
            cs.enable_equality(col)
        cs.create_gate(lambda m: [m.query_selector(s_sq) * (m.query_advice(sq[1], 0) - m.query_advice(sq[0], 0) * m.query_advice(sq[0], 0))])

        def synthesize(c, advice, instances):
            c.fixed[s_id.index][0] = 1
            vals = {"reveal_age": 1, "age": 1, "qr_age": 1, "reveal_gender": 1, "gender": 77, "qr_gender": 77,
                    "reveal_pincode": 0, "pincode": 110051, "qr_pincode": 110051, "reveal_state": 1}
            for i, ch in enumerate(b"Delhi"):
                vals["state%d" % i] = vals["qr_state%d" % i] = ch
            for nm, v in vals.items():
                advice[idc[nm].index][0] = v
            for col, v in zip(ts, (2019, 3, 8, 5, 30, 0, 1552023000)):
                advice[col.index][0] = v
            c.fixed[s_sq.index][0] = 1
            advice[sq[0].index][0] = signal % R
            advice[sq[1].index][0] = signal * signal % R
            instances[sq_inst.index] = []

        return synthesize

    return rsa_sha256_shape(k=k, seed=seed, configure_extra=configure_extra, **kw)


def make(shape_name, seed=7, **override):
    """The circuit of a named SHAPES entry (bench.py --shape)."""
    shape = dict(SHAPES[shape_name])
    shape.update(override)
    fn = full_aadhaar_shape if shape.pop("composite", False) else rsa_sha256_shape
    return fn(seed=seed, **shape)


def canon_limbs(cols):
    """list of columns of canonical Python ints -> (ncols, n, 4) uint64 little-endian limbs (Fr::from_raw input)."""
    out = np.zeros((len(cols), len(cols[0]) if cols else 0, 4), dtype=np.uint64)
    for c, col in enumerate(cols):
        if all(v < (1 << 63) for v in col):
            out[c, :, 0] = np.array(col, dtype=np.uint64)
            continue
        out[c] = np.frombuffer(b"".join(v.to_bytes(32, "little") for v in col), dtype=np.uint64).reshape(-1, 4)
    return out


def check_satisfied(c, rows=None, advice=None, instances=None):
    """MockProver-style check of gates, lookups and copy constraints on the usable rows (Python ints)."""
    desc, n, u = c.desc, c.n, c.usable
    advice = c.advice if advice is None else advice
    instances = c.instances if instances is None else instances
    inst = [list(v) + [0] * (n - len(v)) for v in instances]

    def ev(e, row):
        op = e[0]
        if op == "const":
            return e[1]
        if op == "fixed":
            return c.fixed[e[1]][(row + e[2]) % n]
        if op == "advice":
            return advice[e[1]][(row + e[2]) % n]
        if op == "instance":
            return inst[e[1]][(row + e[2]) % n]
        if op == "neg":
            return (-ev(e[1], row)) % R
        if op == "sum":
            return (ev(e[1], row) + ev(e[2], row)) % R
        if op == "product":
            return ev(e[1], row) * ev(e[2], row) % R
        return ev(e[1], row) * e[2] % R

    def value(col, row):
        if col.kind == 0:
            return advice[col.index][row]
        if col.kind == 1:
            return c.fixed[col.index][row]
        return inst[col.index][row]

    rr = range(u) if rows is None else rows
    for gi, g in enumerate(desc["gates"]):
        for row in rr:
            assert ev(g, row) == 0, "gate %d fails at row %d" % (gi, row)
    for li, lk in enumerate(desc["lookups"]):
        table = {tuple(ev(e, row) for e in lk["tables"]) for row in range(u)}
        for row in rr:
            assert tuple(ev(e, row) for e in lk["inputs"]) in table, "lookup %d fails at row %d" % (li, row)
    cols = c.cs.permutation_columns
    for i, col in enumerate(cols):
        for row in range(n):
            pi, pj = c.assembly.mapping[i][row]
            if (pi, pj) != (i, row):
                assert value(col, row) == value(cols[pi], pj), "copy constraint fails at col %d row %d" % (i, row)
    return True
