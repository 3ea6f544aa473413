"""Circuit fixtures for the prover tests, written against the product's ConstraintSystem mirror the
way the reference's circuits are written against halo2's (`Circuit::configure` + `synthesize`).

  square_circuit      — /root/reference/src/signal.rs:27-76 (SquareCircuit: the circuit the checked-in
                        Solidity verifier is generated for), k >= 4
  lookup_circuit      — small mul/add gates + a range lookup + a 2-column (theta-compressed) lookup +
                        copy constraints across advice/fixed/instance columns
  rsa_sha256_shape    — synthetic circuit with the column/lookup/permutation budget of
                        TestRSASignatureWithHashCircuit1 (/root/reference/src/lib.rs:263-274,295-326):
                        NUM_ADVICE=80 vertical-gate columns (halo2-base FlexGate, q*(a + b*c - d) over
                        4 rotations), 16 range-lookup advice (12-bit table), 8 two-column spread
                        lookups (SHA), 1 constants column, 2 instance columns, all in the permutation
"""
import numpy as np

R = 21888242871839275222246405745257275088548364400416034343698204186575808495617


class Circuit:
    def __init__(self, cs, k):
        self.cs, self.k, self.n = cs, k, 1 << k
        self.desc = cs.describe(k)
        self.usable = self.n - (self.desc["blinding_factors"] + 1)
        self.fixed = [[0] * self.n for _ in range(cs.num_fixed)]
        self.advice = [[0] * self.n for _ in range(cs.num_advice)]
        self.instances = [[] for _ in range(cs.num_instance)]
        self.assembly = None
        self.copies = []  # (perm column, row, perm column, row) in the order they were made

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


def square_circuit(plonk, k=4, signal=5):
    cs = plonk.ConstraintSystem()
    advice = [cs.advice_column(), cs.advice_column()]
    instance = cs.instance_column()
    selector = cs.selector()
    cs.enable_equality(advice[0])
    cs.enable_equality(advice[1])
    cs.enable_equality(instance)

    def gate(meta):
        s = meta.query_selector(selector)
        a = meta.query_advice(advice[0], 0)
        sq = meta.query_advice(advice[1], 0)
        return [s * (sq - a * a)]

    cs.create_gate(gate)
    c = Circuit(cs, k)
    c.assembly = plonk.Assembly(c.n, len(cs.permutation_columns))
    c.fixed[selector.index][0] = 1
    c.advice[0][0] = signal % R
    c.advice[1][0] = signal * signal % R
    c.instances[0] = []  # the reference leaves the instance constraint commented out (signal.rs:72)
    return c


def lookup_circuit(plonk, k=5, seed=1):
    rnd = np.random.RandomState(seed)
    cs = plonk.ConstraintSystem()
    a, b, c_ = cs.advice_column(), cs.advice_column(), cs.advice_column()
    q_mul, q_add, q_rng, q_pair = cs.selector(), cs.selector(), cs.selector(), cs.selector()
    t_rng, t_x, t_y, konst = cs.fixed_column(), cs.fixed_column(), cs.fixed_column(), cs.fixed_column()
    inst = cs.instance_column()
    for col in (a, b, c_, konst, inst):
        cs.enable_equality(col)
    cs.create_gate(lambda m: [m.query_selector(q_mul) * (m.query_advice(a, 0) * m.query_advice(b, 0) - m.query_advice(c_, 0)),
                              m.query_selector(q_add) * (m.query_advice(a, 0) + m.query_advice(b, 1) - m.query_advice(c_, 0))])
    cs.lookup(lambda m: [(m.query_selector(q_rng) * m.query_advice(a, 0), m.query_fixed(t_rng, 0))])
    cs.lookup(lambda m: [(m.query_selector(q_pair) * m.query_advice(b, 0), m.query_fixed(t_x, 0)),
                         (m.query_selector(q_pair) * m.query_advice(c_, 0), m.query_fixed(t_y, 0))])
    c = Circuit(cs, k)
    c.assembly = plonk.Assembly(c.n, len(cs.permutation_columns))
    u = c.usable
    # tables (row 0 holds the all-zero entry so disabled rows look up 0 / (0,0))
    for i in range(u):
        c.fixed[t_rng.index][i] = i % 8
        c.fixed[t_x.index][i] = i % 6
        c.fixed[t_y.index][i] = (i % 6) ** 2 % R if i % 6 else 0
    row = 0
    for i in range(u // 2 - 1):  # mul rows
        x, y = int(rnd.randint(0, 1 << 30)), int(rnd.randint(0, 1 << 30))
        c.fixed[q_mul.index][row] = 1
        c.advice[a.index][row], c.advice[b.index][row], c.advice[c_.index][row] = x, y, x * y % R
        row += 1
    add_rows = []
    while row < u - 1:  # add rows use b at the next row
        c.fixed[q_add.index][row] = 1
        add_rows.append(row)
        row += 1
    for r_ in range(u):
        if c.fixed[q_mul.index][r_] == 0:
            c.advice[a.index][r_] = int(rnd.randint(0, 8))
            c.advice[b.index][r_] = int(rnd.randint(0, 6))
    for r_ in add_rows:
        c.advice[c_.index][r_] = (c.advice[a.index][r_] + c.advice[b.index][r_ + 1]) % R
    # lookups enabled where they hold by construction
    for r_ in range(u):
        if c.fixed[q_mul.index][r_] == 0:
            c.fixed[q_rng.index][r_] = 1
        if r_ == u - 1:  # last usable row: not an add row, c free -> make (b, c) a table pair
            c.fixed[q_pair.index][r_] = 1
            c.advice[c_.index][r_] = c.advice[b.index][r_] ** 2 % R
    # copy constraints: a few equalities inside advice, to a constant, and to the instance column
    c.fixed[konst.index][0] = c.advice[a.index][0]
    c.copy(konst, 0, a, 0)
    c.instances[inst.index] = [c.advice[c_.index][1], c.advice[c_.index][2]]
    c.copy(inst, 0, c_, 1)
    c.copy(inst, 1, c_, 2)
    r1, r2 = add_rows[0], add_rows[3]
    c.advice[a.index][r2] = c.advice[a.index][r1]
    c.advice[c_.index][r2] = (c.advice[a.index][r2] + c.advice[b.index][r2 + 1]) % R
    c.copy(a, r1, a, r2)
    check_satisfied(c)
    return c


def rsa_sha256_shape(plonk, k=15, seed=7, num_advice=80, num_lookup_advice=16, lookup_bits=12, num_spread=8, spread_bits=8,
                     configure_extra=None):
    """Synthetic witness of the reference circuit's shape. Values are built with numpy on canonical
    integers (object arrays only where products are needed). `configure_extra(cs)` may add further
    sub-circuit configurations after this one (as AadhaarQRVerifierCircuit::configure does) and returns
    the function that assigns their cells."""
    rnd = np.random.RandomState(seed)
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

    def spread(v):
        out = 0
        for bit in range(spread_bits):
            out |= ((v >> bit) & 1) << (2 * bit)
        return out

    for i in range(u):
        c.fixed[t_rng.index][i] = i if i < tr else 0
        c.fixed[t_dense.index][i] = i if i < ts else 0
        c.fixed[t_spread.index][i] = spread(i) if i < ts else 0
    # vertical gates: every 4th row starts a gate (a, b, c, d) with d = a + b*c; values are 64-bit limbs
    ngates = (u - 3) // 4
    for ci, (col, s) in enumerate(zip(gate_cols, sels)):
        vals = [int(v) for v in rnd.randint(0, 1 << 62, size=u, dtype=np.int64)]
        for g in range(ngates):
            r0 = 4 * g
            c.fixed[s.index][r0] = 1
            vals[r0 + 3] = (vals[r0] + vals[r0 + 1] * vals[r0 + 2]) % R
        c.advice[col.index][:u] = vals
    for col in lk_cols:
        c.advice[col.index][:u] = [int(v) for v in rnd.randint(0, tr, size=u)]
    for dcol, scol in zip(sp_dense, sp_spread):
        dv = [int(v) for v in rnd.randint(0, ts, size=u)]
        c.advice[dcol.index][:u] = dv
        c.advice[scol.index][:u] = [spread(v) for v in dv]
    # copy constraints. Every cell takes part in at most one copy (tracked in `used`), so fixing up
    # the copied-to cell and its gate output never disturbs an earlier constraint.
    used = set()

    def free_gate(ci):
        while True:
            g = int(rnd.randint(ngates))
            if all((ci, 4 * g + o) not in used for o in range(4)):
                for o in range(4):
                    used.add((ci, 4 * g + o))
                return 4 * g

    def set_input(ci, r0, off, value):
        col = gate_cols[ci]
        c.advice[col.index][r0 + off] = value % R
        c.advice[col.index][r0 + 3] = (c.advice[col.index][r0] + c.advice[col.index][r0 + 1] * c.advice[col.index][r0 + 2]) % R

    for t in range(max(8, u // 8)):  # advice <-> advice
        i1, i2 = int(rnd.randint(num_advice)), int(rnd.randint(num_advice))
        r1, r2 = free_gate(i1), free_gate(i2)
        set_input(i2, r2, 0, c.advice[gate_cols[i1].index][r1])
        c.copy(gate_cols[i1], r1, gate_cols[i2], r2)
    for t in range(16):  # range-checked cells feeding gates
        lc, gi = lk_cols[t % num_lookup_advice], t % num_advice
        r0 = free_gate(gi)
        set_input(gi, r0, 1, c.advice[lc.index][t])
        c.copy(lc, t, gate_cols[gi], r0 + 1)
    for t in range(8):  # constants
        gi = t % num_advice
        r0 = free_gate(gi)
        c.fixed[konst.index][t] = c.advice[gate_cols[gi].index][r0 + 2]
        c.copy(konst, t, gate_cols[gi], r0 + 2)
    c.instances[0], c.instances[1] = [], []
    for i in range(32):  # public inputs: 32 modulus limbs / 32 hash bytes in the reference (src/lib.rs:389-394)
        gi = i % num_advice
        r0 = free_gate(gi)
        c.instances[0].append(c.advice[gate_cols[gi].index][r0 + 1])
        c.copy(inst[0], i, gate_cols[gi], r0 + 1)
        lc = lk_cols[i % num_lookup_advice]
        c.instances[1].append(c.advice[lc.index][20 + i])
        c.copy(inst[1], i, lc, 20 + i)
    if synthesize_extra:
        synthesize_extra(c)
    return c


def full_aadhaar_shape(plonk, k=15, seed=7, signal=5, **kw):
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

        def synthesize(c):
            c.fixed[s_id.index][0] = 1
            vals = {"reveal_age": 1, "age": 1, "qr_age": 1, "reveal_gender": 1, "gender": 77, "qr_gender": 77,
                    "reveal_pincode": 0, "pincode": 110051, "qr_pincode": 110051, "reveal_state": 1}
            for i, ch in enumerate(b"Delhi"):
                vals["state%d" % i] = vals["qr_state%d" % i] = ch
            for nm, v in vals.items():
                c.advice[idc[nm].index][0] = v
            for col, v in zip(ts, (2019, 3, 8, 5, 30, 0, 1552023000)):
                c.advice[col.index][0] = v
            c.fixed[s_sq.index][0] = 1
            c.advice[sq[0].index][0] = signal % R
            c.advice[sq[1].index][0] = signal * signal % R
            c.instances[sq_inst.index] = []

        return synthesize

    return rsa_sha256_shape(plonk, k=k, seed=seed, configure_extra=configure_extra, **kw)


def check_satisfied(c, rows=None):
    """MockProver-style check of gates, lookups and copy constraints on the usable rows (Python ints)."""
    desc, n, u = c.desc, c.n, c.usable
    inst = [list(v) + [0] * (n - len(v)) for v in c.instances]

    def ev(e, row):
        op = e[0]
        if op == "const":
            return e[1]
        if op == "fixed":
            return c.fixed[e[1]][(row + e[2]) % n]
        if op == "advice":
            return c.advice[e[1]][(row + e[2]) % n]
        if op == "instance":
            return inst[e[1]][(row + e[2]) % n]
        if op == "neg":
            return (-ev(e[1], row)) % R
        if op == "sum":
            return (ev(e[1], row) + ev(e[2], row)) % R
        if op == "product":
            return ev(e[1], row) * ev(e[2], row) % R
        return ev(e[1], row) * e[2] % R

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
                assert c.value(col, row) == c.value(cols[pi], pj), "copy constraint fails at col %d row %d" % (i, row)
    return True
