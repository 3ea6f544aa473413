"""ORACLE — TEST INFRASTRUCTURE ONLY. Pure-Python (big-int) restatement of the protocol layer of
halo2_proofs 0.2.0 @ PSE v2023_01_20 (#c7e42e41, /root/reference/Cargo.lock:469-471) [UP]:

  plonk::keygen::{keygen_vk, keygen_pk}          plonk::prover::create_proof
  plonk::{permutation, lookup, vanishing}::prover plonk::evaluation::Evaluator::evaluate_h
  poly::kzg::multiopen::shplonk::ProverSHPLONK    transcript::{Blake2bWrite, Challenge255}
  rand_chacha::ChaCha20Rng (seed_from_u64) + halo2curves Fr::random / from_bytes_wide

and a verifier that checks a proof's algebra from first principles (vanishing identity at x and the
SHPLONK opening equation, in G1, using the test SRS's known tau instead of a pairing).

PARITY UNPINNED vs the reference: /root/reference never calls create_proof and holds no proof bytes
(SURVEY.md §0.1, §8(c)); the upstream crate cannot be built here. RNG draw order, transcript framing
and point encoding follow SURVEY.md Appendix A and recall of the pinned source; the *mathematics*
(every commitment, evaluation and the final opening) is checked by `verify_proof` independently of
how the prover computed it. Sizes: k <= ~8 (lists of Python ints).

Circuit input = the plain-data description produced by ConstraintSystem.describe() (what a Rust fork
would export) — data, not code, so this file imports nothing from the product.
"""
import hashlib
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import pyref as P  # noqa: E402

R, Q = P.R, P.Q
DELTA = P.DELTA
ADVICE, FIXED, INSTANCE = 0, 1, 2


# ------------------------------------------------------------------------------------------ RNG
class ChaCha20Rng:
    """rand_chacha::ChaCha20Rng::seed_from_u64 (rand_core's PCG32 seed expansion), next_u64 stream."""

    def __init__(self, seed_u64):
        state = seed_u64 & 0xFFFFFFFFFFFFFFFF
        key = b""
        for _ in range(8):
            state = (state * 6364136223846793005 + 11634580027462260723) & 0xFFFFFFFFFFFFFFFF
            xorshifted = (((state >> 18) ^ state) >> 27) & 0xFFFFFFFF
            rot = state >> 59
            x = ((xorshifted >> rot) | (xorshifted << ((32 - rot) & 31))) & 0xFFFFFFFF
            key += x.to_bytes(4, "little")
        self.key = [int.from_bytes(key[4 * i:4 * i + 4], "little") for i in range(8)]
        self.counter = 0
        self.buf = []

    def _block(self):
        c = [0x61707865, 0x3320646E, 0x79622D32, 0x6B206574] + self.key + [self.counter & 0xFFFFFFFF, (self.counter >> 32) & 0xFFFFFFFF, 0, 0]
        x = list(c)

        def qr(a, b, cc, d):
            x[a] = (x[a] + x[b]) & 0xFFFFFFFF; x[d] ^= x[a]; x[d] = ((x[d] << 16) | (x[d] >> 16)) & 0xFFFFFFFF
            x[cc] = (x[cc] + x[d]) & 0xFFFFFFFF; x[b] ^= x[cc]; x[b] = ((x[b] << 12) | (x[b] >> 20)) & 0xFFFFFFFF
            x[a] = (x[a] + x[b]) & 0xFFFFFFFF; x[d] ^= x[a]; x[d] = ((x[d] << 8) | (x[d] >> 24)) & 0xFFFFFFFF
            x[cc] = (x[cc] + x[d]) & 0xFFFFFFFF; x[b] ^= x[cc]; x[b] = ((x[b] << 7) | (x[b] >> 25)) & 0xFFFFFFFF

        for _ in range(10):
            qr(0, 4, 8, 12); qr(1, 5, 9, 13); qr(2, 6, 10, 14); qr(3, 7, 11, 15)
            qr(0, 5, 10, 15); qr(1, 6, 11, 12); qr(2, 7, 8, 13); qr(3, 4, 9, 14)
        self.counter += 1
        return [(x[i] + c[i]) & 0xFFFFFFFF for i in range(16)]

    def next_u64(self):
        if len(self.buf) < 2:
            self.buf += self._block()
        lo, hi = self.buf[0], self.buf[1]
        del self.buf[:2]
        return lo | (hi << 32)

    def fr(self):
        """halo2curves Fr::random: from_u512 of 8 next_u64 (little-endian 512-bit integer mod r)."""
        v = 0
        for i in range(8):
            v |= self.next_u64() << (64 * i)
        return v % R


# ------------------------------------------------------------------------------------ transcript
def fr_repr(x):
    return (x % R).to_bytes(32, "little")


def g1_compress(p):
    """halo2curves 0.3.1 G1Affine::to_bytes [UP recall]: x little-endian, sign(y) = y&1 in bit 7 of byte 31."""
    if p is None:
        return bytes(32)
    b = bytearray(p[0].to_bytes(32, "little"))
    b[31] |= (p[1] & 1) << 7
    return bytes(b)


class Blake2bWrite:
    """transcript::Blake2bWrite<_, G1Affine, Challenge255<_>>."""

    def __init__(self):
        self.state = hashlib.blake2b(digest_size=64, person=b"Halo2-Transcript")
        self.proof = bytearray()

    def squeeze_challenge(self):
        self.state.update(b"\x00")
        return int.from_bytes(self.state.copy().digest(), "little") % R

    def common_point(self, p):
        assert p is not None, "cannot write points at infinity to the transcript"
        self.state.update(b"\x01")
        self.state.update(p[0].to_bytes(32, "little"))
        self.state.update(p[1].to_bytes(32, "little"))

    def common_scalar(self, s):
        self.state.update(b"\x02")
        self.state.update(fr_repr(s))

    def write_point(self, p):
        self.common_point(p)
        self.proof += g1_compress(p)

    def write_scalar(self, s):
        self.common_scalar(s)
        self.proof += fr_repr(s)


class Blake2bRead(Blake2bWrite):
    def __init__(self, proof):
        super().__init__()
        self.data, self.pos = bytes(proof), 0

    def read_point(self):
        b = bytearray(self.data[self.pos:self.pos + 32])
        self.pos += 32
        sign = b[31] >> 7
        b[31] &= 0x7F
        x = int.from_bytes(b, "little")
        assert x < Q
        y = pow((x * x * x + 3) % Q, (Q + 1) // 4, Q)
        assert y * y % Q == (x * x * x + 3) % Q, "point not on curve"
        if (y & 1) != sign:
            y = Q - y
        p = (x, y)
        self.common_point(p)
        return p

    def read_scalar(self):
        s = int.from_bytes(self.data[self.pos:self.pos + 32], "little")
        self.pos += 32
        assert s < R
        self.common_scalar(s)
        return s


def keccak256(data):
    """Keccak-256 as the EVM computes it (pad byte 0x01, not SHA3's 0x06); pure Python."""
    RC = [0x0000000000000001, 0x0000000000008082, 0x800000000000808A, 0x8000000080008000, 0x000000000000808B, 0x0000000080000001,
          0x8000000080008081, 0x8000000000008009, 0x000000000000008A, 0x0000000000000088, 0x0000000080008009, 0x000000008000000A,
          0x000000008000808B, 0x800000000000008B, 0x8000000000008089, 0x8000000000008003, 0x8000000000008002, 0x8000000000000080,
          0x000000000000800A, 0x800000008000000A, 0x8000000080008081, 0x8000000000008080, 0x0000000080000001, 0x8000000080008008]
    M = (1 << 64) - 1
    rol = lambda v, n: ((v << n) | (v >> (64 - n))) & M if n else v
    st = [[0] * 5 for _ in range(5)]  # st[x][y]

    def f():
        for rc in RC:
            c = [st[x][0] ^ st[x][1] ^ st[x][2] ^ st[x][3] ^ st[x][4] for x in range(5)]
            d = [c[(x - 1) % 5] ^ rol(c[(x + 1) % 5], 1) for x in range(5)]
            for x in range(5):
                for y in range(5):
                    st[x][y] ^= d[x]
            x, y, cur = 1, 0, st[1][0]
            for t in range(24):  # rho + pi
                x, y = y, (2 * x + 3 * y) % 5
                cur, st[x][y] = st[x][y], rol(cur, ((t + 1) * (t + 2) // 2) % 64)
            for y in range(5):  # chi
                row = [st[x][y] for x in range(5)]
                for x in range(5):
                    st[x][y] = row[x] ^ ((~row[(x + 1) % 5]) & M & row[(x + 2) % 5])
            st[0][0] ^= rc

    rate = 136
    data = bytearray(data)
    data.append(0x01)
    while len(data) % rate:
        data.append(0)
    data[-1] |= 0x80
    for off in range(0, len(data), rate):
        for i in range(rate // 8):
            st[i % 5][i // 5] ^= int.from_bytes(data[off + 8 * i:off + 8 * i + 8], "little")
        f()
    return b"".join(st[i % 5][i // 5].to_bytes(8, "little") for i in range(4))


class Keccak256Write:
    """halo2-solidity-verifier Keccak256Transcript + ChallengeEvm [UP], i.e. what
    /root/reference/solidity_verifier_contract/contract.sol:77-112 re-derives."""

    def __init__(self):
        self.buf = bytearray()
        self.proof = bytearray()

    def squeeze_challenge(self):
        data = bytes(self.buf) + (b"\x01" if len(self.buf) == 32 else b"")
        h = keccak256(data)
        self.buf = bytearray(h)
        return int.from_bytes(h, "big") % R

    def common_point(self, p):
        assert p is not None
        self.buf += p[0].to_bytes(32, "big") + p[1].to_bytes(32, "big")

    def common_scalar(self, s):
        self.buf += (s % R).to_bytes(32, "big")

    def write_point(self, p):
        self.common_point(p)
        self.proof += p[0].to_bytes(32, "big") + p[1].to_bytes(32, "big")

    def write_scalar(self, s):
        self.common_scalar(s)
        self.proof += (s % R).to_bytes(32, "big")


class Keccak256Read(Keccak256Write):
    def __init__(self, proof):
        super().__init__()
        self.data, self.pos = bytes(proof), 0

    def read_point(self):
        x = int.from_bytes(self.data[self.pos:self.pos + 32], "big")
        y = int.from_bytes(self.data[self.pos + 32:self.pos + 64], "big")
        self.pos += 64
        assert x < Q and y < Q and (y * y - x * x * x - 3) % Q == 0, "point not on curve"
        self.common_point((x, y))
        return (x, y)

    def read_scalar(self):
        s = int.from_bytes(self.data[self.pos:self.pos + 32], "big")
        self.pos += 32
        assert s < R
        self.common_scalar(s)
        return s


# ------------------------------------------------------------------------------------ helpers
def batch_invert(v):
    return [pow(x, -1, R) if x else 0 for x in v]


def commit_tau(poly_coeff, tau):
    """Commitment with a test SRS g[i] = tau^i G: MSM(poly, g) = poly(tau) * G (same group element)."""
    return P.g1_mul(P.G1_GEN, P.eval_polynomial(poly_coeff, tau))


def evaluate_expr(e, fixed, advice, instance):
    """fixed/advice/instance: callables (column, rotation) -> value."""
    op = e[0]
    if op == "const":
        return e[1] % R
    if op == "fixed":
        return fixed(e[1], e[2])
    if op == "advice":
        return advice(e[1], e[2])
    if op == "instance":
        return instance(e[1], e[2])
    if op == "neg":
        return (-evaluate_expr(e[1], fixed, advice, instance)) % R
    if op == "sum":
        return (evaluate_expr(e[1], fixed, advice, instance) + evaluate_expr(e[2], fixed, advice, instance)) % R
    if op == "product":
        return evaluate_expr(e[1], fixed, advice, instance) * evaluate_expr(e[2], fixed, advice, instance) % R
    if op == "scaled":
        return evaluate_expr(e[1], fixed, advice, instance) * e[2] % R
    raise ValueError(op)


class Domain(P.EvaluationDomain):
    def rotate_omega(self, x, rot):
        return x * pow(self.omega, rot, R) % R if rot >= 0 else x * pow(self.omega_inv, -rot, R) % R

    def coset_point(self, idx):
        return P.ZETA * pow(self.extended_omega, idx, R) % R


# ------------------------------------------------------------------------------------ keygen
class ProvingKey:
    pass


def keygen(desc, fixed_values, mapping, tau, transcript_repr=None):
    """keygen_vk + keygen_pk. fixed_values[c] = n ints (Lagrange); mapping = Assembly.mapping over
    desc['permutation_columns']. The SRS is the test SRS of known tau."""
    pk = ProvingKey()
    k = desc["k"]
    n = 1 << k
    pk.desc, pk.k, pk.n, pk.tau = desc, k, n, tau
    pk.domain = d = Domain(desc["cs_degree"], k)
    bf = desc["blinding_factors"]
    assert n >= bf + 3, "not enough rows"
    pk.fixed_values = [list(c) for c in fixed_values]
    pk.fixed_polys = [d.lagrange_to_coeff(c) for c in pk.fixed_values]
    pk.fixed_commitments = [commit_tau(p, tau) for p in pk.fixed_polys]
    ncols = len(desc["permutation_columns"])
    omega_powers = [pow(d.omega, i, R) for i in range(n)]
    pk.permutations = []  # sigma_i in Lagrange form
    for i in range(ncols):
        col = []
        for j in range(n):
            pi, pj = mapping[i][j]
            col.append(pow(DELTA, pi, R) * omega_powers[pj] % R)
        pk.permutations.append(col)
    pk.permutation_polys = [d.lagrange_to_coeff(c) for c in pk.permutations]
    pk.permutation_commitments = [commit_tau(p, tau) for p in pk.permutation_polys]
    l0 = [0] * n
    l0[0] = 1
    l_last = [0] * n
    l_last[n - bf - 1] = 1
    l_blind = [0] * n
    for i in range(n - bf, n):
        l_blind[i] = 1
    pk.l0_coeff, pk.l_last_coeff, pk.l_blind_coeff = (d.lagrange_to_coeff(v) for v in (l0, l_last, l_blind))
    # transcript_repr: upstream hashes the Debug string of the pinned VK — not reproducible outside
    # Rust; a fork passes its own value in. Default: hash of our description's repr.
    if transcript_repr is None:
        h = hashlib.blake2b(digest_size=64, person=b"Halo2-Verify-Key")
        s = repr((desc, [P.g1_on_curve(c) and c for c in pk.fixed_commitments], pk.permutation_commitments)).encode()
        h.update(len(s).to_bytes(8, "little"))
        h.update(s)
        transcript_repr = int.from_bytes(h.digest(), "little") % R
    pk.transcript_repr = transcript_repr
    return pk


# ------------------------------------------------------------------------------------ prover
def permute_expression_pair(inp, tab, usable, bf, rng):
    """lookup::prover::permute_expression_pair [UP]."""
    a = sorted(inp[:usable])
    leftover = {}
    for v in tab[:usable]:
        leftover[v] = leftover.get(v, 0) + 1
    s = [0] * usable
    repeated = []
    for row, v in enumerate(a):
        if row == 0 or v != a[row - 1]:
            s[row] = v
            assert leftover.get(v, 0) > 0, "lookup input not in table (ConstraintSystemFailure)"
            leftover[v] -= 1
        else:
            repeated.append(row)
    for v in sorted(leftover):  # BTreeMap iteration: ascending
        for _ in range(leftover[v]):
            s[repeated.pop()] = v
    assert not repeated
    a += [rng.fr() for _ in range(bf + 1)]
    s += [rng.fr() for _ in range(bf + 1)]
    return a, s


def h_constraints(desc, d, get, beta, gamma, theta, x_term, nsets, nlookups):
    """The ordered list of constraint values at one point, as evaluate_h / the verifier's
    expressions() fold them with y. `get(kind, index, rot)` returns the value of a polynomial:
    kinds: fixed/advice/instance, sigma(i), z(set), lz(lookup), la(lookup), ls(lookup), l0,l_last,l_active.
    x_term = the evaluation point X itself."""
    out = []
    F = lambda c, r: get("fixed", c, r)
    A = lambda c, r: get("advice", c, r)
    I = lambda c, r: get("instance", c, r)
    for g in desc["gates"]:
        out.append(evaluate_expr(g, F, A, I))
    bf = desc["blinding_factors"]
    last_rot = -(bf + 1)
    l0, l_last, l_active = get("l0", 0, 0), get("l_last", 0, 0), get("l_active", 0, 0)
    cols = desc["permutation_columns"]
    chunk = desc["cs_degree"] - 2
    colval = lambda kc: get(("advice", "fixed", "instance")[kc[0]], kc[1], 0)
    if nsets:
        out.append((1 - get("z", 0, 0)) * l0 % R)
        zl = get("z", nsets - 1, 0)
        out.append((zl * zl - zl) * l_last % R)
        for s in range(1, nsets):
            out.append((get("z", s, 0) - get("z", s - 1, last_rot)) * l0 % R)
        current_delta = beta * x_term % R
        for s in range(nsets):
            cset = cols[s * chunk:(s + 1) * chunk]
            left = get("z", s, 1)
            for j, kc in enumerate(cset):
                left = left * (colval(kc) + beta * get("sigma", s * chunk + j, 0) + gamma) % R
            right = get("z", s, 0)
            for kc in cset:
                right = right * (colval(kc) + current_delta + gamma) % R
                current_delta = current_delta * DELTA % R
            out.append((left - right) * l_active % R)
    for li in range(nlookups):
        lk = desc["lookups"][li]
        ci = 0
        for e in lk["inputs"]:
            ci = (ci * theta + evaluate_expr(e, F, A, I)) % R
        ct = 0
        for e in lk["tables"]:
            ct = (ct * theta + evaluate_expr(e, F, A, I)) % R
        z, znext = get("lz", li, 0), get("lz", li, 1)
        a, aprev, s = get("la", li, 0), get("la", li, -1), get("ls", li, 0)
        out.append((1 - z) * l0 % R)
        out.append((z * z - z) * l_last % R)
        out.append((znext * (a + beta) % R * (s + gamma) - z * ((ci + beta) * (ct + gamma) % R)) * l_active % R)
        out.append((a - s) * l0 % R)
        out.append((a - s) * (a - aprev) % R * l_active % R)
    return out


def poly_eval_fn(polys_by_kind, d):
    def get_at(x):
        def get(kind, idx, rot):
            if kind in ("l0", "l_last", "l_active"):
                return P.eval_polynomial(polys_by_kind[kind], x)
            return P.eval_polynomial(polys_by_kind[kind][idx], d.rotate_omega(x, rot))
        return get
    return get_at


def create_proof(pk, instances, advice_values, seed, trace=None, transcript="blake2b", multiopen="shplonk"):
    """plonk::prover::create_proof for one circuit instance, KZG + SHPLONK + Blake2b, phase 0 only.
    instances[c] = list of public inputs of instance column c; advice_values[c] = n ints (rows past
    the usable range are overwritten by blinding). Returns the proof bytes."""
    return create_proof_multi(pk, [instances], [advice_values], seed, trace=trace, transcript=transcript, multiopen=multiopen)


def create_proof_multi(pk, instances_list, advice_list, seed, trace=None, transcript="blake2b", multiopen="shplonk"):
    """plonk::prover::create_proof(params, pk, &[circuit; N], &[instances; N], rng, transcript) [UP] — upstream's
    slices: N instances of the SAME circuit in one proof, phase 0 only. Every per-circuit step loops over the
    circuits in order (instances absorbed, advice blinded / committed, lookups permuted, permutation products,
    lookup products), the challenges, the random polynomial and h(X) are shared — h folds the circuits' terms
    in one Horner chain with y, circuit after circuit —, then the evaluations (advice per circuit, fixed,
    random, sigma, then per circuit the permutation products and per circuit the lookups) and one multiopen
    over all queries (per circuit: advice, permutation products, lookups; then fixed, sigma, h, random).
    With N = 1 this is create_proof, statement for statement."""
    desc, d, n, tau = pk.desc, pk.domain, pk.n, pk.tau
    bf = desc["blinding_factors"]
    usable = n - (bf + 1)
    N = len(instances_list)
    assert N >= 1 and len(advice_list) == N
    rng = ChaCha20Rng(seed)
    T = Blake2bWrite() if transcript == "blake2b" else Keccak256Write()
    tr = (lambda *a: trace.append(a)) if trace is not None else (lambda *a: None)
    T.common_scalar(pk.transcript_repr)
    C = [dict() for _ in range(N)]  # per-circuit state
    # instances
    for ci_, instances in enumerate(instances_list):
        inst_values = []
        for col in instances:
            assert len(col) <= usable
            for v in col:
                T.common_scalar(v)
            inst_values.append(list(col) + [0] * (n - len(col)))
        C[ci_]["inst_values"] = inst_values
        C[ci_]["inst_polys"] = [d.lagrange_to_coeff(v) for v in inst_values]
    # advice: per circuit, blind all columns, then draw (unused) commitment blinds, then commit
    for ci_, advice_values in enumerate(advice_list):
        adv = [list(c) for c in advice_values]
        assert len(adv) == desc["num_advice"] and all(len(c) == n for c in adv)
        for c in adv:
            for row in range(usable, n):
                c[row] = rng.fr()
        for _ in adv:
            rng.fr()
        adv_polys = [d.lagrange_to_coeff(c) for c in adv]
        for p in adv_polys:
            cm = commit_tau(p, tau)
            tr("advice_commit", cm)
            T.write_point(cm)
        C[ci_]["adv"], C[ci_]["adv_polys"] = adv, adv_polys
    theta = T.squeeze_challenge()
    tr("theta", theta)
    # lookups: permuted columns
    for st in C:
        adv, inst_values = st["adv"], st["inst_values"]
        lookups = []
        for lk in desc["lookups"]:
            def compress(exprs):
                acc = [0] * n
                for e in exprs:
                    for row in range(n):
                        v = evaluate_expr(e, lambda c, r: pk.fixed_values[c][(row + r) % n], lambda c, r: adv[c][(row + r) % n],
                                          lambda c, r: inst_values[c][(row + r) % n])
                        acc[row] = (acc[row] * theta + v) % R
                return acc
            ci, ct = compress(lk["inputs"]), compress(lk["tables"])
            a, s = permute_expression_pair(ci, ct, usable, bf, rng)
            pa = d.lagrange_to_coeff(a)
            rng.fr()
            ca = commit_tau(pa, tau)
            ps = d.lagrange_to_coeff(s)
            rng.fr()
            cs_ = commit_tau(ps, tau)
            T.write_point(ca)
            T.write_point(cs_)
            tr("lookup_permuted", ca, cs_)
            lookups.append({"ci": ci, "ct": ct, "a": a, "s": s, "pa": pa, "ps": ps})
        st["lookups"] = lookups
    beta = T.squeeze_challenge()
    gamma = T.squeeze_challenge()
    tr("beta_gamma", beta, gamma)
    # permutation grand products
    cols = desc["permutation_columns"]
    chunk = desc["cs_degree"] - 2
    for st in C:
        adv, inst_values = st["adv"], st["inst_values"]
        colvals = lambda kc: (adv, pk.fixed_values, inst_values)[kc[0]][kc[1]]
        sets = []
        deltaomega_col = 1  # delta^j for the running column j
        last_z = 1
        for s0 in range(0, len(cols), chunk):
            cset = cols[s0:s0 + chunk]
            mod = [1] * n
            for j, kc in enumerate(cset):
                vals, sig = colvals(kc), pk.permutations[s0 + j]
                for row in range(n):
                    mod[row] = mod[row] * (beta * sig[row] + gamma + vals[row]) % R
            mod = batch_invert(mod)
            for kc in cset:
                vals = colvals(kc)
                dw = deltaomega_col
                for row in range(n):
                    mod[row] = mod[row] * (dw * beta + gamma + vals[row]) % R
                    dw = dw * d.omega % R
                deltaomega_col = deltaomega_col * DELTA % R
            z = [last_z]
            for row in range(1, n):
                z.append(z[row - 1] * mod[row - 1] % R)
            for row in range(n - bf, n):
                z[row] = rng.fr()
            last_z = z[n - (bf + 1)]
            rng.fr()
            pz = d.lagrange_to_coeff(z)
            cz = commit_tau(pz, tau)
            T.write_point(cz)
            tr("perm_z", cz)
            sets.append(pz)
        st["sets"] = sets
    # lookup grand products
    for st in C:
        for lk in st["lookups"]:
            den = batch_invert([(beta + lk["a"][i]) * (gamma + lk["s"][i]) % R for i in range(n)])
            prod = [den[i] * (lk["ci"][i] + beta) % R * (lk["ct"][i] + gamma) % R for i in range(n)]
            z = [1]
            for row in range(1, n - bf):
                z.append(z[row - 1] * prod[row - 1] % R)
            z += [rng.fr() for _ in range(bf)]
            rng.fr()
            lk["pz"] = d.lagrange_to_coeff(z)
            cz = commit_tau(lk["pz"], tau)
            T.write_point(cz)
            tr("lookup_z", cz)
    # vanishing: random polynomial
    random_poly = [rng.fr() for _ in range(n)]
    rng.fr()
    crand = commit_tau(random_poly, tau)
    T.write_point(crand)
    tr("random_commit", crand)
    y = T.squeeze_challenge()
    tr("y", y)
    # quotient: evaluate every constraint of every circuit on the extended coset, fold with y, divide by Z_H
    ext = d.extended_len()
    rot_scale = 1 << (d.extended_k - d.k)
    to_ext = d.coeff_to_extended
    fixed_cos, sigma_cos = [to_ext(p) for p in pk.fixed_polys], [to_ext(p) for p in pk.permutation_polys]
    for st in C:
        st["cos"] = {"fixed": fixed_cos, "advice": [to_ext(p) for p in st["adv_polys"]],
                     "instance": [to_ext(p) for p in st["inst_polys"]], "sigma": sigma_cos,
                     "z": [to_ext(p) for p in st["sets"]], "lz": [to_ext(l["pz"]) for l in st["lookups"]],
                     "la": [to_ext(l["pa"]) for l in st["lookups"]], "ls": [to_ext(l["ps"]) for l in st["lookups"]]}
    l0e, lle, lbe = to_ext(pk.l0_coeff), to_ext(pk.l_last_coeff), to_ext(pk.l_blind_coeff)
    h = []
    for idx in range(ext):
        acc = 0
        for st in C:
            cos = st["cos"]

            def get(kind, i, rot, idx=idx, cos=cos):
                if kind == "l0":
                    return l0e[idx]
                if kind == "l_last":
                    return lle[idx]
                if kind == "l_active":
                    return (1 - (lle[idx] + lbe[idx])) % R
                return cos[kind][i][(idx + rot * rot_scale) % ext]
            vals = h_constraints(desc, d, get, beta, gamma, theta, d.coset_point(idx), len(st["sets"]), len(st["lookups"]))
            for v in vals:
                acc = (acc * y + v) % R
        h.append(acc)
    h = d.extended_to_coeff(d.divide_by_vanishing_poly(h))
    pieces = [h[i * n:(i + 1) * n] for i in range(d.quotient_poly_degree)]
    for _ in pieces:
        rng.fr()
    for pc in pieces:
        c = commit_tau(pc, tau)
        T.write_point(c)
        tr("h_piece", c)
    x = T.squeeze_challenge()
    tr("x", x)
    xn = pow(x, n, R)
    # evaluations
    ev = lambda poly, rot: P.eval_polynomial(poly, d.rotate_omega(x, rot))
    for st in C:
        for c, r in desc["advice_queries"]:
            T.write_scalar(ev(st["adv_polys"][c], r))
    for c, r in desc["fixed_queries"]:
        T.write_scalar(ev(pk.fixed_polys[c], r))
    h_poly = [0] * n
    for pc in reversed(pieces):
        h_poly = [(a * xn + b) % R for a, b in zip(h_poly, pc)]
    T.write_scalar(ev(random_poly, 0))
    for p in pk.permutation_polys:
        T.write_scalar(ev(p, 0))
    for st in C:
        sets = st["sets"]
        for si, pz in enumerate(sets):
            T.write_scalar(ev(pz, 0))
            T.write_scalar(ev(pz, 1))
            if si + 1 < len(sets):
                T.write_scalar(ev(pz, -(bf + 1)))
    for st in C:
        for lk in st["lookups"]:
            for poly, rot in ((lk["pz"], 0), (lk["pz"], 1), (lk["pa"], 0), (lk["pa"], -1), (lk["ps"], 0)):
                T.write_scalar(ev(poly, rot))
    # multiopen queries, upstream order. poly identity = python object id
    queries = []
    Qy = lambda poly, rot: queries.append((poly, d.rotate_omega(x, rot)))
    for st in C:
        sets = st["sets"]
        for c, r in desc["advice_queries"]:
            Qy(st["adv_polys"][c], r)
        for pz in sets:
            Qy(pz, 0)
            Qy(pz, 1)
        for pz in reversed(sets[:-1]):
            Qy(pz, -(bf + 1))
        for lk in st["lookups"]:
            Qy(lk["pz"], 0); Qy(lk["pa"], 0); Qy(lk["ps"], 0); Qy(lk["pa"], -1); Qy(lk["pz"], 1)
    for c, r in desc["fixed_queries"]:
        Qy(pk.fixed_polys[c], r)
    for p in pk.permutation_polys:
        Qy(p, 0)
    Qy(h_poly, 0)
    Qy(random_poly, 0)
    if multiopen == "gwc":
        gwc_prove(queries, T, tau, n, tr)
    else:
        shplonk_prove(queries, T, tau, n, tr)
    return bytes(T.proof)


def lagrange_interpolate(points, evals):
    """arithmetic::lagrange_interpolate: coefficients of the unique poly of degree < len(points)."""
    m = len(points)
    out = [0] * m
    for j in range(m):
        num = [1]
        den = 1
        for k2 in range(m):
            if k2 == j:
                continue
            num = [0] + num
            for t in range(len(num) - 1):
                num[t] = (num[t] - points[k2] * num[t + 1]) % R
            den = den * (points[j] - points[k2]) % R
        sc = evals[j] * pow(den, -1, R) % R
        for t in range(len(num)):
            out[t] = (out[t] + num[t] * sc) % R
    return out


def intermediate_sets(queries):
    """multiopen::shplonk::construct_intermediate_sets [UP]: queries = [(commitment_key, point)]."""
    super_points = sorted({pt for _, pt in queries})
    com_rot = []  # [(key, set(points))] in first-seen order
    for key, pt in queries:
        for ent in com_rot:
            if ent[0] is key or ent[0] == key:
                ent[1].add(pt)
                break
        else:
            com_rot.append((key, {pt}))
    rot_com = []  # [(sorted points tuple, [keys])]
    for key, pts in com_rot:
        t = tuple(sorted(pts))
        for ent in rot_com:
            if ent[0] == t:
                ent[1].append(key)
                break
        else:
            rot_com.append((t, [key]))
    return rot_com, super_points


def shplonk_prove(queries, T, tau, n, tr):
    # key polynomials by identity
    ids = {}
    keyed = []
    for poly, pt in queries:
        ids.setdefault(id(poly), poly)
        keyed.append((id(poly), pt))
    rot_com, super_points = intermediate_sets(keyed)
    y = T.squeeze_challenge()
    v = T.squeeze_challenge()
    tr("shplonk_y_v", y, v)
    sets = []
    for pts, keys in rot_com:
        coms = []
        for key in keys:
            poly = ids[key]
            evals = [P.eval_polynomial(poly, p) for p in pts]
            coms.append((poly, lagrange_interpolate(list(pts), evals)))
        sets.append((list(pts), coms))
    hx = [0] * n
    vp = 1
    for pts, coms in sets:
        nx = [0] * n
        yp = 1
        for poly, low in coms:
            for i in range(n):
                nx[i] = (nx[i] + yp * (poly[i] - (low[i] if i < len(low) else 0))) % R
            yp = yp * y % R
        for p in pts:
            nx = P.kate_division(nx, p)
        nx += [0] * (n - len(nx))
        hx = [(a + vp * b) % R for a, b in zip(hx, nx)]
        vp = vp * v % R
    c1 = commit_tau(hx, tau)
    T.write_point(c1)
    tr("shplonk_h1", c1)
    u = T.squeeze_challenge()
    tr("u", u)
    zt = 1
    for p in super_points:
        zt = zt * (u - p) % R
    lx = [0] * n
    vp = 1
    z0 = None
    for pts, coms in sets:
        zi = 1
        for p in super_points:
            if p not in pts:
                zi = zi * (u - p) % R
        if z0 is None:
            z0 = zi
        li = [0] * n
        yp = 1
        for poly, low in coms:
            r_u = P.eval_polynomial(low, u)
            for i in range(n):
                li[i] = (li[i] + yp * poly[i]) % R
            li[0] = (li[0] - yp * r_u) % R
            yp = yp * y % R
        lx = [(a + vp * zi % R * b) % R for a, b in zip(lx, li)]
        vp = vp * v % R
    lx = [(a - zt * b) % R for a, b in zip(lx, hx)]
    assert P.eval_polynomial(lx, u) == 0
    h2 = P.kate_division(lx, u)
    z0inv = pow(z0, -1, R)
    h2 = [c * z0inv % R for c in h2]
    c2 = commit_tau(h2, tau)
    T.write_point(c2)
    tr("shplonk_h2", c2)


def gwc_point_sets(queries):
    """multiopen::gwc::construct_intermediate_sets [UP]: group the queries by evaluation point, points in
    first-seen order, the queries of one point in their original order. queries = [(item, point)]."""
    sets = []
    for item, pt in queries:
        for s in sets:
            if s[0] == pt:
                s[1].append(item)
                break
        else:
            sets.append((pt, [item]))
    return sets


def gwc_prove(queries, T, tau, n, tr):
    """multiopen::gwc::ProverGWC::create_proof [UP]: v <- transcript; per point z: the queries' polynomials
    and evaluations are combined with 1, v, v^2, ...; W_z = (sum v^j p_j - sum v^j p_j(z)) / (X - z) is
    committed and written."""
    v = T.squeeze_challenge()
    tr("gwc_v", v)
    for z, polys in gwc_point_sets(queries):
        acc = [0] * n
        ev = 0
        vp = 1
        for poly in polys:
            for i in range(n):
                acc[i] = (acc[i] + vp * poly[i]) % R
            ev = (ev + vp * P.eval_polynomial(poly, z)) % R
            vp = vp * v % R
        acc[0] = (acc[0] - ev) % R
        w = P.kate_division(acc, z)
        c = commit_tau(w + [0] * (n - len(w)), tau)
        T.write_point(c)
        tr("gwc_w", c)


# ------------------------------------------------------------------------------------ verifier
class VerifyingKey:
    """What a verifier holds: the constraint-system description, the commitments of the fixed and
    permutation polynomials, the transcript representative — and, standing in for the pairing, the
    test SRS's tau."""

    def __init__(self, desc, fixed_commitments, permutation_commitments, tau, transcript_repr):
        self.desc, self.k, self.n, self.tau = desc, desc["k"], 1 << desc["k"], tau
        self.domain = Domain(desc["cs_degree"], desc["k"])
        self.fixed_commitments, self.permutation_commitments = list(fixed_commitments), list(permutation_commitments)
        self.transcript_repr = transcript_repr


def lagrange_basis_at(d, n, rows, x):
    """l_i(x) = omega^i (x^n - 1) / (n (x - omega^i)) for i in rows (closed form; upstream: l_i_range)."""
    xn1 = (pow(x, n, R) - 1) % R
    ninv = pow(n, -1, R)
    out = {}
    for i in rows:
        wi = pow(d.omega, i, R)
        out[i] = wi * xn1 % R * ninv % R * pow((x - wi) % R, -1, R) % R
    return out


def verify_proof(pk, instances, proof, transcript="blake2b", multiopen="shplonk"):
    """Checks: transcript re-derivation, the vanishing identity at x, and the SHPLONK opening equation
    (in G1, with the known tau standing in for the pairing). Raises AssertionError on failure."""
    return verify_proof_multi(pk, [instances], proof, transcript=transcript, multiopen=multiopen)


def verify_proof_multi(pk, instances_list, proof, transcript="blake2b", multiopen="shplonk"):
    """plonk::verifier::verify_proof over N instances of the circuit in one proof (the reader of
    create_proof_multi): the same checks, every per-circuit read in a loop over the circuits."""
    desc, d, n, tau = pk.desc, pk.domain, pk.n, pk.tau
    bf = desc["blinding_factors"]
    N = len(instances_list)
    T = Blake2bRead(proof) if transcript == "blake2b" else Keccak256Read(proof)
    T.common_scalar(pk.transcript_repr)
    for instances in instances_list:
        for col in instances:
            for v in col:
                T.common_scalar(v)
    adv_c = [[T.read_point() for _ in range(desc["num_advice"])] for _ in range(N)]
    theta = T.squeeze_challenge()
    lk_c = [[(T.read_point(), T.read_point()) for _ in desc["lookups"]] for _ in range(N)]
    beta = T.squeeze_challenge()
    gamma = T.squeeze_challenge()
    chunk = desc["cs_degree"] - 2
    ncols = len(desc["permutation_columns"])
    nsets = (ncols + chunk - 1) // chunk
    z_c = [[T.read_point() for _ in range(nsets)] for _ in range(N)]
    lz_c = [[T.read_point() for _ in desc["lookups"]] for _ in range(N)]
    rand_c = T.read_point()
    y = T.squeeze_challenge()
    h_c = [T.read_point() for _ in range(d.quotient_poly_degree)]
    x = T.squeeze_challenge()
    xn = pow(x, n, R)
    adv_e = [[T.read_scalar() for _ in desc["advice_queries"]] for _ in range(N)]
    fix_e = [T.read_scalar() for _ in desc["fixed_queries"]]
    rand_e = T.read_scalar()
    sig_e = [T.read_scalar() for _ in range(ncols)]
    z_e = []
    for _ in range(N):
        ze = []
        for s in range(nsets):
            e = {0: T.read_scalar(), 1: T.read_scalar()}
            if s + 1 < nsets:
                e[-(bf + 1)] = T.read_scalar()
            ze.append(e)
        z_e.append(ze)
    lk_e = [[{("lz", 0): T.read_scalar(), ("lz", 1): T.read_scalar(), ("la", 0): T.read_scalar(), ("la", -1): T.read_scalar(),
              ("ls", 0): T.read_scalar()} for _ in desc["lookups"]] for _ in range(N)]
    # l_0, l_last, l_blind at x in closed form; instance evaluations from the public inputs
    # (QUERY_INSTANCE = false): sum_i inst[i] * l_i(x * omega^rot)
    lb = lagrange_basis_at(d, n, [0, n - bf - 1] + list(range(n - bf, n)), x)
    l0, l_last = lb[0], lb[n - bf - 1]
    l_blind = sum(lb[i] for i in range(n - bf, n)) % R
    fq = {q: e for q, e in zip([tuple(q) for q in desc["fixed_queries"]], fix_e)}
    acc = 0
    for ci_ in range(N):
        instances = instances_list[ci_]

        def inst_eval(i, rot, instances=instances):
            col = instances[i]
            if not col:
                return 0
            basis = lagrange_basis_at(d, n, range(len(col)), d.rotate_omega(x, rot))
            return sum(v * basis[j] for j, v in enumerate(col)) % R
        aq = {q: e for q, e in zip([tuple(q) for q in desc["advice_queries"]], adv_e[ci_])}

        def get(kind, i, rot, aq=aq, inst_eval=inst_eval, ci_=ci_):
            if kind == "advice":
                return aq[(i, rot)]
            if kind == "fixed":
                return fq[(i, rot)]
            if kind == "instance":
                return inst_eval(i, rot)
            if kind == "sigma":
                return sig_e[i]
            if kind == "z":
                return z_e[ci_][i][rot]
            if kind in ("lz", "la", "ls"):
                return lk_e[ci_][i][(kind, rot)]
            return {"l0": l0, "l_last": l_last, "l_active": (1 - (l_last + l_blind)) % R}[kind]

        vals = h_constraints(desc, d, get, beta, gamma, theta, x, nsets, len(desc["lookups"]))
        for vv in vals:
            acc = (acc * y + vv) % R
    expected_h = acc * pow(xn - 1, -1, R) % R
    h_commit = None
    for c in reversed(h_c):
        h_commit = P.g1_add(P.g1_mul(h_commit, xn) if h_commit else None, c)
    # queries in the prover's order: (commitment key, point, eval)
    queries = []
    Qv = lambda key, com, rot, e: queries.append((key, com, d.rotate_omega(x, rot), e))
    for ci_ in range(N):
        for (c, r), e in zip(desc["advice_queries"], adv_e[ci_]):
            Qv(("a", ci_, c), adv_c[ci_][c], r, e)
        for s in range(nsets):
            Qv(("z", ci_, s), z_c[ci_][s], 0, z_e[ci_][s][0])
            Qv(("z", ci_, s), z_c[ci_][s], 1, z_e[ci_][s][1])
        for s in reversed(range(nsets - 1)):
            Qv(("z", ci_, s), z_c[ci_][s], -(bf + 1), z_e[ci_][s][-(bf + 1)])
        for li in range(len(desc["lookups"])):
            e = lk_e[ci_][li]
            Qv(("lz", ci_, li), lz_c[ci_][li], 0, e[("lz", 0)]); Qv(("la", ci_, li), lk_c[ci_][li][0], 0, e[("la", 0)])
            Qv(("ls", ci_, li), lk_c[ci_][li][1], 0, e[("ls", 0)]); Qv(("la", ci_, li), lk_c[ci_][li][0], -1, e[("la", -1)])
            Qv(("lz", ci_, li), lz_c[ci_][li], 1, e[("lz", 1)])
    for (c, r), e in zip(desc["fixed_queries"], fix_e):
        Qv(("f", c), pk.fixed_commitments[c], r, e)
    for i in range(ncols):
        Qv(("s", i), pk.permutation_commitments[i], 0, sig_e[i])
    Qv(("h",), h_commit, 0, expected_h)
    Qv(("r",), rand_c, 0, rand_e)
    if multiopen == "gwc":
        # multiopen::gwc::VerifierGWC [UP]: e(sum u^i W_i, [tau]_2) = e(sum u^i z_i W_i + C - e*G, [1]_2), checked
        # in G1 with the known tau. C / e fold each point's commitments / evaluations with powers of v.
        v = T.squeeze_challenge()
        sets = gwc_point_sets([((com, e), pt) for _, com, pt, e in queries])
        ws = [T.read_point() for _ in sets]
        u = T.squeeze_challenge()
        assert T.pos == len(T.data), "trailing bytes in proof"
        lhs = rhs = None
        up = 1
        for (z, items), w in zip(sets, ws):
            cb, eb, vp = None, 0, 1
            for com, e in items:
                cb = P.g1_add(cb, P.g1_mul(com, vp))
                eb = (eb + vp * e) % R
                vp = vp * v % R
            term = P.g1_add(cb, P.g1_neg(P.g1_mul(P.G1_GEN, eb)))
            rhs = P.g1_add(rhs, P.g1_mul(P.g1_add(term, P.g1_mul(w, z)), up))
            lhs = P.g1_add(lhs, P.g1_mul(w, up * tau % R))
            up = up * u % R
        assert lhs == rhs, "GWC opening equation does not hold"
        return True
    # SHPLONK
    coms = {k_: c for k_, c, _, _ in queries}
    evmap = {(k_, pt): e for k_, _, pt, e in queries}
    rot_com, super_points = intermediate_sets([(k_, pt) for k_, _, pt, _ in queries])
    yy = T.squeeze_challenge()
    v = T.squeeze_challenge()
    h1 = T.read_point()
    u = T.squeeze_challenge()
    h2 = T.read_point()
    assert T.pos == len(T.data), "trailing bytes in proof"
    zt = 1
    for p in super_points:
        zt = zt * (u - p) % R
    rhs = None
    vp = 1
    z0 = None
    for pts, keys in rot_com:
        zi = 1
        for p in super_points:
            if p not in pts:
                zi = zi * (u - p) % R
        if z0 is None:
            z0 = zi
        yp = 1
        for key in keys:
            low = lagrange_interpolate(list(pts), [evmap[(key, p)] for p in pts])
            r_u = P.eval_polynomial(low, u)
            term = P.g1_add(coms[key], P.g1_neg(P.g1_mul(P.G1_GEN, r_u)))
            rhs = P.g1_add(rhs, P.g1_mul(term, vp * zi % R * yp % R))
            yp = yp * yy % R
        vp = vp * v % R
    rhs = P.g1_add(rhs, P.g1_neg(P.g1_mul(h1, zt)))
    lhs = P.g1_mul(h2, (tau - u) * z0 % R)
    assert lhs == rhs, "SHPLONK opening equation does not hold"
    return True
