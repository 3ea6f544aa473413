"""ORACLE (second, independent leg) — TEST INFRASTRUCTURE ONLY.

Pure-Python big-integer restatement of the BN254 arithmetic and of the small-size forms of the
hot-path functions (DFT, MSM, coset extension). It shares no code with the C++ oracle
(oracle/*.cpp, 4 x u64 Montgomery) nor with the product (8 x u32 Montgomery on gfx950); the three
must agree. Used to generate tests/golden/*.json (tests/golden/gen_golden.py).

"Parity unpinned" vs the reference: /root/reference holds no vector for this path.
What the reference does pin and this file checks at import:
  solidity_verifier_contract/contract.sol:210-211  q, r
  solidity_verifier_contract/contract.sol:82       y^2 = x^3 + 3
  solidity_verifier_contract/contract.sol:440      delta = 7^(2^28) mod r  (=> generator 7, S = 28)
Upstream algorithms restated ([UP] = not under /root/reference):
  halo2_proofs 0.2.0 @ v2023_01_20 src/arithmetic.rs, src/poly/domain.rs   (Cargo.lock:469-471)
  halo2curves 0.3.1 bn256                                                  (Cargo.lock:484-486)
"""

Q = 21888242871839275222246405745257275088696311157297823662689037894645226208583  # contract.sol:210
R = 21888242871839275222246405745257275088548364400416034343698204186575808495617  # contract.sol:211
CURVE_B = 3  # contract.sol:82
DELTA = 4131629893567559867359510883348571134090853742863529169391034518566172092834  # contract.sol:440
S = 28
GENERATOR = 7
assert pow(GENERATOR, 1 << S, R) == DELTA
ROOT_OF_UNITY = pow(GENERATOR, (R - 1) >> S, R)
assert pow(ROOT_OF_UNITY, 1 << S, R) == 1 and pow(ROOT_OF_UNITY, 1 << (S - 1), R) != 1
# halo2curves Fr::ZETA [UP recall]: the cube root of unity with this value (the other one is ZETA^2).
ZETA = 0x30644E72E131A029048B6E193FD84104CC37A73FEC2BC5E9B8CA0B2D36636F23
assert pow(ZETA, 3, R) == 1 and ZETA != 1
MONT_R = 1 << 256


def omega(k):
    """EvaluationDomain::new [UP]: omega = ROOT_OF_UNITY^(2^(S-k))."""
    return pow(ROOT_OF_UNITY, 1 << (S - k), R)


def to_mont(x, p):
    return x * MONT_R % p


def from_mont(x, p):
    return x * pow(MONT_R, -1, p) % p


def limbs64(x):
    return [(x >> (64 * i)) & 0xFFFFFFFFFFFFFFFF for i in range(4)]


def from_limbs64(l):
    return sum(int(v) << (64 * i) for i, v in enumerate(l))


# ----------------------------------------------------------------------------- G1 (affine, None = identity)
G1_GEN = (1, 2)


def g1_on_curve(p):
    if p is None:
        return True
    x, y = p
    return (y * y - x * x * x - CURVE_B) % Q == 0


def g1_add(p, q):
    if p is None:
        return q
    if q is None:
        return p
    x1, y1 = p
    x2, y2 = q
    if x1 == x2:
        if (y1 + y2) % Q == 0:
            return None
        lam = 3 * x1 * x1 * pow(2 * y1, -1, Q) % Q
    else:
        lam = (y2 - y1) * pow(x2 - x1, -1, Q) % Q
    x3 = (lam * lam - x1 - x2) % Q
    y3 = (lam * (x1 - x3) - y1) % Q
    return (x3, y3)


def g1_neg(p):
    return None if p is None else (p[0], (-p[1]) % Q)


def g1_mul(p, e):
    e %= R
    acc = None
    while e:
        if e & 1:
            acc = g1_add(acc, p)
        p = g1_add(p, p)
        e >>= 1
    return acc


def msm_naive(scalars, bases):
    acc = None
    for s, b in zip(scalars, bases):
        acc = g1_add(acc, g1_mul(b, s))
    return acc


def msm_pippenger(scalars, bases):
    """arithmetic.rs multiexp_serial [UP] on integers (single chunk)."""
    import math

    n = len(bases)
    c = 1 if n < 4 else 3 if n < 32 else math.ceil(math.log(n))
    segments = 256 // c + 1
    acc = None
    for seg in reversed(range(segments)):
        for _ in range(c):
            acc = g1_add(acc, acc)
        buckets = [None] * ((1 << c) - 1)
        for s, b in zip(scalars, bases):
            d = (s >> (seg * c)) & ((1 << c) - 1)
            if d:
                buckets[d - 1] = g1_add(buckets[d - 1], b)
        running = None
        for b in reversed(buckets):
            running = g1_add(running, b)
            acc = g1_add(acc, running)
    return acc


# ----------------------------------------------------------------------------- NTT
def dft_naive(a, w):
    n = len(a)
    return [sum(a[i] * pow(w, i * j, R) for i in range(n)) % R for j in range(n)]


def best_fft(a, w, log_n):
    """arithmetic.rs best_fft [UP] (serial form): bit-reverse, then DIT rounds."""
    n = 1 << log_n
    a = list(a)
    for k in range(n):
        rk = int(format(k, "0%db" % log_n)[::-1], 2) if log_n else 0
        if k < rk:
            a[k], a[rk] = a[rk], a[k]
    tw = [pow(w, i, R) for i in range(n // 2)]
    chunk, tchunk = 2, n // 2
    for _ in range(log_n):
        for base in range(0, n, chunk):
            for i in range(chunk // 2):
                t = a[base + i + chunk // 2] * tw[i * tchunk] % R
                u = a[base + i]
                a[base + i] = (u + t) % R
                a[base + i + chunk // 2] = (u - t) % R
        chunk *= 2
        tchunk //= 2
    return a


def g_to_lagrange(g, k):
    """arithmetic.rs g_to_lagrange [UP]: best_fft over the group with omega^-1 (same bit-reversal + DIT
    network as best_fft above, the twiddle product being a scalar multiplication), then every point
    times 1/n. g: n affine points (None = identity); returns n affine points."""
    n = 1 << k
    w = pow(omega(k), -1, R)
    a = list(g)
    for i in range(n):
        ri = int(format(i, "0%db" % k)[::-1], 2) if k else 0
        if i < ri:
            a[i], a[ri] = a[ri], a[i]
    tw = [pow(w, i, R) for i in range(n // 2)]
    chunk, tchunk = 2, n // 2
    for _ in range(k):
        for base in range(0, n, chunk):
            for i in range(chunk // 2):
                t = g1_mul(a[base + i + chunk // 2], tw[i * tchunk])
                u = a[base + i]
                a[base + i] = g1_add(u, t)
                a[base + i + chunk // 2] = g1_add(u, g1_neg(t))
        chunk *= 2
        tchunk //= 2
    n_inv = pow(n, -1, R)
    return [g1_mul(p, n_inv) for p in a]


class EvaluationDomain:
    """poly/domain.rs EvaluationDomain [UP]: new(j, k), lagrange_to_coeff, coeff_to_extended,
    extended_to_coeff, divide_by_vanishing_poly, rotate_extended, l_i_range."""

    def __init__(self, j, k):
        self.k = k
        self.n = 1 << k
        self.quotient_poly_degree = j - 1
        ek = k
        while (1 << ek) < self.n * self.quotient_poly_degree:
            ek += 1
        self.extended_k = ek
        self.extended_omega = pow(ROOT_OF_UNITY, 1 << (S - ek), R)
        self.omega = pow(self.extended_omega, 1 << (ek - k), R)
        self.omega_inv = pow(self.omega, -1, R)
        self.extended_omega_inv = pow(self.extended_omega, -1, R)
        self.g_coset = ZETA
        self.g_coset_inv = ZETA * ZETA % R
        self.ifft_divisor = pow(self.n, -1, R)
        self.extended_ifft_divisor = pow(1 << ek, -1, R)
        # t_evaluations[i] = 1 / ((zeta * extended_omega^i)^n - 1), period 2^(ek-k)
        orig = pow(ZETA, self.n, R)
        step = pow(self.extended_omega, self.n, R)
        te, cur = [], orig
        while True:
            te.append(cur)
            cur = cur * step % R
            if cur == orig:
                break
        assert len(te) == 1 << (ek - k)
        self.t_evaluations = [pow(t - 1, -1, R) for t in te]

    def extended_len(self):
        return 1 << self.extended_k

    def lagrange_to_coeff(self, a):
        out = best_fft(a, self.omega_inv, self.k)
        return [x * self.ifft_divisor % R for x in out]

    def coeff_to_lagrange(self, a):
        return best_fft(a, self.omega, self.k)

    def _distribute_powers_zeta(self, a, into_coset):
        cp = [self.g_coset, self.g_coset_inv] if into_coset else [self.g_coset_inv, self.g_coset]
        out = []
        for idx, x in enumerate(a):
            i = idx % 3
            out.append(x if i == 0 else x * cp[i - 1] % R)
        return out

    def coeff_to_extended(self, a):
        assert len(a) == self.n
        a = self._distribute_powers_zeta(a, True) + [0] * (self.extended_len() - self.n)
        return best_fft(a, self.extended_omega, self.extended_k)

    def extended_to_coeff(self, a):
        assert len(a) == self.extended_len()
        a = best_fft(a, self.extended_omega_inv, self.extended_k)
        a = [x * self.extended_ifft_divisor % R for x in a]
        a = self._distribute_powers_zeta(a, False)
        return a[: self.n * self.quotient_poly_degree]

    def divide_by_vanishing_poly(self, a):
        m = len(self.t_evaluations)
        return [x * self.t_evaluations[i % m] % R for i, x in enumerate(a)]


def eval_polynomial(poly, x):
    acc = 0
    for c in reversed(poly):
        acc = (acc * x + c) % R
    return acc


def kate_division(a, b):
    """q = (a(X) - a(b)) / (X - b)  (arithmetic.rs kate_division [UP])."""
    q = [0] * (len(a) - 1)
    tmp = 0
    nb = (-b) % R
    for i in reversed(range(len(a) - 1)):
        lead = (a[i + 1] - tmp) % R
        q[i] = lead
        tmp = lead * nb % R
    return q


# ----------------------------------------------------------------------------- deterministic inputs
class SplitMix64:
    """splitmix64, the seed expander BASELINE.md §3 names for synthetic inputs."""

    def __init__(self, seed):
        self.s = seed & 0xFFFFFFFFFFFFFFFF

    def next(self):
        self.s = (self.s + 0x9E3779B97F4A7C15) & 0xFFFFFFFFFFFFFFFF
        z = self.s
        z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & 0xFFFFFFFFFFFFFFFF
        z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & 0xFFFFFFFFFFFFFFFF
        return z ^ (z >> 31)

    def fr(self):
        v = 0
        for i in range(4):
            v |= self.next() << (64 * i)
        return v % R
