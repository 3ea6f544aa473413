"""Test helpers: numpy <-> field conversions and the ctypes wrapper of the CPU oracle.
The oracle is test infrastructure; nothing under anon-aadhaar-halo2_amd/ imports this."""
import ctypes as C
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
R = 21888242871839275222246405745257275088548364400416034343698204186575808495617
Q = 21888242871839275222246405745257275088696311157297823662689037894645226208583
MONT = 1 << 256


def limbs(x):
    return [(x >> (64 * i)) & 0xFFFFFFFFFFFFFFFF for i in range(4)]


def from_limbs(l):
    return sum(int(v) << (64 * i) for i, v in enumerate(l))


def fr_from_int(x):
    return np.array(limbs(x % R * MONT % R), dtype=np.uint64)


def fr_array_from_ints(xs):
    return np.array([limbs(x % R * MONT % R) for x in xs], dtype=np.uint64).reshape(-1, 4)


def fr_to_int(a):
    return from_limbs(a) * pow(MONT, -1, R) % R


def fr_array_to_ints(a):
    inv = pow(MONT, -1, R)
    return [from_limbs(row) * inv % R for row in np.asarray(a).reshape(-1, 4)]


def fq_to_int(a):
    return from_limbs(a) * pow(MONT, -1, Q) % Q


def point_to_ints(p):
    """(8,) uint64 affine -> (x, y) ints or None for the identity."""
    x, y = fq_to_int(p[:4]), fq_to_int(p[4:8])
    return None if x == 0 and y == 0 else (x, y)


def point_from_ints(p):
    if p is None:
        return np.zeros(8, dtype=np.uint64)
    return np.array(limbs(p[0] * MONT % Q) + limbs(p[1] * MONT % Q), dtype=np.uint64)


def splitmix64(seed, count):
    """Vectorised splitmix64 stream (BASELINE.md §3's seed expander)."""
    idx = np.arange(1, count + 1, dtype=np.uint64)
    with np.errstate(over="ignore"):
        z = np.uint64(seed) + idx * np.uint64(0x9E3779B97F4A7C15)
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        return z ^ (z >> np.uint64(31))


def random_fr(n, seed):
    """n pseudo-random Montgomery-form Fr elements: 253 random bits taken as the Montgomery limbs
    (always < r, i.e. a valid representation of some field element)."""
    a = splitmix64(seed, 4 * n).reshape(n, 4).copy()
    a[:, 3] >>= np.uint64(11)
    return a


def skewed_fr(n, seed, oracle):
    """Witness-like column: 70% small (<2^64), 20% zero, 10% uniform."""
    u = random_fr(n, seed)
    sel = splitmix64(seed + 1, n) % np.uint64(10)
    small = np.zeros((n, 4), dtype=np.uint64)
    small[:, 0] = splitmix64(seed + 2, n)
    small = oracle.fr_from_raw(small)
    out = np.where((sel < 7)[:, None], small, u)
    out[(sel >= 7) & (sel < 9)] = 0
    return np.ascontiguousarray(out)


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


class Oracle:
    """ctypes view of oracle/liboracle.so (built by oracle/Makefile)."""

    def __init__(self):
        path = os.path.join(ROOT, "oracle", "liboracle.so")
        if not os.path.exists(path):
            subprocess.check_call(["make"], cwd=os.path.join(ROOT, "oracle"))
        self.L = C.CDLL(path)
        self.L.oracle_max_threads.restype = C.c_int
        # ORACLE_THREADS, else every core this process may use (affinity mask bounded by the cgroup quota: plonk_fast.host_cores)
        sys.path.insert(0, os.path.join(ROOT, "oracle"))
        import plonk_fast as _pf
        self.threads = _pf.threads()

    def _bin(self, fn, a, b):
        a, b = np.ascontiguousarray(a, np.uint64), np.ascontiguousarray(b, np.uint64)
        out = np.empty_like(a)
        getattr(self.L, fn)(_p(a), _p(b), _p(out), C.c_size_t(a.size // 4))
        return out

    def _un(self, fn, a):
        a = np.ascontiguousarray(a, np.uint64)
        out = np.empty_like(a)
        getattr(self.L, fn)(_p(a), _p(out), C.c_size_t(a.size // 4))
        return out

    def fr_mul(self, a, b): return self._bin("oracle_fr_mul", a, b)
    def fr_add(self, a, b): return self._bin("oracle_fr_add", a, b)
    def fr_sub(self, a, b): return self._bin("oracle_fr_sub", a, b)
    def fr_inv(self, a): return self._un("oracle_fr_inv", a)
    def fr_from_raw(self, a): return self._un("oracle_fr_from_raw", a)
    def fr_to_raw(self, a): return self._un("oracle_fr_to_raw", a)
    def fq_mul(self, a, b): return self._bin("oracle_fq_mul", a, b)
    def fq_from_raw(self, a): return self._un("oracle_fq_from_raw", a)
    def fq_to_raw(self, a): return self._un("oracle_fq_to_raw", a)

    def omega(self, k):
        out = np.zeros(4, np.uint64)
        self.L.oracle_fr_omega(C.c_uint32(k), _p(out))
        return out

    def constants(self):
        r, z, d = (np.zeros(4, np.uint64) for _ in range(3))
        self.L.oracle_fr_constants(_p(r), _p(z), _p(d))
        return r, z, d

    def best_fft(self, a, omega, log_n, threads=None):
        a = np.ascontiguousarray(a, np.uint64)
        self.L.oracle_best_fft(_p(a), _p(np.ascontiguousarray(omega)), C.c_uint32(log_n), C.c_int(threads or self.threads))
        return a

    def dft_naive(self, a, omega):
        a = np.ascontiguousarray(a, np.uint64)
        out = np.empty_like(a)
        self.L.oracle_dft_naive(_p(a), C.c_size_t(a.shape[0]), _p(np.ascontiguousarray(omega)), _p(out))
        return out

    def generator(self):
        out = np.zeros(8, np.uint64)
        self.L.oracle_g1_generator(_p(out))
        return out

    def srs_powers(self, tau, n):
        out = np.zeros((n, 8), np.uint64)
        self.L.oracle_srs_powers(_p(np.ascontiguousarray(tau)), _p(out), C.c_size_t(n))
        return out

    def g1_mul_many(self, base, scalars):
        scalars = np.ascontiguousarray(scalars, np.uint64)
        out = np.zeros((scalars.shape[0], 8), np.uint64)
        self.L.oracle_g1_mul_many(_p(np.ascontiguousarray(base)), _p(scalars), _p(out), C.c_size_t(scalars.shape[0]))
        return out

    def on_curve(self, p):
        self.L.oracle_g1_on_curve.restype = C.c_int
        return bool(self.L.oracle_g1_on_curve(_p(np.ascontiguousarray(p))))

    def best_multiexp(self, scalars, bases, threads=None):
        scalars, bases = np.ascontiguousarray(scalars, np.uint64), np.ascontiguousarray(bases, np.uint64)
        out = np.zeros(8, np.uint64)
        self.L.oracle_best_multiexp(_p(scalars), _p(bases), C.c_size_t(scalars.shape[0]), C.c_int(threads or self.threads), _p(out))
        return out

    def msm_naive(self, scalars, bases):
        scalars, bases = np.ascontiguousarray(scalars, np.uint64), np.ascontiguousarray(bases, np.uint64)
        out = np.zeros(8, np.uint64)
        self.L.oracle_msm_naive(_p(scalars), _p(bases), C.c_size_t(scalars.shape[0]), _p(out))
        return out

    def jac_to_affine(self, jac):
        jac = np.ascontiguousarray(jac, np.uint64).reshape(-1, 12)
        out = np.zeros((jac.shape[0], 8), np.uint64)
        self.L.oracle_g1_jac_to_affine(_p(jac), _p(out), C.c_size_t(jac.shape[0]))
        return out

    def eval_polynomial(self, poly, x):
        poly = np.ascontiguousarray(poly, np.uint64)
        out = np.zeros(4, np.uint64)
        self.L.oracle_eval_polynomial(_p(poly), C.c_size_t(poly.shape[0]), _p(np.ascontiguousarray(x)), _p(out))
        return out

    def kate_division(self, a, b):
        a = np.ascontiguousarray(a, np.uint64)
        out = np.zeros((a.shape[0] - 1, 4), np.uint64)
        self.L.oracle_kate_division(_p(a), C.c_size_t(a.shape[0]), _p(np.ascontiguousarray(b)), _p(out))
        return out

    def batch_invert(self, a):
        a = np.ascontiguousarray(a, np.uint64).copy()
        self.L.oracle_batch_invert(_p(a), C.c_size_t(a.shape[0]))
        return a


class OracleDomain:
    """oracle/oracle_domain.cpp: EvaluationDomain restatement."""
    NAMES = ("omega", "omega_inv", "extended_omega", "extended_omega_inv", "g_coset", "g_coset_inv",
             "ifft_divisor", "extended_ifft_divisor")

    def __init__(self, oracle, j, k):
        self.o, self.L = oracle, oracle.L
        self.L.oracle_domain_new.restype = C.c_void_p
        self.L.oracle_domain_extended_k.restype = C.c_uint32
        self.L.oracle_domain_t_len.restype = C.c_uint32
        self.h = C.c_void_p(self.L.oracle_domain_new(C.c_uint32(j), C.c_uint32(k)))
        self.k, self.n, self.j = k, 1 << k, j
        self.extended_k = self.L.oracle_domain_extended_k(self.h)
        for i, nm in enumerate(self.NAMES):
            v = np.zeros(4, np.uint64)
            self.L.oracle_domain_constant(self.h, C.c_int(i), _p(v))
            setattr(self, nm, v)
        t = np.zeros((self.L.oracle_domain_t_len(self.h), 4), np.uint64)
        self.L.oracle_domain_t_evaluations(self.h, _p(t))
        self.t_evaluations = t

    def extended_len(self):
        return 1 << self.extended_k

    def lagrange_to_coeff(self, a):
        a = np.ascontiguousarray(a, np.uint64).copy()
        self.L.oracle_lagrange_to_coeff(self.h, _p(a), C.c_int(self.o.threads))
        return a

    def coeff_to_lagrange(self, a):
        a = np.ascontiguousarray(a, np.uint64).copy()
        self.L.oracle_coeff_to_lagrange(self.h, _p(a), C.c_int(self.o.threads))
        return a

    def coeff_to_extended(self, a):
        a = np.ascontiguousarray(a, np.uint64)
        out = np.zeros((self.extended_len(), 4), np.uint64)
        self.L.oracle_coeff_to_extended(self.h, _p(a), _p(out), C.c_int(self.o.threads))
        return out

    def extended_to_coeff(self, a):
        a = np.ascontiguousarray(a, np.uint64).copy()
        self.L.oracle_extended_to_coeff(self.h, _p(a), C.c_int(self.o.threads))
        return a[: self.n * (self.j - 1)]

    def divide_by_vanishing_poly(self, a):
        a = np.ascontiguousarray(a, np.uint64).copy()
        self.L.oracle_divide_by_vanishing_poly(self.h, _p(a))
        return a


def jac_to_affine_host(oracle, jac):
    """Normalised Jacobian (z = 1 or identity) -> affine (8,) / (m,8); checks the normal form."""
    jac = np.asarray(jac, dtype=np.uint64).reshape(-1, 12)
    one = np.array(limbs(MONT % Q), dtype=np.uint64)
    out = np.zeros((jac.shape[0], 8), dtype=np.uint64)
    for i, j in enumerate(jac):
        if not j[8:].any():  # identity must be exactly (0, 1, 0)
            assert not j[:4].any() and np.array_equal(j[4:8], one), "identity not in normal form"
            continue
        assert np.array_equal(j[8:], one), "result not normalised (z != 1)"
        out[i] = j[:8]
    return out[0] if out.shape[0] == 1 else out


# ------------------------------------------------------------------------------ prover test plumbing
def ints_to_fr(oracle, xs):
    """Python ints (canonical) -> (len,4) uint64 Montgomery, via the oracle's from_raw (fast path)."""
    raw = np.zeros((len(xs), 4), dtype=np.uint64)
    mask = (1 << 64) - 1
    for i, x in enumerate(xs):
        if x:
            raw[i, 0] = x & mask
            raw[i, 1] = (x >> 64) & mask
            raw[i, 2] = (x >> 128) & mask
            raw[i, 3] = x >> 192
    return oracle.fr_from_raw(raw)


def test_srs(oracle, k, tau_int):
    """g[i] = tau^i G and g_lagrange[i] = L_i(tau) G for a known tau (ParamsKZG::setup's values [UP])."""
    n = 1 << k
    g = oracle.srs_powers(fr_from_int(tau_int), n)
    w = fr_to_int(oracle.omega(k))
    tn1 = (pow(tau_int, n, R) - 1) % R
    ninv = pow(n, -1, R)
    scal, wi = [], 1
    for i in range(n):
        scal.append(wi * tn1 % R * ninv % R * pow((tau_int - wi) % R, -1, R) % R)
        wi = wi * w % R
    gl = oracle.g1_mul_many(oracle.generator(), ints_to_fr(oracle, scal))
    return g, gl
