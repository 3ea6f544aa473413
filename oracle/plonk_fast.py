"""ORACLE — TEST INFRASTRUCTURE ONLY. The protocol of oracle/plonk_ref.py (same step order, same
transcripts, same RNG) with every O(n) loop delegated to the C++ oracle (oracle/liboracle.so:
oracle_arith.cpp, oracle_domain.cpp, oracle_plonk.cpp, OpenMP), so that a CPU `create_proof` of the
reference circuit's real shape (k = 15, 112 advice columns, 24 lookups) finishes in tens of seconds.

Uses: (1) byte-for-byte comparison of the MI355X's proof with a CPU prover at the real size
(tests/test_gpu_prover.py), (2) the `cpu_baseline` leg of bench.py — a full create_proof on the host
cores. It must produce exactly plonk_ref.create_proof's bytes (tests/test_oracle_plonk.py checks that
at small k), which in turn are pinned as described in plonk_ref.py / contract_sol.py.

Commitments use the test SRS's known trapdoor: commit(p) = p(s) * G (one Horner evaluation and one
scalar multiplication; the same group element as the MSM over g[i] = s^i G). For timing a real CPU MSM
per commitment, pass `msm_bases=(g, g_lagrange)` — then every commitment is oracle_best_multiexp.
"""
import ctypes as C
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import plonk_ref as PR  # noqa: E402
import pyref as P  # noqa: E402

R = P.R
_L = None


def lib():
    global _L
    if _L is None:
        _L = C.CDLL(os.path.join(HERE, "liboracle.so"))
        _L.oracle_domain_new.restype = C.c_void_p
        _L.oracle_domain_extended_k.restype = C.c_uint32
        _L.oracle_permute_pair.restype = C.c_int
        _L.oracle_max_threads.restype = C.c_int
    return _L


def cgroup_cpu_quota():
    """(text, cores): the CPU bandwidth limit of this process's cgroup — cgroup v2 `cpu.max` ("max 100000" = none) or
    v1 `cpu.cfs_quota_us` / `cpu.cfs_period_us` — as read, and as a number of cores (None = no limit)."""
    for path in ("/sys/fs/cgroup/cpu.max",):
        try:
            txt = open(path).read().strip()
            q, per = txt.split()[:2]
            return txt, (None if q == "max" else float(q) / float(per))
        except (OSError, ValueError):
            pass
    try:
        q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
        per = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
        return "%d %d" % (q, per), (None if q <= 0 else q / per)
    except (OSError, ValueError):
        return None, None


def host_cores():
    """What the CPU baseline may use: the cores of this process's affinity mask, bounded by its cgroup's CPU quota if
    it has one (threads beyond a quota are throttled, not run). rayon's default pool — what upstream's create_proof
    would use (/root/reference/Cargo.lock:890) — is the affinity count; the quota bound keeps the figure honest on a
    box that hands out a share of a larger host."""
    avail = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    txt, quota = cgroup_cpu_quota()
    use = avail if quota is None else max(1, min(avail, int(quota + 0.999)))
    return {"affinity_cores": avail, "cgroup_cpu_max": txt, "cgroup_quota_cores": quota, "usable_cores": use}


_THREADS = None


def set_threads(n):
    """Override the thread count of every OpenMP loop of this module (None = back to the default rule)."""
    global _THREADS
    _THREADS = None if n is None else max(1, int(n))


def threads():
    """OpenMP threads of the oracle's loops: set_threads() > ORACLE_THREADS > all usable host cores (host_cores())."""
    if _THREADS is not None:
        return _THREADS
    env = os.environ.get("ORACLE_THREADS")
    if env:
        return max(1, int(env))
    return max(1, host_cores()["usable_cores"])


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


def fr_from_ints(xs):
    raw = np.frombuffer(b"".join(int(x).to_bytes(32, "little") for x in xs), dtype=np.uint64).reshape(-1, 4).copy() if len(xs) else np.zeros((0, 4), np.uint64)
    out = np.empty_like(raw)
    lib().oracle_fr_from_raw(_p(raw), _p(out), C.c_size_t(len(xs)))
    return out


def fr1(x):
    return fr_from_ints([x % R])[0].copy()


def fr_to_int(a):
    return P.from_mont(P.from_limbs64(a), R)


def _bin(fn, a, b):
    out = np.empty_like(a)
    getattr(lib(), fn)(_p(a), _p(np.ascontiguousarray(b)), _p(out), C.c_size_t(a.shape[0]))
    return out


mul = lambda a, b: _bin("oracle_fr_mul", a, b)
add = lambda a, b: _bin("oracle_fr_add", a, b)


def scale_add_const(a, s_int, c_int):
    out = np.empty_like(a)
    lib().oracle_fr_scale_add_const(_p(out), _p(a), _p(fr1(s_int)), _p(fr1(c_int)), C.c_size_t(a.shape[0]), C.c_int(threads()))
    return out


def axpy(out, a, s_int):
    lib().oracle_fr_axpy(_p(out), _p(a), _p(fr1(s_int)), C.c_size_t(a.shape[0]), C.c_int(threads()))


def eval_poly(poly, x_int):
    out = np.zeros(4, np.uint64)
    lib().oracle_eval_polynomial(_p(poly), C.c_size_t(poly.shape[0]), _p(fr1(x_int)), _p(out))
    return fr_to_int(out)


class FastRng:
    """ChaCha20Rng::seed_from_u64 + Fr::random, the same stream as plonk_ref.ChaCha20Rng, with the block
    function vectorised over all blocks of a request (numpy uint32 lanes)."""

    def __init__(self, seed):
        self.key = np.array(PR.ChaCha20Rng(seed).key, dtype=np.uint32)
        self.counter = 0

    def _blocks(self, nblocks):
        ctr = np.arange(self.counter, self.counter + nblocks, dtype=np.uint64)
        self.counter += nblocks
        st = np.zeros((16, nblocks), dtype=np.uint32)
        st[0:4] = np.array([0x61707865, 0x3320646E, 0x79622D32, 0x6B206574], dtype=np.uint32)[:, None]
        st[4:12] = self.key[:, None]
        st[12] = (ctr & np.uint64(0xFFFFFFFF)).astype(np.uint32)
        st[13] = (ctr >> np.uint64(32)).astype(np.uint32)
        x = st.copy()

        def rotl(v, n_):
            return (v << np.uint32(n_)) | (v >> np.uint32(32 - n_))

        def qr(a, b, c, d_):
            x[a] += x[b]; x[d_] = rotl(x[d_] ^ x[a], 16)
            x[c] += x[d_]; x[b] = rotl(x[b] ^ x[c], 12)
            x[a] += x[b]; x[d_] = rotl(x[d_] ^ x[a], 8)
            x[c] += x[d_]; x[b] = rotl(x[b] ^ x[c], 7)

        with np.errstate(over="ignore"):
            for _ in range(10):
                qr(0, 4, 8, 12); qr(1, 5, 9, 13); qr(2, 6, 10, 14); qr(3, 7, 11, 15)
                qr(0, 5, 10, 15); qr(1, 6, 11, 12); qr(2, 7, 8, 13); qr(3, 4, 9, 14)
            x += st
        return x.T.copy()  # (nblocks, 16) words in stream order

    def fr_many(self, count):
        """`count` Fr::random draws (one 64-byte block each: 8 next_u64 = 16 words) as Python ints."""
        if count == 0:
            return []
        raw = self._blocks(count).astype("<u4").tobytes()
        return [int.from_bytes(raw[64 * i:64 * i + 64], "little") % R for i in range(count)]

    def fr(self):
        return self.fr_many(1)[0]


class FastDomain:
    def __init__(self, j, k):
        self.h = C.c_void_p(lib().oracle_domain_new(C.c_uint32(j), C.c_uint32(k)))
        self.k, self.n, self.j = k, 1 << k, j
        self.extended_k = lib().oracle_domain_extended_k(self.h)
        self.py = PR.Domain(j, k)

    def ext(self):
        return 1 << self.extended_k

    def lagrange_to_coeff(self, a):
        a = a.copy()
        lib().oracle_lagrange_to_coeff(self.h, _p(a), C.c_int(threads()))
        return a

    def coeff_to_extended(self, a):
        out = np.zeros((self.ext(), 4), np.uint64)
        lib().oracle_coeff_to_extended(self.h, _p(np.ascontiguousarray(a)), _p(out), C.c_int(threads()))
        return out

    def h_to_coeff(self, h):
        h = h.copy()
        lib().oracle_divide_by_vanishing_poly(self.h, _p(h))
        lib().oracle_extended_to_coeff(self.h, _p(h), C.c_int(threads()))
        return h[: self.n * (self.j - 1)]


class Flat:
    """Own flattener of the description's expression tuples (independent of the product's)."""

    def __init__(self, desc):
        self.consts, self.words, self.offsets = {}, [], [0]
        exprs = list(desc["gates"])
        self.shape = []
        for lk in desc["lookups"]:
            self.shape += [len(lk["inputs"]), len(lk["tables"])]
            exprs += list(lk["inputs"]) + list(lk["tables"])
        for e in exprs:
            self._emit(e)
            self.offsets.append(len(self.words))
        self.nexprs = len(exprs)
        self.words_a = np.array(self.words, dtype=np.uint32)
        self.offsets_a = np.array(self.offsets, dtype=np.uint32)
        self.shape_a = np.array(self.shape if self.shape else [0], dtype=np.uint32)
        vals = [v for v, _ in sorted(self.consts.items(), key=lambda kv: kv[1])] or [0]
        self.consts_a = fr_from_ints(vals)

    def _c(self, v):
        return self.consts.setdefault(v % R, len(self.consts))

    def _emit(self, e):
        op = e[0]
        if op == "const":
            self.words.append((1 << 24) | self._c(e[1]))
        elif op in ("fixed", "advice", "instance"):
            self.words.append(({"fixed": 2, "advice": 3, "instance": 4}[op] << 24) | (e[1] << 8) | (e[2] + 128))
        elif op == "neg":
            self._emit(e[1])
            self.words.append(5 << 24)
        elif op in ("sum", "product"):
            self._emit(e[1])
            self._emit(e[2])
            self.words.append((6 if op == "sum" else 7) << 24)
        else:
            self._emit(e[1])
            self.words.append((8 << 24) | self._c(e[2]))


def ptr_array(arrs):
    return (C.c_void_p * max(1, len(arrs)))(*[a.ctypes.data for a in arrs])


class HArgs(C.Structure):
    _fields_ = [("words", C.c_void_p), ("offsets", C.c_void_p), ("num_gates", C.c_uint32), ("num_lookups", C.c_uint32),
                ("lookup_shape", C.c_void_p), ("fixed", C.c_void_p), ("advice", C.c_void_p), ("instance", C.c_void_p),
                ("sigma", C.c_void_p), ("zp", C.c_void_p), ("lz", C.c_void_p), ("la", C.c_void_p), ("ls", C.c_void_p),
                ("l0", C.c_void_p), ("l_last", C.c_void_p), ("l_active", C.c_void_p), ("xcoset", C.c_void_p), ("consts", C.c_void_p),
                ("perm_cols", C.c_void_p), ("num_perm", C.c_uint32), ("nsets", C.c_uint32), ("chunk", C.c_uint32),
                ("blinding_factors", C.c_uint32), ("beta", C.c_void_p), ("gamma", C.c_void_p), ("theta", C.c_void_p), ("y", C.c_void_p),
                ("delta", C.c_void_p), ("rows", C.c_size_t), ("rot_scale", C.c_long)]


class FastKey:
    """keygen_pk on arrays. fixed_values: list of int columns; mapping: Assembly.mapping."""

    def __init__(self, desc, fixed_values, mapping, tau, transcript_repr, msm_bases=None):
        self.desc, self.k, self.n, self.tau, self.transcript_repr = desc, desc["k"], 1 << desc["k"], tau, transcript_repr
        self.msm_bases = msm_bases
        n = self.n
        self.dom = d = FastDomain(desc["cs_degree"], self.k)
        self.flat = Flat(desc)
        bf = desc["blinding_factors"]
        self.fixed = [fr_from_ints(c) for c in fixed_values]
        self.fixed_poly = [d.lagrange_to_coeff(c) for c in self.fixed]
        self.fixed_coset = [d.coeff_to_extended(p) for p in self.fixed_poly]
        w = d.py.omega
        op = [1] * n
        for i in range(1, n):
            op[i] = op[i - 1] * w % R
        self.omega_pow = fr_from_ints(op)
        S = len(desc["permutation_columns"])
        dp = [pow(P.DELTA, i, R) for i in range(S)]
        self.sigma = []
        for i in range(S):
            self.sigma.append(fr_from_ints([dp[mapping[i][j][0]] * op[mapping[i][j][1]] % R for j in range(n)]))
        self.sigma_poly = [d.lagrange_to_coeff(c) for c in self.sigma]
        self.sigma_coset = [d.coeff_to_extended(p) for p in self.sigma_poly]
        l0, ll, lb = [0] * n, [0] * n, [0] * n
        l0[0] = 1
        ll[n - bf - 1] = 1
        for i in range(n - bf, n):
            lb[i] = 1
        to_c = lambda v: d.coeff_to_extended(d.lagrange_to_coeff(fr_from_ints(v)))
        self.l0_c, self.ll_c, lb_c = to_c(l0), to_c(ll), to_c(lb)
        one = np.tile(fr1(1), (d.ext(), 1))
        neg = lambda a: scale_add_const(a, R - 1, 0)
        self.lact_c = add(one, neg(add(self.ll_c, lb_c)))
        xc = [0] * d.ext()
        cur = P.ZETA
        for i in range(d.ext()):
            xc[i] = cur
            cur = cur * d.py.extended_omega % R
        self.xcoset = fr_from_ints(xc)
        self.fixed_commitments = [self.commit(p, None) for p in self.fixed_poly]
        self.permutation_commitments = [self.commit(p, None) for p in self.sigma_poly]

    def commit(self, poly_coeff, lagrange):
        """Commitment to a polynomial given in coefficient form (and, when timing real MSMs, its
        Lagrange form)."""
        if self.msm_bases is not None:
            g, gl = self.msm_bases
            out = np.zeros(8, np.uint64)
            if lagrange is not None:
                lib().oracle_best_multiexp(_p(lagrange), _p(gl), C.c_size_t(self.n), C.c_int(threads()), _p(out))
            else:
                lib().oracle_best_multiexp(_p(np.ascontiguousarray(poly_coeff)), _p(g), C.c_size_t(poly_coeff.shape[0]), C.c_int(threads()), _p(out))
            x, y = P.from_mont(P.from_limbs64(out[:4]), P.Q), P.from_mont(P.from_limbs64(out[4:]), P.Q)
            return None if x == 0 and y == 0 else (x, y)
        return P.g1_mul(P.G1_GEN, eval_poly(poly_coeff, self.tau))


def create_proof(pk, instances, advice_values, seed, transcript="blake2b"):
    """Same bytes as plonk_ref.create_proof. advice_values: list of int columns (n each)."""
    desc, d, n = pk.desc, pk.dom, pk.n
    L = lib()
    T_ = threads()
    bf = desc["blinding_factors"]
    usable = n - (bf + 1)
    rng = FastRng(seed)
    T = PR.Blake2bWrite() if transcript == "blake2b" else PR.Keccak256Write()
    T.common_scalar(pk.transcript_repr)
    inst = []
    for col in instances:
        for v in col:
            T.common_scalar(v)
        inst.append(fr_from_ints(list(col) + [0] * (n - len(col))))
    inst_poly = [d.lagrange_to_coeff(c) for c in inst]
    A = desc["num_advice"]
    adv = [fr_from_ints(c) for c in advice_values]
    for c in adv:
        c[usable:] = fr_from_ints(rng.fr_many(bf + 1))
    for _ in adv:
        rng.fr()
    adv_poly = [d.lagrange_to_coeff(c) for c in adv]
    for c, p in zip(adv, adv_poly):
        T.write_point(pk.commit(p, c))
    theta = T.squeeze_challenge()
    fl = pk.flat
    fx_p, ad_p, in_p = ptr_array(pk.fixed), ptr_array(adv), ptr_array(inst)
    lookups = []
    e = len(desc["gates"])
    for li, lk in enumerate(desc["lookups"]):
        ni, nt = len(lk["inputs"]), len(lk["tables"])

        def compress(first, cnt):
            out = np.zeros((n, 4), np.uint64)
            L.oracle_eval_compressed(_p(fl.words_a), _p(fl.offsets_a[first:].copy()), C.c_uint32(cnt), fx_p, ad_p, in_p, _p(fl.consts_a),
                                     C.c_size_t(n), C.c_long(1), _p(fr1(theta)), _p(out), C.c_int(T_))
            return out

        ci, ct = compress(e, ni), compress(e + ni, nt)
        e += ni + nt
        a_ = np.zeros((n, 4), np.uint64)
        s_ = np.zeros((n, 4), np.uint64)
        assert L.oracle_permute_pair(_p(ci), _p(ct), C.c_size_t(usable), _p(a_), _p(s_)), "lookup input not in table"
        a_[usable:] = fr_from_ints(rng.fr_many(bf + 1))
        s_[usable:] = fr_from_ints(rng.fr_many(bf + 1))
        pa = d.lagrange_to_coeff(a_)
        rng.fr()
        ca = pk.commit(pa, a_)
        ps = d.lagrange_to_coeff(s_)
        rng.fr()
        cs_ = pk.commit(ps, s_)
        T.write_point(ca)
        T.write_point(cs_)
        lookups.append({"ci": ci, "ct": ct, "a": a_, "s": s_, "pa": pa, "ps": ps})
    beta = T.squeeze_challenge()
    gamma = T.squeeze_challenge()
    cols = desc["permutation_columns"]
    chunk = desc["cs_degree"] - 2
    colvals = lambda kc: (adv, pk.fixed, inst)[kc[0]][kc[1]]
    sets, sets_lag = [], []
    last_z = 1
    dcol = 1
    for s0 in range(0, len(cols), chunk):
        cset = cols[s0:s0 + chunk]
        mod = None
        for j, kc in enumerate(cset):
            t = add(scale_add_const(pk.sigma[s0 + j], beta, gamma), colvals(kc))
            mod = t if mod is None else mul(mod, t)
        L.oracle_batch_invert(_p(mod), C.c_size_t(n))
        for kc in cset:
            t = add(scale_add_const(pk.omega_pow, dcol * beta % R, gamma), colvals(kc))
            mod = mul(mod, t)
            dcol = dcol * P.DELTA % R
        z = np.zeros((n, 4), np.uint64)
        L.oracle_running_product(_p(fr1(last_z)), _p(mod), C.c_size_t(n), _p(z))
        z[n - bf:] = fr_from_ints(rng.fr_many(bf))
        last_z = fr_to_int(z[usable])
        rng.fr()
        pz = d.lagrange_to_coeff(z)
        T.write_point(pk.commit(pz, z))
        sets.append(pz)
    for lk in lookups:
        den = mul(scale_add_const(lk["a"], 1, beta), scale_add_const(lk["s"], 1, gamma))
        L.oracle_batch_invert(_p(den), C.c_size_t(n))
        prod = mul(mul(den, scale_add_const(lk["ci"], 1, beta)), scale_add_const(lk["ct"], 1, gamma))
        z = np.zeros((n, 4), np.uint64)
        L.oracle_running_product(_p(fr1(1)), _p(prod), C.c_size_t(n), _p(z))
        z[n - bf:] = fr_from_ints(rng.fr_many(bf))
        rng.fr()
        lk["pz"] = d.lagrange_to_coeff(z)
        T.write_point(pk.commit(lk["pz"], z))
    random_poly = fr_from_ints(rng.fr_many(n))
    rng.fr()
    T.write_point(pk.commit(random_poly, None))
    y = T.squeeze_challenge()
    # quotient
    X = d.coeff_to_extended
    adv_c, inst_c = [X(p) for p in adv_poly], [X(p) for p in inst_poly]
    zp_c, lz_c = [X(p) for p in sets], [X(l["pz"]) for l in lookups]
    la_c, ls_c = [X(l["pa"]) for l in lookups], [X(l["ps"]) for l in lookups]
    perm = np.array(cols if cols else [(0, 0)], dtype=np.uint32).reshape(-1, 2)
    keep = [ptr_array(pk.fixed_coset), ptr_array(adv_c), ptr_array(inst_c), ptr_array(pk.sigma_coset), ptr_array(zp_c), ptr_array(lz_c),
            ptr_array(la_c), ptr_array(ls_c)]
    consts = [fr1(v) for v in (beta, gamma, theta, y, P.DELTA)]
    ha = HArgs(fl.words_a.ctypes.data, fl.offsets_a.ctypes.data, len(desc["gates"]), len(lookups), fl.shape_a.ctypes.data,
               C.cast(keep[0], C.c_void_p), C.cast(keep[1], C.c_void_p), C.cast(keep[2], C.c_void_p), C.cast(keep[3], C.c_void_p),
               C.cast(keep[4], C.c_void_p), C.cast(keep[5], C.c_void_p), C.cast(keep[6], C.c_void_p), C.cast(keep[7], C.c_void_p),
               pk.l0_c.ctypes.data, pk.ll_c.ctypes.data, pk.lact_c.ctypes.data, pk.xcoset.ctypes.data, fl.consts_a.ctypes.data,
               perm.ctypes.data, len(cols), len(sets), chunk, bf, consts[0].ctypes.data, consts[1].ctypes.data, consts[2].ctypes.data,
               consts[3].ctypes.data, consts[4].ctypes.data, d.ext(), 1 << (d.extended_k - d.k))
    h_ext = np.zeros((d.ext(), 4), np.uint64)
    L.oracle_evaluate_h(C.byref(ha), _p(h_ext), C.c_int(T_))
    h = d.h_to_coeff(h_ext)
    qd = desc["cs_degree"] - 1
    pieces = [np.ascontiguousarray(h[i * n:(i + 1) * n]) for i in range(qd)]
    for _ in pieces:
        rng.fr()
    for pc in pieces:
        T.write_point(pk.commit(pc, None))
    x = T.squeeze_challenge()
    xn = pow(x, n, R)
    rot = lambda r: d.py.rotate_omega(x, r)
    ev = lambda poly, r: eval_poly(poly, rot(r))
    evals = {}

    def E(poly, r):
        key = (id(poly), r)
        if key not in evals:
            evals[key] = ev(poly, r)
        return evals[key]

    for c, r in desc["advice_queries"]:
        T.write_scalar(E(adv_poly[c], r))
    for c, r in desc["fixed_queries"]:
        T.write_scalar(E(pk.fixed_poly[c], r))
    h_poly = np.zeros((n, 4), np.uint64)
    xp = 1
    for pc in pieces:
        axpy(h_poly, pc, xp)
        xp = xp * xn % R
    T.write_scalar(E(random_poly, 0))
    for p in pk.sigma_poly:
        T.write_scalar(E(p, 0))
    for si, pz in enumerate(sets):
        T.write_scalar(E(pz, 0))
        T.write_scalar(E(pz, 1))
        if si + 1 < len(sets):
            T.write_scalar(E(pz, -(bf + 1)))
    for lk in lookups:
        for poly, r in ((lk["pz"], 0), (lk["pz"], 1), (lk["pa"], 0), (lk["pa"], -1), (lk["ps"], 0)):
            T.write_scalar(E(poly, r))
    queries = []
    Qy = lambda poly, r: queries.append((poly, rot(r), E(poly, r)))
    for c, r in desc["advice_queries"]:
        Qy(adv_poly[c], r)
    for pz in sets:
        Qy(pz, 0)
        Qy(pz, 1)
    for pz in reversed(sets[:-1]):
        Qy(pz, -(bf + 1))
    for lk in lookups:
        Qy(lk["pz"], 0); Qy(lk["pa"], 0); Qy(lk["ps"], 0); Qy(lk["pa"], -1); Qy(lk["pz"], 1)
    for c, r in desc["fixed_queries"]:
        Qy(pk.fixed_poly[c], r)
    for p in pk.sigma_poly:
        Qy(p, 0)
    Qy(h_poly, 0)
    Qy(random_poly, 0)
    _shplonk(pk, queries, T, n)
    return bytes(T.proof)


def _shplonk(pk, queries, T, n):
    ids = {}
    keyed, evmap = [], {}
    for poly, pt, e in queries:
        ids.setdefault(id(poly), poly)
        keyed.append((id(poly), pt))
        evmap[(id(poly), pt)] = e
    rot_com, super_points = PR.intermediate_sets(keyed)
    y = T.squeeze_challenge()
    v = T.squeeze_challenge()
    sets = []
    for pts, keys in rot_com:
        coms = [(ids[k_], PR.lagrange_interpolate(list(pts), [evmap[(k_, p)] for p in pts])) for k_ in keys]
        sets.append((list(pts), coms))
    hx = np.zeros((n, 4), np.uint64)
    Ls = []
    vp = 1
    for pts, coms in sets:
        Li = np.zeros((n, 4), np.uint64)
        low = [0] * len(pts)
        yp = 1
        for poly, lw in coms:
            axpy(Li, poly, yp)
            for t in range(len(lw)):
                low[t] = (low[t] + yp * lw[t]) % R
            yp = yp * y % R
        Ls.append(Li)
        nx = Li.copy()
        nx[: len(low)] = fr_from_ints([(fr_to_int(nx[t]) - low[t]) % R for t in range(len(low))])
        for p in pts:
            q = np.zeros((nx.shape[0] - 1, 4), np.uint64)
            lib().oracle_kate_division(_p(nx), C.c_size_t(nx.shape[0]), _p(fr1(p)), _p(q))
            nx = q
        nx = np.concatenate([nx, np.zeros((n - nx.shape[0], 4), np.uint64)])
        axpy(hx, nx, vp)
        vp = vp * v % R
    T.write_point(pk.commit(hx, None))
    u = T.squeeze_challenge()
    zt = 1
    for p in super_points:
        zt = zt * (u - p) % R
    lx = np.zeros((n, 4), np.uint64)
    vp, z0, cterm = 1, None, 0
    for (pts, coms), Li in zip(sets, Ls):
        zi = 1
        for p in super_points:
            if p not in pts:
                zi = zi * (u - p) % R
        if z0 is None:
            z0 = zi
        ri, yp = 0, 1
        for poly, lw in coms:
            ri = (ri + yp * P.eval_polynomial(lw, u)) % R
            yp = yp * y % R
        wgt = vp * zi % R
        axpy(lx, Li, wgt)
        cterm = (cterm + wgt * ri) % R
        vp = vp * v % R
    axpy(lx, hx, (-zt) % R)
    lx[0] = fr1((fr_to_int(lx[0]) - cterm) % R)
    q = np.zeros((n - 1, 4), np.uint64)
    lib().oracle_kate_division(_p(lx), C.c_size_t(n), _p(fr1(u)), _p(q))
    h2 = scale_add_const(np.concatenate([q, np.zeros((1, 4), np.uint64)]), pow(z0, -1, R), 0)
    T.write_point(pk.commit(h2, None))
