"""Mirror of halo2_proofs::plonk::{ConstraintSystem, Expression, Column, keygen, create_proof}
(v2023_01_20 [UP], /root/reference/Cargo.lock:469-471) — the interface the reference's circuits
implement and call: `Circuit::configure(meta)` (/root/reference/src/lib.rs:295-326,
src/signal.rs:27-49, src/conditional_secrets.rs:81-187, src/timestamp.rs:58-138).

This module is host-side description only: it records columns, queries, gates, lookups and the
permutation exactly the way upstream's ConstraintSystem does (query indices in first-use order,
`degree()`, `blinding_factors()`), and flattens the result into the arrays the C ABI takes
(`amdzk_circuit_*`, include/amdzk.h). All O(n) proving work happens on the device.

Selectors are modelled after upstream's selector compression has run, i.e. as fixed columns.
Field constants are Python ints (canonical); they are converted to Montgomery limbs at export.
"""
from dataclasses import dataclass, field
from typing import List, Tuple

R = 21888242871839275222246405745257275088548364400416034343698204186575808495617

ADVICE, FIXED, INSTANCE = 0, 1, 2  # upstream's `Any` ordering: Advice < Fixed < Instance


@dataclass(frozen=True)
class Column:
    kind: int
    index: int


class Expression:
    """plonk::Expression: Constant | Fixed | Advice | Instance | Negated | Sum | Product | Scaled."""

    __slots__ = ("op", "a", "b")

    def __init__(self, op, a=None, b=None):
        self.op, self.a, self.b = op, a, b

    @staticmethod
    def constant(v):
        return Expression("const", v % R)

    def __neg__(self):
        return Expression("neg", self)

    def __add__(self, o):
        return Expression("sum", self, _lift(o))

    def __radd__(self, o):
        return Expression("sum", _lift(o), self)

    def __sub__(self, o):
        return Expression("sum", self, Expression("neg", _lift(o)))

    def __rsub__(self, o):
        return Expression("sum", _lift(o), Expression("neg", self))

    def __mul__(self, o):
        if isinstance(o, int):
            return Expression("scaled", self, o % R)
        return Expression("product", self, o)

    def __rmul__(self, o):
        return self.__mul__(o)

    def degree(self):
        op = self.op
        if op == "const":
            return 0
        if op in ("fixed", "advice", "instance"):
            return 1
        if op == "neg":
            return self.a.degree()
        if op == "sum":
            return max(self.a.degree(), self.b.degree())
        if op == "product":
            return self.a.degree() + self.b.degree()
        if op == "scaled":
            return self.a.degree()
        raise ValueError(op)

    def to_tuple(self):
        op = self.op
        if op == "const":
            return ("const", self.a)
        if op in ("fixed", "advice", "instance"):
            return (op, self.a, self.b)  # (column index, rotation)
        if op == "neg":
            return ("neg", self.a.to_tuple())
        if op in ("sum", "product"):
            return (op, self.a.to_tuple(), self.b.to_tuple())
        if op == "scaled":
            return ("scaled", self.a.to_tuple(), self.b)
        raise ValueError(op)


def _lift(o):
    return o if isinstance(o, Expression) else Expression.constant(o)


class VirtualCells:
    """What `meta.create_gate(|meta| ...)` / `meta.lookup(|meta| ...)` closures receive."""

    def __init__(self, cs):
        self.cs = cs

    def query_advice(self, col, rot=0):
        assert col.kind == ADVICE
        if self.cs._query(self.cs.advice_queries, col, rot):
            self.cs.num_advice_queries[col.index] += 1
        return Expression("advice", col.index, rot)

    def query_fixed(self, col, rot=0):
        assert col.kind == FIXED
        self.cs._query(self.cs.fixed_queries, col, rot)
        return Expression("fixed", col.index, rot)

    def query_selector(self, col):
        return self.query_fixed(col, 0)

    def query_instance(self, col, rot=0):
        assert col.kind == INSTANCE
        self.cs._query(self.cs.instance_queries, col, rot)
        return Expression("instance", col.index, rot)


class ConstraintSystem:
    def __init__(self):
        self.num_fixed = self.num_advice = self.num_instance = 0
        self.advice_queries: List[Tuple[Column, int]] = []
        self.fixed_queries: List[Tuple[Column, int]] = []
        self.instance_queries: List[Tuple[Column, int]] = []
        self.num_advice_queries: List[int] = []
        self.gates: List[Expression] = []       # one entry per polynomial constraint, in order
        self.lookups: List[Tuple[List[Expression], List[Expression]]] = []
        self.permutation_columns: List[Column] = []
        self.minimum_degree = None

    # ---- columns
    def advice_column(self):
        self.num_advice += 1
        self.num_advice_queries.append(0)
        return Column(ADVICE, self.num_advice - 1)

    def fixed_column(self):
        self.num_fixed += 1
        return Column(FIXED, self.num_fixed - 1)

    selector = fixed_column  # post-compression model

    def instance_column(self):
        self.num_instance += 1
        return Column(INSTANCE, self.num_instance - 1)

    def _query(self, lst, col, rot):
        """query_*_index: register (column, rotation) on first use; True if it was new."""
        if (col, rot) in lst:
            return False
        lst.append((col, rot))
        return True

    def enable_equality(self, col):
        """ConstraintSystem::enable_equality: query at Rotation::cur() + add to the permutation."""
        if col.kind == ADVICE:
            VirtualCells(self).query_advice(col, 0)
        elif col.kind == FIXED:
            VirtualCells(self).query_fixed(col, 0)
        else:
            VirtualCells(self).query_instance(col, 0)
        if col not in self.permutation_columns:
            self.permutation_columns.append(col)

    def create_gate(self, fn):
        polys = fn(VirtualCells(self))
        if isinstance(polys, Expression):
            polys = [polys]
        assert polys, "create_gate: gates must contain at least one constraint"
        self.gates.extend(polys)

    def lookup(self, fn):
        """fn(meta) -> [(input_expr, table_expr), ...]"""
        pairs = fn(VirtualCells(self))
        self.lookups.append(([p[0] for p in pairs], [p[1] for p in pairs]))
        return len(self.lookups) - 1

    # ---- derived quantities (upstream formulas)
    def degree(self):
        d = 3  # permutation::Argument::required_degree() is the constant 3 upstream
        for inputs, tables in self.lookups:
            di = max([1] + [e.degree() for e in inputs])
            dt = max([1] + [e.degree() for e in tables])
            d = max(d, max(4, 2 + di + dt))
        for g in self.gates:
            d = max(d, g.degree())
        return max(d, self.minimum_degree or 1)

    def blinding_factors(self):
        factors = max(self.num_advice_queries) if self.num_advice_queries else 1
        return max(3, factors) + 2

    def minimum_rows(self):
        return self.blinding_factors() + 3

    def describe(self, k):
        """Plain-data description of the constraint system (what a Rust fork would export)."""
        cols = lambda qs: [(c.index, r) for c, r in qs]
        return {
            "k": k,
            "num_fixed": self.num_fixed, "num_advice": self.num_advice, "num_instance": self.num_instance,
            "blinding_factors": self.blinding_factors(), "cs_degree": self.degree(),
            "advice_queries": cols(self.advice_queries), "fixed_queries": cols(self.fixed_queries),
            "instance_queries": cols(self.instance_queries),
            "gates": [g.to_tuple() for g in self.gates],
            "lookups": [{"inputs": [e.to_tuple() for e in i], "tables": [e.to_tuple() for e in t]} for i, t in self.lookups],
            "permutation_columns": [(c.kind, c.index) for c in self.permutation_columns],
        }


class Assembly:
    """plonk::permutation::keygen::Assembly: the copy-constraint cycles. mapping[col][row] =
    (col', row') with columns indexed by their position in `permutation_columns`."""

    def __init__(self, n, ncols):
        self.n, self.ncols = n, ncols
        self.mapping = [[(c, r) for r in range(n)] for c in range(ncols)]
        self.aux = [[(c, r) for r in range(n)] for c in range(ncols)]
        self.sizes = [[1] * n for _ in range(ncols)]

    def copy(self, lc, lr, rc, rr):
        """Assembly::copy (upstream's union by size of two cycles)."""
        left_cycle, right_cycle = self.aux[lc][lr], self.aux[rc][rr]
        if left_cycle == right_cycle:
            return
        if self.sizes[left_cycle[0]][left_cycle[1]] < self.sizes[right_cycle[0]][right_cycle[1]]:
            left_cycle, right_cycle = right_cycle, left_cycle
        self.sizes[left_cycle[0]][left_cycle[1]] += self.sizes[right_cycle[0]][right_cycle[1]]
        i, j = right_cycle
        while True:
            self.aux[i][j] = left_cycle
            i, j = self.mapping[i][j]
            if (i, j) == right_cycle:
                break
        tmp = self.mapping[lc][lr]
        self.mapping[lc][lr] = self.mapping[rc][rr]
        self.mapping[rc][rr] = tmp


# ------------------------------------------------------------------------------------------------
# Export to the C ABI (include/amdzk.h: amdzk_circuit) and the keygen / create_proof entry points.
import ctypes as C  # noqa: E402

import numpy as np  # noqa: E402

_XOP = {"const": 1, "fixed": 2, "advice": 3, "instance": 4, "neg": 5, "sum": 6, "product": 7, "scaled": 8}


class _CCircuit(C.Structure):
    _fields_ = [("k", C.c_uint32), ("num_fixed", C.c_uint32), ("num_advice", C.c_uint32), ("num_instance", C.c_uint32),
                ("blinding_factors", C.c_uint32), ("cs_degree", C.c_uint32),
                ("num_advice_queries", C.c_uint32), ("advice_queries", C.c_void_p),
                ("num_fixed_queries", C.c_uint32), ("fixed_queries", C.c_void_p),
                ("num_instance_queries", C.c_uint32), ("instance_queries", C.c_void_p),
                ("num_gates", C.c_uint32), ("num_lookups", C.c_uint32), ("num_exprs", C.c_uint32),
                ("lookup_shape", C.c_void_p), ("expr_offsets", C.c_void_p), ("expr_words", C.c_void_p),
                ("num_constants", C.c_uint32), ("constants", C.c_void_p),
                ("num_perm_columns", C.c_uint32), ("perm_columns", C.c_void_p)]


def _mont_limbs(v):
    v = (v % R) * (1 << 256) % R
    return [(v >> (64 * i)) & 0xFFFFFFFFFFFFFFFF for i in range(4)]


def flatten_circuit(desc):
    """describe() dict -> (amdzk_circuit, keepalive list)."""
    consts, words, offsets = {}, [], [0]

    def cidx(v):
        return consts.setdefault(v % R, len(consts))

    def emit(e):
        op = e[0]
        if op == "const":
            words.append((_XOP[op] << 24) | cidx(e[1]))
        elif op in ("fixed", "advice", "instance"):
            assert -128 <= e[2] < 128 and e[1] < (1 << 16)
            words.append((_XOP[op] << 24) | (e[1] << 8) | (e[2] + 128))
        elif op == "neg":
            emit(e[1])
            words.append(_XOP[op] << 24)
        elif op in ("sum", "product"):
            emit(e[1])
            emit(e[2])
            words.append(_XOP[op] << 24)
        elif op == "scaled":
            emit(e[1])
            words.append((_XOP[op] << 24) | cidx(e[2]))
        else:
            raise ValueError(op)

    exprs = list(desc["gates"])
    shape = []
    for lk in desc["lookups"]:
        shape += [len(lk["inputs"]), len(lk["tables"])]
        exprs += list(lk["inputs"]) + list(lk["tables"])
    for e in exprs:
        emit(e)
        offsets.append(len(words))
    arrs = {
        "aq": np.array(desc["advice_queries"], dtype=np.int32).reshape(-1, 2),
        "fq": np.array(desc["fixed_queries"], dtype=np.int32).reshape(-1, 2),
        "iq": np.array(desc["instance_queries"], dtype=np.int32).reshape(-1, 2),
        "shape": np.array(shape, dtype=np.uint32),
        "off": np.array(offsets, dtype=np.uint32),
        "words": np.array(words, dtype=np.uint32),
        "consts": np.array([_mont_limbs(v) for v, _ in sorted(consts.items(), key=lambda kv: kv[1])], dtype=np.uint64).reshape(-1, 4),
        "perm": np.array(desc["permutation_columns"], dtype=np.uint32).reshape(-1, 2),
    }
    arrs = {k_: np.ascontiguousarray(v) for k_, v in arrs.items()}
    p = lambda a: a.ctypes.data if a.size else None
    c = _CCircuit(desc["k"], desc["num_fixed"], desc["num_advice"], desc["num_instance"], desc["blinding_factors"], desc["cs_degree"],
                  len(arrs["aq"]), p(arrs["aq"]), len(arrs["fq"]), p(arrs["fq"]), len(arrs["iq"]), p(arrs["iq"]),
                  len(desc["gates"]), len(desc["lookups"]), len(exprs), p(arrs["shape"]), p(arrs["off"]), p(arrs["words"]),
                  len(arrs["consts"]), p(arrs["consts"]), len(arrs["perm"]), p(arrs["perm"]))
    return c, arrs


class ProvingKey:
    """plonk::keygen::{keygen_vk, keygen_pk} result, resident on the device."""

    def __init__(self, ctx, params, desc, fixed_values, mapping, transcript_repr, flags=None):
        """fixed_values: (num_fixed, n, 4) uint64 Montgomery; mapping: Assembly.mapping;
        transcript_repr: (4,) uint64 Montgomery Fr. flags: None = amdzk_keygen (the key's modes from the environment),
        else amdzk_keygen_ex with KEYGEN_FULL_COSETS / KEYGEN_SERIAL or'ed together."""
        self.ctx, self.params, self.desc = ctx, params, desc
        n = 1 << desc["k"]
        cc, keep = flatten_circuit(desc)
        fv = np.ascontiguousarray(fixed_values, dtype=np.uint64).reshape(desc["num_fixed"], n, 4) if desc["num_fixed"] else np.zeros((0, n, 4), np.uint64)
        mp = np.ascontiguousarray(np.array(mapping, dtype=np.uint32).reshape(-1, n, 2)) if len(desc["permutation_columns"]) else np.zeros((0, n, 2), np.uint32)
        tr = np.ascontiguousarray(transcript_repr, dtype=np.uint64).reshape(4)
        h = C.c_void_p()
        if flags is None:
            ctx._chk(ctx.L.amdzk_keygen(ctx.h, params.h, C.byref(cc), fv.ctypes.data if fv.size else None,
                                        mp.ctypes.data if mp.size else None, tr.ctypes.data, C.byref(h)))
        else:
            ctx._chk(ctx.L.amdzk_keygen_ex(ctx.h, params.h, C.byref(cc), fv.ctypes.data if fv.size else None,
                                           mp.ctypes.data if mp.size else None, tr.ctypes.data, int(flags), C.byref(h)))
        self.h = h
        del keep

    def clone_workspace(self):
        """amdzk_pk_clone_workspace: a key that shares this key's material and owns one more circuit instance's per-proof
        workspace — the second, third, ... instance of create_proof_multi, or one more proof of this circuit in flight.
        Free it before this key."""
        c = object.__new__(ProvingKey)
        c.ctx, c.params, c.desc = self.ctx, self.params, self.desc
        h = C.c_void_p()
        self.ctx._chk(self.ctx.L.amdzk_pk_clone_workspace(self.ctx.h, self.h, C.byref(h)))
        c.h = h
        return c

    def commitments(self):
        """(fixed_commitments, permutation_commitments) as (m, 8) uint64 affine points."""
        f = np.zeros((self.desc["num_fixed"], 8), np.uint64)
        p = np.zeros((len(self.desc["permutation_columns"]), 8), np.uint64)
        self.ctx._chk(self.ctx.L.amdzk_pk_commitments(self.h, f.ctypes.data if f.size else None, p.ctypes.data if p.size else None))
        return f, p

    def inspect(self, what):
        """Test hook (amdzk_pk_inspect): what the last create_proof left in the key's workspace. 0: the committed
        polynomials in coefficient form (NP, n, 4), arena order advice | instance | A' | S' | Z_perm | Z_lookup;
        1: the challenges theta, beta, gamma, y (4, 4); 2: the pieces of h(X) (cs_degree - 1, n, 4)."""
        cnt = C.c_size_t(0)
        self.ctx._chk(self.ctx.L.amdzk_pk_inspect(self.ctx.h, self.h, what, None, 0, C.byref(cnt)))
        out = np.zeros((cnt.value, 4), np.uint64)
        self.ctx._chk(self.ctx.L.amdzk_pk_inspect(self.ctx.h, self.h, what, out.ctypes.data, cnt.value, C.byref(cnt)))
        n = 1 << self.desc["k"]
        return out if what == 1 else out.reshape(-1, n, 4)

    def quotient_eval(self, polys, theta, beta, gamma, y):
        """evaluate_h + the quotient (amdzk_quotient_eval_dev): committed polynomials in coefficient form (NP, n, 4)
        and the challenges -> the pieces of h(X) (cs_degree - 1, n, 4)."""
        n = 1 << self.desc["k"]
        polys = np.ascontiguousarray(polys, dtype=np.uint64).reshape(-1, n, 4)
        d_in = self.ctx.alloc(polys.nbytes).upload(polys)
        qdeg = self.desc["cs_degree"] - 1
        d_out = self.ctx.alloc(qdeg * n * 32)
        ch = [np.ascontiguousarray(c, dtype=np.uint64).reshape(4) for c in (theta, beta, gamma, y)]
        try:
            self.ctx._chk(self.ctx.L.amdzk_quotient_eval_dev(self.ctx.h, self.h, d_in.ptr, n, ch[0].ctypes.data, ch[1].ctypes.data,
                                                             ch[2].ctypes.data, ch[3].ctypes.data, d_out.ptr))
            return d_out.download((qdeg, n, 4))
        finally:
            d_in.free(); d_out.free()

    def free(self):
        if self.h:
            self.ctx.L.amdzk_pk_free(self.ctx.h, self.h)
            self.h = None


KEYGEN_FULL_COSETS, KEYGEN_SERIAL = 1, 2  # include/amdzk.h AMDZK_KEYGEN_*
TRANSCRIPT_BLAKE2B, TRANSCRIPT_KECCAK256_EVM = 0, 1
MULTIOPEN_GWC = 0x100  # OR into `transcript`: poly::kzg::multiopen::ProverGWC instead of ProverSHPLONK


def create_proof(ctx, pk, instances, d_advice, seed, advice_stride=None, transcript=TRANSCRIPT_BLAKE2B):
    """plonk::create_proof(params, pk, &[circuit], &[instances], ChaCha20Rng::seed_from_u64(seed), transcript).
    instances: list of (len, 4) uint64 arrays; d_advice: DeviceBuffer (or anything with .ptr) holding
    num_advice columns of n rows. Returns the proof bytes."""
    n = 1 << pk.desc["k"]
    cols = [np.ascontiguousarray(c, dtype=np.uint64).reshape(-1, 4) for c in instances]
    ptrs = (C.c_void_p * max(1, len(cols)))(*[c.ctypes.data if c.size else None for c in cols])
    lens = (C.c_size_t * max(1, len(cols)))(*[c.shape[0] for c in cols])
    need = C.c_size_t(0)
    cap = 1 << 20
    buf = (C.c_uint8 * cap)()
    ctx._chk(ctx.L.amdzk_create_proof_ex(ctx.h, pk.h, ptrs, lens, d_advice.ptr if d_advice is not None else None, advice_stride or n,
                                         C.c_uint64(seed), transcript, buf, cap, C.byref(need)))
    return bytes(buf[: need.value])


def create_proof_multi(ctx, pks, instances_list, d_advice_list, seed, advice_stride=None, transcript=TRANSCRIPT_BLAKE2B):
    """plonk::create_proof(params, pk, &[circuit; N], &[instances; N], ChaCha20Rng::seed_from_u64(seed), transcript):
    N instances of the key's circuit in one proof. pks: the key and N - 1 workspace clones of it
    (ProvingKey.clone_workspace); instances_list[c]: instance c's columns; d_advice_list[c]: its advice on the device."""
    N = len(pks)
    assert N >= 1 and len(instances_list) == N and len(d_advice_list) == N
    n = 1 << pks[0].desc["k"]
    keep, inst_pp, lens_pp = [], (C.POINTER(C.c_void_p) * N)(), (C.POINTER(C.c_size_t) * N)()
    for c, instances in enumerate(instances_list):
        cols = [np.ascontiguousarray(col, dtype=np.uint64).reshape(-1, 4) for col in instances]
        ptrs = (C.c_void_p * max(1, len(cols)))(*[col.ctypes.data if col.size else None for col in cols])
        lens = (C.c_size_t * max(1, len(cols)))(*[col.shape[0] for col in cols])
        keep += [cols, ptrs, lens]
        inst_pp[c] = C.cast(ptrs, C.POINTER(C.c_void_p))
        lens_pp[c] = C.cast(lens, C.POINTER(C.c_size_t))
    keys = (C.c_void_p * N)(*[pk.h for pk in pks])
    adv = (C.c_void_p * N)(*[d.ptr if d is not None else None for d in d_advice_list])
    need = C.c_size_t(0)
    cap = 1 << 22
    buf = (C.c_uint8 * cap)()
    ctx._chk(ctx.L.amdzk_create_proof_multi(ctx.h, keys, N, inst_pp, lens_pp, adv, advice_stride or n, C.c_uint64(seed), transcript, buf, cap,
                                            C.byref(need)))
    return bytes(buf[: need.value])


def proof_size_multi(ctx, pk, n_circuits, transcript=TRANSCRIPT_BLAKE2B):
    return int(ctx.L.amdzk_proof_size_multi(pk.h, n_circuits, transcript))


def proof_size(ctx, pk, transcript=TRANSCRIPT_BLAKE2B):
    return int(ctx.L.amdzk_proof_size(pk.h, transcript))


def proof_random_count(ctx, pk):
    return int(ctx.L.amdzk_proof_random_count(pk.h))


def create_proof_with_scalars(ctx, pk, instances, d_advice, scalars, advice_stride=None, transcript=TRANSCRIPT_BLAKE2B):
    """create_proof with the caller's own RngCore: `scalars` = the Fr::random draws in upstream order."""
    n = 1 << pk.desc["k"]
    cols = [np.ascontiguousarray(c, dtype=np.uint64).reshape(-1, 4) for c in instances]
    ptrs = (C.c_void_p * max(1, len(cols)))(*[c.ctypes.data if c.size else None for c in cols])
    lens = (C.c_size_t * max(1, len(cols)))(*[c.shape[0] for c in cols])
    sc = np.ascontiguousarray(scalars, dtype=np.uint64).reshape(-1, 4)
    need = C.c_size_t(0)
    cap = 1 << 20
    buf = (C.c_uint8 * cap)()
    ctx._chk(ctx.L.amdzk_create_proof_scalars(ctx.h, pk.h, ptrs, lens, d_advice.ptr if d_advice is not None else None, advice_stride or n,
                                              sc.ctypes.data, sc.shape[0], transcript, buf, cap, C.byref(need)))
    return bytes(buf[: need.value])
