"""Mirror of halo2_proofs::arithmetic (v2023_01_20 [UP]) for the functions on the hot path.

Field elements are numpy uint64 arrays of shape (n, 4): halo2curves' in-memory Montgomery limbs.
"""
import ctypes as C

import numpy as np

from ..ffi import _ptr, as_fr_array

NTT_SCALE_NINV = 1


def best_fft(ctx, a, omega, log_n, flags=0):
    """arithmetic::best_fft(a, omega, log_n): in-place NTT of one host column (returns the array)."""
    a = as_fr_array(a)
    assert a.shape[0] == 1 << log_n, "best_fft: a.len() != 1 << log_n"
    omega = np.ascontiguousarray(omega, dtype=np.uint64).reshape(4)
    ctx._chk(ctx.L.amdzk_ntt_fr(ctx.h, _ptr(a), log_n, _ptr(omega), flags))
    return a


def best_fft_dev(ctx, dbuf, omega, log_n, ncols=1, col_stride=None, flags=0):
    """Same on `ncols` device-resident columns."""
    omega = np.ascontiguousarray(omega, dtype=np.uint64).reshape(4)
    if col_stride is None:
        col_stride = 1 << log_n
    ctx._chk(ctx.L.amdzk_ntt_fr_dev(ctx.h, dbuf.ptr, log_n, _ptr(omega), flags, ncols, col_stride))


def best_multiexp(ctx, srs, basis, coeffs):
    """arithmetic::best_multiexp(coeffs, bases) with bases = srs.{g|g_lagrange}[..len].
    Returns the normalised Jacobian point as (12,) uint64 (x, y, z=1 | identity (0,1,0))."""
    coeffs = as_fr_array(coeffs)
    out = np.zeros(12, dtype=np.uint64)
    ctx._chk(ctx.L.amdzk_msm_g1(ctx.h, srs, basis, _ptr(coeffs), coeffs.shape[0], _ptr(out)))
    return out


def best_multiexp_batch(ctx, srs, basis, columns):
    cols = [as_fr_array(c) for c in columns]
    n = cols[0].shape[0]
    assert all(c.shape[0] == n for c in cols)
    ptrs = (C.c_void_p * len(cols))(*[c.ctypes.data for c in cols])
    out = np.zeros((len(cols), 12), dtype=np.uint64)
    ctx._chk(ctx.L.amdzk_msm_g1_batch(ctx.h, srs, basis, ptrs, len(cols), n, _ptr(out)))
    return out


def best_multiexp_dev(ctx, srs, basis, dbuf, ncols, length, col_stride=None):
    if col_stride is None:
        col_stride = length
    out = np.zeros((ncols, 12), dtype=np.uint64)
    ctx._chk(ctx.L.amdzk_msm_g1_dev(ctx.h, srs, basis, dbuf.ptr, ncols, length, col_stride, _ptr(out)))
    return out


def g_to_lagrange(ctx, g, k):
    """arithmetic::g_to_lagrange(g_projective, k): (n, 8) uint64 affine points in the monomial basis ->
    the Lagrange-basis points (1/n)·FFT_{omega^-1}(g), affine."""
    g = np.ascontiguousarray(g, dtype=np.uint64)
    assert g.shape == (1 << k, 8)
    out = np.zeros_like(g)
    ctx._chk(ctx.L.amdzk_g_to_lagrange(ctx.h, _ptr(g), k, _ptr(out)))
    return out


# ---- the PLONK layer's helpers, function by function (include/amdzk.h "function by function"); host arrays in,
# host arrays out: each call uploads, runs the device kernel create_proof itself uses, and downloads.
def _dev_cols(ctx, cols):
    cols = [as_fr_array(c) for c in cols]
    n = cols[0].shape[0]
    assert all(c.shape[0] == n for c in cols)
    buf = ctx.alloc(max(1, len(cols) * n * 32))
    buf.upload(np.ascontiguousarray(np.stack(cols)) if n else np.zeros((0, 4), np.uint64))
    return buf, n


def batch_invert(ctx, a):
    """ff::BatchInvert: inverses of the non-zero elements, zeros stay zero."""
    a = as_fr_array(a)
    buf = ctx.alloc(max(1, a.nbytes)).upload(a)
    ctx._chk(ctx.L.amdzk_batch_invert_dev(ctx.h, buf.ptr, a.shape[0]))
    out = buf.download(a.shape)
    buf.free()
    return out


def batch_invert_assigned(ctx, numerators, denominators=None, in_place=False):
    """poly::batch_invert_assigned: Assigned::Rational(num, den) cells -> num * den^-1 (zero where den = 0); denominators=None:
    every cell Trivial."""
    num = as_fr_array(numerators)
    n = num.shape[0]
    d_num = ctx.alloc(max(1, num.nbytes)).upload(num)
    d_den = None
    if denominators is not None:
        den = as_fr_array(denominators)
        assert den.shape[0] == n
        d_den = ctx.alloc(max(1, den.nbytes)).upload(den)
    d_out = d_num if in_place else ctx.alloc(max(1, num.nbytes))
    ctx._chk(ctx.L.amdzk_batch_invert_assigned_dev(ctx.h, d_num.ptr, d_den.ptr if d_den is not None else None, n, d_out.ptr))
    out = d_out.download(num.shape)
    if not in_place:
        assert np.array_equal(d_num.download(num.shape), num)  # the numerators are read, not written
        d_out.free()
    d_num.free()
    if d_den is not None:
        d_den.free()
    return out


def grand_product(ctx, cols, chain=False, chain_row=0):
    """Running products z[0] = 1, z[i] = z[i-1] * f[i-1] of every column (permutation / lookup commit_product);
    chain: z_c[0] = z_{c-1}[chain_row]."""
    buf, n = _dev_cols(ctx, cols)
    ctx._chk(ctx.L.amdzk_grand_product_dev(ctx.h, buf.ptr, len(cols), n, n, 1 if chain else 0, chain_row))
    out = buf.download((len(cols), n, 4))
    buf.free()
    return out


def eval_polynomial(ctx, polys, points):
    """arithmetic::eval_polynomial(poly, point) for a list of (poly, point) pairs."""
    buf, n = _dev_cols(ctx, polys)
    pts = as_fr_array(points)
    assert pts.shape[0] == len(polys)
    ptrs = (C.c_void_p * len(polys))(*[buf.ptr.value + i * n * 32 for i in range(len(polys))])
    out = np.zeros((len(polys), 4), np.uint64)
    ctx._chk(ctx.L.amdzk_eval_poly_dev(ctx.h, ptrs, _ptr(pts), len(polys), n, _ptr(out)))
    buf.free()
    return out


def poly_axpy(ctx, polys, coefs):
    """sum_j coefs[j] * polys[j]."""
    buf, n = _dev_cols(ctx, polys)
    cf = as_fr_array(coefs)
    ptrs = (C.c_void_p * len(polys))(*[buf.ptr.value + i * n * 32 for i in range(len(polys))])
    out = ctx.alloc(max(1, n * 32))
    ctx._chk(ctx.L.amdzk_poly_axpy_dev(ctx.h, ptrs, _ptr(cf), len(polys), out.ptr, n, 0))
    res = out.download((n, 4))
    buf.free()
    out.free()
    return res


def kate_division(ctx, polys, roots):
    """arithmetic::kate_division(a, root) for each pair: (a(X) - a(root)) / (X - root), n - 1 coefficients."""
    buf, n = _dev_cols(ctx, polys)
    rt = as_fr_array(roots)
    ptrs = (C.c_void_p * len(polys))(*[buf.ptr.value + i * n * 32 for i in range(len(polys))])
    ctx._chk(ctx.L.amdzk_kate_div_dev(ctx.h, ptrs, _ptr(rt), len(polys), n))
    out = buf.download((len(polys), n, 4))
    buf.free()
    assert not out[:, n - 1].any(), "kate_division: top coefficient must be zero"
    return out[:, : n - 1]


def permute_expression_pair(ctx, inputs, tables, usable):
    """lookup::prover::permute_expression_pair for a list of (input, table) column pairs of n rows: returns
    (A', S') as (L, n, 4) arrays, rows >= usable zero (the caller blinds them)."""
    a_buf, n = _dev_cols(ctx, inputs)
    t_buf, n2 = _dev_cols(ctx, tables)
    assert n == n2 and len(inputs) == len(tables)
    s_buf = ctx.alloc(len(inputs) * n * 32)
    try:
        ctx._chk(ctx.L.amdzk_permute_expression_pair_dev(ctx.h, a_buf.ptr, t_buf.ptr, s_buf.ptr, len(inputs), n, usable))
        return a_buf.download((len(inputs), n, 4)), s_buf.download((len(inputs), n, 4))
    finally:
        a_buf.free(); t_buf.free(); s_buf.free()
