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
