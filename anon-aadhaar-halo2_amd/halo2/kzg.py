"""Mirror of halo2_proofs::poly::kzg::commitment::ParamsKZG (v2023_01_20 [UP]) — the resident part."""
import ctypes as C

import numpy as np

from ..ffi import _ptr
from . import arithmetic

BASIS_G = 0
BASIS_G_LAGRANGE = 1


class ParamsKZG:
    """Holds {k, n, g, g_lagrange} on the device. `g`/`g_lagrange`: (n, 8) uint64 affine points."""

    def __init__(self, ctx, k, g=None, g_lagrange=None):
        self.ctx, self.k, self.n = ctx, k, 1 << k
        for a in (g, g_lagrange):
            if a is not None:
                assert a.shape == (self.n, 8) and a.dtype == np.uint64
        self._g = None if g is None else np.ascontiguousarray(g)
        self._gl = None if g_lagrange is None else np.ascontiguousarray(g_lagrange)
        h = C.c_void_p()
        ctx._chk(ctx.L.amdzk_srs_upload(ctx.h, None if g is None else _ptr(self._g),
                                        None if g_lagrange is None else _ptr(self._gl), k, C.byref(h)))
        self.h = h

    @classmethod
    def setup(cls, ctx, k, s, want_host_copy=False):
        """ParamsKZG::setup(k, rng) with the trapdoor `s` ((4,) uint64 Montgomery Fr) given explicitly
        (test / benchmark SRS, as unsafe as upstream's setup). Bases are generated on the device."""
        self = cls.__new__(cls)
        self.ctx, self.k, self.n = ctx, k, 1 << k
        s = np.ascontiguousarray(s, dtype=np.uint64).reshape(4)
        self._g = np.zeros((self.n, 8), np.uint64) if want_host_copy else None
        self._gl = np.zeros((self.n, 8), np.uint64) if want_host_copy else None
        h = C.c_void_p()
        ctx._chk(ctx.L.amdzk_srs_setup(ctx.h, k, _ptr(s), C.byref(h), None if self._g is None else _ptr(self._g),
                                       None if self._gl is None else _ptr(self._gl)))
        self.h = h
        return self

    def write(self, g2=bytes(64), s_g2=bytes(64)):
        """ParamsKZG::write: serialized parameters as bytes (g2 / s_g2 are passed through)."""
        size = self.ctx.L.amdzk_srs_serialized_size(self.k)
        buf = (C.c_uint8 * size)()
        a, b = (C.c_uint8 * 64)(*g2), (C.c_uint8 * 64)(*s_g2)
        self.ctx._chk(self.ctx.L.amdzk_srs_write(self.ctx.h, self.h, a, b, buf, size))
        return bytes(buf)

    @classmethod
    def read(cls, ctx, data):
        """ParamsKZG::read: returns (params, g2 bytes, s_g2 bytes)."""
        self = cls.__new__(cls)
        self.ctx = ctx
        self.k = int.from_bytes(data[:4], "little")
        self.n = 1 << self.k
        self._g = self._gl = None
        h = C.c_void_p()
        g2, s_g2 = (C.c_uint8 * 64)(), (C.c_uint8 * 64)()
        raw = (C.c_uint8 * len(data)).from_buffer_copy(data)
        ctx._chk(ctx.L.amdzk_srs_read(ctx.h, raw, len(data), C.byref(h), g2, s_g2))
        self.h = h
        return self, bytes(g2), bytes(s_g2)

    def downsize(self, k):
        """ParamsKZG::downsize(k): shrink in place to 2^k — g truncated, g_lagrange recomputed on the
        device with g_to_lagrange."""
        h = C.c_void_p()
        self.ctx._chk(self.ctx.L.amdzk_srs_downsize(self.ctx.h, self.h, k, C.byref(h)))
        self.ctx.L.amdzk_srs_free(self.ctx.h, self.h)
        self.h, self.k, self.n = h, k, 1 << k
        self._g = None if self._g is None else np.ascontiguousarray(self._g[: self.n])
        self._gl = None if self._gl is None else self.get_g_lagrange()

    def get_g(self):
        """ParamsKZG::get_g(): (n, 8) uint64 affine points."""
        out = np.zeros((self.n, 8), np.uint64)
        self.ctx._chk(self.ctx.L.amdzk_srs_get(self.ctx.h, self.h, BASIS_G, _ptr(out)))
        return out

    def get_g_lagrange(self):
        out = np.zeros((self.n, 8), np.uint64)
        self.ctx._chk(self.ctx.L.amdzk_srs_get(self.ctx.h, self.h, BASIS_G_LAGRANGE, _ptr(out)))
        return out

    def commit(self, poly_coeff):
        """ParamsKZG::commit(poly, _blind): MSM with g[..len] (the blind is ignored for KZG)."""
        return arithmetic.best_multiexp(self.ctx, self.h, BASIS_G, poly_coeff)

    def commit_lagrange(self, poly_lagrange):
        """ParamsKZG::commit_lagrange(poly, _blind): MSM with g_lagrange[..len]."""
        return arithmetic.best_multiexp(self.ctx, self.h, BASIS_G_LAGRANGE, poly_lagrange)

    def free(self):
        if self.h:
            self.ctx.L.amdzk_srs_free(self.ctx.h, self.h)
            self.h = None
