"""ctypes binding of include/amdzk.h. One `Context` = one amdzk_ctx (one GPU, one host thread)."""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


class AmdzkError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__("amdzk error %d: %s" % (code, msg))
        self.code = code


def lib_path():
    return os.path.join(_HERE, "libamdzk.so")


def lib():
    """Load libamdzk.so (built in-tree by __graft_entry__.build()). Fails loudly if missing."""
    global _LIB
    if _LIB is not None:
        return _LIB
    p = lib_path()
    if not os.path.exists(p):
        raise AmdzkError(-1, "%s not built: run `python -c 'import __graft_entry__ as g; g.build()'`" % p)
    L = C.CDLL(p)
    vp, sz, u32, i32 = C.c_void_p, C.c_size_t, C.c_uint32, C.c_int
    sig = {
        "amdzk_version": (i32, []),
        "amdzk_build_info": (C.c_char_p, []),
        "amdzk_init": (i32, [i32, C.POINTER(vp)]),
        "amdzk_destroy": (None, [vp]),
        "amdzk_last_error": (C.c_char_p, [vp]),
        "amdzk_set_stream": (i32, [vp, vp]),
        "amdzk_sync": (i32, [vp]),
        "amdzk_ctx_device": (i32, [vp]),
        "amdzk_ctx_check_affinity": (i32, [vp]),
        "amdzk_ptr_check_affinity": (i32, [vp, vp]),
        "amdzk_pk_check_affinity": (i32, [vp, vp]),
        "amdzk_dev_alloc": (i32, [vp, sz, C.POINTER(vp)]),
        "amdzk_dev_free": (i32, [vp, vp]),
        "amdzk_dev_upload": (i32, [vp, vp, vp, sz]),
        "amdzk_dev_download": (i32, [vp, vp, vp, sz]),
        "amdzk_dev_memset": (i32, [vp, vp, i32, sz]),
        "amdzk_host_alloc": (i32, [vp, sz, C.POINTER(vp)]),
        "amdzk_host_free": (i32, [vp, vp]),
        "amdzk_dev_upload_async": (i32, [vp, vp, vp, sz]),
        "amdzk_upload_fence": (i32, [vp]),
        "amdzk_srs_upload": (i32, [vp, vp, vp, u32, C.POINTER(vp)]),
        "amdzk_srs_setup": (i32, [vp, u32, vp, C.POINTER(vp), vp, vp]),
        "amdzk_srs_serialized_size": (sz, [u32]),
        "amdzk_srs_write": (i32, [vp, vp, vp, vp, vp, sz]),
        "amdzk_srs_read": (i32, [vp, vp, sz, C.POINTER(vp), vp, vp]),
        "amdzk_srs_free": (None, [vp, vp]),
        "amdzk_srs_downsize": (i32, [vp, vp, u32, C.POINTER(vp)]),
        "amdzk_srs_get": (i32, [vp, vp, i32, vp]),
        "amdzk_g_to_lagrange": (i32, [vp, vp, u32, vp]),
        "amdzk_msm_g1": (i32, [vp, vp, i32, vp, sz, vp]),
        "amdzk_msm_g1_batch": (i32, [vp, vp, i32, C.POINTER(vp), sz, sz, vp]),
        "amdzk_msm_g1_dev": (i32, [vp, vp, i32, vp, sz, sz, sz, vp]),
        "amdzk_ntt_fr": (i32, [vp, vp, u32, vp, u32]),
        "amdzk_ntt_fr_dev": (i32, [vp, vp, u32, vp, u32, sz, sz]),
        "amdzk_fr_from_raw_dev": (i32, [vp, vp, sz]),
        "amdzk_fr_to_repr_dev": (i32, [vp, vp, sz]),
        "amdzk_domain_new": (i32, [vp, u32, u32, C.POINTER(vp)]),
        "amdzk_domain_free": (None, [vp, vp]),
        "amdzk_domain_k": (u32, [vp]),
        "amdzk_domain_extended_k": (u32, [vp]),
        "amdzk_domain_constant": (i32, [vp, i32, vp]),
        "amdzk_lagrange_to_coeff_dev": (i32, [vp, vp, vp, sz, sz]),
        "amdzk_coeff_to_lagrange_dev": (i32, [vp, vp, vp, sz, sz]),
        "amdzk_coeff_to_extended_dev": (i32, [vp, vp, vp, sz, vp, sz, sz]),
        "amdzk_extended_to_coeff_dev": (i32, [vp, vp, vp, sz, sz]),
        "amdzk_divide_by_vanishing_dev": (i32, [vp, vp, vp, sz, sz]),
        "amdzk_set_host_wait": (i32, [vp, i32]),
        "amdzk_keygen": (i32, [vp, vp, vp, vp, vp, vp, C.POINTER(vp)]),
        "amdzk_keygen_ex": (i32, [vp, vp, vp, vp, vp, vp, u32, C.POINTER(vp)]),
        "amdzk_pk_free": (None, [vp, vp]),
        "amdzk_pk_commitments": (i32, [vp, vp, vp]),
        "amdzk_create_proof": (i32, [vp, vp, C.POINTER(vp), C.POINTER(sz), vp, sz, C.c_uint64, vp, sz, C.POINTER(sz)]),
        "amdzk_create_proof_ex": (i32, [vp, vp, C.POINTER(vp), C.POINTER(sz), vp, sz, C.c_uint64, i32, vp, sz, C.POINTER(sz)]),
        "amdzk_pk_clone_workspace": (i32, [vp, vp, C.POINTER(vp)]),
        "amdzk_create_proof_multi": (i32, [vp, C.POINTER(vp), sz, C.POINTER(C.POINTER(vp)), C.POINTER(C.POINTER(sz)), C.POINTER(vp), sz, C.c_uint64, i32,
                                           vp, sz, C.POINTER(sz)]),
        "amdzk_proof_size_multi": (sz, [vp, sz, i32]),
        "amdzk_proof_random_count": (sz, [vp]),
        "amdzk_proof_size": (sz, [vp, i32]),
        "amdzk_create_proof_scalars": (i32, [vp, vp, C.POINTER(vp), C.POINTER(sz), vp, sz, vp, sz, i32, vp, sz, C.POINTER(sz)]),
        "amdzk_ntt_fr_batch": (i32, [vp, C.POINTER(vp), sz, u32, vp, u32]),
        "amdzk_batch_invert_dev": (i32, [vp, vp, sz]),
        "amdzk_batch_invert_assigned_dev": (i32, [vp, vp, vp, sz, vp]),
        "amdzk_grand_product_dev": (i32, [vp, vp, sz, sz, sz, i32, sz]),
        "amdzk_eval_poly_dev": (i32, [vp, C.POINTER(vp), vp, sz, u32, vp]),
        "amdzk_poly_axpy_dev": (i32, [vp, C.POINTER(vp), vp, sz, vp, sz, i32]),
        "amdzk_kate_div_dev": (i32, [vp, C.POINTER(vp), vp, sz, u32]),
        "amdzk_permute_expression_pair_dev": (i32, [vp, vp, vp, vp, sz, u32, u32]),
        "amdzk_quotient_eval_dev": (i32, [vp, vp, vp, sz, vp, vp, vp, vp, vp]),
        "amdzk_pk_inspect": (i32, [vp, vp, i32, vp, sz, C.POINTER(sz)]),
        "amdzk_debug_limb_program": (i32, [vp, sz, vp, sz, C.POINTER(sz), C.POINTER(u32)]),
        "amdzk_pk_h_program": (i32, [vp, vp, sz, C.POINTER(sz)]),
        "amdzk_timer_start": (i32, [vp]),
        "amdzk_timer_stop": (i32, [vp, C.POINTER(C.c_float)]),
        "amdzk_prof_enable": (i32, [vp, i32]),
        "amdzk_prof_reset": (i32, [vp]),
        "amdzk_prof_get": (i32, [vp, C.c_char_p, C.POINTER(C.c_uint64), C.POINTER(C.c_double)]),
        "amdzk_prof_dump": (sz, [vp, C.c_char_p, sz]),
    }
    for name, (res, args) in sig.items():
        fn = getattr(L, name)
        fn.restype = res
        fn.argtypes = args
    L._amdzk_sig = sig
    _LIB = L
    return L


def build_info():
    """amdzk_build_info(): "amdzk <abi> src=<hash> arch=gfx950" -> {"abi": .., "src": .., "arch": ..}."""
    txt = lib().amdzk_build_info().decode()
    parts = txt.split()
    out = {"text": txt, "abi": int(parts[1])}
    out.update(dict(kv.split("=", 1) for kv in parts[2:]))
    return out


def _ptr(a):
    return a.ctypes.data_as(C.c_void_p)


def as_fr_array(a):
    """(n,4) uint64 C-contiguous view/copy: n field elements, Montgomery limbs."""
    a = np.ascontiguousarray(a, dtype=np.uint64)
    if a.ndim == 1:
        a = a.reshape(-1, 4)
    assert a.shape[-1] == 4
    return a


class DeviceBuffer:
    def __init__(self, ctx, nbytes):
        self.ctx, self.nbytes = ctx, nbytes
        p = C.c_void_p()
        ctx._chk(ctx.L.amdzk_dev_alloc(ctx.h, nbytes, C.byref(p)))
        self.ptr = p

    def upload(self, arr):
        arr = np.ascontiguousarray(arr)
        assert arr.nbytes <= self.nbytes
        self.ctx._chk(self.ctx.L.amdzk_dev_upload(self.ctx.h, self.ptr, _ptr(arr), arr.nbytes))
        return self

    def download(self, shape, dtype=np.uint64):
        out = np.empty(shape, dtype=dtype)
        assert out.nbytes <= self.nbytes
        self.ctx._chk(self.ctx.L.amdzk_dev_download(self.ctx.h, _ptr(out), self.ptr, out.nbytes))
        return out

    def free(self):
        if self.ptr:
            self.ctx.L.amdzk_dev_free(self.ctx.h, self.ptr)
            self.ptr = None


class PinnedBuffer:
    """hipHostMalloc'd staging memory (amdzk_host_alloc), viewed as a numpy array."""

    def __init__(self, ctx, nbytes):
        self.ctx, self.nbytes = ctx, nbytes
        p = C.c_void_p()
        ctx._chk(ctx.L.amdzk_host_alloc(ctx.h, nbytes, C.byref(p)))
        self.ptr = p

    def array(self, shape, dtype=np.uint64):
        n = int(np.prod(shape)) * np.dtype(dtype).itemsize
        assert n <= self.nbytes
        buf = (C.c_uint8 * n).from_address(self.ptr.value)
        return np.frombuffer(buf, dtype=dtype).reshape(shape)

    def free(self):
        if self.ptr:
            self.ctx.L.amdzk_host_free(self.ctx.h, self.ptr)
            self.ptr = None


class Context:
    """amdzk_ctx wrapper. Raises AmdzkError (never falls back to the CPU)."""

    def __init__(self, device_id=0):
        self.L = lib()
        h = C.c_void_p()
        rc = self.L.amdzk_init(device_id, C.byref(h))
        if rc != 0:
            raise AmdzkError(rc, "amdzk_init(%d) failed: no usable gfx950 device" % device_id)
        self.h = h

    def _chk(self, rc):
        if rc != 0:
            raise AmdzkError(rc, self.L.amdzk_last_error(self.h).decode())

    def close(self):
        if self.h:
            self.L.amdzk_destroy(self.h)
            self.h = None

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    def sync(self):
        self._chk(self.L.amdzk_sync(self.h))

    def set_host_wait(self, block):
        """amdzk_set_host_wait: block=True, host waits POLL a completion event (20 us of yielding, then 50-us sleeps: each
        wait ends up to ~50 us late, 8-10 waits per proof); False, they spin in hipStreamSynchronize (the default)."""
        self._chk(self.L.amdzk_set_host_wait(self.h, 1 if block else 0))

    def device(self):
        return self.L.amdzk_ctx_device(self.h)

    def check_affinity(self, pk=None, ptr=None):
        """Raise unless the ctx's stream / workspaces (and `pk`'s buffers, and `ptr`) live on the ctx's device."""
        self._chk(self.L.amdzk_ctx_check_affinity(self.h))
        if pk is not None:
            self._chk(self.L.amdzk_pk_check_affinity(self.h, pk))
        if ptr is not None:
            self._chk(self.L.amdzk_ptr_check_affinity(self.h, C.c_void_p(ptr)))

    def set_stream(self, stream_ptr):
        self._chk(self.L.amdzk_set_stream(self.h, C.c_void_p(stream_ptr)))

    def alloc(self, nbytes):
        return DeviceBuffer(self, nbytes)

    def alloc_pinned(self, nbytes):
        return PinnedBuffer(self, nbytes)

    def upload_async(self, dbuf, host_ptr, nbytes):
        """Start a host->device copy on the ctx's copy stream (host_ptr: address of pinned memory)."""
        self._chk(self.L.amdzk_dev_upload_async(self.h, dbuf.ptr, C.c_void_p(host_ptr), nbytes))

    def upload_async_at(self, dbuf, offset, host_ptr, nbytes):
        """upload_async into dbuf + offset (a witness sent as several copies)."""
        assert offset + nbytes <= dbuf.nbytes
        self._chk(self.L.amdzk_dev_upload_async(self.h, C.c_void_p(dbuf.ptr.value + offset), C.c_void_p(host_ptr), nbytes))

    def upload_fence(self):
        self._chk(self.L.amdzk_upload_fence(self.h))

    # ---- timing hooks
    def timer_start(self):
        self._chk(self.L.amdzk_timer_start(self.h))

    def timer_stop(self):
        ms = C.c_float()
        self._chk(self.L.amdzk_timer_stop(self.h, C.byref(ms)))
        return ms.value

    def prof_enable(self, on=True):
        self._chk(self.L.amdzk_prof_enable(self.h, 1 if on else 0))

    def prof_reset(self):
        self._chk(self.L.amdzk_prof_reset(self.h))

    def prof_dump(self):
        need = self.L.amdzk_prof_dump(self.h, None, 0)
        buf = C.create_string_buffer(need + 16)
        self.L.amdzk_prof_dump(self.h, buf, need + 16)
        out = {}
        for line in buf.value.decode().splitlines():
            name, n, ms = line.split()
            out[name] = (int(n), float(ms))
        return out
