"""CPU: libamdzk.so loads, exports every function include/amdzk.h declares, and refuses to run
without a gfx950 device (no CPU fallback). No compute calls here."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_functions():
    text = open(os.path.join(ROOT, "include", "amdzk.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(amdzk_[a-z0-9_]+)\s*\(", text)))


def test_header_declares_the_hot_path():
    names = declared_functions()
    for must in ("amdzk_init", "amdzk_srs_upload", "amdzk_msm_g1", "amdzk_msm_g1_batch", "amdzk_ntt_fr"):
        assert must in names


def test_library_exports_every_declared_symbol(pkg):
    L = pkg.lib()
    for name in declared_functions():
        assert hasattr(L, name), "libamdzk.so does not export %s" % name
    assert L.amdzk_version() >= 1000


def test_binding_covers_header(pkg):
    L = pkg.lib()
    assert set(declared_functions()) == set(L._amdzk_sig.keys())


def test_no_cpu_fallback(pkg):
    import torch

    if torch.cuda.is_available():
        pytest.skip("a GPU is present; covered by the gpu tests")
    with pytest.raises(pkg.AmdzkError):
        pkg.Context(0)


def test_product_does_not_touch_oracle():
    """No file of the shipped package (Python or C++/HIP) references oracle/ or pyref."""
    pkg_dir = os.path.join(ROOT, "anon-aadhaar-halo2_amd")
    for dirpath, _, files in os.walk(pkg_dir):
        for f in files:
            if f.endswith((".py", ".hip", ".hpp", ".cuh", ".cpp", ".h")) or f == "Makefile":
                text = open(os.path.join(dirpath, f), errors="ignore").read()
                for needle in ("oracle/", "liboracle", "pyref", "bn254_ref", "zkutil"):
                    assert needle not in text, "%s references %s" % (os.path.join(dirpath, f), needle)


def test_library_load_sets_hw_queue_default_without_overriding():
    """libamdzk.so raises the HIP runtime's hardware-queue count (GPU_MAX_HW_QUEUES, 4 by default) when it is loaded —
    several proofs in flight need as many queues as contexts (profiles/r02j_hw_queues_sweep.txt) — and never overrides a
    value the host has chosen. Checked in fresh interpreters (the variable is process state)."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    lib = os.path.join(root, "anon-aadhaar-halo2_amd", "libamdzk.so")
    # os.environ does not see setenv() done by C code: ask the C library
    code = ("import ctypes; ctypes.CDLL(%r); libc = ctypes.CDLL(None); libc.getenv.restype = ctypes.c_char_p; "
            "print((libc.getenv(b'GPU_MAX_HW_QUEUES') or b'').decode())" % lib)
    for preset, want in ((None, "16"), ("6", "6")):
        env = {k: v for k, v in os.environ.items() if k != "GPU_MAX_HW_QUEUES"}
        if preset is not None:
            env["GPU_MAX_HW_QUEUES"] = preset
        out = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=120)
        assert out.returncode == 0, out.stderr
        assert out.stdout.strip() == want, (preset, out.stdout, out.stderr)


def test_integration_md_ffi_block_is_the_headers():
    """INTEGRATION.md's Rust `extern "C"` block is generated from include/amdzk.h (tools/gen_rust_ffi.py): every function
    the header declares appears in it, with the generator's signature, and nothing else does (the
    hand-written block had drifted — 28 of 70 functions missing)."""
    import sys
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import gen_rust_ffi as g

    doc = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    i, j = doc.index(g.BEGIN), doc.index(g.END)
    block = doc[i + len(g.BEGIN):j].strip()
    assert block == "```rust\n" + g.block() + "\n```", "run `python tools/gen_rust_ffi.py --update`"
    names = re.findall(r"pub fn (amdzk_\w+)\(", block)
    assert sorted(names) == declared_functions() and len(names) == len(set(names))
    # spot checks of the C -> Rust type mapping
    assert "pub fn amdzk_ntt_fr_batch(ctx: *mut Ctx, cols: *const *mut u64, ncols: usize, log_n: u32, omega: *const u64, flags: u32) -> c_int;" in block
    assert "pub fn amdzk_last_error(ctx: *const Ctx) -> *const c_char;" in block
    assert "pub fn amdzk_destroy(ctx: *mut Ctx);" in block


def test_integration_md_lists_every_environment_variable_the_library_reads():
    """INTEGRATION.md's table of environment variables against the sources: every AMDZK_* name a getenv / env_u32 call in
    csrc/ reads is documented there, and the table names nothing the library does not read."""
    import re
    csrc = os.path.join(ROOT, "anon-aadhaar-halo2_amd", "csrc")
    used = set()
    for fn in os.listdir(csrc):
        if fn.endswith((".hip", ".hpp", ".cuh")):
            text = open(os.path.join(csrc, fn)).read()
            used |= set(re.findall(r'(?:getenv|env_u32)\(\s*"(AMDZK_[A-Z0-9_]+)"', text))
            for pair in re.findall(r'env_u32\([^"]*\?\s*"(AMDZK_[A-Z0-9_]+)"\s*:\s*"(AMDZK_[A-Z0-9_]+)"', text):  # env_u32(c ? "A" : "B", ...)
                used |= set(pair)
    doc = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    sec = doc[doc.index("## Environment variables the library reads"):doc.index("## Python (ctypes) binding")]
    table = "\n".join(ln for ln in sec.splitlines() if ln.startswith("|"))
    listed = set(re.findall(r"`(AMDZK_[A-Z0-9_]+)", table))
    flags = {"AMDZK_KEYGEN_SERIAL", "AMDZK_KEYGEN_FULL_COSETS"}  # header constants the table's text mentions
    assert used, "no getenv found: the pattern is stale"
    assert used - listed == set(), "undocumented: %s" % sorted(used - listed)
    assert listed - used - flags == set(), "documented but not read: %s" % sorted(listed - used - flags)


def test_library_is_stamped_with_its_kernel_sources(pkg):
    """amdzk_build_info() carries the hash of the comment-stripped kernel sources + Makefile the binary was built from
    (csrc/Makefile -> build_stamp.h, tools/src_hash.py); it is bench.kernel_src_hash() of this tree — which is how bench.py
    refuses a stale libamdzk.so that travelled with a snapshot — and a comment edit does not change it."""
    import bench
    info = pkg.build_info()
    assert info["arch"] == "gfx950" and info["abi"] == pkg.lib().amdzk_version() >= 1001
    assert info["src"] == bench.kernel_src_hash(), "libamdzk.so is stale: rebuild (python -c 'import __graft_entry__ as g; g.build()')"
    assert bench.strip_comments("int a; // x\n/* y */ int  b;") == bench.strip_comments("int a;\nint b; // other comment")

    class FakePkg:
        def build_info(self):
            return {"src": "0123456789abcdef"}
    with pytest.raises(SystemExit):
        bench.check_library_stamp(FakePkg())
