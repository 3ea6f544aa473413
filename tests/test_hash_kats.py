"""Known-answer tests for the hashes and the RNG both provers' transcripts rest on: they shrink the
"parity unpinned" surface to what really needs the Rust prover. CPU only.

  * Keccak-256 (the EVM's: pad byte 0x01): the standard digests of b"" and b"abc", for the oracle
    (oracle/plonk_ref.keccak256) and for the product (csrc/hostcrypto.hpp, compiled for the host);
  * the Keccak-f[1600] permutation and the sponge: with SHA-3's pad byte (0x06) the oracle's code must give
    hashlib.sha3_256 on messages around the 136-byte rate;
  * BLAKE2b-512 personalised "Halo2-Transcript": the product's own implementation against hashlib (incl. the
    digest-of-a-running-state use the transcript makes of it, and inputs of exactly one and two blocks);
  * ChaCha20: the all-zero-key block 0 (RFC 7539 A.1 #1 / rand_chacha's `test_chacha_true_values_a`) for the oracle's
    and the product's block function; seed_from_u64's PCG32 key expansion, product == oracle;
  * one-scalar transcripts: the product's Blake2bWrite / Keccak256Write against the oracle's."""
import hashlib
import os
import subprocess
import sys
import tempfile

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import plonk_ref as PR  # noqa: E402

KECCAK_EMPTY = "c5d2460186f7233c927e7db2dcc703c0e500b653ca82273b7bfad8045d85a470"
KECCAK_ABC = "4e03657aea45a94fc7d47ba826c8d667c0d1e6e33a64a036ec44f58fa12d6c45"
CHACHA_ZERO_KEY_BLOCK0 = ("76b8e0ada0f13d90405d6ae55386bd28bdd219b8a08ded1aa836efcc8b770dc7"
                          "da41597c5157488d7724e03fb8d84a376a43b8f41518a11cc387b669b2ee6586")
M200 = bytes(range(200))
M300 = bytes((7 * i + 1) & 0xFF for i in range(300))


@pytest.fixture(scope="module")
def product():
    src = os.path.join(ROOT, "tests", "native", "hostcrypto_kat.cpp")
    with tempfile.TemporaryDirectory() as d:
        exe = os.path.join(d, "kat")
        subprocess.check_call(["g++", "-O1", "-std=c++17", "-o", exe, src])
        out = subprocess.check_output([exe], text=True)
    return dict(line.split() if len(line.split()) == 2 else (line.split()[0], "") for line in out.strip().splitlines())


def test_oracle_keccak256_standard_digests():
    assert PR.keccak256(b"").hex() == KECCAK_EMPTY
    assert PR.keccak256(b"abc").hex() == KECCAK_ABC


def test_oracle_keccak_permutation_is_sha3s(monkeypatch):
    """Same sponge, SHA-3's domain byte: the oracle's Keccak-f and absorption equal hashlib.sha3_256."""
    import inspect
    src = inspect.getsource(PR.keccak256)
    assert "data.append(0x01)" in src
    ns = {}
    exec(src.replace("data.append(0x01)", "data.append(0x06)").replace("def keccak256", "def sha3_256"), ns)  # noqa: S102 - our own source
    for msg in (b"", b"abc", M200[:135], M200[:136], M200[:137], M200, M300):
        assert ns["sha3_256"](msg) == hashlib.sha3_256(msg).digest()


def test_product_keccak256(product):
    assert product["keccak256_empty"] == KECCAK_EMPTY
    assert product["keccak256_abc"] == KECCAK_ABC
    for name, msg in (("keccak256_135", M200[:135]), ("keccak256_136", M200[:136]), ("keccak256_200", M200)):
        assert product[name] == PR.keccak256(msg).hex()


def test_product_blake2b_equals_hashlib(product):
    def h(msg):
        return hashlib.blake2b(msg, digest_size=64, person=b"Halo2-Transcript").hexdigest()
    assert product["blake2b_empty"] == h(b"")
    assert product["blake2b_abc"] == h(b"abc")
    assert product["blake2b_abc_plus_300"] == h(b"abc" + M300)
    assert product["blake2b_128"] == h(M300[:128])
    assert product["blake2b_256"] == h(M300[:256])


def test_chacha20_zero_key_block_and_seed_expansion(product):
    g = PR.ChaCha20Rng(0)
    g.key = [0] * 8
    blk = b"".join(w.to_bytes(4, "little") for w in g._block())
    assert blk.hex() == CHACHA_ZERO_KEY_BLOCK0
    g0 = PR.ChaCha20Rng(0)
    assert product["chacha_seed0_key"] == b"".join(w.to_bytes(4, "little") for w in g0.key).hex()
    assert product["chacha_seed0_block0"] == b"".join(g0.next_u64().to_bytes(8, "little") for _ in range(8)).hex()
    gx = PR.ChaCha20Rng(0x0123456789ABCDEF)
    assert product["chacha_seedX_fr0"] == PR.fr_repr(gx.fr()).hex()
    assert product["chacha_seedX_fr1"] == PR.fr_repr(gx.fr()).hex()


def test_one_scalar_transcripts(product):
    for name, cls in (("keccak", PR.Keccak256Write), ("blake2b", PR.Blake2bWrite)):
        t = cls()
        t.common_scalar(5)
        assert product[name + "_transcript_c1"] == PR.fr_repr(t.squeeze_challenge()).hex()
        assert product[name + "_transcript_c2"] == PR.fr_repr(t.squeeze_challenge()).hex()
        t.write_scalar(77)
        assert product[name + "_transcript_c3"] == PR.fr_repr(t.squeeze_challenge()).hex()
        assert product[name + "_transcript_proof"] == bytes(t.proof).hex()
    # the EVM transcript's first challenge from first principles: keccak256 of the 32-byte big-endian scalar followed
    # by the byte 0x01 (a buffer of exactly 32 bytes gets it: /root/reference/solidity_verifier_contract/contract.sol:89-112)
    assert product["keccak_transcript_c1"] == PR.fr_repr(int.from_bytes(PR.keccak256((5).to_bytes(32, "big") + b"\x01"), "big") % PR.R).hex()
