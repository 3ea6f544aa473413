// Known-answer material for the hashes and the RNG the product's transcripts rest on (csrc/hostcrypto.hpp, host code
// of libamdzk): prints "<name> <hex>" lines that tests/test_hash_kats.py compares with the standard digests
// (Keccak-256 of "" and "abc", the all-zero-key ChaCha20 block), with hashlib (BLAKE2b-512 personalised
// "Halo2-Transcript") and with the oracle's transcripts. Compiled with g++ by the test; no GPU.
#include <stdio.h>
#include <string.h>

#include <string>
#include <vector>

#include "../../anon-aadhaar-halo2_amd/csrc/bn254.cuh"
#include "../../anon-aadhaar-halo2_amd/csrc/hostcrypto.hpp"

static void hex(const char* name, const uint8_t* p, size_t n) {
  printf("%s ", name);
  for (size_t i = 0; i < n; i++) printf("%02x", p[i]);
  printf("\n");
}
static bn254::Fr fr_small(uint64_t v) {
  bn254::Fr r = bn254::Fr::zero();
  memcpy(r.l, &v, 8);
  return bn254::to_mont(r);
}

int main() {
  uint8_t d[64];
  std::vector<uint8_t> m200(200), m300(300);
  for (size_t i = 0; i < 200; i++) m200[i] = (uint8_t)i;
  for (size_t i = 0; i < 300; i++) m300[i] = (uint8_t)(7 * i + 1);
  zkhost::keccak256((const uint8_t*)"", 0, d);
  hex("keccak256_empty", d, 32);
  zkhost::keccak256((const uint8_t*)"abc", 3, d);
  hex("keccak256_abc", d, 32);
  zkhost::keccak256(m200.data(), 135, d);  // one byte short of the rate: padding 0x01 and 0x80 in different bytes
  hex("keccak256_135", d, 32);
  zkhost::keccak256(m200.data(), 136, d);  // exactly the rate: a whole padding block
  hex("keccak256_136", d, 32);
  zkhost::keccak256(m200.data(), 200, d);
  hex("keccak256_200", d, 32);
  {
    zkhost::Blake2b b("Halo2-Transcript");
    b.digest(d);
    hex("blake2b_empty", d, 64);
    b.update("abc", 3);
    b.digest(d);  // state.clone().finalize(): the running state goes on
    hex("blake2b_abc", d, 64);
    b.update(m300.data(), 300);
    b.digest(d);
    hex("blake2b_abc_plus_300", d, 64);
  }
  {
    zkhost::Blake2b b("Halo2-Transcript");
    b.update(m300.data(), 128);  // exactly one block: must not be compressed as a non-final block before the digest
    b.digest(d);
    hex("blake2b_128", d, 64);
    b.update(m300.data() + 128, 128);
    b.digest(d);
    hex("blake2b_256", d, 64);
  }
  {  // one-scalar transcripts
    uint8_t r[32];
    zkhost::Keccak256Write k;
    k.common_scalar(fr_small(5));
    zkhost::fr_to_repr(k.squeeze_challenge(), r);
    hex("keccak_transcript_c1", r, 32);
    zkhost::fr_to_repr(k.squeeze_challenge(), r);  // nothing absorbed since: the 0x01 byte path
    hex("keccak_transcript_c2", r, 32);
    k.write_scalar(fr_small(77));
    zkhost::fr_to_repr(k.squeeze_challenge(), r);
    hex("keccak_transcript_c3", r, 32);
    hex("keccak_transcript_proof", k.proof.data(), k.proof.size());
    zkhost::Blake2bWrite b;
    b.common_scalar(fr_small(5));
    zkhost::fr_to_repr(b.squeeze_challenge(), r);
    hex("blake2b_transcript_c1", r, 32);
    zkhost::fr_to_repr(b.squeeze_challenge(), r);
    hex("blake2b_transcript_c2", r, 32);
    b.write_scalar(fr_small(77));
    zkhost::fr_to_repr(b.squeeze_challenge(), r);
    hex("blake2b_transcript_c3", r, 32);
    hex("blake2b_transcript_proof", b.proof.data(), b.proof.size());
  }
  {  // ChaCha20Rng::seed_from_u64: the expanded key and the first draws
    zkhost::ChaCha20Rng g(0);
    hex("chacha_seed0_key", (const uint8_t*)g.key(), 32);
    uint64_t w[8];
    for (int i = 0; i < 8; i++) w[i] = g.next_u64();
    hex("chacha_seed0_block0", (const uint8_t*)w, 64);
    zkhost::ChaCha20Rng g2(0x0123456789abcdefULL);
    uint8_t r[32];
    zkhost::fr_to_repr(g2.fr(), r);
    hex("chacha_seedX_fr0", r, 32);
    zkhost::fr_to_repr(g2.fr(), r);
    hex("chacha_seedX_fr1", r, 32);
  }
  return 0;
}
