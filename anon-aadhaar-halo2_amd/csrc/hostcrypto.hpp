// Host-side Fiat-Shamir and randomness for the prover driver (control flow that stays on the host:
// SURVEY.md §2.2 U9, Appendix A). Written from the specifications:
//   BLAKE2b  — RFC 7693 (with the 16-byte personalisation parameter block field),
//   ChaCha20 — RFC 8439 block function with a 64-bit block counter, as rand_chacha::ChaCha20Rng,
//   rand_core::SeedableRng::seed_from_u64 — PCG32 seed expansion.
// Mirrors halo2_proofs::transcript::{Blake2bWrite, Challenge255} and halo2curves Fr::random /
// Fr::from_bytes_wide (v2023_01_20 / 0.3.1 [UP]).
#pragma once
#include <stdint.h>
#include <string.h>

#include <vector>

#include "bn254.cuh"

namespace zkhost {

using bn254::Fq;
using bn254::Fr;

// ------------------------------------------------------------------------------ BLAKE2b-512
class Blake2b {
 public:
  explicit Blake2b(const char personal[16]) {
    static const uint64_t iv[8] = {0x6a09e667f3bcc908ULL, 0xbb67ae8584caa73bULL, 0x3c6ef372fe94f82bULL,
                                   0xa54ff53a5f1d36f1ULL, 0x510e527fade682d1ULL, 0x9b05688c2b3e6c1fULL,
                                   0x1f83d9abfb41bd6bULL, 0x5be0cd19137e2179ULL};
    uint8_t param[64];
    memset(param, 0, 64);
    param[0] = 64;  // digest length
    param[2] = 1;   // fanout
    param[3] = 1;   // depth
    memcpy(param + 48, personal, 16);
    for (int i = 0; i < 8; i++) {
      uint64_t p;
      memcpy(&p, param + 8 * i, 8);
      h_[i] = iv[i] ^ p;
    }
    t_[0] = t_[1] = 0;
    buflen_ = 0;
  }
  void update(const void* data, size_t len) {
    const uint8_t* in = (const uint8_t*)data;
    while (len > 0) {
      if (buflen_ == 128) {  // buffer full and more input follows: compress it (not the last block)
        add_counter(128);
        compress(buf_, false);
        buflen_ = 0;
      }
      size_t take = 128 - buflen_;
      if (take > len) take = len;
      memcpy(buf_ + buflen_, in, take);
      buflen_ += take;
      in += take;
      len -= take;
    }
  }
  // Digest of everything absorbed so far; the running state is left untouched (state.clone().finalize()).
  void digest(uint8_t out[64]) const {
    Blake2b c = *this;
    c.add_counter(c.buflen_);
    memset(c.buf_ + c.buflen_, 0, 128 - c.buflen_);
    c.compress(c.buf_, true);
    memcpy(out, c.h_, 64);
  }

 private:
  uint64_t h_[8], t_[2];
  uint8_t buf_[128];
  size_t buflen_;
  void add_counter(uint64_t inc) {
    t_[0] += inc;
    if (t_[0] < inc) t_[1]++;
  }
  static inline uint64_t rotr(uint64_t x, int n) { return (x >> n) | (x << (64 - n)); }
  void compress(const uint8_t block[128], bool last) {
    static const uint64_t iv[8] = {0x6a09e667f3bcc908ULL, 0xbb67ae8584caa73bULL, 0x3c6ef372fe94f82bULL,
                                   0xa54ff53a5f1d36f1ULL, 0x510e527fade682d1ULL, 0x9b05688c2b3e6c1fULL,
                                   0x1f83d9abfb41bd6bULL, 0x5be0cd19137e2179ULL};
    static const uint8_t sigma[12][16] = {
        {0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15}, {14, 10, 4, 8, 9, 15, 13, 6, 1, 12, 0, 2, 11, 7, 5, 3},
        {11, 8, 12, 0, 5, 2, 15, 13, 10, 14, 3, 6, 7, 1, 9, 4}, {7, 9, 3, 1, 13, 12, 11, 14, 2, 6, 5, 10, 4, 0, 15, 8},
        {9, 0, 5, 7, 2, 4, 10, 15, 14, 1, 11, 12, 6, 8, 3, 13}, {2, 12, 6, 10, 0, 11, 8, 3, 4, 13, 7, 5, 15, 14, 1, 9},
        {12, 5, 1, 15, 14, 13, 4, 10, 0, 7, 6, 3, 9, 2, 8, 11}, {13, 11, 7, 14, 12, 1, 3, 9, 5, 0, 15, 4, 8, 6, 2, 10},
        {6, 15, 14, 9, 11, 3, 0, 8, 12, 2, 13, 7, 1, 4, 10, 5}, {10, 2, 8, 4, 7, 6, 1, 5, 15, 11, 9, 14, 3, 12, 13, 0},
        {0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15}, {14, 10, 4, 8, 9, 15, 13, 6, 1, 12, 0, 2, 11, 7, 5, 3}};
    uint64_t m[16], v[16];
    memcpy(m, block, 128);
    for (int i = 0; i < 8; i++) {
      v[i] = h_[i];
      v[i + 8] = iv[i];
    }
    v[12] ^= t_[0];
    v[13] ^= t_[1];
    if (last) v[14] = ~v[14];
#define B2G(a, b, c, d, x, y)        \
  v[a] = v[a] + v[b] + (x);          \
  v[d] = rotr(v[d] ^ v[a], 32);      \
  v[c] = v[c] + v[d];                \
  v[b] = rotr(v[b] ^ v[c], 24);      \
  v[a] = v[a] + v[b] + (y);          \
  v[d] = rotr(v[d] ^ v[a], 16);      \
  v[c] = v[c] + v[d];                \
  v[b] = rotr(v[b] ^ v[c], 63);
    for (int r = 0; r < 12; r++) {
      const uint8_t* s = sigma[r];
      B2G(0, 4, 8, 12, m[s[0]], m[s[1]]);
      B2G(1, 5, 9, 13, m[s[2]], m[s[3]]);
      B2G(2, 6, 10, 14, m[s[4]], m[s[5]]);
      B2G(3, 7, 11, 15, m[s[6]], m[s[7]]);
      B2G(0, 5, 10, 15, m[s[8]], m[s[9]]);
      B2G(1, 6, 11, 12, m[s[10]], m[s[11]]);
      B2G(2, 7, 8, 13, m[s[12]], m[s[13]]);
      B2G(3, 4, 9, 14, m[s[14]], m[s[15]]);
    }
#undef B2G
    for (int i = 0; i < 8; i++) h_[i] ^= v[i] ^ v[i + 8];
  }
};

// ------------------------------------------------------------------------------ field <-> bytes
inline Fr fr_r3() {  // R^3 mod r, for from_u512's high half
  Fr r2 = Fr::r2();
  return bn254::mul(r2, r2);  // (R^2 * R^2) / R = R^3
}
// halo2curves from_u512 / from_bytes_wide: 512-bit little-endian integer mod r, Montgomery form out.
inline Fr fr_from_u512(const uint64_t limbs[8]) {
  Fr d0, d1;
  memcpy(d0.l, limbs, 32);
  memcpy(d1.l, limbs + 4, 32);
  // d0, d1 are arbitrary 256-bit values (possibly >= r). The CIOS loop only needs its FIRST operand
  // (the multiplicand) to be < r; the operand it scans limb by limb may be any 256-bit value
  // (t stays < 2r), so the constants go first.
  return bn254::add(bn254::mul(Fr::r2(), d0), bn254::mul(fr_r3(), d1));
}
inline void fr_to_repr(const Fr& a, uint8_t out[32]) {
  Fr c = bn254::from_mont(a);
  memcpy(out, c.l, 32);
}
inline void fq_to_repr(const Fq& a, uint8_t out[32]) {
  Fq c = bn254::from_mont(a);
  memcpy(out, c.l, 32);
}
// halo2curves 0.3.1 G1Affine::to_bytes [UP recall]: x little-endian; bit 7 of byte 31 = y & 1.
inline void g1_compress(const bn254::G1Affine& p, uint8_t out[32]) {
  if (p.is_inf()) {
    memset(out, 0, 32);
    return;
  }
  uint8_t yb[32];
  fq_to_repr(p.x, out);
  fq_to_repr(p.y, yb);
  out[31] |= (uint8_t)((yb[0] & 1) << 7);
}

// ------------------------------------------------------------------------------ transcript
// halo2_proofs::transcript::TranscriptWrite, as far as the prover uses it.
class TranscriptWrite {
 public:
  virtual ~TranscriptWrite() {}
  virtual Fr squeeze_challenge() = 0;
  virtual bool common_point(const bn254::G1Affine& p) = 0;
  virtual void common_scalar(const Fr& s) = 0;
  virtual bool write_point(const bn254::G1Affine& p) = 0;
  virtual void write_scalar(const Fr& s) = 0;
  std::vector<uint8_t> proof;
};

class Blake2bWrite : public TranscriptWrite {
 public:
  Blake2bWrite() : st_("Halo2-Transcript") {}
  Fr squeeze_challenge() override {
    uint8_t z = 0;  // BLAKE2B_PREFIX_CHALLENGE
    st_.update(&z, 1);
    uint8_t d[64];
    st_.digest(d);
    uint64_t l[8];
    memcpy(l, d, 64);
    return fr_from_u512(l);
  }
  bool common_point(const bn254::G1Affine& p) override {
    if (p.is_inf()) return false;  // "cannot write points at infinity to the transcript"
    uint8_t one = 1, b[32];
    st_.update(&one, 1);
    fq_to_repr(p.x, b);
    st_.update(b, 32);
    fq_to_repr(p.y, b);
    st_.update(b, 32);
    return true;
  }
  void common_scalar(const Fr& s) override {
    uint8_t two = 2, b[32];
    st_.update(&two, 1);
    fr_to_repr(s, b);
    st_.update(b, 32);
  }
  bool write_point(const bn254::G1Affine& p) override {
    if (!common_point(p)) return false;
    uint8_t b[32];
    g1_compress(p, b);
    proof.insert(proof.end(), b, b + 32);
    return true;
  }
  void write_scalar(const Fr& s) override {
    common_scalar(s);
    uint8_t b[32];
    fr_to_repr(s, b);
    proof.insert(proof.end(), b, b + 32);
  }

 private:
  Blake2b st_;
};

// ------------------------------------------------------------------------------ Keccak-256 (EVM)
// Keccak-f[1600] sponge, rate 136, pad10*1 with domain byte 0x01 (the EVM's keccak256, NOT SHA3-256).
inline void keccak256(const uint8_t* data, size_t len, uint8_t out[32]) {
  static const uint64_t RC[24] = {0x0000000000000001ULL, 0x0000000000008082ULL, 0x800000000000808aULL, 0x8000000080008000ULL,
                                  0x000000000000808bULL, 0x0000000080000001ULL, 0x8000000080008081ULL, 0x8000000000008009ULL,
                                  0x000000000000008aULL, 0x0000000000000088ULL, 0x0000000080008009ULL, 0x000000008000000aULL,
                                  0x000000008000808bULL, 0x800000000000008bULL, 0x8000000000008089ULL, 0x8000000000008003ULL,
                                  0x8000000000008002ULL, 0x8000000000000080ULL, 0x000000000000800aULL, 0x800000008000000aULL,
                                  0x8000000080008081ULL, 0x8000000000008080ULL, 0x0000000080000001ULL, 0x8000000080008008ULL};
  static const int ROT[25] = {0, 1, 62, 28, 27, 36, 44, 6, 55, 20, 3, 10, 43, 25, 39, 41, 45, 15, 21, 8, 18, 2, 61, 56, 14};
  uint64_t st[25];
  memset(st, 0, sizeof(st));
  auto permute = [&]() {
    for (int round = 0; round < 24; round++) {
      uint64_t c[5], d[5], b[25];
      for (int x = 0; x < 5; x++) c[x] = st[x] ^ st[x + 5] ^ st[x + 10] ^ st[x + 15] ^ st[x + 20];
      for (int x = 0; x < 5; x++) d[x] = c[(x + 4) % 5] ^ ((c[(x + 1) % 5] << 1) | (c[(x + 1) % 5] >> 63));
      for (int i = 0; i < 25; i++) st[i] ^= d[i % 5];
      for (int x = 0; x < 5; x++)
        for (int y = 0; y < 5; y++) {
          int i = x + 5 * y, r = ROT[i];
          uint64_t v = st[i];
          b[y + 5 * ((2 * x + 3 * y) % 5)] = r ? ((v << r) | (v >> (64 - r))) : v;
        }
      for (int y = 0; y < 5; y++)
        for (int x = 0; x < 5; x++) st[x + 5 * y] = b[x + 5 * y] ^ (~b[(x + 1) % 5 + 5 * y] & b[(x + 2) % 5 + 5 * y]);
      st[0] ^= RC[round];
    }
  };
  const size_t rate = 136;
  size_t off = 0;
  while (len - off >= rate) {
    for (size_t i = 0; i < rate / 8; i++) {
      uint64_t w;
      memcpy(&w, data + off + 8 * i, 8);
      st[i] ^= w;
    }
    permute();
    off += rate;
  }
  uint8_t last[136];
  memset(last, 0, rate);
  memcpy(last, data + off, len - off);
  last[len - off] ^= 0x01;
  last[rate - 1] ^= 0x80;
  for (size_t i = 0; i < rate / 8; i++) {
    uint64_t w;
    memcpy(&w, last + 8 * i, 8);
    st[i] ^= w;
  }
  permute();
  memcpy(out, st, 32);
}

// 256-bit big-endian integer mod r -> Montgomery Fr (ChallengeEvm::new / u256_to_fe)
inline Fr fr_from_be32(const uint8_t be[32]) {
  uint64_t l[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  for (int i = 0; i < 32; i++) l[i / 8] |= (uint64_t)be[31 - i] << (8 * (i % 8));
  return fr_from_u512(l);
}

// halo2-solidity-verifier's Keccak256Transcript + ChallengeEvm, the transcript the reference's
// solidity_verifier_contract/contract.sol re-derives (:89-112 squeeze_challenge / _cont, :77-87 points):
// a running byte buffer of 32-byte big-endian words; challenge = keccak256(buffer) mod r; the hash
// becomes the new buffer, and a squeeze with nothing absorbed since appends the byte 0x01.
class Keccak256Write : public TranscriptWrite {
 public:
  Fr squeeze_challenge() override {
    std::vector<uint8_t> data(buf_);
    if (data.size() == 32) data.push_back(1);
    uint8_t h[32];
    keccak256(data.data(), data.size(), h);
    buf_.assign(h, h + 32);
    return fr_from_be32(h);
  }
  bool common_point(const bn254::G1Affine& p) override {
    if (p.is_inf()) return false;
    uint8_t b[32];
    fq_to_repr(p.x, b);
    for (int i = 31; i >= 0; i--) buf_.push_back(b[i]);
    fq_to_repr(p.y, b);
    for (int i = 31; i >= 0; i--) buf_.push_back(b[i]);
    return true;
  }
  void common_scalar(const Fr& s) override {
    uint8_t b[32];
    fr_to_repr(s, b);
    for (int i = 31; i >= 0; i--) buf_.push_back(b[i]);
  }
  bool write_point(const bn254::G1Affine& p) override {
    if (!common_point(p)) return false;
    uint8_t b[32];
    fq_to_repr(p.x, b);
    for (int i = 31; i >= 0; i--) proof.push_back(b[i]);
    fq_to_repr(p.y, b);
    for (int i = 31; i >= 0; i--) proof.push_back(b[i]);
    return true;
  }
  void write_scalar(const Fr& s) override {
    common_scalar(s);
    uint8_t b[32];
    fr_to_repr(s, b);
    for (int i = 31; i >= 0; i--) proof.push_back(b[i]);
  }

 private:
  std::vector<uint8_t> buf_;
};

// ------------------------------------------------------------------------------ ChaCha20Rng
class ChaCha20Rng {
 public:
  explicit ChaCha20Rng(uint64_t seed) {
    uint64_t state = seed;
    for (int i = 0; i < 8; i++) {  // rand_core seed_from_u64: PCG32 stream into the 32-byte key
      state = state * 6364136223846793005ULL + 11634580027462260723ULL;
      uint32_t xorshifted = (uint32_t)(((state >> 18) ^ state) >> 27);
      uint32_t rot = (uint32_t)(state >> 59);
      key_[i] = (xorshifted >> rot) | (xorshifted << ((32 - rot) & 31));
    }
    counter_ = 0;
    pos_ = 16;
  }
  uint64_t next_u64() {
    uint32_t lo = next_u32(), hi = next_u32();
    return (uint64_t)lo | ((uint64_t)hi << 32);
  }
  Fr fr() {  // Fr::random
    uint64_t l[8];
    for (int i = 0; i < 8; i++) l[i] = next_u64();
    return fr_from_u512(l);
  }
  // Every draw this prover makes is an Fr::random = 8 x next_u64 = exactly one 64-byte ChaCha block, so between draws
  // the stream sits on a block boundary: draw number j is block j of the key stream. That lets a run of draws (the
  // n coefficients of vanishing::prover's random polynomial) be produced on the device, block j0 + i by thread i.
  bool at_block_boundary() const { return pos_ == 16; }
  uint64_t block_counter() const { return counter_; }
  const uint32_t* key() const { return key_; }
  void skip_blocks(uint64_t nblocks) { counter_ += nblocks; }

 private:
  uint32_t key_[8], buf_[16];
  uint64_t counter_;
  int pos_;
  static inline uint32_t rotl(uint32_t x, int n) { return (x << n) | (x >> (32 - n)); }
  uint32_t next_u32() {
    if (pos_ == 16) {
      block();
      pos_ = 0;
    }
    return buf_[pos_++];
  }
  void block() {
    uint32_t c[16] = {0x61707865, 0x3320646e, 0x79622d32, 0x6b206574};
    for (int i = 0; i < 8; i++) c[4 + i] = key_[i];
    c[12] = (uint32_t)counter_;
    c[13] = (uint32_t)(counter_ >> 32);
    c[14] = c[15] = 0;
    uint32_t x[16];
    memcpy(x, c, 64);
#define CQR(a, b, cc, d)                 \
  x[a] += x[b]; x[d] = rotl(x[d] ^ x[a], 16); \
  x[cc] += x[d]; x[b] = rotl(x[b] ^ x[cc], 12); \
  x[a] += x[b]; x[d] = rotl(x[d] ^ x[a], 8);  \
  x[cc] += x[d]; x[b] = rotl(x[b] ^ x[cc], 7);
    for (int r = 0; r < 10; r++) {
      CQR(0, 4, 8, 12) CQR(1, 5, 9, 13) CQR(2, 6, 10, 14) CQR(3, 7, 11, 15)
      CQR(0, 5, 10, 15) CQR(1, 6, 11, 12) CQR(2, 7, 8, 13) CQR(3, 4, 9, 14)
    }
#undef CQR
    for (int i = 0; i < 16; i++) buf_[i] = x[i] + c[i];
    counter_++;
  }
};

}  // namespace zkhost
