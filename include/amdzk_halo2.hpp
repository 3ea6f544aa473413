// amdzk_halo2.hpp — host-side mirror, in C++, of the halo2_proofs interface the reference's circuits are
// written against and its prover is called through (halo2_proofs 0.2.0 @ v2023_01_20 [UP],
// /root/reference/Cargo.lock:469-471):
//
//   plonk::{Column, Expression, VirtualCells, ConstraintSystem}   <- Circuit::configure(meta)
//        /root/reference/src/lib.rs:295-326, src/signal.rs:27-49, src/conditional_secrets.rs:81-187,
//        src/timestamp.rs:58-138
//   plonk::permutation::keygen::Assembly                           <- region.constrain_equal / copy_advice
//   poly::kzg::commitment::ParamsKZG {setup, read, write, downsize, commit, commit_lagrange, get_g}
//   plonk::{keygen_pk -> ProvingKey, create_proof}
//
// Header-only over the C ABI of include/amdzk.h (link libamdzk.so); the same names, argument meaning and
// derived quantities (query order, degree(), blinding_factors()) as upstream. Everything O(n) runs on
// the MI355X; this file only describes circuits and moves bytes. The Python package
// (anon-aadhaar-halo2_amd/halo2/*.py) is the same mirror for the test-suite;
// tests/test_cpp_mirror.py checks that both flatten a circuit to identical C-ABI arrays and that a
// proof made through this header equals the oracle prover's bytes.
#ifndef AMDZK_HALO2_HPP
#define AMDZK_HALO2_HPP

#include <cstdint>
#include <cstring>
#include <functional>
#include <map>
#include <memory>
#include <stdexcept>
#include <string>
#include <utility>
#include <vector>

#include "amdzk.h"

namespace amdzk {
namespace halo2 {

// ------------------------------------------------------------------------------------------ errors
struct Error : std::runtime_error {
  int code;
  Error(int c, const std::string& m) : std::runtime_error(m), code(c) {}
};

class Context {
 public:
  explicit Context(int device = 0) {
    int rc = amdzk_init(device, &h_);
    if (rc != AMDZK_OK) throw Error(rc, "amdzk_init failed");
  }
  ~Context() {
    if (h_) amdzk_destroy(h_);
  }
  Context(const Context&) = delete;
  Context& operator=(const Context&) = delete;
  amdzk_ctx* get() const { return h_; }
  void check(int rc) const {
    if (rc != AMDZK_OK) throw Error(rc, amdzk_last_error(h_));
  }
  // amdzk_set_host_wait: true = this context's calls POLL a completion event while they wait for the device (20 us of
  // yielding, then 50-us sleeps: every wait ends up to ~50 us late, 8-10 waits per proof; for hosts with more proofs in
  // flight than cores to spare), false = they spin in hipStreamSynchronize (the default: lowest latency)
  void set_blocking_waits(bool block) const { check(amdzk_set_host_wait(h_, block ? AMDZK_WAIT_BLOCK : AMDZK_WAIT_SPIN)); }

 private:
  amdzk_ctx* h_ = nullptr;
};

// ------------------------------------------------------------------------------------------ Fr
// halo2curves::bn256::Fr: 4 x u64 little-endian limbs, Montgomery form with R = 2^256. Host arithmetic is
// only used for circuit constants and test witnesses.
struct Fr {
  uint64_t l[4];

  static constexpr uint64_t MOD[4] = {0x43e1f593f0000001ULL, 0x2833e84879b97091ULL, 0xb85045b68181585dULL, 0x30644e72e131a029ULL};
  static constexpr uint64_t INV = 0xc2e1f593efffffffULL;  // -r^-1 mod 2^64
  static constexpr uint64_t R2[4] = {0x1bb8e645ae216da7ULL, 0x53fe3ab1e35c59e3ULL, 0x8c49833d53bb8085ULL, 0x0216d0b17f4e44a5ULL};

  static Fr zero() { return Fr{{0, 0, 0, 0}}; }
  static Fr one() { return from_u64(1); }
  // Fr::from_raw: canonical little-endian limbs (< r) -> Montgomery.
  static Fr from_raw(const uint64_t v[4]) {
    Fr a{{v[0], v[1], v[2], v[3]}}, r2{{R2[0], R2[1], R2[2], R2[3]}};
    return mont_mul(a, r2);
  }
  static Fr from_u64(uint64_t v) {
    uint64_t raw[4] = {v, 0, 0, 0};
    return from_raw(raw);
  }
  // Big-endian hex digits of the canonical value (no 0x prefix), as halo2's Debug prints them.
  static Fr from_hex(const std::string& hex) {
    uint64_t raw[4] = {0, 0, 0, 0};
    int bit = 0;
    for (size_t i = hex.size(); i-- > 0;) {
      char c = hex[i];
      uint64_t d = c >= '0' && c <= '9' ? c - '0' : c >= 'a' && c <= 'f' ? c - 'a' + 10 : c >= 'A' && c <= 'F' ? c - 'A' + 10 : 16;
      if (d > 15 || bit >= 256) throw Error(AMDZK_E_INVALID, "Fr::from_hex: bad digit or too long");
      raw[bit >> 6] |= d << (bit & 63);
      bit += 4;
    }
    return from_raw(raw);
  }
  // Fr::to_repr: canonical limbs.
  void to_repr(uint64_t out[4]) const {
    Fr o{{1, 0, 0, 0}};
    Fr c = mont_mul(*this, o);
    std::memcpy(out, c.l, 32);
  }
  bool is_zero() const { return (l[0] | l[1] | l[2] | l[3]) == 0; }
  bool operator==(const Fr& o) const { return std::memcmp(l, o.l, 32) == 0; }
  bool operator<(const Fr& o) const {  // ordering for std::map keys only
    for (int i = 3; i >= 0; i--)
      if (l[i] != o.l[i]) return l[i] < o.l[i];
    return false;
  }
  Fr operator+(const Fr& o) const {
    Fr r;
    unsigned __int128 c = 0;
    for (int i = 0; i < 4; i++) {
      c += (unsigned __int128)l[i] + o.l[i];
      r.l[i] = (uint64_t)c;
      c >>= 64;
    }
    return (c || geq_mod(r)) ? sub_mod(r) : r;
  }
  Fr operator-() const {
    if (is_zero()) return *this;
    Fr m{{MOD[0], MOD[1], MOD[2], MOD[3]}}, r;
    unsigned __int128 b = 0;
    for (int i = 0; i < 4; i++) {
      unsigned __int128 d = (unsigned __int128)m.l[i] - l[i] - (uint64_t)b;
      r.l[i] = (uint64_t)d;
      b = (d >> 64) & 1;
    }
    return r;
  }
  Fr operator-(const Fr& o) const { return *this + (-o); }
  Fr operator*(const Fr& o) const { return mont_mul(*this, o); }

 private:
  static bool geq_mod(const Fr& a) {
    for (int i = 3; i >= 0; i--)
      if (a.l[i] != MOD[i]) return a.l[i] > MOD[i];
    return true;
  }
  static Fr sub_mod(const Fr& a) {
    Fr r;
    unsigned __int128 b = 0;
    for (int i = 0; i < 4; i++) {
      unsigned __int128 d = (unsigned __int128)a.l[i] - MOD[i] - (uint64_t)b;
      r.l[i] = (uint64_t)d;
      b = (d >> 64) & 1;
    }
    return r;
  }
  static Fr mont_mul(const Fr& a, const Fr& b) {  // CIOS, 4 x 64
    uint64_t t[6] = {0, 0, 0, 0, 0, 0};
    for (int i = 0; i < 4; i++) {
      unsigned __int128 c = 0;
      for (int j = 0; j < 4; j++) {
        c += (unsigned __int128)a.l[j] * b.l[i] + t[j];
        t[j] = (uint64_t)c;
        c >>= 64;
      }
      c += t[4];
      t[4] = (uint64_t)c;
      t[5] = (uint64_t)(c >> 64);
      uint64_t m = t[0] * INV;
      c = (unsigned __int128)m * MOD[0] + t[0];
      c >>= 64;
      for (int j = 1; j < 4; j++) {
        c += (unsigned __int128)m * MOD[j] + t[j];
        t[j - 1] = (uint64_t)c;
        c >>= 64;
      }
      c += t[4];
      t[3] = (uint64_t)c;
      t[4] = t[5] + (uint64_t)(c >> 64);
    }
    Fr r{{t[0], t[1], t[2], t[3]}};
    return (t[4] || geq_mod(r)) ? sub_mod(r) : r;
  }
};

// ------------------------------------------------------------------------------------------ columns
enum class Any : uint32_t { Advice = 0, Fixed = 1, Instance = 2 };  // upstream's ordering of `Any`

struct Column {
  Any kind;
  uint32_t index;
  bool operator==(const Column& o) const { return kind == o.kind && index == o.index; }
};
struct Rotation {
  int32_t v;
  static Rotation cur() { return {0}; }
  static Rotation next() { return {1}; }
  static Rotation prev() { return {-1}; }
};

// ------------------------------------------------------------------------------------------ Expression
// plonk::Expression: Constant | Fixed | Advice | Instance | Negated | Sum | Product | Scaled.
class Expression {
 public:
  enum Op : uint32_t { Constant = 1, Fixed = 2, Advice = 3, Instance = 4, Negated = 5, Sum = 6, Product = 7, Scaled = 8 };
  struct Node {
    Op op;
    Fr c;               // Constant / Scaled
    uint32_t column;    // queries
    int32_t rotation;
    std::shared_ptr<const Node> a, b;
  };

  Expression() = default;
  static Expression constant(const Fr& v) { return Expression(mk(Constant, v, 0, 0, nullptr, nullptr)); }
  static Expression query(Any kind, uint32_t column, int32_t rot) {
    Op op = kind == Any::Advice ? Advice : kind == Any::Fixed ? Fixed : Instance;
    return Expression(mk(op, Fr::zero(), column, rot, nullptr, nullptr));
  }
  Expression operator-() const { return Expression(mk(Negated, Fr::zero(), 0, 0, n_, nullptr)); }
  Expression operator+(const Expression& o) const { return Expression(mk(Sum, Fr::zero(), 0, 0, n_, o.n_)); }
  Expression operator-(const Expression& o) const { return *this + (-o); }  // upstream: Sum(a, Negated(b))
  Expression operator*(const Expression& o) const { return Expression(mk(Product, Fr::zero(), 0, 0, n_, o.n_)); }
  Expression operator*(const Fr& s) const { return Expression(mk(Scaled, s, 0, 0, n_, nullptr)); }

  uint32_t degree() const { return degree(n_.get()); }
  const Node* node() const { return n_.get(); }

 private:
  explicit Expression(std::shared_ptr<const Node> n) : n_(std::move(n)) {}
  static std::shared_ptr<const Node> mk(Op op, const Fr& c, uint32_t col, int32_t rot, std::shared_ptr<const Node> a,
                                        std::shared_ptr<const Node> b) {
    auto n = std::make_shared<Node>();
    n->op = op;
    n->c = c;
    n->column = col;
    n->rotation = rot;
    n->a = std::move(a);
    n->b = std::move(b);
    return n;
  }
  static uint32_t degree(const Node* n) {
    switch (n->op) {
      case Constant: return 0;
      case Fixed: case Advice: case Instance: return 1;
      case Negated: case Scaled: return degree(n->a.get());
      case Sum: return std::max(degree(n->a.get()), degree(n->b.get()));
      case Product: return degree(n->a.get()) + degree(n->b.get());
    }
    return 0;
  }
  std::shared_ptr<const Node> n_;
};

class ConstraintSystem;

// What the closures of create_gate / lookup receive (`meta.query_advice(col, Rotation::cur())`).
class VirtualCells {
 public:
  explicit VirtualCells(ConstraintSystem& cs) : cs_(cs) {}
  Expression query_advice(Column c, Rotation r);
  Expression query_fixed(Column c, Rotation r = Rotation::cur());
  Expression query_instance(Column c, Rotation r);
  Expression query_selector(Column s) { return query_fixed(s, Rotation::cur()); }

 private:
  ConstraintSystem& cs_;
};

// plonk::ConstraintSystem as it stands after Circuit::configure and selector compression (selectors
// are fixed columns here, one each).
class ConstraintSystem {
 public:
  using Query = std::pair<Column, int32_t>;
  struct Lookup {
    std::vector<Expression> inputs, tables;
  };

  uint32_t num_fixed = 0, num_advice = 0, num_instance = 0;
  std::vector<Query> advice_queries, fixed_queries, instance_queries;  // first-use order
  std::vector<uint32_t> num_advice_queries;                            // per advice column
  std::vector<Expression> gates;                                       // one entry per polynomial, in order
  std::vector<Lookup> lookups;
  std::vector<Column> permutation_columns;
  uint32_t minimum_degree = 0;

  Column advice_column() {
    num_advice_queries.push_back(0);
    return {Any::Advice, num_advice++};
  }
  Column fixed_column() { return {Any::Fixed, num_fixed++}; }
  Column selector() { return fixed_column(); }
  Column instance_column() { return {Any::Instance, num_instance++}; }

  // ConstraintSystem::enable_equality: query at Rotation::cur() and add to the permutation argument.
  void enable_equality(Column c) {
    VirtualCells m(*this);
    if (c.kind == Any::Advice) m.query_advice(c, Rotation::cur());
    else if (c.kind == Any::Fixed) m.query_fixed(c, Rotation::cur());
    else m.query_instance(c, Rotation::cur());
    for (const Column& p : permutation_columns)
      if (p == c) return;
    permutation_columns.push_back(c);
  }
  void create_gate(const char* /*name*/, const std::function<std::vector<Expression>(VirtualCells&)>& f) {
    VirtualCells m(*this);
    std::vector<Expression> polys = f(m);
    if (polys.empty()) throw Error(AMDZK_E_INVALID, "create_gate: gates must contain at least one constraint");
    for (auto& p : polys) gates.push_back(p);
  }
  size_t lookup(const char* /*name*/, const std::function<std::vector<std::pair<Expression, Expression>>(VirtualCells&)>& f) {
    VirtualCells m(*this);
    Lookup lk;
    for (auto& pr : f(m)) {
      lk.inputs.push_back(pr.first);
      lk.tables.push_back(pr.second);
    }
    lookups.push_back(std::move(lk));
    return lookups.size() - 1;
  }

  // Upstream formulas.
  uint32_t degree() const {
    uint32_t d = 3;  // permutation::Argument::required_degree()
    for (const Lookup& lk : lookups) {
      uint32_t di = 1, dt = 1;
      for (auto& e : lk.inputs) di = std::max(di, e.degree());
      for (auto& e : lk.tables) dt = std::max(dt, e.degree());
      d = std::max(d, std::max(4u, 2 + di + dt));
    }
    for (auto& g : gates) d = std::max(d, g.degree());
    return std::max(d, std::max(minimum_degree, 1u));
  }
  uint32_t blinding_factors() const {
    uint32_t f = num_advice_queries.empty() ? 1 : 0;
    for (uint32_t q : num_advice_queries) f = std::max(f, q);
    return std::max(3u, f) + 2;
  }
  uint32_t minimum_rows() const { return blinding_factors() + 3; }
  size_t permutation_index(Column c) const {
    for (size_t i = 0; i < permutation_columns.size(); i++)
      if (permutation_columns[i] == c) return i;
    throw Error(AMDZK_E_INVALID, "column is not equality-enabled");
  }

  bool note_query(std::vector<Query>& lst, Column c, int32_t rot) {
    for (auto& q : lst)
      if (q.first == c && q.second == rot) return false;
    lst.push_back({c, rot});
    return true;
  }
};

inline Expression VirtualCells::query_advice(Column c, Rotation r) {
  if (c.kind != Any::Advice) throw Error(AMDZK_E_INVALID, "query_advice: not an advice column");
  if (cs_.note_query(cs_.advice_queries, c, r.v)) cs_.num_advice_queries[c.index]++;
  return Expression::query(Any::Advice, c.index, r.v);
}
inline Expression VirtualCells::query_fixed(Column c, Rotation r) {
  if (c.kind != Any::Fixed) throw Error(AMDZK_E_INVALID, "query_fixed: not a fixed column");
  cs_.note_query(cs_.fixed_queries, c, r.v);
  return Expression::query(Any::Fixed, c.index, r.v);
}
inline Expression VirtualCells::query_instance(Column c, Rotation r) {
  if (c.kind != Any::Instance) throw Error(AMDZK_E_INVALID, "query_instance: not an instance column");
  cs_.note_query(cs_.instance_queries, c, r.v);
  return Expression::query(Any::Instance, c.index, r.v);
}

// ------------------------------------------------------------------------------------------ flattening
// The plain-data form of a ConstraintSystem that amdzk_keygen takes (include/amdzk.h: amdzk_circuit).
struct CircuitData {
  std::vector<int32_t> aq, fq, iq;
  std::vector<uint32_t> lookup_shape, expr_offsets, expr_words, perm_columns;
  std::vector<uint64_t> constants;  // 4 limbs each, Montgomery
  amdzk_circuit c;

  CircuitData(const ConstraintSystem& cs, uint32_t k) {
    auto put = [](std::vector<int32_t>& dst, const std::vector<ConstraintSystem::Query>& qs) {
      for (auto& q : qs) {
        dst.push_back((int32_t)q.first.index);
        dst.push_back(q.second);
      }
    };
    put(aq, cs.advice_queries);
    put(fq, cs.fixed_queries);
    put(iq, cs.instance_queries);
    expr_offsets.push_back(0);
    auto add = [&](const Expression& e) {
      emit(e.node());
      expr_offsets.push_back((uint32_t)expr_words.size());
    };
    for (auto& g : cs.gates) add(g);
    for (auto& lk : cs.lookups) {
      lookup_shape.push_back((uint32_t)lk.inputs.size());
      lookup_shape.push_back((uint32_t)lk.tables.size());
      for (auto& e : lk.inputs) add(e);
      for (auto& e : lk.tables) add(e);
    }
    for (auto& p : cs.permutation_columns) {
      perm_columns.push_back((uint32_t)p.kind);
      perm_columns.push_back(p.index);
    }
    std::memset(&c, 0, sizeof(c));
    c.k = k;
    c.num_fixed = cs.num_fixed;
    c.num_advice = cs.num_advice;
    c.num_instance = cs.num_instance;
    c.blinding_factors = cs.blinding_factors();
    c.cs_degree = cs.degree();
    c.num_advice_queries = (uint32_t)cs.advice_queries.size();
    c.advice_queries = aq.data();
    c.num_fixed_queries = (uint32_t)cs.fixed_queries.size();
    c.fixed_queries = fq.data();
    c.num_instance_queries = (uint32_t)cs.instance_queries.size();
    c.instance_queries = iq.data();
    c.num_gates = (uint32_t)cs.gates.size();
    c.num_lookups = (uint32_t)cs.lookups.size();
    c.num_exprs = (uint32_t)expr_offsets.size() - 1;
    c.lookup_shape = lookup_shape.data();
    c.expr_offsets = expr_offsets.data();
    c.expr_words = expr_words.data();
    c.num_constants = (uint32_t)(constants.size() / 4);
    c.constants = constants.data();
    c.num_perm_columns = (uint32_t)cs.permutation_columns.size();
    c.perm_columns = perm_columns.data();
  }
  CircuitData(const CircuitData&) = delete;
  CircuitData& operator=(const CircuitData&) = delete;

 private:
  std::map<Fr, uint32_t> cidx_;
  uint32_t constant_index(const Fr& v) {
    auto it = cidx_.find(v);
    if (it != cidx_.end()) return it->second;
    uint32_t i = (uint32_t)cidx_.size();
    cidx_[v] = i;
    constants.insert(constants.end(), v.l, v.l + 4);
    return i;
  }
  void emit(const Expression::Node* n) {
    switch (n->op) {
      case Expression::Constant:
        expr_words.push_back((1u << 24) | constant_index(n->c));
        break;
      case Expression::Fixed: case Expression::Advice: case Expression::Instance:
        if (n->rotation < -128 || n->rotation > 127 || n->column >= (1u << 16)) throw Error(AMDZK_E_UNSUPPORTED, "query out of encodable range");
        expr_words.push_back(((uint32_t)n->op << 24) | (n->column << 8) | (uint32_t)(n->rotation + 128));
        break;
      case Expression::Negated:
        emit(n->a.get());
        expr_words.push_back(5u << 24);
        break;
      case Expression::Sum: case Expression::Product:
        emit(n->a.get());
        emit(n->b.get());
        expr_words.push_back((uint32_t)n->op << 24);
        break;
      case Expression::Scaled:
        emit(n->a.get());
        expr_words.push_back((8u << 24) | constant_index(n->c));
        break;
    }
  }
};

// ------------------------------------------------------------------------------------------ Assembly
// plonk::permutation::keygen::Assembly: copy-constraint cycles over the permutation columns.
class Assembly {
 public:
  Assembly(size_t n, size_t ncols) : n_(n), ncols_(ncols), mapping_(2 * n * ncols), aux_(2 * n * ncols), sizes_(n * ncols, 1) {
    for (size_t c = 0; c < ncols; c++)
      for (size_t r = 0; r < n; r++) {
        size_t i = c * n + r;
        mapping_[2 * i] = aux_[2 * i] = (uint32_t)c;
        mapping_[2 * i + 1] = aux_[2 * i + 1] = (uint32_t)r;
      }
  }
  // Assembly::copy: merge the cycles of (left column, row) and (right column, row), union by size.
  void copy(size_t lc, size_t lr, size_t rc, size_t rr) {
    if (lc >= ncols_ || rc >= ncols_ || lr >= n_ || rr >= n_) throw Error(AMDZK_E_INVALID, "Assembly::copy: out of bounds");
    size_t l = lc * n_ + lr, r = rc * n_ + rr;
    size_t lcy = (size_t)aux_[2 * l] * n_ + aux_[2 * l + 1], rcy = (size_t)aux_[2 * r] * n_ + aux_[2 * r + 1];
    if (lcy == rcy) return;
    if (sizes_[lcy] < sizes_[rcy]) std::swap(lcy, rcy);
    sizes_[lcy] += sizes_[rcy];
    size_t i = rcy;
    do {
      aux_[2 * i] = (uint32_t)(lcy / n_);
      aux_[2 * i + 1] = (uint32_t)(lcy % n_);
      i = (size_t)mapping_[2 * i] * n_ + mapping_[2 * i + 1];
    } while (i != rcy);
    std::swap(mapping_[2 * l], mapping_[2 * r]);
    std::swap(mapping_[2 * l + 1], mapping_[2 * r + 1]);
  }
  const uint32_t* mapping() const { return mapping_.data(); }  // amdzk_keygen's perm_mapping layout
  size_t n() const { return n_; }
  size_t num_columns() const { return ncols_; }

 private:
  size_t n_, ncols_;
  std::vector<uint32_t> mapping_, aux_;
  std::vector<uint64_t> sizes_;
};

// ------------------------------------------------------------------------------------------ ParamsKZG
struct G1Affine {
  uint64_t x[4], y[4];  // Montgomery Fq; (0, 0) = identity
};
struct G1 {
  uint64_t x[4], y[4], z[4];  // Jacobian, z = 0 identity; results come back with z = 1
};

class ParamsKZG {
 public:
  ParamsKZG(const Context& ctx, uint32_t k, const G1Affine* g, const G1Affine* g_lagrange) : ctx_(ctx), k_(k) {
    ctx_.check(amdzk_srs_upload(ctx_.get(), (const uint64_t*)g, (const uint64_t*)g_lagrange, k, &h_));
  }
  // ParamsKZG::setup(k, rng) with the trapdoor given explicitly (tests / benchmarks, as unsafe as upstream's).
  static ParamsKZG setup(const Context& ctx, uint32_t k, const Fr& s) {
    amdzk_srs* h = nullptr;
    ctx.check(amdzk_srs_setup(ctx.get(), k, s.l, &h, nullptr, nullptr));
    return ParamsKZG(ctx, k, h);
  }
  static ParamsKZG read(const Context& ctx, const std::vector<uint8_t>& data, uint8_t g2[64] = nullptr, uint8_t s_g2[64] = nullptr) {
    if (data.size() < 4) throw Error(AMDZK_E_INVALID, "ParamsKZG::read: truncated");
    uint32_t k;
    std::memcpy(&k, data.data(), 4);
    amdzk_srs* h = nullptr;
    ctx.check(amdzk_srs_read(ctx.get(), data.data(), data.size(), &h, g2, s_g2));
    return ParamsKZG(ctx, k, h);
  }
  std::vector<uint8_t> write(const uint8_t g2[64], const uint8_t s_g2[64]) const {
    std::vector<uint8_t> out(amdzk_srs_serialized_size(k_));
    ctx_.check(amdzk_srs_write(ctx_.get(), h_, g2, s_g2, out.data(), out.size()));
    return out;
  }
  void downsize(uint32_t k) {
    amdzk_srs* h = nullptr;
    ctx_.check(amdzk_srs_downsize(ctx_.get(), h_, k, &h));
    amdzk_srs_free(ctx_.get(), h_);
    h_ = h;
    k_ = k;
  }
  std::vector<G1Affine> get_g() const { return get(0); }
  std::vector<G1Affine> get_g_lagrange() const { return get(1); }
  G1 commit(const std::vector<Fr>& poly) const { return msm(0, poly); }            // blind ignored for KZG
  G1 commit_lagrange(const std::vector<Fr>& poly) const { return msm(1, poly); }
  uint32_t k() const { return k_; }
  uint64_t n() const { return (uint64_t)1 << k_; }
  const amdzk_srs* handle() const { return h_; }

  ~ParamsKZG() {
    if (h_) amdzk_srs_free(ctx_.get(), h_);
  }
  ParamsKZG(ParamsKZG&& o) noexcept : ctx_(o.ctx_), k_(o.k_), h_(o.h_) { o.h_ = nullptr; }
  ParamsKZG(const ParamsKZG&) = delete;
  ParamsKZG& operator=(const ParamsKZG&) = delete;

 private:
  ParamsKZG(const Context& ctx, uint32_t k, amdzk_srs* h) : ctx_(ctx), k_(k), h_(h) {}
  std::vector<G1Affine> get(int basis) const {
    std::vector<G1Affine> out((size_t)1 << k_);
    ctx_.check(amdzk_srs_get(ctx_.get(), h_, basis, (uint64_t*)out.data()));
    return out;
  }
  G1 msm(int basis, const std::vector<Fr>& poly) const {
    G1 out;
    ctx_.check(amdzk_msm_g1(ctx_.get(), h_, basis, (const uint64_t*)poly.data(), poly.size(), (uint64_t*)&out));
    return out;
  }
  const Context& ctx_;
  uint32_t k_;
  amdzk_srs* h_ = nullptr;
};

// ------------------------------------------------------------------------------------------ keys and proofs
// plonk::keygen_vk + keygen_pk: fixed[c] = the 2^k Lagrange values of fixed column c (selectors included).
class ProvingKey {
 public:
  // flags: AMDZK_KEYGEN_* of include/amdzk.h or'ed together; kEnvDefaults = amdzk_keygen (the modes from the environment)
  static constexpr uint32_t kEnvDefaults = 0xffffffffu;
  ProvingKey(const Context& ctx, const ParamsKZG& params, const ConstraintSystem& cs, const std::vector<std::vector<Fr>>& fixed,
             const Assembly& assembly, const Fr& transcript_repr, uint32_t flags = kEnvDefaults)
      : ctx_(ctx), num_fixed_(cs.num_fixed), num_perm_(cs.permutation_columns.size()), num_advice_(cs.num_advice), k_(params.k()) {
    const size_t n = (size_t)1 << k_;
    if (fixed.size() != cs.num_fixed) throw Error(AMDZK_E_INVALID, "keygen: fixed column count");
    if (assembly.n() != n || assembly.num_columns() != num_perm_) throw Error(AMDZK_E_INVALID, "keygen: assembly shape");
    std::vector<Fr> flat(fixed.size() * n, Fr::zero());
    for (size_t c = 0; c < fixed.size(); c++) {
      if (fixed[c].size() > n) throw Error(AMDZK_E_INVALID, "keygen: fixed column longer than 2^k");
      std::copy(fixed[c].begin(), fixed[c].end(), flat.begin() + c * n);
    }
    CircuitData cd(cs, k_);
    const uint64_t* fx = flat.empty() ? nullptr : (const uint64_t*)flat.data();
    const uint32_t* mp = num_perm_ ? assembly.mapping() : nullptr;
    if (flags == kEnvDefaults) ctx_.check(amdzk_keygen(ctx_.get(), params.handle(), &cd.c, fx, mp, transcript_repr.l, &h_));
    else ctx_.check(amdzk_keygen_ex(ctx_.get(), params.handle(), &cd.c, fx, mp, transcript_repr.l, flags, &h_));
  }
  ~ProvingKey() {
    if (h_) amdzk_pk_free(ctx_.get(), h_);
  }
  ProvingKey(const ProvingKey&) = delete;
  ProvingKey& operator=(const ProvingKey&) = delete;
  // amdzk_pk_clone_workspace: a key sharing this key's material that owns one more circuit instance's per-proof
  // workspace (the second, third, ... instance of a multi-circuit create_proof, or one more proof in flight). Must not
  // outlive this key.
  std::unique_ptr<ProvingKey> clone_workspace() const {
    std::unique_ptr<ProvingKey> c(new ProvingKey(ctx_, num_fixed_, num_perm_, num_advice_, k_));
    ctx_.check(amdzk_pk_clone_workspace(ctx_.get(), h_, &c->h_));
    return c;
  }
  // VerifyingKey::{fixed_commitments, permutation.commitments}
  void commitments(std::vector<G1Affine>& fixed, std::vector<G1Affine>& permutation) const {
    fixed.assign(num_fixed_, G1Affine{});
    permutation.assign(num_perm_, G1Affine{});
    ctx_.check(amdzk_pk_commitments(h_, num_fixed_ ? (uint64_t*)fixed.data() : nullptr, num_perm_ ? (uint64_t*)permutation.data() : nullptr));
  }
  amdzk_pk* handle() const { return h_; }
  uint32_t k() const { return k_; }
  size_t num_advice() const { return num_advice_; }

 private:
  ProvingKey(const Context& ctx, size_t nf, size_t np, size_t na, uint32_t k) : ctx_(ctx), num_fixed_(nf), num_perm_(np), num_advice_(na), k_(k) {}
  const Context& ctx_;
  size_t num_fixed_, num_perm_, num_advice_;
  uint32_t k_;
  amdzk_pk* h_ = nullptr;
};

enum class Transcript : int { Blake2b = AMDZK_TRANSCRIPT_BLAKE2B, Keccak256Evm = AMDZK_TRANSCRIPT_KECCAK256_EVM };
// The `P: Prover` type parameter of create_proof: poly::kzg::multiopen::{ProverSHPLONK, ProverGWC}.
enum class Multiopen : int { Shplonk = 0, Gwc = AMDZK_MULTIOPEN_GWC };

// plonk::create_proof(params, pk, &[circuit], &[instances], ChaCha20Rng::seed_from_u64(seed), transcript) for
// one circuit. `d_advice`: the witness columns resident on the GPU (column c at d_advice + c*stride Fr).
inline std::vector<uint8_t> create_proof(const Context& ctx, const ProvingKey& pk, const std::vector<std::vector<Fr>>& instances,
                                         const void* d_advice, size_t advice_stride, uint64_t rng_seed,
                                         Transcript transcript = Transcript::Blake2b, Multiopen multiopen = Multiopen::Shplonk) {
  const int format = (int)transcript | (int)multiopen;
  std::vector<const uint64_t*> ptrs(std::max<size_t>(1, instances.size()), nullptr);
  std::vector<size_t> lens(std::max<size_t>(1, instances.size()), 0);
  for (size_t i = 0; i < instances.size(); i++) {
    ptrs[i] = instances[i].empty() ? nullptr : (const uint64_t*)instances[i].data();
    lens[i] = instances[i].size();
  }
  size_t need = 0;
  std::vector<uint8_t> proof(amdzk_proof_size(pk.handle(), format));
  ctx.check(amdzk_create_proof_ex(ctx.get(), pk.handle(), ptrs.data(), lens.data(), d_advice, advice_stride, rng_seed, format,
                                  proof.data(), proof.size(), &need));
  proof.resize(need);
  return proof;
}

// plonk::create_proof(params, pk, &[circuit; N], &[instances; N], rng, transcript): N instances of the key's circuit in
// ONE proof (upstream's slices). keys[c]: instance c's workspace — the key itself and ProvingKey::clone_workspace() handles,
// pairwise different; instances[c], d_advice[c] as for one circuit.
inline std::vector<uint8_t> create_proof(const Context& ctx, const std::vector<const ProvingKey*>& keys,
                                         const std::vector<std::vector<std::vector<Fr>>>& instances, const std::vector<const void*>& d_advice,
                                         size_t advice_stride, uint64_t rng_seed, Transcript transcript = Transcript::Blake2b,
                                         Multiopen multiopen = Multiopen::Shplonk) {
  const size_t N = keys.size();
  if (N == 0 || instances.size() != N || d_advice.size() != N) throw Error(AMDZK_E_INVALID, "create_proof: one key, instance set and witness per circuit");
  const int format = (int)transcript | (int)multiopen;
  std::vector<amdzk_pk*> pks(N);
  std::vector<std::vector<const uint64_t*>> ptrs(N);
  std::vector<std::vector<size_t>> lens(N);
  std::vector<const uint64_t* const*> pp(N);
  std::vector<const size_t*> lp(N);
  for (size_t c = 0; c < N; c++) {
    pks[c] = keys[c]->handle();
    ptrs[c].assign(std::max<size_t>(1, instances[c].size()), nullptr);
    lens[c].assign(std::max<size_t>(1, instances[c].size()), 0);
    for (size_t i = 0; i < instances[c].size(); i++) {
      ptrs[c][i] = instances[c][i].empty() ? nullptr : (const uint64_t*)instances[c][i].data();
      lens[c][i] = instances[c][i].size();
    }
    pp[c] = ptrs[c].data();
    lp[c] = lens[c].data();
  }
  size_t need = 0;
  std::vector<uint8_t> proof(amdzk_proof_size_multi(pks[0], N, format));
  ctx.check(amdzk_create_proof_multi(ctx.get(), pks.data(), N, pp.data(), lp.data(), d_advice.data(), advice_stride, rng_seed, format, proof.data(),
                                     proof.size(), &need));
  proof.resize(need);
  return proof;
}

// Convenience for host-resident witnesses (what a WitnessCollection holds after synthesis): uploads the
// advice columns (padded with zeros to 2^k rows), proves, frees.
inline std::vector<uint8_t> create_proof(const Context& ctx, const ProvingKey& pk, const std::vector<std::vector<Fr>>& instances,
                                         const std::vector<std::vector<Fr>>& advice, uint64_t rng_seed,
                                         Transcript transcript = Transcript::Blake2b, Multiopen multiopen = Multiopen::Shplonk) {
  const size_t n = (size_t)1 << pk.k();
  if (advice.size() != pk.num_advice()) throw Error(AMDZK_E_INVALID, "create_proof: advice column count");
  std::vector<Fr> flat(std::max<size_t>(1, advice.size()) * n, Fr::zero());
  for (size_t c = 0; c < advice.size(); c++) {
    if (advice[c].size() > n) throw Error(AMDZK_E_INVALID, "create_proof: advice column longer than 2^k");
    std::copy(advice[c].begin(), advice[c].end(), flat.begin() + c * n);
  }
  void* d = nullptr;
  ctx.check(amdzk_dev_alloc(ctx.get(), flat.size() * sizeof(Fr), &d));
  std::vector<uint8_t> proof;
  try {
    ctx.check(amdzk_dev_upload(ctx.get(), d, flat.data(), flat.size() * sizeof(Fr)));
    proof = create_proof(ctx, pk, instances, d, n, rng_seed, transcript, multiopen);
  } catch (...) {
    amdzk_dev_free(ctx.get(), d);
    throw;
  }
  amdzk_dev_free(ctx.get(), d);
  return proof;
}

// One witness per proof from the host (what a caller does after Circuit::synthesize, /root/reference/src/lib.rs:328-397):
// the advice columns are synthesized into pinned memory, and the upload of proof i+1's witness runs on the ctx's copy
// stream while proof i is proved (amdzk_dev_upload_async / amdzk_upload_fence; double-buffered device staging).
class WitnessStream {
 public:
  WitnessStream(const Context& ctx, const ProvingKey& pk) : ctx_(ctx), pk_(pk), n_((size_t)1 << pk.k()) {
    bytes_ = std::max<size_t>(1, pk.num_advice()) * n_ * sizeof(Fr);
    for (int i = 0; i < 2; i++) {
      ctx_.check(amdzk_dev_alloc(ctx_.get(), bytes_, &dev_[i]));
      ctx_.check(amdzk_host_alloc(ctx_.get(), bytes_, &pin_[i]));
    }
  }
  ~WitnessStream() {
    for (int i = 0; i < 2; i++) {
      if (dev_[i]) amdzk_dev_free(ctx_.get(), dev_[i]);
      if (pin_[i]) amdzk_host_free(ctx_.get(), pin_[i]);
    }
  }
  WitnessStream(const WitnessStream&) = delete;
  WitnessStream& operator=(const WitnessStream&) = delete;

  // create_proof for every (advice, instances, seed) of the batch, in order.
  struct Item {
    const std::vector<std::vector<Fr>>* advice;
    const std::vector<std::vector<Fr>>* instances;
    uint64_t rng_seed;
  };
  std::vector<std::vector<uint8_t>> prove(const std::vector<Item>& items, Transcript transcript = Transcript::Blake2b,
                                          Multiopen multiopen = Multiopen::Shplonk) {
    std::vector<std::vector<uint8_t>> out;
    if (items.empty()) return out;
    stage(0, *items[0].advice);
    for (size_t j = 0; j < items.size(); j++) {
      const int cur = (int)(j & 1);
      ctx_.check(amdzk_upload_fence(ctx_.get()));                    // proof j's witness is (being) uploaded: order the proof behind it
      if (j + 1 < items.size()) stage(cur ^ 1, *items[j + 1].advice);  // proof j-1, the last reader of that buffer, has returned
      out.push_back(create_proof(ctx_, pk_, *items[j].instances, dev_[cur], n_, items[j].rng_seed, transcript, multiopen));
    }
    return out;
  }

 private:
  void stage(int slot, const std::vector<std::vector<Fr>>& advice) {
    if (advice.size() != pk_.num_advice()) throw Error(AMDZK_E_INVALID, "WitnessStream: advice column count");
    Fr* h = (Fr*)pin_[slot];
    for (size_t c = 0; c < advice.size(); c++) {
      if (advice[c].size() > n_) throw Error(AMDZK_E_INVALID, "WitnessStream: advice column longer than 2^k");
      std::copy(advice[c].begin(), advice[c].end(), h + c * n_);
      std::fill(h + c * n_ + advice[c].size(), h + (c + 1) * n_, Fr::zero());
    }
    ctx_.check(amdzk_dev_upload_async(ctx_.get(), dev_[slot], pin_[slot], bytes_));
  }
  const Context& ctx_;
  const ProvingKey& pk_;
  size_t n_, bytes_ = 0;
  void* dev_[2] = {nullptr, nullptr};
  void* pin_[2] = {nullptr, nullptr};
};

}  // namespace halo2
}  // namespace amdzk
#endif /* AMDZK_HALO2_HPP */
