// Drives include/amdzk_halo2.hpp the way a compiled host (the reference's Rust, through its FFI) would:
// circuits are configured in C++ against the mirrored ConstraintSystem, then
//   describe <circuit> <k>                      print the flattened C-ABI arrays (no GPU call)
//   prove <circuit> <k> <witness> <seed> <blake2b|evm>[+gwc] <tau hex> <transcript_repr hex>
//                                               keygen + create_proof on the GPU, print the proof as hex
// Circuit configurations mirror tests/circuits.py (and through it /root/reference/src/signal.rs:27-49,
// src/conditional_secrets.rs:81-187, src/timestamp.rs:58-68, src/lib.rs:295-326); the witness file
// (cells, instances, copy constraints) is written by tests/test_cpp_mirror.py from the Python fixture.
#include <cstdio>
#include <cstdlib>
#include <fstream>
#include <iostream>
#include <sstream>

#include "amdzk_halo2.hpp"

using namespace amdzk::halo2;
using Exprs = std::vector<Expression>;
using Pairs = std::vector<std::pair<Expression, Expression>>;

static void configure_square(ConstraintSystem& cs) {
  Column a0 = cs.advice_column(), a1 = cs.advice_column();
  Column instance = cs.instance_column();
  Column s = cs.selector();
  cs.enable_equality(a0);
  cs.enable_equality(a1);
  cs.enable_equality(instance);
  cs.create_gate("square", [&](VirtualCells& m) {
    Expression sel = m.query_selector(s);
    Expression a = m.query_advice(a0, Rotation::cur());
    Expression sq = m.query_advice(a1, Rotation::cur());
    return Exprs{sel * (sq - a * a)};
  });
}

static void configure_lookup(ConstraintSystem& cs) {
  Column a = cs.advice_column(), b = cs.advice_column(), c = cs.advice_column();
  Column q_mul = cs.selector(), q_add = cs.selector(), q_rng = cs.selector(), q_pair = cs.selector();
  Column t_rng = cs.fixed_column(), t_x = cs.fixed_column(), t_y = cs.fixed_column(), konst = cs.fixed_column();
  Column inst = cs.instance_column();
  for (Column col : {a, b, c, konst, inst}) cs.enable_equality(col);
  // One query per statement: C++ leaves the evaluation order of a * b unspecified, and the order of
  // first use is what numbers the queries (as in Rust, where it is left to right).
  cs.create_gate("mul/add", [&](VirtualCells& m) {
    Expression s0 = m.query_selector(q_mul);
    Expression a0 = m.query_advice(a, Rotation::cur());
    Expression b0 = m.query_advice(b, Rotation::cur());
    Expression c0 = m.query_advice(c, Rotation::cur());
    Expression s1 = m.query_selector(q_add);
    Expression a1 = m.query_advice(a, Rotation::cur());
    Expression b1 = m.query_advice(b, Rotation::next());
    Expression c1 = m.query_advice(c, Rotation::cur());
    return Exprs{s0 * (a0 * b0 - c0), s1 * (a1 + b1 - c1)};
  });
  cs.lookup("range", [&](VirtualCells& m) {
    Expression s = m.query_selector(q_rng);
    Expression v = m.query_advice(a, Rotation::cur());
    Expression t = m.query_fixed(t_rng);
    return Pairs{{s * v, t}};
  });
  cs.lookup("pair", [&](VirtualCells& m) {
    Expression s0 = m.query_selector(q_pair);
    Expression v0 = m.query_advice(b, Rotation::cur());
    Expression t0 = m.query_fixed(t_x);
    Expression s1 = m.query_selector(q_pair);
    Expression v1 = m.query_advice(c, Rotation::cur());
    Expression t1 = m.query_fixed(t_y);
    return Pairs{{s0 * v0, t0}, {s1 * v1, t1}};
  });
}

// tests/circuits.py full_aadhaar_shape(num_advice=5, num_lookup_advice=2, num_spread=2) and its 80/16/8 original.
static void configure_aadhaar(ConstraintSystem& cs, int num_advice, int num_lookup_advice, int num_spread) {
  std::vector<Column> gate_cols, sels, lk_cols, sp_dense, sp_spread;
  for (int i = 0; i < num_advice; i++) gate_cols.push_back(cs.advice_column());
  for (int i = 0; i < num_advice; i++) sels.push_back(cs.selector());
  for (int i = 0; i < num_lookup_advice; i++) lk_cols.push_back(cs.advice_column());
  for (int i = 0; i < num_spread; i++) sp_dense.push_back(cs.advice_column());
  for (int i = 0; i < num_spread; i++) sp_spread.push_back(cs.advice_column());
  Column t_rng = cs.fixed_column(), t_dense = cs.fixed_column(), t_spread = cs.fixed_column(), konst = cs.fixed_column();
  Column inst0 = cs.instance_column(), inst1 = cs.instance_column();
  for (auto* v : {&gate_cols, &lk_cols, &sp_dense, &sp_spread})
    for (Column col : *v) cs.enable_equality(col);
  cs.enable_equality(konst);
  cs.enable_equality(inst0);
  cs.enable_equality(inst1);
  for (int i = 0; i < num_advice; i++) {
    Column col = gate_cols[i], s = sels[i];
    cs.create_gate("vertical", [&](VirtualCells& m) {
      Expression q = m.query_selector(s);
      Expression a = m.query_advice(col, {0}), b = m.query_advice(col, {1}), c = m.query_advice(col, {2}), d = m.query_advice(col, {3});
      return Exprs{q * (a + b * c - d)};
    });
  }
  for (Column col : lk_cols)
    cs.lookup("range", [&](VirtualCells& m) {
      Expression in = m.query_advice(col, Rotation::cur());
      Expression t = m.query_fixed(t_rng);
      return Pairs{{in, t}};
    });
  for (int i = 0; i < num_spread; i++) {
    Column d = sp_dense[i], sp = sp_spread[i];
    cs.lookup("spread", [&](VirtualCells& m) {
      Expression i0 = m.query_advice(d, Rotation::cur());
      Expression t0 = m.query_fixed(t_dense);
      Expression i1 = m.query_advice(sp, Rotation::cur());
      Expression t1 = m.query_fixed(t_spread);
      return Pairs{{i0, t0}, {i1, t1}};
    });
  }
  // IdentityCircuit (src/conditional_secrets.rs:81-187)
  enum { REVEAL_AGE, AGE, QR_AGE, REVEAL_GENDER, GENDER, QR_GENDER, REVEAL_PIN, PIN, QR_PIN, REVEAL_STATE, STATE0, QR_STATE0 = STATE0 + 5, ID_COLS = QR_STATE0 + 5 };
  std::vector<Column> id;
  for (int i = 0; i < ID_COLS; i++) id.push_back(cs.advice_column());
  Column s_id = cs.selector();
  const Fr one = Fr::one();
  for (int r : {REVEAL_AGE, REVEAL_GENDER, REVEAL_PIN, REVEAL_STATE})
    cs.create_gate("boolean", [&](VirtualCells& m) {
      Expression s = m.query_selector(s_id);
      Expression x = m.query_advice(id[r], Rotation::cur());
      Expression y = m.query_advice(id[r], Rotation::cur());
      return Exprs{s * x * (y - Expression::constant(one))};
    });
  cs.create_gate("age", [&](VirtualCells& m) {
    Expression s = m.query_selector(s_id);
    Expression age = m.query_advice(id[AGE], Rotation::cur());
    Expression rv = m.query_advice(id[REVEAL_AGE], Rotation::cur());
    Expression qr = m.query_advice(id[QR_AGE], Rotation::cur());
    return Exprs{s * (age - rv * qr)};
  });
  for (auto pr : {std::make_pair(GENDER, QR_GENDER), std::make_pair(PIN, QR_PIN)})
    cs.create_gate("equal", [&](VirtualCells& m) {
      Expression s = m.query_selector(s_id);
      Expression v = m.query_advice(id[pr.first], Rotation::cur());
      Expression qr = m.query_advice(id[pr.second], Rotation::cur());
      return Exprs{s * (v - qr)};
    });
  cs.create_gate("state", [&](VirtualCells& m) {
    Exprs out;
    for (int i = 0; i < 5; i++) {
      Expression s = m.query_selector(s_id);
      Expression v = m.query_advice(id[STATE0 + i], Rotation::cur());
      Expression qr = m.query_advice(id[QR_STATE0 + i], Rotation::cur());
      out.push_back(s * (v - qr));
    }
    return out;
  });
  // TimestampCircuit (src/timestamp.rs:58-68): seven advice columns, no gate queries them
  for (int i = 0; i < 7; i++) cs.advice_column();
  // SquareCircuit (src/signal.rs:27-49)
  Column q0 = cs.advice_column(), q1 = cs.advice_column();
  Column sq_inst = cs.instance_column();
  Column s_sq = cs.selector();
  cs.enable_equality(q0);
  cs.enable_equality(q1);
  cs.enable_equality(sq_inst);
  cs.create_gate("square", [&](VirtualCells& m) {
    Expression s = m.query_selector(s_sq);
    Expression sq = m.query_advice(q1, Rotation::cur());
    Expression a = m.query_advice(q0, Rotation::cur());
    Expression a2 = m.query_advice(q0, Rotation::cur());
    return Exprs{s * (sq - a * a2)};
  });
}

static void configure(ConstraintSystem& cs, const std::string& name) {
  if (name == "square") configure_square(cs);
  else if (name == "lookup") configure_lookup(cs);
  else if (name == "aadhaar_small") configure_aadhaar(cs, 5, 2, 2);
  else if (name == "aadhaar") configure_aadhaar(cs, 80, 16, 8);
  else throw Error(AMDZK_E_INVALID, "unknown circuit " + name);
}

template <class T>
static void dump(const char* tag, const std::vector<T>& v) {
  std::cout << tag;
  for (auto& x : v) std::cout << ' ' << (long long)x;
  std::cout << '\n';
}

static int describe(const std::string& name, uint32_t k) {
  ConstraintSystem cs;
  configure(cs, name);
  CircuitData cd(cs, k);
  std::cout << "shape " << cd.c.k << ' ' << cd.c.num_fixed << ' ' << cd.c.num_advice << ' ' << cd.c.num_instance << ' ' << cd.c.blinding_factors << ' '
            << cd.c.cs_degree << ' ' << cd.c.num_gates << ' ' << cd.c.num_lookups << ' ' << cd.c.num_exprs << ' ' << cs.minimum_rows() << '\n';
  dump("aq", cd.aq);
  dump("fq", cd.fq);
  dump("iq", cd.iq);
  dump("lookup_shape", cd.lookup_shape);
  dump("expr_offsets", cd.expr_offsets);
  dump("expr_words", cd.expr_words);
  std::cout << "constants";
  for (uint64_t w : cd.constants) std::printf(" %016llx", (unsigned long long)w);
  std::cout << '\n';
  dump("perm", cd.perm_columns);
  return 0;
}

static int prove(int argc, char** argv) {
  const std::string name = argv[2];
  const uint32_t k = (uint32_t)std::atoi(argv[3]);
  const size_t n = (size_t)1 << k;
  ConstraintSystem cs;
  configure(cs, name);
  std::vector<std::vector<Fr>> fixed(cs.num_fixed, std::vector<Fr>(n, Fr::zero())), advice(cs.num_advice, std::vector<Fr>(n, Fr::zero())),
      instances(cs.num_instance);
  Assembly assembly(n, cs.permutation_columns.size());
  std::ifstream f(argv[4]);
  if (!f) throw Error(AMDZK_E_INVALID, "cannot open witness file");
  std::string line;
  while (std::getline(f, line)) {
    std::istringstream is(line);
    char tag;
    is >> tag;
    if (tag == 'F' || tag == 'A') {
      size_t col, row;
      std::string hex;
      is >> col >> row >> hex;
      (tag == 'F' ? fixed : advice).at(col).at(row) = Fr::from_hex(hex);
    } else if (tag == 'I') {
      size_t col;
      std::string hex;
      is >> col;
      while (is >> hex) instances.at(col).push_back(Fr::from_hex(hex));
    } else if (tag == 'C') {
      size_t c1, r1, c2, r2;
      is >> c1 >> r1 >> c2 >> r2;
      assembly.copy(c1, r1, c2, r2);
    }
  }
  const uint64_t seed = std::strtoull(argv[5], nullptr, 10);
  const std::string fmt = argv[6];
  const Transcript tr = fmt.rfind("evm", 0) == 0 ? Transcript::Keccak256Evm : Transcript::Blake2b;
  const Multiopen mo = fmt.find("+gwc") != std::string::npos ? Multiopen::Gwc : Multiopen::Shplonk;
  Context ctx(0);
  ParamsKZG params = ParamsKZG::setup(ctx, k, Fr::from_hex(argv[7]));
  ProvingKey pk(ctx, params, cs, fixed, assembly, Fr::from_hex(argv[8]));
  std::vector<uint8_t> proof = create_proof(ctx, pk, instances, advice, seed, tr, mo);
  std::cout << "proof ";
  for (uint8_t b : proof) std::printf("%02x", b);
  std::cout << '\n';
  {  // the same witness through the streaming path, three proofs with seeds seed, seed + 1, seed + 2 (both staging buffers used twice)
    WitnessStream ws(ctx, pk);
    std::vector<WitnessStream::Item> items;
    for (uint64_t j = 0; j < 3; j++) items.push_back({&advice, &instances, seed + j});
    for (const auto& pr : ws.prove(items, tr, mo)) {
      std::cout << "streamed ";
      for (uint8_t b : pr) std::printf("%02x", b);
      std::cout << '\n';
    }
  }
  {  // upstream's slices: the same circuit twice in ONE proof, through the key and a workspace clone of it
    const size_t n = (size_t)1 << k;
    std::vector<Fr> flat(std::max<size_t>(1, advice.size()) * n, Fr::zero());
    for (size_t c = 0; c < advice.size(); c++) std::copy(advice[c].begin(), advice[c].end(), flat.begin() + c * n);
    void* d = nullptr;
    ctx.check(amdzk_dev_alloc(ctx.get(), flat.size() * sizeof(Fr), &d));
    ctx.check(amdzk_dev_upload(ctx.get(), d, flat.data(), flat.size() * sizeof(Fr)));
    {
      std::unique_ptr<ProvingKey> pk2 = pk.clone_workspace();
      std::vector<uint8_t> two = create_proof(ctx, {&pk, pk2.get()}, {instances, instances}, {d, d}, n, seed, tr, mo);
      std::cout << "multi2 ";
      for (uint8_t b : two) std::printf("%02x", b);
      std::cout << '\n';
    }
    amdzk_dev_free(ctx.get(), d);
  }
  // ParamsKZG surface: commit of the first advice column in both bases (checked by the caller)
  std::vector<G1Affine> fc, pc;
  pk.commitments(fc, pc);
  std::cout << "commitments " << fc.size() << ' ' << pc.size() << '\n';
  return 0;
}

int main(int argc, char** argv) {
  try {
    if (argc == 4 && std::string(argv[1]) == "describe") return describe(argv[2], (uint32_t)std::atoi(argv[3]));
    if (argc == 9 && std::string(argv[1]) == "prove") return prove(argc, argv);
    std::fprintf(stderr, "usage: %s describe <circuit> <k> | prove <circuit> <k> <witness> <seed> <blake2b|evm> <tau hex> <transcript_repr hex>\n", argv[0]);
    return 2;
  } catch (const Error& e) {
    std::fprintf(stderr, "error %d: %s\n", e.code, e.what());
    return 1;
  }
}
