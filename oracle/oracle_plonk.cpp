// ORACLE — TEST INFRASTRUCTURE ONLY ("parity unpinned" vs the reference except where
// oracle/contract_sol.py pins the protocol; see bn254_ref.hpp and DESIGN.md §4).
//
// CPU (OpenMP) restatement of the O(n) loops of halo2_proofs 0.2.0 @ v2023_01_20 [UP]
// (/root/reference/Cargo.lock:469-471) that oracle/plonk_fast.py needs to prove at the reference
// circuit's real size (k = 15):
//   plonk/evaluation.rs   evaluate()/evaluate_h  — expressions on all rows, the h(X) numerator
//   plonk/lookup/prover.rs permute_expression_pair
//   plonk/permutation/prover.rs, lookup/prover.rs — running products
// Arrays are n x 4 u64 Montgomery Fr. Expressions arrive as postfix words produced by plonk_fast.py's
// own flattener: op<<24 | payload; 1 CONST idx, 2 FIXED, 3 ADVICE, 4 INSTANCE (col<<8 | rot+128),
// 5 NEG, 6 ADD, 7 MUL, 8 SCALE idx.
#include <omp.h>
#include <stdlib.h>

#include <algorithm>
#include <vector>

#include "bn254_ref.hpp"

using namespace oref;

namespace {
struct Cols {
  const Fr* const* fixed;
  const Fr* const* advice;
  const Fr* const* instance;
  const Fr* consts;
  size_t mask;       // rows - 1
  long rot_scale;
};

inline Fr eval_expr(const uint32_t* w, uint32_t len, const Cols& c, size_t row, Fr* st) {
  int sp = 0;
  for (uint32_t i = 0; i < len; i++) {
    uint32_t op = w[i] >> 24, pl = w[i] & 0xffffffu;
    switch (op) {
      case 1: st[sp++] = c.consts[pl]; break;
      case 2: case 3: case 4: {
        long rot = (long)(pl & 0xff) - 128;
        size_t idx = (size_t)((long)row + rot * c.rot_scale) & c.mask;
        const Fr* col = op == 2 ? c.fixed[pl >> 8] : op == 3 ? c.advice[pl >> 8] : c.instance[pl >> 8];
        st[sp++] = col[idx];
      } break;
      case 5: st[sp - 1] = st[sp - 1].neg(); break;
      case 6: st[sp - 2] = st[sp - 2] + st[sp - 1]; sp--; break;
      case 7: st[sp - 2] = st[sp - 2] * st[sp - 1]; sp--; break;
      case 8: st[sp - 1] = st[sp - 1] * c.consts[pl]; break;
    }
  }
  return st[0];
}

inline bool canon_less(const uint64_t* a, const uint64_t* b) {
  for (int i = 3; i >= 0; i--)
    if (a[i] != b[i]) return a[i] < b[i];
  return false;
}
struct Key {
  uint64_t v[4];
  bool operator<(const Key& o) const { return canon_less(v, o.v); }
  bool operator==(const Key& o) const { return v[0] == o.v[0] && v[1] == o.v[1] && v[2] == o.v[2] && v[3] == o.v[3]; }
};
}  // namespace

extern "C" {

// out[row] = fold over exprs: acc = acc*theta + expr(row)   (lookup compression; nexprs = 1 and theta
// irrelevant for a plain expression). rows = mask + 1.
void oracle_eval_compressed(const uint32_t* words, const uint32_t* offsets, uint32_t nexprs, const uint64_t* const* fixed,
                            const uint64_t* const* advice, const uint64_t* const* instance, const uint64_t* consts, size_t rows,
                            long rot_scale, const uint64_t* theta, uint64_t* out, int threads) {
  Cols c{(const Fr* const*)fixed, (const Fr* const*)advice, (const Fr* const*)instance, (const Fr*)consts, rows - 1, rot_scale};
  Fr th = *(const Fr*)theta;
  Fr* O = (Fr*)out;
#pragma omp parallel num_threads(threads)
  {
    Fr st[64];
#pragma omp for schedule(static)
    for (size_t row = 0; row < rows; row++) {
      Fr acc = Fr::zero();
      for (uint32_t e = 0; e < nexprs; e++) acc = acc * th + eval_expr(words + offsets[e], offsets[e + 1] - offsets[e], c, row, st);
      O[row] = acc;
    }
  }
}

// lookup::prover::permute_expression_pair (values only; the caller appends the blinding rows).
int oracle_permute_pair(const uint64_t* inp, const uint64_t* tab, size_t usable, uint64_t* out_a, uint64_t* out_s) {
  std::vector<Key> a(usable), t(usable);
  for (size_t i = 0; i < usable; i++) {
    ((const Fr*)inp)[i].to_raw(a[i].v);
    ((const Fr*)tab)[i].to_raw(t[i].v);
  }
  std::sort(a.begin(), a.end());
  std::sort(t.begin(), t.end());
  std::vector<uint8_t> used(usable, 0);
  std::vector<size_t> repeated;
  std::vector<Key> s(usable);
  size_t tp = 0;
  for (size_t row = 0; row < usable; row++) {
    if (row == 0 || !(a[row] == a[row - 1])) {
      while (tp < usable && t[tp] < a[row]) tp++;
      if (tp >= usable || !(t[tp] == a[row])) return 0;
      used[tp++] = 1;
      s[row] = a[row];
    } else {
      repeated.push_back(row);
    }
  }
  for (size_t i = 0; i < usable; i++) {
    if (used[i]) continue;
    if (repeated.empty()) return 0;
    s[repeated.back()] = t[i];
    repeated.pop_back();
  }
  if (!repeated.empty()) return 0;
  for (size_t i = 0; i < usable; i++) {
    ((Fr*)out_a)[i] = Fr::from_raw(a[i].v);
    ((Fr*)out_s)[i] = Fr::from_raw(s[i].v);
  }
  return 1;
}

// z[0] = start, z[i] = z[i-1] * frac[i-1], i < count
void oracle_running_product(const uint64_t* start, const uint64_t* frac, size_t count, uint64_t* z) {
  Fr acc = *(const Fr*)start;
  const Fr* F_ = (const Fr*)frac;
  Fr* Z = (Fr*)z;
  for (size_t i = 0; i < count; i++) {
    Z[i] = acc;
    acc = acc * F_[i];
  }
}

// out[i] += s * a[i]
void oracle_fr_axpy(uint64_t* out, const uint64_t* a, const uint64_t* s, size_t n, int threads) {
  Fr sc = *(const Fr*)s;
  Fr* O = (Fr*)out;
  const Fr* A = (const Fr*)a;
#pragma omp parallel for schedule(static) num_threads(threads)
  for (size_t i = 0; i < n; i++) O[i] = O[i] + A[i] * sc;
}

// out[i] = a[i] * s + c
void oracle_fr_scale_add_const(uint64_t* out, const uint64_t* a, const uint64_t* s, const uint64_t* cst, size_t n, int threads) {
  Fr sc = *(const Fr*)s, cc = *(const Fr*)cst;
  Fr* O = (Fr*)out;
  const Fr* A = (const Fr*)a;
#pragma omp parallel for schedule(static) num_threads(threads)
  for (size_t i = 0; i < n; i++) O[i] = A[i] * sc + cc;
}

// The h(X) numerator on the extended coset, evaluation.rs evaluate_h order: gates, permutation, lookups,
// folded with y. All column arguments are extended-coset arrays of `rows` elements.
struct HArgs {
  const uint32_t* words;
  const uint32_t* offsets;     // expression e = words[offsets[e] .. offsets[e+1])
  uint32_t num_gates, num_lookups;
  const uint32_t* lookup_shape;  // (#inputs, #tables) per lookup; expressions follow the gates
  const uint64_t* const* fixed;
  const uint64_t* const* advice;
  const uint64_t* const* instance;
  const uint64_t* const* sigma;
  const uint64_t* const* zp;     // nsets
  const uint64_t* const* lz;
  const uint64_t* const* la;
  const uint64_t* const* ls;
  const uint64_t* l0;
  const uint64_t* l_last;
  const uint64_t* l_active;
  const uint64_t* xcoset;        // zeta * extended_omega^i
  const uint64_t* consts;
  const uint32_t* perm_cols;     // (kind, index) pairs
  uint32_t num_perm, nsets, chunk, blinding_factors;
  const uint64_t* beta;
  const uint64_t* gamma;
  const uint64_t* theta;
  const uint64_t* y;
  const uint64_t* delta;
  size_t rows;
  long rot_scale;
};

void oracle_evaluate_h(const HArgs* a, uint64_t* out, int threads) {
  Cols c{(const Fr* const*)a->fixed, (const Fr* const*)a->advice, (const Fr* const*)a->instance, (const Fr*)a->consts, a->rows - 1, a->rot_scale};
  const Fr beta = *(const Fr*)a->beta, gamma = *(const Fr*)a->gamma, theta = *(const Fr*)a->theta, y = *(const Fr*)a->y, delta = *(const Fr*)a->delta;
  const Fr one = Fr::one();
  const Fr *l0 = (const Fr*)a->l0, *llast = (const Fr*)a->l_last, *lact = (const Fr*)a->l_active, *xc = (const Fr*)a->xcoset;
  Fr* H = (Fr*)out;
  const long last_rot = -(long)(a->blinding_factors + 1);
  auto at = [&](const uint64_t* col, size_t row, long rot) -> Fr {
    return ((const Fr*)col)[(size_t)((long)row + rot * a->rot_scale) & (a->rows - 1)];
  };
  auto colval = [&](uint32_t j, size_t row) -> Fr {
    uint32_t kind = a->perm_cols[2 * j], idx = a->perm_cols[2 * j + 1];
    const uint64_t* p = kind == 0 ? a->advice[idx] : kind == 1 ? a->fixed[idx] : a->instance[idx];
    return ((const Fr*)p)[row];
  };
#pragma omp parallel num_threads(threads)
  {
    Fr st[64];
#pragma omp for schedule(static)
    for (size_t row = 0; row < a->rows; row++) {
      Fr h = Fr::zero();
      for (uint32_t g = 0; g < a->num_gates; g++)
        h = h * y + eval_expr(a->words + a->offsets[g], a->offsets[g + 1] - a->offsets[g], c, row, st);
      if (a->nsets) {
        h = h * y + (one - at(a->zp[0], row, 0)) * l0[row];
        Fr zl = at(a->zp[a->nsets - 1], row, 0);
        h = h * y + (zl * zl - zl) * llast[row];
        for (uint32_t s = 1; s < a->nsets; s++) h = h * y + (at(a->zp[s], row, 0) - at(a->zp[s - 1], row, last_rot)) * l0[row];
        Fr current_delta = beta * xc[row];
        for (uint32_t s = 0; s < a->nsets; s++) {
          uint32_t lo = s * a->chunk, hi = std::min(a->num_perm, lo + a->chunk);
          Fr left = at(a->zp[s], row, 1);
          for (uint32_t j = lo; j < hi; j++) left = left * (colval(j, row) + beta * ((const Fr*)a->sigma[j])[row] + gamma);
          Fr right = at(a->zp[s], row, 0);
          for (uint32_t j = lo; j < hi; j++) {
            right = right * (colval(j, row) + current_delta + gamma);
            current_delta = current_delta * delta;
          }
          h = h * y + (left - right) * lact[row];
        }
      }
      uint32_t e = a->num_gates;
      for (uint32_t l = 0; l < a->num_lookups; l++) {
        uint32_t ni = a->lookup_shape[2 * l], nt = a->lookup_shape[2 * l + 1];
        Fr ci = Fr::zero(), ct = Fr::zero();
        for (uint32_t i = 0; i < ni; i++) ci = ci * theta + eval_expr(a->words + a->offsets[e + i], a->offsets[e + i + 1] - a->offsets[e + i], c, row, st);
        for (uint32_t i = 0; i < nt; i++)
          ct = ct * theta + eval_expr(a->words + a->offsets[e + ni + i], a->offsets[e + ni + i + 1] - a->offsets[e + ni + i], c, row, st);
        e += ni + nt;
        Fr z = at(a->lz[l], row, 0), zn = at(a->lz[l], row, 1);
        Fr pa = at(a->la[l], row, 0), pap = at(a->la[l], row, -1), ps = at(a->ls[l], row, 0);
        h = h * y + (one - z) * l0[row];
        h = h * y + (z * z - z) * llast[row];
        h = h * y + (zn * (pa + beta) * (ps + gamma) - z * ((ci + beta) * (ct + gamma))) * lact[row];
        h = h * y + (pa - ps) * l0[row];
        h = h * y + (pa - ps) * (pa - pap) * lact[row];
      }
      H[row] = h;
    }
  }
}

}  // extern "C"
