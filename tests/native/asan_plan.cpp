// Host-only sanitizer harness for the tile schedule builder (cfs_plan.hpp is pure
// host C++): builds and decodes schedules of small synthetic matrices -- node
// blocks, bands, hub rows; natural / clustered order; HYB (far entries) and the
// deterministic layout; whole matrix, mirrored and exchange-form shards -- under
// AddressSanitizer + UBSan.  Compiled and run by
// tests/test_native_sanitizers.py (CPU only; GPU sanitizers are not available).
#include <cstdio>
#include <cstdlib>
#include <map>
#include <random>
#include <vector>

#include "cfs_plan.hpp"

struct Csr {
  int n;
  std::vector<int> rp, ci;
  std::vector<double> va;
};

static Csr make(int nodes, int dof, int kind, unsigned seed) {
  std::mt19937 g(seed);
  const int n = nodes * dof;
  std::vector<std::map<int, double>> rows(n);
  auto add = [&](int i, int j, double v) {
    if (i == j) return;
    rows[std::max(i, j)][std::min(i, j)] = v;
  };
  for (int a = 1; a < nodes; a++) {
    const int nb = kind == 0 ? 3 : 1 + (int)(g() % 6);
    for (int k = 0; k < nb; k++) {
      int b = kind == 2 && g() % 50 == 0 ? (int)(g() % a) : a - 1 - (int)(g() % std::min(a, 40));
      if (b < 0) b = 0;
      for (int p = 0; p < dof; p++)
        for (int q = 0; q < dof; q++) add(a * dof + p, b * dof + q, -1.0 + (g() % 1000) * 1e-3);
    }
    for (int p = 1; p < dof; p++)
      for (int q = 0; q < p; q++) add(a * dof + p, a * dof + q, 0.25);
  }
  if (kind == 2) // a hub row
    for (int j = 0; j < n - 1; j += 3) add(n - 1, j, 0.5);
  // full symmetric CSR with diagonal
  std::vector<std::map<int, double>> full(n);
  for (int i = 0; i < n; i++) {
    full[i][i] = 4.0 + i % 3;
    for (auto &e : rows[i]) {
      full[i][e.first] = e.second;
      full[e.first][i] = e.second;
    }
  }
  Csr A;
  A.n = n;
  A.rp.push_back(0);
  for (int i = 0; i < n; i++) {
    for (auto &e : full[i]) {
      A.ci.push_back(e.first);
      A.va.push_back(e.second);
    }
    A.rp.push_back((int)A.ci.size());
  }
  return A;
}

template <typename V> static long run(const Csr &A, int nranks, bool mirror, int flags, int slots, int block) {
  std::vector<V> va(A.va.begin(), A.va.end());
  std::vector<int> rs(nranks + 1);
  cfs_plan::balanced_splits(A.n, A.rp.data(), A.ci.data(), nranks, rs.data());
  long decoded = 0;
  for (int r = 0; r < nranks; r++) {
    cfs_plan::Options o;
    o.max_slots = slots;
    o.block_threads = block;
    o.reorder = !(flags & 8);
    if (flags & 16) o.force_order = 2;
    o.hyb = (flags & 128) != 0;          // far entries (Format::hyb)
    o.deterministic = (flags & 1024) != 0; // two integer words per y slot: smaller windows
    o.mirror_offblock = mirror;
    cfs_plan::SymPlan<V> P;
    if (!cfs_plan::build_plan<V>(A.n, A.rp.data(), A.ci.data(), va.data(), nranks, r,
                                 nranks > 1 ? rs.data() : nullptr, o, P)) {
      if (P.error.find("dense row") == std::string::npos) {
        fprintf(stderr, "unexpected refusal: %s\n", P.error.c_str());
        exit(3);
      }
      continue;
    }
    std::vector<int32_t> rr, cc, fr, fc, ur, uc;
    std::vector<V> vv, fv, uv;
    cfs_plan::decode_plan(P, rr, cc, vv, &fr, &fc, &fv, &ur, &uc, &uv);
    if ((int64_t)fr.size() != P.far_entries || fr.size() != ur.size()) {
      fprintf(stderr, "far entries %zu, mirror images %zu, expected %lld\n", fr.size(), ur.size(),
              (long long)P.far_entries);
      exit(5);
    }
    if ((int64_t)rr.size() != P.nnz_low + P.mirror_entries) {
      fprintf(stderr, "decoded %zu entries, expected %lld\n", rr.size(),
              (long long)(P.nnz_low + P.mirror_entries));
      exit(4);
    }
    decoded += (long)rr.size();
  }
  return decoded;
}

int main() {
  long total = 0;
  int cases = 0;
  for (int kind = 0; kind < 3; kind++)
    for (int dof : {1, 3, 7}) {
      const Csr A = make(700 / dof + 40, dof, kind, 17u * kind + dof);
      for (int flags : {0, 8, 16, 128, 128 | 8, 128 | 16, 1024})
        for (int nranks : {1, 3}) {
          total += run<double>(A, nranks, true, flags, kind == 2 ? 0 : 192, 256);
          total += run<float>(A, nranks, nranks > 1 ? false : true, flags, 0, 512);
          cases += 2;
        }
    }
  printf("asan_plan: %d cases, %ld entries decoded, OK\n", cases, total);
  return 0;
}
