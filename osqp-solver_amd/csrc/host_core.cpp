// host_core.cpp -- see host_core.hpp.  Citations: [REF] = /root/reference,
// "row Ex" = SURVEY.md section 8(a).
#include "host_core.hpp"
#include "sched_format.h"

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <numeric>
#include <set>
#include <chrono>
#include <thread>

#include "../../include/mi_osqp.h"

namespace miosqp {

int validate_settings(const Settings &s) {
  if (s.scaling < 0 || s.max_iter <= 0 || s.check_termination < 0) return 1;
  if (s.adaptive_rho != 0 && s.adaptive_rho != 1) return 1;
  if (s.adaptive_rho_interval < 0 || s.adaptive_rho_tolerance < 1.0) return 1;
  if (!(s.rho > 0.0) || !(s.sigma > 0.0)) return 1;
  if (s.eps_abs < 0.0 || s.eps_rel < 0.0 || (s.eps_abs == 0.0 && s.eps_rel == 0.0)) return 1;
  if (!(s.eps_prim_inf > 0.0) || !(s.eps_dual_inf > 0.0)) return 1;
  if (!(s.alpha > 0.0) || !(s.alpha < 2.0)) return 1;
  if ((s.scaled_termination | 1) != 1 || (s.warm_start | 1) != 1) return 1;
  return 0;
}

// ------------------------------------------------------------------ ordering

// Minimum-degree ordering on the explicit elimination graph (sorted adjacency
// vectors; degree buckets).  Plays the part AMD plays upstream (row E5); any
// permutation yields the same KKT solution up to round-off.
static void min_degree(int N, const std::vector<int> &Kp, const std::vector<int> &Ki,
                       std::vector<int> &perm) {
  std::vector<std::vector<int>> adj(N);
  for (int j = 0; j < N; j++)
    for (int k = Kp[j]; k < Kp[j + 1]; k++) {
      int i = Ki[k];
      if (i != j) { adj[i].push_back(j); adj[j].push_back(i); }
    }
  for (auto &a : adj) { std::sort(a.begin(), a.end()); a.erase(std::unique(a.begin(), a.end()), a.end()); }
  std::vector<char> dead(N, 0);
  // ordered by (degree, index): the smallest index among the nodes of minimum degree goes first
  std::set<std::pair<int, int>> pq;
  std::vector<int> deg(N);
  for (int v = 0; v < N; v++) { deg[v] = (int)adj[v].size(); pq.insert({deg[v], v}); }
  perm.assign(N, 0);
  std::vector<int> tmp;
  int alive = N;
  for (int step = 0; step < N && !pq.empty();) {
    const int v = pq.begin()->second;
    pq.erase(pq.begin());
    if ((int)adj[v].size() == alive - 1) {        // remaining graph is a clique
      perm[step++] = v; dead[v] = 1;
      for (int u = 0; u < N; u++) if (!dead[u]) { perm[step++] = u; dead[u] = 1; }
      break;
    }
    perm[step++] = v; dead[v] = 1; alive--;
    const std::vector<int> S = adj[v];
    for (int u : S) {
      tmp.clear();
      std::set_union(adj[u].begin(), adj[u].end(), S.begin(), S.end(), std::back_inserter(tmp));
      tmp.erase(std::remove_if(tmp.begin(), tmp.end(), [&](int x) { return x == u || x == v; }), tmp.end());
      adj[u].swap(tmp);
      pq.erase({deg[u], u});
      deg[u] = (int)adj[u].size();
      pq.insert({deg[u], u});
    }
    std::vector<int>().swap(adj[v]);
  }
}

// Nested dissection by BFS level structures (George-Liu style): find a pseudo-peripheral
// node, take a middle BFS level as separator, recurse on both sides, number the separator
// last.  On chain-/mesh-like KKT graphs (GOMP trajectories, grids) this gives an
// elimination tree of logarithmic depth, i.e. few device phases; on expander-like graphs it
// degenerates and minimum degree wins (analyze() picks by a cost model).
namespace {
struct NDCtx {
  const std::vector<std::vector<int>> &adj;
  std::vector<int> stamp, level;
  int cur = 0;
  std::vector<int> out;
  int leaf;
};

void nd_leaf(NDCtx &c, const std::vector<int> &nodes) {
  // minimum degree on the induced subgraph
  int k = (int)nodes.size();
  if (k <= 2) { for (int v : nodes) c.out.push_back(v); return; }
  int st = ++c.cur;
  std::vector<int> loc(k);
  for (int i = 0; i < k; i++) { c.stamp[nodes[i]] = st; c.level[nodes[i]] = i; }
  std::vector<int> Kp(k + 1, 0), Ki;
  for (int j = 0; j < k; j++) {
    for (int u : c.adj[nodes[j]]) if (c.stamp[u] == st && c.level[u] < j) Ki.push_back(c.level[u]);
    Ki.push_back(j);
    Kp[j + 1] = (int)Ki.size();
  }
  std::vector<int> perm;
  min_degree(k, Kp, Ki, perm);
  for (int i = 0; i < k; i++) c.out.push_back(nodes[perm[i]]);
}

void nd_rec(NDCtx &c, std::vector<int> nodes) {
  if ((int)nodes.size() <= c.leaf) { nd_leaf(c, nodes); return; }
  // connected components of the induced subgraph, all in one pass (removing a separator of a GOMP-like KKT graph
  // leaves thousands of single constraint nodes behind: peeling them off one per recursion level is quadratic),
  // numbered in the order of their first node
  int st = ++c.cur;
  for (int v : nodes) c.stamp[v] = st;
  {
    const int seen = ++c.cur;
    std::vector<std::vector<int>> comps;
    for (int v : nodes) {
      if (c.stamp[v] != st) continue;
      std::vector<int> q{v};
      c.stamp[v] = seen;
      for (size_t h = 0; h < q.size(); h++)
        for (int u : c.adj[q[h]]) if (c.stamp[u] == st) { c.stamp[u] = seen; q.push_back(u); }
      if (q.size() == nodes.size()) break;                 // connected: carry on below
      comps.push_back(std::move(q));
    }
    if (!comps.empty()) {
      for (auto &comp : comps) nd_rec(c, std::move(comp));       // each in BFS order from its first node
      return;
    }
    for (int v : nodes) c.stamp[v] = st;
  }
  // pseudo-peripheral node + level structure
  auto bfs = [&](int root, std::vector<std::vector<int>> &levels) {
    int seen = ++c.cur;
    levels.clear();
    levels.push_back({root}); c.stamp[root] = seen; c.level[root] = 0;
    while (true) {
      std::vector<int> nxt;
      for (int v : levels.back())
        for (int u : c.adj[v]) if (c.stamp[u] == st) { c.stamp[u] = seen; c.level[u] = (int)levels.size(); nxt.push_back(u); }
      if (nxt.empty()) break;
      levels.push_back(std::move(nxt));
    }
    for (auto &lv : levels) for (int v : lv) c.stamp[v] = st;
  };
  std::vector<std::vector<int>> levels;
  int root = nodes[0];
  for (int rep = 0; rep < 4; rep++) {
    bfs(root, levels);
    int best = levels.back()[0];
    for (int v : levels.back()) if (c.adj[v].size() < c.adj[best].size()) best = v;
    if (best == root) break;
    size_t depth = levels.size();
    std::vector<std::vector<int>> l2;
    bfs(best, l2);
    if (l2.size() <= depth) { if (l2.size() == depth) { levels.swap(l2); root = best; } break; }
    levels.swap(l2); root = best;
  }
  int k = (int)levels.size();
  if (k < 4) { nd_leaf(c, nodes); return; }
  // separator: the smallest level in the middle third (by cumulative size)
  size_t total = nodes.size(), cum = 0;
  int j = -1; size_t bestsz = (size_t)-1;
  for (int i = 0; i < k; i++) {
    if (i >= 1 && i <= k - 2 && cum >= total / 3 && cum + levels[i].size() <= total - total / 3 + levels[i].size()) {
      if (levels[i].size() < bestsz) { bestsz = levels[i].size(); j = i; }
    }
    cum += levels[i].size();
  }
  if (j < 0) j = k / 2;
  std::vector<int> left, right, sep;
  for (int i = 0; i < j; i++) left.insert(left.end(), levels[i].begin(), levels[i].end());
  for (int i = j + 1; i < k; i++) right.insert(right.end(), levels[i].begin(), levels[i].end());
  for (int v : levels[j]) {
    bool touches_right = false;
    for (int u : c.adj[v]) if (c.stamp[u] == st && c.level[u] == j + 1) { touches_right = true; break; }
    (touches_right ? sep : left).push_back(v);
  }
  if (left.empty() || right.empty()) { nd_leaf(c, nodes); return; }
  nd_rec(c, std::move(left));
  nd_rec(c, std::move(right));
  for (int v : sep) c.out.push_back(v);
}
}  // namespace

// leaf: subgraphs of at most that many nodes are ordered by minimum degree.  Small leaves = a deeper dissection = fewer
// dependent chunk levels on chain-like graphs (a GOMP KKT of 60 waypoints: 72 phases per iteration with leaves of 48 nodes, 43
// with leaves of 8, about the same fill); analyze() tries several sizes and keeps the cheapest by its cost model.
static void nested_dissection(int N, const std::vector<int> &Kp, const std::vector<int> &Ki, std::vector<int> &perm, int leaf = 48) {
  std::vector<std::vector<int>> adj(N);
  for (int j = 0; j < N; j++)
    for (int k = Kp[j]; k < Kp[j + 1]; k++) { int i = Ki[k]; if (i != j) { adj[i].push_back(j); adj[j].push_back(i); } }
  for (auto &a : adj) { std::sort(a.begin(), a.end()); a.erase(std::unique(a.begin(), a.end()), a.end()); }
  NDCtx c{adj, std::vector<int>(N, 0), std::vector<int>(N, 0), 0, {}, leaf};
  std::vector<int> all(N);
  std::iota(all.begin(), all.end(), 0);
  nd_rec(c, all);
  perm = c.out;
}

// lower-triangular CSC of the permuted KKT with a natural->permuted entry map
static void build_permuted_lower(Analysis &an) {
  int N = an.N, nnz = an.nnzK();
  std::vector<int> cnt(N + 1, 0);
  std::vector<int> colOf(nnz), rowOf(nnz);
  for (int j = 0; j < N; j++)
    for (int k = an.Kp[j]; k < an.Kp[j + 1]; k++) {
      int a = an.pinv[an.Ki[k]], b = an.pinv[j];
      colOf[k] = std::min(a, b); rowOf[k] = std::max(a, b);
      cnt[colOf[k]]++;
    }
  an.Klp.assign(N + 1, 0);
  for (int j = 0; j < N; j++) an.Klp[j + 1] = an.Klp[j] + cnt[j];
  // sort entries of each column by row so that symbolic merges see sorted lists
  std::vector<int> order(nnz);
  std::iota(order.begin(), order.end(), 0);
  std::sort(order.begin(), order.end(), [&](int a, int b) {
    if (colOf[a] != colOf[b]) return colOf[a] < colOf[b];
    return rowOf[a] < rowOf[b];
  });
  an.Kli.assign(nnz, 0); an.KtoKl.assign(nnz, 0);
  for (int pos = 0; pos < nnz; pos++) { int k = order[pos]; an.Kli[pos] = rowOf[k]; an.KtoKl[k] = pos; }
}

// column structures of L (sorted) + elimination tree
static void symbolic(const Analysis &an, std::vector<std::vector<int>> &cols, std::vector<int> &parent) {
  int N = an.N;
  cols.assign(N, {});
  parent.assign(N, -1);
  std::vector<std::vector<int>> children(N);
  std::vector<int> tmp;
  for (int j = 0; j < N; j++) {
    std::vector<int> &s = cols[j];
    for (int p = an.Klp[j]; p < an.Klp[j + 1]; p++) if (an.Kli[p] > j) s.push_back(an.Kli[p]);
    for (int c : children[j]) {
      tmp.clear();
      const std::vector<int> &cs = cols[c];
      // cs[0] == j ; merge the remainder
      std::set_union(s.begin(), s.end(), cs.begin() + 1, cs.end(), std::back_inserter(tmp));
      s.swap(tmp);
    }
    if (!s.empty()) { parent[j] = s[0]; children[s[0]].push_back(j); }
  }
}

static void postorder(const std::vector<int> &parent, std::vector<int> &post) {
  int N = (int)parent.size();
  std::vector<std::vector<int>> children(N);
  std::vector<int> roots;
  for (int j = 0; j < N; j++) { if (parent[j] >= 0) children[parent[j]].push_back(j); else roots.push_back(j); }
  post.clear(); post.reserve(N);
  std::vector<std::pair<int, size_t>> stack;
  for (int r : roots) {
    stack.push_back({r, 0});
    while (!stack.empty()) {
      auto &[v, ci] = stack.back();
      if (ci < children[v].size()) { int c = children[v][ci++]; stack.push_back({c, 0}); }
      else { post.push_back(v); stack.pop_back(); }
    }
  }
}

// ------------------------------------------------------------ schedule packing

namespace {
struct RowWork { uint32_t row; std::vector<std::pair<uint32_t, int32_t>> ent; };   // ent: (gather index, value source)
// One barrier interval ("phase") of a schedule.  shared[kind]: rows dealt over all waves by load (kind 0: the flush subtracts
// from the row, kind 1: it stores the sum); the rows of one phase never depend on each other across waves.  wave[w]: groups
// of rows that stay on wave w, executed in this order AFTER its share of the shared rows - a later group may gather what
// an earlier group of the same wave has just written (LDS operations of one wave execute in order: no barrier needed).
struct RowGroup { int kind; std::vector<RowWork> rows; };
struct LevelWork { std::vector<RowWork> shared[2]; std::vector<std::vector<RowGroup>> wave; };

int ilog2(int v) { int l = 0; while ((1 << l) < v) l++; return l; }

// Builds one schedule.  Work is first laid out as phases (one per non-empty level, spread over the waves); the steps
// are then numbered WAVE-MAJOR: wave w's steps of all phases are contiguous in memory and form the linear stream the
// device walks (sched_format.h); phase boundaries survive only as barrier counts in the descriptors.
void pack_schedule(const std::vector<LevelWork> &levels, Schedule &sch, int nw, int bt, bool barriers, bool wide,
                   bool dataflow = false, uint32_t shadow = 0, uint32_t pad = 0) {
  sch = Schedule();
  sch.n_levels = (int)levels.size();
  sch.nw = nw; sch.bt = bt; sch.barriers = barriers;
  sch.dataflow = dataflow; sch.shadow = shadow; sch.pad = pad;
  // a unit = work that must stay on one wave, in order
  struct StepSpec { int lt; bool flush; std::vector<const RowWork *> rows; int first_entry; int kind; bool group_start; };
  struct Unit { std::vector<StepSpec> steps; };
  struct PhaseRec { int kind; std::vector<std::vector<StepSpec>> wave; };
  std::vector<PhaseRec> phases;
  // rows -> units (steps that stay together on one wave)
  auto make_units = [&](const std::vector<RowWork> &rows, int kind, std::vector<Unit> &units) {
    std::vector<const RowWork *> longs;
    // short rows (<= 64 entries): lane-group width T and step count S <= 4 chosen to minimise the
    // padded slots T*S (ties -> fewer steps); rows with equal (T,S) are packed 64/T per unit
    std::vector<std::vector<const RowWork *>> byTS(7 * 4);
    for (const RowWork &rw : rows) {
      int len = std::max<int>(1, (int)rw.ent.size());
      if (len > 64) { longs.push_back(&rw); continue; }
      if (kind == 1) { byTS[(len <= kChunk / 2 ? ilog2(kChunk) - 1 : ilog2(kChunk)) * 4].push_back(&rw); continue; }   // phase B: one step per
                                                                               // row; rows of <= 8 entries share an 8-lane group (3 instead of 4 steps per 16-row chunk)
      int bestlt = 6, bestS = 1, bestcost = 1 << 30;
      for (int lt = 0; lt <= 6; lt++) {
        int T = 1 << lt, S = (len + T - 1) / T;
        if (S > 4) continue;
        int cost = T * S * 4 + S;          // slots first, then steps
        if (cost < bestcost) { bestcost = cost; bestlt = lt; bestS = S; }
      }
      byTS[bestlt * 4 + (bestS - 1)].push_back(&rw);
    }
    std::stable_sort(longs.begin(), longs.end(),
                     [](const RowWork *a, const RowWork *b) { return a->ent.size() > b->ent.size(); });
    // long rows: rpt rows share a wave (T = 64/rpt lanes each) so that the level needs <= nw long units
    int rpt = 1;
    while (rpt < 8 && (int)longs.size() > nw * rpt) rpt <<= 1;
    for (size_t i = 0; i < longs.size(); i += rpt) {
      std::vector<const RowWork *> grp(longs.begin() + i, longs.begin() + std::min(longs.size(), i + rpt));
      int T = 64 / rpt, S = 0;
      for (const RowWork *r : grp) S = std::max(S, ((int)r->ent.size() + T - 1) / T);
      Unit u;
      for (int st = 0; st < S; st++) u.steps.push_back({ilog2(T), st == S - 1, grp, st * T, kind, false});
      units.push_back(std::move(u));
    }
    for (int lt = 6; lt >= 0; lt--)
      for (int S = 4; S >= 1; S--) {
        int T = 1 << lt, per = 64 / T;
        const auto &v = byTS[lt * 4 + (S - 1)];
        for (size_t i = 0; i < v.size(); i += per) {
          std::vector<const RowWork *> grp(v.begin() + i, v.begin() + std::min(v.size(), i + per));
          Unit u;
          for (int st = 0; st < S; st++) u.steps.push_back({lt, st == S - 1, grp, st * T, kind, false});
          units.push_back(std::move(u));
        }
      }
  };
  for (size_t L = 0; L < levels.size(); L++) {
    sch.level_first_phase.push_back((int)phases.size());
    const LevelWork &lv = levels[L];
    bool any = !lv.shared[0].empty() || !lv.shared[1].empty();
    for (const auto &wg : lv.wave) for (const RowGroup &g : wg) any = any || !g.rows.empty();
    if (!any) continue;
    PhaseRec ph; ph.kind = lv.shared[0].empty() && !lv.shared[1].empty() ? 1 : 0; ph.wave.resize(nw);
    std::vector<int> load(nw, 0);
    // the rows that stay on their wave count as load before the shared rows are dealt
    std::vector<std::vector<std::vector<Unit>>> own(nw);
    for (int w = 0; w < nw && w < (int)lv.wave.size(); w++)
      for (const RowGroup &g : lv.wave[w]) {
        own[w].emplace_back();
        make_units(g.rows, g.kind, own[w].back());
        for (const Unit &u : own[w].back()) load[w] += (int)u.steps.size();
      }
    for (int kind = 0; kind < 2; kind++) {
      if (lv.shared[kind].empty()) continue;
      std::vector<Unit> units;
      make_units(lv.shared[kind], kind, units);
      // longest-processing-time assignment of units to waves
      std::vector<size_t> order(units.size());
      std::iota(order.begin(), order.end(), 0);
      std::stable_sort(order.begin(), order.end(), [&](size_t a, size_t b) { return units[a].steps.size() > units[b].steps.size(); });
      std::vector<std::vector<size_t>> mine(nw);
      for (size_t k : order) {
        int w = (int)(std::min_element(load.begin(), load.end()) - load.begin());
        load[w] += (int)units[k].steps.size();
        mine[w].push_back(k);
      }
      for (int w = 0; w < nw; w++) {
        bool first = true;
        for (size_t k : mine[w])
          for (StepSpec &st : units[k].steps) { st.group_start = first; first = false; ph.wave[w].push_back(std::move(st)); }
      }
    }
    for (int w = 0; w < nw; w++)
      for (auto &units : own[w]) {
        bool first = true;
        for (Unit &u : units)
          for (StepSpec &st : u.steps) { st.group_start = first; first = false; ph.wave[w].push_back(std::move(st)); }
      }
    phases.push_back(std::move(ph));
  }
  sch.level_first_phase.push_back((int)phases.size());
  sch.n_phases = (int)phases.size();
  // ---- wave-major numbering
  sch.phase.assign((size_t)sch.n_phases * sch.phase_stride(), 0u);
  sch.lvl_pos.assign((size_t)(levels.size() + 1) * nw, 0u);
  sch.tail_bar.assign(nw, 0u);
  sch.wave_range.assign(2 * (size_t)nw, 0u);
  for (int w = 0; w < nw; w++) {
    sch.wave_range[2 * w] = sch.n_steps;
    int last_phase = 0;        // phase of the wave's previous step: barriers owed = phase boundaries crossed since
    size_t L = 0;
    std::vector<uint32_t> group_starts;      // steps that open a group inside a phase: what they gather may have just been written
    for (int p = 0; p < sch.n_phases; p++) {
      while (L < levels.size() && sch.level_first_phase[L] == p) sch.lvl_pos[L++ * nw + w] = sch.n_steps;   // (empty levels share a phase index)
      uint32_t *e = &sch.phase[(size_t)p * sch.phase_stride() + 1 + 4 * w];
      sch.phase[(size_t)p * sch.phase_stride()] = (uint32_t)phases[p].kind;
      e[0] = sch.n_steps;
      const std::vector<StepSpec> &recs = phases[p].wave[w];
      for (size_t k = 0; k < recs.size(); k++) {
        const StepSpec &st = recs[k];
        const uint32_t stepno = sch.n_steps++;
        sch.idx.resize((size_t)sch.n_steps * 64, pad);
        sch.src.resize((size_t)sch.n_steps * 64, MI_SRC_ZERO);
        uint32_t nbar = 0;
        if (barriers && k == 0) { nbar = (uint32_t)(p - last_phase); last_phase = p; }
        uint32_t d = (nbar << 12) | (uint32_t)st.lt | (st.kind == 1 ? MI_D_STORE : 0u), ob = 0;
        if (st.group_start && k > 0) group_starts.push_back(stepno);
        const int T = 1 << st.lt;
        for (int g = 0; g < (int)st.rows.size(); g++) {
          const auto &ent = st.rows[g]->ent;
          for (int en = st.first_entry; en < std::min<int>((int)ent.size(), st.first_entry + T); en++) {
            const uint32_t slot = stepno * 64u + (uint32_t)(g * T + (en - st.first_entry));
            sch.idx[slot] = ent[en].first; sch.src[slot] = ent[en].second;
          }
        }
        if (st.flush) {
          ob = (uint32_t)sch.outA.size();
          for (int g = 0; g < 64 / T; g++) sch.outA.push_back(g < (int)st.rows.size() ? st.rows[g]->row : kNoRow);
          d |= MI_D_FLUSH;
        }
        sch.step.push_back(d);
        sch.step_ob.push_back(ob);
      }
      e[1] = sch.n_steps;
    }
    while (L < levels.size()) sch.lvl_pos[L++ * nw + w] = sch.n_steps;
    sch.lvl_pos[levels.size() * nw + w] = sch.n_steps;
    sch.wave_range[2 * w + 1] = sch.n_steps;
    // gather look-ahead flags: a step may issue the gather of the step MI_D_LOOKAHEAD positions later when
    // no barrier and no level boundary (sub-range walks start there) lies in between
    {
      const uint32_t b0 = sch.wave_range[2 * w], e0 = sch.n_steps;
      std::vector<char> level_start(e0 - b0 + 1, 0);
      for (size_t l2 = 0; l2 <= levels.size(); l2++) level_start[sch.lvl_pos[l2 * nw + w] - b0] = 1;
      for (uint32_t q : group_starts) level_start[q - b0] = 1;      // (no gather is issued ahead across a group boundary)
      auto eligible = [&](uint32_t q) { return MI_D_NBAR(sch.step[q]) == 0 && !level_start[q - b0]; };
      for (uint32_t q = b0; q + MI_D_LOOKAHEAD < e0; q++) {
        bool okk = true;
        for (uint32_t j = 1; j <= MI_D_LOOKAHEAD; j++) okk = okk && eligible(q + j);
        if (okk) { sch.step[q] |= MI_D_AHEAD; sch.step[q + MI_D_LOOKAHEAD] |= MI_D_PRE; }
      }
    }
    // every wave passes exactly n_phases barriers per walk (the last one publishes the final phase)
    sch.tail_bar[w] = barriers ? (uint32_t)(sch.n_phases - last_phase) : 0u;
  }
  sch.n_slots = sch.n_steps * 64u;
  // device index words
  if (wide) sch.idxw64.assign((size_t)sch.n_steps * 64, 0xFFFFFFFF00000000ull);
  else sch.idxw.assign((size_t)sch.n_steps * 64, 0xFFFF0000u);
  for (uint32_t st = 0; st < sch.n_steps; st++) {
    const uint32_t d = sch.step[st], lt = MI_D_LT(d), ob = sch.step_ob[st];
    for (uint32_t ln = 0; ln < 64; ln++) {
      uint32_t row = kNoRow;
      if (d & MI_D_FLUSH) row = sch.outA[ob + (ln >> lt)];
      const size_t slot = (size_t)st * 64 + ln;
      if (wide) sch.idxw64[slot] = (uint64_t)sch.idx[slot] | ((uint64_t)row << 32);
      else sch.idxw[slot] = (sch.idx[slot] & 0xFFFFu) | ((row == kNoRow ? 0xFFFFu : row) << 16);
    }
  }
  if (getenv("MI_OSQP_DEBUG_ORDER")) {
    fprintf(stderr, "[mi_osqp] schedule: levels %zu phases %zu steps %u outA %zu\n", levels.size(), phases.size(), sch.n_steps,
            sch.outA.size());
    {   // critical path: the longest wave of every phase
      uint64_t crit = 0; uint32_t worst = 0; int over8 = 0;
      for (int p = 0; p < sch.n_phases; p++) {
        uint32_t mx = 0;
        for (int w = 0; w < nw; w++) { const uint32_t *e = &sch.phase[(size_t)p * sch.phase_stride() + 1 + 4 * w]; mx = std::max(mx, e[1] - e[0]); }
        crit += mx; worst = std::max(worst, mx); over8 += mx > 8;
      }
      fprintf(stderr, "[mi_osqp]   critical path %llu steps over %d phases (longest phase %u steps, %d phases above 8 steps), %d waves\n",
              (unsigned long long)crit, sch.n_phases, worst, over8, nw);
    }
    uint32_t cnt[7][2] = {};
    for (uint32_t st = 0; st < sch.n_steps; st++) cnt[MI_D_LT(sch.step[st])][(sch.step[st] & MI_D_FLUSH) ? 1 : 0]++;
    for (int lt = 0; lt < 7; lt++) fprintf(stderr, "[mi_osqp]   lane groups of %2d: %u flush steps, %u other steps\n", 1 << lt, cnt[lt][1], cnt[lt][0]);
  }
}
}  // namespace

// The plain form: per level of the chunk dependency graph one phase for the A rows and one for the B rows of its chunks,
// all rows dealt over the waves.
static std::vector<LevelWork> plain_levels(const std::vector<int> &order, std::vector<std::vector<RowWork>> &rowsA,
                                           std::vector<std::vector<RowWork>> &rowsB, const std::vector<int> &lev, int maxlev) {
  std::vector<LevelWork> lw(2 * (size_t)(maxlev + 1));
  for (int c : order) {
    for (RowWork &r : rowsA[c]) lw[2 * (size_t)lev[c]].shared[0].push_back(std::move(r));
    for (RowWork &r : rowsB[c]) lw[2 * (size_t)lev[c] + 1].shared[1].push_back(std::move(r));
  }
  return lw;
}

static void build_tri_schedules(Analysis &an, int nw, int bt, bool df) {
  int N = an.N;
  int nch = (int)an.chunk_start.size() - 1;
  std::vector<int> chunk_of(N);
  for (int c = 0; c < nch; c++) for (int j = an.chunk_start[c]; j < an.chunk_start[c + 1]; j++) chunk_of[j] = c;
  // inverted diagonal blocks and the second vector position of the rows of multi-row chunks
  // Dense tail (host_core.hpp DenseTail): the rows from ts on keep their couplings to the columns before ts
  // (phase A of the forward sweep, gathers of the backward sweep) and nothing else - no in-chunk triangle, no second
  // vector position, no couplings among themselves: x_tail = S^-1 t_tail happens between the two sweeps.
  const int ts = an.dt.k ? an.dt.s : N;
  an.inv_off.assign(nch, -1); an.n_inv = 0;
  an.xloc.resize(N); an.Next = N;
  for (int c = 0; c < nch; c++) {
    int c0 = an.chunk_start[c], r = an.chunk_start[c + 1] - c0;
    const bool multi = r >= 2 && c0 < ts;
    if (multi) { an.inv_off[c] = an.n_inv; an.n_inv += r * (r - 1) / 2; }
    for (int i = 0; i < r; i++) an.xloc[c0 + i] = multi ? an.Next++ : c0 + i;
  }
  an.df = df;
  an.xs_total = df ? 2 * an.Next + 8 : an.Next;
  an.rflag.assign(df ? N : 0, 0);
  if (df) {
    for (int c = 0; c < nch; c++) {
      const int c0 = an.chunk_start[c], c1 = an.chunk_start[c + 1];
      for (int i = c0; i < c1; i++) {
        uint8_t f = (c1 - c0 >= 2 && c0 < ts) ? 4 : 0;
        for (int t = an.Rp[i]; t < an.Rp[i + 1] && !(f & 1); t++) if (an.Rj[t] < c0 && an.Rj[t] < ts) f |= 1;
        for (int q = an.Lp[i]; q < an.Lp[i + 1] && !(f & 2); q++) if (an.Li[q] >= c1) f |= 2;
        an.rflag[i] = f;
      }
    }
  }
  const uint32_t SH = (uint32_t)an.Next, PAD = df ? 2u * (uint32_t)an.Next : 0u;
  // (dataflow form: where a gather finds the FINAL value of the sweep's intermediate results)
  auto f_t = [&](int k) { return (uint32_t)(df && (an.rflag[k] & 1) ? k + (int)SH : k); };                      // forward: t_k
  auto f_x = [&](int j) { return (uint32_t)(df ? an.df_floc(j) : an.xloc[j]); };                                // forward: result of row j
  auto b_t = [&](int k) { return (uint32_t)(df && (an.rflag[k] & 2) ? an.xloc[k] + (int)SH : an.xloc[k]); };    // backward: t_k
  auto b_x = [&](int j) { return (uint32_t)(df ? an.df_bloc(j) : j); };                                         // backward: result of row j
  // ---- forward: rows ascending, sources are columns j < row.  Row i of a chunk: phase A subtracts the
  // couplings to earlier chunks in place (position i); phase B stores inv(L_cc) t at position xloc[i],
  // which is where every later row gathers it.
  {
    std::vector<int> lev(nch, 0);
    int maxlev = 0;
    for (int c = 0; c < nch; c++) {
      int c0 = an.chunk_start[c], c1 = an.chunk_start[c + 1], L = 0;
      for (int i = c0; i < c1; i++)
        for (int t = an.Rp[i]; t < an.Rp[i + 1]; t++) { int j = an.Rj[t]; if (j < c0 && j < ts) L = std::max(L, lev[chunk_of[j]] + 1); }
      lev[c] = L; maxlev = std::max(maxlev, L);
    }
    an.chunk_lev = lev;
    // per chunk: its A rows (couplings to earlier chunks, subtracted in place) and B rows (product with the inverted block)
    std::vector<std::vector<RowWork>> rowsA(nch), rowsB(nch);
    for (int c = 0; c < nch; c++) {
      int c0 = an.chunk_start[c], c1 = an.chunk_start[c + 1], r = c1 - c0;
      for (int i = c0; i < c1; i++) {
        RowWork rw; rw.row = (uint32_t)i;
        for (int t = an.Rp[i]; t < an.Rp[i + 1]; t++) {
          int j = an.Rj[t];
          if (j < c0 && j < ts) rw.ent.push_back({f_x(j), an.Rpos[t]});
        }
        if (!rw.ent.empty()) rowsA[c].push_back(std::move(rw));
        if (r >= 2 && c0 < ts) {
          RowWork rb; rb.row = (uint32_t)an.xloc[i];
          for (int k = c0; k < i; k++) rb.ent.push_back({f_t(k), an.inv_index(c, i - c0, k - c0)});
          rb.ent.push_back({f_t(i), MI_SRC_ONE});
          rowsB[c].push_back(std::move(rb));
        }
      }
    }
    std::vector<int> order(nch);
    std::iota(order.begin(), order.end(), 0);
    std::vector<LevelWork> lw = plain_levels(order, rowsA, rowsB, lev, maxlev);
    pack_schedule(lw, an.fwd, nw, bt, !df, an.wide, df, SH, PAD);
  }
  // ---- backward: columns descending, sources are rows j > column.  Column k of a chunk: phase A works in
  // place at xloc[k] (where the scaled forward result lives), phase B stores inv(L_cc)' t at position k.
  {
    std::vector<int> lev(nch, 0);
    int maxlev = 0;
    for (int c = nch - 1; c >= 0; c--) {
      int c0 = an.chunk_start[c], c1 = an.chunk_start[c + 1], L = 0;
      if (c0 >= ts) { lev[c] = -1; continue; }             // tail rows are final before the sweep starts
      for (int col = c0; col < c1; col++)
        for (int p = an.Lp[col]; p < an.Lp[col + 1]; p++) { int j = an.Li[p]; if (j >= c1) L = std::max(L, lev[chunk_of[j]] + 1); }
      lev[c] = L; maxlev = std::max(maxlev, L);
    }
    std::vector<std::vector<RowWork>> rowsA(nch), rowsB(nch);
    std::vector<int> order;
    for (int c = nch - 1; c >= 0; c--) {
      int c0 = an.chunk_start[c], c1 = an.chunk_start[c + 1], r = c1 - c0;
      if (c0 >= ts) continue;
      order.push_back(c);
      for (int col = c1 - 1; col >= c0; col--) {
        RowWork rw; rw.row = (uint32_t)an.xloc[col];
        for (int p = an.Lp[col]; p < an.Lp[col + 1]; p++) {
          int j = an.Li[p];
          if (j >= c1) rw.ent.push_back({b_x(j), p});
        }
        if (!rw.ent.empty()) rowsA[c].push_back(std::move(rw));
        if (r >= 2) {
          RowWork rb; rb.row = (uint32_t)col;
          rb.ent.push_back({b_t(col), MI_SRC_ONE});
          for (int i = col + 1; i < c1; i++) rb.ent.push_back({b_t(i), an.inv_index(c, i - c0, col - c0)});
          rowsB[c].push_back(std::move(rb));
        }
      }
    }
    std::vector<LevelWork> lw = plain_levels(order, rowsA, rowsB, lev, maxlev);
    pack_schedule(lw, an.bwd, nw, bt, !df, an.wide, df, SH, PAD);
  }
}

// check-SpMV: rows 0..n-1 = P x, n..2n-1 = A' y, 2n..2n+m-1 = A x; gather vector
// is [x ; y]; values come from the combined array [P triu | A].
static void build_chk_schedule(Analysis &an, int nw, int bt) {
  int n = an.n, m = an.m, nnzP = an.Pp[n];
  std::vector<LevelWork> lw(3);
  std::vector<RowWork> px(n), aty(n), ax(m);
  for (int i = 0; i < n; i++) { px[i].row = (uint32_t)i; aty[i].row = (uint32_t)(n + i); }
  for (int r = 0; r < m; r++) ax[r].row = (uint32_t)(2 * n + r);
  for (int c = 0; c < n; c++)
    for (int k = an.Pp[c]; k < an.Pp[c + 1]; k++) {
      int r = an.Pi[k];
      px[r].ent.push_back({(uint32_t)c, k});
      if (r != c) px[c].ent.push_back({(uint32_t)r, k});
    }
  for (int c = 0; c < n; c++)
    for (int k = an.Ap[c]; k < an.Ap[c + 1]; k++) {
      int r = an.Ai[k];
      aty[c].ent.push_back({(uint32_t)(n + r), nnzP + k});
      ax[r].ent.push_back({(uint32_t)c, nnzP + k});
    }
  lw[0].shared[0] = std::move(px); lw[1].shared[0] = std::move(aty); lw[2].shared[0] = std::move(ax);
  pack_schedule(lw, an.chk, nw, bt, false, an.wide);
}


// ------------------------------------------------ dense tail (host_core.hpp DenseTail)

static void build_dense_tail(Analysis &an, int nw) {
  DenseTail &dt = an.dt;
  if (!dt.k) return;
  const int k = dt.k, nb = k / 64;
  dt.nb = nb; dt.nw = nw;
  struct T { int I, J; uint32_t flags; };
  std::vector<std::vector<T>> phases;
  {
    std::vector<T> p0;
    for (int J = 0; J < nb; J++) p0.push_back({J, J, DT_DIAG});
    phases.push_back(std::move(p0));
  }
  // shift p pairs with shift nb - p: the first group adds its row sums to y_r and its column sums to y_c, the second
  // the other way round, which keeps the (vector, 64-row part) targets of one phase disjoint
  for (int p = 1; 2 * p <= nb; p++) {
    std::vector<T> ph;
    for (int J = 0; J + p < nb; J++) ph.push_back({J + p, J, 0u});
    if (2 * p != nb) for (int J = 0; J < p; J++) ph.push_back({J + nb - p, J, DT_SWAP});
    if (!ph.empty()) phases.push_back(std::move(ph));
  }
  dt.n_phases = (int)phases.size();
  dt.task.clear(); dt.src.clear();
  dt.wave_task.assign(nw + 1, 0u); dt.wave_step.assign(nw + 1, 0u); dt.tail_bar.assign(nw, 0u);
  dt.n_steps = 0;
  for (int w = 0; w < nw; w++) {
    dt.wave_task[w] = (uint32_t)(dt.task.size() / 4); dt.wave_step[w] = dt.n_steps;
    int last_phase = 0;
    for (int ph = 0; ph < dt.n_phases; ph++) {
      const std::vector<T> &v = phases[ph];
      bool first = true;
      for (size_t t = 0; t < v.size(); t++) {
        if ((int)((t + (size_t)ph) % (size_t)nw) != w) continue;       // round robin, rotated per phase
        const bool diag = v[t].flags & DT_DIAG;
        const uint32_t nsteps = diag ? 32u : 64u, s0 = diag ? 1u : 0u;
        uint32_t nbar = 0;
        if (first) { nbar = (uint32_t)(ph - last_phase); last_phase = ph; first = false; }
        dt.task.insert(dt.task.end(), {(uint32_t)v[t].I * 64u, (uint32_t)v[t].J * 64u, v[t].flags | (nbar << 8), nsteps});
        for (uint32_t q = 0; q < nsteps; q++)
          for (uint32_t ln = 0; ln < 64; ln++) {
            const uint32_t sh = s0 + q;
            int i = v[t].I * 64 + (int)((ln + sh) % 64u), j = v[t].J * 64 + (int)ln;
            int32_t src;
            if (diag && sh == 32u && ln >= 32u) src = MI_SRC_ZERO;      // the pairs at distance 32 appear twice
            else { if (i < j) std::swap(i, j); src = j * k + i; }
            dt.src.push_back(src);
          }
        dt.n_steps += nsteps;
      }
    }
    dt.tail_bar[w] = (uint32_t)(dt.n_phases - last_phase);
  }
  dt.wave_task[nw] = (uint32_t)(dt.task.size() / 4); dt.wave_step[nw] = dt.n_steps;
}

// The product exactly as the device evaluates it: per task 64 column sums that stay in their lane, 64 row sums that
// rotate; M: k x k column-major (lower triangle read), mdiag: diagonal, xt: t in, x out.
bool replay_dense_tail(const DenseTail &dt, const double *M, const double *mdiag, double *xt) {
  const int k = dt.k, nb = dt.nb;
  std::vector<double> yr(k), yc(k, 0.0);
  for (int i = 0; i < k; i++) yr[i] = mdiag[i] * xt[i];
  bool ok = true;
  std::vector<int> touched_r(nb), touched_c(nb);
  std::vector<uint32_t> pos(dt.nw), epoch(dt.nw, 0u), slot(dt.nw);
  for (int w = 0; w < dt.nw; w++) { pos[w] = dt.wave_task[w]; slot[w] = dt.wave_step[w] * 64u; }
  for (int cur = 0; cur < dt.n_phases; cur++) {
    std::fill(touched_r.begin(), touched_r.end(), 0); std::fill(touched_c.begin(), touched_c.end(), 0);
    for (int w = 0; w < dt.nw; w++)
      while (pos[w] < dt.wave_task[w + 1]) {
        const uint32_t *t = &dt.task[4 * (size_t)pos[w]];
        const uint32_t nbar = t[2] >> 8, flags = t[2] & 255u, nsteps = t[3];
        if ((int)(epoch[w] + nbar) > cur) break;
        epoch[w] += nbar;
        const uint32_t I0 = t[0], J0 = t[1], s0 = (flags & DT_DIAG) ? 1u : 0u;
        double accr[64] = {0.0}, accc[64] = {0.0};      // accr indexed by the row inside the block
        for (uint32_t q = 0; q < nsteps; q++)
          for (uint32_t ln = 0; ln < 64; ln++) {
            const int32_t sc = dt.src[slot[w] + q * 64u + ln];
            const double v = sc == MI_SRC_ZERO ? 0.0 : M[sc];
            const uint32_t il = (ln + s0 + q) % 64u;
            accc[ln] = std::fma(v, xt[I0 + il], accc[ln]);
            accr[il] = std::fma(v, xt[J0 + ln], accr[il]);
          }
        slot[w] += nsteps * 64u;
        std::vector<double> &rv = (flags & DT_SWAP) ? yc : yr, &cv = (flags & DT_SWAP) ? yr : yc;
        std::vector<int> &rt = (flags & DT_SWAP) ? touched_c : touched_r, &ct = (flags & DT_SWAP) ? touched_r : touched_c;
        if (rt[I0 / 64]++ || ct[J0 / 64]++) ok = false;
        for (uint32_t ln = 0; ln < 64; ln++) { rv[I0 + ln] += accr[ln]; cv[J0 + ln] += accc[ln]; }
        pos[w]++;
      }
  }
  for (int w = 0; w < dt.nw; w++) if (pos[w] != dt.wave_task[w + 1] || epoch[w] + dt.tail_bar[w] != (uint32_t)dt.n_phases) ok = false;
  for (int i = 0; i < k; i++) xt[i] = yr[i] + yc[i];
  return ok;
}

// ------------------------------------------------ block factor (device refactor)

static void build_block_factor(Analysis &an) {
  BlockFactor &bf = an.bf;
  bf = BlockFactor();
  const int N = an.N, nch = (int)an.chunk_start.size() - 1;
  std::vector<int> chunk_of(N);
  for (int c = 0; c < nch; c++) for (int j = an.chunk_start[c]; j < an.chunk_start[c + 1]; j++) chunk_of[j] = c;
  auto cw = [&](int c) { return an.chunk_start[c + 1] - an.chunk_start[c]; };
  // blocks of every chunk column, sorted by row chunk (diagonal block first)
  std::vector<std::vector<std::pair<int, uint32_t>>> colblk(nch);   // (I, block id)
  std::vector<std::vector<std::pair<int, uint32_t>>> rowlist(nch);  // per row chunk: (K, block id) with K < I
  std::vector<int> mark(nch, -1);
  // dense tail: its chunk columns (16 wide each) hold the Schur complement S instead of L - every block (I >= J) of
  // the tail exists, receives the updates of the columns before the tail and is neither factorised nor a source
  const int ts = an.dt.k ? an.dt.s : N;
  const int ct0 = an.dt.k ? chunk_of[ts] : nch;            // first tail chunk
  for (int J = 0; J < nch; J++) {
    std::vector<int> rows{J};
    mark[J] = J;
    if (J >= ct0) { for (int I = J + 1; I < nch; I++) rows.push_back(I); }
    else
    for (int col = an.chunk_start[J]; col < an.chunk_start[J + 1]; col++)
      for (int p = an.Lp[col]; p < an.Lp[col + 1]; p++) {
        int I = chunk_of[an.Li[p]];
        if (mark[I] != J) { mark[I] = J; rows.push_back(I); }
      }
    std::sort(rows.begin(), rows.end());
    for (int I : rows) {
      uint32_t id = (uint32_t)bf.n_blocks();
      uint32_t h = (uint32_t)cw(I), w = (uint32_t)cw(J);
      bf.blk.insert(bf.blk.end(), {bf.storage, (uint32_t)an.chunk_start[I], (uint32_t)an.chunk_start[J], (h << 8) | w});
      bf.storage += h * w;
      colblk[J].push_back({I, id});
      if (I != J) rowlist[I].push_back({J, id});
    }
  }
  auto find_blk = [&](int I, int J) -> uint32_t {
    const auto &v = colblk[J];
    auto it = std::lower_bound(v.begin(), v.end(), std::make_pair(I, 0u));
    return it->second;
  };
  auto pos_of = [&](int r, int c) -> uint32_t {      // permuted (row >= col) -> storage position
    int I = chunk_of[r], J = chunk_of[c];
    uint32_t id = find_blk(I, J);
    uint32_t off = bf.blk[4 * id], h = bf.blk[4 * id + 3] >> 8;
    return off + (uint32_t)(c - an.chunk_start[J]) * h + (uint32_t)(r - an.chunk_start[I]);
  };
  // levels over chunk columns (= forward levels of the chunks)
  int nlev = 0;
  for (int c = 0; c < nch; c++) nlev = std::max(nlev, an.chunk_lev[c] + 1);
  bf.n_levels = nlev;
  std::vector<std::vector<int>> cols_of_level(nlev);
  for (int c = 0; c < nch; c++) cols_of_level[an.chunk_lev[c]].push_back(c);
  // Update tasks.  A target block (I,J) gets two of them: its rank-1 sources (one-column
  // chunks) are applied EAGERLY, at the first level after the last such source is final
  // (many targets at once: wide, well balanced), its general sources at level(J).
  struct UT { uint32_t id, tb, tm, te; };
  std::vector<std::vector<UT>> ut_of_level(nlev);
  auto width_of = [&](uint32_t bid) { return bf.blk[4 * bid + 3] & 255u; };
  auto colchunk_level = [&](uint32_t bid) {            // level of the column chunk of a block
    int c0 = (int)bf.blk[4 * bid + 2];
    return an.chunk_lev[chunk_of[c0]];
  };
  for (int L = 0; L < nlev; L++)
    for (int J : cols_of_level[L]) {
      const auto &rj = rowlist[J];
      for (const auto &[I, id] : colblk[J]) {
        if (J >= ct0) continue;          // the tail tiles get their updates in tail_kernel (matrix cores), not through update tasks
        std::vector<std::pair<uint32_t, uint32_t>> tl;
        if (I == J) {
          for (const auto &[K, bid] : rj) if (K < ct0) tl.push_back({bid, bid});
        } else {
          const auto &ri = rowlist[I];
          size_t a = 0, b = 0;
          while (a < ri.size() && b < rj.size()) {
            if (ri[a].first < rj[b].first) a++;
            else if (ri[a].first > rj[b].first) b++;
            else { if (ri[a].first < ct0) tl.push_back({ri[a].second, rj[b].second}); a++; b++; }
          }
        }
        std::stable_partition(tl.begin(), tl.end(), [&](const std::pair<uint32_t, uint32_t> &t) { return width_of(t.first) == 1; });
        size_t n1 = 0;
        int lev1 = 0;
        while (n1 < tl.size() && width_of(tl[n1].first) == 1) { lev1 = std::max(lev1, colchunk_level(tl[n1].first) + 1); n1++; }
        if (n1 && (lev1 >= L || n1 == tl.size()) ) {
          // no earlier level available (or nothing else to do): ONE task, so that never two waves touch a block
          uint32_t tb = (uint32_t)(bf.tri.size() / 2);
          for (size_t q = 0; q < tl.size(); q++) bf.tri.insert(bf.tri.end(), {tl[q].first, tl[q].second});
          uint32_t te = (uint32_t)(bf.tri.size() / 2);
          ut_of_level[std::min(lev1, L)].push_back({id, tb, tb + (uint32_t)n1, te});
          continue;
        }
        if (n1) {
          uint32_t tb = (uint32_t)(bf.tri.size() / 2);
          for (size_t q = 0; q < n1; q++) bf.tri.insert(bf.tri.end(), {tl[q].first, tl[q].second});
          uint32_t te = (uint32_t)(bf.tri.size() / 2);
          ut_of_level[lev1].push_back({id, tb, te, te});
        }
        if (n1 < tl.size()) {
          uint32_t tb = (uint32_t)(bf.tri.size() / 2);
          for (size_t q = n1; q < tl.size(); q++) bf.tri.insert(bf.tri.end(), {tl[q].first, tl[q].second});
          uint32_t te = (uint32_t)(bf.tri.size() / 2);
          ut_of_level[L].push_back({id, tb, tb, te});
        }
      }
    }
  for (int L = 0; L < nlev; L++) {
    uint32_t u0 = (uint32_t)(bf.utask.size() / 4), d0 = (uint32_t)bf.dtask.size(), t0 = (uint32_t)(bf.ttask.size() / 2);
    auto &uts = ut_of_level[L];
    std::stable_sort(uts.begin(), uts.end(), [](const UT &a, const UT &b) { return a.te - a.tb > b.te - b.tb; });
    {
      // small = at most 16 triples and a target of at most 8 columns (the accumulators of the four-tasks-per-wave form);
      // the others first, both classes by size
      auto small = [&](const UT &u) { return u.te - u.tb <= 16u && (bf.blk[4 * u.id + 3] & 255u) <= 8u; };
      std::stable_partition(uts.begin(), uts.end(), [&](const UT &u) { return !small(u); });
      uint32_t nbig = 0;
      while (nbig < uts.size() && !small(uts[nbig])) nbig++;
      bf.ubig.push_back(u0 + nbig);
    }
    for (const UT &u : uts) bf.utask.insert(bf.utask.end(), {u.id, u.tb, u.tm, u.te});
    if (getenv("MI_OSQP_DEBUG_ORDER") && !uts.empty()) {
      // critical path of the U step with round-robin assignment to 16 waves (rank-1 batches of 8, general triples)
      std::vector<double> wv(16, 0.0);
      size_t r1 = 0, gen = 0;
      for (size_t k = 0; k < uts.size(); k++) { wv[k % 16] += (uts[k].tm - uts[k].tb + 7) / 8 * 1.0 + (uts[k].te - uts[k].tm) * 1.5; r1 += uts[k].tm - uts[k].tb; gen += uts[k].te - uts[k].tm; }
      fprintf(stderr, "[mi_osqp] factor level %d: %zu update tasks, rank-1 triples %zu, general %zu, biggest task %u, modelled path %.0f units (ideal %.0f)\n",
              L, uts.size(), r1, gen, uts[0].te - uts[0].tb, *std::max_element(wv.begin(), wv.end()), (r1 / 8.0 + gen * 1.5) / 16);
    }
    // diagonal blocks and triangular solves: the ones of more than 4 columns first, then the narrow ones (four per wave)
    for (int pass = 0; pass < 2; pass++) {
      for (int J : cols_of_level[L]) {
        if (J >= ct0 || (cw(J) > 4) != (pass == 0)) continue;
        for (const auto &[I, id] : colblk[J]) {
          if (I == J) bf.dtask.push_back(id);
          else bf.ttask.insert(bf.ttask.end(), {id, colblk[J][0].second});
        }
      }
      if (pass == 0) { bf.ubig.push_back((uint32_t)bf.dtask.size()); bf.ubig.push_back((uint32_t)(bf.ttask.size() / 2)); }
    }
    bf.lvl.insert(bf.lvl.end(), {u0, (uint32_t)(bf.utask.size() / 4), d0, (uint32_t)bf.dtask.size(), t0, (uint32_t)(bf.ttask.size() / 2)});
  }
  if (getenv("MI_OSQP_DEBUG_ORDER")) {
    size_t cnt[5] = {0, 0, 0, 0, 0}, small = 0;
    for (size_t t = 0; t < bf.utask.size() / 4; t++) {
      const uint32_t w = bf.blk[4 * bf.utask[4 * t] + 3] & 255u, sz = bf.utask[4 * t + 3] - bf.utask[4 * t + 1];
      cnt[w <= 1 ? 0 : w <= 2 ? 1 : w <= 4 ? 2 : w <= 8 ? 3 : 4]++;
      small += sz <= 16;
    }
    fprintf(stderr, "[mi_osqp] update tasks by target width: 1: %zu, 2: %zu, 3-4: %zu, 5-8: %zu, 9-16: %zu; %zu of %zu with <= 16 triples\n", cnt[0], cnt[1], cnt[2], cnt[3], cnt[4], small, bf.utask.size() / 4);
  }
  bf.utask4.resize(bf.utask.size());
  for (size_t t = 0; t < bf.utask.size() / 4; t++) {
    const uint32_t id = bf.utask[4 * t], tb = bf.utask[4 * t + 1], tm = bf.utask[4 * t + 2], te = bf.utask[4 * t + 3];
    const uint32_t hw = bf.blk[4 * id + 3];
    if (tm - tb >= (1u << 22)) bf.overflow = true;                // (4 M one-column sources of one block: analyze() refuses the pattern)
    bf.utask4[4 * t] = bf.blk[4 * id]; bf.utask4[4 * t + 1] = tb; bf.utask4[4 * t + 2] = ((hw >> 8) << 27) | ((hw & 255u) << 22) | ((tm - tb) & 0x3FFFFFu); bf.utask4[4 * t + 3] = te - tm;
  }
  bf.dtask4.resize(4 * bf.dtask.size());
  for (size_t t = 0; t < bf.dtask.size(); t++) for (int k = 0; k < 4; k++) bf.dtask4[4 * t + k] = bf.blk[4 * bf.dtask[t] + k];
  bf.ttask4.resize(2 * bf.ttask.size());
  for (size_t t = 0; t < bf.ttask.size() / 2; t++) {
    const uint32_t id = bf.ttask[2 * t], dg = bf.ttask[2 * t + 1];
    bf.ttask4[4 * t] = bf.blk[4 * id]; bf.ttask4[4 * t + 1] = bf.blk[4 * id + 2]; bf.ttask4[4 * t + 2] = bf.blk[4 * id + 3]; bf.ttask4[4 * t + 3] = bf.blk[4 * dg];
  }
  bf.tri4.resize(2 * bf.tri.size());
  for (size_t q = 0; q < bf.tri.size() / 2; q++) {
    const uint32_t ia = bf.tri[2 * q], ib = bf.tri[2 * q + 1];
    const uint32_t ahw = bf.blk[4 * ia + 3], bh = bf.blk[4 * ib + 3] >> 8;
    bf.tri4[4 * q] = bf.blk[4 * ia]; bf.tri4[4 * q + 1] = bf.blk[4 * ib]; bf.tri4[4 * q + 2] = bf.blk[4 * ia + 2];
    bf.tri4[4 * q + 3] = ((ahw >> 8) << 16) | ((ahw & 255u) << 8) | bh;
  }
  // assembly map: natural KKT entry -> (storage position, value source)
  const int n = an.n, m = an.m, nnzK = an.nnzK();
  bf.asm_dst.assign(nnzK, 0); bf.asm_src.assign(nnzK, 0);
  std::vector<int> colOfK(nnzK);
  for (int j = 0; j < N; j++) for (int k = an.Kp[j]; k < an.Kp[j + 1]; k++) colOfK[k] = j;
  for (int e = 0; e < nnzK; e++) {
    int a = an.pinv[an.Ki[e]], b = an.pinv[colOfK[e]];
    bf.asm_dst[e] = pos_of(std::max(a, b), std::min(a, b));
  }
  for (int k = 0; k < an.Pp[n]; k++) bf.asm_src[an.PtoK[k]] = ((uint32_t)(an.PisDiag[k] ? ASM_P_SIGMA : ASM_P) << 29) | (uint32_t)k;
  for (int pos : an.sigmaOnlyK) bf.asm_src[pos] = (uint32_t)ASM_SIGMA << 29;
  for (int k = 0; k < an.Ap[n]; k++) bf.asm_src[an.AtoK[k]] = ((uint32_t)ASM_A << 29) | (uint32_t)k;
  for (int r = 0; r < m; r++) bf.asm_src[an.rhotoK[r]] = ((uint32_t)ASM_NEG_RHOINV << 29) | (uint32_t)r;
  // canonical L entry -> storage position, and the schedule maps composed with it
  bf.lpos.assign(an.nnzLx(), 0);
  for (int j = 0; j < N; j++) for (int p = an.Lp[j]; p < an.Lp[j + 1]; p++) bf.lpos[p] = (int32_t)pos_of(an.Li[p], j);
  for (int c = 0; c < nch; c++) {
    const int c0 = an.chunk_start[c], r = cw(c);
    if (an.inv_off[c] < 0) continue;
    const uint32_t id = find_blk(c, c), off = bf.blk[4 * id];
    for (int k = 0; k < r; k++) for (int i = k + 1; i < r; i++) bf.lpos[an.inv_index(c, i, k)] = (int32_t)(off + (uint32_t)i * r + (uint32_t)k);
    (void)c0;
  }
  auto compose = [&](const Schedule &s, std::vector<int32_t> &out) {
    out.resize(s.src.size());
    for (size_t k = 0; k < s.src.size(); k++) out[k] = s.src[k] >= 0 ? bf.lpos[s.src[k]] : s.src[k];   // MI_SRC_ZERO / MI_SRC_ONE pass through
  };
  compose(an.fwd, an.fwd_srcblk);
  compose(an.bwd, an.bwd_srcblk);
  if (an.dt.k) {
    const int k = an.dt.k;
    an.dt.sblk.assign((size_t)k * k, 0u);
    for (int j = 0; j < k; j++) for (int i = j; i < k; i++) an.dt.sblk[(size_t)j * k + i] = pos_of(ts + i, ts + j);
    // ---- tail_kernel tables (host_core.hpp DenseTail)
    DenseTail &dt = an.dt;
    const int nt = k / 16;
    // compact entries: L[i, c], i >= ts, c < ts; fragments (tile row, column) -> 16 compact indices
    dt.lt_pos.clear(); dt.ltcol_col.clear();
    std::vector<std::vector<std::pair<int, std::vector<uint16_t>>>> frag(nt);      // per tile row: (compact column, 16 indices), ascending column
    for (int c = 0; c < ts; c++) {
      int ci = -1;
      for (int p = an.Lp[c]; p < an.Lp[c + 1]; p++) {
        const int i = an.Li[p];
        if (i < ts) continue;
        if (ci < 0) { ci = (int)dt.ltcol_col.size(); dt.ltcol_col.push_back((uint32_t)c); }
        const int I = (i - ts) / 16;
        if (frag[I].empty() || frag[I].back().first != ci) frag[I].push_back({ci, std::vector<uint16_t>(16, 0xFFFFu)});
        frag[I].back().second[(i - ts) % 16] = (uint16_t)dt.lt_pos.size();
        dt.lt_pos.push_back((uint32_t)bf.lpos[p]);
      }
    }
    dt.n_lt = (int)dt.lt_pos.size(); dt.n_ltcol = (int)dt.ltcol_col.size();
    const uint16_t zero_slot = (uint16_t)dt.n_lt;          // (analyze() keeps n_lt below 65 535 when it picks a dense tail)
    for (auto &fr : frag) for (auto &f : fr) for (uint16_t &v : f.second) if (v == 0xFFFFu) v = zero_slot;
    struct TileRec { int I, J; std::vector<std::pair<int, int>> src; };           // src: (fragment of I, fragment of J) with a common column
    std::vector<TileRec> tiles;
    for (int I = 0; I < nt; I++)
      for (int J = 0; J <= I; J++) {
        TileRec t{I, J, {}};
        size_t a = 0, b = 0;
        while (a < frag[I].size() && b < frag[J].size()) {
          if (frag[I][a].first < frag[J][b].first) a++;
          else if (frag[I][a].first > frag[J][b].first) b++;
          else { t.src.push_back({(int)a, (int)b}); a++; b++; }
        }
        tiles.push_back(std::move(t));
      }
    // tiles to waves: longest first, dealt in snake order (balanced without a queue on the device)
    const int nw = 16;                    // waves of tail_assemble_kernel (one 1024-thread workgroup per QP)
    std::vector<int> order(tiles.size());
    std::iota(order.begin(), order.end(), 0);
    std::stable_sort(order.begin(), order.end(), [&](int x, int y) { return tiles[x].src.size() > tiles[y].src.size(); });
    std::vector<std::vector<int>> mine(nw);
    for (size_t d = 0; d < order.size(); d++) { const size_t round = d / nw, pos = d % nw; mine[(round & 1) ? nw - 1 - pos : pos].push_back(order[d]); }
    dt.tile_tab.clear(); dt.wave_tiles.assign(nw + 1, 0u); dt.asm_q.clear(); dt.asm_qcol.clear();
    for (int w = 0; w < nw; w++) {
      dt.wave_tiles[w] = (uint32_t)(dt.tile_tab.size() / 4);
      for (int ti : mine[w]) {
        const TileRec &t = tiles[ti];
        const uint32_t q0 = (uint32_t)(dt.asm_q.size() / 64);
        // (padded to whole blocks of 8 quads = 32 sources: tail_kernel walks the tables in straight-line blocks of 8)
        const size_t padded = (t.src.size() + 31) / 32 * 32;
        for (size_t s0 = 0; s0 < padded; s0 += 4) {
          for (int g = 0; g < 4; g++) {
            const bool real = s0 + g < t.src.size();
            const auto *fa = real ? &frag[t.I][t.src[s0 + g].first] : nullptr;
            const auto *fb = real ? &frag[t.J][t.src[s0 + g].second] : nullptr;
            dt.asm_qcol.push_back(real ? (uint16_t)fa->first : (uint16_t)0);
            for (int e = 0; e < 16; e++) dt.asm_q.push_back(real ? ((uint32_t)fa->second[e] | ((uint32_t)fb->second[e] << 16)) : ((uint32_t)zero_slot | ((uint32_t)zero_slot << 16)));
          }
        }
        const uint32_t blk_id = find_blk(ct0 + t.I, ct0 + t.J);
        dt.tile_tab.insert(dt.tile_tab.end(), {((uint32_t)t.I << 16) | (uint32_t)t.J, bf.blk[4 * blk_id], q0, (uint32_t)(dt.asm_q.size() / 64)});
      }
    }
    dt.wave_tiles[nw] = (uint32_t)(dt.tile_tab.size() / 4);
    dt.asm_q64.resize(dt.asm_q.size());
    for (size_t e = 0; e < dt.asm_q.size(); e++) {
      const uint64_t w = dt.asm_q[e], ci = dt.asm_qcol[e / 16];               // (lane / 16 = source of the quad)
      dt.asm_q64[e] = ((w & 0xFFFFu) * 8u) | ((((w >> 16) * 8u) | ((ci * 8u) << 17)) << 32);
    }
    // asm_q is [quad][lane]: lane = source * 16 + tile row -- the loops above emitted exactly that order
    dt.src_tile.resize(dt.src.size());
    for (size_t e = 0; e < dt.src.size(); e++) {
      const int32_t sc = dt.src[e];
      dt.src_tile[e] = sc < 0 ? sc : (int32_t)dt_tile_offset(sc % k, sc / k);
    }
    dt.diag_tile.resize(k);
    for (int i = 0; i < k; i++) dt.diag_tile[i] = dt_tile_offset(i, i);
    dt.task_step.clear();
    { uint32_t pos = 0; for (size_t t = 0; t < dt.task.size() / 4; t++) { dt.task_step.push_back(pos); pos += dt.task[4 * t + 3]; } }
  }
}

// --------------------------------------------------------------------- analyze

int analyze(int64_t n64, int64_t m64, const int64_t *Pp, const int64_t *Pi, const int64_t *Ap,
            const int64_t *Ai, Analysis &an, int nwaves, int bt, int max_extra_rows, int dense_tail_max, int tri_waves, int n_tiles) {
  if (nwaves < 1 || nwaves > 16 || (bt != 1 && bt != 2 && bt != 4) || tri_waves < 0 || tri_waves > 2048) return MI_OSQP_ERR_INVALID_SETTINGS;
  if (n64 <= 0 || m64 < 0 || !Pp || !Ap || n64 + m64 > (int64_t)1 << 30) return MI_OSQP_ERR_INVALID_DATA;
  int n = (int)n64, m = (int)m64, N = n + m;
  an = Analysis();
  an.n = n; an.m = m; an.N = N;
  if (Pp[0] != 0 || Ap[0] != 0) return MI_OSQP_ERR_INVALID_DATA;
  for (int j = 0; j < n; j++) {
    if (Pp[j + 1] < Pp[j] || Ap[j + 1] < Ap[j]) return MI_OSQP_ERR_INVALID_DATA;
    for (int64_t k = Pp[j]; k < Pp[j + 1]; k++) if (Pi[k] < 0 || Pi[k] >= n) return MI_OSQP_ERR_INVALID_DATA;
    for (int64_t k = Ap[j]; k < Ap[j + 1]; k++) if (Ai[k] < 0 || Ai[k] >= m) return MI_OSQP_ERR_INVALID_DATA;
  }
  if (Pp[n] > (int64_t)1 << 30 || Ap[n] > (int64_t)1 << 30) return MI_OSQP_ERR_INVALID_DATA;
  // triu(P)
  an.Pp.assign(n + 1, 0);
  for (int j = 0; j < n; j++) {
    an.Pp[j] = (int)an.Pi.size();
    for (int64_t k = Pp[j]; k < Pp[j + 1]; k++)
      if (Pi[k] <= j) { an.Pi.push_back((int)Pi[k]); an.Psrc.push_back((int)k); }
  }
  an.Pp[n] = (int)an.Pi.size();
  an.Ap.assign(Ap, Ap + n + 1);
  an.Ai.assign(Ai, Ai + Ap[n]);
  // ---- natural upper KKT (row E4)
  int nnzP = an.Pp[n], nnzA = an.Ap[n];
  std::vector<int> cnt(N, 0);
  an.PisDiag.assign(nnzP, 0);
  std::vector<char> hasDiag(n, 0);
  for (int j = 0; j < n; j++)
    for (int k = an.Pp[j]; k < an.Pp[j + 1]; k++) { cnt[j]++; if (an.Pi[k] == j) { hasDiag[j] = 1; an.PisDiag[k] = 1; } }
  for (int j = 0; j < n; j++) if (!hasDiag[j]) cnt[j]++;
  for (int k = 0; k < nnzA; k++) cnt[n + an.Ai[k]]++;
  for (int r = 0; r < m; r++) cnt[n + r]++;
  an.Kp.assign(N + 1, 0);
  for (int j = 0; j < N; j++) an.Kp[j + 1] = an.Kp[j] + cnt[j];
  an.Ki.assign(an.Kp[N], 0);
  an.PtoK.assign(nnzP, 0); an.AtoK.assign(nnzA, 0); an.rhotoK.assign(m, 0);
  std::vector<int> nxt(an.Kp.begin(), an.Kp.end() - 1);
  for (int j = 0; j < n; j++) {
    for (int k = an.Pp[j]; k < an.Pp[j + 1]; k++) { int pos = nxt[j]++; an.Ki[pos] = an.Pi[k]; an.PtoK[k] = pos; }
    if (!hasDiag[j]) { int pos = nxt[j]++; an.Ki[pos] = j; an.sigmaOnlyK.push_back(pos); }
  }
  for (int j = 0; j < n; j++)
    for (int k = an.Ap[j]; k < an.Ap[j + 1]; k++) { int pos = nxt[n + an.Ai[k]]++; an.Ki[pos] = j; an.AtoK[k] = pos; }
  for (int r = 0; r < m; r++) { int pos = nxt[n + r]++; an.Ki[pos] = n + r; an.rhotoK[r] = pos; }
  // ---- ordering: two candidates, chosen by the modelled time of one KKT solve on the device
  // (latency-bound regime of one tile, measured with scripts/trace_phases.py: ~0.4 us per phase, ~40 GB/s of factor stream)
  // 16-bit index words while every vector the streams address stays below 65 535 entries (0xFFFF = "no row")
  an.tri_waves = tri_waves;
  an.wide = N >= 65535 || 2 * n + m >= 65535 || tri_waves > 0;        // (the dataflow form doubles the index range)
  if (max_extra_rows < 0) max_extra_rows = 1 << 30;       // (also the budget of the dense tail's two accumulation vectors)
  if (!an.wide && max_extra_rows > 65534 - N) max_extra_rows = 65534 - N;
  // (relax_zeros = explicit zeros a relaxed supernode may hold; 0 = fundamental supernodes only)
  auto finalize = [&](Analysis &an, const std::vector<int> &perm0, double &cost, int relax_zeros = 0, int *phases_out = nullptr, double *stream_out = nullptr) {
    an.perm = perm0;
    an.pinv.assign(N, 0);
    for (int k = 0; k < N; k++) an.pinv[an.perm[k]] = k;
    build_permuted_lower(an);
    std::vector<std::vector<int>> cols;
    std::vector<int> parent, post;
    symbolic(an, cols, parent);
    postorder(parent, post);       // etree postorder so that supernodes are contiguous
    {
      std::vector<int> p2(N);
      for (int k = 0; k < N; k++) p2[k] = an.perm[post[k]];
      an.perm.swap(p2);
      for (int k = 0; k < N; k++) an.pinv[an.perm[k]] = k;
    }
    build_permuted_lower(an);
    symbolic(an, cols, parent);
    an.etree = parent;
    // Relaxed supernodes: a chain j -> j+1 -> ... of the elimination tree whose columns have ALMOST the same structure becomes
    // one supernode by storing explicit zeros (column c takes the structure {c+1 .. e} u struct(e) of the chain's last column;
    // still a closed symbolic factor).  The factors of the trajectory QPs are made of hundreds of one- and two-column
    // supernodes - every one a level of the sweeps and a handful of 2-entry block tasks of the refactorisation.
    if (relax_zeros > 0) {
      auto close = [&](int s0, int e) {
        for (int c = s0; c < e; c++) {
          std::vector<int> ns;
          ns.reserve((size_t)(e - c) + cols[e].size());
          for (int r = c + 1; r <= e; r++) ns.push_back(r);
          ns.insert(ns.end(), cols[e].begin(), cols[e].end());
          cols[c].swap(ns);
        }
      };
      int s0 = 0;
      for (int j = 0; j < N; j++) {
        bool extend = false;
        if (j + 1 < N && parent[j] == j + 1 && j + 2 - s0 <= kChunk) {
          long z = 0;
          for (int c = s0; c <= j; c++) z += (long)(j + 1 - c) + (long)cols[j + 1].size() - (long)cols[c].size();
          extend = z <= relax_zeros;
        }
        if (!extend) { if (j > s0) close(s0, j); s0 = j + 1; }
      }
    }
    an.Lp.assign(N + 1, 0);
    for (int j = 0; j < N; j++) an.Lp[j + 1] = an.Lp[j] + (int)cols[j].size();
    an.Li.resize(an.Lp[N]);
    for (int j = 0; j < N; j++) std::copy(cols[j].begin(), cols[j].end(), an.Li.begin() + an.Lp[j]);
    // row view of L
    an.Rp.assign(N + 1, 0);
    for (int p = 0; p < an.Lp[N]; p++) an.Rp[an.Li[p] + 1]++;
    for (int i = 0; i < N; i++) an.Rp[i + 1] += an.Rp[i];
    an.Rj.resize(an.Lp[N]); an.Rpos.resize(an.Lp[N]);
    {
      std::vector<int> fill(an.Rp.begin(), an.Rp.end() - 1);
      for (int j = 0; j < N; j++)
        for (int p = an.Lp[j]; p < an.Lp[j + 1]; p++) { int q = fill[an.Li[p]]++; an.Rj[q] = j; an.Rpos[q] = p; }
    }
    // ---- dense tail (host_core.hpp DenseTail): the last k rows (k a multiple of 64) whose triangle of L would cost
    // more bytes per solve (read twice) than the k^2/2 values of the inverted Schur complement (read once); taken
    // when it saves at least a tenth of the factor stream.  MI_OSQP_DENSE_TAIL = 0: never, = k: exactly k rows.
    an.dt = DenseTail();
    int64_t tail_nnz_used = 0;
    {
      const char *e = getenv("MI_OSQP_DENSE_TAIL");
      const int forced = e ? atoi(e) : -1;
      const int kmax = std::min({forced == 0 ? 0 : dense_tail_max, N - 1, max_extra_rows / 2});
      int64_t tail_nnz = 0, best_gain = 0;
      int best_k = 0;
      for (int k = 1; k <= kmax; k++) {
        tail_nnz += (int64_t)cols[N - k].size();
        if (k % 64) continue;
        const int64_t gain = 2 * tail_nnz - (int64_t)k * k / 2 - k;
        if (forced > 0 ? k == forced / 64 * 64 : gain > best_gain) { best_gain = gain; best_k = k; tail_nnz_used = tail_nnz; }
      }
      if (best_k && (forced > 0 || best_gain * 10 >= 2 * (int64_t)an.Lp[N])) {
        // tail_kernel stages the entries of L in the tail rows of the columns before the tail (and the D of those
        // columns) in LDS and addresses them with 16-bit indices: a pattern that exceeds either keeps its triangle
        int64_t n_lt = 0, n_ltcol = 0;
        for (int c = 0; c < N - best_k; c++) {
          const auto lo = std::lower_bound(cols[c].begin(), cols[c].end(), N - best_k);
          n_lt += cols[c].end() - lo; n_ltcol += lo != cols[c].end();
        }
        if (n_lt <= 16382 && n_ltcol <= 4095 && 8 * (n_lt + 1 + n_ltcol) <= 144 * 1024) { an.dt.k = best_k; an.dt.s = N - best_k; }     // (17- / 15-bit byte offsets in asm_q64)
        else tail_nnz_used = 0;
      } else tail_nnz_used = 0;
    }
    const int ts = an.dt.k ? an.dt.s : N;
    // fundamental supernodes: j+1 joins j when parent(j)=j+1 and |col j| = |col j+1| + 1
    an.sn_start.clear(); an.sn_start.push_back(0);
    for (int j = 0; j + 1 < N; j++) {
      bool join = parent[j] == j + 1 && cols[j].size() == cols[j + 1].size() + 1;
      if (!join) an.sn_start.push_back(j + 1);
    }
    an.sn_start.push_back(N);
    // <=16-column chunks of supernodes.  Every row of a multi-row chunk needs a second position in the solve
    // vector (phase B reads t and writes x); when the budget of extra positions (LDS capacity / 16-bit
    // indices) is used up, the remaining supernodes are cut into one-row chunks (more levels, no phase B).
    an.chunk_start.clear();
    int extra_left = max_extra_rows - 2 * an.dt.k;           // the dense tail keeps two accumulation vectors next to the solve vector
    for (size_t s = 0; s + 1 < an.sn_start.size() && an.sn_start[s] < ts; s++) {
      const int send = std::min(an.sn_start[s + 1], ts);     // chunks never straddle the start of the dense tail
      for (int c = an.sn_start[s]; c < send; c += kChunk) {
        const int r = std::min(kChunk, send - c);
        if (r >= 2 && r > extra_left) { for (int j = c; j < c + r; j++) an.chunk_start.push_back(j); continue; }
        if (r >= 2) extra_left -= r;
        an.chunk_start.push_back(c);
      }
    }
    for (int c = ts; c < N; c += kChunk) an.chunk_start.push_back(c);     // the tail: 16 x 16 tiles of the Schur complement
    an.chunk_start.push_back(N);
    // forward chunk levels (the backward sweep has the same depth)
    int nch = (int)an.chunk_start.size() - 1, depth = 0;
    std::vector<int> chunk_of(N), lev(nch, 0);
    for (int c = 0; c < nch; c++) for (int j = an.chunk_start[c]; j < an.chunk_start[c + 1]; j++) chunk_of[j] = c;
    for (int c = 0; c < nch; c++) {
      int c0 = an.chunk_start[c], c1 = an.chunk_start[c + 1], L = 0;
      for (int i = c0; i < c1; i++)
        for (int t = an.Rp[i]; t < an.Rp[i + 1]; t++) { int j = an.Rj[t]; if (j < c0 && j < ts) L = std::max(L, lev[chunk_of[j]] + 1); }
      lev[c] = L; depth = std::max(depth, L + 1);
    }
    // phases of one sweep: one A phase per level + one B phase where the level has multi-row chunks
    std::vector<int> blocks_at(depth, 0);
    for (int c = 0; c < nch; c++) if (an.chunk_start[c + 1] - an.chunk_start[c] >= 2 && an.chunk_start[c] < ts) blocks_at[lev[c]]++;
    int phases = depth;
    for (int L = 0; L < depth; L++) phases += blocks_at[L] ? 1 : 0;
    const double stream = 2.0 * (double)(an.Lp[N] - tail_nnz_used) + 0.5 * (double)an.dt.k * an.dt.k;   // values read per solve
    cost = (2.0 * phases + (an.dt.k ? an.dt.k / 128 + 3 : 0)) * 0.4e-6 + 8.0 * bt * 1.15 * stream / 40e9;
    if (phases_out) *phases_out = 2 * phases + (an.dt.k ? an.dt.k / 128 + 3 : 0);
    if (stream_out) *stream_out = 8.0 * bt * 1.15 * stream;
  };
  {
    const bool dbg_t = getenv("MI_OSQP_DEBUG_ORDER") != nullptr;
    auto now_ = [] { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
    double tt = now_();
    const char *force = getenv("MI_OSQP_ORDERING");       // "md" / "nd": experiments
    // Candidates: minimum degree (works on the explicit elimination graph - quadratic on big meshes: 0.9 s at N = 160 k, where
    // nested dissection takes 0.2 s and wins anyway - so beyond 60 k rows only on request) and nested dissections with
    // several leaf sizes (all of them while the pattern is small enough for that to cost milliseconds; MI_OSQP_ND_LEAF
    // forces one size).  The cheapest by the modelled time of one KKT solve wins; ties go to the earlier candidate.
    // Every candidate (ordering + symbolic analysis of its permutation) runs on a host thread of its own, on its own copy of
    // the analysis so far (the 802-waypoint example: 155 -> 60 ms of the 220 ms a setup spends here).
    struct Cand {
      std::vector<int> perm; int leaf = 0;       // leaf 0 = minimum degree
      Analysis an, rel;                         // with fundamental / relaxed supernodes
      double cost = 0.0, cost_rel = 0.0; int phases = 0, phases_rel = 0; double stream = 0.0, stream_rel = 0.0;
      double t_order = 0.0, t_final = 0.0;
    };
    std::vector<Cand> cands;
    const bool try_md = (N <= 60000 && !(force && force[0] == 'n')) || (force && force[0] == 'm');
    if (try_md) cands.emplace_back();
    if (!(force && force[0] == 'm')) {
      std::vector<int> leaves;
      const char *el = getenv("MI_OSQP_ND_LEAF");
      if (el && atoi(el) >= 2) leaves = {atoi(el)};
      else if (N <= 8000) leaves = {48, 24, 12, 8, 4};
      else if (N <= 60000) leaves = {48, 12};
      else leaves = {48};
      for (int lf : leaves) { cands.emplace_back(); cands.back().leaf = lf; }
    }
    // Relaxed supernodes are a second form of every candidate: fewer phases per sweep (and a third fewer block tasks per
    // refactorisation) against a longer factor stream.  MI_OSQP_RELAX = 0: never, = z: always, up to z zeros per supernode;
    // default: decided below.  Barrier form only.
    const char *er = getenv("MI_OSQP_RELAX");
    const int relax_forced = er ? atoi(er) : -1;
    const int relax_z = relax_forced > 0 ? relax_forced : ((relax_forced < 0 && tri_waves == 0 && N <= 60000) ? 16 : 0);      // (forcing works for every form: experiments)
    auto run_cand = [&](Cand &c) {
      double t0 = now_();
      c.an = an;
      if (c.leaf) nested_dissection(N, an.Kp, an.Ki, c.perm, c.leaf); else min_degree(N, an.Kp, an.Ki, c.perm);
      c.t_order = now_() - t0; t0 = now_();
      if (relax_forced <= 0) finalize(c.an, c.perm, c.cost, 0, &c.phases, &c.stream);
      if (relax_z) { c.rel = an; finalize(c.rel, c.perm, c.cost_rel, relax_z, &c.phases_rel, &c.stream_rel); }
      c.t_final = now_() - t0;
    };
    if (cands.size() > 1 && N >= 1000 && !getenv("MI_OSQP_SERIAL_ANALYSIS")) {
      std::vector<std::thread> th;
      for (size_t c = 1; c < cands.size(); c++) th.emplace_back([&, c] { run_cand(cands[c]); });
      run_cand(cands[0]);
      for (auto &t : th) t.join();
    } else for (Cand &c : cands) run_cand(c);
    // 1. the cheapest candidate with fundamental supernodes (modelled time of one KKT solve of a lone tile);
    // 2. against it the cheapest relaxed candidate, both priced with the stream rate a tile of THIS batch gets: the tiles
    //    share the memory system - a lone tile streams ~40 GB/s, 256 of them ~16 GB/s each (measured per iteration with
    //    MI_OSQP_RELAX forced, scripts/tail_latency_probe.py: 256 x 3-DOF x 60 waypoints 23.8 -> 20.9 us relaxed, 256 x 7-DOF
    //    x 100 waypoints 39.7 -> 47.8 us).
    const double rate = std::min(40e9, 4e12 / std::max(1, n_tiles));
    auto priced = [&](int phases, double stream) { return phases * 0.4e-6 + stream / rate; };
    size_t best = 0, best_rel = 0;
    for (size_t c = 0; c < cands.size(); c++) {
      if (dbg_t) fprintf(stderr, "[mi_osqp] ordering candidate %s leaf %d: modelled solve %.3e s, nnz(L) %d, %d phases; relaxed (%d zeros): %d phases, stream x %.2f (ordering %.1f ms, symbolic %.1f ms)\n",
                         cands[c].leaf ? "nd" : "md", cands[c].leaf, cands[c].cost, relax_forced <= 0 ? cands[c].an.Lp[N] : 0, cands[c].phases, relax_z, cands[c].phases_rel,
                         cands[c].stream > 0 ? cands[c].stream_rel / cands[c].stream : 0.0, 1e3 * cands[c].t_order, 1e3 * cands[c].t_final);
      if (cands[c].cost < cands[best].cost) best = c;
      if (relax_z && priced(cands[c].phases_rel, cands[c].stream_rel) < priced(cands[best_rel].phases_rel, cands[best_rel].stream_rel)) best_rel = c;
    }
    bool take_rel = relax_forced > 0;
    if (relax_z && relax_forced < 0) {
      const double t0 = priced(cands[best].phases, cands[best].stream), t1 = priced(cands[best_rel].phases_rel, cands[best_rel].stream_rel);
      take_rel = t1 < 0.97 * t0;
      if (dbg_t) fprintf(stderr, "[mi_osqp] at %.0f GB/s per tile (%d tiles): fundamental %s leaf %d %.2f us, relaxed %s leaf %d %.2f us -> %s\n", rate / 1e9, n_tiles,
                         cands[best].leaf ? "nd" : "md", cands[best].leaf, 1e6 * t0, cands[best_rel].leaf ? "nd" : "md", cands[best_rel].leaf, 1e6 * t1, take_rel ? "relaxed" : "fundamental");
    }
    if (dbg_t) { fprintf(stderr, "[mi_osqp] ordering candidates x %zu %.1f ms\n", cands.size(), 1e3 * (now_() - tt)); tt = now_(); }
    {
      Cand &w = cands[take_rel ? best_rel : best];
      const int leaf = w.leaf;
      Analysis chosen = std::move(take_rel ? w.rel : w.an);
      an = std::move(chosen); an.ordering = leaf ? 1 : 0; an.relaxed_zeros = take_rel ? relax_z : 0;
    }
  }
  if (tri_waves > 0 && (bt != 1 || an.dt.k)) return MI_OSQP_ERR_INVALID_SETTINGS;
  {
    // the check (SpMV) schedule only reads the symbolic analysis and writes tables of its own: a host thread beside the
    // solve schedules + refactorisation plan (which reads them) for patterns where that pays
    const bool dbg_t = getenv("MI_OSQP_DEBUG_ORDER") != nullptr;
    auto now_ = [] { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
    const double t0 = now_();
    double t_chk = 0.0, t_bf = 0.0;
    auto chk_part = [&] { const double t = now_(); build_chk_schedule(an, tri_waves > 0 ? tri_waves : nwaves, bt); t_chk = now_() - t; };      // (dataflow form: check_kernel runs on the whole grid too)
    auto bf_part = [&] { const double t = now_(); build_block_factor(an); t_bf = now_() - t; };
    const bool par = N >= 1000 && !getenv("MI_OSQP_SERIAL_ANALYSIS");
    std::thread th_chk;
    if (par) th_chk = std::thread(chk_part);
    build_tri_schedules(an, tri_waves > 0 ? tri_waves : nwaves, bt, tri_waves > 0);
    build_dense_tail(an, nwaves);
    const double t_tri = now_() - t0;
    bf_part();
    if (par) th_chk.join(); else chk_part();
    if (an.bf.overflow) return MI_OSQP_ERR_INVALID_DATA;
    if (dbg_t) fprintf(stderr, "[mi_osqp] tables: solve schedules + dense tail %.1f ms, check schedule %.1f ms, block factor %.1f ms; wall %.1f ms\n",
                       1e3 * t_tri, 1e3 * t_chk, 1e3 * t_bf, 1e3 * (now_() - t0));
  }
  // A wave that passes fewer barriers than the others would hang its workgroup (and the GPU): re-count.
  for (const Schedule *sc : {&an.fwd, &an.bwd, &an.chk})
    for (int w = 0; w < sc->nw; w++) {
      uint64_t bars = sc->tail_bar[w];
      for (uint32_t q = sc->wave_range[2 * w]; q < sc->wave_range[2 * w + 1]; q++) bars += MI_D_NBAR(sc->step[q]);
      if (bars != (sc->barriers ? (uint64_t)sc->n_phases : 0u)) return MI_OSQP_ERR_INVALID_DATA;
    }
  return MI_OSQP_OK;
}

// --------------------------------------------------------------- per-QP numeric

void load_qp(const Analysis &an, const Settings &st, const double *Pval, const double *q,
             const double *Aval, const double *l, const double *u, QPNumeric &qp) {
  int n = an.n, m = an.m;
  qp.Pv.resize(an.Pp[n]);
  for (int k = 0; k < an.Pp[n]; k++) qp.Pv[k] = Pval[an.Psrc[k]];
  qp.Av.assign(Aval, Aval + an.Ap[n]);
  if (q) qp.q.assign(q, q + n); else qp.q.assign(n, 0.0);
  qp.l.resize(m); qp.u.resize(m);
  for (int i = 0; i < m; i++) { qp.l[i] = std::max(l[i], -kInfty); qp.u[i] = std::min(u[i], kInfty); }
  qp.D.assign(n, 1.0); qp.Dinv.assign(n, 1.0); qp.E.assign(m, 1.0); qp.Einv.assign(m, 1.0);
  qp.c = qp.cinv = 1.0;
  qp.rho = st.rho;
}

static inline double limit_scaling(double v) {
  v = v < kMinScaling ? 1.0 : v;
  return v > kMaxScaling ? kMaxScaling : v;
}

// Ruiz equilibration of [[P, A'],[A, 0]] + cost normalisation (row E2).  The
// arithmetic is the published algorithm; the loops are fused per pattern walk.
void scale_qp(const Analysis &an, const Settings &st, QPNumeric &qp) {
  int n = an.n, m = an.m;
  std::fill(qp.D.begin(), qp.D.end(), 1.0);
  std::fill(qp.E.begin(), qp.E.end(), 1.0);
  qp.c = 1.0;
  std::vector<double> dn(n), en(m);
  for (int64_t it = 0; it < st.scaling; it++) {
    std::fill(dn.begin(), dn.end(), 0.0);
    std::fill(en.begin(), en.end(), 0.0);
    for (int j = 0; j < n; j++)
      for (int k = an.Pp[j]; k < an.Pp[j + 1]; k++) {
        double a = std::fabs(qp.Pv[k]); int i = an.Pi[k];
        if (a > dn[j]) dn[j] = a;
        if (i != j && a > dn[i]) dn[i] = a;
      }
    for (int j = 0; j < n; j++)
      for (int k = an.Ap[j]; k < an.Ap[j + 1]; k++) {
        double a = std::fabs(qp.Av[k]);
        if (a > dn[j]) dn[j] = a;
        if (a > en[an.Ai[k]]) en[an.Ai[k]] = a;
      }
    for (int j = 0; j < n; j++) dn[j] = 1.0 / std::sqrt(limit_scaling(dn[j]));
    for (int i = 0; i < m; i++) en[i] = 1.0 / std::sqrt(limit_scaling(en[i]));
    for (int j = 0; j < n; j++)
      for (int k = an.Pp[j]; k < an.Pp[j + 1]; k++) { qp.Pv[k] *= dn[an.Pi[k]]; qp.Pv[k] *= dn[j]; }
    for (int j = 0; j < n; j++)
      for (int k = an.Ap[j]; k < an.Ap[j + 1]; k++) { qp.Av[k] *= en[an.Ai[k]]; qp.Av[k] *= dn[j]; }
    for (int j = 0; j < n; j++) { qp.q[j] *= dn[j]; qp.D[j] *= dn[j]; }
    for (int i = 0; i < m; i++) qp.E[i] *= en[i];
    // cost normalisation
    std::fill(dn.begin(), dn.end(), 0.0);
    for (int j = 0; j < n; j++)
      for (int k = an.Pp[j]; k < an.Pp[j + 1]; k++) {
        double a = std::fabs(qp.Pv[k]); int i = an.Pi[k];
        if (a > dn[j]) dn[j] = a;
        if (i != j && a > dn[i]) dn[i] = a;
      }
    double mean = 0.0;
    for (int j = 0; j < n; j++) mean += dn[j];
    mean /= (double)n;
    double nq = 0.0;
    for (int j = 0; j < n; j++) nq = std::max(nq, std::fabs(qp.q[j]));
    nq = limit_scaling(nq);
    double ct = limit_scaling(std::max(mean, nq));
    ct = 1.0 / ct;
    for (double &v : qp.Pv) v *= ct;
    for (double &v : qp.q) v *= ct;
    qp.c *= ct;
  }
  qp.cinv = 1.0 / qp.c;
  for (int j = 0; j < n; j++) qp.Dinv[j] = 1.0 / qp.D[j];
  for (int i = 0; i < m; i++) qp.Einv[i] = 1.0 / qp.E[i];
  for (int i = 0; i < m; i++) { qp.l[i] *= qp.E[i]; qp.u[i] *= qp.E[i]; }
}

// qp holds the same unscaled P, A, q as `rep` did before scale_qp(rep): take over rep's equilibration instead
// of recomputing it (GOMP batches: every trajectory shares P and A, only the bounds differ)
void scale_like(const Analysis &an, const QPNumeric &rep, QPNumeric &qp) {
  qp.Pv = rep.Pv; qp.Av = rep.Av; qp.q = rep.q;
  qp.D = rep.D; qp.Dinv = rep.Dinv; qp.E = rep.E; qp.Einv = rep.Einv; qp.c = rep.c; qp.cinv = rep.cinv;
  for (int i = 0; i < an.m; i++) { qp.l[i] *= qp.E[i]; qp.u[i] *= qp.E[i]; }
}

void unscale_qp(const Analysis &an, QPNumeric &qp) {
  int n = an.n, m = an.m;
  for (int j = 0; j < n; j++)
    for (int k = an.Pp[j]; k < an.Pp[j + 1]; k++) { qp.Pv[k] *= qp.cinv; qp.Pv[k] *= qp.Dinv[an.Pi[k]]; qp.Pv[k] *= qp.Dinv[j]; }
  for (int j = 0; j < n; j++) qp.q[j] *= qp.cinv * qp.Dinv[j];
  for (int j = 0; j < n; j++)
    for (int k = an.Ap[j]; k < an.Ap[j + 1]; k++) { qp.Av[k] *= qp.Einv[an.Ai[k]]; qp.Av[k] *= qp.Dinv[j]; }
  for (int i = 0; i < m; i++) { qp.l[i] *= qp.Einv[i]; qp.u[i] *= qp.Einv[i]; }
}

static inline int8_t row_type(double l, double u) {
  if (l < -kInfty * kMinScaling && u > kInfty * kMinScaling) return -1;
  if (u - l < kRhoTol) return 1;
  return 0;
}
static inline double rho_of_type(int8_t t, double rho) {
  return t < 0 ? kRhoMin : (t > 0 ? kRhoEqOverIneq * rho : rho);
}

void set_rho_vec(const Analysis &an, const Settings &, QPNumeric &qp) {
  int m = an.m;
  qp.rho = std::min(std::max(qp.rho, kRhoMin), kRhoMax);
  qp.rho_vec.resize(m); qp.rho_inv.resize(m); qp.ctype.resize(m);
  for (int i = 0; i < m; i++) {
    qp.ctype[i] = row_type(qp.l[i], qp.u[i]);
    qp.rho_vec[i] = rho_of_type(qp.ctype[i], qp.rho);
    qp.rho_inv[i] = 1.0 / qp.rho_vec[i];
  }
}

int refresh_rho_types(const Analysis &an, QPNumeric &qp) {
  int changed = 0;
  for (int i = 0; i < an.m; i++) {
    int8_t t = row_type(qp.l[i], qp.u[i]);
    if (t != qp.ctype[i]) {
      qp.ctype[i] = t; qp.rho_vec[i] = rho_of_type(t, qp.rho); qp.rho_inv[i] = 1.0 / qp.rho_vec[i]; changed = 1;
    }
  }
  return changed;
}

void apply_rho(const Analysis &an, QPNumeric &qp, double rho_new) {
  qp.rho = std::min(std::max(rho_new, kRhoMin), kRhoMax);
  for (int i = 0; i < an.m; i++) {
    if (qp.ctype[i] == 0) { qp.rho_vec[i] = qp.rho; qp.rho_inv[i] = 1.0 / qp.rho; }
    else if (qp.ctype[i] == 1) { qp.rho_vec[i] = kRhoEqOverIneq * qp.rho; qp.rho_inv[i] = 1.0 / qp.rho_vec[i]; }
  }
}

// Left-looking LDL' on the permuted KKT using the precomputed pattern of L
// (sorted columns + row view).  work: N doubles, zero on entry and exit.
// inv(L_cc) of every multi-row chunk (unit lower triangular), appended to the canonical factor array:
// column k of the inverse by forward substitution on e_k
static void invert_diag_blocks(const Analysis &an, double *Lx) {
  const int nch = (int)an.chunk_start.size() - 1;
  double Lc[kChunk][kChunk], V[kChunk];
  for (int c = 0; c < nch; c++) {
    const int c0 = an.chunk_start[c], c1 = an.chunk_start[c + 1], r = c1 - c0;
    if (an.inv_off[c] < 0) continue;
    for (int i = 0; i < r; i++) for (int k = 0; k < r; k++) Lc[i][k] = 0.0;
    for (int j = c0; j < c1; j++)
      for (int p = an.Lp[j]; p < an.Lp[j + 1] && an.Li[p] < c1; p++) Lc[an.Li[p] - c0][j - c0] = Lx[p];
    for (int k = 0; k + 1 < r; k++) {
      V[k] = 1.0;
      for (int i = k + 1; i < r; i++) {
        double v = 0.0;
        for (int p = k; p < i; p++) v = std::fma(-Lc[i][p], V[p], v);
        V[i] = v;
        Lx[an.inv_index(c, i, k)] = v;
      }
    }
  }
}

int factor_qp(const Analysis &an, const Settings &st, QPNumeric &qp, std::vector<double> &w) {
  int n = an.n, m = an.m, N = an.N, nnzP = an.Pp[n], nnzA = an.Ap[n];
  // permuted lower KKT values
  std::vector<double> Kl(an.nnzK());
  for (int k = 0; k < nnzP; k++) Kl[an.KtoKl[an.PtoK[k]]] = qp.Pv[k] + (an.PisDiag[k] ? st.sigma : 0.0);
  for (int pos : an.sigmaOnlyK) Kl[an.KtoKl[pos]] = st.sigma;
  for (int k = 0; k < nnzA; k++) Kl[an.KtoKl[an.AtoK[k]]] = qp.Av[k];
  for (int r = 0; r < m; r++) Kl[an.KtoKl[an.rhotoK[r]]] = -qp.rho_inv[r];
  qp.Lx.assign(an.nnzLx(), 0.0); qp.Dl.assign(N, 0.0); qp.Dlinv.assign(N, 0.0);
  if ((int)w.size() < N) w.assign(N, 0.0);
  int positive = 0;
  for (int j = 0; j < N; j++) {
    for (int p = an.Klp[j]; p < an.Klp[j + 1]; p++) w[an.Kli[p]] = Kl[p];
    for (int t = an.Rp[j]; t < an.Rp[j + 1]; t++) {
      int k = an.Rj[t], pos = an.Rpos[t];
      double ljk = qp.Lx[pos], f = ljk * qp.Dl[k];
      w[j] -= ljk * f;
      const int end = an.Lp[k + 1];
      for (int p = pos + 1; p < end; p++) w[an.Li[p]] -= qp.Lx[p] * f;
    }
    double d = w[j]; w[j] = 0.0;
    if (d == 0.0) { for (int p = an.Lp[j]; p < an.Lp[j + 1]; p++) w[an.Li[p]] = 0.0; return MI_OSQP_ERR_NONCONVEX; }
    if (d > 0.0) positive++;
    qp.Dl[j] = d;
    double dinv = 1.0 / d;
    qp.Dlinv[j] = dinv;
    for (int p = an.Lp[j]; p < an.Lp[j + 1]; p++) { int i = an.Li[p]; qp.Lx[p] = w[i] * dinv; w[i] = 0.0; }
  }
  invert_diag_blocks(an, qp.Lx.data());
  // dense tail reference: M = S^-1 = L22^-T D2^-1 L22^-1 from the complete factor (W = L22^-1 by forward substitution)
  qp.Minv.clear();
  if (an.dt.k) {
    const int k = an.dt.k, ts = an.dt.s;
    std::vector<double> W((size_t)k * k, 0.0);           // column-major, unit lower triangular
    for (int c = 0; c < k; c++) {
      double *wc = &W[(size_t)c * k];
      wc[c] = 1.0;
      for (int j = c; j < k; j++) {                      // column-oriented forward substitution on e_c
        const double v = wc[j];
        if (v == 0.0) continue;
        for (int p = an.Lp[ts + j]; p < an.Lp[ts + j + 1]; p++) wc[an.Li[p] - ts] -= qp.Lx[p] * v;
      }
    }
    qp.Minv.assign((size_t)k * k, 0.0);
    for (int a = 0; a < k; a++)
      for (int b = a; b < k; b++) {
        double sum = 0.0;
        for (int r = b; r < k; r++) sum += W[(size_t)a * k + r] * qp.Dlinv[ts + r] * W[(size_t)b * k + r];
        qp.Minv[(size_t)a * k + b] = qp.Minv[(size_t)b * k + a] = sum;
      }
  }
  return positive == n ? MI_OSQP_OK : MI_OSQP_ERR_NONCONVEX;
}

void direct_kkt_solve(const Analysis &an, const QPNumeric &qp, const double *rhs, double *sol) {
  int N = an.N;
  std::vector<double> b(N);
  for (int k = 0; k < N; k++) b[k] = rhs[an.perm[k]];
  for (int j = 0; j < N; j++) { double v = b[j]; for (int p = an.Lp[j]; p < an.Lp[j + 1]; p++) b[an.Li[p]] -= qp.Lx[p] * v; }
  for (int j = 0; j < N; j++) b[j] *= qp.Dlinv[j];
  for (int j = N - 1; j >= 0; j--) { double v = b[j]; for (int p = an.Lp[j]; p < an.Lp[j + 1]; p++) v -= qp.Lx[p] * b[an.Li[p]]; b[j] = v; }
  for (int k = 0; k < N; k++) sol[an.perm[k]] = b[k];
}

size_t phys_index(const Schedule &s, uint32_t slot, int b) { return (size_t)slot * s.bt + b; }

// Sequential interpreter of one schedule (tests): every wave walks its stream exactly as the device does;
// waves are interleaved epoch by epoch (epoch = number of barriers passed).  Returns false when the streams
// are not race-free / deadlock-free: barrier totals differ between waves, or within one epoch an entry of the
// solve vector is gathered and written, or written twice (the only read allowed next to a write is the
// writer's own read-modify-write) -- this is also what lets the device issue gathers ahead of flushes.
static bool replay(const Schedule &s, const double *canon, double *xs, size_t xs_len, bool tri, double *out) {
  const int nw = s.nw;
  std::vector<uint32_t> pos(nw), epoch(nw, 0u);
  for (int w = 0; w < nw; w++) pos[w] = s.wave_range[2 * w];
  // per entry of the vector: the epoch of its last write / read and who did it (-2 = several waves).  Within one epoch an
  // entry may be written and read by ONE wave (its LDS operations execute in order; the interpreter walks a wave's steps in
  // stream order too); any other mix of a write with another access in the same epoch is a race.
  std::vector<int64_t> wr_epoch(tri ? xs_len : 0, -1), rd_epoch(tri ? xs_len : 0, -1);
  std::vector<int> wr_wave(tri ? xs_len : 0, -1), rd_wave(tri ? xs_len : 0, -1);
  std::vector<std::vector<double>> acc(nw, std::vector<double>(64, 0.0));
  // gathers issued ahead (MI_D_AHEAD on step q = the gather of step q + MI_D_LOOKAHEAD happens at step q): the values a
  // later step will use are taken when the device takes them
  std::vector<std::vector<std::vector<double>>> ahead(nw);
  std::vector<uint32_t> ahead_first(nw, 0u);
  bool ok = true;
  int64_t cur = 0;
  auto gather = [&](int w, uint32_t e) {
    if (tri) {
      if (wr_epoch[e] == cur && wr_wave[e] != w) ok = false;
      if (rd_epoch[e] != cur) { rd_epoch[e] = cur; rd_wave[e] = w; } else if (rd_wave[e] != w) rd_wave[e] = -2;
    }
    return xs[e];
  };
  auto store = [&](int w, uint32_t e, double v) {
    if (tri) {
      if (wr_epoch[e] == cur && wr_wave[e] != w) ok = false;
      if (rd_epoch[e] == cur && rd_wave[e] != w) ok = false;
      wr_epoch[e] = cur; wr_wave[e] = w; xs[e] = v;
    } else out[e] = v;
  };
  for (;; cur++) {
    bool done = true;
    for (int w = 0; w < nw; w++) {
      while (pos[w] < s.wave_range[2 * w + 1]) {
        const uint32_t st = pos[w], d = s.step[st];
        if ((int64_t)epoch[w] + MI_D_NBAR(d) > cur) break;          // the wave waits at a barrier
        epoch[w] += MI_D_NBAR(d);
        if (MI_D_TYPE(d) == MI_D_TYPE_ROW) {
          const uint32_t lt = MI_D_LT(d), T = 1u << lt;
          std::vector<double> mine;
          if (d & MI_D_PRE) {                                      // gathered MI_D_LOOKAHEAD steps ago
            if (ahead[w].empty() || ahead_first[w] != st) return false;
            mine = std::move(ahead[w].front()); ahead[w].erase(ahead[w].begin()); ahead_first[w] = st + 1;      // (consecutive PRE steps follow consecutive AHEAD steps)
          }
          if (d & MI_D_AHEAD) {
            const uint32_t s2 = st + MI_D_LOOKAHEAD;
            if (s2 >= s.wave_range[2 * w + 1] || !(s.step[s2] & MI_D_PRE) || MI_D_NBAR(s.step[s2])) return false;
            std::vector<double> v(64, 0.0);
            for (uint32_t ln = 0; ln < 64; ln++) { const uint32_t slot = s2 * 64 + ln; if (s.src[slot] != MI_SRC_ZERO) v[ln] = gather(w, s.idx[slot]); }
            if (ahead[w].empty()) ahead_first[w] = s2;
            ahead[w].push_back(std::move(v));
          }
          for (uint32_t ln = 0; ln < 64; ln++) {
            const uint32_t slot = st * 64 + ln;
            if (s.src[slot] != MI_SRC_ZERO)
              acc[w][ln] += (s.src[slot] == MI_SRC_ONE ? 1.0 : canon[s.src[slot]]) * ((d & MI_D_PRE) ? mine[ln] : gather(w, s.idx[slot]));
          }
          if (d & MI_D_FLUSH) {
            for (uint32_t g = 0; g < 64 / T; g++) {
              double sum = 0.0;
              for (uint32_t ln = g * T; ln < (g + 1) * T; ln++) sum += acc[w][ln];
              const uint32_t row = s.outA[s.step_ob[st] + g];
              if (row != kNoRow) {
                const bool sub = tri && !(d & MI_D_STORE);
                store(w, row, sub ? gather(w, row) - sum : sum);
              }
            }
            std::fill(acc[w].begin(), acc[w].end(), 0.0);
          }
        }
        pos[w]++;
      }
      if (pos[w] < s.wave_range[2 * w + 1]) done = false;
    }
    if (done) break;
    if (cur > (int64_t)s.n_phases + 1) return false;             // some wave waits for a barrier that never comes
  }
  if (s.barriers)
    for (int w = 0; w < nw; w++) if (epoch[w] + s.tail_bar[w] != (uint32_t)s.n_phases) ok = false;
  return ok;
}

// Dataflow form (Schedule::dataflow): the waves advance round robin, one step at a time; a step whose gathers meet a
// "not yet" entry waits.  False when the streams deadlock, when an entry is written twice in one sweep or when a subtracting
// flush reads an entry the sweep itself writes (its old value must be there before the sweep starts).
static bool replay_dataflow(const Schedule &s, const double *canon, double *xs, size_t xs_len, const std::vector<char> &pending) {
  const int nw = s.nw;
  std::vector<uint32_t> pos(nw);
  for (int w = 0; w < nw; w++) pos[w] = s.wave_range[2 * w];
  std::vector<char> wait(pending), written(xs_len, 0);      // wait[e]: the sweep has yet to write e
  std::vector<std::vector<double>> acc(nw, std::vector<double>(64, 0.0));
  bool ok = true;
  for (;;) {
    bool done = true, progress = false;
    for (int w = 0; w < nw; w++) {
      if (pos[w] >= s.wave_range[2 * w + 1]) continue;
      done = false;
      const uint32_t st = pos[w], d = s.step[st];
      if (MI_D_NBAR(d)) return false;
      if (MI_D_TYPE(d) == MI_D_TYPE_ROW) {
        bool ready = true;
        for (uint32_t ln = 0; ln < 64 && ready; ln++) {
          const uint32_t slot = st * 64 + ln;
          if (s.src[slot] != MI_SRC_ZERO && wait[s.idx[slot]]) ready = false;
          if (s.src[slot] == MI_SRC_ZERO && s.idx[slot] != s.pad) ok = false;
        }
        if (!ready) continue;
        const uint32_t lt = MI_D_LT(d), T = 1u << lt;
        for (uint32_t ln = 0; ln < 64; ln++) {
          const uint32_t slot = st * 64 + ln;
          if (s.src[slot] != MI_SRC_ZERO) acc[w][ln] += (s.src[slot] == MI_SRC_ONE ? 1.0 : canon[s.src[slot]]) * xs[s.idx[slot]];
        }
        if (d & MI_D_FLUSH) {
          for (uint32_t g = 0; g < 64 / T; g++) {
            double sum = 0.0;
            for (uint32_t ln = g * T; ln < (g + 1) * T; ln++) sum += acc[w][ln];
            const uint32_t row = s.outA[s.step_ob[st] + g];
            if (row == kNoRow) continue;
            const bool sub = !(d & MI_D_STORE);
            const uint32_t dst = sub ? row + s.shadow : row;
            if (dst >= xs_len || written[dst] || !pending[dst] || (sub && (pending[row] || written[row]))) { ok = false; continue; }
            xs[dst] = sub ? xs[row] - sum : sum;
            written[dst] = 1; wait[dst] = 0;
          }
          std::fill(acc[w].begin(), acc[w].end(), 0.0);
        }
      }
      pos[w]++; progress = true;
    }
    if (done) break;
    if (!progress) return false;                              // every unfinished wave waits: deadlock
  }
  for (size_t e = 0; e < xs_len; e++) if (pending[e] && !written[e]) ok = false;     // a reset entry nobody writes would stall its readers
  return ok;
}

static bool replay_kkt_solve_df(const Analysis &an, const QPNumeric &qp, const double *rhs, double *sol) {
  const int N = an.N, SH = an.Next;
  std::vector<double> xs(an.xs_total, 0.0);
  std::vector<char> pend(an.xs_total, 0);
  // E6 as the device does it: right-hand side at p, "not yet" at every entry the forward sweep writes
  for (int k = 0; k < N; k++) {
    xs[k] = rhs[an.perm[k]];
    if (an.rflag[k] & 1) pend[k + SH] = 1;
    if (an.rflag[k] & 4) pend[an.xloc[k]] = 1;
  }
  bool ok = replay_dataflow(an.fwd, qp.Lx.data(), xs.data(), xs.size(), pend);
  // the middle: D^-1, results to xloc; "not yet" at every entry the backward sweep writes
  std::fill(pend.begin(), pend.end(), 0);
  for (int k = 0; k < N; k++) {
    const double v = xs[an.df_floc(k)] * qp.Dlinv[k];
    xs[an.xloc[k]] = v;
    if (an.rflag[k] & 2) pend[an.xloc[k] + SH] = 1;
    if (an.rflag[k] & 4) pend[k] = 1;
  }
  ok = replay_dataflow(an.bwd, qp.Lx.data(), xs.data(), xs.size(), pend) && ok;
  for (int k = 0; k < N; k++) sol[an.perm[k]] = xs[an.df_bloc(k)];
  return ok;
}

bool replay_kkt_solve(const Analysis &an, const QPNumeric &qp, const double *rhs, double *sol) {
  if (an.df) return replay_kkt_solve_df(an, qp, rhs, sol);
  int N = an.N;
  std::vector<double> xs(an.Next, 0.0);
  for (int k = 0; k < N; k++) xs[k] = rhs[an.perm[k]];
  bool ok = replay(an.fwd, qp.Lx.data(), xs.data(), xs.size(), true, nullptr);
  const int ts = an.dt.k ? an.dt.s : N;
  for (int k = 0; k < ts; k++) xs[an.xloc[k]] *= qp.Dlinv[k];
  if (an.dt.k) {
    std::vector<double> md(an.dt.k);
    for (int i = 0; i < an.dt.k; i++) md[i] = qp.Minv[(size_t)i * an.dt.k + i];
    ok = replay_dense_tail(an.dt, qp.Minv.data(), md.data(), xs.data() + ts) && ok;
  }
  ok = replay(an.bwd, qp.Lx.data(), xs.data(), xs.size(), true, nullptr) && ok;
  for (int k = 0; k < N; k++) sol[an.perm[k]] = xs[k];
  return ok;
}

int replay_block_factor(const Analysis &an, const Settings &st, const QPNumeric &qp, QPNumeric &out) {
  const BlockFactor &bf = an.bf;
  const int n = an.n, N = an.N;
  std::vector<double> S(bf.storage, 0.0), D(N, 0.0);
  for (int e = 0; e < an.nnzK(); e++) {
    uint32_t kind = bf.asm_src[e] >> 29, idx = bf.asm_src[e] & 0x1FFFFFFFu;
    double v = 0.0;
    switch (kind) {
      case ASM_P: v = qp.Pv[idx]; break;
      case ASM_P_SIGMA: v = qp.Pv[idx] + st.sigma; break;
      case ASM_SIGMA: v = st.sigma; break;
      case ASM_A: v = qp.Av[idx]; break;
      default: v = -qp.rho_inv[idx]; break;
    }
    S[bf.asm_dst[e]] = v;
  }
  auto B = [&](uint32_t id, uint32_t &off, uint32_t &r0, uint32_t &c0, uint32_t &h, uint32_t &w) {
    off = bf.blk[4 * id]; r0 = bf.blk[4 * id + 1]; c0 = bf.blk[4 * id + 2]; h = bf.blk[4 * id + 3] >> 8; w = bf.blk[4 * id + 3] & 255u;
  };
  int positive = 0;
  for (int L = 0; L < bf.n_levels; L++) {
    const uint32_t *lv = &bf.lvl[6 * L];
    for (uint32_t t = lv[0]; t < lv[1]; t++) {
      uint32_t off, r0, c0, h, w; B(bf.utask[4 * t], off, r0, c0, h, w);
      for (uint32_t q = bf.utask[4 * t + 1]; q < bf.utask[4 * t + 3]; q++) {
        uint32_t ao, ar, ac, ah, aw, bo, br, bc, bh, bw;
        B(bf.tri[2 * q], ao, ar, ac, ah, aw); B(bf.tri[2 * q + 1], bo, br, bc, bh, bw);
        for (uint32_t j = 0; j < w; j++) for (uint32_t i = 0; i < h; i++) {
          double acc = 0.0;
          for (uint32_t k = 0; k < aw; k++) acc += S[ao + k * ah + i] * D[ac + k] * S[bo + k * bh + j];
          S[off + j * h + i] -= acc;
        }
      }
    }
    for (uint32_t t = lv[2]; t < lv[3]; t++) {
      uint32_t off, r0, c0, h, w; B(bf.dtask[t], off, r0, c0, h, w);
      for (uint32_t j = 0; j < w; j++) {
        double d = S[off + j * h + j];
        if (d == 0.0) return MI_OSQP_ERR_NONCONVEX;
        if (d > 0.0) positive++;
        D[c0 + j] = d;
        for (uint32_t i = j + 1; i < h; i++) S[off + j * h + i] /= d;
        for (uint32_t k = j + 1; k < w; k++) for (uint32_t i = k; i < h; i++) S[off + k * h + i] -= S[off + j * h + i] * d * S[off + j * h + k];
      }
    }
    for (uint32_t t = lv[4]; t < lv[5]; t++) {
      uint32_t off, r0, c0, h, w, doff, dr, dc, dh, dw;
      B(bf.ttask[2 * t], off, r0, c0, h, w); B(bf.ttask[2 * t + 1], doff, dr, dc, dh, dw);
      for (uint32_t i = 0; i < h; i++)
        for (uint32_t j = 0; j < w; j++) {
          double v = S[off + j * h + i];
          for (uint32_t k = 0; k < j; k++) v -= S[off + k * h + i] * D[c0 + k] * S[doff + k * dh + j];
          S[off + j * h + i] = v / D[c0 + j];
        }
    }
  }
  // inverted diagonal blocks, exactly as factor_kernel stores them: inv(L_JJ)[i,k] (i > k) at (row k, col i) of B(J,J)
  for (uint32_t t = 0; t < bf.dtask.size(); t++) {
    uint32_t off, r0, c0, h, w; B(bf.dtask[t], off, r0, c0, h, w);
    // same recurrence and order as fct_diag: X[i,j] = -L[i,j] - sum_{p=j+1}^{i-1} X[i,p] L[p,j], j descending
    // (X[i,p] lives at (row p, col i) of the block, L[p,j] at (row p, col j))
    for (int j = (int)w - 2; j >= 0; j--)
      for (uint32_t i = (uint32_t)j + 1; i < w; i++) {
        double x = -S[off + (uint32_t)j * h + i];
        for (uint32_t p2 = (uint32_t)j + 1; p2 < i; p2++) x = std::fma(-S[off + i * h + p2], S[off + (uint32_t)j * h + p2], x);
        S[off + i * h + (uint32_t)j] = x;
      }
  }
  out.Lx.resize(an.nnzLx()); out.Dl = D; out.Dlinv.resize(N);
  for (int p = 0; p < an.nnzLx(); p++) out.Lx[p] = S[bf.lpos[p]];
  for (int j = 0; j < N; j++) out.Dlinv[j] = 1.0 / D[j];
  out.Minv.clear();
  if (an.dt.k) {
    // the tail blocks now hold the Schur complement; invert it the way dense_inverse_kernel does: symmetric sweep with
    // 16 x 16 pivot tiles  (P = inv(A_pp); G = A_:p P; A -= G A_p:; A_:p = G; A_pp = -P;  result = -inv(S))
    const int k = an.dt.k;
    const DenseTail &dt = an.dt;
    std::vector<double> A((size_t)k * k);
    {   // assembly of S exactly as tail_kernel does it: KKT block minus the quads of 4 source columns (one MFMA step each)
      std::vector<double> La((size_t)dt.n_lt + 1, 0.0), Dc(dt.n_ltcol);
      for (int e = 0; e < dt.n_lt; e++) La[e] = S[dt.lt_pos[e]];
      for (int ci = 0; ci < dt.n_ltcol; ci++) Dc[ci] = D[dt.ltcol_col[ci]];
      for (size_t t = 0; t < dt.tile_tab.size() / 4; t++) {
        const uint32_t I = dt.tile_tab[4 * t] >> 16, J = dt.tile_tab[4 * t] & 0xFFFFu, boff = dt.tile_tab[4 * t + 1];
        double tile[16][16];
        for (int r = 0; r < 16; r++) for (int c = 0; c < 16; c++) {
          const int rr = (I == J && r < c) ? c : r, cc = (I == J && r < c) ? r : c;
          tile[r][c] = S[boff + (uint32_t)cc * 16 + (uint32_t)rr];
        }
        for (uint32_t q = dt.tile_tab[4 * t + 2]; q < dt.tile_tab[4 * t + 3]; q++)
          for (int r = 0; r < 16; r++) for (int c = 0; c < 16; c++) {
            double acc = tile[r][c];
            for (int g = 0; g < 4; g++) {
              const double av = -La[dt.asm_q[(size_t)q * 64 + g * 16 + r] & 0xFFFFu];
              const double bv = La[dt.asm_q[(size_t)q * 64 + g * 16 + c] >> 16] * Dc[dt.asm_qcol[(size_t)q * 4 + g]];
              acc = std::fma(av, bv, acc);
            }
            tile[r][c] = acc;
          }
        for (int r = 0; r < 16; r++) for (int c = 0; c < 16; c++) {
          const size_t i = (size_t)I * 16 + r, j = (size_t)J * 16 + c;
          if (I != J || r >= c) A[j * k + i] = A[i * k + j] = tile[r][c];
        }
      }
    }
    std::vector<double> G((size_t)k * 16), C((size_t)k * 16);
    for (int p0 = 0; p0 < k; p0 += 16) {
      double T[16][16];
      for (int a = 0; a < 16; a++) for (int b = 0; b < 16; b++) T[a][b] = A[(size_t)(p0 + b) * k + p0 + a];
      for (int kk = 0; kk < 16; kk++) {                 // scalar sweep of the pivot tile: T <- -inv(T); its pivots are those of LDL'
        const double d = T[kk][kk];
        if (d == 0.0) return MI_OSQP_ERR_NONCONVEX;
        if (d > 0.0) positive++;
        const double di = 1.0 / d;
        for (int a = 0; a < 16; a++) for (int b = 0; b < 16; b++) if (a != kk && b != kk) T[a][b] -= T[a][kk] * di * T[kk][b];
        for (int a = 0; a < 16; a++) if (a != kk) { T[a][kk] *= di; T[kk][a] = T[a][kk]; }
        T[kk][kk] = -di;
      }
      for (int r = 0; r < k; r++)
        for (int c = 0; c < 16; c++) {
          const bool piv = r >= p0 && r < p0 + 16;
          C[(size_t)c * k + r] = piv ? 0.0 : A[(size_t)(p0 + c) * k + r];
        }
      for (int r = 0; r < k; r++)
        for (int c = 0; c < 16; c++) {
          double g = 0.0;
          for (int q = 0; q < 16; q++) g = std::fma(C[(size_t)q * k + r], -T[q][c], g);      // P = -T
          G[(size_t)c * k + r] = g;
        }
      for (int j = 0; j < k; j++) for (int i = 0; i < k; i++) {
        double acc = A[(size_t)j * k + i];
        for (int q = 0; q < 16; q++) acc = std::fma(-G[(size_t)q * k + i], C[(size_t)q * k + j], acc);
        A[(size_t)j * k + i] = acc;
      }
      for (int r = 0; r < k; r++) {
        if (r >= p0 && r < p0 + 16) continue;
        for (int c = 0; c < 16; c++) A[(size_t)(p0 + c) * k + r] = A[(size_t)r * k + p0 + c] = G[(size_t)c * k + r];
      }
      for (int a = 0; a < 16; a++) for (int b = 0; b < 16; b++) A[(size_t)(p0 + b) * k + p0 + a] = T[a][b];
    }
    out.Minv.resize((size_t)k * k);
    for (size_t e = 0; e < A.size(); e++) out.Minv[e] = -A[e];
  }
  return positive == n ? MI_OSQP_OK : MI_OSQP_ERR_NONCONVEX;
}

void replay_spmv(const Analysis &an, const QPNumeric &qp, const double *x, const double *y,
                 double *Px, double *Aty, double *Ax) {
  int n = an.n, m = an.m;
  std::vector<double> xs(n + m), out(2 * n + m, 0.0), pa(qp.Pv.size() + qp.Av.size());
  std::copy(x, x + n, xs.begin()); std::copy(y, y + m, xs.begin() + n);
  std::copy(qp.Pv.begin(), qp.Pv.end(), pa.begin());
  std::copy(qp.Av.begin(), qp.Av.end(), pa.begin() + qp.Pv.size());
  replay(an.chk, pa.data(), xs.data(), xs.size(), false, out.data());
  std::copy(out.begin(), out.begin() + n, Px);
  std::copy(out.begin() + n, out.begin() + 2 * n, Aty);
  std::copy(out.begin() + 2 * n, out.end(), Ax);
}

}  // namespace miosqp
