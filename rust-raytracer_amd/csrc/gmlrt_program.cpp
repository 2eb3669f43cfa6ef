// Derives the traversal program the kernel actually runs from a validated scene blob.
//
// 1. First-element shortcut.  trace() and the shadow test look only at the FIRST
//    element of the root's intersection list (raytracer.rs:52-58, 74-79).  For a
//    union the first element of the result is the nearer of its operands' first
//    elements, the left one on a tie: the sweep of primitives.rs:262-313 meets that
//    Entry first, the rule turns true, and the span it opens is at least as long as
//    the operand's own first span, so the 1e-10 sliver rule cannot pop it — provided
//    that operand span is itself not a sliver, which holds for every operator output;
//    a two-event primitive list can be one, and the kernel re-runs such a ray through
//    the full list algorithm.  So the maximal tree of unions hanging from the root
//    needs no lists at all: its non-union children ("frontier children") are
//    evaluated one after the other and only the nearest first Entry is kept.
//
//    The order in which frontier children are visited is free as long as a tie between two
//    candidates at the same distance still goes to the one that comes first in the blob's
//    program; leaves are numbered in program order, so the kernel breaks ties by leaf index.
//    That freedom is used to regroup the frontier children spatially: a binary tree of boxes
//    over their bounds (the cut of the centroid order, along any axis, that the surface-area heuristic
//    prices lowest; nearer to the camera first), emitted in depth-first order with skip offsets.  The blob's own bounds on
//    union nodes of the frontier are not used (a chain of unions bounds little).
//
// 2. Stream layout.  The program is one byte stream of 16-byte-aligned records with
//    their operands inline (a bound's sphere, a leaf's inverse matrix), so that one
//    wave-uniform address gives everything a step needs in a single round of scalar
//    loads instead of word -> index -> record.
#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstddef>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <stdexcept>
#include <vector>

#include <map>
#include <string>
#include <tuple>

#include "gmlrt_device.hpp"
#include "gmlrt_scene.hpp"
#include "gmlrt_surface_vm.h"

namespace gmlrt {

namespace {
// A box is chosen over the sphere when its area is below this fraction of the sphere's
// (its test costs somewhat more).  The environment override exists for tuning runs.
double box_bias() {
  static const double v = [] {
    const char* e = std::getenv("GMLRT_BOX_BIAS");
    return e && *e ? std::atof(e) : 0.8;
  }();
  return v;
}
// Lazy operand evaluation inside frontier children (skip records, cheaper intersection operand first); the
// environment override exists for tuning runs.
bool lazy_operands() {
  static const bool v = [] { const char* e = std::getenv("GMLRT_LAZY_OPERANDS"); return !(e && *e == '0'); }();
  return v;
}
bool sah_split() {
  static const bool v = [] { const char* e = std::getenv("GMLRT_SAH"); return !(e && *e == '0'); }();  // tuning runs
  return v;
}
struct PNode {
  int a = -1, b = -1;    // children (operators)
  uint32_t word = 0;     // the leaf / operator word
  int bound = -1;        // index of the bound attached to this subtree, -1 if none
  uint32_t first = 0;    // index of the subtree's first program word (its bound word if it has one)
  uint32_t last = 0;     // index of its own word
  uint32_t nl = 1;       // leaves below (and including) this node
};

struct Stream {
  std::vector<uint8_t> bytes;
  size_t put(const void* p, size_t n) {
    const size_t at = bytes.size();
    bytes.insert(bytes.end(), (const uint8_t*)p, (const uint8_t*)p + n);
    return at;
  }
  void pad16() { while (bytes.size() % 16) bytes.push_back(0); }
};
}  // namespace

std::vector<uint8_t> derive_stream_program(const void* blob, uint32_t* need_stack, uint64_t* need_events, uint32_t* slice_off) {
  gmlrt_blob_header H;
  std::memcpy(&H, blob, sizeof(H));
  const uint8_t* base = (const uint8_t*)blob;
  const uint32_t* nodes = (const uint32_t*)(base + H.off_nodes);
  const gmlrt_leaf* leaves = (const gmlrt_leaf*)(base + H.off_leaves);
  const gmlrt_bound* bounds = (const gmlrt_bound*)(base + H.off_bounds);

  // ---- rebuild the tree
  std::vector<PNode> tree;
  std::vector<int> st;
  struct Open { uint32_t end; int bound; uint32_t at; };
  std::vector<Open> open;
  for (uint32_t i = 0; i < H.n_nodes; i++) {
    const uint32_t w = nodes[i];
    const uint32_t kind = w >> 30;
    if (kind == 2) {
      open.push_back({i + 1 + (w & 0x7FFFu), (int)((w >> 15) & 0x7FFFu), i});
      continue;
    }
    PNode n;
    n.word = w;
    n.first = n.last = i;
    if (kind == 1) {
      n.b = st.back(); st.pop_back();
      n.a = st.back(); st.pop_back();
      n.first = tree[(size_t)n.a].first;
      n.nl = tree[(size_t)n.a].nl + tree[(size_t)n.b].nl;
    }
    while (!open.empty() && open.back().end == i + 1) {
      n.bound = open.back().bound;  // the validator guarantees this is exactly this subtree
      n.first = open.back().at;
      open.pop_back();
    }
    st.push_back((int)tree.size());
    tree.push_back(n);
  }
  if (st.size() != 1) throw std::runtime_error("derive_stream_program: malformed program");

  *need_stack = 1;
  *need_events = 2;
  Stream S;
  // Expected number of frontier bound tests per ray, for choosing between two arrangements of
  // the frontier: a bound is tested when the one around it was entered, which a random ray
  // does with a probability proportional to that one's surface area.
  double cost = 0.0, ref_area = 0.0;
  std::vector<double> open_p;
  auto smaller_area = [](const gmlrt_bound& g) {
    double a = 4.0 * 3.14159265358979323846 * g.radius * g.radius;
    const double e[3] = {g.hi[0] - g.lo[0], g.hi[1] - g.lo[1], g.hi[2] - g.lo[2]};
    if (e[0] < 0.0 || e[1] < 0.0 || e[2] < 0.0) return 0.0;
    const double b = 2.0 * (e[0] * e[1] + e[1] * e[2] + e[2] * e[0]);
    return std::isfinite(b) && b < a ? b : a;
  };
  auto put_bound = [&](const gmlrt_bound& g, bool frontier) -> size_t {
    if (frontier) {
      cost += open_p.empty() ? 1.0 : open_p.back();
      const double pr = ref_area > 0.0 ? smaller_area(g) / ref_area : 1.0;
      open_p.push_back(pr < 1.0 ? pr : 1.0);
    }
    // whichever of the two volumes a random ray is less likely to meet (smaller surface area)
    bool box_ok = true;
    for (int i = 0; i < 3; i++) box_ok = box_ok && std::isfinite(g.lo[i]) && std::isfinite(g.hi[i]);
    if (box_ok) {
      const double e[3] = {g.hi[0] - g.lo[0], g.hi[1] - g.lo[1], g.hi[2] - g.lo[2]};
      const bool empty = e[0] < 0.0 || e[1] < 0.0 || e[2] < 0.0;
      const double box_area = 2.0 * (e[0] * e[1] + e[1] * e[2] + e[2] * e[0]);
      const double sphere_area = 4.0 * 3.14159265358979323846 * g.radius * g.radius;
      if (empty || box_area < box_bias() * sphere_area) {
        ProgBox r;
        std::memset(&r, 0, sizeof(r));
        r.kind = kRecBox | kRecIsBox | (frontier ? kRecFlag : 0u);
        // the slab test is symmetric in lo and hi, so an empty solid (lo > hi) is stated as a
        // point no ray of interest can reach
        for (int i = 0; i < 3; i++) {
          const double lo = empty ? 1e30 : g.lo[i], hi = empty ? 1e30 : g.hi[i];
          // to single precision, outwards (a value beyond the float range becomes +-infinity: "no information")
          float flo = (float)lo, fhi = (float)hi;
          if ((double)flo > lo) flo = std::nextafterf(flo, -INFINITY);
          if ((double)fhi < hi) fhi = std::nextafterf(fhi, INFINITY);
          r.lohi[2 * i] = flo;
          r.lohi[2 * i + 1] = fhi;
          r.mag[i] = std::nextafterf(std::max(std::fabs(flo), std::fabs(fhi)), INFINITY);
        }
        return S.put(&r, sizeof(r));
      }
    }
    ProgBound r;
    std::memset(&r, 0, sizeof(r));
    r.kind = kRecBound | kRecIsBound | (frontier ? kRecFlag : 0u);
    r.skip_to = 0;  // patched when the subtree is complete
    for (int i = 0; i < 3; i++) r.center[i] = g.center[i];
    r.radius2 = g.radius * g.radius;
    r.radius = g.radius;
    return S.put(&r, sizeof(r));
  };
  auto put_leaf = [&](uint32_t leaf, bool take) {
    ProgLeaf r;
    std::memset(&r, 0, sizeof(r));
    r.kind = kRecLeaf | kRecIsLeaf | (take ? kRecFlag : 0u) | (leaves[leaf].type << 16);
    r.leaf = leaf;
    std::memcpy(r.inv_m, leaves[leaf].inv_m, sizeof(r.inv_m));
    S.put(&r, sizeof(r));
  };
  auto put_small = [&](uint32_t kind, uint32_t arg) {
    ProgSmall r;
    std::memset(&r, 0, sizeof(r));
    r.kind = kind | (kind == kRecOp ? kRecIsOp : kind == kRecTake ? kRecIsTake : kind == kRecSkipEmpty ? kRecIsSkip : 0u);
    r.arg = arg;
    S.put(&r, sizeof(r));
  };
  auto patch_skip = [&](size_t at) {
    const uint32_t to = (uint32_t)S.bytes.size();
    std::memcpy(S.bytes.data() + at + offsetof(ProgBound, skip_to), &to, 4);
  };
  auto close_frontier_bound = [&](size_t at) {
    patch_skip(at);
    open_p.pop_back();
  };

  // ---- the frontier: operands of the union tree under the root, in program order
  // (explicit stack instead of recursion: union chains can be tens of thousands deep)
  std::vector<int> items;
  {
    std::vector<int> work{st[0]};
    while (!work.empty()) {
      const int ni = work.back();
      work.pop_back();
      const PNode& n = tree[(size_t)ni];
      if ((n.word >> 30) == 1 && (n.word & 3u) == GMLRT_UNION) {
        work.push_back(n.b);  // reversed: the left child comes off the stack first
        work.push_back(n.a);
      } else {
        items.push_back(ni);
      }
    }
  }
  // reference area for the cost estimate: the box around everything that has a bound
  {
    gmlrt_bound all;
    for (int k = 0; k < 3; k++) { all.lo[k] = INFINITY; all.hi[k] = -INFINITY; all.center[k] = 0.0; }
    all.radius = INFINITY;
    for (int ni : items) {
      if (tree[(size_t)ni].bound < 0) continue;
      const gmlrt_bound& b = bounds[tree[(size_t)ni].bound];
      for (int k = 0; k < 3; k++) { all.lo[k] = std::min(all.lo[k], b.lo[k]); all.hi[k] = std::max(all.hi[k], b.hi[k]); }
    }
    ref_area = smaller_area(all);
    if (!std::isfinite(ref_area)) ref_area = 0.0;
  }
  // centres of the frontier children's boxes (the sorts below compare them many times)
  std::vector<double> cent(tree.size() * 3, 0.0);
  for (int ni : items)
    if (tree[(size_t)ni].bound >= 0) {
      const gmlrt_bound& g = bounds[tree[(size_t)ni].bound];
      for (int k = 0; k < 3; k++) cent[(size_t)ni * 3 + (size_t)k] = 0.5 * (g.lo[k] + g.hi[k]);
    }
  std::vector<size_t> skip_open;  // skip records waiting for the offset of the record after their operator
  auto emit_item = [&](int ni) {
    const PNode& n = tree[(size_t)ni];
    size_t bound_at = 0;
    if (n.bound >= 0) bound_at = put_bound(bounds[n.bound], true);
    if ((n.word >> 30) == 0) {
      put_leaf(n.word & 0x3FFFFFFFu, true);  // frontier leaf: candidate for the nearest Entry, no list
    } else {
      // Frontier child with an operator at its top: its subtree in evaluation order, then take the first
      // element.  Two liberties are taken with the blob's postfix order, neither of which changes a list:
      // * after the first operand of an intersection or a difference comes a skip record: when that operand's
      //   list is empty for every ray of the wave, so is the operator's result whatever the other operand
      //   yields (primitives.rs:177-187: a && b, a && !b), and the other operand's records are jumped over;
      // * an intersection may evaluate its right operand first when that one is the cheaper to find empty
      //   (fewer leaves; bounded rather than a half-space).  The operator record then says so (kOpSwapped)
      //   and the merge gives ties to the list that was pushed second, which is the reference's left operand.
      struct Step { int node; int stage; size_t a, b; };
      std::vector<Step> work;
      std::vector<uint64_t> sim;  // list capacities on the operand stack
      uint64_t sim_sp = 0;
      work.push_back({ni, 0, 0, 0});
      auto n_leaves_of = [&](int x) { return tree[(size_t)x].nl; };
      auto unbounded = [&](int x) {
        const PNode& q = tree[(size_t)x];
        return (q.word >> 30) == 0 && leaves[q.word & 0x3FFFFFFFu].type == GMLRT_PLANE;
      };
      while (!work.empty()) {
        Step stp = work.back();
        work.pop_back();
        const PNode& q = tree[(size_t)stp.node];
        if (stp.stage == 0) {
          size_t bound_rec = (size_t)-1;
          if (q.bound >= 0 && stp.node != ni) bound_rec = put_bound(bounds[q.bound], false);
          if ((q.word >> 30) == 0) {
            put_leaf(q.word & 0x3FFFFFFFu, false);
            if (bound_rec != (size_t)-1) patch_skip(bound_rec);
            // capacity the kernel's stacks need for this evaluation order (worst case: every leaf yields two events)
            sim.emplace_back(2);
            sim_sp += 2;
            *need_events = std::max(*need_events, sim_sp);
            *need_stack = std::max(*need_stack, (uint32_t)sim.size());
            continue;
          }
          const uint32_t rule = q.word & 3u;
          bool swapped = false;
          if (rule == GMLRT_INTERSECT && lazy_operands()) {
            const uint64_t score_a = (uint64_t)n_leaves_of(q.a) + (unbounded(q.a) ? 1000000u : 0u);
            const uint64_t score_b = (uint64_t)n_leaves_of(q.b) + (unbounded(q.b) ? 1000000u : 0u);
            swapped = score_b < score_a;
          }
          const int first = swapped ? q.b : q.a, second = swapped ? q.a : q.b;
          // popped in reverse: first operand, (skip record), second operand, operator, close bound
          work.push_back({stp.node, 3, bound_rec, swapped ? 1u : 0u});
          work.push_back({second, 0, 0, 0});
          if (rule != GMLRT_UNION && lazy_operands()) work.push_back({stp.node, 1, 0, 0});
          work.push_back({first, 0, 0, 0});
        } else if (stp.stage == 1) {
          skip_open.push_back(S.bytes.size());
          put_small(kRecSkipEmpty, 0);
        } else {
          const uint32_t rule = q.word & 3u;
          // (the leaf-operand flags of the blob's word are not used by the traversal)
          put_small(kRecOp, rule | (stp.b ? kOpSwapped : 0u));
          {
            const uint64_t lb = sim.back(); sim.pop_back();
            const uint64_t la = sim.back(); sim.pop_back();
            *need_events = std::max(*need_events, sim_sp + la + lb);  // the merge writes its output above both operands
            sim.push_back(la + lb);
          }
          if (rule != GMLRT_UNION && lazy_operands()) {
            const uint32_t to = (uint32_t)S.bytes.size();
            std::memcpy(S.bytes.data() + skip_open.back() + offsetof(ProgSmall, arg), &to, 4);
            skip_open.pop_back();
          }
          if (stp.a != (size_t)-1) patch_skip(stp.a);
        }
      }
      put_small(kRecTake, 0);
    }
    if (n.bound >= 0) close_frontier_bound(bound_at);
  };

  // Arrangement 1: the union tree as the blob has it, with the bounds the blob put on its nodes
  // (a scene built recursively is a good hierarchy already).
  auto emit_as_given = [&]() {
    struct Item { int node; int stage; size_t bound_at; };
    std::vector<Item> work;
    work.push_back({st[0], 0, 0});
    while (!work.empty()) {
      Item it = work.back();
      work.pop_back();
      if (it.stage == 1) {  // close a union's bound: it covers everything emitted since it was opened
        close_frontier_bound(it.bound_at);
        continue;
      }
      const PNode& n = tree[(size_t)it.node];
      if ((n.word >> 30) == 1 && (n.word & 3u) == GMLRT_UNION) {
        if (n.bound >= 0) work.push_back({it.node, 1, put_bound(bounds[n.bound], true)});
        work.push_back({n.b, 0, 0});  // reversed: the left child is emitted first
        work.push_back({n.a, 0, 0});
      } else {
        emit_item(it.node);
      }
    }
  };

  // Arrangement 2: the frontier children regrouped spatially (objects added to the scene one by
  // one make a chain of unions, which bounds little).  Children without a finite bound
  // (half-spaces) are visited by every ray, and so is, in effect, a child that spans most of the
  // scene (a floor slab): both stay outside the tree and come first, so that a hit on them can
  // already cull what lies behind it.
  auto emit_regrouped = [&](const std::vector<int>& its) {
  std::vector<int> grouped;
  {
    double lo[3] = {INFINITY, INFINITY, INFINITY}, hi[3] = {-INFINITY, -INFINITY, -INFINITY};
    auto box_area = [](const double* l, const double* h) {
      const double e[3] = {h[0] - l[0], h[1] - l[1], h[2] - l[2]};
      if (e[0] < 0.0 || e[1] < 0.0 || e[2] < 0.0) return 0.0;
      return 2.0 * (e[0] * e[1] + e[1] * e[2] + e[2] * e[0]);
    };
    for (int ni : items) {  // (all of them, also for a slice: what counts as "spans the scene" does not depend on the slice)
      if (tree[(size_t)ni].bound < 0) continue;
      const gmlrt_bound& b = bounds[tree[(size_t)ni].bound];
      for (int k = 0; k < 3; k++) { lo[k] = std::min(lo[k], b.lo[k]); hi[k] = std::max(hi[k], b.hi[k]); }
    }
    const double all_area = box_area(lo, hi);
    for (int ni : its) {
      const int bi = tree[(size_t)ni].bound;
      if (bi < 0 || !(box_area(bounds[bi].lo, bounds[bi].hi) <= 0.25 * all_area)) emit_item(ni);
      else grouped.push_back(ni);
    }
  }
  struct Range { size_t lo, hi; int stage; size_t bound_at; };
  auto centroid = [&](int ni, int axis) { return cent[(size_t)ni * 3 + (size_t)axis]; };
  const double cam[3] = {0.0, 0.0, -1.0};  // raytracer.rs:138
  std::vector<Range> ranges;
  ranges.push_back({0, grouped.size(), 0, 0});
  while (!ranges.empty()) {
    Range r = ranges.back();
    ranges.pop_back();
    if (r.stage == 1) { close_frontier_bound(r.bound_at); continue; }
    const size_t n = r.hi - r.lo;
    if (n == 0) continue;
    if (n == 1) { emit_item(grouped[r.lo]); continue; }
    // box of the group and of its centroids
    gmlrt_bound g;
    double clo[3], chi[3];
    for (int k = 0; k < 3; k++) { g.lo[k] = clo[k] = INFINITY; g.hi[k] = chi[k] = -INFINITY; }
    for (size_t i = r.lo; i < r.hi; i++) {
      const gmlrt_bound& b = bounds[tree[(size_t)grouped[i]].bound];
      for (int k = 0; k < 3; k++) {
        g.lo[k] = std::min(g.lo[k], b.lo[k]);
        g.hi[k] = std::max(g.hi[k], b.hi[k]);
        const double c = centroid(grouped[i], k);
        clo[k] = std::min(clo[k], c);
        chi[k] = std::max(chi[k], c);
      }
    }
    bool finite = true;
    double h2 = 0.0;
    for (int k = 0; k < 3; k++) {
      finite = finite && std::isfinite(g.lo[k]) && std::isfinite(g.hi[k]);
      g.center[k] = 0.5 * (g.lo[k] + g.hi[k]);
      const double h = 0.5 * (g.hi[k] - g.lo[k]);
      h2 += h * h;
    }
    g.radius = std::sqrt(h2) * (1.0 + 1e-9) + 1e-12;  // the box's circumscribed sphere
    int axis = 0;
    for (int k = 1; k < 3; k++) if (chi[k] - clo[k] > chi[axis] - clo[axis]) axis = k;
    size_t mid = r.lo + n / 2;
    auto by_centroid = [&](int ax) {
      return [&, ax](int x, int y) {
        const double cx = centroid(x, ax), cy = centroid(y, ax);
        return cx < cy || (cx == cy && x < y);
      };
    };
    if (sah_split()) {
      // surface-area heuristic: along each axis, the cut of the centroid order that minimises
      // area(left) * count(left) + area(right) * count(right)
      auto area_of = [](const double* l, const double* h) {
        const double e[3] = {h[0] - l[0], h[1] - l[1], h[2] - l[2]};
        if (e[0] < 0.0 || e[1] < 0.0 || e[2] < 0.0) return 0.0;
        return 2.0 * (e[0] * e[1] + e[1] * e[2] + e[2] * e[0]);
      };
      double best_cost = INFINITY;
      int best_axis = axis;
      size_t best_cut = n / 2;
      std::vector<int> tmp(grouped.begin() + (long)r.lo, grouped.begin() + (long)r.hi);
      std::vector<double> right_area(n + 1, 0.0);
      for (int ax = 0; ax < 3; ax++) {
        std::sort(tmp.begin(), tmp.end(), by_centroid(ax));
        double l[3] = {INFINITY, INFINITY, INFINITY}, h[3] = {-INFINITY, -INFINITY, -INFINITY};
        for (size_t i = n; i-- > 0;) {
          const gmlrt_bound& b = bounds[tree[(size_t)tmp[i]].bound];
          for (int k = 0; k < 3; k++) { l[k] = std::min(l[k], b.lo[k]); h[k] = std::max(h[k], b.hi[k]); }
          right_area[i] = area_of(l, h);
        }
        for (int k = 0; k < 3; k++) { l[k] = INFINITY; h[k] = -INFINITY; }
        for (size_t i = 0; i + 1 < n; i++) {
          const gmlrt_bound& b = bounds[tree[(size_t)tmp[i]].bound];
          for (int k = 0; k < 3; k++) { l[k] = std::min(l[k], b.lo[k]); h[k] = std::max(h[k], b.hi[k]); }
          const double c = area_of(l, h) * (double)(i + 1) + right_area[i + 1] * (double)(n - i - 1);
          if (c < best_cost) { best_cost = c; best_axis = ax; best_cut = i + 1; }
        }
      }
      axis = best_axis;
      mid = r.lo + best_cut;
      std::sort(grouped.begin() + (long)r.lo, grouped.begin() + (long)r.hi, by_centroid(axis));
    } else {
    std::nth_element(grouped.begin() + (long)r.lo, grouped.begin() + (long)mid, grouped.begin() + (long)r.hi, by_centroid(axis));
    }
    // the half whose centroids lie nearer to the camera goes first
    double dl = 0.0, dr = 0.0;
    for (int k = 0; k < 3; k++) {
      double ml = 0.0, mr = 0.0;
      for (size_t i = r.lo; i < mid; i++) ml += centroid(grouped[i], k);
      for (size_t i = mid; i < r.hi; i++) mr += centroid(grouped[i], k);
      ml = ml / (double)(mid - r.lo) - cam[k];
      mr = mr / (double)(r.hi - mid) - cam[k];
      dl += ml * ml;
      dr += mr * mr;
    }
    if (dr < dl) std::rotate(grouped.begin() + (long)r.lo, grouped.begin() + (long)mid, grouped.begin() + (long)r.hi);
    const size_t cut = dr < dl ? r.lo + (r.hi - mid) : mid;
    // the root group of a scene that is bounded as a whole still gets its box: rays that miss everything leave at once
    if (finite && std::isfinite(g.radius)) ranges.push_back({0, 0, 1, put_bound(g, true)});
    ranges.push_back({cut, r.hi, 0, 0});  // popped second
    ranges.push_back({r.lo, cut, 0, 0});  // popped first
  }
  };

  // emit both, keep the one with the smaller expected number of bound tests
  const auto t_a = std::chrono::steady_clock::now();
  emit_as_given();
  const auto t_b = std::chrono::steady_clock::now();
  const double cost_given = cost;
  std::vector<uint8_t> given;
  given.swap(S.bytes);
  cost = 0.0;
  open_p.clear();
  emit_regrouped(items);
  const auto t_c = std::chrono::steady_clock::now();
  bool regroup = cost < cost_given;
  if (const char* e = std::getenv("GMLRT_REGROUP")) regroup = *e == '1';  // tuning runs
  if (std::getenv("GMLRT_VERBOSE"))
    std::fprintf(stderr, "[gmlrt] frontier of %zu children: expected bound tests per ray %.2f as given, %.2f regrouped -> %s\n",
                 items.size(), cost_given, cost, regroup ? "regrouped" : "as given");
  if (!regroup) S.bytes.swap(given);
  put_small(kRecEnd, 0);
  // ---- slices: the same frontier dealt out to kProgSlices programs of their own, for the four waves of a block
  // that trace the SAME rays together (each wave walks one slice, the nearest candidate of the four is the
  // nearest of the whole frontier: the rule that picks it, distance then leaf index, does not care in which
  // order or by whom the candidates were found).  Children are dealt round-robin in the order of the
  // regrouped arrangement, neighbours to different slices, so that the pieces a ray passes are spread over
  // the waves; every slice gets its own hierarchy of boxes.  Offsets in a slice are offsets into the whole
  // buffer, like everywhere else.
  for (uint32_t k = 0; k < kProgSlices; k++) slice_off[k] = 0;
  static const bool no_slices = std::getenv("GMLRT_NO_SLICES") != nullptr;  // tuning runs
  if (!no_slices && items.size() >= 2 * (size_t)kProgSlices && tree[(size_t)st[0]].nl >= 32) {
    // spatial order: a walk through the emitted frontier, children in the order their first leaf appears
    std::vector<int> spatial;
    {
      std::map<uint32_t, int> by_first_leaf;
      for (int ni : items) {
        int q = ni;
        while ((tree[(size_t)q].word >> 30) == 1) q = tree[(size_t)q].a;
        by_first_leaf[tree[(size_t)q].word & 0x3FFFFFFFu] = ni;
      }
      std::vector<char> seen(tree.size(), 0);
      for (size_t at = 0; at + 16 <= S.bytes.size();) {
        uint32_t kind, leaf;
        std::memcpy(&kind, S.bytes.data() + at, 4);
        std::memcpy(&leaf, S.bytes.data() + at + 4, 4);
        const uint32_t kk = kind & 0xFu;
        if (kk == kRecEnd) break;
        if (kk == kRecLeaf) {
          auto it = by_first_leaf.find(leaf);
          if (it != by_first_leaf.end() && !seen[(size_t)it->second]) { seen[(size_t)it->second] = 1; spatial.push_back(it->second); }
        }
        at += kk == kRecLeaf ? sizeof(ProgLeaf) : (kk == kRecBound || kk == kRecBox) ? sizeof(ProgBound) : sizeof(ProgSmall);
      }
      for (int ni : items) if (!seen[(size_t)ni]) spatial.push_back(ni);  // (a lazily swapped first operand hides a child's first leaf)
    }
    // Which slice a child goes to: the one where it adds least to  sum over the slice's children of
    // leaves(child) * leaves(other) / (distance^2 + size^2)  -- children of one slice end up far from each other
    // in every direction (a ray meets neighbours, not the whole board, and the waves should share what it
    // meets), and the slices hold about the same number of leaves.  Children without a finite box are met by
    // every ray: they are dealt round-robin.
    std::vector<std::vector<int>> deal(kProgSlices);
    {
      struct Placed { double c[3], size2, nl; };
      std::vector<std::vector<Placed>> placed(kProgSlices);
      std::vector<double> load(kProgSlices, 0.0);
      uint32_t rr = 0;
      for (int ni : spatial) {
        Placed me;
        me.nl = (double)tree[(size_t)ni].nl;
        bool finite = tree[(size_t)ni].bound >= 0;
        if (finite) {
          const gmlrt_bound& g = bounds[tree[(size_t)ni].bound];
          me.size2 = 0.0;
          for (int k = 0; k < 3; k++) { me.c[k] = 0.5 * (g.lo[k] + g.hi[k]); const double e = g.hi[k] - g.lo[k]; me.size2 += e * e; }
          finite = std::isfinite(me.c[0]) && std::isfinite(me.c[1]) && std::isfinite(me.c[2]) && std::isfinite(me.size2);
        }
        // (quadratic in the number of children, a division each: a frontier of hundreds is dealt round-robin in its spatial order)
        if (spatial.size() > 128 || !finite) { deal[rr++ % kProgSlices].push_back(ni); continue; }
        uint32_t best = 0;
        double best_v = INFINITY;
        for (uint32_t k = 0; k < kProgSlices; k++) {
          double v = 1e-9 * load[k];
          for (const Placed& o : placed[k]) {
            const double d2 = (me.c[0] - o.c[0]) * (me.c[0] - o.c[0]) + (me.c[1] - o.c[1]) * (me.c[1] - o.c[1]) + (me.c[2] - o.c[2]) * (me.c[2] - o.c[2]);
            v += o.nl / (d2 + 0.25 * (me.size2 + o.size2) + 1e-300);
          }
          if (v < best_v) { best_v = v; best = k; }
        }
        deal[best].push_back(ni);
        placed[best].push_back(me);
        load[best] += me.nl;
      }
    }
    for (uint32_t k = 0; k < kProgSlices; k++) {
      std::vector<int>& mine = deal[k];
      std::sort(mine.begin(), mine.end());  // program order within the slice, as the whole frontier has it
      while (S.bytes.size() % 64) S.bytes.push_back(0);
      slice_off[k] = (uint32_t)S.bytes.size();
      open_p.clear();
      emit_regrouped(mine);
      put_small(kRecEnd, 0);
      if (std::getenv("GMLRT_VERBOSE")) {
        uint32_t nl = 0;
        for (int ni : mine) nl += tree[(size_t)ni].nl;
        std::fprintf(stderr, "[gmlrt] slice %u: %zu children, %u leaves, at byte %u\n", k, mine.size(), nl, slice_off[k]);
      }
    }
  }
  if (std::getenv("GMLRT_VERBOSE")) {
    auto ms = [](std::chrono::steady_clock::time_point a, std::chrono::steady_clock::time_point b) { return std::chrono::duration<double, std::milli>(b - a).count(); };
    std::fprintf(stderr, "[gmlrt] derivation: as given %.3f ms, regrouped %.3f ms, slices %.3f ms\n", ms(t_a, t_b), ms(t_b, t_c), ms(t_c, std::chrono::steady_clock::now()));
    size_t n_kind[8] = {0};
    for (size_t at = 0; at + 16 <= S.bytes.size();) {
      uint32_t kind;
      std::memcpy(&kind, S.bytes.data() + at, 4);
      const uint32_t k = kind & 0xFu;
      n_kind[k & 7]++;
      if (k == kRecEnd) break;
      at += k == kRecLeaf ? sizeof(ProgLeaf) : (k == kRecBound || k == kRecBox) ? sizeof(ProgBound) : sizeof(ProgSmall);
    }
    std::fprintf(stderr, "[gmlrt] program: %zu bytes; records: %zu boxes, %zu spheres, %zu leaves, %zu operators, %zu skip-if-empty, %zu takes\n",
                 S.bytes.size(), n_kind[kRecBox], n_kind[kRecBound], n_kind[kRecLeaf], n_kind[kRecOp], n_kind[kRecSkipEmpty], n_kind[kRecTake]);
  }
  S.bytes.resize(S.bytes.size() + 64, 0);  // the traversal reads 64 bytes at every record, the last one included
  if (S.bytes.size() >= 0xFFFFFFF0ull) throw std::runtime_error("derived program too large");
  return S.bytes;
}

std::vector<uint8_t> derive_surface_stream(const void* blob, std::vector<uint32_t>* leaf_surf) {
  gmlrt_blob_header H;
  std::memcpy(&H, blob, sizeof(H));
  const uint8_t* base = (const uint8_t*)blob;
  const gmlrt_leaf* leaves = (const gmlrt_leaf*)(base + H.off_leaves);
  const uint64_t* code = (const uint64_t*)(base + H.off_code);
  const uint64_t* consts = (const uint64_t*)(base + H.off_consts);
  std::vector<uint8_t> out;
  // (code, constants, face) -> first record; and record bytes -> first record, so that programs
  // that come out the same (a function that never looks at `face`, two faces that take the same
  // arm) are stored and, more to the point, run once per wave
  std::map<std::tuple<uint32_t, uint32_t, uint32_t>, uint32_t> seen;
  std::map<std::string, uint32_t> by_content;
  leaf_surf->assign((size_t)H.n_leaves * kSurfFaces, kNoSurfProg);
  for (uint32_t i = 0; i < H.n_leaves; i++) {
    if (leaves[i].surf_code == GMLRT_SURF_CONST) continue;
    // faces a primitive of this type can report (primitives.rs:403-412, 544-555, 726-734, 897-904, 989-996)
    const uint32_t n_faces = leaves[i].type == GMLRT_CUBE ? 6u : leaves[i].type == GMLRT_CYLINDER ? 3u
                             : leaves[i].type == GMLRT_CONE ? 2u : 1u;
    std::vector<uint8_t> general;
    for (uint32_t f = 0; f < n_faces; f++) {
      const auto key = std::make_tuple(leaves[i].surf_code, leaves[i].surf_const, f);
      auto it = seen.find(key);
      if (it == seen.end()) {
        if (general.empty()) expand_surface(code + leaves[i].surf_code, consts + leaves[i].surf_const, general);
        std::vector<uint8_t> spec;
        specialize_surface(general.data(), (int64_t)f, spec);
        const std::string bytes((const char*)spec.data(), spec.size());
        auto ct = by_content.find(bytes);
        if (ct == by_content.end()) {
          ct = by_content.emplace(bytes, (uint32_t)(out.size() / sizeof(SurfInstr))).first;
          out.insert(out.end(), spec.begin(), spec.end());
        }
        it = seen.emplace(key, ct->second).first;
      }
      (*leaf_surf)[(size_t)i * kSurfFaces + f] = it->second;
    }
  }
  if (std::getenv("GMLRT_VERBOSE")) {
    size_t general_records = 0;
    std::map<std::pair<uint32_t, uint32_t>, int> distinct;
    for (uint32_t i = 0; i < H.n_leaves; i++)
      if (leaves[i].surf_code != GMLRT_SURF_CONST && distinct.emplace(std::make_pair(leaves[i].surf_code, leaves[i].surf_const), 0).second) {
        std::vector<uint8_t> g;
        expand_surface(code + leaves[i].surf_code, consts + leaves[i].surf_const, g);
        general_records += g.size() / sizeof(SurfInstr);
      }
    std::fprintf(stderr, "[gmlrt] surface programs: %zu as compiled (%zu records), %zu after specialising per face (%zu records)\n",
                 distinct.size(), general_records, by_content.size(), out.size() / sizeof(SurfInstr));
  }
  if (out.empty()) out.resize(sizeof(SurfInstr), 0);  // a lone END record: never run, keeps the buffer non-empty
  return out;
}

}  // namespace gmlrt
