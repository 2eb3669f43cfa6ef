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
// 2. Stream layout.  The program is one byte stream of 16-byte-aligned records with
//    their operands inline (a bound's sphere, a leaf's inverse matrix), so that one
//    wave-uniform address gives everything a step needs in a single round of scalar
//    loads instead of word -> index -> record.
#include <cmath>
#include <cstddef>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <stdexcept>
#include <vector>

#include <map>

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
struct PNode {
  int a = -1, b = -1;    // children (operators)
  uint32_t word = 0;     // the leaf / operator word
  int bound = -1;        // index of the bound attached to this subtree, -1 if none
  uint32_t first = 0;    // index of the subtree's first program word (its bound word if it has one)
  uint32_t last = 0;     // index of its own word
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

std::vector<uint8_t> derive_stream_program(const void* blob) {
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

  Stream S;
  auto put_bound = [&](int idx, bool frontier) -> size_t {
    // whichever of the two volumes a random ray is less likely to meet (smaller surface area)
    const gmlrt_bound& g = bounds[idx];
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
        r.kind = kRecBox | (frontier ? kRecFlag : 0u);
        // the slab test is symmetric in lo and hi, so an empty solid (lo > hi) is stated as a
        // point no ray of interest can reach
        for (int i = 0; i < 3; i++) { r.lo[i] = empty ? 1e300 : g.lo[i]; r.hi[i] = empty ? 1e300 : g.hi[i]; }
        return S.put(&r, sizeof(r));
      }
    }
    ProgBound r;
    std::memset(&r, 0, sizeof(r));
    r.kind = kRecBound | (frontier ? kRecFlag : 0u);
    r.skip_to = 0;  // patched when the subtree is complete
    for (int i = 0; i < 3; i++) r.center[i] = bounds[idx].center[i];
    r.radius2 = bounds[idx].radius * bounds[idx].radius;
    r.radius = bounds[idx].radius;
    return S.put(&r, sizeof(r));
  };
  auto put_leaf = [&](uint32_t leaf, bool take) {
    ProgLeaf r;
    std::memset(&r, 0, sizeof(r));
    r.kind = kRecLeaf | (take ? kRecFlag : 0u) | (leaves[leaf].type << 16);
    r.leaf = leaf;
    std::memcpy(r.inv_m, leaves[leaf].inv_m, sizeof(r.inv_m));
    S.put(&r, sizeof(r));
  };
  auto put_small = [&](uint32_t kind, uint32_t arg) {
    ProgSmall r;
    std::memset(&r, 0, sizeof(r));
    r.kind = kind;
    r.arg = arg;
    S.put(&r, sizeof(r));
  };
  auto patch_skip = [&](size_t at) {
    const uint32_t to = (uint32_t)S.bytes.size();
    std::memcpy(S.bytes.data() + at + offsetof(ProgBound, skip_to), &to, 4);
  };

  // explicit stack instead of recursion: union chains can be tens of thousands deep
  struct Item { int node; int stage; size_t bound_at; };
  std::vector<Item> work;
  work.push_back({st[0], 0, 0});
  while (!work.empty()) {
    Item it = work.back();
    work.pop_back();
    if (it.stage == 1) {  // close a frontier bound: it covers everything emitted since it was opened
      patch_skip(it.bound_at);
      continue;
    }
    const PNode& n = tree[(size_t)it.node];
    const bool is_union = (n.word >> 30) == 1 && (n.word & 3u) == GMLRT_UNION;
    if (n.bound >= 0) work.push_back({it.node, 1, put_bound(n.bound, true)});
    if (is_union) {
      // pushed in reverse so that the left child is emitted first
      work.push_back({n.b, 0, 0});
      work.push_back({n.a, 0, 0});
    } else if ((n.word >> 30) == 0) {
      put_leaf(n.word & 0x3FFFFFFFu, true);  // frontier leaf: candidate for the nearest Entry, no list
    } else {
      // frontier child with an operator at its top: its words in order, then take the first element
      const uint32_t begin = n.bound >= 0 ? n.first + 1 : n.first;
      std::vector<std::pair<uint32_t, size_t>> inner;  // (word index one past the subtree, record offset)
      for (uint32_t i = begin; i <= n.last; i++) {
        const uint32_t w = nodes[i];
        if ((w >> 30) == 2) inner.emplace_back(i + 1 + (w & 0x7FFFu), put_bound((int)((w >> 15) & 0x7FFFu), false));
        else if ((w >> 30) == 0) put_leaf(w & 0x3FFFFFFFu, false);
        else put_small(kRecOp, w & 0xFu);
        while (!inner.empty() && inner.back().first == i + 1) {
          patch_skip(inner.back().second);
          inner.pop_back();
        }
      }
      put_small(kRecTake, 0);
    }
  }
  put_small(kRecEnd, 0);
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
  std::map<std::pair<uint32_t, uint32_t>, uint32_t> seen;
  leaf_surf->assign(H.n_leaves, kNoSurfProg);
  for (uint32_t i = 0; i < H.n_leaves; i++) {
    if (leaves[i].surf_code == GMLRT_SURF_CONST) continue;
    const auto key = std::make_pair(leaves[i].surf_code, leaves[i].surf_const);
    auto it = seen.find(key);
    if (it == seen.end()) {
      it = seen.emplace(key, (uint32_t)(out.size() / sizeof(SurfInstr))).first;
      expand_surface(code + key.first, consts + key.second, out);
    }
    (*leaf_surf)[i] = it->second;
  }
  if (out.empty()) out.resize(sizeof(SurfInstr), 0);  // a lone END record: never run, keeps the buffer non-empty
  return out;
}

}  // namespace gmlrt
