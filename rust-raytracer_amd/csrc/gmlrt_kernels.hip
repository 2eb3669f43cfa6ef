// HIP kernels of the GML `render` path for gfx950 (wave64).
//
// One ray per lane.  Every lane runs a small state machine over the rays of its
// pixel (camera ray, one shadow ray per light that faces the hit, mirror bounce;
// /root/reference/src/raytracer.rs:44-111) and all ray kinds share one scene
// traversal, so a wave traverses 64 rays at a time whatever their kind.  A lane
// whose pixel is finished pulls the next pixel from a global counter.
//
// The traversal evaluates the CSG tree as a postfix program that is the same for
// every ray, so control flow is wave-uniform and leaf data is fetched with scalar
// loads; what differs per lane is the list of entry/exit events each node yields
// (primitives.rs:253-315), kept on a per-lane stack in device memory.
//
// Arithmetic is f64 in the reference's operation order and the file is built with
// -ffp-contract=off, so intermediate values equal the CPU restatement's bit for
// bit; only libm-class functions (pow, atan2, asin ...) may differ in the last ulp.
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstring>

#include "gmlrt_device.hpp"
#include "gmlrt_surface_vm.h"

namespace gmlrt {
namespace {

constexpr int kBlock = 256;
constexpr int kWave = 64;
constexpr uint32_t kLeafMask = 0xFFFFFu;
constexpr uint32_t kFaceShift = 20;
constexpr uint32_t kOrigExit = 1u << 23;  // the primitive produced this event as an Exit
constexpr uint32_t kCurExit = 1u << 24;   // current type after operator switches
constexpr int kCtxFixed = 27;             // f64 fields of the shading context before the bounce stack
constexpr int kCtxPerLevel = 7;

enum RayKind : int { kIdle = 0, kView = 1, kShadow = 2 };

struct KArgs {
  const gmlrt_leaf* leaves;
  const uint32_t* nodes;
  const gmlrt_light* lights;
  const uint64_t* code;
  const uint64_t* consts;
  uint32_t n_leaves, n_nodes, n_lights, max_stack;
  double amb[3];
  double c_x, c_y, pw;  // camera, raytracer.rs:131-136
  int32_t depth;
  int32_t w, h;
  int32_t tile_rows, part, nparts;
  int32_t rows_part;     // rows this partition renders
  uint32_t tiles_x;      // 8x8 pixel tiles per row band
  uint32_t n_slots;      // tiles_x * tiles_y * 64 pixel slots (padding included)
  uint8_t* rgb8;
  double* rgbf;
  double* ev_dist;
  uint32_t* ev_meta;
  uint32_t ev_cap;       // event slots per lane
  double* ctx;
  uint32_t ctx_fields;
  unsigned int* next_slot;
  unsigned long long* ray_counts;  // primary, shadow, reflect, shaded hits
};

struct V3 { double x, y, z; };
__device__ inline V3 vadd(V3 a, V3 b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
__device__ inline V3 vsub(V3 a, V3 b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
__device__ inline V3 vmul(V3 a, double s) { return {a.x * s, a.y * s, a.z * s}; }
__device__ inline V3 vneg(V3 a) { return {-a.x, -a.y, -a.z}; }
__device__ inline V3 vcmul(V3 a, V3 b) { return {a.x * b.x, a.y * b.y, a.z * b.z}; }
__device__ inline double vdot(V3 a, V3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
__device__ inline V3 vnormalize(V3 a) { return vmul(a, 1.0 / sqrt(vdot(a, a))); }

// ---------------------------------------------------------------- primitives
// Ray into object space: primitives.rs:350-354 (same prologue in all five).
__device__ inline void to_object(const double* __restrict__ im, V3 o, V3 d, double& scale, V3& rd, V3& rp) {
  V3 td = {im[0] * d.x + im[1] * d.y + im[2] * d.z, im[4] * d.x + im[5] * d.y + im[6] * d.z,
           im[8] * d.x + im[9] * d.y + im[10] * d.z};
  scale = 1.0 / sqrt(vdot(td, td));
  rd = vmul(td, scale);
  rp = {im[0] * o.x + im[1] * o.y + im[2] * o.z + im[3], im[4] * o.x + im[5] * o.y + im[6] * o.z + im[7],
        im[8] * o.x + im[9] * o.y + im[10] * o.z + im[11]};
}

// Result of one primitive test: bit 0 = an Entry event (tA, fA), bit 1 = an Exit event (tB, fB).
struct PrimHit { int bits; double tA, tB; int fA, fB; };

// primitives.rs:355-393
__device__ inline PrimHit isect_sphere(V3 p, V3 d) {
  PrimHit r = {0, 0.0, 0.0, 0, 0};
  double s = vdot(vneg(p), d);
  double lsq = vdot(p, p);
  if (s < 0.0 && lsq > 1.0) return r;
  double msq = lsq - s * s;
  if (msq > 1.0) return r;
  double q = sqrt(1.0 - msq);
  double t1 = s + q, t2 = s - q;
  if (t1 > t2) { double t = t1; t1 = t2; t2 = t; }
  if (t1 > 0.0) { r.bits |= 1; r.tA = t1; }
  if (t2 > 0.0) { r.bits |= 2; r.tB = t2; }
  return r;
}

// primitives.rs:468-534; SLABS = [(3,2),(4,5),(1,0)]
__device__ inline PrimHit isect_cube(V3 rp, V3 d) {
  PrimHit r = {0, 0.0, 0.0, 0, 0};
  const double eps = 1e-15;
  bool has = false;
  double tmin = 0.0, tmax = 0.0;
  int fmin = 0, fmax = 0;
  const double pe[3] = {0.5 - rp.x, 0.5 - rp.y, 0.5 - rp.z};
  const double df[3] = {d.x, d.y, d.z};
  const int f1s[3] = {3, 4, 1}, f2s[3] = {2, 5, 0};
#pragma unroll
  for (int i = 0; i < 3; i++) {
    int face1 = f1s[i], face2 = f2s[i];
    const double e = pe[i], f = df[i];
    if (fabs(f) > eps) {
      double t1 = (e + 0.5) / f, t2 = (e - 0.5) / f;
      if (t1 > t2) { double t = t1; t1 = t2; t2 = t; int ff = face1; face1 = face2; face2 = ff; }
      if (!(has && tmin >= t1)) { tmin = t1; fmin = face1; }
      if (!(has && tmax <= t2)) { tmax = t2; fmax = face2; }
      has = true;
      if (tmin > tmax) return r;
      if (tmax < 0.0) return r;
    } else if (-e - 0.5 > 0.0 || -e + 0.5 < 0.0) {
      return r;
    }
  }
  if (has && tmin > 0.0) { r.bits |= 1; r.tA = tmin; r.fA = fmin; }
  if (has && tmax > 0.0) { r.bits |= 2; r.tB = tmax; r.fB = fmax; }
  return r;
}

// (t, face) pairs ordered like Rust tuples
__device__ inline bool tf_gt(double ta, int fa, double tb, int fb) { return ta != tb ? ta > tb : fa > fb; }
__device__ inline bool tf_lt(double ta, int fa, double tb, int fb) { return ta != tb ? ta < tb : fa < fb; }

// primitives.rs:592-606 and 778-792: both roots carry face 0
__device__ inline bool solve_quadratic(double a, double b, double c, double& t1, double& t2) {
  double sq = b * b - 4.0 * a * c;
  if (sq < 0.0) return false;
  double root = sqrt(sq);
  double r1 = (-b - root) / (2.0 * a), r2 = (-b + root) / (2.0 * a);
  if (r1 > r2) { double t = r1; r1 = r2; r2 = t; }
  t1 = r1; t2 = r2;
  return true;
}

// primitives.rs:609-621: bottom cap face 2, top cap face 1
__device__ inline void cyl_caps(double py, double dy, double& ta, int& fa, double& tb, int& fb) {
  double dinv = 1.0 / dy;
  double t1 = -py * dinv, t2 = (-py + 1.0) * dinv;
  ta = t1; fa = 2; tb = t2; fb = 1;
  if (t1 > t2) { ta = t2; fa = 1; tb = t1; fb = 2; }
}

// primitives.rs:633-717
__device__ inline PrimHit isect_cylinder(V3 p, V3 d) {
  PrimHit r = {0, 0.0, 0.0, 0, 0};
  const double eps = 1e-7;
  double tmin, tmax;
  int fmin, fmax;
  if (fabs(d.y) + eps >= 1.0) {
    double frsd = 1.0 - p.x * p.x - p.z * p.z;
    if (frsd < 0.0) return r;
    cyl_caps(p.y, d.y, tmin, fmin, tmax, fmax);
  } else if (fabs(d.y) < eps) {
    if (p.y < 0.0 || p.y > 1.0) return r;
    if (!solve_quadratic(d.x * d.x + d.z * d.z, 2.0 * (p.x * d.x + p.z * d.z), p.x * p.x + p.z * p.z - 1.0, tmin, tmax))
      return r;
    fmin = 0; fmax = 0;
  } else {
    double tc1, tc2;
    if (!solve_quadratic(d.x * d.x + d.z * d.z, 2.0 * (p.x * d.x + p.z * d.z), p.x * p.x + p.z * p.z - 1.0, tc1, tc2))
      return r;
    cyl_caps(p.y, d.y, tmin, fmin, tmax, fmax);
    if (tf_gt(tc1, 0, tmin, fmin)) { tmin = tc1; fmin = 0; }
    if (tf_lt(tc2, 0, tmax, fmax)) { tmax = tc2; fmax = 0; }
    if (tmin > tmax) return r;
    if (tmax < 0.0) return r;
  }
  if (tmin > 0.0) { r.bits |= 1; r.tA = tmin; r.fA = fmin; }
  if (tmax > 0.0) { r.bits |= 2; r.tB = tmax; r.fB = fmax; }
  return r;
}

// primitives.rs:805-887
__device__ inline PrimHit isect_cone(V3 p, V3 d) {
  PrimHit r = {0, 0.0, 0.0, 0, 0};
  double tc1, tc2;
  if (!solve_quadratic(d.x * d.x + d.z * d.z - d.y * d.y, 2.0 * (p.x * d.x + p.z * d.z - p.y * d.y),
                       p.x * p.x + p.z * p.z - p.y * p.y, tc1, tc2))
    return r;
  const double cminy = p.y + tc1 * d.y, cmaxy = p.y + tc2 * d.y;
  const bool k1 = 0.0 <= cminy && cminy <= 1.0, k2 = 0.0 <= cmaxy && cmaxy <= 1.0;
  double ta, tb;
  int fa = 0, fb = 0;
  if (k1 && k2) {
    ta = tc1; tb = tc2;
  } else if (k1 || k2) {
    // one hit on the mantle, the other on the base cap (face 1); two-element sort
    // that swaps when the cap is nearer or the comparison is unordered
    ta = k1 ? tc1 : tc2;
    tb = (-p.y + 1.0) / d.y;
    fb = 1;
    if (!(tb >= ta)) { double t = ta; ta = tb; tb = t; fa = 1; fb = 0; }
  } else {
    return r;
  }
  if (ta > 0.0) { r.bits |= 1; r.tA = ta; r.fA = fa; }
  if (tb > 0.0) { r.bits |= 2; r.tB = tb; r.fB = fb; }
  return r;
}

// primitives.rs:945-981: a single event, Entry when the ray heads into y <= 0
__device__ inline PrimHit isect_plane(V3 p, V3 d) {
  PrimHit r = {0, 0.0, 0.0, 0, 0};
  const double denom = d.y;
  if (fabs(denom) < 1e-7) return r;
  const double t = -p.y / denom;
  if (t < 0.0) return r;
  if (denom > 0.0) { r.bits = 2; r.tB = t; }
  else { r.bits = 1; r.tA = t; }
  return r;
}

__device__ inline PrimHit isect_prim(uint32_t type, V3 rp, V3 rd) {
  switch (type) {
    case GMLRT_SPHERE: return isect_sphere(rp, rd);
    case GMLRT_CUBE: return isect_cube(rp, rd);
    case GMLRT_CYLINDER: return isect_cylinder(rp, rd);
    case GMLRT_CONE: return isect_cone(rp, rd);
    default: return isect_plane(rp, rd);
  }
}

// (u, v) of primitives.rs:403-412, 544-555, 726-734, 897-904, 989-996
__device__ inline void prim_uv(uint32_t type, int face, V3 o, double& u, double& v) {
  switch (type) {
    case GMLRT_SPHERE: v = (o.y + 1.0) / 2.0; u = atan2(o.x, o.z); break;
    case GMLRT_CUBE:
      if (face <= 1) { u = o.x; v = o.y; }
      else if (face <= 3) { u = o.z; v = o.y; }
      else { u = o.x; v = o.z; }
      break;
    case GMLRT_CYLINDER:
    case GMLRT_CONE:
      if (face == 0) { u = atan2(o.x, o.z); v = o.y; }
      else { u = (o.x + 1.0) / 2.0; v = (o.z + 1.0) / 2.0; }
      break;
    default: u = o.x; v = o.z; break;
  }
}

// object-space normal of primitives.rs:414-420, 557-565, 735-743, 906-913, 998-1006
__device__ inline V3 prim_normal(uint32_t type, int face, V3 o) {
  switch (type) {
    case GMLRT_SPHERE: return o;
    case GMLRT_CUBE:
      switch (face) {
        case 0: return {0.0, 0.0, -1.0};
        case 1: return {0.0, 0.0, 1.0};
        case 2: return {-1.0, 0.0, 0.0};
        case 3: return {1.0, 0.0, 0.0};
        case 4: return {0.0, 1.0, 0.0};
        default: return {0.0, -1.0, 0.0};
      }
    case GMLRT_CYLINDER:
      if (face == 0) return {o.x, 0.0, o.z};
      return face == 1 ? V3{0.0, 1.0, 0.0} : V3{0.0, -1.0, 0.0};
    case GMLRT_CONE:
      if (face == 0) return {o.x, -o.y, o.z};
      return {0.0, 1.0, 0.0};
    default: return {0.0, 1.0, 0.0};
  }
}

// ---------------------------------------------------------------- CSG traversal
struct Hit { bool any; double dist; uint32_t meta; };

__device__ inline bool csg_rule(uint32_t rule, bool a, bool b) {
  return rule == GMLRT_UNION ? (a || b) : rule == GMLRT_INTERSECT ? (a && b) : (a && !b);
}

// Sort-then-sweep of primitives.rs:262-313 for two non-empty lists that sit one
// above the other on the lane's event stack (A at a0, B right after it).  Both
// inputs are already sorted, so the stable sort of their concatenation is a merge
// that takes A on ties.  Output goes above B and is then moved down onto A.
__device__ inline uint32_t merge_lists(uint32_t rule, double* __restrict__ evD, uint32_t* __restrict__ evM, uint32_t a0,
                                       uint32_t la, uint32_t lb) {
  const uint32_t b0 = a0 + la, out0 = b0 + lb;
  uint32_t i = 0, j = 0, k = 0;
  bool in1 = false, in2 = false, inside = false;
  double prevt = 0.0;
  double da = evD[(size_t)a0 * kWave], db = evD[(size_t)b0 * kWave];
  uint32_t ma = evM[(size_t)a0 * kWave], mb = evM[(size_t)b0 * kWave];
  const uint32_t total = la + lb;
  for (uint32_t n = 0; n < total; n++) {
    const bool takeA = (j >= lb) || (i < la && db >= da);
    const double d = takeA ? da : db;
    const uint32_t m = takeA ? ma : mb;
    if (takeA) {
      i++;
      if (i < la) { da = evD[(size_t)(a0 + i) * kWave]; ma = evM[(size_t)(a0 + i) * kWave]; }
    } else {
      j++;
      if (j < lb) { db = evD[(size_t)(b0 + j) * kWave]; mb = evM[(size_t)(b0 + j) * kWave]; }
    }
    const bool entry = !(m & kCurExit);
    if (takeA) in1 = entry; else in2 = entry;
    const bool ni = csg_rule(rule, in1, in2);
    if (inside && !ni) {
      if (d - prevt < 1e-10) {
        k--;  // sliver: drop the Entry that opened it
      } else {
        evD[(size_t)(out0 + k) * kWave] = d;
        evM[(size_t)(out0 + k) * kWave] = m | kCurExit;
        k++;
      }
    }
    if (!inside && ni) {
      evD[(size_t)(out0 + k) * kWave] = d;
      evM[(size_t)(out0 + k) * kWave] = m & ~kCurExit;
      k++;
      prevt = d;
    }
    inside = ni;
  }
  for (uint32_t n = 0; n < k; n++) {
    evD[(size_t)(a0 + n) * kWave] = evD[(size_t)(out0 + n) * kWave];
    evM[(size_t)(a0 + n) * kWave] = evM[(size_t)(out0 + n) * kWave];
  }
  return k;
}

// Node::intersect of the whole scene for one ray per lane; returns the first
// element of the root list (all that raytracer.rs:52-58, 74-79 look at).
// `lens` holds, per operand-stack level and lane, the length of that list; bit 15
// marks a two-event primitive list shorter than 1e-10 (it vanishes in any operator
// whose other operand is empty, primitives.rs:295-298).
__device__ inline Hit intersect_scene(const KArgs& A, bool active, V3 o, V3 d, double* __restrict__ evD,
                                      uint32_t* __restrict__ evM, uint16_t* __restrict__ lens) {
  Hit h = {false, 0.0, 0u};
  if (A.n_nodes == 1) {
    // The root is a bare primitive: its raw list is what trace sees, a lone Exit included.
    const gmlrt_leaf* L = A.leaves;
    double scale; V3 rd, rp;
    to_object(L->inv_m, o, d, scale, rd, rp);
    PrimHit ph = isect_prim(L->type, rp, rd);
    if (active && ph.bits) {
      h.any = true;
      if (ph.bits & 1) { h.dist = scale * ph.tA; h.meta = ((uint32_t)ph.fA << kFaceShift); }
      else { h.dist = scale * ph.tB; h.meta = ((uint32_t)ph.fB << kFaceShift) | kOrigExit | kCurExit; }
    }
    return h;
  }
  uint32_t sp = 0;     // events on this lane's stack
  uint32_t level = 0;  // operand stack depth (wave-uniform)
  const int tid = threadIdx.x;
  for (uint32_t pc = 0; pc < A.n_nodes; pc++) {
    const uint32_t word = A.nodes[pc];
    if ((word >> 30) == 0) {
      const gmlrt_leaf* L = A.leaves + (word & 0x3FFFFFFFu);
      double scale; V3 rd, rp;
      to_object(L->inv_m, o, d, scale, rd, rp);
      PrimHit ph = isect_prim(L->type, rp, rd);
      uint32_t n = 0;
      // a list that starts with Exit never raises its operand's counter above zero
      // (primitives.rs:254-293 start both counters at 0): same as no list at all
      if (active && (ph.bits & 1)) {
        const uint32_t leaf = word & kLeafMask;
        const double d0 = scale * ph.tA;
        evD[(size_t)sp * kWave] = d0;
        evM[(size_t)sp * kWave] = leaf | ((uint32_t)ph.fA << kFaceShift);
        n = 1;
        if (ph.bits & 2) {
          const double d1 = scale * ph.tB;
          evD[(size_t)(sp + 1) * kWave] = d1;
          evM[(size_t)(sp + 1) * kWave] = leaf | ((uint32_t)ph.fB << kFaceShift) | kOrigExit | kCurExit;
          n = (d1 - d0 < 1e-10) ? (2u | 0x8000u) : 2u;
        }
      }
      lens[level * kBlock + tid] = (uint16_t)n;
      sp += n & 0x7FFFu;
      level++;
    } else {
      const uint32_t rule = word & 3u;
      const uint32_t lenB = lens[(level - 1) * kBlock + tid], lenA = lens[(level - 2) * kBlock + tid];
      const uint32_t lb = lenB & 0x7FFFu, la = lenA & 0x7FFFu;
      const bool slvA = lenA >> 15, slvB = lenB >> 15;
      const uint32_t a0 = sp - la - lb;
      uint32_t lout = 0;
      bool general = false;
      if (rule == GMLRT_UNION) {
        if (lb == 0) lout = slvA ? 0 : la;
        else if (la == 0) lout = slvB ? 0 : lb;  // B already sits at a0
        else general = true;
      } else if (rule == GMLRT_INTERSECT) {
        general = la != 0 && lb != 0;
      } else {
        if (la != 0) {
          if (lb == 0) lout = slvA ? 0 : la;
          else general = true;
        }
      }
      if (general) lout = merge_lists(rule, evD, evM, a0, la, lb);
      lens[(level - 2) * kBlock + tid] = (uint16_t)lout;
      sp = a0 + lout;
      level--;
    }
  }
  const uint32_t n = lens[tid] & 0x7FFFu;
  if (active && n) {
    h.any = true;
    h.dist = evD[0];
    h.meta = evM[0];
  }
  return h;
}

// ---------------------------------------------------------------- lights (lights.rs:42-120)
__device__ inline void light_direction(const gmlrt_light& l, V3 pos, V3& dir, bool& has_dist, double& dist) {
  if (l.kind == GMLRT_LIGHT_DIRECTIONAL) {
    dir = {l.dir[0], l.dir[1], l.dir[2]};
    has_dist = false;
    dist = 0.0;
    return;
  }
  V3 dd = vsub({l.pos[0], l.pos[1], l.pos[2]}, pos);
  double dl = sqrt(vdot(dd, dd));
  dir = vmul(dd, 1.0 / dl);
  has_dist = true;
  dist = dl;
}
__device__ inline V3 light_intensity(const gmlrt_light& l, V3 pos) {
  const V3 col = {l.color[0], l.color[1], l.color[2]};
  if (l.kind == GMLRT_LIGHT_DIRECTIONAL) return col;
  V3 dd = vsub({l.pos[0], l.pos[1], l.pos[2]}, pos);
  double dsq = vdot(dd, dd);
  if (l.kind == GMLRT_LIGHT_POINT) return vmul(col, 100.0 / (99.0 + dsq));
  const V3 D = {l.dir[0], l.dir[1], l.dir[2]};
  double invlend = 1.0 / sqrt(dsq);
  double cosangle = -vdot(dd, D) * invlend;
  if (cosangle < l.coscutoff) return {0.0, 0.0, 0.0};
  double i = pow(vdot(D, vmul(dd, -invlend)), l.exp);
  return vmul(col, i * 100.0 / (99.0 + dsq));
}

// raytracer.rs:21-25
__device__ inline uint8_t quantise(double p) {
  double v = 255.0 * fmin(fmax(p, 0.0), 1.0);
  if (!(v == v)) return 0;
  return (uint8_t)v;
}

#define CTX(f) ctx[(size_t)(f) * kWave]
__device__ inline V3 ctx_get3(const double* ctx, int f) { return {CTX(f), CTX(f + 1), CTX(f + 2)}; }
__device__ inline void ctx_set3(double* ctx, int f, V3 v) { CTX(f) = v.x; CTX(f + 1) = v.y; CTX(f + 2) = v.z; }
// context fields
enum { F_POS = 0, F_NRM = 3, F_SC = 6, F_KD = 9, F_KS = 10, F_N = 11, F_C = 12, F_DIFF = 15, F_SPEC = 18,
       F_RPOS = 21, F_RDIR = 24 };

__global__ __launch_bounds__(kBlock) void render_kernel(KArgs A) {
  extern __shared__ uint16_t lens[];
  const int tid = threadIdx.x;
  const int lane = tid & (kWave - 1);
  const size_t wave = (size_t)blockIdx.x * (kBlock / kWave) + (size_t)(tid / kWave);
  double* __restrict__ evD = A.ev_dist + wave * A.ev_cap * kWave + lane;
  uint32_t* __restrict__ evM = A.ev_meta + wave * A.ev_cap * kWave + lane;
  double* __restrict__ ctx = A.ctx + wave * A.ctx_fields * kWave + lane;

  int kind = kIdle;
  uint32_t pix = 0;   // output index (row-in-partition * w + x)
  int level = 0;      // bounce level of the current view ray
  uint32_t li = 0;    // light being tested
  V3 o = {0, 0, 0}, d = {0, 0, 1};
  uint32_t n_primary = 0, n_shadow = 0, n_reflect = 0, n_hits = 0;
  bool more = true;   // wave-uniform: pixel slots left to hand out

  for (;;) {
    // ---- hand new pixels to idle lanes
    if (more) {
      const unsigned long long idle = __ballot(kind == kIdle);
      if (idle) {
        const uint32_t cnt = (uint32_t)__popcll(idle);
        uint32_t base = 0;
        if (lane == 0) base = atomicAdd(A.next_slot, cnt);
        base = (uint32_t)__shfl((int)base, 0);
        if (kind == kIdle) {
          const uint32_t slot = base + (uint32_t)__popcll(idle & ((1ull << lane) - 1ull));
          if (slot < A.n_slots) {
            // slot -> 8x8 tile -> pixel of this partition's packed rows
            const uint32_t tile = slot >> 6, in = slot & 63u;
            const uint32_t x = (tile % A.tiles_x) * 8u + (in & 7u);
            const uint32_t yl = (tile / A.tiles_x) * 8u + (in >> 3);
            if (x < (uint32_t)A.w && yl < (uint32_t)A.rows_part) {
              uint32_t y = yl;
              if (A.nparts > 1) {
                const uint32_t k = yl / (uint32_t)A.tile_rows;
                y = ((uint32_t)A.part + k * (uint32_t)A.nparts) * (uint32_t)A.tile_rows + yl % (uint32_t)A.tile_rows;
              }
              // raytracer.rs:138-142
              const double xf = (double)x, yf = (double)y;
              V3 dir = {A.c_x + (xf + 0.5) * A.pw, A.c_y - (yf + 0.5) * A.pw, 1.0};
              o = {0.0, 0.0, -1.0};
              d = vnormalize(dir);
              kind = kView;
              level = 0;
              pix = yl * (uint32_t)A.w + x;
              n_primary++;
            }
          }
        }
        if (base + cnt >= A.n_slots) more = false;
      }
    }
    if (!__ballot(kind != kIdle)) {
      if (!more) break;
      continue;
    }

    // ---- one scene traversal for whatever ray each lane holds
    const Hit h = intersect_scene(A, kind != kIdle, o, d, evD, evM, lens);

    bool advance = false, finish = false;
    V3 C = {0.0, 0.0, 0.0};  // value of trace() at `level` when finish is set

    if (kind == kView) {
      if (!h.any || (h.meta & kCurExit)) {
        finish = true;  // raytracer.rs:55-57, 108-110
      } else {
        // ---- raytracer.rs:58-68: surface, position, normal of the hit
        const gmlrt_leaf* L = A.leaves + (h.meta & kLeafMask);
        const int face = (int)((h.meta >> kFaceShift) & 7u);
        double scale; V3 rd, rp;
        to_object(L->inv_m, o, d, scale, rd, rp);
        const uint32_t type = L->type;
        PrimHit ph = isect_prim(type, rp, rd);
        const double od = (h.meta & kOrigExit) ? ph.tB : ph.tA;
        const V3 opos = vadd(rp, vmul(rd, od));
        double u, v;
        prim_uv(type, face, opos, u, v);
        double s6[6];
        const uint32_t mycode = L->surf_code;
        const uint64_t* kc = A.consts + L->surf_const;
        if (mycode == GMLRT_SURF_CONST) {
#pragma unroll
          for (int q = 0; q < 6; q++) s6[q] = vm_as_f64(kc[q]);
        } else {
          // run each distinct program once for the lanes that share it
          bool pending = true;
          while (pending) {
            const uint32_t cur = __builtin_amdgcn_readfirstlane(mycode);
            if (mycode == cur) {
              surface_vm(A.code + cur, kc, (int64_t)face, u, v, s6);
              pending = false;
            }
          }
        }
        const V3 sc = {s6[0], s6[1], s6[2]};
        const double kd = s6[3], ks = s6[4], sn = s6[5];
        const V3 c = vmul(vcmul({A.amb[0], A.amb[1], A.amb[2]}, sc), kd);  // get_ambient
        const double* m = L->m;
        const V3 pos = {m[0] * opos.x + m[1] * opos.y + m[2] * opos.z + m[3],
                        m[4] * opos.x + m[5] * opos.y + m[6] * opos.z + m[7],
                        m[8] * opos.x + m[9] * opos.y + m[10] * opos.z + m[11]};
        const V3 no = prim_normal(type, face, opos);
        const double* im = L->inv_m;  // transform_normal: transpose(inv_m) * n  (transform.rs:34-36)
        V3 nw = {im[0] * no.x + im[4] * no.y + im[8] * no.z, im[1] * no.x + im[5] * no.y + im[9] * no.z,
                 im[2] * no.x + im[6] * no.y + im[10] * no.z};
        V3 normal = vnormalize(nw);
        if (((h.meta & kCurExit) != 0) != ((h.meta & kOrigExit) != 0)) normal = vneg(normal);
        ctx_set3(ctx, F_POS, pos);
        ctx_set3(ctx, F_NRM, normal);
        ctx_set3(ctx, F_SC, sc);
        CTX(F_KD) = kd; CTX(F_KS) = ks; CTX(F_N) = sn;
        ctx_set3(ctx, F_C, c);
        ctx_set3(ctx, F_DIFF, {0.0, 0.0, 0.0});
        ctx_set3(ctx, F_SPEC, {0.0, 0.0, 0.0});
        ctx_set3(ctx, F_RPOS, o);
        ctx_set3(ctx, F_RDIR, d);
        li = 0;
        advance = true;
        n_hits++;
      }
    } else if (kind == kShadow) {
      // ---- raytracer.rs:76-93: is light `li` visible along the shadow ray just traced
      const gmlrt_light& l = A.lights[li];
      const V3 pos = ctx_get3(ctx, F_POS);
      V3 ldir; bool has_ld; double ld;
      light_direction(l, pos, ldir, has_ld, ld);
      bool lit = !h.any;
      if (!lit && has_ld && ld < h.dist) lit = true;
      if (lit) {
        const V3 normal = ctx_get3(ctx, F_NRM);
        const V3 sc = ctx_get3(ctx, F_SC);
        const double kd = CTX(F_KD), ks = CTX(F_KS);
        const double df = vdot(normal, ldir);
        const V3 ic = vcmul(sc, light_intensity(l, pos));
        if (kd > 0.0) ctx_set3(ctx, F_DIFF, vadd(ctx_get3(ctx, F_DIFF), vmul(ic, df)));
        if (ks > 0.0) {
          // get_specular, raytracer.rs:34-42
          const V3 raypos = ctx_get3(ctx, F_RPOS);
          const V3 halfway = vnormalize(vadd(ldir, vnormalize(vsub(raypos, pos))));
          const double sp = vdot(normal, halfway);
          if (sp > 0.0) ctx_set3(ctx, F_SPEC, vadd(ctx_get3(ctx, F_SPEC), vmul(ic, pow(sp, CTX(F_N)))));
          else ctx_set3(ctx, F_SPEC, vadd(ctx_get3(ctx, F_SPEC), {0.0, 0.0, 0.0}));
        }
      }
      li++;
      advance = true;
    }

    if (advance) {
      // ---- raytracer.rs:69-74: next light that faces the surface gets a shadow ray
      const V3 pos = ctx_get3(ctx, F_POS);
      const V3 normal = ctx_get3(ctx, F_NRM);
      bool spawned = false;
      while (li < A.n_lights) {
        V3 ldir; bool has_ld; double ld;
        light_direction(A.lights[li], pos, ldir, has_ld, ld);
        if (vdot(normal, ldir) > 0.0) {
          o = vadd(pos, vmul(ldir, 1e-7));
          d = ldir;
          kind = kShadow;
          n_shadow++;
          spawned = true;
          break;
        }
        li++;
      }
      if (!spawned) {
        // ---- raytracer.rs:96-107
        const double kd = CTX(F_KD), ks = CTX(F_KS);
        const V3 diff = vmul(ctx_get3(ctx, F_DIFF), kd);
        const V3 spec = vmul(ctx_get3(ctx, F_SPEC), ks);
        const V3 c = vadd(ctx_get3(ctx, F_C), vadd(diff, spec));
        if (ks > 0.0 && (A.depth - level) > 0) {
          const V3 raydir = ctx_get3(ctx, F_RDIR);
          const V3 refl = vnormalize(vsub(raydir, vmul(normal, 2.0 * vdot(raydir, normal))));
          // park (c, sc, ks) until the bounce below returns
          const int f = kCtxFixed + level * kCtxPerLevel;
          ctx_set3(ctx, f, c);
          ctx_set3(ctx, f + 3, ctx_get3(ctx, F_SC));
          CTX(f + 6) = ks;
          o = vadd(pos, vmul(refl, 1e-7));
          d = refl;
          kind = kView;
          level++;
          n_reflect++;
        } else {
          C = c;
          finish = true;
        }
      }
    }

    if (finish) {
      // unwind the parked levels: c + (rc * sc) * ks, raytracer.rs:103-104
      for (int lv = level - 1; lv >= 0; lv--) {
        const int f = kCtxFixed + lv * kCtxPerLevel;
        const V3 c = ctx_get3(ctx, f);
        const V3 sc = ctx_get3(ctx, f + 3);
        const double ks = CTX(f + 6);
        C = vadd(c, vmul(vcmul(C, sc), ks));
      }
      const size_t oidx = (size_t)pix * 3;
      A.rgb8[oidx] = quantise(C.x);
      A.rgb8[oidx + 1] = quantise(C.y);
      A.rgb8[oidx + 2] = quantise(C.z);
      if (A.rgbf) { A.rgbf[oidx] = C.x; A.rgbf[oidx + 1] = C.y; A.rgbf[oidx + 2] = C.z; }
      kind = kIdle;
    }
  }

  // ---- ray counters: one atomic per wave and kind
  for (int off = 32; off > 0; off >>= 1) {
    n_primary += (uint32_t)__shfl_down((int)n_primary, off);
    n_shadow += (uint32_t)__shfl_down((int)n_shadow, off);
    n_reflect += (uint32_t)__shfl_down((int)n_reflect, off);
    n_hits += (uint32_t)__shfl_down((int)n_hits, off);
  }
  if (lane == 0) {
    atomicAdd(&A.ray_counts[0], (unsigned long long)n_primary);
    atomicAdd(&A.ray_counts[1], (unsigned long long)n_shadow);
    atomicAdd(&A.ray_counts[2], (unsigned long long)n_reflect);
    atomicAdd(&A.ray_counts[3], (unsigned long long)n_hits);
  }
}

std::string hip_err(const char* what, hipError_t e) {
  return std::string(what) + ": " + hipGetErrorString(e);
}
#define HIP_TRY(expr)                                   \
  do {                                                  \
    hipError_t e_ = (expr);                             \
    if (e_ != hipSuccess) return hip_err(#expr, e_);    \
  } while (0)

}  // namespace

std::string dev_count(int* n) {
  *n = 0;
  hipError_t e = hipGetDeviceCount(n);
  if (e != hipSuccess) { *n = 0; return hip_err("hipGetDeviceCount", e); }
  return "";
}

std::string dev_alloc(int device, size_t bytes, void** out) {
  HIP_TRY(hipSetDevice(device));
  HIP_TRY(hipMalloc(out, bytes ? bytes : 1));
  return "";
}
void dev_free(int device, void* p) {
  if (!p) return;
  if (hipSetDevice(device) == hipSuccess) (void)hipFree(p);
}
std::string dev_copy_to_host(int device, void* dst, const void* src, size_t bytes) {
  HIP_TRY(hipSetDevice(device));
  HIP_TRY(hipMemcpy(dst, src, bytes, hipMemcpyDeviceToHost));
  return "";
}

std::string dev_scene_create(const void* blob, size_t bytes, int device, DeviceScene* out) {
  HIP_TRY(hipSetDevice(device));
  out->device = device;
  out->blob_bytes = bytes;
  std::memcpy(&out->hdr, blob, sizeof(out->hdr));
  HIP_TRY(hipMalloc(&out->d_blob, bytes));
  HIP_TRY(hipMemcpy(out->d_blob, blob, bytes, hipMemcpyHostToDevice));
  HIP_TRY(hipMalloc(&out->d_counters, 64));
  hipEvent_t a, b;
  HIP_TRY(hipEventCreate(&a));
  HIP_TRY(hipEventCreate(&b));
  out->ev_start = a;
  out->ev_stop = b;
  return "";
}

void dev_scene_destroy(DeviceScene* s) {
  if (!s || s->device < 0) return;
  if (hipSetDevice(s->device) != hipSuccess) return;
  (void)hipFree(s->d_blob);
  (void)hipFree(s->d_ev_dist);
  (void)hipFree(s->d_ev_meta);
  (void)hipFree(s->d_ctx);
  (void)hipFree(s->d_counters);
  if (s->ev_start) (void)hipEventDestroy((hipEvent_t)s->ev_start);
  if (s->ev_stop) (void)hipEventDestroy((hipEvent_t)s->ev_stop);
  *s = DeviceScene{};
}

std::string dev_render(DeviceScene* s, const gmlrt_params& p, void* d_rgb8, void* d_rgbf, void* stream_v,
                       gmlrt_stats* stats) {
  HIP_TRY(hipSetDevice(s->device));
  hipStream_t stream = (hipStream_t)stream_v;
  const gmlrt_blob_header& H = s->hdr;
  const uint8_t* base = (const uint8_t*)s->d_blob;

  KArgs A;
  std::memset(&A, 0, sizeof(A));
  A.leaves = (const gmlrt_leaf*)(base + H.off_leaves);
  A.nodes = (const uint32_t*)(base + H.off_nodes);
  A.lights = (const gmlrt_light*)(base + H.off_lights);
  A.code = (const uint64_t*)(base + H.off_code);
  A.consts = (const uint64_t*)(base + H.off_consts);
  A.n_leaves = H.n_leaves;
  A.n_nodes = H.n_nodes;
  A.n_lights = H.n_lights;
  A.max_stack = H.max_stack;
  for (int i = 0; i < 3; i++) A.amb[i] = p.amb[i];
  // raytracer.rs:131-136
  const double w_world = 2.0 * std::tan(0.5 * (p.fov_deg * (3.14159265358979323846 / 180.0)));
  const double h_world = (double)p.height * w_world / (double)p.width;
  A.c_x = -0.5 * w_world;
  A.c_y = 0.5 * h_world;
  A.pw = w_world / (double)p.width;
  A.depth = (int32_t)p.depth;
  A.w = (int32_t)p.width;
  A.h = (int32_t)p.height;
  const int32_t nparts = p.nparts > 1 ? p.nparts : 1;
  A.tile_rows = p.tile_rows > 0 ? p.tile_rows : 8;
  A.part = nparts > 1 ? p.part : 0;
  A.nparts = nparts;
  const int64_t rows = gmlrt_part_rows(p.height, A.tile_rows, A.part, nparts);
  A.rows_part = (int32_t)rows;
  A.tiles_x = (uint32_t)((p.width + 7) / 8);
  const uint64_t tiles_y = (uint64_t)((rows + 7) / 8);
  const uint64_t n_slots = (uint64_t)A.tiles_x * tiles_y * 64ull;
  if (n_slots >= 0xFFFF0000ull) return "image too large for one launch";
  A.n_slots = (uint32_t)n_slots;
  A.rgb8 = (uint8_t*)d_rgb8;
  A.rgbf = (double*)d_rgbf;

  if (stats) {
    stats->rays_primary = stats->rays_shadow = stats->rays_reflect = stats->hits = 0;
    stats->kernel_ms = 0.0;
    stats->leaf_tests = 0;
  }
  if (rows == 0) return "";

  // ---- launch geometry: a persistent grid sized to what the device keeps resident
  hipDeviceProp_t prop;
  HIP_TRY(hipGetDeviceProperties(&prop, s->device));
  const size_t lds_bytes = (size_t)H.max_stack * kBlock * sizeof(uint16_t);
  if (lds_bytes > 64 * 1024) return "CSG tree nests deeper than the operand stack supports";
  int blocks_per_cu = 0;
  HIP_TRY(hipOccupancyMaxActiveBlocksPerMultiprocessor(&blocks_per_cu, render_kernel, kBlock, lds_bytes));
  if (blocks_per_cu < 1) blocks_per_cu = 1;
  uint64_t grid = (uint64_t)prop.multiProcessorCount * (uint64_t)blocks_per_cu;
  const uint64_t need = (n_slots + kBlock - 1) / kBlock;
  if (grid > need) grid = need;
  if (grid < 1) grid = 1;

  // ---- workspace
  const size_t lanes = (size_t)grid * kBlock;
  const size_t ev_cap = H.max_events < 2 ? 2 : H.max_events;
  const size_t ctx_fields = (size_t)kCtxFixed + (size_t)kCtxPerLevel * (size_t)(p.depth + 1);
  if (lanes > s->lanes_alloc || ev_cap > s->ev_slots_alloc) {
    (void)hipFree(s->d_ev_dist);
    (void)hipFree(s->d_ev_meta);
    s->d_ev_dist = s->d_ev_meta = nullptr;
    const size_t l2 = lanes > s->lanes_alloc ? lanes : s->lanes_alloc;
    const size_t e2 = ev_cap > s->ev_slots_alloc ? ev_cap : s->ev_slots_alloc;
    HIP_TRY(hipMalloc(&s->d_ev_dist, l2 * e2 * sizeof(double)));
    HIP_TRY(hipMalloc(&s->d_ev_meta, l2 * e2 * sizeof(uint32_t)));
    s->ev_slots_alloc = e2;
    if (l2 > s->lanes_alloc) {
      // the context array is laid out per lane too
      (void)hipFree(s->d_ctx);
      s->d_ctx = nullptr;
      s->ctx_fields_alloc = 0;
    }
    s->lanes_alloc = l2;
  }
  if (!s->d_ctx || ctx_fields > s->ctx_fields_alloc) {
    (void)hipFree(s->d_ctx);
    s->d_ctx = nullptr;
    HIP_TRY(hipMalloc(&s->d_ctx, s->lanes_alloc * ctx_fields * sizeof(double)));
    s->ctx_fields_alloc = ctx_fields;
  }
  A.ev_dist = (double*)s->d_ev_dist;
  A.ev_meta = (uint32_t*)s->d_ev_meta;
  A.ev_cap = (uint32_t)s->ev_slots_alloc;
  A.ctx = (double*)s->d_ctx;
  A.ctx_fields = (uint32_t)s->ctx_fields_alloc;
  A.next_slot = (unsigned int*)s->d_counters;
  A.ray_counts = (unsigned long long*)((uint8_t*)s->d_counters + 8);

  HIP_TRY(hipMemsetAsync(s->d_counters, 0, 64, stream));
  HIP_TRY(hipEventRecord((hipEvent_t)s->ev_start, stream));
  hipLaunchKernelGGL(render_kernel, dim3((unsigned)grid), dim3(kBlock), lds_bytes, stream, A);
  HIP_TRY(hipGetLastError());
  HIP_TRY(hipEventRecord((hipEvent_t)s->ev_stop, stream));
  unsigned long long counts[4] = {0, 0, 0, 0};
  HIP_TRY(hipMemcpyAsync(counts, (uint8_t*)s->d_counters + 8, 32, hipMemcpyDeviceToHost, stream));
  HIP_TRY(hipStreamSynchronize(stream));
  float ms = 0.f;
  HIP_TRY(hipEventElapsedTime(&ms, (hipEvent_t)s->ev_start, (hipEvent_t)s->ev_stop));
  if (stats) {
    stats->rays_primary = counts[0];
    stats->rays_shadow = counts[1];
    stats->rays_reflect = counts[2];
    stats->hits = counts[3];
    stats->kernel_ms = (double)ms;
    stats->leaf_tests = (counts[0] + counts[1] + counts[2]) * (uint64_t)H.n_leaves;
  }
  return "";
}

}  // namespace gmlrt
