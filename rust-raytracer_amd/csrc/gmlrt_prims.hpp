// Device-side geometry of the GML primitives in f64: ray -> object space, the five
// intersectors, per-face (u,v) and object-space normals.  Each function restates
// the reference function cited above it with the same operation order; the
// translation unit that includes this header is built with -ffp-contract=off.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/gmlrt.h"

namespace gmlrt {

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
// (in two steps, so that a caller can look at the ray before it pays for the square root and the division)
__device__ inline void to_object_raw(const double* __restrict__ im, V3 o, V3 d, V3& td, V3& rp) {
  td = {im[0] * d.x + im[1] * d.y + im[2] * d.z, im[4] * d.x + im[5] * d.y + im[6] * d.z,
        im[8] * d.x + im[9] * d.y + im[10] * d.z};
  rp = {im[0] * o.x + im[1] * o.y + im[2] * o.z + im[3], im[4] * o.x + im[5] * o.y + im[6] * o.z + im[7],
        im[8] * o.x + im[9] * o.y + im[10] * o.z + im[11]};
}
__device__ inline void to_object(const double* __restrict__ im, V3 o, V3 d, double& scale, V3& rd, V3& rp) {
  V3 td;
  to_object_raw(im, o, d, td, rp);
  scale = 1.0 / sqrt(vdot(td, td));
  rd = vmul(td, scale);
}

// Result of one primitive test: bit 0 = an Entry event (tA, fA), bit 1 = an Exit event (tB, fB).
struct PrimHit { int bits; double tA, tB; int fA, fB; };

// primitives.rs:355-393
__device__ inline PrimHit isect_sphere(V3 p, V3 d) {
  PrimHit r = {0, 0.0, 0.0, 0, 0};
  double s = vdot(vneg(p), d);
  double lsq = vdot(p, p);
  if ((s < 0.0) & (lsq > 1.0)) return r;
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
  const bool k1 = (0.0 <= cminy) & (cminy <= 1.0), k2 = (0.0 <= cmaxy) & (cmaxy <= 1.0);
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

// ---------------------------------------------------------------- a look before the test
// Can the ray p + t * d have an Entry event on the unit primitive at all?  d is the direction in object space BEFORE
// it is normalised (a = d . d), so the answer costs neither the square root nor the division of to_object, and none
// of the intersector's.  "false" is a promise: the intersector above, given p and d / |d|, reports no Entry (bit 0
// of PrimHit::bits) for this ray.  "true" promises nothing.
//
// Every condition below is a consequence of the ray having an Entry, in real arithmetic, relaxed by a slack of
// 1e-9 relative to the magnitudes of the terms it compares (1e-5 where the hardware's approximate reciprocal and
// square root are used: they are good to 2^-23).  The intersectors' own rounding errors are of the order of 1e-15 of
// the same magnitudes, so a ray turned away here is turned away by them too.  The arithmetic here need not be
// reproducible, only conservative: fused operations are welcome, and a NaN anywhere makes a comparison false,
// which keeps the ray.
__device__ inline bool prim_may_enter(uint32_t type, V3 p, V3 d, double a) {
  constexpr double kE = 1e-9, kA = 1e-5;
  switch (type) {
    case GMLRT_SPHERE: {
      // Entry <=> the line passes within the radius, the origin lies outside and the sphere lies ahead
      const double S = -__builtin_fma(p.x, d.x, __builtin_fma(p.y, d.y, p.z * d.z));  // s * |d|
      const double lsq = __builtin_fma(p.x, p.x, __builtin_fma(p.y, p.y, p.z * p.z));
      const double la = lsq * a;
      const bool behind = (S < 0.0) & (S * S > 1e-24 * la);
      const bool inside = lsq < 1.0 - kE;
      const bool wide = __builtin_fma(-S, S, la) > __builtin_fma(kE, la + a, a);
      return !(behind | inside | wide);
    }
    case GMLRT_CUBE: {
      // slabs with approximate reciprocals; Entry <=> the three intervals overlap and the overlap begins ahead
      const double rx = __builtin_amdgcn_rcp(d.x), ry = __builtin_amdgcn_rcp(d.y), rz = __builtin_amdgcn_rcp(d.z);
      const double x0 = -p.x * rx, x1 = (1.0 - p.x) * rx, y0 = -p.y * ry, y1 = (1.0 - p.y) * ry, z0 = -p.z * rz, z1 = (1.0 - p.z) * rz;
      const double tmin = fmax(fmax(fmin(x0, x1), fmin(y0, y1)), fmin(z0, z1));
      const double tmax = fmin(fmin(fmax(x0, x1), fmax(y0, y1)), fmax(z0, z1));
      // (1 / |d_i| >= 1 / |d|: the second term is 1e-9 of the cube's edge, in units of t)
      const double m = __builtin_fma(kA, fabs(tmin) + fabs(tmax), kE * fmin(fmin(fabs(rx), fabs(ry)), fabs(rz)));
      // The intersector leaves out a slab whose direction component is below 1e-15 of the direction's length and
      // then takes an origin ON a face of that slab for inside, whatever the ray does from there: a ray with such a
      // component (below 1e-14, to be clear of the threshold) is not judged here.
      const double tiny = 1e-28 * a;
      const bool flat = (d.x * d.x <= tiny) | (d.y * d.y <= tiny) | (d.z * d.z <= tiny);
      return flat | !((tmin - tmax > m) | (tmin < -m));
    }
    case GMLRT_CYLINDER: {
      const double A = __builtin_fma(d.x, d.x, d.z * d.z), Bh = __builtin_fma(p.x, d.x, p.z * d.z);
      const double r2 = __builtin_fma(p.x, p.x, p.z * p.z), C = r2 - 1.0;
      // beyond a cap and not heading back; outside the mantle and heading away from the axis; inside the solid
      const bool beyond = ((p.y > 1.0 + kE) & (d.y >= 0.0)) | ((p.y < -kE) & (d.y <= 0.0));
      const bool away = (C > kE * (r2 + 1.0)) & (Bh > 0.0) & (Bh * Bh > 1e-20 * A * r2);
      const bool inside = (C < -kE) & (p.y > kE) & (p.y < 1.0 - kE);
      // the line misses the infinite cylinder
      // (the slack is taken from the magnitudes of the terms, not of their sums: Bh and C are differences that may cancel)
      const double disc = __builtin_fma(Bh, Bh, -A * C);
      const double m1 = fabs(p.x * d.x) + fabs(p.z * d.z);
      const bool wide = disc < -kE * __builtin_fma(m1, m1, A * (r2 + 1.0));
      // the chord inside the infinite cylinder lies beyond one cap with both ends.  Not asked of a ray within 1e-3 of
      // parallel or orthogonal to the axis: there the intersector leaves its general case and looks at the origin only.
      const double rt = __builtin_amdgcn_sqrt(fmax(disc, 0.0)), rA = __builtin_amdgcn_rcp(A);
      const double ya = __builtin_fma((-Bh - rt) * rA, d.y, p.y), yb = __builtin_fma((-Bh + rt) * rA, d.y, p.y);
      const double m = kA * __builtin_fma((fabs(Bh) + rt) * rA, fabs(d.y), fabs(p.y) + 1.0);
      const bool general = (d.y * d.y > 1e-6 * a) & (A > 1e-6 * a);
      const bool off = general & (((ya > 1.0 + m) & (yb > 1.0 + m)) | ((ya < -m) & (yb < -m)));
      return !(beyond | away | inside | wide | off);
    }
    case GMLRT_CONE: {
      const double A = __builtin_fma(d.x, d.x, __builtin_fma(d.z, d.z, -d.y * d.y));
      const double Bh = __builtin_fma(p.x, d.x, __builtin_fma(p.z, d.z, -p.y * d.y));
      const double r2 = __builtin_fma(p.x, p.x, p.z * p.z), py2 = p.y * p.y, C = r2 - py2;
      // every point of the cone has 0 <= y <= 1: beyond that range and not heading back, all events lie behind
      const bool beyond = ((p.y > 1.0 + kE) & (d.y >= 0.0)) | ((p.y < -kE) & (d.y <= 0.0));
      // (the slack is taken from the magnitudes of the terms, not of their sums: A, Bh and C are differences that may
      // cancel, and where they do, a ray along the mantle, the intersector's own roots are rounding noise)
      const double disc = __builtin_fma(Bh, Bh, -A * C);
      const double m1 = fabs(p.x * d.x) + fabs(p.z * d.z) + fabs(p.y * d.y);
      const bool wide = disc < -kE * __builtin_fma(m1, m1, a * (r2 + py2));
      // both crossings of the double cone lie outside 0 <= y <= 1 (asked only where the quadratic is well conditioned)
      const double rt = __builtin_amdgcn_sqrt(fmax(disc, 0.0)), rA = __builtin_amdgcn_rcp(A);
      const double ya = __builtin_fma((-Bh - rt) * rA, d.y, p.y), yb = __builtin_fma((-Bh + rt) * rA, d.y, p.y);
      const double m = kA * __builtin_fma((fabs(Bh) + rt) * fabs(rA), fabs(d.y), fabs(p.y) + 1.0);
      const bool off = (fabs(A) > 1e-6 * a) & ((ya > 1.0 + m) | (ya < -m)) & ((yb > 1.0 + m) | (yb < -m));
      return !(beyond | wide | off);
    }
    default:
      // plane: an Entry <=> the ray heads into y <= 0 from y >= 0 (exact: the signs survive the normalisation)
      return !(d.y >= 0.0) & !(p.y < 0.0);
  }
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

}  // namespace gmlrt
