// Host check of the look before the leaf test (gmlrt_prims.hpp: prim_may_enter): for random and for degenerate rays
// (lattice coordinates, zero direction components, origins 1e-7 off a surface, huge and tiny object-space scales) the
// intersector must report no Entry whenever the look says none is possible.  The hardware's approximate reciprocal and
// square root are stood in for by exact ones with a relative error of up to 2^-22 of either sign.
// Usage: look_fuzz <cases per primitive and mode> <seed>; prints one line per primitive, exit code 1 on a violation.
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <random>
#define __device__
static double g_pert = 1.0;
static inline double fuzz_rcp(double x) { return (1.0 / x) * g_pert; }
static inline double fuzz_sqrt(double x) { return std::sqrt(x) * g_pert; }
#define __builtin_amdgcn_rcp(x) fuzz_rcp(x)
#define __builtin_amdgcn_sqrt(x) fuzz_sqrt(x)
#define GMLRT_PRIMS_HOST_CHECK 1
#include "../../rust-raytracer_amd/csrc/gmlrt_prims.hpp"
using namespace gmlrt;

int main(int argc, char** argv) {
  const long n = argc > 1 ? std::atol(argv[1]) : 200000;
  std::mt19937_64 rng(argc > 2 ? std::atoi(argv[2]) : 1);
  std::uniform_real_distribution<double> U(-1.0, 1.0);
  const double lat[] = {-2.0, -1.0, -0.5, 0.0, 0.5, 1.0, 2.0, 1e-7, -1e-7, 1.0 + 1e-7, 1.0 - 1e-7, 0.5 + 1e-7, 1e-15, 1e-16, -1e-16, 3.0, 0.25};
  const double dl[] = {0.0, 0.0, 1.0, -1.0, 0.5, -0.5, 1e-8, -1e-8, 1e-15, 1e-16, 2.0, -0.0, 1e-7, -1e-7, 0.70710678118654752};
  const char* names[5] = {"sphere", "cube", "cylinder", "cone", "plane"};
  long bad_total = 0;
  for (uint32_t type = 0; type < 5; type++) {
    long bad = 0, tests = 0, noentry = 0, caught = 0;
    for (int mode = 0; mode < 5; mode++) {
      for (long i = 0; i < n; i++) {
        V3 p, d;
        auto L = [&]() { return lat[rng() % (sizeof(lat) / sizeof(lat[0]))]; };
        auto D = [&]() { return dl[rng() % (sizeof(dl) / sizeof(dl[0]))]; };
        if (mode == 0) {
          const double ps = std::pow(10.0, 3.0 * U(rng)), ds = std::pow(10.0, 2.0 * U(rng));
          p = {3.0 * U(rng) * ps, 3.0 * U(rng) * ps, 3.0 * U(rng) * ps};
          d = {U(rng) * ds, U(rng) * ds, U(rng) * ds};
        } else if (mode == 1) {
          p = {L(), L(), L()};
          d = {D(), D(), D()};
          if (rng() % 4 == 0) { const double s = std::pow(10.0, 3.0 * U(rng)); d = {d.x * s, d.y * s, d.z * s}; }
        } else if (mode == 2) {
          // an origin 1e-7 along a random direction off a random point near the primitive, lattice or random direction
          p = {L() + 1e-7 * U(rng), L() + 1e-7 * U(rng), L() + 1e-7 * U(rng)};
          d = rng() % 2 ? V3{D(), D(), D()} : V3{U(rng), U(rng), U(rng)};
        } else if (mode == 4) {
          // grazing: from a point on (or 1e-7 off) the mantle of the sphere / cylinder / cone, along the mantle or nearly so
          const double ang = 3.14159265358979 * U(rng), h = 0.5 + 0.5 * U(rng);
          const double rr = type == GMLRT_CONE ? h : 1.0;
          V3 q = {rr * std::cos(ang), type == GMLRT_SPHERE ? 0.0 : h, rr * std::sin(ang)};
          V3 tdir = type == GMLRT_CONE ? V3{std::cos(ang), 1.0, std::sin(ang)} : V3{-std::sin(ang), type == GMLRT_SPHERE ? U(rng) : 0.0, std::cos(ang)};
          if (rng() % 2) tdir = {tdir.x + 1e-8 * U(rng), tdir.y + 1e-8 * U(rng), tdir.z + 1e-8 * U(rng)};
          if (rng() % 3 == 0) tdir = {tdir.x + 1e-3 * U(rng), tdir.y + 1e-3 * U(rng), tdir.z + 1e-3 * U(rng)};
          const double back = rng() % 2 ? std::pow(10.0, 2.0 * U(rng)) : 0.0, off = rng() % 2 ? 1e-7 * U(rng) : 0.0, ds = std::pow(10.0, 2.0 * U(rng));
          p = {q.x - back * tdir.x + off, q.y - back * tdir.y + off, q.z - back * tdir.z - off};
          if (rng() % 8 == 0 && type == GMLRT_CONE) p = {1e-7 * U(rng), 1e-7 * U(rng), 1e-7 * U(rng)};  // at the apex
          d = {tdir.x * ds, tdir.y * ds, tdir.z * ds};
        } else {
          // aimed at the primitive from outside: a point in the unit box region as target
          const V3 q = {U(rng), 0.5 + 0.5 * U(rng) * 1.2, U(rng)};
          const double r = std::pow(10.0, 1.5 * U(rng) + 0.5);
          const V3 o = {r * U(rng), r * U(rng), r * U(rng)};
          p = o;
          d = {q.x - o.x, q.y - o.y, q.z - o.z};
          if (rng() % 3 == 0) { if (rng() % 2) d.y = 0.0; else { d.x = 0.0; d.z = 0.0; } }
        }
        if (d.x == 0.0 && d.y == 0.0 && d.z == 0.0) continue;
        g_pert = 1.0 + ((rng() & 1) ? 1.0 : -1.0) * std::ldexp(1.0, -22) * (double)(rng() % 1025) / 1024.0;
        const double a = vdot(d, d);
        const double scale = 1.0 / std::sqrt(a);
        const V3 rd = vmul(d, scale);
        const PrimHit ph = isect_prim(type, p, rd);
        const bool may = prim_may_enter(type, p, d, a);
        tests++;
        if (!(ph.bits & 1)) { noentry++; if (!may) caught++; }
        if ((ph.bits & 1) && !may) {
          if (bad < 5)
            std::printf("VIOLATION %s mode %d: p = (%.17g, %.17g, %.17g) d = (%.17g, %.17g, %.17g): Entry at %.17g face %d\n", names[type], mode,
                        p.x, p.y, p.z, d.x, d.y, d.z, ph.tA, ph.fA);
          bad++;
        }
      }
    }
    std::printf("%-8s %ld rays, %ld without an Entry, %ld of those turned away by the look (%.1f %%), %ld violations\n", names[type], tests, noentry,
                caught, 100.0 * (double)caught / (double)(noentry ? noentry : 1), bad);
    bad_total += bad;
  }
  return bad_total ? 1 : 0;
}
