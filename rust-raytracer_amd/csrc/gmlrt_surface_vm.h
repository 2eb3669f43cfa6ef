// Interpreter of the surface bytecode declared in include/gmlrt.h, written once
// and compiled for both sides: the device kernel runs it per hit, the host runs
// it in unit tests to check the surface compiler against the GML evaluator.
// Scalar semantics follow /root/reference/src/evaluator.rs (line numbers per op
// in include/gmlrt.h).
#pragma once
#include <math.h>
#include <stdint.h>
#include <string.h>

#include "../../include/gmlrt.h"

#ifdef __HIPCC__
#define GMLRT_HD __host__ __device__ inline
#else
#define GMLRT_HD inline
#endif

#define GMLRT_SURF_MAX_REGS 64

namespace gmlrt {

GMLRT_HD double vm_as_f64(uint64_t b) { double d; memcpy(&d, &b, 8); return d; }
GMLRT_HD uint64_t vm_from_f64(double d) { uint64_t b; memcpy(&b, &d, 8); return b; }

// `f.floor() as i64` (evaluator.rs:388-391): saturating, NaN -> 0
GMLRT_HD int64_t vm_floor_i64(double x) {
  double f = floor(x);
  if (!(f == f)) return 0;
  if (f >= 9223372036854775808.0) return INT64_MAX;
  if (f <= -9223372036854775808.0) return INT64_MIN;
  return (int64_t)f;
}
// evaluator.rs:212-218.  The reference panics on a zero divisor (and on
// i64::MIN / -1); the bytecode defines both as 0 so that a SELECT may discard them.
GMLRT_HD bool vm_fits_i32(int64_t x) { return (uint64_t)(x + 0x7FFFFFFFLL) < 0xFFFFFFFFULL; }  // |x| < 2^31
GMLRT_HD int64_t vm_divi(int64_t a, int64_t b) {
  if (b == 0 || (a == INT64_MIN && b == -1)) return 0;
  // Operands are nearly always small (floor() of a texture coordinate, a literal): a
  // 32-bit division gives the same quotient for a fraction of the 64-bit one's cost on the device.
  if (vm_fits_i32(a) && vm_fits_i32(b)) return (int64_t)((int32_t)a / (int32_t)b);
  return a / b;
}
GMLRT_HD int64_t vm_modi(int64_t a, int64_t b) {
  if (b != 0 && vm_fits_i32(a) && vm_fits_i32(b)) return (int64_t)((int32_t)a % (int32_t)b);
  return (int64_t)((uint64_t)a - (uint64_t)b * (uint64_t)vm_divi(a, b));
}
GMLRT_HD double vm_to_radians(double d) { return d * (3.14159265358979323846 / 180.0); }
GMLRT_HD double vm_to_degrees(double r) { return r * (180.0 / 3.14159265358979323846); }
GMLRT_HD double vm_sin(double x) { double r = sin(vm_to_radians(x)); return fabs(r) < 1e-15 ? 0.0 : r; }
GMLRT_HD double vm_cos(double x) { double r = cos(vm_to_radians(x)); return fabs(r) < 1e-15 ? 0.0 : r; }
GMLRT_HD double vm_frac(double x) { return x - trunc(x); }
GMLRT_HD double vm_clampf(double x) { return fmin(fmax(x, 0.0), 1.0); }

// Register storage of the interpreter: registers [0, n_fast) live at fast[i * fast_stride]
// (the device hands in a lane-strided slice of LDS, the host a plain array), the
// rest at slow[i * slow_stride] (device: the lane's context record in memory).
// The pointer types are template parameters so that the device can pass pointers
// that carry their address space (LDS, global, constant) in the type.
template <class FastP, class SlowP>
struct VmRegs {
  FastP fast;
  uint32_t fast_stride, n_fast;
  SlowP slow;
  uint32_t slow_stride;
  bool writable;  // false: this lane only runs along with its wave and must not touch the storage
  // fast registers gap_at and up sit gap_len slots further on (the device keeps the slots in between
  // for the program's results); n_fast counts registers, not slots
  uint32_t gap_at, gap_len;
};
template <class RegsT>
GMLRT_HD uint32_t vm_slot(const RegsT& R, uint32_t i) { return i >= R.gap_at ? i + R.gap_len : i; }
// kAllFast: the caller guarantees that every register the program names is below n_fast.
template <bool kAllFast, class RegsT>
GMLRT_HD uint64_t vm_get(const RegsT& R, uint32_t i) {
  if (kAllFast) return R.fast[(size_t)vm_slot(R, i) * R.fast_stride];
  if (i < R.n_fast) return R.fast[(size_t)vm_slot(R, i) * R.fast_stride];
  return R.slow[(size_t)i * R.slow_stride];
}
template <bool kAllFast, class RegsT>
GMLRT_HD void vm_set(const RegsT& R, uint32_t i, uint64_t v) {
  if (!R.writable) return;
  if (kAllFast || i < R.n_fast) R.fast[(size_t)vm_slot(R, i) * R.fast_stride] = v;
  else R.slow[(size_t)i * R.slow_stride] = v;
}
template <bool kAllFast, class ConstP, class RegsT>
GMLRT_HD uint64_t vm_operand(ConstP consts, const RegsT& R, uint32_t x) {
  if (x & 0x8000u) return consts[x & 0x7FFFu];
  return vm_get<kAllFast>(R, x & (GMLRT_SURF_MAX_REGS - 1));
}

// Runs `code` until GMLRT_SOP_END.  consts points at this surface's first constant.
template <bool kAllFast, class CodeP, class ConstP, class RegsT>
GMLRT_HD void surface_vm(CodeP code, ConstP consts, int64_t face, double u, double v, double out[6], const RegsT R,
                         bool* trapped = nullptr) {
  vm_set<kAllFast>(R, 0, (uint64_t)face);
  vm_set<kAllFast>(R, 1, vm_from_f64(u));
  vm_set<kAllFast>(R, 2, vm_from_f64(v));
#define OPND(x) vm_operand<kAllFast>(consts, R, (x))
  for (uint32_t pc = 0;; pc++) {
    const uint64_t w = code[pc];
    const uint32_t op = (uint32_t)(w & 0xFF);
    if (op == GMLRT_SOP_END) break;
    const uint32_t dst = (uint32_t)((w >> 8) & 0xFF);
    const uint32_t oa = (uint32_t)((w >> 16) & 0xFFFF), ob = (uint32_t)((w >> 32) & 0xFFFF),
                   oc = (uint32_t)((w >> 48) & 0xFFFF);
    const uint64_t a = OPND(oa), b = OPND(ob);
    const double fa = vm_as_f64(a), fb = vm_as_f64(b);
    const int64_t ia = (int64_t)a, ib = (int64_t)b;
    uint64_t res = 0;
    switch (op) {
      case GMLRT_SOP_MOV: res = a; break;
      case GMLRT_SOP_ADDI: res = a + b; break;
      case GMLRT_SOP_SUBI: res = a - b; break;
      case GMLRT_SOP_MULI: res = a * b; break;
      case GMLRT_SOP_DIVI: res = (uint64_t)vm_divi(ia, ib); break;
      case GMLRT_SOP_MODI: res = (uint64_t)vm_modi(ia, ib); break;
      case GMLRT_SOP_EQI: res = ia == ib; break;
      case GMLRT_SOP_LESSI: res = ia < ib; break;
      case GMLRT_SOP_REAL: res = vm_from_f64((double)ia); break;
      case GMLRT_SOP_FLOOR: res = (uint64_t)vm_floor_i64(fa); break;
      case GMLRT_SOP_ADDF: res = vm_from_f64(fa + fb); break;
      case GMLRT_SOP_SUBF: res = vm_from_f64(fa - fb); break;
      case GMLRT_SOP_MULF: res = vm_from_f64(fa * fb); break;
      case GMLRT_SOP_DIVF: res = vm_from_f64(fa / fb); break;
      case GMLRT_SOP_EQF: res = fa == fb; break;
      case GMLRT_SOP_LESSF: res = fa < fb; break;
      case GMLRT_SOP_SQRT: res = vm_from_f64(sqrt(fa)); break;
      case GMLRT_SOP_FRAC: res = vm_from_f64(vm_frac(fa)); break;
      case GMLRT_SOP_CLAMPF: res = vm_from_f64(vm_clampf(fa)); break;
      case GMLRT_SOP_SIN: res = vm_from_f64(vm_sin(fa)); break;
      case GMLRT_SOP_COS: res = vm_from_f64(vm_cos(fa)); break;
      case GMLRT_SOP_ASIN: res = vm_from_f64(vm_to_degrees(asin(fa))); break;
      case GMLRT_SOP_ACOS: res = vm_from_f64(vm_to_degrees(acos(fa))); break;
      case GMLRT_SOP_SELECT: res = a ? b : OPND(oc); break;
      case GMLRT_SOP_OUT: {
        const double fc = vm_as_f64(OPND(oc));
        if (dst == 0) { out[0] = fa; out[1] = fb; out[2] = fc; }
        else if (dst == 3) { out[3] = fa; out[4] = fb; out[5] = fc; }
        continue;
      }
      case GMLRT_SOP_TRAP:
        if (a && trapped) *trapped = true;
        continue;
      default: break;
    }
    vm_set<kAllFast>(R, dst & (GMLRT_SURF_MAX_REGS - 1), res);
  }
#undef OPND
}

// ---- expanded form.  What the device interprets is not the blob's word code but an
// expansion of it, one 32-byte record per instruction with the constants inline, made
// once per distinct (code, constants) pair when a scene is uploaded: one wave-uniform
// load brings everything an instruction needs, and the record after it is requested
// before the current one executes.
struct SurfInstr {
  uint32_t head;  // op in bits 7..0, dst in 15..8, bit 16/17/18: a/b/c is an immediate (else a register index)
  uint32_t pad;
  uint64_t a, b, c;
};
#define GMLRT_SI_IMM_A 0x10000u
#define GMLRT_SI_IMM_B 0x20000u
#define GMLRT_SI_IMM_C 0x40000u

// field by field: a struct cannot be copied out of an address-space-qualified pointer
template <class ProgP>
GMLRT_HD SurfInstr vm_fetch(ProgP prog, uint32_t pc) {
  SurfInstr I;
  I.head = prog[pc].head;
  I.pad = 0;
  I.a = prog[pc].a;
  I.b = prog[pc].b;
  I.c = prog[pc].c;
  return I;
}

// Where a program's results go: put(first, a, b, c) stores result slots first .. first + 2 (first is 0
// for the colour, 3 for kd, ks, n).  The host and the plain interpreter collect them in an array; the device
// stores them where the shading code reads them, so that no result is carried through the interpreter loop.
struct VmArrayOut {
  double* out;
  bool* trapped = nullptr;
  GMLRT_HD void put(uint32_t first, double a, double b, double c) const { out[first] = a; out[first + 1] = b; out[first + 2] = c; }
  GMLRT_HD void trap() const { if (trapped) *trapped = true; }
};

template <bool kAllFast, class ProgP, class RegsT, class OutT>
GMLRT_HD void surface_stream_vm(ProgP prog, int64_t face, double u, double v, const OutT out, const RegsT R) {
  vm_set<kAllFast>(R, 0, (uint64_t)face);
  vm_set<kAllFast>(R, 1, vm_from_f64(u));
  vm_set<kAllFast>(R, 2, vm_from_f64(v));
  SurfInstr I = vm_fetch(prog, 0);
  for (uint32_t pc = 1;; pc++) {
    const uint32_t head = I.head;
    const uint32_t op = head & 0xFFu;
    if (op == GMLRT_SOP_END) break;
    const SurfInstr next = vm_fetch(prog, pc);  // always exists: the program ends with an END record
    const uint32_t dst = (head >> 8) & 0xFFu;
    const uint64_t a = (head & GMLRT_SI_IMM_A) ? I.a : vm_get<kAllFast>(R, (uint32_t)I.a);
    const uint64_t b = (head & GMLRT_SI_IMM_B) ? I.b : vm_get<kAllFast>(R, (uint32_t)I.b);
    const double fa = vm_as_f64(a), fb = vm_as_f64(b);
    const int64_t ia = (int64_t)a, ib = (int64_t)b;
    uint64_t res = 0;
    switch (op) {
      case GMLRT_SOP_MOV: res = a; break;
      case GMLRT_SOP_ADDI: res = a + b; break;
      case GMLRT_SOP_SUBI: res = a - b; break;
      case GMLRT_SOP_MULI: res = a * b; break;
      case GMLRT_SOP_DIVI: res = (uint64_t)vm_divi(ia, ib); break;
      case GMLRT_SOP_MODI: res = (uint64_t)vm_modi(ia, ib); break;
      case GMLRT_SOP_EQI: res = ia == ib; break;
      case GMLRT_SOP_LESSI: res = ia < ib; break;
      case GMLRT_SOP_REAL: res = vm_from_f64((double)ia); break;
      case GMLRT_SOP_FLOOR: res = (uint64_t)vm_floor_i64(fa); break;
      case GMLRT_SOP_ADDF: res = vm_from_f64(fa + fb); break;
      case GMLRT_SOP_SUBF: res = vm_from_f64(fa - fb); break;
      case GMLRT_SOP_MULF: res = vm_from_f64(fa * fb); break;
      case GMLRT_SOP_DIVF: res = vm_from_f64(fa / fb); break;
      case GMLRT_SOP_EQF: res = fa == fb; break;
      case GMLRT_SOP_LESSF: res = fa < fb; break;
      case GMLRT_SOP_SQRT: res = vm_from_f64(sqrt(fa)); break;
      case GMLRT_SOP_FRAC: res = vm_from_f64(vm_frac(fa)); break;
      case GMLRT_SOP_CLAMPF: res = vm_from_f64(vm_clampf(fa)); break;
      case GMLRT_SOP_SIN: res = vm_from_f64(vm_sin(fa)); break;
      case GMLRT_SOP_COS: res = vm_from_f64(vm_cos(fa)); break;
      case GMLRT_SOP_ASIN: res = vm_from_f64(vm_to_degrees(asin(fa))); break;
      case GMLRT_SOP_ACOS: res = vm_from_f64(vm_to_degrees(acos(fa))); break;
      case GMLRT_SOP_SELECT: {
        const uint64_t c = (head & GMLRT_SI_IMM_C) ? I.c : vm_get<kAllFast>(R, (uint32_t)I.c);
        res = a ? b : c;
        break;
      }
      case GMLRT_SOP_OUT: {
        const uint64_t c = (head & GMLRT_SI_IMM_C) ? I.c : vm_get<kAllFast>(R, (uint32_t)I.c);
        const double fc = vm_as_f64(c);
        if (dst == 0 || dst == 3) out.put(dst, fa, fb, fc);
        I = next;
        continue;
      }
      case GMLRT_SOP_TRAP:
        if (a) out.trap();
        I = next;
        continue;
      default: break;
    }
    vm_set<kAllFast>(R, dst & (GMLRT_SURF_MAX_REGS - 1), res);
    I = next;
  }
}

}  // namespace gmlrt
