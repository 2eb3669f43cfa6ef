// Fragment of gmlrt_kernels.hip (included inside its anonymous namespace): everything
// trace() does around its intersect calls (raytracer.rs:53-110), cut at those calls.
//
// after_traversal consumes the traversal result of the ray a lane holds and leaves
// either the next ray to traverse in the lane state or an idle lane with its pixel
// written.  It runs once per traced ray, out of line, so that the traversal loop
// keeps a small register footprint.  Values that must survive the next traversal
// are parked in the lane's context record in device memory; the code is laid out
// so that each path does one batch of loads from it, not a chain of them.

// Context fields (f64 each, field-major: field f of this lane at ctx[f * 64]).
#define CTX(f) ctx[(size_t)(f) * kWave]
__device__ inline V3 ctx_get3(const double* ctx, int f) { return {CTX(f), CTX(f + 1), CTX(f + 2)}; }
__device__ inline void ctx_set3(double* ctx, int f, V3 v) { CTX(f) = v.x; CTX(f + 1) = v.y; CTX(f + 2) = v.z; }
enum { F_POS = 0, F_NRM = 3, F_SC = 6, F_KD = 9, F_KS = 10, F_N = 11, F_C = 12, F_DIFF = 15, F_SPEC = 18,
       F_RPOS = 21, F_RDIR = 24 };

// The traversal result travels as two scalars (distance; meta with kHitAny set when
// there is a hit): together with the lane state an aggregate would exceed what the
// calling convention passes in registers and go through the stack.
__device__ GMLRT_AFTER_INLINE LaneState after_traversal(LaneState S, const double h_dist, const uint32_t h_meta,
                                                  const ShadeArgs* __restrict__ SA_v, double* __restrict__ ctx_v,
                                                  uint64_t* __restrict__ vm_fast_v) {
  const ShadeArgs* __restrict__ SA = const_space(wave_uniform(SA_v));
  double* __restrict__ ctx = global_space(ctx_v);
  const ScenePtrs P = {const_space(wave_uniform(SA->leaves)), nullptr, const_space(wave_uniform(SA->lights)),
                       const_space(wave_uniform(SA->code)), const_space(wave_uniform(SA->consts))};
  const Hit h = {(h_meta & kHitAny) != 0, h_dist, h_meta & ~kHitAny, false};
  int kind = st_kind(S.st);
  int level = (int)((S.st >> 2) & 0x7FFu);
  uint32_t li = (S.st >> 13) & 0x7FFu;
  uint32_t events = 0;
  const uint32_t n_lights = SA->n_lights;

  bool advance = false, finish = false;
  V3 C = {0.0, 0.0, 0.0};  // value of trace() at `level` when finish is set
  // the shading point: filled by either path below when `advance` is set
  V3 pos = {0, 0, 0}, normal = {0, 0, 0}, sc = {0, 0, 0}, diff = {0, 0, 0}, spec = {0, 0, 0};
  double kd = 0.0, ks = 0.0;
  bool fresh = false;       // the point was shaded in this call: its ambient term and view ray are in registers
  V3 c_amb = {0, 0, 0};

  DIAG(unsigned long long tq0 = __builtin_readcyclecounter(), tq1, tq2, tq3, tq4, tq5, tq6;)
  // ---- raytracer.rs:58-62: geometry of a view ray's hit in the primitive's object space
  bool shade = false;
  const gmlrt_leaf* L = P.leaves;
  uint32_t type = 0, mycode = GMLRT_SURF_CONST, myconst = 0;
  int face = 0;
  double u = 0.0, v = 0.0;
  V3 opos = {0, 0, 0};
  if (kind == kView) {
    if (!h.any || (h.meta & kCurExit)) {
      finish = true;  // raytracer.rs:55-57, 108-110
    } else {
      shade = true;
      L = P.leaves + (h.meta & kLeafMask);
      face = (int)((h.meta >> kFaceShift) & 7u);
      type = L->type;
      mycode = L->surf_code;
      myconst = L->surf_const;
      double scale; V3 rd, rp;
      to_object(L->inv_m, S.o, S.d, scale, rd, rp);
      PrimHit ph = isect_prim(type, rp, rd);
      const double od = (h.meta & kOrigExit) ? ph.tB : ph.tA;
      opos = vadd(rp, vmul(rd, od));
      prim_uv(type, face, opos, u, v);
    }
  }

  // ---- raytracer.rs:63: the surface function.  Each distinct (program, constants) pair among
  // the wave's hits is interpreted once, by all lanes in uniform control flow (so the
  // bytecode is fetched with scalar loads); a lane keeps the result of its own pair.
  // Lanes without a hit run along on zeros; the bytecode has no trapping operation.
  double s6[6] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
  DIAG(tq1 = __builtin_readcyclecounter();)
  {
    typedef GMLRT_AS(3) uint64_t* LdsP;
    typedef GMLRT_AS(1) uint64_t* GlobalP;
    typedef const GMLRT_AS(4) uint64_t* ConstP;
    const VmRegs<LdsP, GlobalP> R = {
        (LdsP)(uintptr_t)vm_fast_v, (uint32_t)kBlock, SA->n_vm_fast,
        (GlobalP)(uintptr_t)(ctx_v + (size_t)(SA->ctx_fields - GMLRT_SURF_MAX_REGS) * kWave), (uint32_t)kWave};
    const ConstP code = (ConstP)(uintptr_t)P.code, consts = (ConstP)(uintptr_t)P.consts;
    const bool all_fast = SA->vm_all_fast != 0;
    bool need = shade && mycode != GMLRT_SURF_CONST;
    unsigned long long todo = __ballot(need);
    while (todo) {
      const int src = __builtin_ctzll(todo);
      const uint32_t cur = (uint32_t)__builtin_amdgcn_readlane((int)mycode, src);
      const uint32_t curk = (uint32_t)__builtin_amdgcn_readlane((int)myconst, src);
      double t6[6] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
      if (all_fast) surface_vm<true>(code + cur, consts + curk, (int64_t)face, u, v, t6, R);
      else surface_vm<false>(code + cur, consts + curk, (int64_t)face, u, v, t6, R);
      if (need && mycode == cur && myconst == curk) {
#pragma unroll
        for (int q = 0; q < 6; q++) s6[q] = t6[q];
        need = false;
      }
      todo = __ballot(need);
    }
  }

  DIAG(tq2 = __builtin_readcyclecounter();)
  if (shade) {
    {
      if (mycode == GMLRT_SURF_CONST) {
        const uint64_t* kc = P.consts + myconst;
#pragma unroll
        for (int q = 0; q < 6; q++) s6[q] = vm_as_f64(kc[q]);
      }
      sc = {s6[0], s6[1], s6[2]};
      kd = s6[3];
      ks = s6[4];
      c_amb = vmul(vcmul({SA->amb[0], SA->amb[1], SA->amb[2]}, sc), kd);  // get_ambient
      // ---- raytracer.rs:64-68: position and normal in world space
      const double* m = L->m;
      pos = {m[0] * opos.x + m[1] * opos.y + m[2] * opos.z + m[3], m[4] * opos.x + m[5] * opos.y + m[6] * opos.z + m[7],
             m[8] * opos.x + m[9] * opos.y + m[10] * opos.z + m[11]};
      const V3 no = prim_normal(type, face, opos);
      const double* im = L->inv_m;  // transform_normal: transpose(inv_m) * n  (transform.rs:34-36)
      V3 nw = {im[0] * no.x + im[4] * no.y + im[8] * no.z, im[1] * no.x + im[5] * no.y + im[9] * no.z,
               im[2] * no.x + im[6] * no.y + im[10] * no.z};
      normal = vnormalize(nw);
      if (((h.meta & kCurExit) != 0) != ((h.meta & kOrigExit) != 0)) normal = vneg(normal);
      ctx_set3(ctx, F_POS, pos);
      ctx_set3(ctx, F_NRM, normal);
      ctx_set3(ctx, F_SC, sc);
      CTX(F_KD) = kd; CTX(F_KS) = ks; CTX(F_N) = s6[5];
      ctx_set3(ctx, F_C, c_amb);
      ctx_set3(ctx, F_RPOS, S.o);
      ctx_set3(ctx, F_RDIR, S.d);
      fresh = true;
      li = 0;
      events |= kStHit;
      advance = true;
    }
  }
  DIAG(tq3 = __builtin_readcyclecounter();)
  if (kind == kShadow) {
    // ---- raytracer.rs:76-93: is light `li` visible along the shadow ray just traced
    const gmlrt_light& l = P.lights[li];
    pos = ctx_get3(ctx, F_POS);
    normal = ctx_get3(ctx, F_NRM);
    sc = ctx_get3(ctx, F_SC);
    kd = CTX(F_KD);
    ks = CTX(F_KS);
    const double sn = CTX(F_N);
    diff = ctx_get3(ctx, F_DIFF);
    spec = ctx_get3(ctx, F_SPEC);
    const V3 raypos = ctx_get3(ctx, F_RPOS);
    V3 ldir; bool has_ld; double ld;
    light_direction(l, pos, ldir, has_ld, ld);
    bool lit = !h.any;
    if (!lit && has_ld && ld < h.dist) lit = true;
    if (lit) {
      const double df = vdot(normal, ldir);
      const V3 ic = vcmul(sc, light_intensity(l, pos));
      if (kd > 0.0) diff = vadd(diff, vmul(ic, df));
      if (ks > 0.0) {
        // get_specular, raytracer.rs:34-42
        const V3 halfway = vnormalize(vadd(ldir, vnormalize(vsub(raypos, pos))));
        const double sp = vdot(normal, halfway);
        if (sp > 0.0) spec = vadd(spec, vmul(ic, pow(sp, sn)));
        else spec = vadd(spec, {0.0, 0.0, 0.0});
      }
    }
    li++;
    advance = true;
  }

  DIAG(tq4 = __builtin_readcyclecounter();)
  if (advance) {
    // ---- raytracer.rs:69-74: next light that faces the surface gets a shadow ray
    bool spawned = false;
    while (li < n_lights) {
      V3 ldir; bool has_ld; double ld;
      light_direction(P.lights[li], pos, ldir, has_ld, ld);
      if (vdot(normal, ldir) > 0.0) {
        // the accumulators wait in memory while the shadow ray is traced
        ctx_set3(ctx, F_DIFF, diff);
        ctx_set3(ctx, F_SPEC, spec);
        S.o = vadd(pos, vmul(ldir, 1e-7));
        S.d = ldir;
        kind = kShadow;
        events |= kStShadow;
        spawned = true;
        break;
      }
      li++;
    }
    if (!spawned) {
      // ---- raytracer.rs:96-107
      const V3 c0 = fresh ? c_amb : ctx_get3(ctx, F_C);
      const V3 c = vadd(c0, vadd(vmul(diff, kd), vmul(spec, ks)));
      if (ks > 0.0 && (SA->depth - level) > 0) {
        const V3 raydir = fresh ? S.d : ctx_get3(ctx, F_RDIR);
        const V3 refl = vnormalize(vsub(raydir, vmul(normal, 2.0 * vdot(raydir, normal))));
        // park (c, sc, ks) until the bounce below returns
        const int f = kCtxFixed + level * kCtxPerLevel;
        ctx_set3(ctx, f, c);
        ctx_set3(ctx, f + 3, sc);
        CTX(f + 6) = ks;
        S.o = vadd(pos, vmul(refl, 1e-7));
        S.d = refl;
        kind = kView;
        level++;
        events |= kStReflect;
      } else {
        C = c;
        finish = true;
      }
    }
  }

  DIAG(tq5 = __builtin_readcyclecounter();)
  if (finish) {
    // unwind the parked levels: c + (rc * sc) * ks, raytracer.rs:103-104
    for (int lv = level - 1; lv >= 0; lv--) {
      const int f = kCtxFixed + lv * kCtxPerLevel;
      const V3 c = ctx_get3(ctx, f);
      const V3 psc = ctx_get3(ctx, f + 3);
      const double pks = CTX(f + 6);
      C = vadd(c, vmul(vcmul(C, psc), pks));
    }
    const size_t oidx = (size_t)S.pix * 3;
    uint8_t* rgb8 = global_space(wave_uniform(SA->rgb8));
    double* rgbf = global_space(wave_uniform(SA->rgbf));
    rgb8[oidx] = quantise(C.x);
    rgb8[oidx + 1] = quantise(C.y);
    rgb8[oidx + 2] = quantise(C.z);
    if (rgbf) { rgbf[oidx] = C.x; rgbf[oidx + 1] = C.y; rgbf[oidx + 2] = C.z; }
    kind = kIdle;
  }
  DIAG(tq6 = __builtin_readcyclecounter();
       const uint32_t lid = __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
       if (lid == (uint32_t)__builtin_amdgcn_readfirstlane((int)lid)) {
         // the wave's own row of the per-wave table: plain adds, no contention
         unsigned long long* dq = global_space(wave_uniform(SA->diag_waves)) +
                                  ((size_t)blockIdx.x * (kBlock / kWave) + threadIdx.x / kWave) * 8;
         dq[0] += tq1 - tq0; dq[1] += tq2 - tq1; dq[2] += tq3 - tq2; dq[3] += tq4 - tq3; dq[4] += tq5 - tq4;
         dq[5] += tq6 - tq5;
       })
  S.st = (uint32_t)kind | ((uint32_t)level << 2) | (li << 13) | events;
  return S;
}
