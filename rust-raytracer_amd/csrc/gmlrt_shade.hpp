// Fragment of gmlrt_kernels.hip (included inside its anonymous namespace): everything
// trace() does around its intersect calls (raytracer.rs:53-110), cut at those calls.
//
// after_traversal consumes the traversal result of the ray a lane holds and leaves
// either the next ray to traverse in the lane state or an idle lane with its pixel
// written.  It runs once per traced ray, out of line, so that the traversal loop
// keeps a small register footprint.  Values that must survive the next traversal
// are parked in the lane's context record.  The function is written so that few
// values are live at any point (each phase stores what it produced and the next
// one loads what it needs): what it keeps in registers beyond the caller-saved set
// costs a save and a restore at every call.

// The lane's context record.  "Hot" fields are read and written around every shadow
// ray; when the block's LDS has room for them they live there (kLds), lane-strided, in
// the region the surface bytecode's registers use while a hit is being shaded (the two
// never hold live data at the same time).  Otherwise they are the first fields of the
// lane's record in device memory.  "Cold" fields always live in that record.
//   hot field f of this lane:  kLds ? lds[f * 256] : ctx[f * 64]      cold: ctx[f * 64]
enum { H_POS = 0, H_NRM = 3, H_SC = 6, H_KD = 9, H_KS = 10, H_N = 11, H_DIFF = 12, H_SPEC = 15, kHotFields = 18,
       C_RPOS = 18, C_RDIR = 21 };
static_assert(kCtxFixed == 24, "cold fields end where the parked bounce records begin");
#define COLD(f) ctx[(size_t)(f) * kWave]
#define HOT(f) hot[(size_t)(f) * kHotStride]
__device__ inline V3 cold_get3(const double* ctx, int f) { return {COLD(f), COLD(f + 1), COLD(f + 2)}; }
__device__ inline void cold_set3(double* ctx, int f, V3 v) { COLD(f) = v.x; COLD(f + 1) = v.y; COLD(f + 2) = v.z; }
#define HOT_GET3(f) V3{HOT(f), HOT((f) + 1), HOT((f) + 2)}
#define HOT_SET3(f, v) do { const V3 v_ = (v); HOT(f) = v_.x; HOT((f) + 1) = v_.y; HOT((f) + 2) = v_.z; } while (0)

// The traversal result travels as two scalars (distance; meta with kHitAny set when
// there is a hit): together with the lane state an aggregate would exceed what the
// calling convention passes in registers and go through the stack.
//
// What the call did (kStShadow / kStReflect / kStHit) is written through events_out, which
// points at a local variable of the caller, on purpose: a call that is handed the address of
// one of its caller's stack slots cannot be marked as a tail call, and only then may the
// compiler treat this function (internal, not recursive) as one that preserves no registers.
// Without that it saves and restores every callee-saved register it touches, about fifty
// VGPRs to scratch memory and three dozen SGPRs, on every call; with it the caller spills
// only the handful of values it really keeps across the call.
//
// The lane's context record and its slice of the block's LDS region are found from the thread's own position
// (ShadeArgs has the bases): as arguments they would be four more registers, and arguments beyond what the
// calling convention passes in registers travel through scratch memory at every call.
//
// kSlot: rays change lanes between traversals (the kernel that regroups the rays of a block), so the context belongs to
// the ray, not to the lane: bits 31..24 of S.st name the block's context slot the ray owns, and that slot takes the place
// of the thread's position in every address below.  The bits are handed back unchanged.
//
// kLevels: the frame is rendered level by level; a bounce ray leaves this lane (the caller queues it for the next
// launch), so what trace() keeps across its recursive call, (c, sc, ks), is parked per PIXEL and level, where the
// lane that ends the pixel's chain of bounces, in whatever launch, finds it.
// The direction of the camera ray of output pixel `pix` (row-in-band * w + x), made the way the pixel hand-out of the
// kernel makes it (GMLRT_CAMERA_RAY; raytracer.rs:138-142): the same operations on the same values, so the same bits.
__device__ inline V3 camera_ray_dir(const ShadeArgs* __restrict__ SA, uint32_t pix) {
  const uint32_t w = (uint32_t)SA->w;
  const uint32_t yl = pix / w, x = pix - yl * w;
  const uint32_t yp = yl + (uint32_t)SA->band_row0;
  uint32_t y = yp;
  if (SA->nparts > 1) {
    const uint32_t k = yp / (uint32_t)SA->tile_rows;
    y = ((uint32_t)SA->part + k * (uint32_t)SA->nparts) * (uint32_t)SA->tile_rows + yp % (uint32_t)SA->tile_rows;
  }
  const double xf = (double)x, yf = (double)y;
  const V3 dir = {SA->c_x + (xf + 0.5) * SA->pw, SA->c_y - (yf + 0.5) * SA->pw, 1.0};
  return vnormalize(dir);
}

template <bool kLds, bool kSlot, bool kLevels = false>
__device__ GMLRT_AFTER_INLINE LaneState after_traversal(LaneState S, const double h_dist, const uint32_t h_meta_any,
                                                  const ShadeArgs* __restrict__ SA_v, uint32_t* __restrict__ events_out) {
  const ShadeArgs* __restrict__ SA = const_space(wave_uniform(SA_v));
  extern __shared__ __align__(16) uint16_t lens[];
  const uint32_t my = kSlot ? (S.st >> 24) : threadIdx.x;
  uint64_t* const vm_fast_v = (uint64_t*)((uint8_t*)lens + SA->vm_lds_off) + my;
  double* const ctx_v = SA->ctx + ((size_t)blockIdx.x * (kBlock / kWave) + my / kWave) * SA->ctx_fields * kWave + (my & (kWave - 1));
  double* __restrict__ ctx = global_space(ctx_v);
  constexpr size_t kHotStride = kLds ? (size_t)kBlock : (size_t)kWave;
  double* hot;
  if constexpr (kLds) hot = (double*)(GMLRT_AS(3) double*)(uintptr_t)vm_fast_v;
  else hot = ctx;
  const gmlrt_leaf* __restrict__ leaves = const_space(wave_uniform(SA->leaves));
  const gmlrt_light* __restrict__ lights = const_space(wave_uniform(SA->lights));
  const bool h_any = (h_meta_any & kHitAny) != 0;
  const uint32_t h_meta = h_meta_any & ~kHitAny;
  int kind = st_kind(S.st);
  int level = (int)((S.st >> 2) & 0x7FFu);
  uint32_t li = (S.st >> 13) & 0x7FFu;
  uint32_t events = 0;
  const uint32_t n_lights = SA->n_lights;
  bool advance = false, finish = false;

  DIAG(unsigned long long tq0 = __builtin_readcyclecounter(), tq1, tq2, tq3, tq4, tq5, tq6, tq_runs = 0, tq_lanes = 0;)
  // ---- raytracer.rs:58-62: geometry of a view ray's hit in the primitive's object space
  bool shade = false;
  uint32_t mysurf = kNoSurfProg;
  int face = 0;
  double u = 0.0, v = 0.0;
  V3 opos = {0, 0, 0};
  if (kind == kView) {
    if (!h_any || (h_meta & kCurExit)) {
      finish = true;  // raytracer.rs:55-57, 108-110
    } else {
      shade = true;
      const gmlrt_leaf* L = leaves + (h_meta & kLeafMask);
      face = (int)((h_meta >> kFaceShift) & 7u);
      const uint32_t type = L->type;
      mysurf = const_space(wave_uniform(SA->leaf_surf))[(h_meta & kLeafMask) * kSurfFaces + (uint32_t)face];
      double scale; V3 rd, rp;
      to_object(L->inv_m, S.o, S.d, scale, rd, rp);
      PrimHit ph = isect_prim(type, rp, rd);
      const double od = (h_meta & kOrigExit) ? ph.tB : ph.tA;
      opos = vadd(rp, vmul(rd, od));
      prim_uv(type, face, opos, u, v);
      // the view ray is needed again for the highlight and the bounce; S is free from here on.  A camera ray is not
      // kept: its origin is the eye and its direction follows from the pixel (camera_ray_dir), which is cheaper than
      // 48 bytes a lane out to memory and back for every pixel that shows anything.
      if (kLevels || level != 0) {
        cold_set3(ctx, C_RPOS, S.o);
        cold_set3(ctx, C_RDIR, S.d);
      }
    }
  }

  // ---- raytracer.rs:63: the surface function.  Each distinct program among the wave's hits
  // is interpreted once, by all lanes in uniform control flow (so its records come by
  // scalar loads); a lane keeps the result of its own program.  Lanes without a hit run
  // along on zeros without storing anything; the bytecode has no trapping operation.
  DIAG(tq1 = __builtin_readcyclecounter();)
  // The results (colour, kd, ks, n) go straight to their hot fields H_SC .. H_N, stored only by the
  // lanes that run their own program.  With the hot fields in LDS those six slots lie inside the region
  // the interpreter's registers use, so registers 6 and up are kept six slots further on.
  {
    typedef GMLRT_AS(3) uint64_t* LdsP;
    typedef GMLRT_AS(1) uint64_t* GlobalP;
    typedef const GMLRT_AS(4) SurfInstr* ProgP;
    const VmRegs<LdsP, GlobalP> R = {
        (LdsP)(uintptr_t)vm_fast_v, (uint32_t)kBlock, SA->n_vm_fast,
        (GlobalP)(uintptr_t)(ctx_v + (size_t)(SA->ctx_fields - GMLRT_SURF_MAX_REGS) * kWave), (uint32_t)kWave,
        shade,  // the region holds live context fields of the lanes that are not shading a hit
        kLds ? (uint32_t)H_SC : (uint32_t)GMLRT_SURF_MAX_REGS, kLds ? 6u : 0u};
    struct HotOut {
      double* hot;
      bool mine;
      uint32_t* err;
      __device__ void put(uint32_t first, double a, double b, double c) const {
        if (mine) { HOT(H_SC + first) = a; HOT(H_SC + first + 1) = b; HOT(H_SC + first + 2) = c; }
      }
      // the reference would have aborted on this hit's surface evaluation (lanes that only run along do not count)
      __device__ void trap() const { if (mine) *err = 1u; }
    };
    static_assert(H_KD == H_SC + 3 && H_KS == H_SC + 4 && H_N == H_SC + 5, "result slots 0..5 map onto H_SC..H_N");
    const ProgP progs = (ProgP)(uintptr_t)wave_uniform(SA->surf_prog);
    const bool all_fast = SA->vm_all_fast != 0;
    bool need = shade && mysurf != kNoSurfProg;
    unsigned long long todo = __ballot(need);
    DIAG(unsigned long long vm_runs = 0; const unsigned long long vm_lanes = (unsigned long long)__popcll(todo);)
    while (todo) {
      DIAG(vm_runs++;)
      const int src = __builtin_ctzll(todo);
      const uint32_t cur = (uint32_t)__builtin_amdgcn_readlane((int)mysurf, src);
      const HotOut out = {hot, need && mysurf == cur, global_space(wave_uniform(SA->surf_error))};
      if (all_fast) surface_stream_vm<true>(progs + cur, (int64_t)face, u, v, out, R);
      else surface_stream_vm<false>(progs + cur, (int64_t)face, u, v, out, R);
      if (mysurf == cur) need = false;
      todo = __ballot(need);
    }
    DIAG(tq_runs = vm_runs; tq_lanes = vm_lanes;)
  }

  DIAG(tq2 = __builtin_readcyclecounter();)
  if (shade) {
    // ---- raytracer.rs:63-68: surface result, position and normal in world space
    const gmlrt_leaf* L = leaves + (h_meta & kLeafMask);
    if (mysurf == kNoSurfProg) {
      const uint64_t* kc = const_space(wave_uniform(SA->consts)) + L->surf_const;
#pragma unroll
      for (int q = 0; q < 6; q++) HOT(H_SC + q) = vm_as_f64(kc[q]);
    }
    HOT_SET3(H_DIFF, (V3{0.0, 0.0, 0.0}));
    HOT_SET3(H_SPEC, (V3{0.0, 0.0, 0.0}));
    const double* m = L->m;
    const V3 pos = {m[0] * opos.x + m[1] * opos.y + m[2] * opos.z + m[3], m[4] * opos.x + m[5] * opos.y + m[6] * opos.z + m[7],
                    m[8] * opos.x + m[9] * opos.y + m[10] * opos.z + m[11]};
    HOT_SET3(H_POS, pos);
    const V3 no = prim_normal(L->type, face, opos);
    const double* im = L->inv_m;  // transform_normal: transpose(inv_m) * n  (transform.rs:34-36)
    const V3 nw = {im[0] * no.x + im[4] * no.y + im[8] * no.z, im[1] * no.x + im[5] * no.y + im[9] * no.z,
                   im[2] * no.x + im[6] * no.y + im[10] * no.z};
    V3 normal = vnormalize(nw);
    if (((h_meta & kCurExit) != 0) != ((h_meta & kOrigExit) != 0)) normal = vneg(normal);
    HOT_SET3(H_NRM, normal);
    li = 0;
    events |= kStHit;
    advance = true;
  }
  DIAG(tq3 = __builtin_readcyclecounter();)
  if (kind == kShadow) {
    // ---- raytracer.rs:76-93: is light `li` visible along the shadow ray just traced
    const gmlrt_light& l = lights[li];
    const V3 pos = HOT_GET3(H_POS);
    V3 ldir; bool has_ld; double ld;
    light_direction(l, pos, ldir, has_ld, ld);
    bool lit = !h_any;
    if (!lit && has_ld && ld < h_dist) lit = true;
    if (lit) {
      const V3 normal = HOT_GET3(H_NRM);
      const double df = vdot(normal, ldir);
      const V3 ic = vcmul(HOT_GET3(H_SC), light_intensity(l, pos));
      if (HOT(H_KD) > 0.0) HOT_SET3(H_DIFF, vadd(HOT_GET3(H_DIFF), vmul(ic, df)));
      if (HOT(H_KS) > 0.0) {
        // get_specular, raytracer.rs:34-42
        const V3 raypos = (kLevels || level != 0) ? cold_get3(ctx, C_RPOS) : V3{0.0, 0.0, -1.0};
        const V3 halfway = vnormalize(vadd(ldir, vnormalize(vsub(raypos, pos))));
        const double sp = vdot(normal, halfway);
        V3 add = {0.0, 0.0, 0.0};
        if (sp > 0.0) add = vmul(ic, pow(sp, HOT(H_N)));
        HOT_SET3(H_SPEC, vadd(HOT_GET3(H_SPEC), add));
      }
    }
    li++;
    advance = true;
  }

  DIAG(tq4 = __builtin_readcyclecounter();)
  V3 C = {0.0, 0.0, 0.0};  // value of trace() at `level` when finish is set
  if (advance) {
    // ---- raytracer.rs:69-74: next light that faces the surface gets a shadow ray
    const V3 pos = HOT_GET3(H_POS), normal = HOT_GET3(H_NRM);
    bool spawned = false;
    while (li < n_lights) {
      V3 ldir; bool has_ld; double ld;
      light_direction(lights[li], pos, ldir, has_ld, ld);
      if (vdot(normal, ldir) > 0.0) {
        S.o = vadd(pos, vmul(ldir, 1e-7));
        S.d = ldir;
        // all the light test will ask is whether the first hit lies at or before the light
        // (any hit at all for a light at infinity); rounded down so that the float never exceeds it
        S.t_stop = has_ld ? (float)(ld * (1.0 - 1e-6)) : __builtin_inff();
        kind = kShadow;
        events |= kStShadow;
        spawned = true;
        break;
      }
      li++;
    }
    if (!spawned) {
      // ---- raytracer.rs:96-107
      const V3 sc = HOT_GET3(H_SC);
      const double kd = HOT(H_KD), ks = HOT(H_KS);
      const V3 c0 = vmul(vcmul({SA->amb[0], SA->amb[1], SA->amb[2]}, sc), kd);  // get_ambient, raytracer.rs:30-32
      const V3 c = vadd(c0, vadd(vmul(HOT_GET3(H_DIFF), kd), vmul(HOT_GET3(H_SPEC), ks)));
      if (ks > 0.0 && (SA->depth - level) > 0) {
        const V3 raydir = (kLevels || level != 0) ? cold_get3(ctx, C_RDIR) : camera_ray_dir(SA, S.pix);
        const V3 refl = vnormalize(vsub(raydir, vmul(normal, 2.0 * vdot(raydir, normal))));
        // park (c, sc, ks) until the bounce below returns
        if constexpr (kLevels) {
          double* __restrict__ pk = global_space(wave_uniform(SA->parked)) + ((size_t)S.pix * SA->parked_stride + (uint32_t)level) * 8;
          pk[0] = c.x; pk[1] = c.y; pk[2] = c.z; pk[3] = sc.x; pk[4] = sc.y; pk[5] = sc.z; pk[6] = ks;
        } else {
          const int f = kCtxFixed + level * kCtxPerLevel;
          cold_set3(ctx, f, c);
          cold_set3(ctx, f + 3, sc);
          COLD(f + 6) = ks;
        }
        S.o = vadd(pos, vmul(refl, 1e-7));
        S.d = refl;
        S.t_stop = -__builtin_inff();
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
      V3 c, psc;
      double pks;
      if constexpr (kLevels) {
        const double* __restrict__ pk = global_space(wave_uniform(SA->parked)) + ((size_t)S.pix * SA->parked_stride + (uint32_t)lv) * 8;
        c = {pk[0], pk[1], pk[2]};
        psc = {pk[3], pk[4], pk[5]};
        pks = pk[6];
      } else {
        const int f = kCtxFixed + lv * kCtxPerLevel;
        c = cold_get3(ctx, f);
        psc = cold_get3(ctx, f + 3);
        pks = COLD(f + 6);
      }
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
                                  ((size_t)blockIdx.x * (kBlock / kWave) + threadIdx.x / kWave) * kDiagWaveSlots;
         dq[0] += tq1 - tq0; dq[1] += tq2 - tq1; dq[2] += tq3 - tq2; dq[3] += tq4 - tq3; dq[4] += tq5 - tq4;
         dq[5] += tq6 - tq5; dq[6] += tq_runs; dq[7] += tq_lanes;
       })
  *events_out = events;
  S.st = (uint32_t)kind | ((uint32_t)level << 2) | (li << 13) | (kSlot ? my << 24 : 0u);
  if constexpr (!kSlot) {
    // a new ray: is it one that must be traced exactly, without bounds (ray_special_case)?  The first classes of each
    // kind come out of the launch record, all in one round of scalar loads; what a scene has beyond them, through the list.
    if (kind != kIdle && (SA->n_axes | SA->n_cone_forms) != 0u) {
      const bool odd = ray_special_inline(SA, S.d);
      if (odd) S.st |= kStOdd;
    }
  }
  return S;
}
