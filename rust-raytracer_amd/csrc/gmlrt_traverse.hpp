// Fragment of gmlrt_kernels.hip (included inside its anonymous namespace, after
// merge_lists and the event-meta constants): the traversal the kernel runs.
//
// It interprets the derived program of gmlrt_program.cpp, a byte stream of records
// with inline operands, so each step is one round of scalar loads from a
// wave-uniform address.  The tree of unions under the root keeps only the nearest
// first Entry of its children (`best`); every other operator is evaluated with
// lists exactly as intersect_general does.
//
// Two things cut a lane's work short without changing what its caller sees:
// * a subtree whose bound cannot be reached before the lane's current `best` cannot
//   supply a nearer Entry (an equal one loses the tie to the earlier candidate anyway);
// * t_stop: the caller only wants to know whether the first hit lies at or before that
//   distance (a shadow ray's test against its light, raytracer.rs:76-79); as soon as a
//   lane has such a candidate it stops taking part, and bounds are skipped by the wave
//   once no lane is left that needs them.  Camera and bounce rays pass -infinity.

#ifndef GMLRT_TRAVERSE_INLINE
#define GMLRT_TRAVERSE_INLINE inline
#endif
#if defined(GMLRT_LEAF_LOOK_CHECK) && GMLRT_LEAF_LOOK
#define LOOKCHK(...) __VA_ARGS__
#else
#define LOOKCHK(...)
#endif
// Per-lane flags of the traversal, kept as bits of one integer register.  (As separate
// booleans they are wave masks in scalar register pairs, and every join of control flow
// in the loop below merges each of them with three scalar instructions.)  "A candidate has
// been found" is bit 31 of best_m, the bit the caller's interface uses for it anyway.
constexpr uint32_t kFlActive = 1u, kFlRedo = 2u;

__device__ GMLRT_TRAVERSE_INLINE Hit intersect_first(const gmlrt_leaf* __restrict__ leaves, const uint8_t* __restrict__ prog,
                                      const uint32_t pc_start, uint32_t n_leaves, bool active_in, V3 o, V3 d, const double t_stop,
                                      double* __restrict__ evD, uint32_t* __restrict__ evM,
                                      uint16_t* __restrict__ lens, Diag& dg) {
  Hit h = {false, 0.0, 0u, false};
  if (n_leaves == 1) {
    // The root is a bare primitive: its raw list is what trace sees, a lone Exit included.
    double scale; V3 rd, rp;
    to_object(leaves->inv_m, o, d, scale, rd, rp);
    PrimHit ph = isect_prim(leaves->type, rp, rd);
    if (active_in && ph.bits) {
      h.any = true;
      if (ph.bits & 1) { h.dist = scale * ph.tA; h.meta = ((uint32_t)ph.fA << kFaceShift); }
      else { h.dist = scale * ph.tB; h.meta = ((uint32_t)ph.fB << kFaceShift) | kOrigExit | kCurExit; }
    }
    return h;
  }
  uint32_t fl = active_in ? kFlActive : 0u;
  double best = 0.0;
  uint32_t best_m = 0u;  // bit 31 (kHitAny): a candidate has been found
  uint32_t sp = 0;       // events on this lane's stack
  uint32_t level = 0;    // operand stack depth (wave-uniform)
  const int tid = threadIdx.x;
  uint32_t pc = pc_start;  // byte offset of the current record (wave-uniform): 0, or where a slice of the frontier starts
  // For the box tests, in single precision: t = p * inv - o * inv for a slab plane p, one fused operation per
  // plane.  Against the exact value the computed t is off by at most
  //     E = 2^-21 * (|p| + |o|) * |inv|      (p rounded outwards to f32: 2^-23 relative; inv and o * inv rounded
  //                                            to f32: 2^-24 each; the fused operation: 2^-24 of |t| <= (|p| + |o|) |inv|)
  // and every slab interval is widened by the largest E of the three axes before the intervals are compared, so
  // a ray that meets the box in exact arithmetic is never turned away.  A zero direction component gives
  // inv = +-inf and t = NaN (inf - inf) or +-inf; min/max skip a NaN operand, which leaves that slab out, and a
  // NaN or infinite E keeps the subtree: conservative in every case.
  const double inv_dx = 1.0 / d.x, inv_dy = 1.0 / d.y, inv_dz = 1.0 / d.z;
  const float inv_x = (float)inv_dx, inv_y = (float)inv_dy, inv_z = (float)inv_dz;
  const float noi_x = -(float)(o.x * inv_dx), noi_y = -(float)(o.y * inv_dy), noi_z = -(float)(o.z * inv_dz);
  constexpr float kErr = 4.76837158203125e-07f;  // 2^-21
  const float ie_x = kErr * __builtin_fabsf(inv_x), ie_y = kErr * __builtin_fabsf(inv_y), ie_z = kErr * __builtin_fabsf(inv_z);
  const float oe_x = __builtin_fabsf((float)o.x) * ie_x, oe_y = __builtin_fabsf((float)o.y) * ie_y, oe_z = __builtin_fabsf((float)o.z) * ie_z;
// (the element is copied first: a bit cast applied directly to a vector element lvalue read element 0 whatever the index)
#define REC_F32(i) __uint_as_float((uint32_t)R[2 + (i)])
  typedef uint32_t u32x16 __attribute__((ext_vector_type(16)));
#define REC_F64(i) __builtin_bit_cast(double, (uint64_t)R[2 + 2 * (i)] | ((uint64_t)R[3 + 2 * (i)] << 32))
  for (;;) {
    // The first 64 bytes of the record, whatever its kind, in one scalar load: fetching the
    // kind first and the operands after the branch on it would be two dependent round trips
    // per record.  (The program is padded so that this never reads past its buffer.)
    u32x16 R = *(const u32x16*)(const void*)(prog + pc);
    uint32_t kf = R[0];
    DIAG(dg.c_t = __builtin_readcyclecounter();)
#define DIAG_KIND(q) DIAG({ const unsigned long long n_ = __builtin_readcyclecounter(); dg.c_kind[q] += n_ - dg.c_t; dg.n_kind[q]++; dg.c_t = n_; })
    level = (uint32_t)__builtin_amdgcn_readfirstlane((int)level);  // wave-uniform by construction: say so
    // Bounds come in runs (the way down a hierarchy) and change nothing a lane carries, only where
    // the wave reads next: they get a loop of their own, whose only state is scalar.  (In one loop
    // with the other record kinds every bound record paid for copying the lanes' candidate, flags
    // and stack height from one iteration to the next.)
    // (The loop asks for a bit that only box records carry, not for the kind: a test of the kind is merged by the
    // compiler with the dispatch on the kind below, and every box then pays for that dispatch.  Which lanes are
    // active and which have a candidate does not change inside a run of bounds: those are wave masks set up once,
    // and the test of a bound is a few vector compares whose masks are combined with scalar operations.  Both
    // kinds of bound record are the same size.)
    if (kf & (kRecIsBox | kRecIsBound)) {
      const unsigned long long m_active = __ballot((fl & kFlActive) != 0);
#if GMLRT_TMAX_CULL
      const unsigned long long m_cand = __ballot((best_m & kHitAny) != 0);
#endif
      do {
        // A bound in front of a subtree: if no lane's ray comes near it, every list below is
        // empty.  Under the union frontier that means "nothing to take"; inside a frontier
        // child the subtree's (empty) list is pushed.  The tests only have to be conservative,
        // not reproducible, so they use fused multiply-adds.  Any NaN compares false and keeps
        // the subtree.
        unsigned long long m_hit;
        if (kf & kRecIsBox) {
          // ProgBox: skip_to, (lo, hi) per axis in f32, largest magnitude per axis.  Slab test.
          const float x1 = __builtin_fmaf(REC_F32(0), inv_x, noi_x), x2 = __builtin_fmaf(REC_F32(1), inv_x, noi_x);
          const float y1 = __builtin_fmaf(REC_F32(2), inv_y, noi_y), y2 = __builtin_fmaf(REC_F32(3), inv_y, noi_y);
          const float z1 = __builtin_fmaf(REC_F32(4), inv_z, noi_z), z2 = __builtin_fmaf(REC_F32(5), inv_z, noi_z);
          const float err = __builtin_fmaxf(__builtin_fmaxf(__builtin_fmaf(REC_F32(6), ie_x, oe_x), __builtin_fmaf(REC_F32(7), ie_y, oe_y)),
                                            __builtin_fmaf(REC_F32(8), ie_z, oe_z));
          const float tnear = __builtin_fmaxf(__builtin_fmaxf(__builtin_fminf(x1, x2), __builtin_fminf(y1, y2)), __builtin_fmaxf(__builtin_fminf(z1, z2), 0.0f));
          const float tfar = __builtin_fminf(__builtin_fminf(__builtin_fmaxf(x1, x2), __builtin_fmaxf(y1, y2)), __builtin_fmaxf(z1, z2));
          const float tn_lo = tnear - err;  // no entry before this
          m_hit = __ballot(!(tfar + err < tn_lo)) & m_active;
#if GMLRT_TMAX_CULL
          m_hit &= ~(m_cand & __ballot((double)tn_lo > best));
#endif
        } else {
          // the same with the subtree's sphere.  ProgBound: skip_to, center[3], radius2, radius.  |perp|^2 =
          // |oc|^2 - tca^2 (d has unit length); the slack added to r^2 covers the cancellation in that difference.
          const double ocx = REC_F64(0) - o.x, ocy = REC_F64(1) - o.y, ocz = REC_F64(2) - o.z;
          const double oc2 = __builtin_fma(ocx, ocx, __builtin_fma(ocy, ocy, ocz * ocz));
          const double tca = __builtin_fma(ocx, d.x, __builtin_fma(ocy, d.y, ocz * d.z));
          const double r2 = REC_F64(3);
          m_hit = m_active & ~__ballot(__builtin_fma(-tca, tca, oc2) > __builtin_fma(oc2, 1e-9, r2)) &
                  ~(__ballot(tca < 0.0) & __ballot(oc2 > r2));
#if GMLRT_TMAX_CULL
          m_hit &= ~(m_cand & __ballot(tca - REC_F64(4) > best));
#endif
        }
        DIAG(dg.n_bound++;)
        DIAG(dg.b_lanes += (unsigned long long)__popcll(m_hit);)
        if (m_hit) {
          DIAG(dg.k_entered[dg.kind]++; dg.k_blanes[dg.kind] += (unsigned long long)__popcll(m_hit);)
          pc += (uint32_t)sizeof(ProgBox);
          R = *(const u32x16*)(const void*)(prog + pc);
        } else {
          DIAG(dg.n_skip++;)
          if (!(kf & kRecFlag)) {
            lens[level * kBlock + tid] = 0;
            level++;
          }
          pc = R[1];
          R = *(const u32x16*)(const void*)(prog + pc);
        }
        kf = R[0];
      } while (kf & (kRecIsBox | kRecIsBound));
      DIAG_KIND(0);
    }
    if (kf & kRecIsLeaf) {
      // ProgLeaf: leaf, inv_m[12]; the matrix's last five elements lie beyond the first 64 bytes
      const ProgLeaf* __restrict__ L = (const ProgLeaf*)(prog + pc);
      const double im[12] = {REC_F64(0), REC_F64(1), REC_F64(2), REC_F64(3), REC_F64(4), REC_F64(5), REC_F64(6),
                             L->inv_m[7], L->inv_m[8], L->inv_m[9], L->inv_m[10], L->inv_m[11]};
      V3 td, rp;
      to_object_raw(im, o, d, td, rp);
      const double td2 = vdot(td, td);
      DIAG(dg.n_leaf++;)
      pc += (uint32_t)sizeof(ProgLeaf);
#if GMLRT_LEAF_LOOK
      // A look at the rays in the primitive's object space before the direction is normalised: when no active lane's
      // ray can have an Entry on the primitive (prim_may_enter: conservative), the leaf's list is empty for every
      // lane that matters (a list without an Entry counts as empty, see below) and the square root, the division
      // and the intersector are skipped.  In half of all leaf tests of a frame no lane has an Entry.
      const bool may = ((fl & kFlActive) != 0) & prim_may_enter((kf >> 16) & 7u, rp, td, td2);
#ifndef GMLRT_LEAF_LOOK_CHECK
      if (!__ballot(may)) {
        DIAG(dg.l_nohit++; if (dg.leaf_dbg && (threadIdx.x & 63) == (unsigned)__builtin_ctzll(__ballot(true))) {
               atomicAdd(&dg.leaf_dbg[R[1] * kLeafDbg], 1ull); atomicAdd(&dg.leaf_dbg[R[1] * kLeafDbg + 1], 1ull); atomicAdd(&dg.leaf_dbg[R[1] * kLeafDbg + 3], 1ull); })
        if (!(kf & kRecFlag)) {
          lens[level * kBlock + tid] = 0;
          level++;
        }
        DIAG_KIND(1);
        continue;
      }
#endif
#endif
      const double scale = 1.0 / sqrt(td2);
      const V3 rd = vmul(td, scale);
      PrimHit ph = isect_prim((kf >> 16) & 7u, rp, rd);
      const uint32_t leaf = R[1];
      // a list that starts with Exit never raises its operand's counter above zero
      // (primitives.rs:254-293 start both counters at 0): same as no list at all
      const bool entry = ((fl & kFlActive) != 0) & ((ph.bits & 1) != 0);
      const double d0 = scale * ph.tA, d1 = scale * ph.tB;
      const bool sliver = ((ph.bits & 2) != 0) & (d1 - d0 < 1e-10);
      DIAG(dg.l_lanes += (unsigned long long)__popcll(__ballot((fl & kFlActive) != 0)); dg.l_hits += (unsigned long long)__popcll(__ballot(entry));
           if (!__ballot(entry)) dg.l_nohit++;
           { const unsigned long long be = __ballot(entry);
             // (check build: the look decides nothing; counted are the tests it would have skipped and the lanes with
             // an Entry that it turned away, which must be none)
             LOOKCHK(const unsigned long long bm = __ballot(may); const unsigned long long bv = __ballot(entry & !may);)
             if (dg.leaf_dbg && (threadIdx.x & 63) == (unsigned)__builtin_ctzll(__ballot(true))) { atomicAdd(&dg.leaf_dbg[leaf * kLeafDbg], 1ull); if (!be) atomicAdd(&dg.leaf_dbg[leaf * kLeafDbg + 1], 1ull);
               atomicAdd(&dg.leaf_dbg[leaf * kLeafDbg + 2], (unsigned long long)__popcll(be));
               LOOKCHK(if (!bm) atomicAdd(&dg.leaf_dbg[leaf * kLeafDbg + 3], 1ull);
                       atomicAdd(&dg.leaf_dbg[leaf * kLeafDbg + 4], (unsigned long long)__popcll(bv));)
             } })
      if (kf & kRecFlag) {
        // frontier leaf: no list, just a candidate for the nearest first Entry
        // (selects, not branches: as nested conditions this was five levels of lane-mask bookkeeping for eight
        // instructions of work)
        const bool nearer = !(best_m & kHitAny) | (d0 < best) | ((d0 == best) & (leaf < (best_m & kLeafMask)));
        const bool upd = entry & !sliver & nearer;
        fl |= (entry & sliver) ? kFlRedo : 0u;
        best = upd ? d0 : best;
        best_m = upd ? (leaf | ((uint32_t)ph.fA << kFaceShift) | kHitAny) : best_m;
#if GMLRT_TSTOP
        fl &= (entry & ((best_m & kHitAny) != 0) & (best <= t_stop)) ? ~kFlActive : ~0u;
#endif
        DIAG_KIND(2);
        continue;
      }
      uint32_t n = 0;
      if (entry) {
        evD[(size_t)sp * kWave] = d0;
        evM[(size_t)sp * kWave] = leaf | ((uint32_t)ph.fA << kFaceShift);
        n = 1;
        if (ph.bits & 2) {
          evD[(size_t)(sp + 1) * kWave] = d1;
          evM[(size_t)(sp + 1) * kWave] = leaf | ((uint32_t)ph.fB << kFaceShift) | kOrigExit | kCurExit;
          n = sliver ? (2u | 0x8000u) : 2u;
        }
      }
      lens[level * kBlock + tid] = (uint16_t)n;
      sp += n & 0x7FFFu;
      level++;
      DIAG_KIND(3);
      continue;
    }
    if (kf & kRecIsOp) {
      const uint32_t rule = R[1] & 3u;
      const uint32_t lenB = lens[(level - 1) * kBlock + tid], lenA = lens[(level - 2) * kBlock + tid];
      const uint32_t lb = lenB & 0x7FFFu, la = lenA & 0x7FFFu;
      const bool slvA = lenA >> 15, slvB = lenB >> 15;
      const uint32_t a0 = sp - la - lb;
      // One empty operand: the result is the other list as it stands (union; a difference's left operand), minus
      // a lone sliver, or nothing.  Both non-empty: the sweep.  (Selects: the rule is the wave's, the lengths are
      // the lanes'.)
      const bool general = (la != 0) & (lb != 0);
      const uint32_t keepA = slvA ? 0u : la, keepB = slvB ? 0u : lb;  // (B already sits at a0 when A is empty)
      uint32_t lout = 0;
      if (rule == GMLRT_UNION) lout = lb == 0 ? keepA : la == 0 ? keepB : 0u;
      else if (rule != GMLRT_INTERSECT) lout = lb == 0 ? keepA : 0u;
      DIAG(if (__ballot(general)) dg.n_merge++;)
      if (general) lout = merge_any(R[1] & (3u | kOpSwapped), evD, evM, a0, la, lb);
      lens[(level - 2) * kBlock + tid] = (uint16_t)lout;
      sp = a0 + lout;
      level--;
      pc += (uint32_t)sizeof(ProgSmall);
      DIAG(if (__ballot(general)) { DIAG_KIND(5); } else { DIAG_KIND(4); })
      continue;
    }
    if (kf & kRecIsSkip) {
      // between the two operands of an intersection or a difference: the first one's list is empty for every
      // ray of the wave, so the operator's result is empty too; that empty list stays where it is and the
      // second operand and the operator are jumped over
      const uint32_t l = lens[(level - 1) * kBlock + tid] & 0x7FFFu;
      pc = __ballot(l != 0) ? pc + (uint32_t)sizeof(ProgSmall) : R[1];
      DIAG_KIND(6);
      continue;
    }
    if (kf & kRecIsTake) {
      // a frontier child is complete: its list is the only one on the stack and, being an
      // operator's output, starts with an Entry and holds no sliver
      const uint32_t n = lens[tid] & 0x7FFFu;
      DIAG(dg.n_take++;)
      if ((fl & kFlActive) && n) {
        const double d0 = evD[0];
        const uint32_t m0 = evM[0];
        // (both loaded values are used right here, without a nested condition: a load still on its way when the
        // loop goes round makes every record that follows wait for all memory traffic, event stores included,
        // before it may touch that register)
        const bool nearer = !(best_m & kHitAny) | (d0 < best) | ((d0 == best) & ((m0 & kLeafMask) < (best_m & kLeafMask)));
        best = nearer ? d0 : best;
        best_m = nearer ? (m0 | kHitAny) : best_m;
#if GMLRT_TSTOP
        if (best <= t_stop) fl &= ~kFlActive;
#endif
      }
      sp = 0;
      level = 0;
      pc += (uint32_t)sizeof(ProgSmall);
      DIAG_KIND(7);
      continue;
    }
    break;  // kRecEnd
  }
#undef DIAG_KIND
#undef REC_F64
#undef REC_F32
  h.any = (best_m & kHitAny) != 0;
  h.dist = best;
  h.meta = best_m & ~kHitAny;
  h.redo = (fl & kFlRedo) != 0;
  return h;
}
