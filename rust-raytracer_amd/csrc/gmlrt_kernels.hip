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
// The traversal loop is the hot code and is kept lean in registers; everything
// that happens once per traced ray (surface bytecode, lights, bounce bookkeeping)
// lives in one out-of-line function that parks its state in a per-lane context
// record in device memory.
//
// Arithmetic is f64 in the reference's operation order and the file is built with
// -ffp-contract=off, so intermediate values equal the CPU restatement's bit for
// bit; only libm-class functions (pow, atan2, asin ...) may differ in the last ulp.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <mutex>
#include <thread>
#include <vector>
#include <cstdlib>
#include <cstdio>
#include <chrono>
#include <cstring>

#include "gmlrt_device.hpp"
#include "gmlrt_prims.hpp"
#include "gmlrt_surface_vm.h"

namespace gmlrt {
namespace {

constexpr int kBlock = 256;
constexpr int kWave = 64;
constexpr uint32_t kLeafMask = 0xFFFFFu;
constexpr uint32_t kFaceShift = 20;
constexpr uint32_t kOrigExit = 1u << 23;  // the primitive produced this event as an Exit
constexpr uint32_t kCurExit = 1u << 24;   // current type after operator switches
constexpr uint32_t kHitAny = 1u << 31;    // set in the meta word handed to after_traversal: the ray hit something
constexpr int kCtxFixed = 24;             // f64 fields of the shading context before the bounce stack
constexpr int kCtxPerLevel = 7;
#ifndef GMLRT_TILE_W
#define GMLRT_TILE_W 8
#endif
// pixels of a tile: one wave's worth.  Measured, chess / fractal 4K in Mrays/s: 4 x 16 pixels 5410 / 11780,
// 8 x 8 5300 / 12190, 16 x 4 4790 / 11940, 32 x 2 4050 / 11090.
constexpr uint32_t kTileW = GMLRT_TILE_W, kTileH = 64 / GMLRT_TILE_W;
static_assert(kTileW * kTileH == 64, "a tile is one wave of pixels");
constexpr uint32_t kTileBatch = 4;      // tiles per claim when no batch table is given
constexpr uint32_t kTileBatchMax = 16;  // most tiles a batch of the table holds
// Atomics on one address are served one after the other, device-wide (about 14 ns each, measured): the
// batch counter is therefore split into several, far enough apart to be different memory channels.
constexpr uint32_t kClaimCounters = 8, kClaimStride = 4096;
// Level-by-level rendering: a queued ray is 8 words (origin, direction, pixel), a wave takes room for
// kQueueChunk of them with one atomic (one per ray-pushing wave and round would be one per 4 ns of a 4K frame
// on a single address, which serves one per 14 ns).
constexpr uint32_t kRayWords = 8, kQueueChunk = 256, kKeyInvalid = 0xFFFFFFFFu;
// Sort keys: the grid cell of the ray's origin (up to 8 x 8 cells over x and z, two layers in y) and its direction
// (octant, and 4 x 4 bins inside the octant).
constexpr uint32_t kKeyBitsMax = 3, kKeyDirBits = 7, kKeysMax = 1u << (2 * kKeyBitsMax + 1 + kKeyDirBits);
// The queued rays are put in the order of their keys by a counting sort whose blocks each take kSortSegment
// consecutive slots of the queue and count them in LDS first: a few keys hold most of the rays, and atomics on one
// address are served one at a time, so the global counters see one addition per block and key, not one per wave.
constexpr uint32_t kSortSegment = 4096;
static_assert(kSortSegment % kQueueChunk == 0 && kSortSegment % 64 == 0, "segments are whole chunks and whole runs of 64 rays");
constexpr uint32_t kMaxLevels = 1000;  // (level L claims its work from word L of each 4 KB claim counter)
// Words of the level-by-level counter buffer: chunks handed out of the queue that feeds level l, rays in that level's
// sorted table, the overflow flag, rays per key, where each key's run of the sorted table is filled next.
constexpr size_t kQcChunks = 0, kQcSorted = kMaxLevels + 2, kQcOverflow = 2 * (kMaxLevels + 2), kQcHist = kQcOverflow + 60,
                 kQcCursor = kQcHist + kKeysMax, kQcWords = kQcCursor + kKeysMax;
// Layout of the per-scene counter buffer: the claim counters, 256 bytes of diagnostic counters, the ShadeArgs
// record, one set of ray counts per block.
constexpr size_t kOffDiag = (size_t)kClaimCounters * kClaimStride, kOffShadeArgs = kOffDiag + 256, kOffBlockCounts = kOffDiag + 1024;
constexpr size_t kMaxGridBlocks = 8192;
constexpr size_t kOffSurfError = kOffBlockCounts - 8;  // inside the ShadeArgs slot, right in front of the block counts
constexpr size_t kCountersBytes = kOffBlockCounts + kMaxGridBlocks * 4 * sizeof(unsigned long long);
constexpr size_t kNoAllocScratchBytes = GMLRT_NOALLOC_SCRATCH_BYTES;
constexpr size_t kStagingBytes = 1024 + kMaxGridBlocks * 4 * sizeof(unsigned long long);  // host side: ShadeArgs, then the block counts
[[maybe_unused]] constexpr int kDiagWaveSlots = 28;  // diagnostic build: counters per wave (12.. : cycles and counts per record kind)

enum RayKind : int { kIdle = 0, kView = 1, kShadow = 2 };

// The reference's cylinder has two special cases, taken when the ray's direction in the cylinder's object space is
// within 1e-7 of parallel or of orthogonal to the axis (primitives.rs:644-663): they test only the ray's ORIGIN
// against the radius (or the two caps) and report the cap (or mantle) hit wherever the ray has drifted to by then,
// which may lie outside the solid, and so outside every bound built around it.  Such a ray must meet no bound at
// all.  The host lists the cylinders' axes in world space, one class per distinct direction, each with the cosines
// beyond / below which a unit direction may be in one of the two cases for some cylinder of the class (generous
// margins: a flagged ray only takes the slower exact path); the kernel flags a lane whose ray matches a class, keeps
// it out of the bounded traversal and traces it with the full-list algorithm without bounds.  About one ray in 10^6
// per class.  kAxisWords doubles a class: axis x y z, cosine above which "parallel", cosine below which "orthogonal".
__device__ inline bool cylinder_special_case(const double* __restrict__ axes, uint32_t n_axes, V3 d) {
  bool odd = false;
  for (uint32_t k = 0; k < n_axes; k++) {
    const double* __restrict__ a = axes + (size_t)k * kAxisWords;
    const double c = fabs(d.x * a[0] + d.y * a[1] + d.z * a[2]);
    odd |= (c >= a[3]) | (c <= a[4]);
  }
  return odd;
}

// The reference's cone has a case of the same kind: for a ray that runs along the mantle (parallel to a generator)
// the quadratic of primitives.rs:778-792 loses its leading coefficient, a = dx^2 + dz^2 - dy^2 comes out as a few
// units of rounding, and the "root" (-b + sqrt(b^2 - 4ac)) / 2a is the quotient of two rounding errors.  When that
// number happens to give a height between 0 and 1 the reference takes it for the one crossing of the mantle and adds
// the crossing of the base PLANE as the other (primitives.rs:844-848), wherever that lies: a hit that need not be
// anywhere near the solid, and that every bound around the solid culls.  (Found by the lattice fuzz: one scene in
// 20 000, a light at 45 degrees over a cone on its side.)  A ray counts as in the regime when |a| <= 1e-8 of its
// squared length in the cone's object space (well above the rounding of a itself, 4e-16; beyond it the small root is
// off by at most 2.2e-8 |b|, which the cone's bounds allow for: see leaf_bound, and a hundred times fewer rays take
// the walk through the whole tree than with 1e-6, thirty in an eighth of the chess frame); a = d^T (M^T J M) d for the world direction d, M the linear part of the cone's
// inverse transform and J = diag(1, -1, 1), so the host lists one symmetric form per distinct cone shape and
// orientation, scaled by 1 / trace(M^T M) (then |d_object|^2 <= |d|^2 and the bound is conservative), and the kernel
// flags a lane whose unit direction gives |d^T Q d| <= 1e-8 for some form; such a lane is traced exactly and without
// bounds, like a cylinder's.  kAxisWords doubles a form: q00 q11 q22 2q01 2q02 2q12 bound.
__device__ inline bool cone_special_case(const double* __restrict__ forms, uint32_t n_forms, V3 d) {
  const double xx = d.x * d.x, yy = d.y * d.y, zz = d.z * d.z, xy = d.x * d.y, xz = d.x * d.z, yz = d.y * d.z;
  bool odd = false;
  for (uint32_t k = 0; k < n_forms; k++) {
    const double* __restrict__ f = forms + (size_t)k * kAxisWords;
    const double q = __builtin_fma(f[0], xx, __builtin_fma(f[1], yy, __builtin_fma(f[2], zz, __builtin_fma(f[3], xy, __builtin_fma(f[4], xz, f[5] * yz)))));
    odd |= !(fabs(q) > f[6]);
  }
  return odd;
}

// Both looks at a new ray's direction.  The answer travels with the ray as a bit of its state word (kStOdd), set where
// the ray is made: in the per-ray code, which is out of line and has registers to spare.  (Asked in the kernel's main
// loop, once a round, the loop over the classes cost the traversal next to it 5 % on the chess frame: scalar registers
// held across it.)
__device__ inline bool ray_special_case(const double* __restrict__ axes, uint32_t n_axes, uint32_t n_forms, V3 d) {
  return (n_axes && cylinder_special_case(axes, n_axes, d)) || (n_forms && cone_special_case(axes + (size_t)n_axes * kAxisWords, n_forms, d));
}
constexpr uint32_t kOddInline = 4;
constexpr uint32_t kStOdd = 1u << 24;  // (not in the kernel that regroups the rays of a block, where bits 31..24 name the ray's context slot)

// Read-only scene sections.  Passed to the kernel as separate `const __restrict__`
// parameters so that wave-uniform reads of them become scalar loads.
struct ScenePtrs {
  const gmlrt_leaf* __restrict__ leaves;
  const uint32_t* __restrict__ nodes;
  const gmlrt_light* __restrict__ lights;
  const uint64_t* __restrict__ code;
  const uint64_t* __restrict__ consts;
};

struct KArgs {
  uint32_t n_leaves, n_nodes, n_lights, max_stack;
  uint32_t n_prog;       // words of the derived program
  double amb[3];
  double c_x, c_y, pw;  // camera, raytracer.rs:131-136
  int32_t depth;
  int32_t w, h;
  int32_t tile_rows, part, nparts;
  int32_t rows_part;     // rows this launch renders: the partition's packed rows band_row0 .. band_row0 + rows_part - 1
  int32_t band_row0;
  uint32_t tiles_x;      // 8x8 pixel tiles per row band
  uint32_t n_slots;      // tiles_x * tiles_y * 64 pixel slots (padding included)
  uint8_t* rgb8;
  double* rgbf;
  double* ev_dist;
  uint32_t* ev_meta;
  uint32_t ev_cap;       // event slots per lane
  double* ctx;
  uint32_t ctx_fields;
  uint32_t n_vm_fast;    // surface-bytecode registers kept in LDS (the rest live in the context record)
  uint32_t vm_lds_off;   // byte offset of that LDS region (after the operand-stack lengths)
  uint32_t lds_ctx;      // 1: the region also holds the hot fields of the shading context
  const uint32_t* tile_order;  // tile to render for the i-th hand-out; null: i itself
  uint32_t stride;             // no tile_order: hand-out i renders tile (i * stride) % n_tiles (1: image order)
  const uint32_t* batch_start; // hand-outs [batch_start[b], batch_start[b + 1]) make up batch b; null: kTileBatch each
  uint32_t n_batches;
  uint32_t n_tiles;
  unsigned long long* tile_dbg;  // diagnostic build only: per tile, 4 counters
  unsigned long long* leaf_dbg;  // diagnostic build only: per leaf: tests, tests without an Entry in any lane, lanes with an Entry,
                                 // tests the look before the test skipped, lanes with an Entry that it turned away (check build; none)
  uint32_t* tile_cost;         // per tile: how long its wave spent on it (wall-clock ticks); null: not wanted
  // Tiles that set the length of a launch are rendered first, each by the four waves of a block together: block b
  // takes entries b, b + gridDim, ... of this list before it joins the ordinary hand-out.  slice_off: where the
  // program slice of each wave of the block starts.
  const uint32_t* tiles_shared;
  uint32_t n_tiles_shared;
  uint32_t tail_lds_off;           // byte offset of two LDS words for sharing the last tiles of a launch; 0: not done
  uint32_t regroup_mode;           // bits 1..0: what the rays are sorted by (tuning); bit 2: new tiles only for an idle block
  uint64_t* xchg;                  // the regrouping kernel: 256 x 8 words per block, the ray states on their way to their new lanes
  uint32_t slice_off[kProgSlices];
  unsigned int* next_batch;        // kClaimCounters counters, kClaimStride bytes apart: counter g hands out batches g, g + kClaimCounters, ...
  unsigned long long* block_counts;  // per block: primary, shadow, reflect, shaded hits
  unsigned long long* diag;        // diagnostic build only: per-phase cycle and event totals
  // Directions in which the reference's cylinder test leaves its general case (see kAxisWords): n_axes classes.
  const double* axes;
  uint32_t n_axes;
  uint32_t n_cone_forms;  // the cones' forms follow the cylinders' classes in `axes` (see cone_special_case)
  // Rendering level by level (render_kernel<.., 3>): a launch traces the view rays of ONE bounce level and their
  // shadow rays; a mirror bounce leaves its lane for a device-wide queue, which the next launch traces regrouped.
  uint32_t level;                  // bounce level this launch starts at: 0 camera rays from the pixel tiles, else rays from q_in
  uint32_t level_push;             // a bounce to this level is queued; one to a level below it stays in its lane
  const uint64_t* q_in;            // this level's rays, kRayWords words each ...
  const uint32_t* q_sorted;        // ... in the order of this table (positions into q_in) ...
  const uint32_t* q_in_count;      // ... which holds this many entries
  uint64_t* q_out;                 // the next level's rays, written in chunks of kQueueChunk records
  uint32_t* q_out_keys;            // their sort keys (kKeyInvalid: a slot of a chunk that was never filled)
  uint32_t* q_out_chunks;          // chunks handed out so far
  uint32_t* q_overflow;            // set when the queue is full (never, with the capacity the host gives it)
  uint32_t q_cap_chunks;
  uint32_t key_bits;               // sort keys: cells per axis = 1 << key_bits ...
  uint32_t key_mode;               // ... 0 no sorting (one key), 1 cell then direction octant, 2 octant then cell
  double key_lo[3], key_inv[3];
  // Sampled hand-out (render_kernel<true, 1>, a launch that knows no tile costs): see kSmClasses.
  uint32_t* sm_claimed;            // per tile: kTileFree until a wave takes the tile (an atomic exchange decides who)
  uint32_t* sm_list;               // kSmClasses lists of n_tiles entries: tiles published as dear (kTileFree: entry not written yet)
  uint32_t* sm_cnt;                // counters of the lists and statistics of the samples: see kSmLine
  uint32_t sm_groups_x, sm_groups; // groups of 4 x 4 tiles per row of groups, and in all
  uint32_t sm_stride;              // sample i is the one of group (i * sm_stride) % sm_groups
  uint32_t sm_lds_off;             // byte offset of the block's copy of the classes' milestones in LDS (kSmClasses words of 8 bytes)
  uint32_t sm_share[3], sm_factor[3];  // per class: share of the samples (of 256) that opens it; multiple of the average a sample must have run
};

// Sampled hand-out.  A frame that is rendered once knows no tile costs, and a launch ends when its last dear tile does
// (a tile between mirrors takes twenty times the average, and dear tiles come in regions: a piece and what it
// reflects).  So one tile of every 4 x 4 group, the sample, is handed out first, one per claim, and its wave watches the
// clock: a sample still running when seven eighths (96 %, 99 %) of all samples have finished, and for 3 (6, 12) times
// as long as those took on average, publishes the other tiles of its group in a list of that class (the shares make the
// classes the same part of every scene, the factors keep a scene of evenly priced tiles from publishing anything); a wave that needs work looks into the lists first, the highest class first, and takes a few
// tiles at a time from them; only then does it claim a batch of the plain order (a stride over all tiles).  Who renders a tile is decided by an atomic exchange on the tile's word, so every tile is rendered
// exactly once whichever way it was found.  Nothing here waits for another wave: an entry that a taker finds not
// yet written is passed over (the plain order reaches that tile in the end), and every loop is bounded.
constexpr uint32_t kSmClasses = 3, kSmGroup = 4, kTileFree = 0xFFFFFFFFu;
// Counters, each in a 128-byte line of its own (every wave of the device reads and bumps them, and the memory system
// serves accesses to one line one after the other): per class c, entries published at word (2c) * 32 and entries taken
// at (2c + 1) * 32; then one 64-bit word of statistics: how many samples have finished and the ticks they took.
// In the line after it, per class, what that word held when the share of samples that opens the class had finished (0 until then).
constexpr uint32_t kSmLine = 32, kSmStatWord = 2 * kSmClasses * kSmLine, kSmMileWord = kSmStatWord + kSmLine, kSmCntWords = kSmMileWord + kSmLine;
// the share of all samples (of 256) that must have finished before a sample still running may publish its group in class c

// how many times the finished samples' average a sample must have run to publish its group in class c (a wave takes
// one tile at a time from a list: dear tiles must spread over many waves)

// One entry of the highest class that has any (lane 0 of a wave only): class in bits 31..28, number of entries (1) in
// 27..24, the entry's position in 23..0; kTileFree: nothing published that is not taken.  The place is taken with an
// unconditional add: a compare-and-swap that fails whenever another wave got in between lets the whole device take
// one entry per round trip to memory (measured: 30 000 entries, 17 ms).  A wave that finds it has gone beyond what is
// published gives its place back; when that crosses other waves' adds an entry can be handed out twice or passed
// over, which costs nothing but the entry's priority: who renders a tile is decided by the exchange on the tile's own
// word, and the plain order reaches every tile that no list delivered.
__device__ inline uint32_t sm_take_dear(const KArgs& A, unsigned long long* __restrict__ stat_seen) {
  // (the milestone words are written three times a launch; the statistics word itself takes an atomic add per sample, and
  // loads from a line under atomic traffic are slow for everybody: 0.4 ms of a 2 ms frame, measured)
#pragma unroll
  for (uint32_t c = 0; c < kSmClasses; c++)
    stat_seen[c] = __hip_atomic_load((const unsigned long long*)(A.sm_cnt + kSmMileWord) + c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  uint32_t r[kSmClasses], t[kSmClasses];
#pragma unroll
  for (uint32_t c = 0; c < kSmClasses; c++) {
    r[c] = __hip_atomic_load(A.sm_cnt + (2 * c) * kSmLine, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    t[c] = __hip_atomic_load(A.sm_cnt + (2 * c + 1) * kSmLine, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
#pragma unroll
  for (int c = (int)kSmClasses - 1; c >= 0; c--) {
    if (t[c] >= r[c]) continue;
    const uint32_t at = atomicAdd(A.sm_cnt + (2 * c + 1) * kSmLine, 1u);
    if (at < r[c]) return ((uint32_t)c << 28) | (1u << 24) | at;
    atomicSub(A.sm_cnt + (2 * c + 1) * kSmLine, 1u);
  }
  return kTileFree;
}

// What the out-of-line per-ray code needs of the launch parameters.  It lives in
// device memory and is reached through a pointer: a by-value copy would travel
// through the stack (scratch memory) at every call.
struct ShadeArgs {
  double amb[3];
  int32_t depth;
  uint32_t n_lights;
  uint32_t ctx_fields;
  uint32_t n_vm_fast;  // surface-bytecode registers kept in LDS
  uint32_t vm_all_fast;  // 1: that is all the registers the scene's bytecode uses
  uint32_t pad;
  uint8_t* rgb8;
  double* rgbf;
  const gmlrt_leaf* leaves;
  const gmlrt_light* lights;
  const uint64_t* code;
  const uint64_t* consts;
  const uint32_t* leaf_surf;   // per leaf and face (kSurfFaces slots a leaf): first record of the surface program to run
  const SurfInstr* surf_prog;  // expanded surface programs
  unsigned long long* diag_waves;  // diagnostic build only: 8 counters per wave
  uint32_t* surf_error;        // set to 1 by a lane whose own surface program raised GMLRT_SOP_TRAP
  double* ctx;                 // context records of all waves (wave-major, then field, then lane)
  uint32_t vm_lds_off;         // byte offset of the block's register / hot-context region in its LDS
  // the camera (raytracer.rs:131-142) and the row layout of the launch: the per-ray code makes the camera ray of a
  // pixel again where it needs it (highlight, first bounce) instead of keeping it in the lane's context record
  double c_x, c_y, pw;
  int32_t w, tile_rows, part, nparts, band_row0;
  const double* axes;          // the cylinders' and cones' special directions (see ray_special_case)
  uint32_t n_axes, n_cone_forms;
  // the first kOddInline classes of each kind once more, in the record itself: the per-ray code fetches them in one round
  // of scalar loads, without first fetching a pointer (slots beyond a scene's classes hold classes that no ray matches)
  double odd_axis[kOddInline][4];  // axis x y z, cosine above which "parallel"
  double odd_axis_lo[kOddInline];  // cosine below which "orthogonal"
  double odd_form[kOddInline][6];  // q00 q11 q22 2q01 2q02 2q12 (the bound is kConeFormBound for every form)
  uint32_t parked_stride;      // level-by-level rendering: parked records per pixel (the depth of the call)
  double* parked;              // ... and the records: 8 words for pixel p and level l at (p * parked_stride + l) * 8
};

static_assert(sizeof(ShadeArgs) <= kOffSurfError - kOffShadeArgs, "ShadeArgs outgrew its slot in the counter buffer");

// Diagnostic build (-DGMLRT_DIAG): wave-level totals of what the traversal did and
// where the cycles went.  Never part of the shipped library; numbers from it are
// shares, not timings.
struct Diag { unsigned long long t_refill, t_trav, t_after, n_trav, n_bound, n_skip, n_leaf, n_merge, n_take, lanes, n_redo,
              b_lanes, l_lanes, l_hits, k_trav[3], k_leaf[3], l_nohit, k_entered[3], k_blanes[3], k_active[3], c_kind[8], n_kind[8], c_t; int kind;
              unsigned long long* leaf_dbg; };
// -DGMLRT_DIAG_SHARED: where the four waves of a block spend their time on shared tiles (own slice, waiting for
// the others' candidates, per-ray code, waiting for the next rays), per wave of the block.  Diagnostic builds only.
#ifdef GMLRT_DIAG_SHARED
#define DIAGS(...) __VA_ARGS__
#else
#define DIAGS(...)
#endif
#ifdef GMLRT_DIAG
#define DIAG(...) __VA_ARGS__
#else
#define DIAG(...)
#endif

// Pointers reach an out-of-line device function as generic ("flat") addresses in
// vector registers.  These helpers restate what the caller knows: which memory the
// pointer is in (so accesses become global_/ds_ instructions instead of flat_) and,
// for wave_uniform, that every lane holds the same value (so read-only data can be
// fetched with scalar loads).
#define GMLRT_AS(n) __attribute__((address_space(n)))
template <class T>
__device__ inline T* global_space(T* p) { return (T*)(GMLRT_AS(1) T*)(uintptr_t)p; }
template <class T>
__device__ inline T* lds_space(T* p) { return (T*)(GMLRT_AS(3) T*)p; }
template <class T>
__device__ inline const T* const_space(const T* p) { return (const T*)(const GMLRT_AS(4) T*)(uintptr_t)p; }
template <class T>
__device__ inline T* wave_uniform(T* p) {
  const uint64_t v = (uint64_t)p;
  const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)v), hi = __builtin_amdgcn_readfirstlane((uint32_t)(v >> 32));
  return (T*)(((uint64_t)hi << 32) | lo);
}

// ray_special_case for a new ray, asked of the launch record: the first kOddInline classes of each kind lie inline in
// it and arrive in one round of scalar loads; what a scene has beyond them comes through the list.
__device__ inline bool ray_special_inline(const ShadeArgs* __restrict__ SA, V3 d) {
  const uint32_t na = SA->n_axes, nf = SA->n_cone_forms;
  bool odd = false;
  if (na) {
#pragma unroll
    for (uint32_t k = 0; k < kOddInline; k++) {
      if (k >= na) break;
      const double c = fabs(__builtin_fma(d.x, SA->odd_axis[k][0], __builtin_fma(d.y, SA->odd_axis[k][1], d.z * SA->odd_axis[k][2])));
      odd |= (c >= SA->odd_axis[k][3]) | (c <= SA->odd_axis_lo[k]);
    }
    if (na > kOddInline) odd |= cylinder_special_case(const_space(wave_uniform(SA->axes)) + (size_t)kOddInline * kAxisWords, na - kOddInline, d);
  }
  if (nf) {
    const double xx = d.x * d.x, yy = d.y * d.y, zz = d.z * d.z, xy = d.x * d.y, xz = d.x * d.z, yz = d.y * d.z;
#pragma unroll
    for (uint32_t k = 0; k < kOddInline; k++) {
      if (k >= nf) break;
      const double* __restrict__ f = SA->odd_form[k];
      const double q = __builtin_fma(f[0], xx, __builtin_fma(f[1], yy, __builtin_fma(f[2], zz, __builtin_fma(f[3], xy, __builtin_fma(f[4], xz, f[5] * yz)))));
      odd |= !(fabs(q) > kConeFormBound);
    }
    if (nf > kOddInline) odd |= cone_special_case(const_space(wave_uniform(SA->axes)) + (size_t)(na + kOddInline) * kAxisWords, nf - kOddInline, d);
  }
  return odd;
}
// (out of line: for the rays that are not made by the per-ray code, i.e. camera rays and rays taken from a queue.  Asked
// of the class list in device memory instead of the launch record, a load after a load for every class, the camera rays'
// flags cost the chess frame 1.2 %; this way 0.4 %.)
__device__ __noinline__ uint32_t ray_special_bit(const ShadeArgs* __restrict__ sa_v, double dx, double dy, double dz) {
  return ray_special_inline(const_space(wave_uniform(sa_v)), V3{dx, dy, dz}) ? kStOdd : 0u;
}

// ---------------------------------------------------------------- CSG traversal
// First element of the root list.  redo: the shortcut traversal met a case it does
// not cover and this lane must be traced again with the full list algorithm.
struct Hit { bool any; double dist; uint32_t meta; bool redo; };

__device__ inline bool csg_rule(uint32_t rule, bool a, bool b) {
  return rule == GMLRT_UNION ? (a || b) : rule == GMLRT_INTERSECT ? (a && b) : (a && !b);
}

// Sort-then-sweep of primitives.rs:262-313 for two non-empty lists that sit one
// above the other on the lane's event stack (A at a0, B right after it).  Both
// inputs are already sorted, so the stable sort of their concatenation is a merge
// that takes A on ties.  Output goes above B and is then moved down onto A.
//
// kOpSwapped in `rule_sw`: the list at a0 is the reference's RIGHT operand (obj2) and the one after it the left
// (the traversal evaluated the right operand first); ties then go to the second list and the first feeds in2.
__device__ __noinline__ uint32_t merge_lists(uint32_t rule_sw, double* __restrict__ evD_v, uint32_t* __restrict__ evM_v,
                                             uint32_t a0, uint32_t la, uint32_t lb) {
  const uint32_t rule = rule_sw & 3u;
  const bool swapped = (rule_sw & kOpSwapped) != 0;
  double* __restrict__ evD = global_space(evD_v);
  uint32_t* __restrict__ evM = global_space(evM_v);
  const uint32_t b0 = a0 + la, out0 = b0 + lb;
  uint32_t i = 0, j = 0, k = 0;
  bool in1 = false, in2 = false, inside = false;
  double prevt = 0.0;
  double da = evD[(size_t)a0 * kWave], db = evD[(size_t)b0 * kWave];
  uint32_t ma = evM[(size_t)a0 * kWave], mb = evM[(size_t)b0 * kWave];
  const uint32_t total = la + lb;
  for (uint32_t n = 0; n < total; n++) {
    const bool takeA = (j >= lb) || (i < la && (swapped ? db > da : db >= da));
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
    if (takeA != swapped) in1 = entry; else in2 = entry;
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

// The same for short lists (1..4 events each), which is nearly all of them: every event is
// fetched up front with independent loads, merged and swept in registers, and the output
// written straight over A.  merge_lists above walks the lists in memory, one dependent load
// after the other; on a frontier child with a dozen nested operators that chain of round
// trips was the longest thing a ray could wait for.
//
// Order: the reference sorts the concatenation A ++ B stably by distance, i.e. by the total
// order (distance, position in A ++ B).  Both halves arrive sorted in that order (padding:
// +infinity), so Batcher's odd-even merge network for 4 + 4 elements yields exactly that
// sequence.  The position travels in three free bits of the meta word.
constexpr uint32_t kPosShift = 25;  // bits 27..25 of a meta word, free in event storage
__device__ inline void merge_cx(double& d0, uint32_t& m0, double& d1, uint32_t& m1) {
  const bool swap = d0 > d1 || (d0 == d1 && (m0 >> kPosShift) > (m1 >> kPosShift));
  const double dl = swap ? d1 : d0, dh = swap ? d0 : d1;
  const uint32_t ml = swap ? m1 : m0, mh = swap ? m0 : m1;
  d0 = dl; d1 = dh; m0 = ml; m1 = mh;
}
// kN = 4: up to 4 + 4 events; kN = 2: up to 2 + 2 (two primitives), the most common case, with a third of
// the comparators and half the loads and sweep steps.
template <int kN>
__device__ __noinline__ uint32_t merge_small(uint32_t rule_sw, double* __restrict__ evD_v, uint32_t* __restrict__ evM_v,
                                             uint32_t a0, uint32_t la, uint32_t lb) {
  const uint32_t rule = rule_sw & 3u;
  // swapped operands (see merge_lists): the first list is the reference's obj2, so its events take the
  // positions 4.. of A ++ B and the second list's the positions 0..
  const uint32_t pos_first = (rule_sw & kOpSwapped) ? 4u : 0u, pos_second = 4u - pos_first;
  double* __restrict__ evD = global_space(evD_v);
  uint32_t* __restrict__ evM = global_space(evM_v);
  const uint32_t b0 = a0 + la;
  double d[2 * kN];
  uint32_t m[2 * kN];
#pragma unroll
  for (uint32_t i = 0; i < (uint32_t)kN; i++) {
    // clamped index: always a slot of the list, so all the loads can be in flight together
    const uint32_t ia = a0 + (i < la ? i : la - 1u), ib = b0 + (i < lb ? i : lb - 1u);
    d[i] = evD[(size_t)ia * kWave];
    m[i] = evM[(size_t)ia * kWave];
    d[kN + i] = evD[(size_t)ib * kWave];
    m[kN + i] = evM[(size_t)ib * kWave];
  }
#pragma unroll
  for (uint32_t i = 0; i < (uint32_t)kN; i++) {
    if (i >= la) d[i] = __builtin_inf();
    if (i >= lb) d[kN + i] = __builtin_inf();
    // the position in A ++ B, with B's numbered from 4 in both sizes (bit 2 = "from B")
    m[i] = (m[i] & ~(7u << kPosShift)) | ((pos_first + i) << kPosShift);
    m[kN + i] = (m[kN + i] & ~(7u << kPosShift)) | ((pos_second + i) << kPosShift);
  }
  if constexpr (kN == 4) {
    merge_cx(d[0], m[0], d[4], m[4]); merge_cx(d[1], m[1], d[5], m[5]);
    merge_cx(d[2], m[2], d[6], m[6]); merge_cx(d[3], m[3], d[7], m[7]);
    merge_cx(d[2], m[2], d[4], m[4]); merge_cx(d[3], m[3], d[5], m[5]);
    merge_cx(d[1], m[1], d[2], m[2]); merge_cx(d[3], m[3], d[4], m[4]); merge_cx(d[5], m[5], d[6], m[6]);
  } else {
    // odd-even merge of two sorted pairs
    merge_cx(d[0], m[0], d[2], m[2]); merge_cx(d[1], m[1], d[3], m[3]);
    merge_cx(d[1], m[1], d[2], m[2]);
  }
  // the sweep of primitives.rs:262-313, as in merge_lists
  const uint32_t total = la + lb;
  uint32_t k = 0;
  bool in1 = false, in2 = false, inside = false;
  double prevt = 0.0;
#pragma unroll
  for (uint32_t n = 0; n < 2u * kN; n++) {
    if (n < total) {
      const bool from_b = ((m[n] >> kPosShift) & 4u) != 0;
      const uint32_t mm = m[n] & ~(7u << kPosShift);
      const bool entry = !(mm & kCurExit);
      if (from_b) in2 = entry; else in1 = entry;
      const bool ni = csg_rule(rule, in1, in2);
      if (inside && !ni) {
        if (d[n] - prevt < 1e-10) {
          k--;  // sliver: drop the Entry that opened it
        } else {
          evD[(size_t)(a0 + k) * kWave] = d[n];
          evM[(size_t)(a0 + k) * kWave] = mm | kCurExit;
          k++;
        }
      }
      if (!inside && ni) {
        evD[(size_t)(a0 + k) * kWave] = d[n];
        evM[(size_t)(a0 + k) * kWave] = mm & ~kCurExit;
        k++;
        prevt = d[n];
      }
      inside = ni;
    }
  }
  return k;
}
__device__ inline uint32_t merge_any(uint32_t rule, double* __restrict__ evD, uint32_t* __restrict__ evM, uint32_t a0,
                                     uint32_t la, uint32_t lb) {
  uint32_t out = 0;
  // (callers enter with the lanes that have two non-empty lists; the size class is chosen for the wave)
  const bool pair = la <= 2u && lb <= 2u, small = la <= 4u && lb <= 4u;
  if (!__ballot(!pair)) out = merge_small<2>(rule, evD, evM, a0, la, lb);
  else if (small) out = merge_small<4>(rule, evD, evM, a0, la, lb);
  else out = merge_lists(rule, evD, evM, a0, la, lb);
  return out;
}

__device__ inline double readlane_f64(double v, int j) {
  const uint64_t u = __builtin_bit_cast(uint64_t, v);
  const uint32_t lo = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)u, j), hi = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(u >> 32), j);
  return __builtin_bit_cast(double, ((uint64_t)hi << 32) | lo);
}
#ifdef GMLRT_DIAG
// diagnostic build: calls of intersect_general, leaves left after the marking, words of the tree visited, operators that
// merged two lists, time of the marking and of the walk (100 MHz clock)
__device__ unsigned long long g_general_stats[12];
#endif
// One primitive's test, and the look before it, for the full-list tracer; out of line.  What the tracer costs the
// kernel is not its own time (one ray in 10^5 gets there) but its registers: the compiler hands the kernel the exact
// set of registers the tracer and everything below it overwrite, and the kernel's loop keeps its long-lived values in
// the rest across the call.  A version of the tracer that also kept one ray's tests of 64 leaves in registers, a leaf
// per lane (22 more live registers; its walk was a quarter shorter), took registers the loop lives in: ten more
// values of the loop went to scratch, one of them reloaded inside the traversal, and every frame of every scene was
// 2.4 % slower.  The kernel's own frame (96 bytes a lane) is the figure to watch when this function changes.
__device__ __noinline__ PrimHit leaf_test(const gmlrt_leaf* L, V3 o, V3 d, double& scale) {
  V3 rd, rp;
  to_object(L->inv_m, o, d, scale, rd, rp);
  return isect_prim(L->type, rp, rd);
}
__device__ __noinline__ bool leaf_look(const gmlrt_leaf* L, V3 o, V3 d) {
  V3 td, rp;
  to_object_raw(L->inv_m, o, d, td, rp);
  return prim_may_enter(L->type, rp, td, vdot(td, td));
}
constexpr uint32_t kMarkLeavesFrom = 24;  // scenes with fewer leaves: the walk alone is shorter than the marking
// Node::intersect of the whole scene for one ray per lane, every operator evaluated
// with full lists; returns the first element of the root list (all that
// raytracer.rs:52-58, 74-79 look at).  This is the complete algorithm; the kernel
// normally runs intersect_first below and comes here only for the rare ray it flags.
// `lens` holds, per operand-stack level and lane, the length of that list; bit 15
// marks a two-event primitive list shorter than 1e-10 (it vanishes in any operator
// whose other operand is empty, primitives.rs:295-298).
__device__ __noinline__ Hit intersect_general(const gmlrt_leaf* __restrict__ leaves_v, const uint32_t* __restrict__ nodes_v,
                                              const gmlrt_bound* __restrict__ bounds_v, uint32_t n_nodes,
                                              uint32_t n_leaves, bool active, bool no_bounds, V3 o, V3 d, double* __restrict__ evD_v,
                                              uint32_t* __restrict__ evM_v, uint16_t* __restrict__ lens_v) {
  const gmlrt_leaf* __restrict__ leaves = global_space(wave_uniform(leaves_v));
  const uint32_t* __restrict__ nodes = global_space(wave_uniform(nodes_v));
  const gmlrt_bound* __restrict__ bounds = global_space(wave_uniform(bounds_v));
  double* __restrict__ evD = global_space(evD_v);
  uint32_t* __restrict__ evM = global_space(evM_v);
  // (as flat stores the writes of the list lengths held up every later use of a loaded value by the time of a trip
  // through the memory pipeline: 0.4 us for each word of the tree)
  uint16_t* __restrict__ lens = lds_space(lens_v);
  Hit h = {false, 0.0, 0u, false};
  if (n_leaves == 1) {
    // The root is a bare primitive: its raw list is what trace sees, a lone Exit included.
    double scale; V3 rd, rp;
    to_object(leaves->inv_m, o, d, scale, rd, rp);
    PrimHit ph = isect_prim(leaves->type, rp, rd);
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
  // Which leaves can matter at all.  The walk below visits every leaf of the scene for the one or two lanes that come
  // here (a ray traced without bounds skips nothing), one after the other, and each visit waits for the leaf's matrix:
  // 300 us for the 261 leaves of the chess scene, which sets the length of a launch when a frame is split over
  // several devices.  So the look that the walk would take at each leaf (prim_may_enter: conservative, bounds play
  // no part in it) is taken beforehand with the roles turned round: each of the wave's rays in turn is given to all
  // 64 lanes, every lane takes a leaf of its own, and a leaf on which none of the rays can have an Entry is marked
  // as empty.  Lane w keeps the marks of leaves 64 w .. 64 w + 63.
  const bool marked = n_leaves >= kMarkLeavesFrom && n_nodes <= 64u * 64u;
  uint64_t marks = 0, live = 0;
  DIAG(const unsigned long long gs_t0 = wall_clock64(); unsigned long long gs_words = 0, gs_merges = 0, gs_leaf_t = 0, gs_leaf_n = 0, gs_op_t = 0, gs_op_n = 0, gs_op0_t = 0, gs_op0_n = 0;)
  if (marked) {
    const uint32_t lane = (uint32_t)tid & 63u;
    for (unsigned long long todo = __ballot(active); todo; todo &= todo - 1ull) {
      const int j = __builtin_ctzll(todo);
      const V3 oj = {readlane_f64(o.x, j), readlane_f64(o.y, j), readlane_f64(o.z, j)};
      const V3 dj = {readlane_f64(d.x, j), readlane_f64(d.y, j), readlane_f64(d.z, j)};
      for (uint32_t w = 0; w * 64u < n_leaves; w++) {
        const uint32_t i = w * 64u + lane;
        const bool may = i < n_leaves && leaf_look(leaves + i, oj, dj);
        const uint64_t m = __ballot(may);
        if (lane == w) marks |= m;
      }
    }
    // The same marks per word of the tree: lane w keeps, for words 64 w .. 64 w + 63, which of them are leaves that
    // are not empty.  A subtree behind a bound word that holds none of them is empty and is jumped over.
    for (uint32_t w = 0; w * 64u < n_nodes; w++) {
      const uint32_t at = w * 64u + lane;
      const uint32_t word = at < n_nodes ? nodes[at] : 0xC0000000u;
      const uint32_t i = (word >> 30) == 0 ? (word & 0x3FFFFFFFu) : 0u;
      const uint32_t mlo = (uint32_t)__shfl((int)(uint32_t)marks, (int)(i >> 6)), mhi = (uint32_t)__shfl((int)(uint32_t)(marks >> 32), (int)(i >> 6));
      const bool is_live = (word >> 30) == 0 && ((((i & 32u) ? mhi : mlo) >> (i & 31u)) & 1u);
      const uint64_t m = __ballot(is_live);
      if (lane == w) live = m;
    }
  }
  DIAG(const unsigned long long gs_t1 = wall_clock64();)
  // The words of the tree come 64 at a time, one per lane, and are handed to the wave from there: a load per word
  // through the scalar cache waits for the L2 nearly every time (the cache is shared with waves that stream the
  // traversal program through it), half a microsecond a word.
  uint32_t words = 0, words_at = 0xFFFFFFFFu;
  // Bit l: the list at operand level l (below 64) is empty in every lane because nothing below it was looked at.  An
  // operator over two such lists has nothing to read and nothing to write: most of the operators of a large scene.
  uint64_t empty_levels = 0;
  auto level_bit = [](uint32_t l) -> uint64_t { return l < 64u ? 1ull << l : 0ull; };
  for (uint32_t pc = 0; pc < n_nodes; pc++) {
    if ((pc >> 6) != words_at) {
      words_at = pc >> 6;
      const uint32_t at = words_at * 64u + ((uint32_t)tid & 63u);
      words = at < n_nodes ? nodes[at] : 0u;
    }
    const uint32_t word = (uint32_t)__builtin_amdgcn_readlane((int)words, (int)(pc & 63u));
    DIAG(gs_words++;)
    if ((word >> 30) == 2 && marked) {
      // words pc + 1 .. pc + n of the tree
      const uint32_t first = pc + 1u, last = pc + (word & 0x7FFFu);
      bool any_live = false;
      for (uint32_t w = first >> 6; w <= (last >> 6) && !any_live; w++) {
        const uint64_t lw = (uint64_t)(uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)live, (int)w) |
                            ((uint64_t)(uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(live >> 32), (int)w) << 32);
        const uint32_t from = w == (first >> 6) ? (first & 63u) : 0u, to = w == (last >> 6) ? (last & 63u) : 63u;
        const uint64_t range = (~0ull << from) & (~0ull >> (63u - to));
        any_live = (lw & range) != 0;
      }
      if (!any_live) {
        lens[level * kBlock + tid] = 0;
        empty_levels |= level_bit(level);
        level++;
        pc += word & 0x7FFFu;
        continue;
      }
    }
    if ((word >> 30) == 0 && marked) {
      const uint32_t i = word & 0x3FFFFFFFu;
      const uint32_t mw = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(marks >> (i & 32u)), (int)(i >> 6));
      if (!((mw >> (i & 31u)) & 1u)) {
        lens[level * kBlock + tid] = 0;
        empty_levels |= level_bit(level);
        level++;
        continue;
      }
    }
    if ((word >> 30) == 2) {
      // Bounding sphere in front of a subtree: if no lane's ray comes near it, every
      // list below is empty, so is the subtree's: push that and jump over its words.
      if (!__ballot(active && !no_bounds)) continue;  // rays traced without bounds only: nothing to ask
      const gmlrt_bound* __restrict__ B = bounds + ((word >> 15) & 0x7FFFu);
      const V3 oc = {B->center[0] - o.x, B->center[1] - o.y, B->center[2] - o.z};
      const double tca = vdot(oc, d);
      const V3 perp = vsub(oc, vmul(d, tca));
      const double r2 = B->radius * B->radius;
      // any NaN compares false and keeps the subtree; a lane in one of the cylinder's special cases (no_bounds) keeps
      // every subtree
      const bool miss = !active || (!no_bounds && (vdot(perp, perp) > r2 || (tca < 0.0 && vdot(oc, oc) > r2)));
      if (!__ballot(!miss)) {
        lens[level * kBlock + tid] = 0;
        empty_levels |= level_bit(level);
        level++;
        pc += word & 0x7FFFu;
      }
      continue;
    }
    if ((word >> 30) == 0) {
      DIAG(const unsigned long long gs_l0 = wall_clock64();)
      double scale;
      PrimHit ph;
      ph = leaf_test(leaves + (word & 0x3FFFFFFFu), o, d, scale);
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
      empty_levels &= ~level_bit(level);
      sp += n & 0x7FFFu;
      level++;
      DIAG(gs_leaf_t += wall_clock64() - gs_l0; gs_leaf_n++;)
    } else {
      DIAG(const unsigned long long gs_o0 = wall_clock64();)
      const uint64_t both = level_bit(level - 1u) | level_bit(level - 2u);
      if (level <= 64u && (empty_levels & both) == both) {
        // empty with empty, under any rule: empty, and the zero already stands at level - 2
        empty_levels &= ~level_bit(level - 1u);
        level--;
        DIAG(gs_op0_t += wall_clock64() - gs_o0; gs_op0_n++;)
        continue;
      }
      empty_levels &= ~both;
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
      DIAG(if (__ballot(general)) gs_merges++;)
      if (general) lout = merge_any(rule, evD, evM, a0, la, lb);
      lens[(level - 2) * kBlock + tid] = (uint16_t)lout;
      sp = a0 + lout;
      level--;
      DIAG(gs_op_t += wall_clock64() - gs_o0; gs_op_n++;)
    }
  }
  DIAG(if ((tid & 63) == 0) {
         atomicAdd(&g_general_stats[6], gs_leaf_t << 32 | gs_leaf_n); atomicAdd(&g_general_stats[7], gs_op_t << 32 | gs_op_n);
         atomicAdd(&g_general_stats[8], gs_op0_t << 32 | gs_op0_n);
         const unsigned long long now = wall_clock64();
         atomicAdd(&g_general_stats[0], 1ull); atomicAdd(&g_general_stats[1], (unsigned long long)__popcll(live));
         atomicAdd(&g_general_stats[2], gs_words); atomicAdd(&g_general_stats[3], gs_merges);
         atomicAdd(&g_general_stats[4], gs_t1 - gs_t0); atomicAdd(&g_general_stats[5], now - gs_t1);
       })
  const uint32_t n = lens[tid] & 0x7FFFu;
  if (active && n) {
    h.any = true;
    h.dist = evD[0];
    h.meta = evM[0];
  }
  return h;
}

// Measured on chess 4K depth 5 (Mrays/s): none 3384, TSTOP 3387, TSTOP+TMAX_CULL 3534, TMAX_CULL 3493;
// an all-lanes-done test in front of every record cost more than it saved (3265).
#ifndef GMLRT_TSTOP
#define GMLRT_TSTOP 1
#endif
#ifndef GMLRT_TMAX_CULL
#define GMLRT_TMAX_CULL 1
#endif
#ifndef GMLRT_TOP_EXIT
#define GMLRT_TOP_EXIT 0
#endif
#ifndef GMLRT_LEAF_LOOK
#define GMLRT_LEAF_LOOK 1
#endif
[[maybe_unused]] constexpr int kLeafDbg = 5;  // diagnostic build: counters per leaf
#include "gmlrt_traverse.hpp"

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

// What a lane carries from one traversal to the next.  Kept to 60 bytes so that it
// travels to and from the out-of-line code in registers (a larger struct would be
// passed through the stack, i.e. scratch memory).
struct LaneState {
  V3 o, d;       // ray to traverse
  float t_stop;  // the traversal may stop at the first hit at or before this distance (-inf: never)
  uint32_t pix;  // output index (row-in-partition * w + x)
  uint32_t st;   // bits 1..0 RayKind, 12..2 bounce level, 23..13 light being tested
};
// what one call of after_traversal did: spawned a shadow ray / a bounce / shaded a hit
constexpr uint32_t kStShadow = 1u << 24, kStReflect = 1u << 25, kStHit = 1u << 26;
__device__ inline int st_kind(uint32_t st) { return (int)(st & 3u); }

#ifndef GMLRT_AFTER_INLINE
#define GMLRT_AFTER_INLINE __noinline__
#endif
#include "gmlrt_shade.hpp"

// Idle lanes are given new pixels only once at least this many of the wave's lanes are
// idle: rays that start together stay in step (camera rays of one tile, then their
// shadow rays, then their bounces), which is what lets a whole wave skip a bounded subtree.
#ifndef GMLRT_REFILL_MIN
#define GMLRT_REFILL_MIN 64
#endif
#ifndef GMLRT_WAVES_PER_SIMD
#define GMLRT_WAVES_PER_SIMD 4  // measured on chess 4K depth 5: 3 -> 1796, 4 -> 1955, 5 -> 1655, 6 -> 1794 Mrays/s
#endif
// kLdsCtx: the block's LDS has room for the hot fields of the shading context (see gmlrt_shade.hpp);
// a kernel per case, so that the loop below holds a single call site of after_traversal.
// kMode: 0 the ordinary kernel; 1 the launch has tiles that the waves of a block render together; 2 the rays of a block are
// regrouped before every traversal (both need the LDS context region); 3 one launch per bounce level, a bounce ray
// is queued for the next launch instead of staying in its lane.  Kernels of their own, so that the ordinary one
// carries none of that.
template <bool kLdsCtx, int kMode>
__global__ __launch_bounds__(kBlock, GMLRT_WAVES_PER_SIMD) void render_kernel(
    const gmlrt_leaf* __restrict__ leaves, const uint32_t* __restrict__ nodes, const gmlrt_light* __restrict__ lights,
    const uint64_t* __restrict__ code, const uint64_t* __restrict__ consts, const gmlrt_bound* __restrict__ bounds,
    const uint8_t* __restrict__ prog, const ShadeArgs* __restrict__ shade_args, const KArgs A) {
  constexpr bool kShared = kMode == 1, kRegroup = kMode == 2, kLevels = kMode == 3;
  // (waves that share rays get the owner's new camera rays only at a round that began with every lane idle)
  static_assert(!kShared || GMLRT_REFILL_MIN == kWave, "sharing tiles among the waves of a block assumes whole-wave refills");
  extern __shared__ __align__(16) uint16_t lens[];
  const int tid = threadIdx.x;
  const int lane = tid & (kWave - 1);
  const size_t wave = (size_t)blockIdx.x * (kBlock / kWave) + (size_t)(tid / kWave);
  double* __restrict__ evD = A.ev_dist + wave * A.ev_cap * kWave + lane;
  uint32_t* __restrict__ evM = A.ev_meta + wave * A.ev_cap * kWave + lane;

  LaneState S;
  S.o = {0, 0, 0};
  S.d = {0, 0, 1};
  S.t_stop = -__builtin_inff();
  S.pix = 0;
  S.st = kIdle | (kRegroup ? (uint32_t)tid << 24 : 0u);  // (regrouping: the context slot the ray owns travels in bits 31..24)
  // ray counts of the whole wave, kept as scalars (a per-lane count would be one more vector register to
  // save around every call of the per-ray code)
  uint32_t n_primary = 0, n_shadow = 0, n_reflect = 0, n_hits = 0;
  // Pixels are handed out in 8x8 tiles.  A wave owns one tile at a time and feeds
  // its idle lanes from it, so the rays a wave traverses together stay close on
  // screen (that is what lets a whole wave skip a bounded subtree); a new tile comes
  // from a global counter when the current one is used up.  All three are wave-uniform.
  bool more = true;        // the global counter still has tiles
  uint32_t tile_base = 0;  // next slot of the wave's tile
  uint32_t tile_left = 0;  // slots of it not handed out yet
  uint32_t batch_pos = 0, batch_end = 0;  // hand-outs claimed from the global counter and not started yet
  uint32_t claim_g = blockIdx.x % kClaimCounters, claim_tries = 0;  // counter in use, counters found empty
  // Level-by-level rendering, levels above 0: the "tiles" are runs of 64 queued rays in sorted order, and how many
  // there are is only known on the device.  q_pos .. q_end: room left in the wave's chunk of the outgoing queue.
  [[maybe_unused]] uint32_t n_in = 0, q_pos = 0, q_end = 0;
  uint32_t n_tiles_l = A.n_tiles, n_batches_l = A.n_batches;
  [[maybe_unused]] uint32_t tile_batch_l = kTileBatch;
  if constexpr (kLevels) {
    if (A.level > 0) {
      n_in = *A.q_in_count;
      n_tiles_l = (n_in + 63u) / 64u;
      // (few runs of rays: one per claim, so that no wave ends the launch with several of them in a row)
      tile_batch_l = n_tiles_l > 16u * gridDim.x * (uint32_t)(kBlock / kWave) ? kTileBatch : 1u;
      n_batches_l = (n_tiles_l + tile_batch_l - 1u) / tile_batch_l;
    }
  }
  // Tiles differ a lot in cost (sky: one traversal; a piece between mirrors: a dozen heavy ones) and
  // a wave follows its tile through every bounce, so the launch ends when the last expensive tile
  // does.  Each wave therefore notes how long its tiles took; the host sorts that into the hand-out
  // order (dearest first) of later launches with the same geometry.
  uint32_t cur_tile = 0xFFFFFFFFu;
  unsigned long long tile_t0 = 0;
  // sampled hand-out: the wave's tile is a sample; the class it publishes its group in next; which hand-outs of the
  // claimed batch [batch_base, batch_end) this wave won; dear tiles taken from a list and not started yet
  [[maybe_unused]] bool cur_is_sample = false;
  [[maybe_unused]] uint32_t sm_class = 0, sm_next = 0, batch_ok = 0, batch_base = 0, dq_n = 0, dq0 = 0, dq1 = 0, dq2 = 0, dq3 = 0;
  DIAG(unsigned long long dbg_trav0 = 0, dbg_leaf0 = 0, dbg_merge0 = 0, dbg_redo0 = 0;)
  Diag dg = {};
  DIAG(dg.leaf_dbg = A.leaf_dbg;)
  DIAG(unsigned long long t0 = __builtin_readcyclecounter(); const unsigned long long w_start = wall_clock64();
       unsigned long long w_tiles = 0;)

  // slot `in` of an 8x8 tile -> pixel of this partition's packed rows -> camera ray in S (the lane stays idle on a padding slot).
  // (Text, not a lambda: as a lambda the ordinary kernel came out with three more spilled registers.)
#define GMLRT_CAMERA_RAY(tile_, in_)                                                                               \
  do {                                                                                                             \
    const uint32_t ct_ = (tile_), ci_ = (in_);                                                                     \
    const uint32_t x = (ct_ % A.tiles_x) * kTileW + (ci_ % kTileW);                                                \
    const uint32_t yl = (ct_ / A.tiles_x) * kTileH + (ci_ / kTileW);                                               \
    if (x < (uint32_t)A.w && yl < (uint32_t)A.rows_part) {                                                         \
      const uint32_t yp = yl + (uint32_t)A.band_row0; /* packed row of the partition */                            \
      uint32_t y = yp;                                                                                             \
      if (A.nparts > 1) {                                                                                          \
        const uint32_t k = yp / (uint32_t)A.tile_rows;                                                             \
        y = ((uint32_t)A.part + k * (uint32_t)A.nparts) * (uint32_t)A.tile_rows + yp % (uint32_t)A.tile_rows;      \
      }                                                                                                            \
      /* raytracer.rs:138-142 */                                                                                   \
      const double xf = (double)x, yf = (double)y;                                                                 \
      V3 dir = {A.c_x + (xf + 0.5) * A.pw, A.c_y - (yf + 0.5) * A.pw, 1.0};                                        \
      S.o = {0.0, 0.0, -1.0};                                                                                      \
      S.d = vnormalize(dir);                                                                                       \
      S.t_stop = -__builtin_inff();                                                                                \
      S.st = kView | (kRegroup ? S.st & 0xFF000000u                                                                \
                               : (A.n_axes | A.n_cone_forms) ? ray_special_bit(shade_args, S.d.x, S.d.y, S.d.z) : 0u); /* level 0 */           \
      S.pix = yl * (uint32_t)A.w + x;                                                                              \
    }                                                                                                              \
  } while (0)
  // Shared tiles: the four waves of the block hold the SAME 64 rays, each walks its own slice of the frontier,
  // wave 0 keeps the nearest of the four candidates per ray, runs the per-ray code and hands the next rays to the
  // other three.  A tile takes about a third of the time that way; the hand-out gives such treatment to the few
  // tiles whose cost would otherwise set the length of the launch.  Everything that decides control flow here is
  // the same in all four waves (the list position, and ray states that are copies of wave 0's), so the block's
  // barriers are met by all of them.  The exchange goes through the hot-context slots of the lanes of waves 1-3,
  // which hold nothing while those waves run no per-ray code: field f of thread t is xq[f * 256 + t].
  //
  // The same protocol serves the END of a launch without any list (A.tail_lds_off != 0): a wave that finds nothing
  // left to hand out counts itself in (an LDS counter) and waits at a block barrier; a wave that still traces and
  // sees the other three counted in joins them at that barrier and becomes the owner: from then on the four trace
  // the owner's rays together, its remaining tiles included.  (All four counted in: everybody leaves.)  The last
  // tiles of a launch, which a single wave would trail behind an idle device, take a third of the time.
  const uint32_t wib = (uint32_t)__builtin_amdgcn_readfirstlane(tid / kWave);
  uint32_t shared_next = blockIdx.x;
  bool shared = false;   // this wave traces rays that all four waves of the block hold
  bool tail = false;     // ... because the block has gone over to sharing its last busy wave's work
  bool owner = false;    // this wave runs the per-ray code and hands out the rays
  uint32_t owner_wib = 0;
  uint32_t pc0 = 0, shared_tile = 0;
  DIAGS(unsigned long long ds_phase[4] = {0, 0, 0, 0}, ds_t = 0;)
  [[maybe_unused]] double* const xq = (double*)((uint8_t*)lens + A.vm_lds_off);
  // [0] waves that have run out of work, [1] the owner's number
  [[maybe_unused]] volatile uint32_t* const tailc = (volatile uint32_t*)((uint8_t*)lens + A.tail_lds_off);
  // sampled hand-out: what the block last saw of the samples' statistics (finished samples and their ticks, one word)
  // (read and written as an LDS word by name: through a plain volatile pointer the read became a flat load with a wait
  // for everything the wave has in flight to memory, stores included, which cost a wave with a sample 15 us a look)
  typedef GMLRT_AS(3) unsigned long long* LdsWord;
  [[maybe_unused]] const LdsWord sm_seen = (LdsWord)(uintptr_t)((uint8_t*)lens + A.sm_lds_off);
  if constexpr (kShared) {
    if (A.tail_lds_off || A.sm_claimed) {
      if (tid == 0) {
        if (A.tail_lds_off) { tailc[0] = 0; tailc[1] = 0; }
        if (A.sm_claimed) for (uint32_t c = 0; c < kSmClasses; c++) __hip_atomic_store(sm_seen + c, 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      }
      __syncthreads();
    }
  }
// the owner's rays to the other three, through fields 2..9 of the lanes of the wave after the owner (one of the three:
// its context slots hold nothing); a block barrier between the writing and the reading
#define GMLRT_HAND_OUT_RAYS()                                                                               \
  do {                                                                                                      \
    double* const xs = xq + 2 * kBlock + ((owner_wib + 1u) & 3u) * kWave + lane;                            \
    if (owner) {                                                                                            \
      xs[0] = S.o.x; xs[kBlock] = S.o.y; xs[2 * kBlock] = S.o.z;                                            \
      xs[3 * kBlock] = S.d.x; xs[4 * kBlock] = S.d.y; xs[5 * kBlock] = S.d.z;                               \
      ((uint64_t*)xs)[6 * kBlock] = (uint64_t)__float_as_uint(S.t_stop) | ((uint64_t)S.pix << 32);          \
      ((uint64_t*)xs)[7 * kBlock] = (uint64_t)S.st;                                                         \
    }                                                                                                       \
    __syncthreads();                                                                                        \
    if (!owner) {                                                                                           \
      S.o = {xs[0], xs[kBlock], xs[2 * kBlock]};                                                            \
      S.d = {xs[3 * kBlock], xs[4 * kBlock], xs[5 * kBlock]};                                               \
      const uint64_t tp = ((uint64_t*)xs)[6 * kBlock];                                                      \
      S.t_stop = __uint_as_float((uint32_t)tp);                                                             \
      S.pix = (uint32_t)(tp >> 32);                                                                         \
      S.st = (uint32_t)((uint64_t*)xs)[7 * kBlock];                                                         \
    }                                                                                                       \
  } while (0)

  [[maybe_unused]] uint32_t n_idle = 0;  // regrouping: idle lanes of the whole block
  for (;;) {
    if constexpr (kRegroup) {
      // ---- the block's 256 rays, sorted into its four waves by what they are: camera rays, their shadow rays,
      // rays after a bounce by the quadrant of their direction, shadow rays after a bounce; idle lanes last, so
      // that a wave either is full of rays of a kind or has nothing to do (and takes a new tile when all its
      // lanes are idle).  A counting sort that keeps the order within a kind: rays of one tile stay neighbours.
      // A ray takes its state along (through device memory: 64 bytes a ray) and the number of its context slot.
      constexpr uint32_t kKeys = 8;
      const uint32_t knd = (uint32_t)st_kind(S.st), lvl = (S.st >> 2) & 0x7FFu;
      const uint32_t key = knd == kIdle ? kKeys - 1 : (A.regroup_mode & 3u) == 1u ? 0u : lvl == 0 ? (knd == kView ? 0u : 1u)
                           : (A.regroup_mode & 3u) == 2u ? (knd == kView ? 2u : 3u)
                           : knd == kShadow ? 6u : 2u + (S.d.x < 0.0 ? 1u : 0u) + (S.d.z < 0.0 ? 2u : 0u);
      // counts of this wave, in the operand-stack slots of its own lanes (dead between its traversals)
      uint32_t rank = 0;
#pragma unroll
      for (uint32_t k = 0; k < kKeys; k++) {
        const unsigned long long b = __ballot(key == k);
        if (key == k) rank = (uint32_t)__popcll(b & ((1ull << lane) - 1ull));
        if (lane == 0) lens[wib * kWave + k] = (uint16_t)__popcll(b);
      }
      __syncthreads();
      uint32_t pos = rank;
      n_idle = 0;
#pragma unroll
      for (uint32_t w = 0; w < (uint32_t)(kBlock / kWave); w++) {
#pragma unroll
        for (uint32_t k = 0; k < kKeys; k++) {
          const uint32_t c = lens[w * kWave + k];
          if (k < key || (k == key && w < wib)) pos += c;
          if (k == kKeys - 1) n_idle += c;
        }
      }
      uint64_t* const xg = A.xchg + (size_t)blockIdx.x * kBlock * 8;
      {
        uint64_t* const xo = xg + pos;
        xo[0] = __builtin_bit_cast(uint64_t, S.o.x); xo[kBlock] = __builtin_bit_cast(uint64_t, S.o.y); xo[2 * kBlock] = __builtin_bit_cast(uint64_t, S.o.z);
        xo[3 * kBlock] = __builtin_bit_cast(uint64_t, S.d.x); xo[4 * kBlock] = __builtin_bit_cast(uint64_t, S.d.y); xo[5 * kBlock] = __builtin_bit_cast(uint64_t, S.d.z);
        xo[6 * kBlock] = (uint64_t)__float_as_uint(S.t_stop) | ((uint64_t)S.pix << 32);
        xo[7 * kBlock] = (uint64_t)S.st;
      }
      __syncthreads();  // (the counts are not read after this barrier, the traversal may overwrite them)
      {
        const uint64_t* const xi = xg + tid;
        S.o = {__builtin_bit_cast(double, xi[0]), __builtin_bit_cast(double, xi[kBlock]), __builtin_bit_cast(double, xi[2 * kBlock])};
        S.d = {__builtin_bit_cast(double, xi[3 * kBlock]), __builtin_bit_cast(double, xi[4 * kBlock]), __builtin_bit_cast(double, xi[5 * kBlock])};
        const uint64_t tp = xi[6 * kBlock];
        S.t_stop = __uint_as_float((uint32_t)tp);
        S.pix = (uint32_t)(tp >> 32);
        S.st = (uint32_t)xi[7 * kBlock];
      }
      // (the next round's states are written after its first barrier, which no wave passes before every wave has read these)
    }
    if constexpr (kShared) {
      // a sample that has outlasted the next multiple of the average publishes its group (see kSmClasses); it looks when
      // it has run 64 us and then every time it has run an eighth as long again
      if (A.sm_claimed && cur_is_sample && sm_class < kSmClasses && !shared && wall_clock64() - tile_t0 > (unsigned long long)sm_next) {
        const unsigned long long now = wall_clock64();
        sm_next += sm_next >> 3;
        // (the class's milestone as a wave of this block last saw it while it looked for work: LDS, no device-wide load here)
        const unsigned long long v = __hip_atomic_load(sm_seen + sm_class, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        const unsigned long long fin_n = v >> 40, fin_sum = v & ((1ull << 40) - 1ull);
        // class c: still running when that share of all samples has finished (v != 0), and for several times as long as those took on average
        const uint32_t is_dear = v != 0ull && (now - tile_t0) * fin_n > (unsigned long long)A.sm_factor[sm_class] * fin_sum ? 1u : 0u;
        if (__builtin_amdgcn_readfirstlane((int)is_dear)) {
          const uint32_t tiles_y = n_tiles_l / A.tiles_x;
          const uint32_t gx = (cur_tile % A.tiles_x) / kSmGroup, gy = (cur_tile / A.tiles_x) / kSmGroup;
          const uint32_t tx = gx * kSmGroup + ((uint32_t)lane & 3u), ty = gy * kSmGroup + (((uint32_t)lane >> 2) & 3u);
          const uint32_t tl = ty * A.tiles_x + tx;
          bool valid = lane < 16 && tx < A.tiles_x && ty < tiles_y;
          if (valid) valid = __hip_atomic_load(A.sm_claimed + tl, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == kTileFree;
          const unsigned long long vm = __ballot(valid);
          if (vm) {
            uint32_t at = 0;
            if (lane == 0) at = atomicAdd(A.sm_cnt + (2 * sm_class) * kSmLine, (uint32_t)__popcll(vm));
            at = (uint32_t)__builtin_amdgcn_readfirstlane((int)at);
            if (valid)
              __hip_atomic_store(A.sm_list + (size_t)sm_class * n_tiles_l + at + (uint32_t)__popcll(vm & ((1ull << lane) - 1ull)), tl,
                                 __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          }
          sm_class++;
          sm_next = (uint32_t)(now - tile_t0);  // (the next class may be due already: look again on the next round)
        }
      }
    }
    // ---- hand new pixels to idle lanes
    unsigned long long idle = __ballot(st_kind(S.st) == kIdle);
    [[maybe_unused]] const unsigned long long idle_at_top = idle;
    if constexpr (kShared) {
      if (!tail && (shared || shared_next < A.n_tiles_shared) && idle == ~0ull) {
        if (shared && A.tile_cost && tid == 0) A.tile_cost[shared_tile] = (uint32_t)(wall_clock64() - tile_t0);
        shared = shared_next < A.n_tiles_shared;
        pc0 = 0;
        owner = false;
        if (shared) {
          shared_tile = A.tiles_shared[shared_next];
          if (A.tile_cost) tile_t0 = wall_clock64();
          GMLRT_CAMERA_RAY(shared_tile, (uint32_t)lane);
          shared_next += gridDim.x;
          pc0 = A.slice_off[wib];
          owner = wib == 0;
          owner_wib = 0;
          idle = __ballot(st_kind(S.st) == kIdle);
          if (owner) n_primary += (uint32_t)__popcll(~idle);
        }
      }
      // the other three waves of the block have run out of work and wait: share what this one still has
      if (A.tail_lds_off && !shared && idle != ~0ull && (uint32_t)__builtin_amdgcn_readfirstlane((int)tailc[0]) == 3u) {
        if (lane == 0) tailc[1] = wib;
        __syncthreads();  // the barrier the other three wait at
        tail = shared = owner = true;
        owner_wib = wib;
        pc0 = A.slice_off[0];
        GMLRT_HAND_OUT_RAYS();
      }
    }
    while ((!shared || (tail && owner)) && (uint32_t)__popcll(idle) >= (uint32_t)GMLRT_REFILL_MIN && (tile_left || more) &&
           (!kRegroup || !(A.regroup_mode & 4u) || n_idle == (uint32_t)kBlock)) {  // (regrouping, bit 2: new tiles only when the whole block is idle)
      bool sm_done = false;  // the sampled hand-out below has found the wave's next tile (or that there is none)
      if constexpr (kShared) {
        if (tile_left == 0 && A.sm_claimed) {
          sm_done = true;
          const unsigned long long now = wall_clock64();
          // (one word: finished samples in bits 63..40, the ticks they took in the bits below)
          if (cur_is_sample && lane == 0) {
            const unsigned long long add = (1ull << 40) | ((now - tile_t0) & ((1ull << 40) - 1ull));
            const unsigned long long after = atomicAdd((unsigned long long*)(A.sm_cnt + kSmStatWord), add) + add;
            // the one wave whose sample completes a class's share leaves the word as it is now for everybody to read
#pragma unroll
            for (uint32_t c = 0; c < kSmClasses; c++)
              if ((after >> 40) == ((unsigned long long)A.sm_share[c] * A.sm_groups + 255ull) / 256ull)
                __hip_atomic_store((unsigned long long*)(A.sm_cnt + kSmMileWord) + c, after, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          }
          unsigned long long stat_seen[kSmClasses] = {};
          cur_is_sample = false;
          sm_class = 0;
          sm_next = 6400u;
          // hand-out i: below sm_groups the sample of a group (its second tile in both directions, or the last one of a
          // narrower group), then the plain stride over all tiles; the samples come up there again and are found taken
#define GMLRT_SM_TILE(i_, out_)                                                                                       \
          do {                                                                                                        \
            const uint32_t hi_ = (i_);                                                                                \
            if (hi_ < A.sm_groups) {                                                                                  \
              const uint32_t g_ = (uint32_t)(((uint64_t)hi_ * A.sm_stride) % A.sm_groups);                            \
              const uint32_t sx_ = (g_ % A.sm_groups_x) * kSmGroup + 1u, sy_ = (g_ / A.sm_groups_x) * kSmGroup + 1u;  \
              const uint32_t ty_n_ = n_tiles_l / A.tiles_x;                                                           \
              (out_) = (sy_ < ty_n_ ? sy_ : ty_n_ - 1u) * A.tiles_x + (sx_ < A.tiles_x ? sx_ : A.tiles_x - 1u);       \
            } else {                                                                                                  \
              (out_) = (uint32_t)(((uint64_t)(hi_ - A.sm_groups) * A.stride) % n_tiles_l);                            \
            }                                                                                                         \
          } while (0)
          uint32_t found = kTileFree;
          bool none_left = false;
          for (;;) {
            while (batch_pos < batch_end && !((batch_ok >> (batch_pos - batch_base)) & 1u)) batch_pos++;
            if (batch_pos < batch_end) {
              GMLRT_SM_TILE(batch_pos, found);
              cur_is_sample = batch_pos < A.sm_groups;
              batch_pos++;
              break;
            }
            // dear tiles first: those taken earlier, then a look into the lists
            if (dq_n) {
              found = dq0; dq0 = dq1; dq1 = dq2; dq2 = dq3;
              dq_n--;
              break;
            }
            uint32_t run = kTileFree;
            if (lane == 0) {
              run = sm_take_dear(A, stat_seen);
#pragma unroll
              for (uint32_t c = 0; c < kSmClasses; c++)  // (for the waves of the block that hold a sample)
                __hip_atomic_store(sm_seen + c, stat_seen[c], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            }
            run = (uint32_t)__builtin_amdgcn_readfirstlane((int)run);
            if (run != kTileFree) {
              // the entries of the run, one lane each; an entry not written yet, or a tile that has been taken since it
              // was published, drops out
              const uint32_t rc = run >> 28, rn = (run >> 24) & 15u, rt = run & 0xFFFFFFu;
              uint32_t mine = kTileFree;
              bool got = false;
              if ((uint32_t)lane < rn) {
                mine = __hip_atomic_load(A.sm_list + (size_t)rc * n_tiles_l + rt + (uint32_t)lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (mine != kTileFree) got = atomicExch(A.sm_claimed + mine, 0u) == kTileFree;
              }
              unsigned long long gm = __ballot(got);
              dq_n = (uint32_t)__popcll(gm);
              dq0 = dq1 = dq2 = dq3 = kTileFree;
              if (gm) { dq0 = (uint32_t)__builtin_amdgcn_readlane((int)mine, __builtin_ctzll(gm)); gm &= gm - 1ull; }
              if (gm) { dq1 = (uint32_t)__builtin_amdgcn_readlane((int)mine, __builtin_ctzll(gm)); gm &= gm - 1ull; }
              if (gm) { dq2 = (uint32_t)__builtin_amdgcn_readlane((int)mine, __builtin_ctzll(gm)); gm &= gm - 1ull; }
              if (gm) { dq3 = (uint32_t)__builtin_amdgcn_readlane((int)mine, __builtin_ctzll(gm)); }
              continue;  // (an empty run: look again, the lists may hold more)
            }
            // then the next batch of the plain order: batches below sm_groups are one sample each
            uint32_t b = n_batches_l;
            while (claim_tries < kClaimCounters) {
              uint32_t k = 0;
              if (lane == 0) k = atomicAdd((unsigned int*)((uint8_t*)A.next_batch + (size_t)claim_g * kClaimStride), 1u);
              k = (uint32_t)__builtin_amdgcn_readfirstlane((int)k);
              const uint64_t cand = (uint64_t)k * kClaimCounters + claim_g;
              if (cand < n_batches_l) { b = (uint32_t)cand; break; }
              claim_g = (claim_g + 1u) % kClaimCounters;
              claim_tries++;
            }
            if (b >= n_batches_l) { none_left = true; break; }
            if (b < A.sm_groups) {
              batch_pos = b;
              batch_end = b + 1u;
            } else {
              batch_pos = A.sm_groups + (b - A.sm_groups) * kTileBatch;
              const uint32_t top = A.sm_groups + n_tiles_l;
              batch_end = batch_pos + kTileBatch < top ? batch_pos + kTileBatch : top;
            }
            batch_base = batch_pos;
            // the tiles of the batch, one lane each: which of them are still free
            uint32_t was = 0u;
            if ((uint32_t)lane < batch_end - batch_pos) {
              uint32_t bt;
              GMLRT_SM_TILE(batch_pos + (uint32_t)lane, bt);
              was = atomicExch(A.sm_claimed + bt, 0u);
            }
            batch_ok = (uint32_t)__ballot(was == kTileFree);
          }
#undef GMLRT_SM_TILE
          cur_tile = 0xFFFFFFFFu;
          if (none_left) { more = false; break; }
          cur_tile = found;
          tile_t0 = wall_clock64();  // (after the search: a sample's time is its own)
#ifdef GMLRT_SM_TIMING
          if (lane == 0) { atomicAdd((unsigned long long*)(A.sm_cnt + kSmStatWord) + 1, (unsigned long long)tile_t0 - now); atomicAdd((unsigned long long*)(A.sm_cnt + kSmStatWord) + 2, 1ull); }
#endif
          tile_base = 0;
          tile_left = 64u;

        }
      }
      if (tile_left == 0 && !sm_done) {
        bool none = false;
        if (batch_pos == batch_end) {
          // One atomic on the global counter claims a batch of tiles, not a tile: all waves of the
          // device add to the same address, which the memory system serves one after the other
          // (about 14 ns each, measured); per tile that alone took 1.8 ms of a 4K frame, the
          // whole frame time of a simple scene.
          // counter g hands out batches g, g + 8, g + 16, ...; a wave starts with the counter of its block
          // (blocks go to the chip's eight dies in turn, so a die mostly keeps to one counter) and moves
          // on to the next when one has run out; all eight empty: nothing left
          uint32_t b = n_batches_l;
          while (claim_tries < kClaimCounters) {
            uint32_t k = 0;
            if constexpr (kLevels) {
              // once one counter has run out, a look before the atomic: the others have mostly run out too, and are
              // passed by without one (at the end of a launch every wave finds all of them so, and atomics on one
              // address are served one at a time; a stale value read here can only be too small, which costs the atomic
              // it would have saved)
              const uint32_t seen = claim_tries == 0 ? 0u : __hip_atomic_load((unsigned int*)((uint8_t*)A.next_batch + (size_t)claim_g * kClaimStride),
                                                                              __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
              if (claim_tries != 0 && (uint64_t)seen * kClaimCounters + claim_g >= n_batches_l) {
                claim_g = (claim_g + 1u) % kClaimCounters;
                claim_tries++;
                continue;
              }
            }
            if (lane == 0) k = atomicAdd((unsigned int*)((uint8_t*)A.next_batch + (size_t)claim_g * kClaimStride), 1u);
            k = (uint32_t)__shfl((int)k, 0);
            k = (uint32_t)__builtin_amdgcn_readfirstlane((int)k);
            const uint64_t cand = (uint64_t)k * kClaimCounters + claim_g;
            if (cand < n_batches_l) { b = (uint32_t)cand; break; }
            claim_g = (claim_g + 1u) % kClaimCounters;
            claim_tries++;
          }
          if (b >= n_batches_l) {
            none = true;
          } else if (A.batch_start) {
            batch_pos = A.batch_start[b];
            batch_end = A.batch_start[b + 1];
          } else {
            batch_pos = b * (kLevels ? tile_batch_l : kTileBatch);
            batch_end = batch_pos + (kLevels ? tile_batch_l : kTileBatch) < n_tiles_l ? batch_pos + (kLevels ? tile_batch_l : kTileBatch) : n_tiles_l;
          }
        }
        const uint32_t t = none ? 0xFFFFFFFFu : batch_pos * 64u;
        const unsigned long long now = A.tile_cost ? wall_clock64() : 0ull;
        if (A.tile_cost && cur_tile != 0xFFFFFFFFu && lane == 0) A.tile_cost[cur_tile] = (uint32_t)(now - tile_t0);
        DIAG(if (cur_tile != 0xFFFFFFFFu && lane == 0 && A.tile_dbg) {
               A.tile_dbg[cur_tile * 4 + 0] = dg.n_trav - dbg_trav0; A.tile_dbg[cur_tile * 4 + 1] = dg.n_leaf - dbg_leaf0;
               A.tile_dbg[cur_tile * 4 + 2] = dg.n_merge - dbg_merge0; A.tile_dbg[cur_tile * 4 + 3] = dg.n_redo - dbg_redo0; }
             dbg_trav0 = dg.n_trav; dbg_leaf0 = dg.n_leaf; dbg_merge0 = dg.n_merge; dbg_redo0 = dg.n_redo;)
        cur_tile = 0xFFFFFFFFu;
        if (none) { more = false; break; }
        batch_pos++;
        tile_base = t;
        tile_left = 64u;
        cur_tile = A.tile_order ? A.tile_order[t >> 6] : (uint32_t)(((uint64_t)(t >> 6) * A.stride) % n_tiles_l);
        tile_t0 = now;
        DIAG(w_tiles++;)
      }
      const uint32_t want = (uint32_t)__popcll(idle);
      const uint32_t take = want < tile_left ? want : tile_left;
      const uint32_t rank = (uint32_t)__popcll(idle & ((1ull << lane) - 1ull));
      if (st_kind(S.st) == kIdle && rank < take) {
        if (kLevels && A.level > 0) {
          // the ray queued at this place of the sorted order (past the end of the table: the lane stays idle)
          const uint32_t qi = cur_tile * 64u + ((tile_base + rank) & 63u);
          if (qi < n_in) {
            typedef double d2 __attribute__((ext_vector_type(2)));
            const d2* __restrict__ r = (const d2*)(A.q_in + (size_t)A.q_sorted[qi] * kRayWords);
            const d2 r0 = r[0], r1 = r[1], r2 = r[2], r3 = r[3];
            S.o = {r0.x, r0.y, r1.x};
            S.d = {r1.y, r2.x, r2.y};
            S.pix = (uint32_t)__builtin_bit_cast(uint64_t, r3.x);
            S.t_stop = -__builtin_inff();
            S.st = kView | (A.level << 2) | ((A.n_axes | A.n_cone_forms) ? ray_special_bit(shade_args, S.d.x, S.d.y, S.d.z) : 0u);
          }
        } else {
          GMLRT_CAMERA_RAY(cur_tile, (tile_base + rank) & 63u);
        }
      }
      tile_base += take;
      tile_left -= take;
      const unsigned long long idle_now = __ballot(st_kind(S.st) == kIdle);  // lanes that drew a padding slot are still idle
      if (!kLevels || A.level == 0) n_primary += (uint32_t)__popcll(idle & ~idle_now);
      idle = idle_now;
    }
    if constexpr (kRegroup) {
      // the whole block was idle before the hand-out above: did any wave get a tile?  (the same answer in all four)
      if (n_idle == (uint32_t)kBlock && !__syncthreads_or(__ballot(st_kind(S.st) != kIdle) != 0ull)) break;
      if (!__ballot(st_kind(S.st) != kIdle)) continue;  // a wave of idle lanes with no tile to take: on to the next round's barriers
    } else {
    if constexpr (kShared) {
      if (tail && idle_at_top == ~0ull) {
        // the shared rays are all done (the same in all four waves): the owner has looked for another tile of its own
        // above; what it found goes to the others, and if it found nothing everybody leaves.  The hand-out at the end
        // of the round before used the same slots and no barrier lies between its reads and these writes: one here,
        // which all four waves reach (they hold copies of the same ray states).
        __syncthreads();
        GMLRT_HAND_OUT_RAYS();
        if (!__ballot(st_kind(S.st) != kIdle)) break;
      }
    }
    if (!__ballot(st_kind(S.st) != kIdle)) {
      if (shared && !tail) continue;  // a shared tile of padding slots only: on to the next one
      if constexpr (kShared) {
        if (A.tail_lds_off && !tail) {
          // nothing in flight and nothing left to hand out: count this wave in and wait for the rest of the block
          if (lane == 0) atomicAdd((uint32_t*)tailc, 1u);
          __syncthreads();
          if ((uint32_t)__builtin_amdgcn_readfirstlane((int)tailc[0]) == (uint32_t)(kBlock / kWave)) break;  // all four: done
          tail = shared = true;
          owner = false;
          owner_wib = (uint32_t)__builtin_amdgcn_readfirstlane((int)tailc[1]);
          pc0 = A.slice_off[(wib - owner_wib) & 3u];
          GMLRT_HAND_OUT_RAYS();  // (the owner still traces, or it would have counted itself in: some lane is active now)
        } else {
          break;
        }
      } else {
        break;  // nothing in flight and nothing left to hand out
      }
    }
    }
    DIAG(unsigned long long t1 = __builtin_readcyclecounter(); dg.t_refill += t1 - t0; dg.n_trav++;
         dg.lanes += (unsigned long long)__popcll(__ballot(st_kind(S.st) != kIdle));)

    DIAG(const unsigned long long dk_leaf0 = dg.n_leaf;
         // what the wave is tracing: 0 camera rays only, 1 shadow rays of camera hits among them, 2 anything after a bounce
         const int dk = __ballot(st_kind(S.st) != kIdle && ((S.st >> 2) & 0x7FFu) != 0) ? 2 : __ballot(st_kind(S.st) == kShadow) ? 1 : 0;
         dg.kind = dk; dg.k_active[dk] += (unsigned long long)__popcll(__ballot(st_kind(S.st) != kIdle));)
    // ---- one scene traversal for whatever ray each lane holds
    DIAGS(ds_t = __builtin_readcyclecounter();)
    // (a ray in one of the cylinder's special cases takes no part: see cylinder_special_case)
    bool odd_ray;
    if constexpr (kRegroup) odd_ray = st_kind(S.st) != kIdle && ray_special_case(A.axes, A.n_axes, A.n_cone_forms, S.d);
    else odd_ray = st_kind(S.st) != kIdle && (S.st & kStOdd) != 0;
    Hit h = intersect_first(leaves, prog, pc0, A.n_leaves, st_kind(S.st) != kIdle && !odd_ray, S.o, S.d, (double)S.t_stop, evD, evM, lens, dg);
    if (odd_ray) h.redo = true;
    if constexpr (kShared) {
      if (shared) {
        // (a shadow ray's slice stops at its first candidate at or before the light; the nearest of the four is
        // then at or before the light too, which is all the light test asks)
        uint64_t* const xm = (uint64_t*)xq + kBlock;
        DIAGS({ const unsigned long long n_ = __builtin_readcyclecounter(); ds_phase[0] += n_ - ds_t; ds_t = n_; })
        if (!owner) {
          xq[tid] = h.dist;
          xm[tid] = (uint64_t)(h.meta | (h.any ? kHitAny : 0u)) | (h.redo ? 1ull << 32 : 0ull);
          h.redo = false;
        }
        __syncthreads();
        DIAGS({ const unsigned long long n_ = __builtin_readcyclecounter(); ds_phase[1] += n_ - ds_t; ds_t = n_; })
        if (owner) {
          for (uint32_t k = 1; k < (uint32_t)(kBlock / kWave); k++) {
            const uint32_t w = (owner_wib + k) & 3u;
            const double dw = xq[w * kWave + lane];
            const uint64_t mw = xm[w * kWave + lane];
            if (mw >> 32) h.redo = true;
            if ((mw & kHitAny) && (!h.any || dw < h.dist || (dw == h.dist && ((uint32_t)mw & kLeafMask) < (h.meta & kLeafMask)))) {
              h.any = true; h.dist = dw; h.meta = (uint32_t)mw & ~kHitAny;
            }
          }
        }
      }
    }
    if (__ballot(h.redo)) {
      DIAG(const unsigned long long redo_t0 = wall_clock64();)
      // a primitive right under the root's unions gave a span shorter than 1e-10: only the
      // full sweep knows whether it survives (primitives.rs:295-298)
      const Hit g = intersect_general(leaves, nodes, bounds, A.n_nodes, A.n_leaves, h.redo,
                                      odd_ray, S.o, S.d, evD, evM, lens);
      if (h.redo) h = g;
      DIAG(dg.n_redo += 1ull + ((wall_clock64() - redo_t0) << 16);)  // rounds, and their time on the 100 MHz clock
    }
    // ---- then the per-ray bookkeeping, out of line
    DIAG(unsigned long long t2 = __builtin_readcyclecounter(); dg.t_trav += t2 - t1;
         dg.k_trav[dk]++; dg.k_leaf[dk] += dg.n_leaf - dk_leaf0;)
    uint32_t events = 0;  // its address goes to the callee: see the note at after_traversal
    if (st_kind(S.st) != kIdle && (!shared || owner)) {
      const uint32_t hm = h.any ? (h.meta | kHitAny) : 0u;
      S = after_traversal<kLdsCtx, kRegroup, kLevels>(S, h.dist, hm, shade_args, &events);
    }
    if constexpr (kLevels) {
      // ---- a mirror bounce leaves its lane: the ray goes to the queue of the next level, with the key the rays of
      // that level are grouped by, and the lane is idle.  (What trace() keeps across the bounce has been parked per pixel.)
      const bool mine = (events & kStReflect) != 0 && ((S.st >> 2) & 0x7FFu) == A.level_push;
      const unsigned long long mb = __ballot(mine);
      if (mb) {
        const uint32_t n = (uint32_t)__popcll(mb), rank = (uint32_t)__popcll(mb & ((1ull << lane) - 1ull));
        const uint32_t room = q_end - q_pos, first = q_pos;
        uint32_t second = 0;
        bool full = false;
        if (n > room) {
          uint32_t c = 0;
          if (lane == 0) c = atomicAdd(A.q_out_chunks, 1u);
          c = (uint32_t)__builtin_amdgcn_readfirstlane((int)c);
          if (c >= A.q_cap_chunks) {
            // (cannot happen with the capacity the host gives the queue; the rays are dropped and the launch says so)
            full = true;
            if (lane == 0) *A.q_overflow = 1u;
            q_pos = q_end = first + room;
          } else {
            second = c * kQueueChunk;
            q_pos = second + (n - room);
            q_end = second + kQueueChunk;
          }
        } else {
          q_pos += n;
        }
        uint32_t key = 0;
        if (mine && !(full && rank >= room)) {
          const uint32_t at = rank < room ? first + rank : second + (rank - room);
          typedef double d2 __attribute__((ext_vector_type(2)));
          d2* __restrict__ r = (d2*)(A.q_out + (size_t)at * kRayWords);
          r[0] = d2{S.o.x, S.o.y};
          r[1] = d2{S.o.z, S.d.x};
          r[2] = d2{S.d.y, S.d.z};
          r[3] = d2{__builtin_bit_cast(double, (uint64_t)S.pix), 0.0};
          if (A.key_mode) {
            // grid cell of the origin (clamped into the scene's box) and bin of the direction
            const int top = (1 << A.key_bits) - 1;
            const double fx = (S.o.x - A.key_lo[0]) * A.key_inv[0], fy = (S.o.y - A.key_lo[1]) * A.key_inv[1], fz = (S.o.z - A.key_lo[2]) * A.key_inv[2];
            const uint32_t cx = (uint32_t)(fx > 0.0 ? (fx < (double)top ? (int)fx : top) : 0);
            const uint32_t cz = (uint32_t)(fz > 0.0 ? (fz < (double)top ? (int)fz : top) : 0);
            const uint32_t cy = fy >= 1.0 ? 1u : 0u;
            uint32_t cell = 0;
#pragma unroll
            for (uint32_t b = 0; b < kKeyBitsMax; b++) cell |= (((cx >> b) & 1u) << (2 * b)) | (((cz >> b) & 1u) << (2 * b + 1));
            cell |= cy << (2 * A.key_bits);
            const float ax = __builtin_fabsf((float)S.d.x), ay = __builtin_fabsf((float)S.d.y), az = __builtin_fabsf((float)S.d.z);
            const float inv = 4.0f / (ax + ay + az);
            const uint32_t qa = (uint32_t)__builtin_fminf(ax * inv, 3.0f), qb = (uint32_t)__builtin_fminf(az * inv, 3.0f);
            const uint32_t dir = ((S.d.x < 0.0 ? 1u : 0u) | (S.d.z < 0.0 ? 2u : 0u) | (S.d.y < 0.0 ? 4u : 0u)) << 4 | qa << 2 | qb;
            key = A.key_mode == 1 ? (cell << kKeyDirBits) | dir : (dir << (2 * A.key_bits + 1)) | cell;
          }
          A.q_out_keys[at] = key;
        }
        if (mine) S.st = kIdle;
      }
    }
    if constexpr (kShared) {
      if (shared) {
        DIAGS({ const unsigned long long n_ = __builtin_readcyclecounter(); ds_phase[2] += n_ - ds_t; ds_t = n_; })
        GMLRT_HAND_OUT_RAYS();
        DIAGS({ const unsigned long long n_ = __builtin_readcyclecounter(); ds_phase[3] += n_ - ds_t; ds_t = n_; })
        // (the next round's candidates go to fields 0 and 1, and its rays are written after the barrier that follows
        // them, or, when the rays are all done and the owner hands out a new tile at the top of the loop, after the
        // barrier placed there: nothing here is overwritten while a wave may still read it)
      }
    }
    n_shadow += (uint32_t)__popcll(__ballot((events & kStShadow) != 0));
    n_reflect += (uint32_t)__popcll(__ballot((events & kStReflect) != 0));
    n_hits += (uint32_t)__popcll(__ballot((events & kStHit) != 0));
    DIAG(t0 = __builtin_readcyclecounter(); dg.t_after += t0 - t2;)
  }

  if (A.tile_cost && cur_tile != 0xFFFFFFFFu && lane == 0) A.tile_cost[cur_tile] = (uint32_t)(wall_clock64() - tile_t0);
  if constexpr (kLevels) {
    // the slots of the wave's last chunk that no ray went to
    for (uint32_t i = q_pos + (uint32_t)lane; i < q_end; i += (uint32_t)kWave) A.q_out_keys[i] = kKeyInvalid;
  }

  // ---- ray counters of the waves, summed over the block (in the first bytes of the operand-stack lengths, each wave in the slots of
  // its own lanes: the other waves may still be tracing) and stored per block; the host adds the blocks
  // up.  Atomics on four device-wide counters, one set per wave, took 0.24 ms of every launch.
  {
    uint32_t* __restrict__ part = (uint32_t*)lens + (tid / kWave) * (kWave / 2);
    if (lane == 0) { part[0] = n_primary; part[1] = n_shadow; part[2] = n_reflect; part[3] = n_hits; }
    __syncthreads();
    if (tid < 4) {
      unsigned long long sum = 0;
      for (int w = 0; w < kBlock / kWave; w++) sum += ((const uint32_t*)lens)[w * (kWave / 2) + tid];
      // (level by level: the launches of a frame add up; the host cleared the counts before the first)
      if constexpr (kLevels) A.block_counts[(size_t)blockIdx.x * 4 + tid] += sum;
      else A.block_counts[(size_t)blockIdx.x * 4 + tid] = sum;
    }
  }
  DIAGS(if (lane == 0) for (int q = 0; q < 4; q++) atomicAdd(&A.diag[wib * 4 + q], ds_phase[q]);)
  if (lane == 0) {
    DIAG(for (int q = 0; q < 8; q++) { shade_args->diag_waves[wave * kDiagWaveSlots + 12 + q] = dg.c_kind[q]; shade_args->diag_waves[wave * kDiagWaveSlots + 20 + q] = dg.n_kind[q]; })
    DIAG(shade_args->diag_waves[wave * kDiagWaveSlots + 8] = w_start; shade_args->diag_waves[wave * kDiagWaveSlots + 9] = wall_clock64();
         shade_args->diag_waves[wave * kDiagWaveSlots + 10] = w_tiles;)
    DIAG(atomicAdd(&A.diag[0], dg.t_refill); atomicAdd(&A.diag[1], dg.t_trav); atomicAdd(&A.diag[2], dg.t_after);
         atomicAdd(&A.diag[3], dg.n_trav); atomicAdd(&A.diag[4], dg.n_bound); atomicAdd(&A.diag[5], dg.n_skip);
         atomicAdd(&A.diag[6], dg.n_leaf); atomicAdd(&A.diag[7], dg.n_merge); atomicAdd(&A.diag[8], dg.n_take);
         atomicAdd(&A.diag[9], dg.lanes); atomicAdd(&A.diag[16], dg.b_lanes); atomicAdd(&A.diag[17], dg.l_lanes);
         atomicAdd(&A.diag[18], dg.l_hits);
         for (int q = 0; q < 3; q++) atomicAdd(&A.diag[19 + q], dg.k_trav[q]);
         for (int q = 0; q < 2; q++) atomicAdd(&A.diag[22 + q], dg.k_leaf[q]);  // the third is the rest of n_leaf; the table ends at slot 23
         atomicAdd(&A.diag[10], dg.l_nohit);
         for (int q = 0; q < 3; q++) { atomicAdd(&A.diag[24 + q], dg.k_entered[q]); atomicAdd(&A.diag[27 + q], dg.k_blanes[q]); }
         for (int q = 0; q < 2; q++) atomicAdd(&A.diag[30 + q], dg.k_active[q]);
         )
  }
}

// ---- level-by-level rendering: between two launches the queued rays are put in the order of their keys, a counting
// sort of their positions (the ray records stay where they were written) in three steps: rays per key; where each
// key's run starts; every position to its place.  Slots of chunks that were never filled drop out.
// Steps 1 and 3: block g takes slots [g * kSortSegment, (g + 1) * kSortSegment) of the queue and counts them per key
// in LDS (which also numbers the rays of a key within the block); kPlace = false adds the counts to the global
// histogram, kPlace = true takes the block's share of every key's run and writes the positions there.
template <bool kPlace>
__global__ __launch_bounds__(kBlock) void levels_sort_kernel(const uint32_t* __restrict__ keys, const uint32_t* __restrict__ n_chunks,
                                                             uint32_t cap_chunks, uint32_t n_keys, uint32_t* __restrict__ counters,
                                                             uint32_t* __restrict__ sorted) {
  __shared__ uint32_t bins[kKeysMax];
  const uint32_t tid = threadIdx.x;
  const uint32_t chunks = *n_chunks < cap_chunks ? *n_chunks : cap_chunks;
  const uint32_t n_flat = chunks * kQueueChunk;
  constexpr uint32_t kPer = kSortSegment / kBlock;
  for (uint32_t seg = blockIdx.x; seg * kSortSegment < n_flat; seg += gridDim.x) {
    const uint32_t base = seg * kSortSegment;
    for (uint32_t k = tid; k < n_keys; k += kBlock) bins[k] = 0;
    __syncthreads();
    uint32_t key[kPer], rank[kPer];
#pragma unroll
    for (uint32_t j = 0; j < kPer; j++) {
      const uint32_t i = base + j * kBlock + tid;
      key[j] = i < n_flat ? keys[i] : kKeyInvalid;
      rank[j] = key[j] != kKeyInvalid ? atomicAdd(&bins[key[j]], 1u) : 0u;
    }
    __syncthreads();
    for (uint32_t k = tid; k < n_keys; k += kBlock) {
      const uint32_t c = bins[k];
      if (kPlace) bins[k] = c ? atomicAdd(counters + k, c) : 0u;
      else if (c) atomicAdd(counters + k, c);
    }
    __syncthreads();
    if constexpr (kPlace) {
#pragma unroll
      for (uint32_t j = 0; j < kPer; j++)
        if (key[j] != kKeyInvalid) sorted[bins[key[j]] + rank[j]] = base + j * kBlock + tid;
      __syncthreads();
    }
  }
}
// Step 2, one block: where each key's run starts in the sorted table (exclusive prefix sums of the histogram, which
// is cleared for the next launch's rays), and how many rays there are.
constexpr int kScanThreads = 1024;
__global__ __launch_bounds__(kScanThreads) void levels_scan_kernel(uint32_t* __restrict__ hist, uint32_t* __restrict__ cursor,
                                                                    uint32_t n_keys, uint32_t* __restrict__ n_sorted) {
  __shared__ uint32_t part[kScanThreads];
  const uint32_t tid = threadIdx.x;
  const uint32_t per = (n_keys + kScanThreads - 1) / kScanThreads;
  const uint32_t k0 = tid * per < n_keys ? tid * per : n_keys, k1 = k0 + per < n_keys ? k0 + per : n_keys;
  uint32_t sum = 0;
  for (uint32_t k = k0; k < k1; k++) sum += hist[k];
  part[tid] = sum;
  __syncthreads();
  for (uint32_t step = 1; step < (uint32_t)kScanThreads; step <<= 1) {
    const uint32_t add = tid >= step ? part[tid - step] : 0u;
    __syncthreads();
    part[tid] += add;
    __syncthreads();
  }
  uint32_t run = part[tid] - sum;  // rays of all keys before k0
  for (uint32_t k = k0; k < k1; k++) {
    const uint32_t c = hist[k];
    cursor[k] = run;
    hist[k] = 0;
    run += c;
  }
  if (tid == kScanThreads - 1) *n_sorted = part[tid];
}

// Scratch of the entry point that allocates nothing (dev_render with no_alloc): claim counters, launch
// record and per-block ray counts, then the event stacks and the context records of as many lanes as
// fit.  Every device the library runs on holds its own copy of the array, at its own address; host code
// never uses the array's name as an address, only what hipGetSymbolAddress returned for that device.
__device__ __align__(256) uint8_t scratch[kNoAllocScratchBytes];
std::mutex g_scratch_mu;
constexpr int kMaxDevices = 64;
void* g_scratch_ptr[kMaxDevices] = {nullptr};

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

namespace {
// Page-locked staging of one host worker (one per GPU partition of gmlrt_render): a ring of slots that device
// memory is copied into by DMA while the worker copies the previous slot on into the caller's pageable buffer.
// (hipMemcpy into pageable memory does the same inside the runtime, one step after the other: 1.9 ms for the
// 24.9 MB of a 4K frame against 0.45 ms of DMA into page-locked memory and 0.8 ms of memcpy, which overlap here.)
struct HostStage {
  static constexpr int kSlots = 3;
  int device = -1;
  void* slot[kSlots] = {nullptr, nullptr, nullptr};
  size_t slot_bytes = 0;
  hipStream_t stream = nullptr;
  hipEvent_t ev[kSlots] = {nullptr, nullptr, nullptr};
  // device output buffers of the worker, kept between calls
  void* d8 = nullptr;
  void* df = nullptr;
  size_t d8_bytes = 0, df_bytes = 0;
  std::mutex copy_mu;  // one copy at a time through this stage's slots, stream and events
};
constexpr size_t kStageSlotBytes = 4u << 20;
constexpr size_t kCopyThreads = 4;  // stage index = worker * kCopyThreads + thread
std::mutex g_stage_mu;
std::vector<HostStage*> g_stage;  // by worker index; entries are created once and live until dev_release_cached

std::string stage_for(int worker, int device, HostStage** out) {
  HostStage* st = nullptr;
  {
    std::lock_guard<std::mutex> lk(g_stage_mu);
    if ((size_t)worker >= g_stage.size()) g_stage.resize((size_t)worker + 1, nullptr);
    if (!g_stage[(size_t)worker]) g_stage[(size_t)worker] = new HostStage();
    st = g_stage[(size_t)worker];
  }
  if (st->device != device) {
    // (a worker index keeps its device for the life of the process in practice: partition g runs on GPU g mod n)
    if (st->device >= 0 && hipSetDevice(st->device) == hipSuccess) {
      for (int i = 0; i < HostStage::kSlots; i++) { if (st->slot[i]) (void)hipHostFree(st->slot[i]); if (st->ev[i]) (void)hipEventDestroy(st->ev[i]); }
      if (st->stream) (void)hipStreamDestroy(st->stream);
      (void)hipFree(st->d8);
      (void)hipFree(st->df);
      st->device = -1;
      for (int i = 0; i < HostStage::kSlots; i++) { st->slot[i] = nullptr; st->ev[i] = nullptr; }
      st->stream = nullptr;
      st->d8 = st->df = nullptr;
      st->d8_bytes = st->df_bytes = 0;
    }
    HIP_TRY(hipSetDevice(device));
    // (the stage names its device only once all of it exists: a call after a failure half-way builds it again
    // instead of using null slots)
    if (!st->stream) HIP_TRY(hipStreamCreateWithFlags(&st->stream, hipStreamNonBlocking));
    for (int i = 0; i < HostStage::kSlots; i++) {
      if (!st->slot[i]) HIP_TRY(hipHostMalloc(&st->slot[i], kStageSlotBytes, hipHostMallocDefault));
      if (!st->ev[i]) HIP_TRY(hipEventCreateWithFlags(&st->ev[i], hipEventDisableTiming));
    }
    st->slot_bytes = kStageSlotBytes;
    st->device = device;
  }
  *out = st;
  return "";
}
}  // namespace

std::string dev_output_buffers(int worker, int device, size_t rgb8_bytes, size_t rgbf_bytes, void** d8, void** df) {
  HostStage* st = nullptr;
  std::string e = stage_for(worker * (int)kCopyThreads, device, &st);
  if (!e.empty()) return e;
  HIP_TRY(hipSetDevice(device));
  if (rgb8_bytes > st->d8_bytes) {
    (void)hipFree(st->d8);
    st->d8 = nullptr;
    st->d8_bytes = 0;
    HIP_TRY(hipMalloc(&st->d8, rgb8_bytes));
    st->d8_bytes = rgb8_bytes;
  }
  if (rgbf_bytes > st->df_bytes) {
    (void)hipFree(st->df);
    st->df = nullptr;
    st->df_bytes = 0;
    HIP_TRY(hipMalloc(&st->df, rgbf_bytes));
    st->df_bytes = rgbf_bytes;
  }
  *d8 = st->d8;
  *df = rgbf_bytes ? st->df : nullptr;
  return "";
}

std::string dev_copy_rows_to_host(int worker, int device, void* image, const void* d_rows, size_t row_bytes, int64_t height,
                                  int32_t tile_rows, int32_t part, int32_t nparts) {
  HIP_TRY(hipSetDevice(device));
  const int64_t rows = gmlrt_part_rows(height, tile_rows, part, nparts);
  if (rows <= 0) return "";
  const size_t total = (size_t)rows * row_bytes;
  // a unit is what goes to one contiguous place of the image: the whole partition when it is the whole image,
  // else one row tile (the partition's tiles lie nparts tiles apart in the image)
  const size_t band = (size_t)tile_rows * row_bytes;
  const size_t unit = nparts > 1 ? band : row_bytes;
  HostStage* st = nullptr;
  std::string e = stage_for(worker * (int)kCopyThreads, device, &st);
  if (!e.empty()) return e;
  if (unit > st->slot_bytes) {
    // rows too long for the staging slots: the runtime's own staged copies
    if (nparts <= 1) {
      HIP_TRY(hipMemcpy(image, d_rows, total, hipMemcpyDeviceToHost));
      return "";
    }
    const int64_t full = rows / tile_rows, rest = rows % tile_rows;
    uint8_t* dst = (uint8_t*)image + (size_t)part * band;
    if (full > 0)
      HIP_TRY(hipMemcpy2D(dst, (size_t)nparts * band, d_rows, band, band, (size_t)full, hipMemcpyDeviceToHost));
    if (rest > 0)
      HIP_TRY(hipMemcpy(dst + (size_t)full * (size_t)nparts * band, (const uint8_t*)d_rows + (size_t)full * band,
                        (size_t)rest * row_bytes, hipMemcpyDeviceToHost));
    return "";
  }
  // Large transfers are cut into ranges of whole units, each with a thread and a staging ring of its own: one
  // thread copies out of page-locked memory at 10 to 30 GB/s, the DMA engines deliver more than 50.
  const size_t n_units = (total + unit - 1) / unit;
  // (from 8 MB on: the 24.9 MB of a 4K frame took 0.93 ms through one thread, as long as its memcpy alone)
  static const size_t many_from = [] { const char* e = std::getenv("GMLRT_COPY_THREADS_FROM_MB"); return (size_t)(e ? std::atoi(e) : 8) << 20; }();  // tuning runs
  size_t n_threads = total >= many_from ? kCopyThreads : 1;
  if (n_threads > n_units) n_threads = n_units;
  auto copy_range = [&](HostStage* hs, size_t unit_begin, size_t unit_end) -> std::string {
    // (calls for one device share its stages: two callers in the same slots would interleave their chunks)
    std::lock_guard<std::mutex> one_at_a_time(hs->copy_mu);
    if (hipSetDevice(device) != hipSuccess) return "hipSetDevice failed in a copy thread";
    const size_t begin = unit_begin * unit, end = std::min(total, unit_end * unit);
    const size_t chunk = hs->slot_bytes / unit * unit;  // whole units per slot
    const size_t n_chunks = (end - begin + chunk - 1) / chunk;
    constexpr size_t K = HostStage::kSlots;
    auto issue = [&](size_t c) -> hipError_t {
      const size_t off = begin + c * chunk, len = std::min(chunk, end - off);
      hipError_t r = hipMemcpyAsync(hs->slot[c % K], (const uint8_t*)d_rows + off, len, hipMemcpyDeviceToHost, hs->stream);
      if (r != hipSuccess) return r;
      return hipEventRecord(hs->ev[c % K], hs->stream);
    };
    for (size_t c = 0; c < K - 1 && c < n_chunks; c++) HIP_TRY(issue(c));
    for (size_t c = 0; c < n_chunks; c++) {
      HIP_TRY(hipEventSynchronize(hs->ev[c % K]));
      if (c + K - 1 < n_chunks) HIP_TRY(issue(c + K - 1));  // into the slot emptied one step ago
      const size_t off = begin + c * chunk, len = std::min(chunk, end - off);
      const uint8_t* src = (const uint8_t*)hs->slot[c % K];
      if (nparts <= 1) {
        std::memcpy((uint8_t*)image + off, src, len);
      } else {
        for (size_t o = 0; o < len; o += band) {
          const size_t k = (off + o) / band;  // the partition's k-th tile is tile part + k * nparts of the image
          std::memcpy((uint8_t*)image + ((size_t)part + k * (size_t)nparts) * band, src + o, std::min(band, len - o));
        }
      }
    }
    return "";
  };
  if (n_threads == 1) return copy_range(st, 0, n_units);
  std::vector<HostStage*> stages(n_threads, st);
  for (size_t t = 1; t < n_threads; t++) {
    e = stage_for(worker * (int)kCopyThreads + (int)t, device, &stages[t]);
    if (!e.empty()) return e;
  }
  std::vector<std::string> errs(n_threads);
  std::vector<std::thread> th;
  for (size_t t = 1; t < n_threads; t++)
    th.emplace_back([&, t] { errs[t] = copy_range(stages[t], n_units * t / n_threads, n_units * (t + 1) / n_threads); });
  errs[0] = copy_range(stages[0], 0, n_units / n_threads);
  for (auto& x : th) x.join();
  for (auto& x : errs) if (!x.empty()) return x;
  return "";
}

namespace {
// Scratch of the last destroyed scene per device (event stacks, context records, counters, tile
// tables, timing events): allocating gigabytes of address space per render call costs milliseconds.
std::mutex g_spare_mu;
std::vector<DeviceScene> g_spare;

void free_workspace(DeviceScene* s) {
  (void)hipFree(s->d_ev_dist);
  (void)hipFree(s->d_ev_meta);
  (void)hipFree(s->d_ctx);
  (void)hipFree(s->d_counters);
  (void)hipFree(s->d_tile_cost);
  (void)hipFree(s->d_tile_order);
  (void)hipFree(s->d_tile_batch);
  (void)hipFree(s->d_tile_batch_img);
  (void)hipFree(s->d_tile_order_rest);
  (void)hipFree(s->d_tile_batch_rest);
  (void)hipFree(s->d_tiles_shared);
  (void)hipFree(s->d_sampled);
  (void)hipFree(s->d_xchg);
  (void)hipFree(s->d_queue[0]);
  (void)hipFree(s->d_queue[1]);
  (void)hipFree(s->d_qkeys);
  (void)hipFree(s->d_qsorted);
  (void)hipFree(s->d_parked);
  (void)hipFree(s->d_qcnt);
  if (s->h_staging) (void)hipHostFree(s->h_staging);
  if (s->ev_start) (void)hipEventDestroy((hipEvent_t)s->ev_start);
  if (s->ev_stop) (void)hipEventDestroy((hipEvent_t)s->ev_stop);
}

// ---- hand-out order of the pixel tiles from per-tile costs (ticks of the first launch with a geometry)
// order: tiles sorted by cost class (a few classes up to the dearest tile), dearest class first, image order
// within a class: neighbouring tiles walk the same parts of the scene program, and waves that run side by
// side should keep sharing them in cache.
void tiles_by_cost(const std::vector<uint32_t>& ticks, uint32_t n_classes, std::vector<uint32_t>* order) {
  const size_t n_tiles = ticks.size();
  uint32_t top = 1;
  for (uint32_t c : ticks) top = std::max(top, c);
  // counting sort by class, stable
  std::vector<uint32_t> cls(n_tiles), start(n_classes + 1, 0);
  for (size_t i = 0; i < n_tiles; i++) {
    cls[i] = n_classes - 1 - (uint32_t)((uint64_t)ticks[i] * n_classes / ((uint64_t)top + 1));  // 0 = dearest
    start[cls[i] + 1]++;
  }
  for (uint32_t c = 0; c < n_classes; c++) start[c + 1] += start[c];
  order->resize(n_tiles);
  for (size_t i = 0; i < n_tiles; i++) (*order)[start[cls[i]]++] = (uint32_t)i;
}
// Batches over an order (null: image order), each about the same cost: a dear tile is a batch of its own,
// cheap ones (sky, a bare floor) go up to kTileBatchMax together.  The target is a small share of what a wave
// does in the launch, so that the last batches still even out the end.  (Batches that shrink towards the end,
// each a fixed share of what remains per wave, measured slower: 4740 against 4830 Mrays/s on the chess frame.)
// Two limits on the number of batches: enough of them that the last ones even out the end of the launch (two
// dozen a wave), and few enough that the claims, which the memory system serves one at a time (about 14 ns
// each), stay a small part of the launch: at most one claim per 56 ns of it.
void batches_by_cost(const std::vector<uint32_t>& ticks, const std::vector<uint32_t>* order, uint64_t waves, double launch_ms,
                     std::vector<uint32_t>* start) {
  const size_t n_tiles = ticks.size();
  unsigned long long total = 0;
  for (uint32_t c : ticks) total += c;
  static const unsigned long long per_wave = [] { const char* e = std::getenv("GMLRT_BATCHES_PER_WAVE"); return e ? (unsigned long long)std::atoi(e) : 24ull; }();  // tuning runs
  unsigned long long n_target = waves * per_wave;
  static const double claim_ns = [] { const char* e = std::getenv("GMLRT_CLAIM_NS"); return e ? std::atof(e) : 56.0; }();  // tuning runs
  const unsigned long long by_claims = (unsigned long long)(launch_ms * 1e6 / claim_ns) + 1ull;
  if (n_target > by_claims) n_target = by_claims;
  const unsigned long long target = total / n_target + 1ull;
  start->clear();
  start->reserve(n_tiles + 1);
  unsigned long long acc = 0;
  uint32_t in_batch = 0;
  for (size_t i = 0; i < n_tiles; i++) {
    const unsigned long long c = ticks[order ? (*order)[i] : i];
    if (in_batch == 0 || in_batch >= kTileBatchMax || acc + c > target) {
      start->push_back((uint32_t)i);
      acc = 0;
      in_batch = 0;
    }
    acc += c;
    in_batch++;
  }
  start->push_back((uint32_t)n_tiles);  // number of batches = size - 1
}
}  // namespace

// ---- streaming a frame to the host band by band (the pixel sink of gmlrt_render_to_sink)
struct BandPipe {
  int device = -1;
  size_t band_pixels = 0;            // capacity of a slot in pixels
  void* d8[2] = {nullptr, nullptr};  // device: quantised band (the kernel always writes it)
  void* df[2] = {nullptr, nullptr};  // device: f64 band
  double* host[2] = {nullptr, nullptr};  // page-locked: f64 band
  hipStream_t copy_stream = nullptr;
  hipEvent_t copied[2] = {nullptr, nullptr};
  bool in_flight[2] = {false, false};
};

std::string band_pipe_create(int device, size_t band_pixels, BandPipe** out) {
  *out = nullptr;
  HIP_TRY(hipSetDevice(device));
  BandPipe* bp = new BandPipe();
  bp->device = device;
  bp->band_pixels = band_pixels;
  *out = bp;  // the caller destroys it whatever happens below
  HIP_TRY(hipStreamCreateWithFlags(&bp->copy_stream, hipStreamNonBlocking));
  for (int i = 0; i < 2; i++) {
    HIP_TRY(hipMalloc(&bp->d8[i], band_pixels * 3));
    HIP_TRY(hipMalloc(&bp->df[i], band_pixels * 3 * sizeof(double)));
    HIP_TRY(hipHostMalloc((void**)&bp->host[i], band_pixels * 3 * sizeof(double), hipHostMallocDefault));
    HIP_TRY(hipEventCreateWithFlags(&bp->copied[i], hipEventDisableTiming));
  }
  return "";
}

void band_pipe_destroy(BandPipe* bp) {
  if (!bp) return;
  if (hipSetDevice(bp->device) == hipSuccess) {
    if (bp->copy_stream) (void)hipStreamSynchronize(bp->copy_stream);
    for (int i = 0; i < 2; i++) {
      (void)hipFree(bp->d8[i]);
      (void)hipFree(bp->df[i]);
      if (bp->host[i]) (void)hipHostFree(bp->host[i]);
      if (bp->copied[i]) (void)hipEventDestroy(bp->copied[i]);
    }
    if (bp->copy_stream) (void)hipStreamDestroy(bp->copy_stream);
  }
  delete bp;
}

std::string band_pipe_render(BandPipe* bp, DeviceScene* s, const gmlrt_params& p, int slot, int64_t row0, int64_t rows,
                             gmlrt_stats* stats) {
  HIP_TRY(hipSetDevice(bp->device));
  if ((size_t)rows * (size_t)p.width > bp->band_pixels) return "band larger than its buffers";
  // the device buffers of this slot are free once the copy queued from them two bands ago has finished
  if (bp->in_flight[slot]) HIP_TRY(hipEventSynchronize(bp->copied[slot]));
  bp->in_flight[slot] = false;
  std::string e = dev_render(s, p, bp->d8[slot], bp->df[slot], nullptr, stats, false, row0, rows);  // returns when the kernel is done
  if (!e.empty()) return e;
  HIP_TRY(hipMemcpyAsync(bp->host[slot], bp->df[slot], (size_t)rows * (size_t)p.width * 3 * sizeof(double), hipMemcpyDeviceToHost,
                         bp->copy_stream));
  HIP_TRY(hipEventRecord(bp->copied[slot], bp->copy_stream));
  bp->in_flight[slot] = true;
  return "";
}

std::string band_pipe_wait(BandPipe* bp, int slot, const double** host) {
  HIP_TRY(hipEventSynchronize(bp->copied[slot]));
  *host = bp->host[slot];
  return "";
}

void dev_release_cached(void) {
  {
    std::lock_guard<std::mutex> lk(g_stage_mu);
    for (HostStage* st : g_stage) {
      if (!st) continue;
      if (st->device >= 0 && hipSetDevice(st->device) == hipSuccess) {
        for (int i = 0; i < HostStage::kSlots; i++) { if (st->slot[i]) (void)hipHostFree(st->slot[i]); if (st->ev[i]) (void)hipEventDestroy(st->ev[i]); }
        if (st->stream) (void)hipStreamDestroy(st->stream);
        (void)hipFree(st->d8);
        (void)hipFree(st->df);
      }
      delete st;
    }
    g_stage.clear();
  }
  std::lock_guard<std::mutex> lk(g_spare_mu);
  for (auto& w : g_spare)
    if (hipSetDevice(w.device) == hipSuccess) free_workspace(&w);
  g_spare.clear();
}

std::string dev_scene_create(const void* blob, size_t bytes, int device, DeviceScene* out) {
  HIP_TRY(hipSetDevice(device));
  {
    std::lock_guard<std::mutex> lk(g_spare_mu);
    for (size_t i = 0; i < g_spare.size(); i++)
      if (g_spare[i].device == device) {
        *out = g_spare[i];
        g_spare.erase(g_spare.begin() + (long)i);
        break;
      }
  }
  out->device = device;
  out->blob_bytes = bytes;
  std::memcpy(&out->hdr, blob, sizeof(out->hdr));
  out->levels_refused = false;
  {
    // The box the sort keys of queued rays are laid over: the bounded primitives' own boxes, taken to world space
    // (a half-space has none; a scene of nothing else gets a box around the origin).  Only the grouping of the rays
    // depends on it, never a result.
    const gmlrt_blob_header& H = out->hdr;
    const gmlrt_leaf* L = (const gmlrt_leaf*)((const uint8_t*)blob + H.off_leaves);
    double lo[3] = {1e300, 1e300, 1e300}, hi[3] = {-1e300, -1e300, -1e300};
    for (uint32_t i = 0; i < H.n_leaves; i++) {
      if (L[i].type == GMLRT_PLANE) continue;
      const double olo[3] = {L[i].type == GMLRT_CUBE ? 0.0 : -1.0, L[i].type == GMLRT_SPHERE ? -1.0 : 0.0, L[i].type == GMLRT_CUBE ? 0.0 : -1.0};
      for (int c = 0; c < 8; c++) {
        const double q[3] = {(c & 1) ? 1.0 : olo[0], (c & 2) ? 1.0 : olo[1], (c & 4) ? 1.0 : olo[2]};
        for (int a = 0; a < 3; a++) {
          const double v = L[i].m[a * 4] * q[0] + L[i].m[a * 4 + 1] * q[1] + L[i].m[a * 4 + 2] * q[2] + L[i].m[a * 4 + 3];
          if (v == v) { lo[a] = std::min(lo[a], v); hi[a] = std::max(hi[a], v); }
        }
      }
    }
    for (int a = 0; a < 3; a++) {
      if (!(hi[a] > lo[a]) || !(hi[a] - lo[a] < 1e30)) { lo[a] = -10.0; hi[a] = 10.0; }
      out->key_lo[a] = lo[a];
      out->key_inv[a] = 1.0 / (hi[a] - lo[a]);  // cells per unit for ONE cell over the box; the launch scales it
    }
  }
  {
    // The classes of rays that must meet no bound (see cylinder_special_case, cone_special_case): made on the host
    // from the leaves' transforms (gmlrt_scene.cpp: special_ray_classes).
    std::vector<double> axes;
    special_ray_classes(blob, &axes, &out->n_axes, &out->n_cone_forms);
    if (std::getenv("GMLRT_NO_SPECIAL_CASES")) { axes.clear(); out->n_axes = out->n_cone_forms = 0; }  // tuning runs: what the flags cost
    if (const char* e = std::getenv("GMLRT_ALL_RAYS_WITHOUT_BOUNDS"); e && *e == '1') {
      // tests: a class that every direction belongs to, so that every ray is traced by the full-list algorithm
      axes.insert(axes.begin(), {0.0, 1.0, 0.0, -1.0, 2.0, 0.0, 0.0, 0.0});
      out->n_axes++;
    }
    if (std::getenv("GMLRT_VERBOSE"))
      std::fprintf(stderr, "[gmlrt] rays traced without bounds: %u classes of cylinder axes and plane normals, %u cone forms\n", out->n_axes, out->n_cone_forms);
    out->d_axes = nullptr;
    out->h_axes = axes;
    if (!axes.empty()) {
      HIP_TRY(hipMalloc(&out->d_axes, axes.size() * sizeof(double)));
      HIP_TRY(hipMemcpy(out->d_axes, axes.data(), axes.size() * sizeof(double), hipMemcpyHostToDevice));
    }
  }
  HIP_TRY(hipMalloc(&out->d_blob, bytes));
  HIP_TRY(hipMemcpy(out->d_blob, blob, bytes, hipMemcpyHostToDevice));
  {
    uint32_t need_stack = 1;
    uint64_t need_events = 2;
    const auto t_derive0 = std::chrono::steady_clock::now();
    std::vector<uint8_t> prog = derive_stream_program(blob, &need_stack, &need_events, out->slice_off);
    if (std::getenv("GMLRT_VERBOSE"))
      std::fprintf(stderr, "[gmlrt] traversal program derived in %.3f ms\n",
                   std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_derive0).count());
    out->max_stack = std::max(out->hdr.max_stack, need_stack);
    out->max_events = std::max<uint64_t>(out->hdr.max_events, need_events);
    if (out->max_events > 0xFFFFFFFFull) return "too large: event stack";
    out->n_prog = (uint32_t)prog.size();
    HIP_TRY(hipMalloc(&out->d_prog, prog.size()));
    HIP_TRY(hipMemcpy(out->d_prog, prog.data(), prog.size(), hipMemcpyHostToDevice));
  }
  {
    // highest register any surface program writes (OUT's dst field is a slot selector, not a register)
    const gmlrt_blob_header& H = out->hdr;
    const uint64_t* code = (const uint64_t*)((const uint8_t*)blob + H.off_code);
    uint32_t top = 2;  // face, u, v
    for (uint64_t i = 0; i < H.n_code_words; i++) {
      const uint32_t op = (uint32_t)(code[i] & 0xFF), dst = (uint32_t)((code[i] >> 8) & 0xFF);
      if (op != GMLRT_SOP_END && op != GMLRT_SOP_OUT && op != GMLRT_SOP_TRAP && dst > top) top = dst;
    }
    out->n_vm_regs = H.n_code_words ? (top & (GMLRT_SURF_MAX_REGS - 1)) + 1 : 0;
    if (top >= GMLRT_SURF_MAX_REGS) out->n_vm_regs = GMLRT_SURF_MAX_REGS;
  }
  {
    std::vector<uint32_t> leaf_surf;
    std::vector<uint8_t> surf = derive_surface_stream(blob, &leaf_surf);
    HIP_TRY(hipMalloc(&out->d_surf, surf.size()));
    HIP_TRY(hipMemcpy(out->d_surf, surf.data(), surf.size(), hipMemcpyHostToDevice));
    HIP_TRY(hipMalloc(&out->d_leaf_surf, leaf_surf.size() * sizeof(uint32_t)));
    HIP_TRY(hipMemcpy(out->d_leaf_surf, leaf_surf.data(), leaf_surf.size() * sizeof(uint32_t), hipMemcpyHostToDevice));
  }
  if (!out->d_counters) HIP_TRY(hipMalloc(&out->d_counters, kCountersBytes));
  // through page-locked memory: copies to and from pageable memory are staged and wait
  if (!out->h_staging) HIP_TRY(hipHostMalloc(&out->h_staging, kStagingBytes, hipHostMallocDefault));
  if (device < kMaxDevices) {
    // this device's copy of the static scratch array (once per device, with the device current)
    std::lock_guard<std::mutex> lk(g_scratch_mu);
    if (!g_scratch_ptr[device]) {
      void* ptr = nullptr;
      HIP_TRY(hipGetSymbolAddress(&ptr, HIP_SYMBOL(scratch)));
      g_scratch_ptr[device] = ptr;
    }
    out->d_static_scratch = g_scratch_ptr[device];
  }
  {
    // device facts and the occupancy of this scene's kernel, asked once per handle (the queries cost tens
    // of microseconds, a fifth of a simple frame; and a render call that must not allocate makes none)
    const size_t lens_bytes = (((size_t)out->max_stack * kBlock * sizeof(uint16_t)) + 15) & ~(size_t)15;
    hipDeviceProp_t q;
    HIP_TRY(hipGetDeviceProperties(&q, device));
    int b = 1;
    if (lens_bytes <= 64 * 1024) HIP_TRY(hipOccupancyMaxActiveBlocksPerMultiprocessor(&b, render_kernel<true, 0>, kBlock, lens_bytes));
    out->dev_cus = q.multiProcessorCount;
    out->dev_lds_block = q.sharedMemPerBlock;
    out->dev_lds_cu = q.maxSharedMemoryPerMultiProcessor;
    out->dev_blocks_per_cu = b < 1 ? 1 : b;
  }
  if (!out->ev_start) {
    hipEvent_t a, b;
    HIP_TRY(hipEventCreate(&a));
    HIP_TRY(hipEventCreate(&b));
    out->ev_start = a;
    out->ev_stop = b;
  }
  return "";
}

void dev_scene_destroy(DeviceScene* s) {
  if (!s || s->device < 0) return;
  if (hipSetDevice(s->device) != hipSuccess) return;
  (void)hipFree(s->d_blob);
  (void)hipFree(s->d_prog);
  (void)hipFree(s->d_surf);
  (void)hipFree(s->d_leaf_surf);
  (void)hipFree(s->d_axes);
  // the scratch stays for the next scene on this device (one set per device)
  DeviceScene spare;
  spare.device = s->device;
  spare.d_ev_dist = s->d_ev_dist; spare.d_ev_meta = s->d_ev_meta; spare.d_ctx = s->d_ctx;
  spare.d_counters = s->d_counters;
  spare.ev_slots_alloc = s->ev_slots_alloc; spare.ctx_fields_alloc = s->ctx_fields_alloc;
  spare.d_tile_cost = s->d_tile_cost; spare.d_tile_order = s->d_tile_order; spare.d_tile_batch = s->d_tile_batch; spare.d_tile_batch_img = s->d_tile_batch_img;
  spare.d_tile_order_rest = s->d_tile_order_rest; spare.d_tile_batch_rest = s->d_tile_batch_rest; spare.d_tiles_shared = s->d_tiles_shared;
  spare.d_xchg = s->d_xchg; spare.xchg_blocks = s->xchg_blocks; spare.d_sampled = s->d_sampled;
  spare.d_queue[0] = s->d_queue[0]; spare.d_queue[1] = s->d_queue[1]; spare.d_qkeys = s->d_qkeys; spare.d_qsorted = s->d_qsorted;
  spare.d_parked = s->d_parked; spare.d_qcnt = s->d_qcnt; spare.queue_cap = s->queue_cap; spare.parked_records = s->parked_records;
  spare.h_staging = s->h_staging; spare.tiles_alloc = s->tiles_alloc;
  spare.ev_start = s->ev_start; spare.ev_stop = s->ev_stop;
  bool kept = false;
  {
    std::lock_guard<std::mutex> lk(g_spare_mu);
    bool have = false;
    for (auto& w : g_spare) have = have || w.device == spare.device;
    if (!have) { g_spare.push_back(spare); kept = true; }
  }
  if (!kept) free_workspace(&spare);
  *s = DeviceScene{};
}

std::string dev_render(DeviceScene* s, const gmlrt_params& p, void* d_rgb8, void* d_rgbf, void* stream_v,
                       gmlrt_stats* stats, bool no_alloc, int64_t band_row0, int64_t band_rows) {
  HIP_TRY(hipSetDevice(s->device));
  hipStream_t stream = (hipStream_t)stream_v;
  const gmlrt_blob_header& H = s->hdr;
  const uint8_t* base = (const uint8_t*)s->d_blob;

  const gmlrt_leaf* d_leaves = (const gmlrt_leaf*)(base + H.off_leaves);
  const uint32_t* d_nodes = (const uint32_t*)(base + H.off_nodes);
  const gmlrt_light* d_lights = (const gmlrt_light*)(base + H.off_lights);
  const uint64_t* d_code = (const uint64_t*)(base + H.off_code);
  const uint64_t* d_consts = (const uint64_t*)(base + H.off_consts);
  const gmlrt_bound* d_bounds = (const gmlrt_bound*)(base + H.off_bounds);

  KArgs A;
  std::memset(&A, 0, sizeof(A));
  A.n_leaves = H.n_leaves;
  A.n_nodes = H.n_nodes;
  A.n_prog = s->n_prog;
  A.n_lights = H.n_lights;
  A.max_stack = s->max_stack;
  A.axes = (const double*)s->d_axes;
  A.n_axes = s->n_axes;
  A.n_cone_forms = s->n_cone_forms;
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
  const int64_t rows_all = gmlrt_part_rows(p.height, A.tile_rows, A.part, nparts);
  if (band_row0 < 0 || band_row0 > rows_all) return "band outside the partition";
  const int64_t rows = band_rows < 0 || band_row0 + band_rows > rows_all ? rows_all - band_row0 : band_rows;
  A.rows_part = (int32_t)rows;
  A.band_row0 = (int32_t)band_row0;
  A.tiles_x = (uint32_t)((p.width + kTileW - 1) / kTileW);
  const uint64_t tiles_y = (uint64_t)((rows + kTileH - 1) / kTileH);
  const uint64_t n_slots = (uint64_t)A.tiles_x * tiles_y * 64ull;
  if (n_slots >= 0xFFFF0000ull) return "image too large for one launch";
  A.n_slots = (uint32_t)n_slots;
  A.rgb8 = (uint8_t*)d_rgb8;
  A.rgbf = (double*)d_rgbf;

  if (stats) {
    stats->rays_primary = stats->rays_shadow = stats->rays_reflect = stats->hits = 0;
    stats->kernel_ms = 0.0;
    stats->tiles_shared = 0;
    stats->handout_way = 0;
  }
  if (rows == 0) return "";

  // ---- launch geometry: a persistent grid sized to what the device keeps resident
  const size_t lens_bytes = (((size_t)s->max_stack * kBlock * sizeof(uint16_t)) + 15) & ~(size_t)15;
  if (lens_bytes > 64 * 1024) return "CSG tree nests deeper than the operand stack supports";
  struct { int multiProcessorCount; size_t sharedMemPerBlock, maxSharedMemoryPerMultiProcessor; } prop = {
      s->dev_cus, s->dev_lds_block, s->dev_lds_cu};
  const int blocks_per_cu = s->dev_blocks_per_cu;
  // The surface bytecode's registers go to whatever LDS is left at that occupancy.
  size_t n_vm_fast = s->n_vm_regs;
  {
    const size_t per_block = (size_t)prop.sharedMemPerBlock < (size_t)prop.maxSharedMemoryPerMultiProcessor / blocks_per_cu
                                 ? (size_t)prop.sharedMemPerBlock
                                 : (size_t)prop.maxSharedMemoryPerMultiProcessor / blocks_per_cu;
    const size_t room = per_block > lens_bytes + 1024 ? (per_block - lens_bytes - 1024) / ((size_t)kBlock * 8) : 0;
    if (n_vm_fast > room) n_vm_fast = room;
  }
  // With room for all of them the hot context fields go to LDS as well, into the same region.
  const size_t hot_bytes = (size_t)kHotFields * kBlock * sizeof(double);
  size_t lds_spare = 0;  // bytes of the block's LDS share that nothing uses
  {
    const size_t per_block = (size_t)prop.sharedMemPerBlock < (size_t)prop.maxSharedMemoryPerMultiProcessor / blocks_per_cu
                                 ? (size_t)prop.sharedMemPerBlock
                                 : (size_t)prop.maxSharedMemoryPerMultiProcessor / blocks_per_cu;
    A.lds_ctx = lens_bytes + hot_bytes <= per_block ? 1u : 0u;
    lds_spare = A.lds_ctx ? per_block - lens_bytes - hot_bytes : 0;
  }
  // (six of the region's slots are kept for the programs' results: gmlrt_shade.hpp)
  if (A.lds_ctx) n_vm_fast = s->n_vm_regs < (uint32_t)kHotFields - 6u ? s->n_vm_regs : (uint32_t)kHotFields - 6u;
  size_t lds_bytes = lens_bytes + (A.lds_ctx ? hot_bytes : n_vm_fast * kBlock * sizeof(uint64_t));
  A.n_vm_fast = (uint32_t)n_vm_fast;
  A.vm_lds_off = (uint32_t)lens_bytes;
  uint64_t grid = (uint64_t)prop.multiProcessorCount * (uint64_t)blocks_per_cu;
  if (const char* e = std::getenv("GMLRT_GRID_BLOCKS")) {  // tuning runs
    const long v = std::atol(e);
    if (v > 0 && (uint64_t)v < grid) grid = (uint64_t)v;
  }
  const uint64_t need = (n_slots + kBlock - 1) / kBlock;
  if (grid > need) grid = need;
  if (grid > kMaxGridBlocks) grid = kMaxGridBlocks;
  if (grid < 1) grid = 1;

  // ---- workspace
  const size_t ev_cap = (size_t)s->max_events;
  // shading context + one parked record per bounce level + the surface bytecode's registers (at the end)
  const size_t ctx_fields =
      (size_t)kCtxFixed + (size_t)kCtxPerLevel * (size_t)(p.depth + 1) + (size_t)GMLRT_SURF_MAX_REGS;
  uint8_t* d_counters = (uint8_t*)s->d_counters;
  if (no_alloc) {
    // Everything the kernel writes besides its outputs comes out of this device's copy of the static array:
    // counters | event distances | context records | event meta words.  The grid shrinks to the lanes that fit.
    uint8_t* base_ptr = (uint8_t*)s->d_static_scratch;
    if (!base_ptr) return "too large: no static scratch on this device";
    constexpr size_t counters_bytes = (kCountersBytes + 255) & ~(size_t)255;
    const size_t per_lane = ev_cap * (sizeof(double) + sizeof(uint32_t)) + ctx_fields * sizeof(double);
    const uint64_t fit = (kNoAllocScratchBytes - counters_bytes) / (per_lane * kBlock);
    if (fit < 1) return "too large: one block of this scene needs more than the static scratch buffer holds";
    if (grid > fit) grid = fit;
    const size_t lanes_na = (size_t)grid * kBlock;
    d_counters = base_ptr;
    A.ev_dist = (double*)(base_ptr + counters_bytes);
    A.ctx = (double*)(base_ptr + counters_bytes + lanes_na * ev_cap * sizeof(double));
    A.ev_meta = (uint32_t*)(base_ptr + counters_bytes + lanes_na * (ev_cap + ctx_fields) * sizeof(double));
    A.ev_cap = (uint32_t)ev_cap;
    A.ctx_fields = (uint32_t)ctx_fields;
  } else {
  // (a size on record is cleared together with its pointer and set again only after the allocation
  // has succeeded: a failed hipMalloc must not leave a size that vouches for a null buffer)
  const size_t lanes = (size_t)grid * kBlock;
  if (lanes * ev_cap > s->ev_slots_alloc) {
    (void)hipFree(s->d_ev_dist);
    (void)hipFree(s->d_ev_meta);
    s->d_ev_dist = s->d_ev_meta = nullptr;
    s->ev_slots_alloc = 0;
    HIP_TRY(hipMalloc(&s->d_ev_dist, lanes * ev_cap * sizeof(double)));
    HIP_TRY(hipMalloc(&s->d_ev_meta, lanes * ev_cap * sizeof(uint32_t)));
    s->ev_slots_alloc = lanes * ev_cap;
  }
  if (lanes * ctx_fields > s->ctx_fields_alloc) {
    (void)hipFree(s->d_ctx);
    s->d_ctx = nullptr;
    s->ctx_fields_alloc = 0;
    HIP_TRY(hipMalloc(&s->d_ctx, lanes * ctx_fields * sizeof(double)));
    s->ctx_fields_alloc = lanes * ctx_fields;
  }
  A.ev_dist = (double*)s->d_ev_dist;
  A.ev_meta = (uint32_t*)s->d_ev_meta;
  A.ev_cap = (uint32_t)ev_cap;
  A.ctx = (double*)s->d_ctx;
  A.ctx_fields = (uint32_t)ctx_fields;
  }
  const size_t lanes = (size_t)grid * kBlock;
  (void)lanes;
  // ---- tile order: costs of an earlier launch with the same geometry, dearest first
  const size_t n_tiles = (size_t)(n_slots / 64);
  if (!no_alloc && n_tiles > s->tiles_alloc) {
    (void)hipFree(s->d_tile_cost);
    (void)hipFree(s->d_tile_order);
    s->d_tile_cost = s->d_tile_order = nullptr;
    (void)hipFree(s->d_tile_batch);
    (void)hipFree(s->d_tile_batch_img);
    s->d_tile_batch = s->d_tile_batch_img = nullptr;
    (void)hipFree(s->d_tile_order_rest);
    (void)hipFree(s->d_tile_batch_rest);
    (void)hipFree(s->d_tiles_shared);
    (void)hipFree(s->d_sampled);
    s->d_tile_order_rest = s->d_tile_batch_rest = s->d_tiles_shared = s->d_sampled = nullptr;
    s->tiles_alloc = 0;
    s->order_valid = false;

    HIP_TRY(hipMalloc(&s->d_tile_cost, n_tiles * sizeof(uint32_t)));
    HIP_TRY(hipMalloc(&s->d_tile_order, n_tiles * sizeof(uint32_t)));
    HIP_TRY(hipMalloc(&s->d_tile_batch, (n_tiles + 1) * sizeof(uint32_t)));
    HIP_TRY(hipMalloc(&s->d_tile_batch_img, (n_tiles + 1) * sizeof(uint32_t)));
    HIP_TRY(hipMalloc(&s->d_tile_order_rest, n_tiles * sizeof(uint32_t)));
    HIP_TRY(hipMalloc(&s->d_tile_batch_rest, (n_tiles + 1) * sizeof(uint32_t)));
    HIP_TRY(hipMalloc(&s->d_tiles_shared, n_tiles * sizeof(uint32_t)));
    HIP_TRY(hipMalloc(&s->d_sampled, ((size_t)(1 + kSmClasses) * n_tiles + kSmCntWords) * sizeof(uint32_t)));
    s->tiles_alloc = n_tiles;
  }
  gmlrt_params key = p;
  key.amb[0] = key.amb[1] = key.amb[2] = 0.0;  // the ambient colour moves no ray
  key.tile_rows = A.tile_rows; key.part = A.part; key.nparts = A.nparts;
  key.reserved = (int32_t)(band_row0 * 1315423911ll + rows);  // a band of the partition is a geometry of its own
  if (!no_alloc) {
    if (s->order_valid && std::memcmp(&key, &s->order_key, sizeof(key)) != 0) s->order_valid = false;
    if (!s->order_valid) { s->order_state = 0; s->order_trials = 0; }
  }
  static const bool no_order_env = std::getenv("GMLRT_NO_TILE_ORDER") != nullptr;  // tuning runs
  // (a call that must not allocate neither learns a hand-out order nor uses one: the tables are allocations)
  const bool no_order = no_order_env || no_alloc;
  int way = !s->order_valid || no_order ? 0 : s->order_state >= 1 && s->order_state <= 4 ? s->order_state : s->order_state == 5 ? s->order_way : 0;
  // ---- level by level (way 5): one launch per bounce level, bounce rays queued and regrouped in between.  It needs
  // no tile costs, so a frame rendered once runs as fast as a repeated one.  Buffers: two ray queues (a level can hold
  // one ray per pixel, plus the chunks the waves have not filled), keys and sorted positions, one parked record per
  // pixel and level.  A frame that can have no bounce, a call that must not allocate, or buffers beyond the limit
  // (GMLRT_LEVELS_MAX_GB, default 24) stay with the one-launch kernel.
  bool levels = false;
  size_t q_cap_chunks = 0;
  const size_t n_pix = (size_t)rows * (size_t)p.width;
  {
    const char* e = std::getenv("GMLRT_LEVELS");  // read at every call (tests switch between the two kernels)
    const bool wanted = e && *e == '1';
    if (wanted && !no_alloc && p.depth > 0 && p.depth <= (int64_t)kMaxLevels && !s->levels_refused) {
      const size_t waves = (size_t)grid * (kBlock / kWave);
      q_cap_chunks = (n_pix + kQueueChunk - 1) / kQueueChunk + waves + 1;
      const size_t cap = q_cap_chunks * kQueueChunk;
      const size_t parked = n_pix * (size_t)p.depth;
      static const double max_gb = [] { const char* m = std::getenv("GMLRT_LEVELS_MAX_GB"); return m ? std::atof(m) : 24.0; }();
      const double bytes = (double)cap * (2.0 * kRayWords * 8 + 8) + (double)parked * 64.0;
      if (cap < 0xFFFF0000ull && bytes <= max_gb * 1073741824.0) {
        levels = true;
        if (cap > s->queue_cap) {
          for (int q = 0; q < 2; q++) { (void)hipFree(s->d_queue[q]); s->d_queue[q] = nullptr; }
          (void)hipFree(s->d_qkeys);
          (void)hipFree(s->d_qsorted);
          s->d_qkeys = s->d_qsorted = nullptr;
          s->queue_cap = 0;
          bool ok = true;
          for (int q = 0; q < 2; q++) ok = ok && hipMalloc(&s->d_queue[q], cap * kRayWords * sizeof(uint64_t)) == hipSuccess;
          ok = ok && hipMalloc(&s->d_qkeys, cap * sizeof(uint32_t)) == hipSuccess;
          ok = ok && hipMalloc(&s->d_qsorted, cap * sizeof(uint32_t)) == hipSuccess;
          if (ok) s->queue_cap = cap; else levels = false;
        }
        if (levels && parked > s->parked_records) {
          (void)hipFree(s->d_parked);
          s->d_parked = nullptr;
          s->parked_records = 0;
          if (hipMalloc(&s->d_parked, parked * 8 * sizeof(double)) == hipSuccess) s->parked_records = parked; else levels = false;
        }
        if (levels && !s->d_qcnt && hipMalloc(&s->d_qcnt, kQcWords * sizeof(uint32_t)) != hipSuccess) { s->d_qcnt = nullptr; levels = false; }
        if (!levels) {
          (void)hipGetLastError();  // out of memory: not an error of this call, the one-launch kernel renders the frame
          s->levels_refused = true;
        }
      }
    }
  }
  if (levels) way = 5;
  // The first launch with a geometry times its tiles and the host makes the other ways' tables from that: worth it
  // only when a wave renders several tiles (a frame of fewer is never timed and stays with way 0).
  const bool learning = !levels && s->order_state == 0 && !no_order && s->learn_order && n_tiles >= 2 * (size_t)grid * (kBlock / kWave);
  A.tile_order = way == 1 || way == 4 ? (const uint32_t*)s->d_tile_order : way == 3 ? (const uint32_t*)s->d_tile_order_rest : nullptr;
  A.batch_start = way == 1 || way == 4 ? (const uint32_t*)s->d_tile_batch : way == 2 ? (const uint32_t*)s->d_tile_batch_img
                  : way == 3 ? (const uint32_t*)s->d_tile_batch_rest : nullptr;
  A.n_tiles = (uint32_t)n_tiles;
  A.stride = 1;
  A.n_batches = way == 1 || way == 4 ? s->n_batches : way == 2 ? s->n_batches_img : way == 3 ? s->n_batches_rest
                : (uint32_t)((n_tiles + kTileBatch - 1) / kTileBatch);
  // The end of a launch shared among the waves of a block (see the kernel): for launches that know no tile costs and
  // are not there to measure them (a frame rendered once ends on whatever dear tile came up last), and together with
  // the shared tiles of way 3.  Needs the frontier's slices, the LDS context region and 16 spare bytes of LDS.
  A.tail_lds_off = 0;
  {
    const bool tail_share = [] { const char* e = std::getenv("GMLRT_TAIL_SHARE"); return !(e && *e == '0'); }();  // tuning runs
    const bool timing = learning;
    if (tail_share && A.lds_ctx && s->slice_off[0] != 0 && lds_spare >= 16 && ((way == 0 && !timing) || way == 3)) {
      A.tail_lds_off = (uint32_t)lds_bytes;
      lds_bytes += 16;
    }
  }
  A.xchg = nullptr;
  if (way == 4) {
    if (s->xchg_blocks < grid) {
      (void)hipFree(s->d_xchg);
      s->d_xchg = nullptr;
      s->xchg_blocks = 0;
      HIP_TRY(hipMalloc(&s->d_xchg, (size_t)grid * kBlock * 8 * sizeof(uint64_t)));
      s->xchg_blocks = grid;
    }
    A.xchg = (uint64_t*)s->d_xchg;
    A.regroup_mode = [] { const char* e = std::getenv("GMLRT_REGROUP_MODE"); return e ? (uint32_t)std::atoi(e) : 0u; }();  // tuning runs
  }
  A.tiles_shared = way == 3 ? (const uint32_t*)s->d_tiles_shared : nullptr;
  A.n_tiles_shared = way == 3 ? s->n_tiles_shared : 0u;
  for (uint32_t k = 0; k < kProgSlices; k++) A.slice_off[k] = s->slice_off[k];
  static const bool no_stride = std::getenv("GMLRT_IMAGE_ORDER") != nullptr;  // tuning runs
  if (way == 0 && !no_stride && H.n_leaves > 16) {
    // No tile costs known (a handle that renders once, or the first launch of one that will learn them, or a
    // frame on which neither learnt way paid).  In image order the dear tiles of a frame tend to
    // sit together (the lower rows of a scene on a floor) and the launch ends on them; a golden-ratio
    // stride, made coprime with the tile count, spreads neighbours over the whole launch instead
    // (chess 4K: 7.5 -> 6.4 ms, fractal 2.45 -> 2.15 ms; a scene of a few primitives loses 3 %).
    auto gcd = [](uint64_t a, uint64_t b) { while (b) { const uint64_t t = a % b; a = b; b = t; } return a; };
    uint64_t st = (uint64_t)((double)n_tiles * 0.6180339887) | 1ull;
    while (st > 1 && gcd(st, n_tiles) != 1) st += 2;
    if (st < n_tiles) A.stride = (uint32_t)st;
  }
  // The sampled hand-out (way 6, see kSmClasses; GMLRT_SAMPLED=1): for a launch that knows no costs and is not there to measure
  // them, of a scene with enough in it that tiles differ (the same bar as for the stride) and enough tiles that their order matters.
  bool sampled = false;
  {
    const char* e = std::getenv("GMLRT_SAMPLED");  // tuning runs and tests; read at every call
    const uint64_t waves = grid * (uint64_t)(kBlock / kWave);
    // (on request only: measured, it gains 3 % on the first chess frame and loses on the fractal at every setting tried)
    sampled = e && *e == '1' && way == 0 && !learning && !no_alloc && A.lds_ctx && A.stride > 1 && p.depth >= 1 && s->d_sampled &&
              n_tiles >= 8 * waves && n_tiles <= s->tiles_alloc && n_tiles < (1u << 24);  // (list positions travel in 24 bits)
  }
  if (sampled && lds_spare < 64) sampled = false;
  if (sampled) {
    {
      // seven eighths, 96 %, 99 % of the samples; 3, 6, 12 times their average (GMLRT_SM_TUNE="s0,s1,s2,f0,f1,f2": tuning runs)
      const uint32_t dflt[6] = {224, 245, 253, 3, 6, 12};
      uint32_t v[6];
      for (int i = 0; i < 6; i++) v[i] = dflt[i];
      if (const char* e = std::getenv("GMLRT_SM_TUNE")) {
        int i = 0;
        for (const char* q = e; *q && i < 6; i++) { v[i] = (uint32_t)std::strtoul(q, const_cast<char**>(&q), 10); while (*q == ',') q++; }
      }
      for (int c = 0; c < 3; c++) { A.sm_share[c] = v[c] < 1 ? 1 : v[c] > 256 ? 256 : v[c]; A.sm_factor[c] = v[3 + c]; }
    }
    A.sm_lds_off = (uint32_t)((lds_bytes + 7) & ~(size_t)7);
    lds_bytes = (size_t)A.sm_lds_off + 8 * kSmClasses;
    uint32_t* base_w = (uint32_t*)s->d_sampled;
    A.sm_claimed = base_w;
    A.sm_list = base_w + n_tiles;
    A.sm_cnt = base_w + (size_t)(1 + kSmClasses) * n_tiles;
    const uint64_t tiles_y_all = n_tiles / A.tiles_x;
    A.sm_groups_x = (A.tiles_x + kSmGroup - 1) / kSmGroup;
    A.sm_groups = A.sm_groups_x * (uint32_t)((tiles_y_all + kSmGroup - 1) / kSmGroup);
    auto gcd = [](uint64_t a, uint64_t b) { while (b) { const uint64_t t = a % b; a = b; b = t; } return a; };
    uint64_t st = (uint64_t)((double)A.sm_groups * 0.6180339887) | 1ull;
    while (st > 1 && gcd(st, A.sm_groups) != 1) st += 2;
    A.sm_stride = st < A.sm_groups ? (uint32_t)st : 1u;
    A.n_batches = A.sm_groups + (uint32_t)((n_tiles + kTileBatch - 1) / kTileBatch);
    HIP_TRY(hipMemsetAsync(base_w, 0xFF, (size_t)(1 + kSmClasses) * n_tiles * sizeof(uint32_t), stream));
    HIP_TRY(hipMemsetAsync(A.sm_cnt, 0, kSmCntWords * sizeof(uint32_t), stream));
    way = 6;
  }
  A.tile_cost = learning ? (uint32_t*)s->d_tile_cost : nullptr;  // only the learning launch times its tiles
  static const bool verbose_shared = std::getenv("GMLRT_VERBOSE") != nullptr;
  if (verbose_shared && way == 3 && s->order_state == 3) A.tile_cost = (uint32_t*)s->d_tile_cost;  // (what sharing did to the dear tiles, for the log)
#ifdef GMLRT_DIAG
  unsigned long long* d_tile_dbg = nullptr;
  HIP_TRY(hipMalloc(&d_tile_dbg, n_tiles * 4 * sizeof(unsigned long long)));
  HIP_TRY(hipMemsetAsync(d_tile_dbg, 0, n_tiles * 4 * sizeof(unsigned long long), stream));
  A.tile_dbg = d_tile_dbg;
  unsigned long long* d_leaf_dbg = nullptr;
  HIP_TRY(hipMalloc(&d_leaf_dbg, (size_t)H.n_leaves * kLeafDbg * sizeof(unsigned long long)));
  HIP_TRY(hipMemsetAsync(d_leaf_dbg, 0, (size_t)H.n_leaves * kLeafDbg * sizeof(unsigned long long), stream));
  A.leaf_dbg = d_leaf_dbg;
#endif
  A.next_batch = (unsigned int*)d_counters;
  A.block_counts = (unsigned long long*)(d_counters + kOffBlockCounts);
  A.diag = (unsigned long long*)(d_counters + kOffDiag);
  // The ray counts of the blocks and the surface-error word go straight to page-locked host memory (every block
  // stores its four counts once, at its end): no copy behind the kernel, one operation less for the stream to start
  // and finish in every launch.  (Level by level the counts add up over several launches: on the device.)
  static const bool counts_to_host = [] { const char* e = std::getenv("GMLRT_COUNTS_TO_HOST"); return !e || *e != '0'; }();  // tuning runs
  const bool zero_copy = counts_to_host && !levels;
  if (zero_copy) A.block_counts = (unsigned long long*)((uint8_t*)s->h_staging + 1024);

  // (with the static scratch this clears what the previous call on this device left in the counters)
  HIP_TRY(hipMemsetAsync(d_counters, 0, kOffShadeArgs, stream));  // claim and diagnostic counters; every block writes its counts
  ShadeArgs SA;
  std::memset(&SA, 0, sizeof(SA));
  for (int i = 0; i < 3; i++) SA.amb[i] = p.amb[i];
  SA.depth = A.depth;
  SA.n_lights = A.n_lights;
  SA.ctx_fields = A.ctx_fields;
  SA.n_vm_fast = A.n_vm_fast;
  SA.vm_all_fast = A.n_vm_fast >= s->n_vm_regs ? 1u : 0u;
  SA.rgb8 = A.rgb8;
  SA.rgbf = A.rgbf;
  SA.leaves = d_leaves;
  SA.lights = d_lights;
  SA.code = d_code;
  SA.consts = d_consts;
  SA.leaf_surf = (const uint32_t*)s->d_leaf_surf;
  SA.surf_prog = (const SurfInstr*)s->d_surf;
  SA.surf_error = zero_copy ? (uint32_t*)((uint8_t*)s->h_staging + 1024 - 8) : (uint32_t*)(d_counters + kOffSurfError);
  SA.ctx = A.ctx;
  SA.c_x = A.c_x; SA.c_y = A.c_y; SA.pw = A.pw;
  SA.axes = A.axes; SA.n_axes = A.n_axes; SA.n_cone_forms = A.n_cone_forms;
  for (uint32_t k = 0; k < kOddInline; k++) {
    const bool ha = k < s->n_axes && s->h_axes.size() >= (size_t)(k + 1) * kAxisWords;
    const double* a = ha ? s->h_axes.data() + (size_t)k * kAxisWords : nullptr;
    SA.odd_axis[k][0] = ha ? a[0] : 0.0; SA.odd_axis[k][1] = ha ? a[1] : 0.0; SA.odd_axis[k][2] = ha ? a[2] : 0.0;
    SA.odd_axis[k][3] = ha ? a[3] : 2.0;   // (no cosine reaches 2, none is below -1)
    SA.odd_axis_lo[k] = ha ? a[4] : -1.0;
    const bool hf = k < s->n_cone_forms && s->h_axes.size() >= (size_t)(s->n_axes + k + 1) * kAxisWords;
    const double* f = hf ? s->h_axes.data() + (size_t)(s->n_axes + k) * kAxisWords : nullptr;
    for (int q = 0; q < 6; q++) SA.odd_form[k][q] = hf ? f[q] : (q < 3 ? 1.0 : 0.0);  // (the unit form: 1 for every unit direction)
  }
  SA.w = A.w; SA.tile_rows = A.tile_rows; SA.part = A.part; SA.nparts = A.nparts; SA.band_row0 = A.band_row0;
  SA.vm_lds_off = A.vm_lds_off;
  SA.parked = levels ? (double*)s->d_parked : nullptr;
  SA.parked_stride = (uint32_t)p.depth;
#ifdef GMLRT_DIAG
  unsigned long long* d_diag_waves = nullptr;
  HIP_TRY(hipMalloc(&d_diag_waves, lanes / kWave * kDiagWaveSlots * sizeof(unsigned long long)));
  HIP_TRY(hipMemsetAsync(d_diag_waves, 0, lanes / kWave * kDiagWaveSlots * sizeof(unsigned long long), stream));
  SA.diag_waves = d_diag_waves;
#endif
  // (the whole slot: the zeros behind the record clear the surface-error word of the previous launch)
  std::memset(s->h_staging, 0, kOffBlockCounts - kOffShadeArgs);
  if (zero_copy) *(unsigned long long*)((uint8_t*)s->h_staging + 1024 - 8) = 0ull;  // (the word the lanes themselves write, then)
  std::memcpy(s->h_staging, &SA, sizeof(SA));
  HIP_TRY(hipMemcpyAsync(d_counters + kOffShadeArgs, s->h_staging, kOffBlockCounts - kOffShadeArgs, hipMemcpyHostToDevice, stream));
  HIP_TRY(hipEventRecord((hipEvent_t)s->ev_start, stream));
  if (levels) {
    uint32_t* qc = (uint32_t*)s->d_qcnt;
    // (tuning runs and tests; read at every call)
    const uint32_t key_mode = [] { const char* e = std::getenv("GMLRT_LEVELS_SORT"); const int v = e ? std::atoi(e) : 1; return (uint32_t)(v < 0 || v > 2 ? 1 : v); }();
    const uint32_t key_bits = [] { const char* e = std::getenv("GMLRT_LEVELS_KEY_BITS"); const int v = e ? std::atoi(e) : (int)kKeyBitsMax; return (uint32_t)(v < 1 ? 1 : v > (int)kKeyBitsMax ? (int)kKeyBitsMax : v); }();
    const uint32_t n_keys = key_mode ? 1u << (2 * key_bits + 1 + kKeyDirBits) : 1u;
    A.key_mode = key_mode;
    A.key_bits = key_bits;
    for (int a = 0; a < 3; a++) { A.key_lo[a] = s->key_lo[a]; A.key_inv[a] = s->key_inv[a] * (a == 1 ? 2.0 : (double)(1u << key_bits)); }
    A.q_sorted = (const uint32_t*)s->d_qsorted;
    A.q_out_keys = (uint32_t*)s->d_qkeys;
    A.q_overflow = qc + kQcOverflow;
    A.q_cap_chunks = (uint32_t)q_cap_chunks;
    A.n_batches = (uint32_t)((n_tiles + kTileBatch - 1) / kTileBatch);
    // (the launches of a frame add their ray counts up, and each level claims its work from counters of its own)
    HIP_TRY(hipMemsetAsync(qc, 0, (kQcHist + n_keys) * sizeof(uint32_t), stream));
    HIP_TRY(hipMemsetAsync(d_counters + kOffBlockCounts, 0, (size_t)grid * 4 * sizeof(unsigned long long), stream));
    const unsigned sort_blocks = (unsigned)std::min<size_t>((q_cap_chunks * kQueueChunk + kSortSegment - 1) / kSortSegment, (size_t)prop.multiProcessorCount * 8);
    // The launches: starts[g] is the level launch g begins at; a bounce to a level below the next launch's start stays
    // in its lane (GMLRT_LEVEL_CUTS: the starts after 0, a comma-separated list; default 2, 3, ... depth).
    std::vector<int64_t> starts = {0};
    {
      const char* e = std::getenv("GMLRT_LEVEL_CUTS");  // tuning runs
      if (e && *e) {
        for (const char* q = e; *q;) {
          const long v = std::strtol(q, const_cast<char**>(&q), 10);
          if (v > starts.back() && v <= p.depth) starts.push_back(v);
          while (*q == ',') q++;
          if (!(*q >= '0' && *q <= '9')) break;
        }
      } else {
        // (the first bounce stays in its lane: the 64 rays of a tile that have just left neighbouring points of one
        // surface are a better group than anything the sort makes of them; chess 4K depth 5: 3.6 ms against 4.0)
        for (int64_t L = 2; L <= p.depth; L++) starts.push_back(L);
      }
    }
    for (size_t g = 0; g < starts.size(); g++) {
      const int64_t L = starts[g], L_next = g + 1 < starts.size() ? starts[g + 1] : p.depth + 1;
      if (L > 0) {
        if ((g & 7) == 0) {
          // a deep `depth`: every eighth launch the host looks whether any ray is left (the frames of the
          // configurations never get here)
          HIP_TRY(hipMemcpyAsync(s->h_staging, qc + kQcChunks + L, sizeof(uint32_t), hipMemcpyDeviceToHost, stream));
          HIP_TRY(hipStreamSynchronize(stream));
          if (*(const uint32_t*)s->h_staging == 0) break;
        }
        hipLaunchKernelGGL((levels_sort_kernel<false>), dim3(sort_blocks), dim3(kBlock), 0, stream, (const uint32_t*)s->d_qkeys,
                           (const uint32_t*)(qc + kQcChunks + L), (uint32_t)q_cap_chunks, n_keys, qc + kQcHist, (uint32_t*)nullptr);
        hipLaunchKernelGGL(levels_scan_kernel, dim3(1), dim3(kScanThreads), 0, stream, qc + kQcHist, qc + kQcCursor, n_keys, qc + kQcSorted + L);
        hipLaunchKernelGGL((levels_sort_kernel<true>), dim3(sort_blocks), dim3(kBlock), 0, stream, (const uint32_t*)s->d_qkeys,
                           (const uint32_t*)(qc + kQcChunks + L), (uint32_t)q_cap_chunks, n_keys, qc + kQcCursor, (uint32_t*)s->d_qsorted);
      }
      A.level = (uint32_t)L;
      A.level_push = (uint32_t)L_next;
      A.q_in = (const uint64_t*)s->d_queue[g & 1];
      A.q_in_count = qc + kQcSorted + L;
      A.q_out = (uint64_t*)s->d_queue[(g + 1) & 1];
      A.q_out_chunks = qc + kQcChunks + L_next;
      A.next_batch = (unsigned int*)d_counters + L;  // the claim counters of the launch: word L of each counter's 4 KB
      if (A.lds_ctx)
        hipLaunchKernelGGL((render_kernel<true, 3>), dim3((unsigned)grid), dim3(kBlock), lds_bytes, stream, d_leaves, d_nodes, d_lights,
                           d_code, d_consts, d_bounds, (const uint8_t*)s->d_prog, (const ShadeArgs*)(d_counters + kOffShadeArgs), A);
      else
        hipLaunchKernelGGL((render_kernel<false, 3>), dim3((unsigned)grid), dim3(kBlock), lds_bytes, stream, d_leaves, d_nodes, d_lights,
                           d_code, d_consts, d_bounds, (const uint8_t*)s->d_prog, (const ShadeArgs*)(d_counters + kOffShadeArgs), A);
      HIP_TRY(hipGetLastError());
    }
  } else
  if (A.lds_ctx && A.xchg)
    hipLaunchKernelGGL((render_kernel<true, 2>), dim3((unsigned)grid), dim3(kBlock), lds_bytes, stream, d_leaves, d_nodes, d_lights,
                       d_code, d_consts, d_bounds, (const uint8_t*)s->d_prog, (const ShadeArgs*)(d_counters + kOffShadeArgs), A);
  else if (A.lds_ctx && (A.n_tiles_shared || A.tail_lds_off || A.sm_claimed))
    hipLaunchKernelGGL((render_kernel<true, 1>), dim3((unsigned)grid), dim3(kBlock), lds_bytes, stream, d_leaves, d_nodes, d_lights,
                       d_code, d_consts, d_bounds, (const uint8_t*)s->d_prog, (const ShadeArgs*)(d_counters + kOffShadeArgs), A);
  else if (A.lds_ctx)
    hipLaunchKernelGGL((render_kernel<true, 0>), dim3((unsigned)grid), dim3(kBlock), lds_bytes, stream, d_leaves, d_nodes, d_lights,
                       d_code, d_consts, d_bounds, (const uint8_t*)s->d_prog, (const ShadeArgs*)(d_counters + kOffShadeArgs), A);
  else
    hipLaunchKernelGGL((render_kernel<false, 0>), dim3((unsigned)grid), dim3(kBlock), lds_bytes, stream, d_leaves, d_nodes, d_lights,
                       d_code, d_consts, d_bounds, (const uint8_t*)s->d_prog, (const ShadeArgs*)(d_counters + kOffShadeArgs), A);
  HIP_TRY(hipGetLastError());
  HIP_TRY(hipEventRecord((hipEvent_t)s->ev_stop, stream));
  // the surface-error word and, right behind it, the per-block ray counts
  const unsigned long long* per_block = (const unsigned long long*)((uint8_t*)s->h_staging + 1024);
  if (!zero_copy)
    HIP_TRY(hipMemcpyAsync((uint8_t*)s->h_staging + 1024 - 8, d_counters + kOffSurfError, 8 + (size_t)grid * 4 * sizeof(unsigned long long),
                           hipMemcpyDeviceToHost, stream));
  *(uint32_t*)((uint8_t*)s->h_staging + 1024 - 16) = 0;
  if (levels) HIP_TRY(hipMemcpyAsync((uint8_t*)s->h_staging + 1024 - 16, (uint32_t*)s->d_qcnt + kQcOverflow, sizeof(uint32_t), hipMemcpyDeviceToHost, stream));
  HIP_TRY(hipStreamSynchronize(stream));
  if (*(const uint32_t*)((const uint8_t*)s->h_staging + 1024 - 16) != 0) return "ray queue overflow";
  unsigned long long counts[4] = {0, 0, 0, 0};
  for (size_t b = 0; b < (size_t)grid; b++)
    for (int k = 0; k < 4; k++) counts[k] += per_block[b * 4 + k];
  float ms = 0.f;
  HIP_TRY(hipEventElapsedTime(&ms, (hipEvent_t)s->ev_start, (hipEvent_t)s->ev_stop));
  if (A.sm_claimed && std::getenv("GMLRT_VERBOSE")) {
    std::vector<uint32_t> c(kSmCntWords);
    HIP_TRY(hipMemcpy(c.data(), A.sm_cnt, c.size() * sizeof(uint32_t), hipMemcpyDeviceToHost));
    const unsigned long long st = *(const unsigned long long*)(c.data() + kSmStatWord);
#ifdef GMLRT_SM_TIMING
    std::fprintf(stderr, "[gmlrt] sampled hand-out: %llu searches for the next tile, %.2f us each\n", ((const unsigned long long*)(c.data() + kSmStatWord))[2],
                 (double)((const unsigned long long*)(c.data() + kSmStatWord))[1] / 100.0 / (double)std::max(1ull, ((const unsigned long long*)(c.data() + kSmStatWord))[2]));
#endif
    std::fprintf(stderr, "[gmlrt] sampled hand-out: %u samples of %u tiles, %llu finished, mean %.1f us; published / taken per class: %u / %u, %u / %u, %u / %u; %.3f ms\n",
                 A.sm_groups, A.n_tiles, st >> 40, (st >> 40) ? (double)(st & ((1ull << 40) - 1ull)) / (double)(st >> 40) / 100.0 : 0.0, c[0], c[kSmLine],
                 c[2 * kSmLine], c[3 * kSmLine], c[4 * kSmLine], c[5 * kSmLine], (double)ms);
  }
  if (no_alloc) {
    // nothing to learn and nothing to record
  } else if (s->order_state >= 1 && s->order_state <= 4 && way == s->order_state) {
    // a trial: the learnt order trades cache locality between neighbouring waves for a shorter tail (scenes
    // with evenly priced tiles lose); the batch table over the image order only saves atomics; sharing the
    // dearest tiles among the waves of a block costs total work and pays when those tiles set the launch's length
    // (ways 3 and 4 run other kernels than the launches before them: the first launch of a kernel in a process runs
    // 0.15 ms longer than its later ones, code that no cache holds yet, which is more than the difference between
    // the ways on a partition of a frame; they are timed twice and the better time counts)
    const bool again = (way == 3 || way == 4) && s->order_trials == 0;
    s->order_ms[way] = s->order_trials ? std::min(s->order_ms[way], ms) : ms;
    s->order_trials = again ? 1 : 0;
    if (verbose_shared && way == 3 && !s->learnt_cost.empty()) {
      std::vector<uint32_t> now(n_tiles), list(s->n_tiles_shared);
      HIP_TRY(hipMemcpy(now.data(), s->d_tile_cost, n_tiles * sizeof(uint32_t), hipMemcpyDeviceToHost));
      HIP_TRY(hipMemcpy(list.data(), s->d_tiles_shared, list.size() * sizeof(uint32_t), hipMemcpyDeviceToHost));
      double before = 0, after = 0;
      for (uint32_t t : list) { before += s->learnt_cost[t]; after += now[t]; }
      std::fprintf(stderr, "[gmlrt] shared tiles: %.1f us each alone, %.1f us each shared; dearest %.1f -> %.1f, median %.1f -> %.1f\n",
                   before / list.size() / 100.0, after / list.size() / 100.0, s->learnt_cost[list[0]] / 100.0, now[list[0]] / 100.0,
                   s->learnt_cost[list[list.size() / 2]] / 100.0, now[list[list.size() / 2]] / 100.0);
    }
    if (!again) s->order_state++;
    if (s->order_state == 3 && s->n_tiles_shared == 0) { s->order_state = 4; s->order_ms[3] = 0.f; }  // no tile dear enough: nothing to try
    // The kernel that regroups the rays of a block is on trial only on request: it has lost on every scene measured
    // (1.7 to 2.7 times slower: two block barriers a round put the four waves in step, and each round then lasts as
    // long as its slowest wave), so a handle does not spend a launch on it unasked.
    const bool regroup_rays = [] { const char* e = std::getenv("GMLRT_REGROUP_RAYS"); return e && *e == '1'; }();
    if (s->order_state == 4 && !(A.lds_ctx && regroup_rays)) { s->order_state = 5; s->order_ms[4] = 0.f; }
    if (s->order_state == 5) {
      s->order_way = 0;
      float best = 0.98f * s->order_ms[0];
      for (int k = 1; k <= 4; k++)
        if (s->order_ms[k] > 0.f && s->order_ms[k] < best) { best = s->order_ms[k]; s->order_way = k; }
      if (std::getenv("GMLRT_VERBOSE"))
        std::fprintf(stderr, "[gmlrt] tile hand-out: no costs known %.3f ms, learnt order %.3f ms, image order with batch table %.3f ms, "
                     "learnt order with %u shared tiles %.3f ms, learnt order with the rays of a block regrouped %.3f ms -> way %d\n",
                     s->order_ms[0], s->order_ms[1], s->order_ms[2], s->n_tiles_shared, s->order_ms[3], s->order_ms[4], s->order_way);
    }
  } else if (learning) {
    // first launch with this geometry: turn its per-tile costs into the order of the next ones
    // (worth it only when a wave renders several tiles)
    std::vector<uint32_t> cost(n_tiles), order(n_tiles);
    HIP_TRY(hipMemcpy(cost.data(), s->d_tile_cost, n_tiles * sizeof(uint32_t), hipMemcpyDeviceToHost));
    // Sorted by cost class only (a few classes up to the dearest tile), image order within a class:
    // neighbouring tiles walk the same parts of the scene program, and waves that run side by side
    // should keep sharing them in cache.
    static const uint32_t n_classes = [] { const char* e = std::getenv("GMLRT_TILE_CLASSES"); return e ? (uint32_t)std::atoi(e) : 16u; }();
    if (std::getenv("GMLRT_VERBOSE")) {
      std::vector<uint32_t> sorted(cost);
      std::sort(sorted.begin(), sorted.end());
      size_t at = 0;
      for (size_t i = 0; i < n_tiles; i++) if (cost[i] > cost[at]) at = i;
      unsigned long long sum = 0;
      for (uint32_t c : cost) sum += c;
      std::fprintf(stderr, "[gmlrt] tile cost (us, 100 MHz clock): p50 %.1f p90 %.1f p99 %.1f max %.1f (tile x %zu y %zu), sum %.0f over %zu tiles\n",
                   sorted[n_tiles / 2] / 100.0, sorted[n_tiles * 9 / 10] / 100.0, sorted[n_tiles * 99 / 100] / 100.0,
                   cost[at] / 100.0, at % A.tiles_x, at / A.tiles_x, (double)sum / 100.0, n_tiles);
      if (const char* watch = std::getenv("GMLRT_TILE_WATCH")) {  // "x,y;x,y;...": what these tiles cost
        for (const char* q = watch; *q;) {
          unsigned x = 0, y = 0;
          if (std::sscanf(q, "%u,%u", &x, &y) == 2 && x < A.tiles_x && (size_t)y * A.tiles_x + x < n_tiles)
            std::fprintf(stderr, "[gmlrt] tile x %u y %u: %.1f us\n", x, y, cost[(size_t)y * A.tiles_x + x] / 100.0);
          while (*q && *q != ';') q++;
          if (*q) q++;
        }
      }
#ifdef GMLRT_DIAG
      {
        std::vector<unsigned long long> dbg(n_tiles * 4);
        HIP_TRY(hipMemcpy(dbg.data(), d_tile_dbg, dbg.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost));
        std::vector<size_t> idx(n_tiles);
        for (size_t i = 0; i < n_tiles; i++) idx[i] = i;
        std::sort(idx.begin(), idx.end(), [&](size_t a, size_t b) { return cost[a] > cost[b]; });
        for (size_t r = 0; r < 12 && r < n_tiles; r++) {
          const size_t q = idx[r * (r < 6 ? 1 : n_tiles / 40)];
          std::fprintf(stderr, "[gmlrt diag] tile rank %zu (x %zu y %zu): %.1f us, traversals %llu, leaves %llu, merges %llu, redo %llu\n",
                       r < 6 ? r : r * (n_tiles / 40), q % A.tiles_x, q / A.tiles_x, cost[q] / 100.0, dbg[q * 4], dbg[q * 4 + 1],
                       dbg[q * 4 + 2], dbg[q * 4 + 3] & 0xFFFFull);
        }
        // and the tiles with the most rounds traced without bounds
        std::sort(idx.begin(), idx.end(), [&](size_t a, size_t b) { return (dbg[a * 4 + 3] & 0xFFFFull) > (dbg[b * 4 + 3] & 0xFFFFull); });
        unsigned long long redo_tiles = 0, redo_rounds = 0;
        for (size_t i = 0; i < n_tiles; i++) { redo_tiles += dbg[i * 4 + 3] != 0; redo_rounds += dbg[i * 4 + 3] & 0xFFFFull; }
        std::fprintf(stderr, "[gmlrt diag] %llu rounds without bounds in %llu tiles\n", redo_rounds, redo_tiles);
        unsigned long long gs[12];
        HIP_TRY(hipMemcpyFromSymbol(gs, HIP_SYMBOL(g_general_stats), sizeof(gs)));
        std::fprintf(stderr, "[gmlrt diag] full-list traversals so far: %llu, live leaves (lane 0's 64 words only) %llu, words visited %llu, merges %llu, marking %.1f us, walk %.1f us\n",
                     gs[0], gs[1], gs[2], gs[3], gs[4] / 100.0, gs[5] / 100.0);
        std::fprintf(stderr, "[gmlrt diag] of the walk: %llu leaves tested %.1f us, %llu operators with a list %.1f us, %llu operators without %.1f us\n",
                     gs[6] & 0xFFFFFFFFull, (gs[6] >> 32) / 100.0, gs[7] & 0xFFFFFFFFull, (gs[7] >> 32) / 100.0, gs[8] & 0xFFFFFFFFull, (gs[8] >> 32) / 100.0);
        for (size_t r = 0; r < 8 && r < n_tiles && dbg[idx[r] * 4 + 3]; r++) {
          const size_t q = idx[r];
          std::fprintf(stderr, "[gmlrt diag] tile (x %zu y %zu): %.1f us, traversals %llu, leaves %llu, merges %llu, redo %llu in %.1f us\n",
                       q % A.tiles_x, q / A.tiles_x, cost[q] / 100.0, dbg[q * 4], dbg[q * 4 + 1], dbg[q * 4 + 2], dbg[q * 4 + 3] & 0xFFFFull,
                       (double)(dbg[q * 4 + 3] >> 16) / 100.0);
        }
      }
#endif
    }
    if (verbose_shared) s->learnt_cost = cost;
    tiles_by_cost(cost, n_classes, &order);
    HIP_TRY(hipMemcpy(s->d_tile_order, order.data(), n_tiles * sizeof(uint32_t), hipMemcpyHostToDevice));
    {
      const uint64_t waves = (uint64_t)grid * (kBlock / kWave);
      std::vector<uint32_t> start;
      batches_by_cost(cost, &order, waves, (double)ms, &start);
      s->n_batches = (uint32_t)start.size() - 1;
      HIP_TRY(hipMemcpy(s->d_tile_batch, start.data(), start.size() * sizeof(uint32_t), hipMemcpyHostToDevice));
      batches_by_cost(cost, nullptr, waves, (double)ms, &start);
      s->n_batches_img = (uint32_t)start.size() - 1;
      HIP_TRY(hipMemcpy(s->d_tile_batch_img, start.data(), start.size() * sizeof(uint32_t), hipMemcpyHostToDevice));
      // Way 3: a tile that alone takes more than a share of what a wave has to do in the whole launch is rendered
      // by a block's four waves together, dearest first, before anything else; the rest keeps the learnt order.
      s->n_tiles_shared = 0;
      s->n_batches_rest = 0;
      // The threshold: sharing makes the dearest tile about 2.4 times shorter (measured; the per-ray code stays
      // serial and the slices never balance) at 4 waves' expense, so nothing cheaper than the dearest tile's
      // shared time is worth sharing, and nothing that a wave could do alone within its fair part of the launch.
      // (tuning runs and tests: a fixed fraction of a wave's fair part instead; 0 switches sharing off.  Read at every learning launch.)
      const double share = [] { const char* e = std::getenv("GMLRT_SHARED_TILE_SHARE"); return e ? std::atof(e) : -1.0; }();
      if (A.lds_ctx && s->slice_off[0] != 0 && share != 0.0) {
        unsigned long long total = 0;
        uint32_t top = 0;
        for (uint32_t c : cost) { total += c; top = std::max(top, c); }
        const double per_wave = (double)total / (double)waves;
        const double threshold = share > 0.0 ? share * per_wave : std::max((double)top / 2.4, per_wave);
        std::vector<uint32_t> dear, rest_cost;
        std::vector<uint32_t> keep;  // tiles that stay in the ordinary hand-out, in image order
        for (size_t i = 0; i < n_tiles; i++) {
          if ((double)cost[i] > threshold) dear.push_back((uint32_t)i);
          else keep.push_back((uint32_t)i);
        }
        // at most a few per block: beyond that the tiles are not what sets the length of the launch
        std::sort(dear.begin(), dear.end(), [&](uint32_t a, uint32_t b) { return cost[a] > cost[b] || (cost[a] == cost[b] && a < b); });
        const size_t cap = 4 * (size_t)grid;
        while (dear.size() > cap) { keep.push_back(dear.back()); dear.pop_back(); }
        if (!dear.empty() && !keep.empty()) {
          std::sort(keep.begin(), keep.end());
          rest_cost.resize(keep.size());
          for (size_t i = 0; i < keep.size(); i++) rest_cost[i] = cost[keep[i]];
          std::vector<uint32_t> rest_order;
          tiles_by_cost(rest_cost, n_classes, &rest_order);
          batches_by_cost(rest_cost, &rest_order, waves, (double)ms, &start);
          for (uint32_t& t : rest_order) t = keep[t];
          s->n_batches_rest = (uint32_t)start.size() - 1;
          s->n_tiles_shared = (uint32_t)dear.size();
          HIP_TRY(hipMemcpy(s->d_tile_order_rest, rest_order.data(), rest_order.size() * sizeof(uint32_t), hipMemcpyHostToDevice));
          HIP_TRY(hipMemcpy(s->d_tile_batch_rest, start.data(), start.size() * sizeof(uint32_t), hipMemcpyHostToDevice));
          HIP_TRY(hipMemcpy(s->d_tiles_shared, dear.data(), dear.size() * sizeof(uint32_t), hipMemcpyHostToDevice));
        }
      }
      if (std::getenv("GMLRT_VERBOSE"))
        std::fprintf(stderr, "[gmlrt] %zu tiles in %u batches (learnt order), %u batches (image order); %u tiles dear enough to share\n",
                     n_tiles, s->n_batches, s->n_batches_img, s->n_tiles_shared);
    }
    s->order_key = key;
    s->order_valid = true;
    s->order_state = 1;
    s->order_trials = 0;
    s->order_ms[0] = ms;
  }
#ifdef GMLRT_DIAG_SHARED
  if (A.n_tiles_shared) {
    unsigned long long v[16];
    HIP_TRY(hipMemcpy(v, (uint8_t*)d_counters + kOffDiag, sizeof(v), hipMemcpyDeviceToHost));
    for (int w = 0; w < 4; w++)
      std::fprintf(stderr, "[gmlrt shared] wave %d of a block, Mcycles over %u tiles: own slice %.1f, wait for candidates %.1f, per-ray code %.1f, wait for rays %.1f\n",
                   w, A.n_tiles_shared, v[w * 4] / 1e6, v[w * 4 + 1] / 1e6, v[w * 4 + 2] / 1e6, v[w * 4 + 3] / 1e6);
  }
#endif
#ifdef GMLRT_DIAG
  {
    unsigned long long dgv[16];
    HIP_TRY(hipMemcpy(dgv, (uint8_t*)s->d_counters + kOffDiag, 10 * sizeof(unsigned long long), hipMemcpyDeviceToHost));
    {
      std::vector<unsigned long long> per_wave(lanes / kWave * kDiagWaveSlots);
      HIP_TRY(hipMemcpy(per_wave.data(), d_diag_waves, per_wave.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost));
      (void)hipFree(d_diag_waves);
      unsigned long long vm_runs = 0, vm_lanes = 0;
      for (int q = 0; q < 6; q++) dgv[10 + q] = 0;
      for (size_t wv = 0; wv < lanes / kWave; wv++) {
        for (int q = 0; q < 6; q++) dgv[10 + q] += per_wave[wv * kDiagWaveSlots + q];
        vm_runs += per_wave[wv * kDiagWaveSlots + 6];
        vm_lanes += per_wave[wv * kDiagWaveSlots + 7];
      }
      {
        // when did each wave finish, relative to the first start (100 MHz wall clock)
        unsigned long long first = ~0ull, last = 0, sum = 0, tiles_max = 0, tiles_sum = 0;
        const size_t nw = lanes / kWave;
        std::vector<unsigned long long> ends;
        for (size_t wv = 0; wv < nw; wv++) first = std::min(first, per_wave[wv * kDiagWaveSlots + 8]);
        for (size_t wv = 0; wv < nw; wv++) {
          const unsigned long long e = per_wave[wv * kDiagWaveSlots + 9] - first;
          ends.push_back(e);
          last = std::max(last, e);
          sum += e;
          tiles_max = std::max(tiles_max, per_wave[wv * kDiagWaveSlots + 10]);
          tiles_sum += per_wave[wv * kDiagWaveSlots + 10];
        }
        std::sort(ends.begin(), ends.end());
        std::fprintf(stderr, "[gmlrt diag] wave end times (us): p10 %.0f p50 %.0f p90 %.0f p99 %.0f max %.0f mean %.0f; tiles per wave mean %.1f max %llu\n",
                     ends[nw / 10] / 100.0, ends[nw / 2] / 100.0, ends[nw * 9 / 10] / 100.0, ends[nw * 99 / 100] / 100.0, last / 100.0,
                     (double)sum / (double)nw / 100.0, (double)tiles_sum / (double)nw, tiles_max);
      }
      {
        // cycles of the traversal by record kind (the clock reads themselves are in there: shares, not timings)
        unsigned long long ck[8] = {}, nk[8] = {}, all = 0;
        for (size_t wv = 0; wv < lanes / kWave; wv++)
          for (int q = 0; q < 8; q++) { ck[q] += per_wave[wv * kDiagWaveSlots + 12 + q]; nk[q] += per_wave[wv * kDiagWaveSlots + 20 + q]; }
        for (int q = 0; q < 8; q++) all += ck[q];
        const char* kn[8] = {"runs of bounds", "leaves skipped by the look", "frontier leaves tested", "list leaves tested", "operators without a merge",
                             "operators with a merge", "skip-if-empty records", "take records"};
        for (int q = 0; q < 8; q++)
          std::fprintf(stderr, "[gmlrt diag] traversal cycles: %-28s %5.1f %%  (%llu times, %.0f cycles each)\n", kn[q], 100.0 * (double)ck[q] / (double)(all ? all : 1),
                       nk[q], (double)ck[q] / (double)(nk[q] ? nk[q] : 1));
      }
      std::fprintf(stderr, "[gmlrt diag] surface program runs %llu (%.2f per traversal), lanes wanting one %.1f per traversal\n",
                   vm_runs, (double)vm_runs / ((double)dgv[3] > 0 ? (double)dgv[3] : 1.0),
                   (double)vm_lanes / ((double)dgv[3] > 0 ? (double)dgv[3] : 1.0));
    }
    {
      unsigned long long ex[9];
      HIP_TRY(hipMemcpy(ex, (uint8_t*)s->d_counters + kOffDiag + 16 * 8, 8 * sizeof(ex[0]), hipMemcpyDeviceToHost));
      ex[8] = dgv[6] - ex[6] - ex[7];
      const double entered = (double)(dgv[4] - dgv[5]) > 0 ? (double)(dgv[4] - dgv[5]) : 1.0, nl = (double)dgv[6] > 0 ? (double)dgv[6] : 1.0;
      std::fprintf(stderr, "[gmlrt diag] lanes whose ray meets an entered bound %.1f; per leaf test: lanes taking part %.1f, with an Entry %.1f\n",
                   (double)ex[0] / entered, (double)ex[1] / nl, (double)ex[2] / nl);
      const char* names[3] = {"camera", "shadow of camera hits", "after a bounce"};
      for (int q = 0; q < 3; q++)
        std::fprintf(stderr, "[gmlrt diag] waves tracing %-22s: %llu traversals, %.1f leaves each\n", names[q], ex[3 + q],
                     (double)ex[6 + q] / ((double)ex[3 + q] > 0 ? (double)ex[3 + q] : 1.0));
    }
    if (std::getenv("GMLRT_VERBOSE")) {
      std::vector<unsigned long long> ld((size_t)H.n_leaves * kLeafDbg);
      HIP_TRY(hipMemcpy(ld.data(), d_leaf_dbg, ld.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost));
      std::vector<gmlrt_leaf> hl(H.n_leaves);
      HIP_TRY(hipMemcpy(hl.data(), d_leaves, hl.size() * sizeof(gmlrt_leaf), hipMemcpyDeviceToHost));
      unsigned long long by_type[5][kLeafDbg] = {};
      for (uint32_t i = 0; i < H.n_leaves; i++) for (int q = 0; q < kLeafDbg; q++) by_type[hl[i].type][q] += ld[(size_t)i * kLeafDbg + q];
      const char* tn[5] = {"sphere", "cube", "cylinder", "cone", "plane"};
      for (int t = 0; t < 5; t++)
        std::fprintf(stderr, "[gmlrt diag] %-8s leaf tests %llu, without an Entry in any lane %.1f %%, lanes with an Entry %.1f; skipped by the look before the test %.1f %%, lanes wrongly turned away %llu\n", tn[t], by_type[t][0],
                     100.0 * (double)by_type[t][1] / ((double)by_type[t][0] > 0 ? (double)by_type[t][0] : 1.0),
                     (double)by_type[t][2] / ((double)by_type[t][0] > 0 ? (double)by_type[t][0] : 1.0),
                     100.0 * (double)by_type[t][3] / ((double)by_type[t][0] > 0 ? (double)by_type[t][0] : 1.0), by_type[t][4]);
      if (std::getenv("GMLRT_DIAG_LEAVES"))
        for (uint32_t i = 0; i < H.n_leaves; i++)
          std::fprintf(stderr, "[gmlrt diag] leaf %u %s: tests %llu nohit %llu entries %llu\n", i, tn[hl[i].type], ld[(size_t)i * kLeafDbg], ld[(size_t)i * kLeafDbg + 1], ld[(size_t)i * kLeafDbg + 2]);
      (void)hipFree(d_leaf_dbg);
    }
    {
      unsigned long long ex2[8], nohit;
      HIP_TRY(hipMemcpy(ex2, (uint8_t*)s->d_counters + kOffDiag + 24 * 8, 8 * sizeof(ex2[0]), hipMemcpyDeviceToHost));
      HIP_TRY(hipMemcpy(&nohit, (uint8_t*)s->d_counters + kOffDiag + 10 * 8, sizeof(nohit), hipMemcpyDeviceToHost));
      std::fprintf(stderr, "[gmlrt diag] leaf tests in which no lane has an Entry: %.1f %%\n", 100.0 * (double)nohit / ((double)dgv[6] > 0 ? (double)dgv[6] : 1.0));
      const char* names[3] = {"camera", "shadow of camera hits", "after a bounce"};
      for (int q = 0; q < 3; q++)
        std::fprintf(stderr, "[gmlrt diag] waves tracing %-22s: entered bounds %llu, lanes whose ray meets one %.1f\n", names[q], ex2[q],
                     (double)ex2[3 + q] / ((double)ex2[q] > 0 ? (double)ex2[q] : 1.0));
      std::fprintf(stderr, "[gmlrt diag] active lanes per traversal: camera waves %.1f, shadow-of-camera waves %.1f\n",
                   (double)ex2[6] / ((double)dgv[3] > 0 ? 129600.0 : 1.0), (double)ex2[7] / 73852.0);
    }
    const double nt = (double)dgv[3] > 0 ? (double)dgv[3] : 1.0;
    std::fprintf(stderr,
                 "[gmlrt diag] wave-traversals %llu (avg lanes %.1f); per traversal: bounds %.1f (skipped %.1f) leaves %.1f "
                 "merges %.2f takes %.1f; cycles/traversal: refill %.0f traverse %.0f after %.0f\n",
                 dgv[3], (double)dgv[9] / nt, (double)dgv[4] / nt, (double)dgv[5] / nt, (double)dgv[6] / nt, (double)dgv[7] / nt,
                 (double)dgv[8] / nt, (double)dgv[0] / nt, (double)dgv[1] / nt, (double)dgv[2] / nt);
    std::fprintf(stderr,
                 "[gmlrt diag] after, cycles/traversal: hit geometry %.0f surface vm %.0f shade+store %.0f shadow result %.0f "
                 "next ray %.0f finish %.0f\n",
                 (double)dgv[10] / nt, (double)dgv[11] / nt, (double)dgv[12] / nt, (double)dgv[13] / nt, (double)dgv[14] / nt,
                 (double)dgv[15] / nt);
  }
#endif
  if (stats) {
    stats->rays_primary = counts[0];
    stats->rays_shadow = counts[1];
    stats->rays_reflect = counts[2];
    stats->hits = counts[3];
    stats->kernel_ms = (double)ms;
    stats->tiles_shared = A.n_tiles_shared;
    stats->handout_way = (uint32_t)way;
    stats->surface_errors = *(const uint32_t*)((const uint8_t*)s->h_staging + 1024 - 8) ? 1u : 0u;
  }
  return "";
}

}  // namespace gmlrt
