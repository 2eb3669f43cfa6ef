// Internal interface between the C ABI layer (gmlrt_api.cpp) and the HIP
// translation unit (gmlrt_kernels.hip).
#pragma once
#include <cstddef>
#include <cstdint>
#include <string>

#include "../../include/gmlrt.h"

#include <vector>

namespace gmlrt {

// Records of the derived traversal program (gmlrt_program.cpp): a byte stream of
// 16-byte-aligned records, operands inline.  `kind` = record kind in bits 3..0,
// kRecFlag in bit 8, a leaf's primitive type in bits 18..16.
constexpr uint32_t kRecEnd = 0, kRecBound = 1, kRecLeaf = 2, kRecOp = 3, kRecTake = 4, kRecBox = 5, kRecSkipEmpty = 6;
constexpr uint32_t kOpSwapped = 0x10u;  // operator arg: the right operand was pushed first (ties still go to the left one)
constexpr uint32_t kRecFlag = 0x100u;  // bound: under the union frontier (a skip pushes no list); leaf: frontier leaf
// One bit per record kind, beside the kind itself: the traversal asks for these bits one after the other, most frequent
// kind first (a chain of tests of the kind would be turned into one dispatch by the compiler, with state flags and
// a common exit that every record then walks through).  kRecEnd has none.
constexpr uint32_t kRecIsBox = 0x200u, kRecIsLeaf = 0x400u, kRecIsOp = 0x800u, kRecIsTake = 0x1000u, kRecIsSkip = 0x2000u,
                   kRecIsBound = 0x4000u;
struct ProgBound {   // 64 bytes
  uint32_t kind;
  uint32_t skip_to;  // byte offset of the first record after the bounded subtree
  double center[3];
  double radius2;
  double radius;
  double pad[2];     // to 64 bytes: a record fetch that straddles two cache lines is two requests
};
struct ProgBox {     // 64 bytes: the same as ProgBound with the subtree's box instead of its sphere
  uint32_t kind;
  uint32_t skip_to;
  // The box in single precision, rounded outwards, as three (lo, hi) pairs, and per axis the larger magnitude
  // of the two: the test runs in f32 with an explicit bound on its own rounding error (it only has to be
  // conservative, never reproducible).
  float lohi[6];     // lo.x hi.x lo.y hi.y lo.z hi.z
  float mag[3];      // max(|lo|, |hi|) per axis, rounded up
  uint32_t pad[5];
};
struct ProgLeaf {    // 128 bytes
  uint32_t kind;
  uint32_t leaf;     // index into the blob's leaf section
  double inv_m[12];
  uint64_t pad;
  uint64_t pad2[2];  // to 128 bytes
};
struct ProgSmall {   // 16 bytes: operator (arg = rule | kOpSwapped), take, end, skip-if-empty (arg = byte offset to jump to)
  uint32_t kind;
  uint32_t arg;
  uint64_t pad;
};
static_assert(sizeof(ProgBox) == 64 && offsetof(ProgBox, skip_to) == offsetof(ProgBound, skip_to) && sizeof(ProgBound) == 64 && sizeof(ProgLeaf) == 128 && sizeof(ProgSmall) == 16, "record sizes");
// need_stack / need_events: operand-stack levels and event slots per ray this program's evaluation order can
// reach (it may differ from the blob's postfix order, for which the blob header states them).
// slice_off[kProgSlices]: byte offsets of the frontier's slices, programs of their own behind the main one for
// the waves of a block that trace the same rays together (all zero: the scene has no frontier worth slicing).
constexpr uint32_t kProgSlices = 4;
std::vector<uint8_t> derive_stream_program(const void* blob, uint32_t* need_stack, uint64_t* need_events, uint32_t* slice_off);
// Expanded surface programs (SurfInstr records, gmlrt_surface_vm.h): every leaf's program
// specialised for each face its primitive can report, identical results stored once;
// leaf_surf[leaf * kSurfFaces + face] = index of the first record of the program to run for a
// hit on that face, or kNoSurfProg for a constant surface.
constexpr uint32_t kNoSurfProg = 0xFFFFFFFFu;
constexpr uint32_t kSurfFaces = 8;  // slots per leaf (a cube has six faces)
std::vector<uint8_t> derive_surface_stream(const void* blob, std::vector<uint32_t>* leaf_surf);

// Device-resident copy of one scene blob plus the per-device scratch the render
// kernel needs (per-ray event stacks, shading contexts, counters).
struct DeviceScene {
  int device = -1;
  void* d_blob = nullptr;
  size_t blob_bytes = 0;
  void* d_prog = nullptr;       // derived program (record stream)
  uint32_t n_prog = 0;          // its size in bytes
  uint32_t slice_off[kProgSlices] = {0, 0, 0, 0};  // where the frontier's slices start in it (zeros: none)
  uint32_t max_stack = 1;       // operand-stack levels / event slots per ray: the larger of what the blob states for its
  uint64_t max_events = 2;      // postfix order (full-list traversal) and what the derived program's order needs
  uint32_t n_vm_regs = 0;       // registers the scene's surface bytecode uses
  void* d_surf = nullptr;       // expanded surface programs
  void* d_leaf_surf = nullptr;  // u32 per leaf: first record of its program
  void* d_axes = nullptr;       // directions in which a cylinder's test leaves its general case (gmlrt_kernels.hip: cylinder_special_case)
  uint32_t n_axes = 0;
  std::vector<double> h_axes;   // host copy of d_axes (the first classes travel in the launch record as well)
  uint32_t n_cone_forms = 0;    // quadratic forms, behind the cylinders' classes in d_axes: directions along a cone's mantle (cone_special_case)
  gmlrt_blob_header hdr{};
  // workspace, grown on demand and kept for the life of the handle
  void* d_ev_dist = nullptr;
  void* d_ev_meta = nullptr;
  void* d_ctx = nullptr;
  void* d_counters = nullptr;
  size_t ev_slots_alloc = 0;    // event slots currently allocated (lanes x slots per lane)
  size_t ctx_fields_alloc = 0;  // f64 context fields currently allocated (lanes x fields per lane)
  // hand-out order of the pixel tiles, learnt from the first launch with a given geometry
  void* d_tile_cost = nullptr;  // u32 per tile, written by every launch
  void* d_tile_order = nullptr; // u32 per tile: tiles sorted by decreasing cost
  void* d_tile_batch = nullptr; // u32 per batch + 1: first hand-out of each batch of that order
  uint32_t n_batches = 0;
  size_t tiles_alloc = 0;
  bool order_valid = false;     // d_tile_order holds an order for order_key
  bool learn_order = true;      // false for a handle that renders once: learning would never pay
  // Ways of handing out the tiles (6, the sampled hand-out of launches that know no costs and learn none, is described
  // in gmlrt_kernels.hip): 0 image order, fixed small batches; 1 the learnt order with its batch
  // table; 2 image order with a batch table of its own; 3 the learnt order with the dearest tiles taken out and
  // rendered first, each by the four waves of a block together (only when the launch has tiles dear enough to set
  // its length); 4 the learnt order, run by the kernel that sorts the rays of a block into its waves before every
  // traversal.  order_state counts the launches with this geometry: 0 nothing learnt (runs way 0 and times its
  // tiles), 1 .. 4: that way on trial, 5: decided (order_way).
  int order_state = 0;
  int order_way = 0;            // the fastest of them (a trial has to beat way 0 by 2 %)
  float order_ms[5] = {0.f, 0.f, 0.f, 0.f, 0.f};  // kernel time of the launch that ran each way (0: not tried)
  int order_trials = 0;         // launches the way on trial has had (ways 3 and 4 get two: see where they are timed)
  void* d_tile_batch_img = nullptr;      // way 2's batch table
  uint32_t n_batches_img = 0;
  void* d_tile_order_rest = nullptr;     // way 3: order and batch table of the tiles that are not shared, and the shared ones
  void* d_tile_batch_rest = nullptr;
  void* d_tiles_shared = nullptr;
  uint32_t n_batches_rest = 0, n_tiles_shared = 0;
  void* d_sampled = nullptr;             // sampled hand-out (way 6, launches that know no costs): a word per tile, three lists of tiles, counters
  void* d_xchg = nullptr;                // way 4: ray states on their way between lanes, 16 KB a block
  uint64_t xchg_blocks = 0;
  std::vector<uint32_t> learnt_cost;     // verbose runs only: the costs the tables were made from
  gmlrt_params order_key{};     // the launch parameters the order belongs to (ambient zeroed)
  void* h_staging = nullptr;    // page-locked host memory for the launch record and the per-block ray counts
  int dev_cus = 0;              // device facts, asked once: compute units, LDS per block and per unit,
  size_t dev_lds_block = 0, dev_lds_cu = 0;
  int dev_blocks_per_cu = 0;    // resident blocks per unit for this scene's LDS need
  // Rendering level by level (hand-out way 5): bounce rays leave their lane for a device-wide queue and the next
  // launch traces them regrouped.  Two ray queues used in turn, the keys and the sorted positions of the queued rays,
  // one parked (c, sc, ks) record per pixel and bounce level, counters; the scene's box for the sort keys.
  void* d_queue[2] = {nullptr, nullptr};
  void* d_qkeys = nullptr;
  void* d_qsorted = nullptr;
  void* d_parked = nullptr;
  void* d_qcnt = nullptr;
  size_t queue_cap = 0;           // ray records each queue holds
  size_t parked_records = 0;      // parked records allocated (pixels x levels)
  double key_lo[3] = {0, 0, 0};   // sort keys: a grid over this box ...
  double key_inv[3] = {0, 0, 0};  // ... with this many cells per unit on each axis
  bool levels_refused = false;    // the buffers of way 5 could not be had for this handle's frames: do not ask again
  void* d_static_scratch = nullptr;  // this device's copy of the library's static scratch array (never freed)
  void* ev_start = nullptr;     // hipEvent_t pair used to time the kernel
  void* ev_stop = nullptr;
};

// Each returns "" on success or a description of the HIP failure.
std::string dev_count(int* n);
std::string dev_scene_create(const void* blob, size_t bytes, int device, DeviceScene* out);
void dev_scene_destroy(DeviceScene* s);
// Launches the render kernel for the partition in `p` and waits for it.
// d_rgb8 / d_rgbf are device pointers (rgbf may be null).
// no_alloc: use the library's static scratch array instead of the handle's allocated workspace and make no
// allocation of any kind (fewer lanes in flight when the array holds fewer; no learnt tile order).
// band_row0 / band_rows: render only that range of the partition's packed rows (band_rows < 0: to the end);
// the output buffers then hold the band alone.
std::string dev_render(DeviceScene* s, const gmlrt_params& p, void* d_rgb8, void* d_rgbf, void* stream,
                       gmlrt_stats* stats, bool no_alloc = false, int64_t band_row0 = 0, int64_t band_rows = -1);
// Plain device buffers (tests, tools).
std::string dev_alloc(int device, size_t bytes, void** out);
void dev_free(int device, void* p);
std::string dev_copy_to_host(int device, void* dst, const void* src, size_t bytes);
// The host-buffer entry point runs one worker per GPU partition; `worker` names its cached resources (device
// output buffers, page-locked staging), which live until dev_release_cached.  rgbf_bytes = 0: no f64 output.
std::string dev_output_buffers(int worker, int device, size_t rgb8_bytes, size_t rgbf_bytes, void** d8, void** df);
// Copies the packed rows one partition rendered (device) to their places in the full image (pageable host memory),
// through the worker's page-locked staging: partition `part` of `nparts` owns the row tiles part, part + nparts, ...
// of `tile_rows` rows each.
std::string dev_copy_rows_to_host(int worker, int device, void* image, const void* d_rows, size_t row_bytes, int64_t height,
                                  int32_t tile_rows, int32_t part, int32_t nparts);
// Streaming a frame to the host band by band: two device buffers and two page-locked host buffers per GPU, so that
// band k + 1 renders while band k is copied and band k - 1 is handed to the caller's pixel sink.
struct BandPipe;
std::string band_pipe_create(int device, size_t band_pixels, BandPipe** out);  // *out is set even on failure: destroy it
void band_pipe_destroy(BandPipe* bp);
// Renders rows [row0, row0 + rows) of the partition (packed rows) into slot 0 or 1 and queues the copy of the f64
// pixels into that slot's host buffer.  Call only when the caller has finished with the slot's previous contents.
std::string band_pipe_render(BandPipe* bp, DeviceScene* s, const gmlrt_params& p, int slot, int64_t row0, int64_t rows,
                             gmlrt_stats* stats);
// Waits until the slot's host buffer holds the band queued last; may be called from another thread than render.
std::string band_pipe_wait(BandPipe* bp, int slot, const double** host);
// Frees what dev_scene_destroy keeps for the next scene on the same device (scratch buffers, events).
void dev_release_cached(void);

// Directions of rays for which the reference's cylinder and cone leave geometry (primitives.rs:644-663, 778-792,
// 844-858) and report hits outside the solid: such rays must meet no bound.  kAxisWords doubles a class.  An axis
// class: unit vector x y z, the cosine at or above which a unit direction counts as parallel to it, the cosine at or
// below which as orthogonal (a value no cosine reaches switches a test off).  A cone form: q00 q11 q22 2q01 2q02 2q12 of
// the symmetric form Q = M^T diag(1, -1, 1) M / trace(M^T M) (M: the linear part of the cone's inverse transform),
// then the bound: a unit direction d with |d^T Q d| <= bound counts as running along the mantle.
constexpr uint32_t kAxisWords = 8;
#ifndef GMLRT_CONE_FORM_BOUND
#define GMLRT_CONE_FORM_BOUND 1e-8
#endif
constexpr double kConeFormBound = GMLRT_CONE_FORM_BOUND;
void special_ray_classes(const void* blob, std::vector<double>* classes, uint32_t* n_axes, uint32_t* n_cone_forms);

}  // namespace gmlrt
