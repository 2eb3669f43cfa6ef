// Internal interface between the C ABI layer (gmlrt_api.cpp) and the HIP
// translation unit (gmlrt_kernels.hip).
#pragma once
#include <cstddef>
#include <cstdint>
#include <string>

#include "../../include/gmlrt.h"

namespace gmlrt {

// Device-resident copy of one scene blob plus the per-device scratch the render
// kernel needs (per-ray event stacks, shading contexts, counters).
struct DeviceScene {
  int device = -1;
  void* d_blob = nullptr;
  size_t blob_bytes = 0;
  gmlrt_blob_header hdr{};
  // workspace, grown on demand and kept for the life of the handle
  void* d_ev_dist = nullptr;
  void* d_ev_meta = nullptr;
  void* d_ctx = nullptr;
  void* d_counters = nullptr;
  size_t ev_slots_alloc = 0;    // event slots per lane currently allocated
  size_t ctx_fields_alloc = 0;  // f64 fields per lane currently allocated
  size_t lanes_alloc = 0;
  void* ev_start = nullptr;     // hipEvent_t pair used to time the kernel
  void* ev_stop = nullptr;
};

// Each returns "" on success or a description of the HIP failure.
std::string dev_count(int* n);
std::string dev_scene_create(const void* blob, size_t bytes, int device, DeviceScene* out);
void dev_scene_destroy(DeviceScene* s);
// Launches the render kernel for the partition in `p` and waits for it.
// d_rgb8 / d_rgbf are device pointers (rgbf may be null).
std::string dev_render(DeviceScene* s, const gmlrt_params& p, void* d_rgb8, void* d_rgbf, void* stream,
                       gmlrt_stats* stats);
// Plain device buffers for the host-buffer entry point.
std::string dev_alloc(int device, size_t bytes, void** out);
void dev_free(int device, void* p);
std::string dev_copy_to_host(int device, void* dst, const void* src, size_t bytes);

}  // namespace gmlrt
