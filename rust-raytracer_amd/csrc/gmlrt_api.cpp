// C ABI of include/gmlrt.h: argument checking, scene handles, the in-process
// multi-GPU driver (one host thread per GPU, interleaved row tiles, final gather
// into the caller's framebuffer) and the GML program front-end handle.
#include <chrono>
#include <condition_variable>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "../../include/gmlrt.h"
#include "gml_front.hpp"
#include "gmlrt_device.hpp"
#include "gmlrt_scene.hpp"

namespace {

thread_local std::string g_last_error;
int fail(int code, const std::string& msg) {
  g_last_error = msg;
  return code;
}

std::mutex g_mu;
int g_n_gpus = 0;  // 0 = not initialised

// what validate_blob found, as a status: malformed, or well-formed but beyond the kernel's limits
int blob_status(const std::string& v) {
  if (v.empty()) return GMLRT_OK;
  return fail(v.rfind("too large:", 0) == 0 ? GMLRT_ERR_SCENE_TOO_LARGE : GMLRT_ERR_BAD_BLOB, v);
}

// Nothing may leave an extern "C" function as an exception (the derivation of the device programs
// throws on a program too large for its offsets, any container on allocation failure).
template <class F>
int guarded(F&& body) {
  try {
    return body();
  } catch (gmlrt::SceneTooLarge& e) {
    return fail(GMLRT_ERR_SCENE_TOO_LARGE, e.what());
  } catch (std::bad_alloc&) {
    return fail(GMLRT_ERR_SCENE_TOO_LARGE, "out of host memory");
  } catch (std::exception& e) {
    return fail(GMLRT_ERR_HIP, e.what());
  }
}

}  // namespace

struct gmlrt_scene {
  gmlrt::DeviceScene dev;
};

struct gmlrt_program {
  struct Call {
    gmlrt_params params;
    std::string file;
    std::vector<uint8_t> blob;
  };
  std::vector<Call> calls;
};

// ---- the host-buffer entry points: one worker per GPU partition, with what it keeps between calls
namespace {
struct Worker {
  // the scene uploaded last by this worker: a caller that renders the same scene again (same bytes) gets the
  // handle back with everything derived from the blob, and from then on the handle learns its tile order
  gmlrt::DeviceScene ds;
  std::vector<uint8_t> blob;
  int device = -1;
  int uses = 0;
};
std::mutex g_render_mu;         // the host-buffer entry points run one at a time
std::vector<Worker*> g_workers;  // by partition index; under g_render_mu

void drop_scene(Worker* W) {
  if (W->device >= 0) gmlrt::dev_scene_destroy(&W->ds);
  W->ds = gmlrt::DeviceScene{};
  W->blob.clear();
  W->device = -1;
  W->uses = 0;
}

// Scene handle of worker W for `blob` on `device` ("" on success).
std::string worker_scene(Worker* W, const void* blob, size_t blob_bytes, int device) {
  if (W->device == device && W->blob.size() == blob_bytes && std::memcmp(W->blob.data(), blob, blob_bytes) == 0) {
    W->uses++;
    W->ds.learn_order = true;  // rendered more than once: learning the hand-out order pays
    return "";
  }
  drop_scene(W);
  std::string e = gmlrt::dev_scene_create(blob, blob_bytes, device, &W->ds);
  if (!e.empty()) {
    gmlrt::dev_scene_destroy(&W->ds);
    W->ds = gmlrt::DeviceScene{};
    return e;
  }
  W->blob.assign((const uint8_t*)blob, (const uint8_t*)blob + blob_bytes);
  W->device = device;
  W->uses = 1;
  W->ds.learn_order = false;  // as far as anyone knows this handle renders once
  return "";
}

// Number of partitions a frame is rendered in, and the GPUs behind them.
// GMLRT_EMULATE_PARTS=N splits the frame into N partitions however many GPUs there are (partition g runs on GPU
// g mod n): the multi-GPU path on a box with one GPU, for tests.
void partitions(int64_t h, int32_t tile_rows, int opened, int* n_parts, int* real_gpus) {
  *real_gpus = opened;
  int n = opened;
  if (const char* e = std::getenv("GMLRT_EMULATE_PARTS")) {
    const int v = std::atoi(e);
    if (v >= 1 && v <= 64) n = v;
  }
  const int64_t n_tiles = (h + tile_rows - 1) / tile_rows;  // no point in giving a GPU nothing to do
  if ((int64_t)n > n_tiles) n = (int)n_tiles;
  *n_parts = n;
}

int surface_failure() {
  return fail(GMLRT_ERR_SURFACE, "a surface function divided by zero or indexed an array out of range on a shaded hit "
                                 "(the reference aborts there; the image holds 0 / the first element instead)");
}
void add_stats(gmlrt_stats* total, const gmlrt_stats& st, bool sum_kernel_ms) {
  total->surface_errors |= st.surface_errors;
  total->tiles_shared += st.tiles_shared;
  if (st.handout_way > total->handout_way) total->handout_way = st.handout_way;
  total->rays_primary += st.rays_primary;
  total->rays_shadow += st.rays_shadow;
  total->rays_reflect += st.rays_reflect;
  total->hits += st.hits;
  if (sum_kernel_ms) total->kernel_ms += st.kernel_ms;
  else if (st.kernel_ms > total->kernel_ms) total->kernel_ms = st.kernel_ms;
}

template <class F>
std::string no_throw(F&& body) {
  try {
    return body();
  } catch (gmlrt::SceneTooLarge& e) {
    return std::string("too large: ") + e.what();
  } catch (std::bad_alloc&) {
    return "too large: out of host memory";
  } catch (std::exception& e) {
    return e.what();
  }
}
int status_of(const std::string& e) {
  return fail(e.rfind("too large:", 0) == 0 ? GMLRT_ERR_SCENE_TOO_LARGE : GMLRT_ERR_HIP, e);
}
void release_workers() {
  std::lock_guard<std::mutex> lk(g_render_mu);
  for (Worker* W : g_workers) {
    drop_scene(W);
    delete W;
  }
  g_workers.clear();
}
}  // namespace


extern "C" {

const char* gmlrt_strerror(int status) {
  switch (status) {
    case GMLRT_OK: return "ok";
    case GMLRT_ERR_INVALID_ARG: return "invalid argument";
    case GMLRT_ERR_BAD_BLOB: return "scene blob failed validation";
    case GMLRT_ERR_NO_DEVICE: return "no usable GPU";
    case GMLRT_ERR_HIP: return "HIP runtime call failed";
    case GMLRT_ERR_GML: return "GML program error";
    case GMLRT_ERR_UNSUPPORTED_SURFACE: return "surface function outside the supported bytecode subset";
    case GMLRT_ERR_IO: return "I/O error";
    case GMLRT_ERR_SCENE_TOO_LARGE: return "scene too large";
    case GMLRT_ERR_NOT_INITIALISED: return "gmlrt_init has not been called";
    case GMLRT_ERR_SINK: return "a renderer callback reported failure";
    case GMLRT_ERR_SURFACE: return "a surface function failed at run time (division by zero or array index out of range)";
    default: return "unknown status";
  }
}

const char* gmlrt_last_error(void) { return g_last_error.c_str(); }

int64_t gmlrt_part_rows(int64_t height, int32_t tile_rows, int32_t part, int32_t nparts) {
  if (height <= 0) return 0;
  if (nparts <= 1) return height;
  if (tile_rows <= 0 || part < 0 || part >= nparts) return 0;
  const int64_t n_tiles = (height + tile_rows - 1) / tile_rows;
  int64_t rows = 0;
  for (int64_t t = part; t < n_tiles; t += nparts) {
    const int64_t y0 = t * tile_rows;
    rows += (y0 + tile_rows <= height) ? tile_rows : (height - y0);
  }
  return rows;
}

int gmlrt_part_row_index(int64_t height, int32_t tile_rows, int32_t part, int32_t nparts, int64_t* rows_out) {
  if (!rows_out || height < 0) return fail(GMLRT_ERR_INVALID_ARG, "gmlrt_part_row_index: bad argument");
  if (nparts <= 1) {
    for (int64_t y = 0; y < height; y++) rows_out[y] = y;
    return GMLRT_OK;
  }
  if (tile_rows <= 0 || part < 0 || part >= nparts) return fail(GMLRT_ERR_INVALID_ARG, "gmlrt_part_row_index: bad partition");
  const int64_t n_tiles = (height + tile_rows - 1) / tile_rows;
  int64_t k = 0;
  for (int64_t t = part; t < n_tiles; t += nparts)
    for (int64_t y = t * tile_rows; y < (t + 1) * tile_rows && y < height; y++) rows_out[k++] = y;
  return GMLRT_OK;
}

int gmlrt_device_count(void) {
  int n = 0;
  std::string e = gmlrt::dev_count(&n);
  if (!e.empty()) { g_last_error = e; return 0; }
  return n;
}

int gmlrt_init(int n_gpus) {
  std::lock_guard<std::mutex> lk(g_mu);
  int avail = 0;
  std::string e = gmlrt::dev_count(&avail);
  if (!e.empty() || avail <= 0) return fail(GMLRT_ERR_NO_DEVICE, e.empty() ? "no HIP device visible" : e);
  if (n_gpus < 0 || n_gpus > avail) return fail(GMLRT_ERR_INVALID_ARG, "gmlrt_init: more GPUs requested than visible");
  g_n_gpus = n_gpus == 0 ? avail : n_gpus;
  return GMLRT_OK;
}

void gmlrt_shutdown(void) {
  release_workers();
  std::lock_guard<std::mutex> lk(g_mu);
  g_n_gpus = 0;
  gmlrt::dev_release_cached();
}

static int check_params(const gmlrt_params* p) {
  if (!p) return fail(GMLRT_ERR_INVALID_ARG, "params is NULL");
  if (p->width <= 0 || p->height <= 0 || p->width > (1 << 20) || p->height > (1 << 20))
    return fail(GMLRT_ERR_INVALID_ARG, "image size out of range");
  if (p->depth < 0 || p->depth > 1024) {
    // a negative depth never bounces (raytracer.rs:100 tests depth > 0); treat it as 0 below
    if (p->depth > 1024) return fail(GMLRT_ERR_INVALID_ARG, "depth above 1024");
  }
  if (p->nparts > 1 && (p->tile_rows <= 0 || p->part < 0 || p->part >= p->nparts))
    return fail(GMLRT_ERR_INVALID_ARG, "bad row partition");
  return GMLRT_OK;
}

int gmlrt_scene_create(const void* blob, size_t blob_bytes, int device, gmlrt_scene** out) {
  if (!out) return fail(GMLRT_ERR_INVALID_ARG, "out is NULL");
  *out = nullptr;
  return guarded([&]() -> int {
  if (int rc = blob_status(gmlrt::validate_blob(blob, blob_bytes))) return rc;
  int avail = 0;
  std::string e = gmlrt::dev_count(&avail);
  if (!e.empty() || avail <= 0) return fail(GMLRT_ERR_NO_DEVICE, e.empty() ? "no HIP device visible" : e);
  if (device < 0 || device >= avail) return fail(GMLRT_ERR_INVALID_ARG, "device index out of range");
  auto s = std::make_unique<gmlrt_scene>();
  e = gmlrt::dev_scene_create(blob, blob_bytes, device, &s->dev);
  if (!e.empty()) {
    gmlrt::dev_scene_destroy(&s->dev);
    return fail(GMLRT_ERR_HIP, e);
  }
  *out = s.release();
  return GMLRT_OK;
  });
}

void gmlrt_scene_destroy(gmlrt_scene* s) {
  if (!s) return;
  gmlrt::dev_scene_destroy(&s->dev);
  delete s;
}

static int render_device(gmlrt_scene* s, const gmlrt_params* params, void* d_rgb8, void* d_rgbf64, void* hip_stream,
                         gmlrt_stats* stats, bool no_alloc) {
  if (!s || !d_rgb8) return fail(GMLRT_ERR_INVALID_ARG, "scene or output pointer is NULL");
  int rc = check_params(params);
  if (rc != GMLRT_OK) return rc;
  gmlrt_params p = *params;
  if (p.depth < 0) p.depth = 0;
  const auto t0 = std::chrono::steady_clock::now();
  gmlrt_stats st;
  std::memset(&st, 0, sizeof(st));
  return guarded([&]() -> int {
    std::string e = gmlrt::dev_render(&s->dev, p, d_rgb8, d_rgbf64, hip_stream, &st, no_alloc);
    if (!e.empty()) return fail(e.rfind("too large:", 0) == 0 ? GMLRT_ERR_SCENE_TOO_LARGE : GMLRT_ERR_HIP, e);
    st.total_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    st.n_gpus = 1;
    if (stats) *stats = st;
    return st.surface_errors ? surface_failure() : GMLRT_OK;
  });
}

int gmlrt_copy_rows_to_host(int device, const void* d_rows, void* image, size_t row_bytes, int64_t height,
                            int32_t tile_rows, int32_t part, int32_t nparts) {
  if (!d_rows || !image || row_bytes == 0 || height <= 0 || device < 0)
    return fail(GMLRT_ERR_INVALID_ARG, "gmlrt_copy_rows_to_host: bad argument");
  if (nparts > 1 && (tile_rows <= 0 || part < 0 || part >= nparts)) return fail(GMLRT_ERR_INVALID_ARG, "bad row partition");
  return guarded([&]() -> int {
    // (worker slots from 128 up: the in-process workers of gmlrt_render use the low ones)
    std::string e = gmlrt::dev_copy_rows_to_host(128 + device, device, image, d_rows, row_bytes, height, tile_rows > 0 ? tile_rows : 8,
                                                 nparts > 1 ? part : 0, nparts > 1 ? nparts : 1);
    return e.empty() ? GMLRT_OK : fail(GMLRT_ERR_HIP, e);
  });
}

int gmlrt_render_device(gmlrt_scene* s, const gmlrt_params* params, void* d_rgb8, void* d_rgbf64, void* hip_stream,
                        gmlrt_stats* stats) {
  return render_device(s, params, d_rgb8, d_rgbf64, hip_stream, stats, false);
}

int gmlrt_render_device_noalloc(gmlrt_scene* s, const gmlrt_params* params, void* d_rgb8, void* d_rgbf64,
                                void* hip_stream, gmlrt_stats* stats) {
  return render_device(s, params, d_rgb8, d_rgbf64, hip_stream, stats, true);
}

int gmlrt_render(const void* blob, size_t blob_bytes, const gmlrt_params* params, uint8_t* rgb8, double* rgbf64,
                 gmlrt_stats* stats) {
  if (!rgb8) return fail(GMLRT_ERR_INVALID_ARG, "rgb8 is NULL");
  int rc = check_params(params);
  if (rc != GMLRT_OK) return rc;
  int opened;
  {
    std::lock_guard<std::mutex> lk(g_mu);
    opened = g_n_gpus;
  }
  if (opened <= 0) return fail(GMLRT_ERR_NOT_INITIALISED, "call gmlrt_init first");
  if (int brc = blob_status(gmlrt::validate_blob(blob, blob_bytes))) return brc;
  std::lock_guard<std::mutex> render_lock(g_render_mu);
  const auto t0 = std::chrono::steady_clock::now();
  const int64_t w = params->width, h = params->height;
  const int32_t tile_rows = params->tile_rows > 0 ? params->tile_rows : 8;
  int n_parts, real_gpus;
  partitions(h, tile_rows, opened, &n_parts, &real_gpus);
  while ((int)g_workers.size() < n_parts) g_workers.push_back(new Worker());

  struct PerPart {
    std::string err;
    gmlrt_stats st;
  };
  std::vector<PerPart> per((size_t)n_parts);
  const bool verbose = std::getenv("GMLRT_VERBOSE") != nullptr;
  auto work = [&](int g) {
    const int dev = g % real_gpus;
    PerPart& P = per[(size_t)g];
    std::memset(&P.st, 0, sizeof(P.st));
    Worker* W = g_workers[(size_t)g];
    gmlrt_params p = *params;
    if (p.depth < 0) p.depth = 0;
    p.tile_rows = tile_rows;
    p.part = g;
    p.nparts = n_parts;
    const int64_t rows = gmlrt_part_rows(h, tile_rows, g, n_parts);
    auto tick = [&, last = std::chrono::steady_clock::now()](const char* what) mutable {
      if (!verbose) return;
      const auto now = std::chrono::steady_clock::now();
      std::fprintf(stderr, "[gmlrt] part %d %s: %.2f ms\n", g, what, std::chrono::duration<double, std::milli>(now - last).count());
      last = now;
    };
    P.err = no_throw([&]() -> std::string {
      std::string e = worker_scene(W, blob, blob_bytes, dev);
      if (!e.empty()) return e;
      tick(W->uses > 1 ? "scene (kept from the last call)" : "scene upload");
      const size_t n8 = (size_t)rows * (size_t)w * 3;
      void *d8 = nullptr, *df = nullptr;
      e = gmlrt::dev_output_buffers(g, dev, n8, rgbf64 ? n8 * sizeof(double) : 0, &d8, &df);
      if (!e.empty()) return e;
      tick("output buffers");
      e = gmlrt::dev_render(&W->ds, p, d8, df, nullptr, &P.st);
      if (!e.empty()) return e;
      tick("render (workspace + kernel)");
      // the final gather: each partition's rows go to their place in the caller's framebuffer
      e = gmlrt::dev_copy_rows_to_host(g, dev, rgb8, d8, (size_t)w * 3, h, tile_rows, g, n_parts);
      if (!e.empty()) return e;
      if (rgbf64) e = gmlrt::dev_copy_rows_to_host(g, dev, rgbf64, df, (size_t)w * 3 * sizeof(double), h, tile_rows, g, n_parts);
      tick("copy to host");
      return e;
    });
    if (!P.err.empty()) drop_scene(W);
  };
  if (n_parts == 1) {
    work(0);
  } else {
    std::vector<std::thread> th;
    for (int g = 0; g < n_parts; g++) th.emplace_back(work, g);
    for (auto& t : th) t.join();
  }
  gmlrt_stats total;
  std::memset(&total, 0, sizeof(total));
  for (auto& P : per) {
    if (!P.err.empty()) return status_of(P.err);
    add_stats(&total, P.st, false);
  }
  total.total_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
  total.n_gpus = (uint32_t)n_parts;
  if (stats) *stats = total;
  return total.surface_errors ? surface_failure() : GMLRT_OK;
}

// The frame is rendered in bands of whole "rounds" (one row tile per partition), top to bottom.  Each partition
// has a producer thread that renders its share of band k + 1 while the share of band k travels to page-locked
// host memory and the calling thread hands band k - 1 to the sink, pixel by pixel in scanline order: the caller
// sees its first pixels after one band, and the host never holds more than two bands per partition.
int gmlrt_render_to_sink(const void* blob, size_t blob_bytes, const gmlrt_params* params, const char* filename,
                         const gmlrt_renderer* sink, gmlrt_stats* stats) {
  if (!sink || !sink->new_image || !sink->push_pixel) return fail(GMLRT_ERR_INVALID_ARG, "sink needs new_image and push_pixel");
  int rc = check_params(params);
  if (rc != GMLRT_OK) return rc;
  int opened;
  {
    std::lock_guard<std::mutex> lk(g_mu);
    opened = g_n_gpus;
  }
  if (opened <= 0) return fail(GMLRT_ERR_NOT_INITIALISED, "call gmlrt_init first");
  if (int brc = blob_status(gmlrt::validate_blob(blob, blob_bytes))) return brc;
  std::lock_guard<std::mutex> render_lock(g_render_mu);
  const auto t0 = std::chrono::steady_clock::now();
  const int64_t w = params->width, h = params->height;
  const int32_t tile_rows = params->tile_rows > 0 ? params->tile_rows : 8;
  int n_parts, real_gpus;
  partitions(h, tile_rows, opened, &n_parts, &real_gpus);
  while ((int)g_workers.size() < n_parts) g_workers.push_back(new Worker());

  // band height: about 2^20 pixels per partition and band (a few milliseconds of kernel on a heavy scene, a
  // 25 MB host buffer), in whole rounds; GMLRT_SINK_BAND_ROWS overrides the rows per partition (tests)
  const int64_t round_rows = (int64_t)tile_rows * n_parts;
  int64_t part_rows_per_band = ((1ll << 20) / w / tile_rows) * tile_rows;
  if (const char* e = std::getenv("GMLRT_SINK_BAND_ROWS")) part_rows_per_band = (std::atoll(e) / tile_rows) * tile_rows;
  if (part_rows_per_band < tile_rows) part_rows_per_band = tile_rows;
  const int64_t rounds_per_band = part_rows_per_band / tile_rows;
  const int64_t n_rounds = (h + round_rows - 1) / round_rows;
  const int64_t n_bands = (n_rounds + rounds_per_band - 1) / rounds_per_band;

  struct Part {
    gmlrt::BandPipe* pipe = nullptr;
    std::string err;
    gmlrt_stats st;
    int64_t issued = 0;    // bands whose copy has been queued
    int64_t consumed = 0;  // bands the sink is done with
  };
  std::vector<Part> parts((size_t)n_parts);
  std::mutex mu;
  std::condition_variable cv;
  bool stop = false;
  for (auto& P : parts) std::memset(&P.st, 0, sizeof(P.st));

  auto produce = [&](int g) {
    const int dev = g % real_gpus;
    Part& P = parts[(size_t)g];
    Worker* W = g_workers[(size_t)g];
    gmlrt_params p = *params;
    if (p.depth < 0) p.depth = 0;
    p.tile_rows = tile_rows;
    p.part = g;
    p.nparts = n_parts;
    const int64_t rows_all = gmlrt_part_rows(h, tile_rows, g, n_parts);
    std::string err = no_throw([&]() -> std::string {
      std::string e = worker_scene(W, blob, blob_bytes, dev);
      if (!e.empty()) return e;
      W->ds.learn_order = false;  // every band is a geometry of its own
      e = gmlrt::band_pipe_create(dev, (size_t)part_rows_per_band * (size_t)w, &P.pipe);
      if (!e.empty()) return e;
      for (int64_t b = 0; b < n_bands; b++) {
        {
          std::unique_lock<std::mutex> lk(mu);
          cv.wait(lk, [&] { return stop || P.consumed + 2 > b; });  // slot b % 2 held band b - 2
          if (stop) return "";
        }
        const int64_t row0 = b * part_rows_per_band;
        const int64_t rows = std::max<int64_t>(0, std::min(part_rows_per_band, rows_all - row0));
        if (rows > 0) {
          gmlrt_stats st;
          std::memset(&st, 0, sizeof(st));
          e = gmlrt::band_pipe_render(P.pipe, &W->ds, p, (int)(b & 1), row0, rows, &st);
          if (!e.empty()) return e;
          add_stats(&P.st, st, true);
        }
        {
          std::lock_guard<std::mutex> lk(mu);
          P.issued = b + 1;
        }
        cv.notify_all();
      }
      return "";
    });
    if (!err.empty()) {
      std::lock_guard<std::mutex> lk(mu);
      P.err = err;
      stop = true;
    }
    cv.notify_all();
  };

  std::vector<std::thread> th;
  for (int g = 0; g < n_parts; g++) th.emplace_back(produce, g);

  int result = GMLRT_OK;
  std::string sink_err;
  if (sink->new_image(sink->user, filename ? filename : "", w, h) != 0) {
    result = GMLRT_ERR_SINK;
    sink_err = "new_image failed";
  }
  for (int64_t b = 0; b < n_bands && result == GMLRT_OK; b++) {
    // the band's share of every partition must have arrived
    std::vector<const double*> host((size_t)n_parts, nullptr);
    for (int g = 0; g < n_parts && result == GMLRT_OK; g++) {
      Part& P = parts[(size_t)g];
      {
        std::unique_lock<std::mutex> lk(mu);
        cv.wait(lk, [&] { return stop || P.issued > b; });
        if (stop) { result = GMLRT_ERR_HIP; break; }
      }
      const int64_t row0 = b * part_rows_per_band;
      if (gmlrt_part_rows(h, tile_rows, g, n_parts) > row0) {
        std::string e = gmlrt::band_pipe_wait(P.pipe, (int)(b & 1), &host[(size_t)g]);
        if (!e.empty()) { result = GMLRT_ERR_HIP; sink_err = e; }
      }
    }
    if (result != GMLRT_OK) break;
    // scanline order over the band's rounds: round r holds one row tile of every partition, in partition order
    const int64_t y_begin = b * rounds_per_band * round_rows;
    const int64_t y_end = std::min(h, (b + 1) * rounds_per_band * round_rows);
    for (int64_t y = y_begin; y < y_end && result == GMLRT_OK; y++) {
      const int64_t tile = y / tile_rows;
      const int g = (int)(tile % n_parts);
      const int64_t packed = (tile / n_parts) * tile_rows + y % tile_rows - b * part_rows_per_band;  // row inside the band's buffer
      const double* row = host[(size_t)g] + (size_t)packed * (size_t)w * 3;
      for (int64_t x = 0; x < w; x++)
        if (sink->push_pixel(sink->user, row + x * 3) != 0) {
          result = GMLRT_ERR_SINK;
          sink_err = "push_pixel failed";
          break;
        }
    }
    {
      std::lock_guard<std::mutex> lk(mu);
      for (auto& P : parts) P.consumed = b + 1;
    }
    cv.notify_all();
  }
  {
    std::lock_guard<std::mutex> lk(mu);
    if (result != GMLRT_OK) stop = true;
  }
  cv.notify_all();
  for (auto& t : th) t.join();
  gmlrt_stats total;
  std::memset(&total, 0, sizeof(total));
  std::string dev_err;
  for (int g = 0; g < n_parts; g++) {
    Part& P = parts[(size_t)g];
    gmlrt::band_pipe_destroy(P.pipe);
    if (!P.err.empty()) {
      dev_err = P.err;
      drop_scene(g_workers[(size_t)g]);
    }
    // partitions render side by side: the frame's kernel time is the slowest partition's sum over its bands
    add_stats(&total, P.st, false);
  }
  if (!dev_err.empty()) return status_of(dev_err);
  if (result == GMLRT_ERR_SINK) return fail(GMLRT_ERR_SINK, sink_err);
  if (result != GMLRT_OK) return fail(result, sink_err.empty() ? "render stopped" : sink_err);
  if (sink->done && sink->done(sink->user) != 0) return fail(GMLRT_ERR_SINK, "done failed");
  total.total_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
  total.n_gpus = (uint32_t)n_parts;
  if (stats) *stats = total;
  return total.surface_errors ? surface_failure() : GMLRT_OK;
}

int gmlrt_write_ppm(const char* path, const uint8_t* rgb8, int64_t w, int64_t h) {
  if (!path || !rgb8 || w <= 0 || h <= 0) return fail(GMLRT_ERR_INVALID_ARG, "gmlrt_write_ppm: bad argument");
  FILE* f = std::fopen(path, "wb");
  if (!f) return fail(GMLRT_ERR_IO, std::string("cannot create ") + path);
  bool ok = std::fprintf(f, "P6 %lld %lld 255\n", (long long)w, (long long)h) > 0;
  const size_t n = (size_t)w * (size_t)h * 3;
  ok = ok && std::fwrite(rgb8, 1, n, f) == n;
  ok = (std::fclose(f) == 0) && ok;
  return ok ? GMLRT_OK : fail(GMLRT_ERR_IO, std::string("short write to ") + path);
}

// ------------------------------------------------------------------ GML programs
static int program_open(const std::string& text, gmlrt_program** out) {
  auto prog = std::make_unique<gmlrt_program>();
  int rc = GMLRT_OK;
  std::string msg;
  try {
    gml::run_with_big_stack([&] {
      gml::Evaluator ev([&](const gml::RenderCall& rc_) {
        gmlrt_program::Call c;
        std::memset(&c.params, 0, sizeof(c.params));
        for (int i = 0; i < 3; i++) c.params.amb[i] = rc_.amb[(size_t)i];
        c.params.depth = rc_.depth;
        c.params.fov_deg = rc_.fov;
        c.params.width = rc_.w;
        c.params.height = rc_.h;
        c.params.tile_rows = 8;
        c.params.part = 0;
        c.params.nparts = 1;
        c.file = rc_.file;
        c.blob = gmlrt::flatten_scene(rc_);
        prog->calls.push_back(std::move(c));
      });
      gml::Env env;
      gml::Stack st;
      ev.run(text, env, st);
    });
  } catch (gmlrt::UnsupportedSurface& e) {
    rc = GMLRT_ERR_UNSUPPORTED_SURFACE;
    msg = e.what();
  } catch (gmlrt::SceneTooLarge& e) {
    rc = GMLRT_ERR_SCENE_TOO_LARGE;
    msg = e.what();
  } catch (std::bad_alloc&) {
    rc = GMLRT_ERR_SCENE_TOO_LARGE;
    msg = "out of host memory";
  } catch (std::exception& e) {
    rc = GMLRT_ERR_GML;
    msg = e.what();
  }
  if (rc != GMLRT_OK) return fail(rc, msg);
  *out = prog.release();
  return GMLRT_OK;
}

int gmlrt_program_open_file(const char* path, gmlrt_program** out) {
  if (!path || !out) return fail(GMLRT_ERR_INVALID_ARG, "gmlrt_program_open_file: NULL argument");
  *out = nullptr;
  std::string text;
  try {
    text = gml::preprocess_file(path);
  } catch (std::exception& e) {
    return fail(GMLRT_ERR_IO, e.what());
  }
  return program_open(text, out);
}

int gmlrt_program_open_text(const char* gml_text, gmlrt_program** out) {
  if (!gml_text || !out) return fail(GMLRT_ERR_INVALID_ARG, "gmlrt_program_open_text: NULL argument");
  *out = nullptr;
  return program_open(gml_text, out);
}

void gmlrt_program_close(gmlrt_program* p) { delete p; }

int gmlrt_program_num_renders(const gmlrt_program* p) { return p ? (int)p->calls.size() : 0; }

int gmlrt_program_render_call(const gmlrt_program* p, int idx, gmlrt_params* params, const char** filename,
                              const void** blob, size_t* blob_bytes) {
  if (!p || idx < 0 || idx >= (int)p->calls.size()) return fail(GMLRT_ERR_INVALID_ARG, "render call index out of range");
  const auto& c = p->calls[(size_t)idx];
  if (params) *params = c.params;
  if (filename) *filename = c.file.c_str();
  if (blob) *blob = c.blob.data();
  if (blob_bytes) *blob_bytes = c.blob.size();
  return GMLRT_OK;
}

}  // extern "C"
