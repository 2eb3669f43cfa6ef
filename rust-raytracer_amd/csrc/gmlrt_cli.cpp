// gmlrt — command-line runner, the counterpart of the reference's `rungml`
// (/root/reference/src/rungml/src/main.rs:49-84): evaluates a GML file (with its
// #include lines) and renders every `render` call on the GPU(s) through the C ABI.
// Unlike the reference at this snapshot it does write the PPM the program names.
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/gmlrt.h"

static void usage(const char* prog) {
  std::printf(
      "Usage: %s FILE [options]\n"
      "  -n, --no-render      evaluate only\n"
      "  --width N --height N --depth N   override the operands of every render call\n"
      "  --gpus N             GPUs to use (default: all visible)\n"
      "  --out-dir DIR        directory for the output files (default: .)\n"
      "  -h, --help\n",
      prog);
}

int main(int argc, char** argv) {
  std::string input, out_dir = ".";
  long width = 0, height = 0, depth = -1;
  int gpus = 0;
  bool no_render = false;
  for (int i = 1; i < argc; i++) {
    std::string a = argv[i];
    auto need = [&](const char* name) -> const char* {
      if (i + 1 >= argc) { std::fprintf(stderr, "%s needs a value\n", name); std::exit(2); }
      return argv[++i];
    };
    if (a == "-h" || a == "--help") { usage(argv[0]); return 0; }
    else if (a == "-n" || a == "--no-render") no_render = true;
    else if (a == "--width") width = std::atol(need("--width"));
    else if (a == "--height") height = std::atol(need("--height"));
    else if (a == "--depth") depth = std::atol(need("--depth"));
    else if (a == "--gpus") gpus = std::atoi(need("--gpus"));
    else if (a == "--out-dir") out_dir = need("--out-dir");
    else if (input.empty()) input = a;
    else { std::fprintf(stderr, "unexpected argument %s\n", a.c_str()); return 2; }
  }
  if (input.empty()) { usage(argv[0]); return 0; }

  gmlrt_program* prog = nullptr;
  int rc = gmlrt_program_open_file(input.c_str(), &prog);
  if (rc != GMLRT_OK) {
    std::fprintf(stderr, "%s: %s: %s\n", input.c_str(), gmlrt_strerror(rc), gmlrt_last_error());
    return 1;
  }
  const int n = gmlrt_program_num_renders(prog);
  if (no_render || n == 0) { gmlrt_program_close(prog); return 0; }
  rc = gmlrt_init(gpus);
  if (rc != GMLRT_OK) {
    std::fprintf(stderr, "gmlrt_init: %s: %s\n", gmlrt_strerror(rc), gmlrt_last_error());
    return 1;
  }
  for (int i = 0; i < n; i++) {
    gmlrt_params p;
    const char* file;
    const void* blob;
    size_t bytes;
    gmlrt_program_render_call(prog, i, &p, &file, &blob, &bytes);
    if (width > 0) p.width = width;
    if (height > 0) p.height = height;
    if (depth >= 0) p.depth = depth;
    std::vector<uint8_t> img((size_t)p.width * (size_t)p.height * 3);
    gmlrt_stats st;
    rc = gmlrt_render(blob, bytes, &p, img.data(), nullptr, &st);
    if (rc != GMLRT_OK) {
      std::fprintf(stderr, "render %d: %s: %s\n", i, gmlrt_strerror(rc), gmlrt_last_error());
      return 1;
    }
    const std::string path = out_dir + "/" + file;
    rc = gmlrt_write_ppm(path.c_str(), img.data(), p.width, p.height);
    if (rc != GMLRT_OK) {
      std::fprintf(stderr, "%s: %s\n", gmlrt_strerror(rc), gmlrt_last_error());
      return 1;
    }
    const double rays = (double)(st.rays_primary + st.rays_shadow + st.rays_reflect);
    std::printf("%s  %lldx%lld depth %lld  gpus %u  rays %.0f (primary %llu shadow %llu reflect %llu)  kernel %.3f ms  %.1f Mrays/s\n",
                path.c_str(), (long long)p.width, (long long)p.height, (long long)p.depth, st.n_gpus, rays,
                (unsigned long long)st.rays_primary, (unsigned long long)st.rays_shadow,
                (unsigned long long)st.rays_reflect, st.kernel_ms, st.kernel_ms > 0 ? rays / st.kernel_ms / 1e3 : 0.0);
  }
  gmlrt_shutdown();
  gmlrt_program_close(prog);
  return 0;
}
