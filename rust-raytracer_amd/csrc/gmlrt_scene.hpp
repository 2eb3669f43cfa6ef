// Host-side scene flattening: turns the operands of one `render` call
// (gml::RenderCall) into the contiguous blob declared in include/gmlrt.h.
//   - every leaf primitive with its finished (m, inv_m) pair
//   - the CSG tree as a postfix program
//   - the lights with their constructor-derived fields
//   - every surface closure compiled to straight-line bytecode
#pragma once
#include <string>
#include <vector>

#include "../../include/gmlrt.h"
#include "gml_front.hpp"

namespace gmlrt {

struct UnsupportedSurface : std::runtime_error {
  using std::runtime_error::runtime_error;
};

struct SceneTooLarge : std::runtime_error {
  using std::runtime_error::runtime_error;
};
// Limits of the kernel's packed per-lane fields: lights (11 bits of the lane state) and events in
// one list of the operand stack (15 bits of its length word).
constexpr uint32_t kMaxLights = 2047;
constexpr uint64_t kMaxListEvents = 32767;

// Result of compiling one surface closure (evaluator.rs:776-792 protocol).
struct SurfaceProgram {
  bool constant = false;           // body never reads face/u/v: consts[0..6) = r g b kd ks n
  std::vector<uint64_t> code;      // bytecode words, ends with GMLRT_SOP_END
  std::vector<uint64_t> consts;    // 8-byte constants (f64 or i64 bit patterns)
};

// Symbolically executes the closure with (face,u,v) unknown.  Both arms of a
// data-dependent `if` are evaluated and joined with SELECT, so the result has no
// control flow.  Throws UnsupportedSurface / gml::GmlError.
SurfaceProgram compile_surface(const gml::Value& closure);

// Host runs of the bytecode (used by unit tests to check the compiler against the
// tree-walking evaluator).  run_surface_program goes the way the device does: it
// expands the words into records with inline constants and interprets those;
// run_surface_words interprets the blob's words directly.
// *trapped (optional): the evaluation raised GMLRT_SOP_TRAP, i.e. the reference would have aborted on it.
void run_surface_program(const SurfaceProgram& p, int64_t face, double u, double v, double out[6], bool* trapped = nullptr);
void run_surface_words(const SurfaceProgram& p, int64_t face, double u, double v, double out[6], bool* trapped = nullptr);

// Expansion of one (code, constants) pair: appends 32-byte SurfInstr records (END
// included) to `out`.  `code` points at the program's first word, `consts` at the
// surface's first constant.
void expand_surface(const uint64_t* code, const uint64_t* consts, std::vector<uint8_t>& out);

// The program `records` (SurfInstr records up to and including END) with `face` known:
// constants propagated, operations on constants folded, dead code dropped.  Appends the
// result (END included) to `out`.  What the device runs for a hit on that face.
void specialize_surface(const uint8_t* records, int64_t face, std::vector<uint8_t>& out);
// Host run of the specialised program (unit tests: must equal run_surface_program bit for bit).
void run_surface_specialized(const SurfaceProgram& p, int64_t face, double u, double v, double out[6], bool* trapped = nullptr);

// Flattens one render call.  Throws UnsupportedSurface / gml::GmlError.
std::vector<uint8_t> flatten_scene(const gml::RenderCall& rc);

// Structural validation of a blob received over the C ABI.  Returns "" if fine; a message that
// starts with "too large:" means a well-formed scene beyond the kernel's limits.
std::string validate_blob(const void* blob, size_t bytes);

}  // namespace gmlrt
