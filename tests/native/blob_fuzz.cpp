// Mutates valid scene blobs (bit flips, byte writes, truncations) and runs what gmlrt_scene_create runs on the host:
// the validator, then, if it accepts, the derivation of the traversal and surface programs.  Built with ASan/UBSan.
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <iterator>
#include <random>
#include <vector>
#include "gmlrt_device.hpp"
#include "gmlrt_scene.hpp"
int main(int argc, char** argv) {
  std::ifstream f(argv[1], std::ios::binary);
  std::vector<uint8_t> b0((std::istreambuf_iterator<char>(f)), std::istreambuf_iterator<char>());
  const int n = std::atoi(argv[2]);
  std::mt19937_64 rng(std::atoi(argv[3]));
  long accepted = 0, rejected = 0, threw = 0;
  for (int it = 0; it < n; it++) {
    std::vector<uint8_t> b = b0;
    const int kind = (int)(rng() % 4);
    const int edits = 1 + (int)(rng() % 4);
    for (int e = 0; e < edits; e++) {
      const size_t at = (kind == 3 ? rng() % 256 : rng() % b.size());  // kind 3: the header
      if (kind == 0) b[at] ^= (uint8_t)(1u << (rng() % 8));
      else if (kind == 1) b[at] = (uint8_t)rng();
      else if (kind == 2) { uint32_t v = (uint32_t)rng(); if (rng() % 2) v = 0xFFFFFFFFu >> (rng() % 32); std::memcpy(&b[at & ~(size_t)3], &v, std::min<size_t>(4, b.size() - (at & ~(size_t)3))); }
      else b[at] = (uint8_t)rng();
    }
    size_t bytes = b.size();
    if (rng() % 8 == 0) bytes = rng() % (b.size() + 1);
    std::vector<uint8_t> exact(b.begin(), b.begin() + (long)bytes);  // (own allocation: reads past the end are caught)
    const std::string err = gmlrt::validate_blob(exact.data(), exact.size());
    if (!err.empty()) { rejected++; continue; }
    accepted++;
    try {
      uint32_t ns; uint64_t ne; uint32_t so[4];
      auto p = gmlrt::derive_stream_program(exact.data(), &ns, &ne, so);
      std::vector<uint32_t> ls;
      auto q = gmlrt::derive_surface_stream(exact.data(), &ls);
      (void)p; (void)q;
    } catch (const std::exception&) { threw++; }
  }
  std::printf("%s: %d mutants, %ld accepted by the validator (%ld of them refused by the derivation), %ld rejected\n", argv[1], n, accepted, threw, rejected);
}
