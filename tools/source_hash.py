#!/usr/bin/env python3
"""First 16 hex digits of the SHA-256 over the sources that decide what the render kernel executes (the HIP
translation unit, its headers, and the host code that derives the device programs).  profiles/*_pmc_summary.json
records it; bench.py replays a summary's counters only when the hash of the tree it runs from is the same."""
import hashlib, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FILES = ["gmlrt_kernels.hip", "gmlrt_traverse.hpp", "gmlrt_shade.hpp", "gmlrt_prims.hpp", "gmlrt_surface_vm.h",
         "gmlrt_device.hpp", "gmlrt_program.cpp", "gmlrt_scene.cpp"]


def source_hash():
    h = hashlib.sha256()
    for f in FILES:
        h.update(open(os.path.join(ROOT, "rust-raytracer_amd", "csrc", f), "rb").read())
    return h.hexdigest()[:16]


if __name__ == "__main__":
    print(source_hash())
