#!/usr/bin/env python3
"""Wall time of the host-buffer entry point (gmlrt_render: upload, launch, copy back into pageable memory) for the
headline frame, next to its kernel time: as a one-shot render (another scene rendered in between, so that the library
has nothing of this one left) and repeated (the library keeps the last scene's handle, which learns its tile order).
Usage on the GPU box: python tools/host_path_timing.py"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "rust-raytracer_amd"))
import gmlrt
call = gmlrt.Program(path=os.path.join(ROOT, "scenes", "chess.gml")).calls[0].with_overrides(3840, 2160, 5)
other = gmlrt.Program(path=os.path.join(ROOT, "scenes", "spheres.gml")).calls[0]
gmlrt.render(call)  # process start-up: runtime, code object, workspace
for want in (False, True):
    for mode in ("one-shot", "repeated"):
        for i in range(4):
            if mode == "one-shot":
                gmlrt.render(other)
            t0 = time.perf_counter()
            img, f64, st = gmlrt.render(call, want_f64=want)
            wall = (time.perf_counter() - t0) * 1e3
            print(f"f64={want} {mode} run {i}: wall {wall:.2f} ms (library total {st.total_ms:.2f} ms, kernel {st.kernel_ms:.2f} ms), "
                  f"{st.rays / wall / 1e3:.0f} Mrays/s end to end", flush=True)
