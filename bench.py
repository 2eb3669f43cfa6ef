#!/usr/bin/env python3
"""Headline benchmark: chess.gml at 3840x2160, depth 5, through the C ABI on N GPUs.

A step is one full render of the frame: every rank renders its interleaved row
tiles on its own GPU (scene and output buffers already resident in HBM), then the
rows are gathered to rank 0 (the only collective).  Rank 0 prints one JSON line.

  python bench.py --gpus 1 --steps 10 --warmup 2
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
         --master-port P bench.py --gpus N --steps K --warmup W

The other configurations (parity-test cases; a scene's own depth unless --depth is given):
  python bench.py --scene scenes/fractal.gml                      (likewise cone-fractal, spotlight, house, spheres)
  python bench.py --gpus 8 --scene scenes/large.gml --width 7680 --height 4320 --depth 8
"""
import argparse
import ctypes
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(ROOT, "rust-raytracer_amd"))

import numpy as np  # noqa: E402
import torch  # noqa: E402  (before gmlrt: one HIP runtime per process)
import torch.distributed as dist  # noqa: E402

import gmlrt  # noqa: E402
import rowtiles  # noqa: E402

SCENE = os.path.join(ROOT, "scenes", "chess.gml")
WIDTH, HEIGHT, DEPTH = 3840, 2160, 5
TILE_ROWS = 8

# Algorithmic flops per primitive test (FMA = 2, sqrt/div = 1, affine 3x4 transforms), including the
# 45-flop common prologue: sphere, cube, cylinder, cone, plane.  Shading: ~150 per shaded hit.
F_PRIM = {0: 61, 1: 60, 2: 70, 3: 77, 4: 47}
F_SHADE = 150
PEAK_FP64_VALU_TFLOPS = 78.6   # MI355X vector FP64 (the path computes in f64)
PEAK_FP32_VALU_TFLOPS = 157.3  # MI355X vector FP32, the figure the project metric names
PMC_SUMMARIES = os.path.join(ROOT, "profiles", "r03_*_pmc_summary.json")


def measured_counters(workload):
    """Counters of a workload collected with rocprofv3 --pmc (tools/profile_round.sh) and committed under profiles/:
    executed f64 flops, memory-side bytes and VALU busy fraction, all per launch.  What a launch executes does not depend
    on when it runs, so the counts are replayed here and divided by the kernel time measured live; they are replayed
    only while the kernel sources are the ones they were taken from (tools/source_hash.py) and only for the workload
    they were taken on.  Returns (derived figures, file name) or (None, None)."""
    import glob
    try:
        sys.path.insert(0, os.path.join(ROOT, "tools"))
        from source_hash import source_hash
        now = source_hash()
    except (OSError, ImportError):
        return None, None
    for path in sorted(glob.glob(PMC_SUMMARIES)):
        try:
            doc = json.load(open(path))
            if doc.get("workload") == workload and doc.get("kernel_source_hash") == now:
                return doc["derived"], os.path.basename(path)
        except (OSError, KeyError, ValueError):
            continue
    return None, None


def spawn_ranks(n, argv):
    """`python bench.py --gpus N` without a launcher: start N ranks with torch.distributed.run as a child process
    (before this process has touched the GPU) and pass its exit code on."""
    import socket
    import subprocess
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n), "--master-addr",
           "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + argv
    return subprocess.call(cmd)


def scene_flops_per_ray(blob):
    import struct
    hdr = struct.unpack_from("<IIQIIIIIIIIQQQQQ", blob, 0)
    n_leaves, off_leaves = hdr[3], hdr[11]
    leaf_size = 208
    total = 0
    for i in range(n_leaves):
        (ptype,) = struct.unpack_from("<I", blob, off_leaves + i * leaf_size + 192)
        total += F_PRIM[ptype]
    return total, n_leaves


def cpu_baseline(whole_frame=False, scene_path=SCENE, width=WIDTH, height=HEIGHT, depth=DEPTH):
    """The f64 CPU oracle (restated reference algorithm) on a bounded sample of the same frame; with whole_frame
    (--cpu-whole-frame, a quarter of a minute of 16 threads) every row of it with all threads as well."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import oracle_py
    threads = int(os.environ.get("OMP_NUM_THREADS", "0")) or min(16, os.cpu_count() or 1)
    prog = oracle_py.OracleProgram(path=scene_path)
    name = os.path.basename(scene_path)
    WIDTH, HEIGHT, DEPTH = width, height, depth
    whole = None
    if whole_frame:
        t0 = time.time()
        _, _, cw = prog.render(0, w=WIDTH, h=HEIGHT, depth=DEPTH, threads=threads)
        tw = time.time() - t0
        whole = {"value": round(sum(cw.values()) / tw / 1e6, 4), "unit": "Mrays/s", "cores": threads, "rays": sum(cw.values()),
                 "wall_s": round(tw, 2), "sample": "every row of the frame"}
    # single thread, as the reference runs: every 120th row (18 rows)
    t0 = time.time()
    _, _, c1 = prog.render(0, w=WIDTH, h=HEIGHT, depth=DEPTH, y0=60, y1=HEIGHT, ystep=120, threads=1)
    t1 = time.time() - t0
    # all cores: every 12th row (180 rows)
    t0 = time.time()
    _, _, cn = prog.render(0, w=WIDTH, h=HEIGHT, depth=DEPTH, y0=6, y1=HEIGHT, ystep=12, threads=threads)
    tn = time.time() - t0
    r1, rn = sum(c1.values()), sum(cn.values())
    return {
        "value": round(rn / tn / 1e6, 4), "unit": "Mrays/s", "cores": threads, "kind": "port",
        "sample": "%s %dx%d depth %d, every 12th row (%d rows, %d rays, %.1f s wall on %d threads); "
                  "single thread on every 120th row: %.4f Mrays/s (%d rays, %.1f s)"
                  % (name, WIDTH, HEIGHT, DEPTH, len(range(6, HEIGHT, 12)), rn, tn, threads, r1 / t1 / 1e6, r1, t1),
        "single_thread_value": round(r1 / t1 / 1e6, 4),
        "whole_frame": whole,
    }


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-whole-frame", action="store_true", help="also time the CPU oracle on every row of the frame (about 15 s of 16 threads)")
    ap.add_argument("--no-host-path", action="store_true", help="skip the first-frame / host-buffer measurements")
    ap.add_argument("--backend", default=None, help="process group backend (default nccl; gloo for CPU rehearsals of the launch path)")
    ap.add_argument("--gather", choices=["host", "rccl"], default="host")
    ap.add_argument("--verify", action="store_true", help="rank 0 compares the gathered frame with one it renders alone")
    ap.add_argument("--dry-run", action="store_true", help="set up the ranks, print what would run, render nothing")
    ap.add_argument("--scene", default=SCENE)
    ap.add_argument("--width", type=int, default=WIDTH)
    ap.add_argument("--height", type=int, default=HEIGHT)
    ap.add_argument("--depth", type=int, default=None, help="default: 5 for the headline scene, else what the scene file says")
    args = ap.parse_args()

    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        # no launcher around this process: be one (never print a one-GPU line for an N-GPU request)
        raise SystemExit(spawn_ranks(args.gpus, sys.argv[1:]))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    # (ranks beyond the visible GPUs share them: rehearsals of the N-rank path on a box with fewer GPUs, gloo only)
    local_rank = int(os.environ.get("LOCAL_RANK", "0")) % max(1, torch.cuda.device_count())
    if world != args.gpus:
        raise SystemExit("bench.py: --gpus %d but the launcher started %d ranks" % (args.gpus, world))
    backend = args.backend or "nccl"
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            try:
                dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))
            except Exception as e:  # noqa: BLE001
                # RCCL unusable on this node (every rank sees the same failure at start-up): the path's only
                # exchange is the host framebuffer, so the bookkeeping collectives can go over gloo
                print("bench.py: nccl process group failed (%s); using gloo for the bookkeeping collectives" % e, file=sys.stderr)
                if dist.is_initialized():
                    dist.destroy_process_group()
                backend = "gloo"
                dist.init_process_group(backend="gloo")
        else:
            dist.init_process_group(backend=backend)
    if args.dry_run:
        # the launch path alone (tests run it on CPU with gloo): every rank reports, rank 0 prints the roll call
        me = torch.tensor([rank], dtype=torch.int64)
        ranks = [torch.zeros(1, dtype=torch.int64) for _ in range(world)]
        if world > 1:
            dist.all_gather(ranks, me)
        else:
            ranks = [me]
        if rank == 0:
            print(json.dumps({"dry_run": True, "n_gpus": world, "ranks": [int(r) for r in ranks],
                              "rows_per_rank": [gmlrt.part_rows(args.height, TILE_ROWS, r, world) for r in range(world)]}), flush=True)
        if world > 1:
            dist.destroy_process_group()
        return
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the render path has no CPU fallback")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)

    # ---- inputs resident in HBM before the timed region
    depth = args.depth if args.depth is not None else (DEPTH if os.path.abspath(args.scene) == SCENE else None)
    call = gmlrt.Program(path=args.scene).calls[0].with_overrides(args.width, args.height, depth)
    w, h = call.params.width, call.params.height
    flops_per_ray, n_leaves = scene_flops_per_ray(call.blob)
    scene = gmlrt.DeviceScene(call.blob, local_rank)
    p = gmlrt.Params()
    ctypes.memmove(ctypes.byref(p), ctypes.byref(call.params), ctypes.sizeof(p))
    p.tile_rows, p.part, p.nparts = TILE_ROWS, rank, world
    # padded so every rank gathers equal shapes
    out = torch.zeros((rowtiles.padded_rows(h, TILE_ROWS, world), w, 3), dtype=torch.uint8, device=dev)
    # The final framebuffer gather, the only exchange of a step.  "host" (default): the frame is one host buffer all
    # ranks have mapped and every rank copies its rows from its GPU straight to their places, all at once; "rccl":
    # dist.gather of the packed rows into rank 0's HBM.
    gathered = shared = None
    if world > 1 and args.gather == "rccl":
        gathered = [torch.empty_like(out) for _ in range(world)] if rank == 0 else None
    elif world > 1:
        try:
            shared = rowtiles.SharedFramebuffer(h, w, 3, tile_rows=TILE_ROWS)
        except RuntimeError:  # raised on every rank alike: fall back to the collective
            gathered = [torch.empty_like(out) for _ in range(world)] if rank == 0 else None
    stream = torch.cuda.current_stream().cuda_stream
    gather_s = [0.0]
    # With the host framebuffer, a step's rows travel to the host while the next step renders: two device buffers
    # in turn, and one thread per rank that does the copying (the library call blocks and runs on a stream of its
    # own).  Every step still renders the whole partition and delivers all of its rows; the clock stops only after
    # the last copy has landed.
    outs = [out, torch.zeros_like(out)] if shared is not None else [out]
    copy_jobs = None
    if shared is not None:
        import queue, threading
        copy_jobs = queue.Queue()
        copy_err = []

        def copier():
            while True:
                buf = copy_jobs.get()
                try:
                    if buf is None:
                        return
                    t = time.perf_counter()
                    shared.put_rows_from_device(local_rank, buf.data_ptr())
                    gather_s[0] += time.perf_counter() - t
                except Exception as e:  # noqa: BLE001 (reported by the main thread)
                    copy_err.append(e)
                finally:
                    copy_jobs.task_done()
        threading.Thread(target=copier, daemon=True).start()
    n_step = [0]

    def step():
        buf = outs[n_step[0] % len(outs)]
        n_step[0] += 1
        if copy_jobs is not None:
            # the rows this buffer held (the step before the last one's) must have left for the host before the kernel
            # writes it again: at most len(outs) - 1 copies may still be on their way
            while copy_jobs.unfinished_tasks >= len(outs):
                time.sleep(0)
        st = scene.render_into(p, buf.data_ptr(), None, stream)  # returns when the kernel is done
        if copy_jobs is not None:
            copy_jobs.put(buf)
        elif world > 1:
            t = time.perf_counter()
            rowtiles.gather_to_rank0(buf, gathered)
            torch.cuda.synchronize()
            gather_s[0] += time.perf_counter() - t
        return st

    def drain():
        if copy_jobs is not None:
            copy_jobs.join()
            if copy_err:
                raise copy_err[0]

    # Scene preparation, like the upload above: the library times the pixel tiles in the first launch with a
    # given geometry and tries up to three other ways of handing them out in the launches that follow, one of them twice (gmlrt_kernels.hip).
    learn = [scene.render_into(p, out.data_ptr(), None, stream) for _ in range(5)]
    learn_ms = [x.kernel_ms for x in learn]
    st = learn[-1]
    for _ in range(args.warmup):
        step()
    drain()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    gather_s[0] = 0.0
    t0 = time.perf_counter()
    rays = kernel_ms = hits = 0
    for _ in range(args.steps):
        st = step()
        rays += st.rays
        hits += st.hits
        kernel_ms += st.kernel_ms
    drain()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    out = outs[(n_step[0] - 1) % len(outs)] if n_step[0] else out

    # whole-job figures: max time over ranks, rays summed over ranks
    red_dev = dev if backend == "nccl" else torch.device("cpu")
    agg = torch.tensor([elapsed, kernel_ms / max(args.steps, 1), gather_s[0] / max(args.steps, 1)], dtype=torch.float64, device=red_dev)
    tot = torch.tensor([rays, hits], dtype=torch.float64, device=red_dev)
    if world > 1:
        dist.all_reduce(agg, op=dist.ReduceOp.MAX)
        dist.all_reduce(tot, op=dist.ReduceOp.SUM)
    elapsed_max, kernel_ms_max, gather_s_max = agg.tolist()
    rays_all, hits_all = tot.tolist()
    # what every rank did, for reading a scaling run: its kernel time per step, how its handle ended up dealing its
    # tiles, how many tiles it shared among the waves of a block, its share of the rays
    mine = torch.tensor([kernel_ms / max(args.steps, 1), float(st.handout_way), float(st.tiles_shared), rays / max(args.steps, 1),
                         gather_s[0] * 1e3 / max(args.steps, 1), elapsed * 1e3 / max(args.steps, 1)], dtype=torch.float64, device=red_dev)
    per_rank = [torch.zeros_like(mine) for _ in range(world)]
    if world > 1:
        dist.all_gather(per_rank, mine)
    else:
        per_rank = [mine]
    per_rank = [t.tolist() for t in per_rank]

    if rank == 0:
        ms_per_step = elapsed_max * 1e3 / args.steps
        rays_per_step = rays_all / args.steps
        value = rays_per_step / (ms_per_step * 1e-3) / 1e6
        # roofline of the render kernel on rank 0's GPU: its own work / its own HIP-event duration
        my_rays = rays / args.steps
        my_hits = hits / args.steps
        my_kernel_ms = kernel_ms / args.steps
        workload = "%s %dx%d depth %d" % (os.path.basename(args.scene), w, h, call.params.depth)
        headline = workload == "chess.gml 3840x2160 depth 5"
        pmc, pmc_file = measured_counters(workload) if world == 1 else (None, None)
        # what the reference's brute force would execute for these rays (every primitive for every ray): a rate of
        # useful results, most of which the kernel obtains by culling; never a fraction of a hardware peak
        equiv = (my_rays * flops_per_ray + my_hits * F_SHADE) / (my_kernel_ms * 1e-3) / 1e12
        roofline = {"bound": "valu", "achieved": None, "peak": PEAK_FP64_VALU_TFLOPS, "unit": "TFLOP/s", "frac": None,
                    "kernel": "render_kernel", "kernel_ms": round(my_kernel_ms, 4),
                    "kernel_Mrays_per_s": round(my_rays / (my_kernel_ms * 1e-3) / 1e6, 3),
                    "traffic": None, "algorithmic_equiv_tflops": round(equiv, 3),
                    "algorithmic_equiv_note": "rays x the flops the reference's brute force spends per ray (%d on this scene) + "
                                              "150 per shaded hit, over kernel time: useful results, not executed work" % flops_per_ray}
        if pmc:
            achieved = pmc["f64_flops_per_launch"] / (my_kernel_ms * 1e-3) / 1e12
            roofline.update({
                "achieved": round(achieved, 3), "frac": round(achieved / PEAK_FP64_VALU_TFLOPS, 4),
                "frac_of_fp32_valu_peak": round(achieved / PEAK_FP32_VALU_TFLOPS, 4),
                "achieved_note": "executed f64 flops per launch (rocprofv3 SQ_INSTS_VALU_{ADD,MUL,TRANS}_F64 + 2 x FMA_F64, x mean "
                                 "active lanes; %s, taken from these kernel sources) / kernel time measured here" % pmc_file,
                "f64_flops_per_launch": pmc["f64_flops_per_launch"],
                "traffic": pmc["fetch_bytes_per_launch"] + pmc["write_bytes_per_launch"],
                "valu_busy_measured": pmc["valu_busy_fraction"]})
        else:
            roofline["achieved_note"] = ("no executed-flop count for this workload and these kernel sources under profiles/ "
                                         "(tools/profile_round.sh collects one)")
        line = {
            "metric": "Mrays/s (primary+shadow+reflect), " + ("chess.gml 4K depth 5" if headline else workload),
            "value": round(value, 3), "unit": "Mrays/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(ms_per_step, 4), "higher_is_better": True,
            "scaling": "strong", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": workload,
                       "leaves": n_leaves, "flops_per_ray": flops_per_ray,
                       "rays_per_frame": int(rays_per_step), "parallelism": "row-tiles x%d (tile %d rows)" % (world, TILE_ROWS),
                       "gather": ("host framebuffer shared by the ranks, every rank copies its own rows while its next step renders" if shared is not None else
                                  "dist.gather to rank 0" if world > 1 else "none")},
            "kernel_ms_max_over_ranks": round(kernel_ms_max, 4),
            "gather_ms_per_step": round(gather_s_max * 1e3, 4) if world > 1 else None,
            "learning_launch_kernel_ms": [round(x, 4) for x in learn_ms],
            "per_rank": {"kernel_ms": [round(r[0], 4) for r in per_rank], "kernel_ms_mean": round(sum(r[0] for r in per_rank) / world, 4),
                         "handout_way": [int(r[1]) for r in per_rank], "tiles_shared": [int(r[2]) for r in per_rank],
                         "rays_per_step": [int(r[3]) for r in per_rank], "gather_ms": [round(r[4], 4) for r in per_rank],
                         "ms_per_step": [round(r[5], 4) for r in per_rank]},
            "process_group_backend": backend if world > 1 else None,
            "roofline": roofline,
        }
        if args.verify:
            whole = gmlrt.Params()
            ctypes.memmove(ctypes.byref(whole), ctypes.byref(call.params), ctypes.sizeof(whole))
            whole.tile_rows, whole.part, whole.nparts = TILE_ROWS, 0, 1
            ref = torch.zeros((h, w, 3), dtype=torch.uint8, device=dev)
            scene.render_into(whole, ref.data_ptr(), None, stream)
            if shared is not None:
                got = torch.from_numpy(np.array(shared.frame))
            elif world > 1:
                got = rowtiles.assemble(gathered, h, TILE_ROWS).cpu()
            else:
                got = out[:h].cpu()
            line["verified"] = bool((got == ref.cpu()).all())
        if world == 1 and not args.no_host_path:
            line["one_shot"] = one_shot(call)
        if world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(args.cpu_whole_frame, args.scene, w, h, call.params.depth)
        print(json.dumps(line), flush=True)
    scene.close()
    if shared is not None:
        shared.close()
    if world > 1:
        dist.destroy_process_group()


def one_shot(call):
    """The same frame the way the reference's `render` is called: once, with host buffers (gmlrt_render: validate and
    upload the scene, render with no tile order learnt, copy the rows into pageable memory).  A small other scene is
    rendered in between so that the library has nothing of this one left.  Medians of five."""
    other = gmlrt.Program(path=os.path.join(ROOT, "scenes", "spheres.gml")).calls[0]
    gmlrt.render(call)
    wall, kern, wall64 = [], [], []
    for want in (False, True):
        for _ in range(5):
            gmlrt.render(other)
            t = time.perf_counter()
            _, _, st = gmlrt.render(call, want_f64=want)
            (wall64 if want else wall).append((time.perf_counter() - t) * 1e3)
            if not want:
                kern.append(st.kernel_ms)
    med = lambda v: round(sorted(v)[len(v) // 2], 3)  # noqa: E731
    return {"first_frame_kernel_ms": med(kern), "e2e_ms_host_buffers": med(wall), "e2e_ms_host_buffers_with_f64_pixels": med(wall64),
            "e2e_Mrays_per_s": round(call_rays(call) / med(wall) / 1e3, 1),
            "note": "gmlrt_render into pageable numpy arrays allocated per call; never part of `value`"}


_RAYS = {}


def call_rays(call):
    key = id(call)
    if key not in _RAYS:
        _RAYS[key] = gmlrt.render(call)[2].rays
    return _RAYS[key]


if __name__ == "__main__":
    main()
