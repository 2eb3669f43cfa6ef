"""The N>1 path on CPU: two gloo ranks partition a frame into interleaved row tiles, each fills its packed rows
(here with the CPU oracle standing in for the GPU kernel, which this container cannot run), gather to rank 0 and
reassemble; the result must equal the single-process frame byte for byte."""
import os
import subprocess
import sys
import textwrap

import numpy as np

from conftest import ROOT, SCENES

WORKER = textwrap.dedent("""
    import os, sys
    import numpy as np
    import torch, torch.distributed as dist
    sys.path.insert(0, os.path.join({root!r}, "rust-raytracer_amd"))
    sys.path.insert(0, os.path.join({root!r}, "oracle"))
    import rowtiles, oracle_py
    dist.init_process_group(backend="gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    W, H, DEPTH, TILE = 48, 37, 2, 8          # 37 rows: a ragged last tile
    prog = oracle_py.OracleProgram(path={scene!r})
    rows = rowtiles.my_rows(H, TILE, rank, world)
    local = torch.zeros((rowtiles.padded_rows(H, TILE, world), W, 3), dtype=torch.uint8)
    full, _, _ = prog.render(0, w=W, h=H, depth=DEPTH)
    # each rank renders ONLY its rows (one oracle call per row keeps the stand-in simple)
    for k, y in enumerate(rows):
        img, _, _ = prog.render(0, w=W, h=H, depth=DEPTH, y0=int(y), y1=int(y) + 1)
        local[k] = torch.from_numpy(img[0])
    got = rowtiles.gather_to_rank0(local)
    if rank == 0:
        frame = rowtiles.assemble(got, H, TILE).numpy()
        assert frame.shape == full.shape and (frame == full).all(), "reassembled frame differs"
        assert sum(len(rowtiles.my_rows(H, TILE, r, world)) for r in range(world)) == H
        print("MULTIRANK_OK", world)
    dist.barrier()
    dist.destroy_process_group()
""")


def test_two_ranks_gloo(tmp_path):
    script = tmp_path / "worker.py"
    script.write_text(WORKER.format(root=ROOT, scene=os.path.join(SCENES, "spheres.gml")))
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", OMP_NUM_THREADS="1")
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2",
                          "--master-addr", "127.0.0.1", "--master-port", "29611", str(script)],
                         env=env, capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "MULTIRANK_OK 2" in out.stdout
