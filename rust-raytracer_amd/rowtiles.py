"""Row-tile partition of a frame over ranks and the final framebuffer gather.

Image rows are independent (raytracer.rs:137-146), so a frame is cut into tiles of `tile_rows` rows dealt
round-robin to the ranks (rank r renders tiles r, r+world, ...: cheap sky rows and expensive board rows are spread
evenly).  Nothing is exchanged while rendering; the only collective is this gather of the finished rows to rank 0.
"""
import torch
import torch.distributed as dist

import gmlrt


def my_rows(height, tile_rows, rank, world):
    """Image rows (ascending) that `rank` renders; its output buffer holds them packed in this order."""
    return gmlrt.part_row_index(height, tile_rows, rank, world)


def padded_rows(height, tile_rows, world):
    """Every rank's buffer is padded to this many rows so that the gather moves equal shapes."""
    return max(gmlrt.part_rows(height, tile_rows, r, world) for r in range(world))


def gather_to_rank0(local, gathered=None, dst=0):
    """Gathers each rank's packed rows [padded_rows, w, 3] on `dst`.  Returns the list on dst, None elsewhere."""
    world = dist.get_world_size()
    rank = dist.get_rank()
    if rank == dst and gathered is None:
        gathered = [torch.empty_like(local) for _ in range(world)]
    dist.gather(local, gathered if rank == dst else None, dst=dst)
    return gathered if rank == dst else None


def assemble(gathered, height, tile_rows):
    """Puts the gathered per-rank rows back at their image positions: [height, w, 3]."""
    world = len(gathered)
    w = gathered[0].shape[1]
    frame = torch.empty((height, w, 3), dtype=gathered[0].dtype, device=gathered[0].device)
    for r in range(world):
        rows = torch.as_tensor(my_rows(height, tile_rows, r, world), device=frame.device)
        frame[rows] = gathered[r][: len(rows)]
    return frame
