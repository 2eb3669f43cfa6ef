"""Row-tile partition of a frame over ranks and the final framebuffer gather.

Image rows are independent (raytracer.rs:137-146), so a frame is cut into tiles of `tile_rows` rows dealt
round-robin to the ranks (rank r renders tiles r, r+world, ...: cheap sky rows and expensive board rows are spread
evenly).  Nothing is exchanged while rendering; the only collective is this gather of the finished rows to rank 0.
"""
import os

import numpy as np
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


class SharedFramebuffer:
    """The frame as one host buffer that every rank of the node has mapped (a file in /dev/shm): the final gather
    is then each rank copying the rows it rendered from its own GPU straight to their places, all ranks at once, each
    over its own GPU's host link; no rank receives the others' rows first and nothing crosses the GPU-to-GPU links.
    Rank 0 creates the file; everybody calls close() at the end (rank 0 removes the file)."""

    def __init__(self, height, width, channels=3, dtype=np.uint8, tile_rows=8):
        self.rank = dist.get_rank() if dist.is_initialized() else 0
        self.world = dist.get_world_size() if dist.is_initialized() else 1
        self.tile_rows = tile_rows
        shape = (height, width, channels)
        name = [None]
        if self.rank == 0:
            try:
                path = "/dev/shm/gmlrt_fb_%d_%d_%s" % (os.getpid(), int(os.environ.get("MASTER_PORT", "0")), os.urandom(6).hex())
                # created here and nowhere else: never follows a link somebody planted, readable by this user only
                fd = os.open(path, os.O_CREAT | os.O_EXCL | os.O_RDWR, 0o600)
                try:
                    os.ftruncate(fd, int(np.prod(shape)) * np.dtype(dtype).itemsize)
                except OSError:
                    os.close(fd)
                    os.unlink(path)
                    raise
                os.close(fd)
                name = [path]
            except OSError:
                name = [None]  # no shared memory file system here: every rank learns it below and raises alike
        if self.world > 1:
            dist.broadcast_object_list(name, src=0)  # also orders the creation before everybody's open
        if name[0] is None:
            raise RuntimeError("no shared host memory for the framebuffer (/dev/shm not writable)")
        self.path = name[0]
        try:
            self.frame = np.memmap(self.path, dtype=dtype, mode="r+", shape=shape)
        except Exception:
            if self.rank == 0:  # do not leave the file behind
                try:
                    os.unlink(self.path)
                except OSError:
                    pass
            raise

    def put_rows_from_device(self, device, d_rows_ptr):
        """This rank's packed rows (device memory) to their image positions."""
        gmlrt.copy_rows_to_host(device, d_rows_ptr, self.frame, self.tile_rows, self.rank, self.world)

    def put_rows_from_host(self, local):
        """The same from a host array [rows, w, c] (CPU rehearsals of the multi-rank path)."""
        rows = my_rows(self.frame.shape[0], self.tile_rows, self.rank, self.world)
        self.frame[rows] = np.asarray(local)[: len(rows)]

    def close(self):
        if self.world > 1:
            dist.barrier()
        self.frame = None  # drops the mapping
        if self.rank == 0:
            try:
                os.unlink(self.path)
            except OSError:
                pass
