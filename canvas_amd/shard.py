"""Frame sharding across the GPUs of one node: one process per GPU, frame g -> rank g % world.

Frames are independent random-access units in the reference (docs/sphinx/framework.rst:16-18;
workspace bookkeeping is a pure function of the frame index, src/cprocess/workspace.c:243-307), so
there is no exchange step on the frame path and no collective on it.  RCCL (torch.distributed
backend "nccl") is used for exactly two things:
  * one broadcast of the packed parameter block -- 3x3 matrix (36 B) + the transfer tables in use
    (128 KiB each) -- from rank 0 at start / on parameter change; latency-bound, never link-bound;
  * one all-gather of per-rank {frames, checksum, seconds} at the end.
The same code runs over gloo on CPU tensors (tests/test_shard_gloo.py, world_size 2).
"""
import ctypes as C

import numpy as np

LUT_BYTES = 65536 * 2
MATRIX_BYTES = 9 * 4


def frames_of_rank(rank, world, count, first=0):
    """The first `count` global frame indices owned by `rank` (g % world == rank), from `first` up."""
    start = first + ((rank - first) % world)
    return [start + j * world for j in range(count)]


def owner_of_frame(frame, world):
    return frame % world


def pack_parameters(lib, matrix, lut_ids):
    """Rank 0: matrix + host copies of the tables -> one byte block."""
    block = np.empty(MATRIX_BYTES + LUT_BYTES * len(lut_ids), np.uint8)
    block[:MATRIX_BYTES] = np.ascontiguousarray(matrix, np.float32).reshape(9).view(np.uint8)
    for i, which in enumerate(lut_ids):
        p = lib.cvs_lut_host(which)
        if not p:
            raise RuntimeError("transfer table %d unavailable" % which)
        tab = np.ctypeslib.as_array(p, shape=(65536,))
        block[MATRIX_BYTES + i * LUT_BYTES: MATRIX_BYTES + (i + 1) * LUT_BYTES] = tab.view(np.uint8)
    return block


def unpack_parameters(lib, block, lut_ids, install=True):
    block = np.ascontiguousarray(block, np.uint8)
    matrix = block[:MATRIX_BYTES].view(np.float32).copy()
    for i, which in enumerate(lut_ids):
        tab = block[MATRIX_BYTES + i * LUT_BYTES: MATRIX_BYTES + (i + 1) * LUT_BYTES].view(np.uint16).copy()
        if install:
            rc = lib.cvs_lut_install(which, tab.ctypes.data_as(C.POINTER(C.c_uint16)))
            if rc != 0:
                raise RuntimeError("cvs_lut_install(%d) failed" % which)
    return matrix


def broadcast_parameters(lib, dist, rank, matrix, lut_ids, device=None):
    """Every rank ends up with rank 0's matrix and tables.  dist=None: single process."""
    if dist is None:
        for which in lut_ids:
            if not lib.cvs_lut_device(which):
                raise RuntimeError("transfer table %d unavailable" % which)
        return np.ascontiguousarray(matrix, np.float32).reshape(9)
    import torch
    if device is None:
        device = "cuda" if dist.get_backend() == "nccl" else "cpu"
    nbytes = MATRIX_BYTES + LUT_BYTES * len(lut_ids)
    if rank == 0:
        t = torch.from_numpy(pack_parameters(lib, matrix, lut_ids)).to(device)
    else:
        t = torch.empty(nbytes, dtype=torch.uint8, device=device)
    dist.broadcast(t, src=0)
    block = t.cpu().numpy()
    # rank 0 keeps its own tables; the others install what they received
    return unpack_parameters(lib, block, lut_ids, install=(rank != 0))


def checksum52(hex_digest):
    """The first 52 bits of a SHA-256 hex digest: what fits a float64 exactly, so that it can ride in gather_stats."""
    return int(hex_digest[:13], 16)


def gather_stats(dist, frames_done, checksum, seconds, extra=(), device=None):
    """All ranks' (frames_done, checksum, seconds, *extra) on every rank: list of tuples, index = rank.
    `checksum` < 2**52 (checksum52 of the digest of one rendered frame); `extra`: further floats per rank (verified
    flag, launch time, ...).  This is the one collective at the end of a run."""
    extra = [float(x) for x in extra]
    if dist is None:
        return [(int(frames_done), int(checksum), float(seconds), *extra)]
    import torch
    if device is None:
        device = "cuda" if dist.get_backend() == "nccl" else "cpu"
    mine = torch.tensor([float(frames_done), float(checksum % (1 << 52)), float(seconds)] + extra, dtype=torch.float64, device=device)
    out = [torch.empty_like(mine) for _ in range(dist.get_world_size())]
    dist.all_gather(out, mine)
    return [(int(o[0].item()), int(o[1].item()), float(o[2].item()), *[float(x) for x in o[3:].tolist()]) for o in out]
