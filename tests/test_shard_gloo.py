"""The N>1 path on CPU: two processes, gloo backend, the same sharding/broadcast code bench.py
runs over RCCL.  No GPU, no pixels: a stand-in object plays the library's LUT store."""
import ctypes as C
import os
import socket
import subprocess
import sys
import textwrap

import numpy as np

from canvas_amd import shard

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_round_robin_assignment_is_a_partition():
    for world in (1, 2, 4, 8):
        owned = [shard.frames_of_rank(r, world, 5) for r in range(world)]
        flat = sorted(g for o in owned for g in o)
        assert flat == list(range(5 * world))
        for r, o in enumerate(owned):
            assert all(shard.owner_of_frame(g, world) == r for g in o)
    assert shard.frames_of_rank(1, 4, 3, first=10) == [13, 17, 21]


class FakeStore:
    """cvs_lut_host / cvs_lut_install / cvs_lut_device of the real library, minus the GPU."""

    def __init__(self, tables=None):
        self.tables = dict(tables or {})
        self.installed = {}

    def cvs_lut_host(self, which):
        t = self.tables.get(which)
        return None if t is None else t.ctypes.data_as(C.POINTER(C.c_uint16))

    def cvs_lut_device(self, which):
        return 1 if which in self.tables else None

    def cvs_lut_install(self, which, ptr):
        self.installed[which] = np.ctypeslib.as_array(ptr, shape=(65536,)).copy()
        return 0


def test_pack_unpack_round_trip():
    rng = np.random.default_rng(0)
    tabs = {0: rng.integers(0, 65536, 65536).astype(np.uint16), 3: rng.integers(0, 65536, 65536).astype(np.uint16)}
    m = rng.normal(size=9).astype(np.float32)
    block = shard.pack_parameters(FakeStore(tabs), m, [0, 3])
    assert block.nbytes == 36 + 2 * 131072
    dst = FakeStore()
    m2 = shard.unpack_parameters(dst, block, [0, 3])
    assert np.array_equal(m, m2)
    assert np.array_equal(dst.installed[0], tabs[0]) and np.array_equal(dst.installed[3], tabs[3])


WORKER = textwrap.dedent("""
    import os, sys, json
    import numpy as np
    sys.path.insert(0, %(root)r)
    import torch.distributed as dist
    from canvas_amd import shard
    from tests.test_shard_gloo import FakeStore
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    dist.init_process_group("gloo")
    rng = np.random.default_rng(123)
    table = rng.integers(0, 65536, 65536).astype(np.uint16)
    if rank == 0:
        store, m = FakeStore({0: table}), np.arange(9, dtype=np.float32) * 0.5
    else:
        store, m = FakeStore(), np.zeros(9, np.float32)
    got = shard.broadcast_parameters(store, dist, rank, m, [0])
    ok_m = bool(np.array_equal(got, np.arange(9, dtype=np.float32) * 0.5))
    ok_t = rank == 0 or bool(np.array_equal(store.installed[0], table))
    mine = shard.frames_of_rank(rank, world, 6)
    stats = shard.gather_stats(dist, len(mine), sum(mine), 0.25 * (rank + 1), extra=(1 - rank, 2.5 + rank))
    print(json.dumps({"rank": rank, "ok_m": ok_m, "ok_t": ok_t, "frames": mine, "stats": stats}), flush=True)
    dist.barrier()
    dist.destroy_process_group()
""")


def test_two_rank_broadcast_and_gather_over_gloo(tmp_path):
    """Two ranks started by the SAME launcher code bench.py uses for `--gpus N` (canvas_amd.launch.spawn_ranks),
    gloo instead of RCCL: the parameter broadcast and the end-of-run all-gather with the per-rank proof fields."""
    import json
    from canvas_amd import launch
    script = tmp_path / "worker.py"
    script.write_text(WORKER % {"root": ROOT})
    rc, outs = launch.spawn_ranks(2, [sys.executable, str(script)], timeout=240, relay=False)
    assert rc == 0, outs
    outs = sorted((json.loads(o.strip().splitlines()[-1]) for o in outs), key=lambda o: o["rank"])
    assert all(o["ok_m"] and o["ok_t"] for o in outs)
    assert outs[0]["frames"] == [0, 2, 4, 6, 8, 10] and outs[1]["frames"] == [1, 3, 5, 7, 9, 11]
    for o in outs:
        assert [s[0] for s in o["stats"]] == [6, 6]
        assert [s[1] for s in o["stats"]] == [30, 36]
        assert [round(s[2], 2) for s in o["stats"]] == [0.25, 0.5]
        assert [s[3:] for s in o["stats"]] == [[1.0, 2.5], [0.0, 3.5]]          # the extra fields ride along, rank by rank


def test_a_failing_rank_takes_the_others_down(tmp_path):
    from canvas_amd import launch
    script = tmp_path / "w.py"
    script.write_text("import os, sys, time\nif os.environ['RANK'] == '1': sys.exit(7)\ntime.sleep(60)\n")
    import time
    t0 = time.monotonic()
    rc, _ = launch.spawn_ranks(2, [sys.executable, str(script)], timeout=30, relay=False)
    assert rc == 7 and time.monotonic() - t0 < 20


def test_bench_refuses_more_ranks_than_devices():
    """`python bench.py --gpus 2` with no launcher starts its own ranks -- or, with fewer than 2 devices visible, says so
    and exits before touching anything."""
    from canvas_amd import launch
    if launch.visible_gpu_count() >= 2:
        import pytest
        pytest.skip("two GPUs are visible here")
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"],
                       env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=120)
    assert p.returncode == 2
    assert "2 ranks wanted, %d devices visible" % launch.visible_gpu_count() in p.stderr
    assert p.stdout == ""


def test_checksum52_fits_a_float64():
    d = "f" * 64
    v = shard.checksum52(d)
    assert v == (1 << 52) - 1 and float(v) == v
