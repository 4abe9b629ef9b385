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


# ---------------------------------------------------------------- eight ranks on one host

WORKER8 = textwrap.dedent("""
    import os, sys, json
    import numpy as np
    sys.path.insert(0, %(root)r)
    from canvas_amd import launch, shard
    from tests.test_shard_gloo import FakeStore
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    with launch.ResultOnly() as out:
        os.write(1, ("NCCL version 2.x banner of rank %%d on fd 1\\n" %% rank).encode())      # what RCCL does on communicator creation
        sys.stderr.write("rank %%d says hello\\n" %% rank)
        import torch.distributed as dist
        dist.init_process_group("gloo")
        table = np.random.default_rng(5).integers(0, 65536, 65536).astype(np.uint16)
        store = FakeStore({0: table}) if rank == 0 else FakeStore()
        m = shard.broadcast_parameters(store, dist, rank, np.arange(9, dtype=np.float32), [0])
        ok = bool(np.array_equal(m, np.arange(9, dtype=np.float32))) and (rank == 0 or bool(np.array_equal(store.installed[0], table)))
        mine = shard.frames_of_rank(rank, world, 4)
        stats = shard.gather_stats(dist, len(mine), sum(mine), 0.1, extra=(float(ok),))
        print("stray print of rank %%d" %% rank)                     # a library printing through Python's stdout: also not the result
        if rank == 0:
            out.emit(json.dumps({"world": world, "frames_per_rank": [s[0] for s in stats], "sums": [s[1] for s in stats], "ok": [s[3] for s in stats]}))
        dist.barrier()
        dist.destroy_process_group()
""")


def test_eight_ranks_one_result_line_and_tagged_stderr(tmp_path):
    """`bench.py --gpus 8` as the driver will see it, rehearsed on CPU: eight ranks through the launcher's own spawn_ranks,
    gloo in RCCL's place, every rank writing a banner to file descriptor 1 -- the launcher's stdout must carry exactly ONE
    line (rank 0's result), and every stderr line must say which rank wrote it."""
    import json
    script = tmp_path / "worker8.py"
    script.write_text(WORKER8 % {"root": ROOT})
    driver = tmp_path / "driver.py"
    driver.write_text("import sys\nsys.path.insert(0, %r)\nfrom canvas_amd import launch\nrc, _ = launch.spawn_ranks(8, [sys.executable, %r], timeout=400)\nsys.exit(rc)\n" % (ROOT, str(script)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    p = subprocess.run([sys.executable, str(driver)], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=600)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [l for l in p.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, p.stdout
    res = json.loads(lines[0])
    assert res["world"] == 8 and res["frames_per_rank"] == [4] * 8 and res["ok"] == [1.0] * 8
    assert res["sums"] == [sum(shard.frames_of_rank(r, 8, 4)) for r in range(8)]
    for r in range(8):
        assert "[rank %d] rank %d says hello" % (r, r) in p.stderr
        assert "[rank %d] NCCL version 2.x banner of rank %d on fd 1" % (r, r) in p.stderr
        assert "[rank %d] stray print of rank %d" % (r, r) in p.stderr
    assert all(l.startswith("[rank ") for l in p.stderr.splitlines() if l.strip())


def test_rank_placement_from_a_sysfs_tree(tmp_path):
    """place_rank reads the KFD topology and the PCI device of the rank's GPU from sysfs (no HIP call): two GPUs on node 0,
    two on node 1 of a made-up tree -> each rank gets the CPUs local to its GPU, shared with its node mate."""
    from canvas_amd import launch
    root = tmp_path / "sys"
    gpus = [(0x0300, 0, "0-3"), (0x0400, 0, "0-3"), (0x8300, 1, "4-7"), (0x8400, 1, "4-7")]
    nodes = root / "class" / "kfd" / "kfd" / "topology" / "nodes"
    (nodes / "0").mkdir(parents=True)
    (nodes / "0" / "properties").write_text("cpu_cores_count 8\nsimd_count 0\n")           # the CPU node: not a GPU
    for i, (loc, numa, cpus) in enumerate(gpus):
        (nodes / str(i + 1)).mkdir()
        (nodes / str(i + 1) / "properties").write_text("cpu_cores_count 0\nsimd_count 1024\nlocation_id %d\ndomain 0\n" % loc)
        dev = root / "bus" / "pci" / "devices" / ("0000:%02x:%02x.%d" % ((loc >> 8) & 0xFF, (loc >> 3) & 0x1F, loc & 7))
        dev.mkdir(parents=True)
        (dev / "numa_node").write_text("%d\n" % numa)
        (dev / "local_cpulist").write_text(cpus + "\n")
    assert launch.gpu_numa_cpus(2, str(root)) == {"numa_node": 1, "pci": "0000:83:00.0", "cpus": {4, 5, 6, 7}}
    assert launch.gpu_numa_cpus(7, str(root)) is None
    if len(os.sched_getaffinity(0)) >= 8 and set(range(8)) <= os.sched_getaffinity(0):
        before = os.sched_getaffinity(0)
        try:
            got = launch.place_rank(3, 4, str(root))
            assert got["numa_node"] == 1 and os.sched_getaffinity(0) == {6, 7}, got     # node 1's second rank: the second half of 4-7
        finally:
            os.sched_setaffinity(0, before)
    assert launch.place_rank(0, 1, str(root))["method"].startswith("none")
