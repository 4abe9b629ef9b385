"""One process per GPU, started by the measuring script itself.

`python bench.py --gpus N` must run N ranks whether or not something like `torch.distributed.run` started it.
When WORLD_SIZE is absent, `ensure_ranks(N)` re-runs the calling script N times as child processes with
RANK / LOCAL_RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT set, relays their output and exits with the worst
exit code.  It runs BEFORE the first HIP call of the parent: a process that has initialised the GPU must not
fork-and-exec on this pool, and the parent never needs the GPU at all (it only waits).

Frames are independent units (reference: docs/sphinx/framework.rst:16-18), so the ranks share nothing but the
rendezvous; see shard.py for what travels over RCCL.
"""
import glob
import os
import socket
import subprocess
import sys
import threading
import time


def visible_gpu_count():
    """GPUs this process may use, WITHOUT touching the HIP runtime: the KFD topology in sysfs (nodes with
    SIMDs are GPUs), narrowed by HIP_VISIBLE_DEVICES / ROCR_VISIBLE_DEVICES when set."""
    n = 0
    for props in glob.glob("/sys/class/kfd/kfd/topology/nodes/*/properties"):
        try:
            with open(props) as f:
                for line in f:
                    if line.startswith("simd_count"):
                        if int(line.split()[1]) > 0:
                            n += 1
                        break
        except (OSError, ValueError, IndexError):
            pass
    for var in ("HIP_VISIBLE_DEVICES", "ROCR_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES"):
        v = os.environ.get(var)
        if v is not None:
            listed = [x for x in v.split(",") if x.strip() != ""]
            n = min(n, len(listed))
    return n


def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def spawn_ranks(n, argv, env=None, timeout=None, relay=True):
    """Start `argv` n times (rank r gets RANK=r, LOCAL_RANK=r, WORLD_SIZE=n, rendezvous on 127.0.0.1).
    Returns (worst exit code, [stdout of rank r]).  stderr of the children is relayed to this process's stderr line by
    line, every line tagged "[rank r] " (eight ranks' RCCL banners and tracebacks stay attributable).
    A rank that fails takes the others down (exact PIDs, no pattern kill)."""
    base = dict(os.environ if env is None else env)
    base.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(free_port()), WORLD_SIZE=str(n),
                HSA_ENABLE_IPC_MODE_LEGACY=base.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    procs, chunks, readers = [], [[] for _ in range(n)], []
    err_lock = threading.Lock()

    def drain(pipe, into):                              # a rank may print more than a pipe holds (64 KiB): read as it writes
        for line in pipe:
            into.append(line)
        pipe.close()

    def tag(pipe, r):
        for line in pipe:
            with err_lock:                              # whole lines, one rank at a time
                sys.stderr.write("[rank %d] %s" % (r, line if line.endswith("\n") else line + "\n"))
                sys.stderr.flush()
        pipe.close()

    for r in range(n):
        e = dict(base, RANK=str(r), LOCAL_RANK=str(r), CANVAS_STDERR_TAGGED="1")
        procs.append(subprocess.Popen(list(argv), env=e, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, errors="replace"))
        readers.append(threading.Thread(target=drain, args=(procs[r].stdout, chunks[r]), daemon=True))
        readers[-1].start()
        readers.append(threading.Thread(target=tag, args=(procs[r].stderr, r), daemon=True))
        readers[-1].start()
    deadline = None if timeout is None else time.monotonic() + timeout
    worst = 0
    pending = set(range(n))
    while pending:
        for r in sorted(pending):
            rc = procs[r].poll()
            if rc is None:
                continue
            pending.discard(r)
            if rc != 0:
                worst = worst or rc
                for q in pending:                       # one rank down: the rendezvous can never complete
                    procs[q].terminate()
        if pending:
            if deadline is not None and time.monotonic() > deadline:
                sys.stderr.write("ranks %s still running after %.0f s: killed\n" % (sorted(pending), timeout))
                for q in pending:
                    procs[q].kill()
                worst = worst or 124
                deadline = None
            time.sleep(0.05)
    for t in readers:
        t.join(10)
    outs = ["".join(c) for c in chunks]
    if relay:
        for r in range(n):
            if outs[r]:
                sys.stdout.write(outs[r])
        sys.stdout.flush()
    return worst, outs


def ensure_ranks(gpus, argv=None, need_devices=True, timeout=3000):
    """Call first thing in main(), before any GPU call.  Returns normally inside a rank (or when one rank is
    all that was asked for); otherwise spawns the ranks, waits (at most `timeout` seconds: a hung rendezvous ends with
    a message and exit code 124 instead of hanging the caller), and exits the process."""
    if gpus <= 1 or "WORLD_SIZE" in os.environ:
        return
    if need_devices:
        have = visible_gpu_count()
        if have < gpus:
            sys.stderr.write("%d ranks wanted, %d devices visible: not starting\n" % (gpus, have))
            sys.exit(2)
    rc, _ = spawn_ranks(gpus, [sys.executable] + (list(sys.argv) if argv is None else list(argv)), timeout=timeout)
    sys.exit(rc)


# ---------------------------------------------------------------- stdout carries the result and nothing else

class ResultOnly:
    """`with ResultOnly() as out: ...; out.emit(line)`: inside the block everything any library writes to file descriptor 1
    (RCCL prints its version, host name and library path there on communicator creation; eight ranks, eight banners) goes
    to stderr; emit() writes one line to the REAL stdout.  A measuring script's stdout is then exactly its result line."""

    def __enter__(self):
        sys.stdout.flush()
        try:
            self.real = os.dup(1)
            os.dup2(2, 1)
        except OSError:                                 # no usable stderr: leave stdout alone
            self.real = None
        return self

    def emit(self, line):
        sys.stdout.flush()
        if self.real is not None:
            os.dup2(self.real, 1)
        print(line, flush=True)
        if self.real is not None:
            os.dup2(2, 1)

    def __exit__(self, *exc):
        sys.stdout.flush()
        if self.real is not None:
            os.dup2(self.real, 1)
            os.close(self.real)
            self.real = None
        return False


# ---------------------------------------------------------------- where a rank's host threads run

def _cpulist(text):
    cpus = set()
    for part in text.strip().split(","):
        if not part:
            continue
        a, _, b = part.partition("-")
        cpus.update(range(int(a), int(b or a) + 1))
    return cpus


def gpu_numa_cpus(local_rank, sysfs="/sys"):
    """CPUs of the NUMA node GPU `local_rank` hangs off, from sysfs alone (no HIP call): the KFD topology lists the GPUs in
    the order the runtime numbers them (nodes with SIMDs; `domain` and `location_id` give the PCI address), and the PCI
    device says which node and CPUs are local to it.  None when the box does not say (containers often hide it)."""
    nodes = []
    for props in sorted(glob.glob(sysfs + "/class/kfd/kfd/topology/nodes/*/properties"), key=lambda p: int(p.split("/")[-2])):
        try:
            kv = dict(line.split()[:2] for line in open(props) if len(line.split()) >= 2)
            if int(kv.get("simd_count", "0")) > 0:
                nodes.append(kv)
        except (OSError, ValueError):
            pass
    order = os.environ.get("HIP_VISIBLE_DEVICES") or os.environ.get("ROCR_VISIBLE_DEVICES")
    index = local_rank
    if order:
        try:
            index = int([x for x in order.split(",") if x.strip()][local_rank])
        except (ValueError, IndexError):
            return None
    if index >= len(nodes):
        return None
    try:
        loc, dom = int(nodes[index]["location_id"]), int(nodes[index].get("domain", "0"))
        bdf = "%04x:%02x:%02x.%d" % (dom, (loc >> 8) & 0xFF, (loc >> 3) & 0x1F, loc & 7)
        dev = "%s/bus/pci/devices/%s" % (sysfs, bdf)
        node = int(open(dev + "/numa_node").read())
        cpus = _cpulist(open(dev + "/local_cpulist").read())
    except (OSError, ValueError, KeyError):
        return None
    return {"numa_node": node, "pci": bdf, "cpus": cpus} if cpus else None


def place_rank(local_rank, world, sysfs="/sys"):
    """Call before the first HIP call of a rank.  Eight ranks on one host are eight enqueue threads plus RCCL's proxy and the
    runtime's helper threads: left alone they wander over both sockets.  Each rank is confined to the CPUs local to ITS GPU
    (its NUMA node, shared evenly with the other ranks whose GPUs sit on the same node); where sysfs does not say, to an
    even share of the CPUs the process may use.  Returns a dict for the result line; never raises."""
    try:
        allowed = sorted(os.sched_getaffinity(0))
    except (AttributeError, OSError):
        return {"method": "none (no sched_getaffinity)"}
    if world <= 1:
        return {"method": "none (one rank)", "cpus_allowed": len(allowed)}
    info = gpu_numa_cpus(local_rank, sysfs)
    try:
        if info:
            mates = [r for r in range(world) if (gpu_numa_cpus(r, sysfs) or {}).get("numa_node") == info["numa_node"]]
            local = sorted(info["cpus"] & set(allowed)) or allowed
            k = max(1, len(local) // max(1, len(mates)))
            at = mates.index(local_rank) if local_rank in mates else 0
            mine = local[at * k:(at + 1) * k] or local
            os.sched_setaffinity(0, mine)
            return {"method": "CPUs of the GPU's NUMA node (sysfs), shared among the node's ranks", "numa_node": info["numa_node"], "pci": info["pci"],
                    "cpus": "%d-%d (%d)" % (mine[0], mine[-1], len(mine))}
        k = len(allowed) // world
        if k < 4:                                       # too few to be worth fencing: the runtime's helper threads need room
            return {"method": "none (%d CPUs allowed for %d ranks)" % (len(allowed), world)}
        mine = allowed[local_rank * k:(local_rank + 1) * k] or allowed
        os.sched_setaffinity(0, mine)
        return {"method": "even share of the allowed CPUs (sysfs names no NUMA node for the GPU)", "cpus": "%d-%d (%d)" % (mine[0], mine[-1], len(mine))}
    except OSError as e:
        return {"method": "none (%s)" % e}


def tag_own_stderr(rank):
    """For ranks started by someone else's launcher (torch.distributed.run hands every rank the same stderr): from here on
    every line this PROCESS writes to file descriptor 2 -- Python's, RCCL's, the HIP runtime's -- reaches the real stderr
    as "[rank r] line".  A pipe and a reader thread; undone at exit.  No-op when spawn_ranks already tags (it set
    CANVAS_STDERR_TAGGED) or when CANVAS_TAG_STDERR=0."""
    if os.environ.get("CANVAS_STDERR_TAGGED") == "1" or os.environ.get("CANVAS_TAG_STDERR") == "0":
        return False
    import atexit
    try:
        sys.stderr.flush()
        rd, wr = os.pipe()
        real = os.dup(2)
        os.dup2(wr, 2)
        os.close(wr)
    except OSError:
        return False
    prefix = ("[rank %d] " % rank).encode()

    def pump():
        buf = b""
        while True:
            try:
                chunk = os.read(rd, 65536)
            except OSError:
                break
            if not chunk:
                break
            buf += chunk
            *lines, buf = buf.split(b"\n")
            if lines:
                os.write(real, b"".join(prefix + l + b"\n" for l in lines))
        if buf:
            os.write(real, prefix + buf + b"\n")

    t = threading.Thread(target=pump, daemon=True)
    t.start()

    def undo():
        try:
            sys.stderr.flush()
            os.dup2(real, 2)                             # closes the pipe's last write end: the pump sees EOF and drains
            t.join(5)
        except OSError:
            pass
    atexit.register(undo)
    os.environ["CANVAS_STDERR_TAGGED"] = "1"
    return True
