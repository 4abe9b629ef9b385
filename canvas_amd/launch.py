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
    Returns (worst exit code, [stdout of rank r]).  stderr of the children goes to this process's stderr.
    A rank that fails takes the others down (exact PIDs, no pattern kill)."""
    base = dict(os.environ if env is None else env)
    base.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(free_port()), WORLD_SIZE=str(n),
                HSA_ENABLE_IPC_MODE_LEGACY=base.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    procs, chunks, readers = [], [[] for _ in range(n)], []

    def drain(pipe, into):                              # a rank may print more than a pipe holds (64 KiB): read as it writes
        for line in pipe:
            into.append(line)
        pipe.close()

    for r in range(n):
        e = dict(base, RANK=str(r), LOCAL_RANK=str(r))
        procs.append(subprocess.Popen(list(argv), env=e, stdout=subprocess.PIPE, stderr=None, text=True))
        readers.append(threading.Thread(target=drain, args=(procs[r].stdout, chunks[r]), daemon=True))
        readers[r].start()
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
