"""Self-launch of the multi-GPU bench: one child process per GPU.

`python bench.py --gpus N` started WITHOUT torch.distributed.run has no peers; this module makes
the parent start them: N children of the same command line with RANK / LOCAL_RANK / WORLD_SIZE /
MASTER_ADDR / MASTER_PORT set (what torch.distributed.run would export), rank 0's stdout relayed
so that its JSON line is the parent's last stdout line, the worst child exit code returned, and
a deadline after which the children are ended by PID and the parent exits non-zero.

The parent must not have touched the GPU: nothing here imports torch or loads the HIP library
(a process that has initialised HIP must neither fork GPU workers nor exec; see the task's
environment notes), and nothing is ever re-exec'ed.
"""
import os
import signal
import socket
import subprocess
import sys
import threading
import time


def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def child_env(rank, world, port, base=None):
    env = dict(os.environ if base is None else base)
    env.update({"RANK": str(rank), "LOCAL_RANK": str(rank), "WORLD_SIZE": str(world),
                "LOCAL_WORLD_SIZE": str(world), "MASTER_ADDR": "127.0.0.1", "MASTER_PORT": str(port)})
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")     # dmabuf IPC only on this pool (RCCL needs it)
    return env


def spawn_ranks(world, argv, timeout_s=900.0, out=None, err=None, poll_s=0.05):
    """Run `argv` as `world` rank processes.  Rank 0's stdout goes to `out` line by line (default
    sys.stdout), every rank's stderr (and the other ranks' stdout) to `err` prefixed with the
    rank.  Returns the exit code: 0 when every rank returned 0, otherwise the worst child code
    (a signal death counts as 128 + signal), 124 when the deadline passed.  As soon as one rank
    fails the others are ended too — a rank that died inside a collective would leave its peers
    waiting for the store timeout."""
    out = sys.stdout if out is None else out
    err = sys.stderr if err is None else err
    port = free_port()
    procs, pumps = [], []
    lock = threading.Lock()

    def pump(stream, sink, prefix):
        for line in iter(stream.readline, ""):
            with lock:
                sink.write(prefix + line if prefix else line)
                sink.flush()
        stream.close()

    for rank in range(world):
        p = subprocess.Popen(argv, env=child_env(rank, world, port), stdout=subprocess.PIPE, stderr=subprocess.PIPE,
                             text=True, bufsize=1, start_new_session=True)
        procs.append(p)
        pumps.append(threading.Thread(target=pump, args=(p.stdout, out if rank == 0 else err,
                                                         "" if rank == 0 else "[rank %d] " % rank), daemon=True))
        pumps.append(threading.Thread(target=pump, args=(p.stderr, err, "[rank %d] " % rank), daemon=True))
    for t in pumps:
        t.start()

    def end_all():
        for p in procs:                              # exact process groups we started, never a pattern
            if p.poll() is None:
                try:
                    os.killpg(p.pid, signal.SIGTERM)
                except (ProcessLookupError, PermissionError):
                    pass
        t_end = time.monotonic() + 5.0
        for p in procs:
            while p.poll() is None and time.monotonic() < t_end:
                time.sleep(poll_s)
            if p.poll() is None:
                try:
                    os.killpg(p.pid, signal.SIGKILL)
                except (ProcessLookupError, PermissionError):
                    pass
                p.wait()

    deadline = time.monotonic() + float(timeout_s)
    code = 0
    while True:
        states = [p.poll() for p in procs]
        bad = [s for s in states if s not in (None, 0)]
        if bad:
            code = max(128 - s if s < 0 else s for s in bad)
            end_all()
            break
        if all(s == 0 for s in states):
            break
        if time.monotonic() > deadline:
            with lock:
                err.write("launch: deadline of %.0f s passed, ending %d rank processes\n" % (timeout_s, world))
                err.flush()
            end_all()
            code = 124
            break
        time.sleep(poll_s)
    for t in pumps:
        t.join(timeout=5.0)
    return code
