"""One-node rank launcher without a port to lose: `python -m cbgbench_amd.launch --nproc N [-m module | script.py] args...`.

The N ranks of a sampling / training job (one process per GPU; SURVEY.md 8e, the reference's pocket loop sample.py:159) meet through a
torch.distributed FileStore on a path created here (`CBGX_RDZV_FILE`, read by sharding.init_process_group), so no TCP port is chosen
ahead of time by anybody: there is no pick-a-port-then-close window for another process to take (the EADDRINUSE that stopped round 5's
GPU suite), and any number of launches can run on one box at once.  gloo / RCCL still open their own data sockets, each bound to port 0
by the library that listens on it.

Under an external launcher (`python -m torch.distributed.run ... --master-port P bench.py`, the driver's N-GPU entry) nothing here is
used: RANK / WORLD_SIZE / MASTER_* come from that launcher and sharding.init_process_group takes the env:// route.
"""
import os
import shutil
import subprocess
import sys
import tempfile
import time

RETRY_MARKERS = ("EADDRINUSE", "address already in use", "Address already in use")


def rendezvous_dir():
    """a private directory for one job's FileStore (the store creates the file itself)"""
    return tempfile.mkdtemp(prefix="cbgx_rdzv_")


def rank_env(rank, world, rdzv_file, base=None):
    env = dict(os.environ if base is None else base)
    for k in ("MASTER_ADDR", "MASTER_PORT", "TORCHELASTIC_RUN_ID"):      # an outer launcher's rendezvous is not this job's
        env.pop(k, None)
    env.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), LOCAL_WORLD_SIZE=str(world), CBGX_RDZV_FILE=rdzv_file)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")      # dmabuf IPC only on this pool (RCCL / tensor sharing across processes)
    env.setdefault("GLOO_SOCKET_IFNAME", "lo")             # one node: the container hostname may not resolve
    return env


def _tail(path, n=4000):
    try:
        with open(path, "rb") as f:
            f.seek(0, 2)
            size = f.tell()
            f.seek(max(0, size - n))
            return f.read().decode("utf-8", "replace")
    except OSError:
        return ""


def _run_once(n, argv, env, poll_s=0.05):
    """start the n ranks, wait for all; the first failing rank ends the others (its peers would otherwise sit in a collective until the
    store times out).  Returns (exit code, {rank: stderr tail})."""
    d = rendezvous_dir()
    procs, logs = [], []
    try:
        for r in range(n):
            log = open(os.path.join(d, f"rank{r}.stderr"), "wb")
            logs.append(log)
            # rank 0's stdout is the job's stdout (the one JSON line); stderr goes to a file AND is replayed below
            procs.append(subprocess.Popen(argv, env=rank_env(r, n, os.path.join(d, "store"), env), stderr=log))
        code, failed = 0, None
        live = set(range(n))
        while live:
            for r in sorted(live):
                rc = procs[r].poll()
                if rc is None:
                    continue
                live.discard(r)
                if rc != 0 and code == 0:
                    code, failed = rc, r
            if code != 0:
                break
            time.sleep(poll_s)
        if code != 0:
            deadline = time.time() + 10
            for r in sorted(live):
                procs[r].terminate()           # exact PIDs this function started
            for r in sorted(live):
                try:
                    procs[r].wait(timeout=max(0.1, deadline - time.time()))
                except subprocess.TimeoutExpired:
                    procs[r].kill()
                    procs[r].wait()
        for log in logs:
            log.close()
        tails = {r: _tail(os.path.join(d, f"rank{r}.stderr")) for r in range(n)}
        return code, failed, tails
    finally:
        for log in logs:
            if not log.closed:
                log.close()
        shutil.rmtree(d, ignore_errors=True)


def launch(n, argv, env=None, retries=3, out=sys.stderr):
    """Run `argv` as n ranks on this node.  Returns the job's exit code (0 = every rank exited 0).  A failure whose stderr names an
    address collision is retried (up to `retries` launches in total); every failure prints the failing rank's stderr tail -- the
    traceback is never reduced to a bare exit code."""
    code = 1
    for attempt in range(1, max(1, retries) + 1):
        code, failed, tails = _run_once(n, list(argv), env)
        if code == 0:
            for r in sorted(tails):         # warnings of a passing run stay visible, as they would without the capture
                out.write(tails[r])
            return 0
        out.write(f"[cbgx launch] rank {failed} of {n} exited with code {code} (attempt {attempt}); its stderr tail:\n{tails.get(failed, '')}\n")
        for r in sorted(tails):
            if r != failed and tails[r].strip():
                out.write(f"[cbgx launch] rank {r} stderr tail:\n{tails[r][-1500:]}\n")
        if not any(m in t for t in tails.values() for m in RETRY_MARKERS):
            break
        out.write("[cbgx launch] address collision in a library-owned socket; launching again\n")
        time.sleep(0.5 * attempt)
    return code


def main(argv=None):
    argv = list(sys.argv[1:] if argv is None else argv)
    usage = "usage: python -m cbgbench_amd.launch --nproc N [-m module | script.py] args..."
    if len(argv) < 3 or argv[0] not in ("--nproc", "--nproc-per-node"):
        raise SystemExit(usage)
    n = int(argv[1])
    rest = argv[2:]
    if rest and rest[0] == "--":
        rest = rest[1:]
    if not rest:
        raise SystemExit(usage)
    return launch(n, [sys.executable] + rest)


if __name__ == "__main__":
    sys.exit(main())
