"""Child-process launcher of the test suite.

`conftest.py` starts this script once, when the pytest session starts -- before anything in the pytest process has
touched HIP -- and every child process a test needs afterwards (rank workers, bench.py, the C++ drivers under bin/,
hipcc for the stand-in transport) is started HERE, on request over a pipe, never by a fork of the pytest process
itself. Why: in round 2 the pytest parent died with SIGSEGV inside subprocess.Popen (fork / vfork of a process that
held a live HIP runtime, its ROCr worker threads, PyTorch and three OpenMP runtimes). This process
imports neither HIP nor torch nor numpy, has one thread, and is all a GPU test ever forks from.

Protocol (one JSON document per line on stdin / stdout):
  request  {"cmds": [{"argv": [...], "env": {...} | null, "cwd": str | null}, ...], "timeout": seconds}
  reply    {"results": [{"returncode": int, "stdout": str, "stderr": str}, ...]}
The commands of one request run side by side; when one of them fails the others are given `grace` seconds and are
then killed (the way torch.multiprocessing.start_processes(join=True) treats its workers).
"""
import json
import os
import signal
import subprocess
import sys
import tempfile
import time


def _kill_group(p):
    """Ends the process group child `p` leads (start_new_session=True below): the child AND whatever it started in turn --
    the ranks under torch.distributed.run, hipcc's subprocesses -- which would otherwise keep the GPU. Exactly the group
    created here, nothing found by pattern."""
    try:
        os.killpg(p.pid, signal.SIGKILL)
    except (ProcessLookupError, PermissionError):
        pass
    if p.poll() is None:
        p.kill()


def run_group(cmds, timeout, grace=20.0):
    procs, files = [], []
    try:
        for c in cmds:
            out, err = tempfile.TemporaryFile(), tempfile.TemporaryFile()   # files, not pipes: no reader threads needed
            files.append((out, err))
            procs.append(subprocess.Popen(c["argv"], env=c.get("env"), cwd=c.get("cwd"), stdin=subprocess.DEVNULL,
                                          stdout=out, stderr=err, start_new_session=True))
    except Exception:
        # a later command could not be started (missing executable): the ranks already running would sit in their
        # rendezvous holding the GPU -- end them, reap them, drop their files, then report the error
        for p in procs:
            _kill_group(p)
        for p in procs:
            p.wait()
        for out, err in files:
            out.close()
            err.close()
        raise
    deadline = time.monotonic() + timeout
    failed_at = None
    while any(p.poll() is None for p in procs):
        now = time.monotonic()
        if failed_at is None and any(p.returncode not in (None, 0) for p in procs):
            failed_at = now
        if now > deadline or (failed_at is not None and now > failed_at + grace):
            for p in procs:
                if p.poll() is None:
                    _kill_group(p)
            break
        time.sleep(0.02)
    results = []
    for p, (out, err) in zip(procs, files):
        p.wait()
        if p.returncode != 0:
            _kill_group(p)          # a failed or killed child may have left grandchildren in its group
        texts = []
        for f in (out, err):
            f.seek(0)
            texts.append(f.read().decode("utf-8", errors="replace"))
            f.close()
        results.append({"returncode": p.returncode, "stdout": texts[0], "stderr": texts[1]})
    return results


def main():
    for line in sys.stdin:                      # EOF (the pytest process is gone) ends the loop
        line = line.strip()
        if not line:
            continue
        req = json.loads(line)
        try:
            reply = {"results": run_group(req["cmds"], float(req.get("timeout", 600)))}
        except Exception as exc:                # e.g. an executable that does not exist
            reply = {"error": f"{type(exc).__name__}: {exc}"}
        sys.stdout.write(json.dumps(reply) + "\n")
        sys.stdout.flush()


if __name__ == "__main__":
    main()
