"""tests/launcher.py itself (CPU): a group whose later command cannot be started leaves nothing running, and a killed
child takes the processes it started with it (process groups, not just the direct child)."""
import os
import sys
import time

import pytest

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import launcher  # noqa: E402


def _alive(pid):
    try:
        os.kill(pid, 0)
    except ProcessLookupError:
        return False
    try:                              # a zombie still answers signal 0
        with open(f"/proc/{pid}/stat") as f:
            return f.read().rsplit(")", 1)[1].split()[0] != "Z"
    except OSError:
        return False


def test_failed_start_of_a_later_command_ends_the_earlier_ones(tmp_path):
    pidfile = tmp_path / "pid"
    sleeper = [sys.executable, "-c", f"import os,time; open({str(pidfile)!r},'w').write(str(os.getpid())); time.sleep(120)"]
    with pytest.raises(FileNotFoundError):
        launcher.run_group([{"argv": sleeper}, {"argv": [str(tmp_path / "no-such-executable")]}], timeout=60)
    deadline = time.monotonic() + 10
    while not pidfile.exists() and time.monotonic() < deadline:
        time.sleep(0.05)
    if pidfile.exists():              # (the first child may have been ended before it wrote its pid)
        pid = int(pidfile.read_text())
        while _alive(pid) and time.monotonic() < deadline:
            time.sleep(0.05)
        assert not _alive(pid)


def test_timeout_ends_grandchildren_too(tmp_path):
    pidfile = tmp_path / "grandchild"
    child = ("import subprocess, sys, time\n"
             f"p = subprocess.Popen([sys.executable, '-c', 'import time; time.sleep(120)'])\n"
             f"open({str(pidfile)!r}, 'w').write(str(p.pid))\n"
             "time.sleep(120)\n")
    t0 = time.monotonic()
    res = launcher.run_group([{"argv": [sys.executable, "-c", child]}], timeout=3)
    assert res[0]["returncode"] != 0 and time.monotonic() - t0 < 30
    pid = int(pidfile.read_text())
    deadline = time.monotonic() + 10
    while _alive(pid) and time.monotonic() < deadline:
        time.sleep(0.05)
    assert not _alive(pid)
