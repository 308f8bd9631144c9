"""bench.py --gpus N started plainly (no torch.distributed.run): the parent starts N rank processes itself,
relays rank 0's one JSON line and refuses a line that reports another n_gpus. CPU-side checks of that launcher
(the ranks here are a stub script; the real two-rank run through the stand-in transport is in test_dist_gpu.py)."""
import argparse
import json
import os
import subprocess
import sys
import textwrap

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def _stub(tmp_path, body):
    p = tmp_path / "rank_stub.py"
    p.write_text(textwrap.dedent(body))
    return str(p)


def test_launcher_gives_every_rank_the_torchrun_environment_and_relays_one_line(tmp_path, capsys):
    import bench
    stub = _stub(tmp_path, """
        import json, os, sys
        rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
        assert os.environ["LOCAL_RANK"] == str(rank) and os.environ["MASTER_ADDR"] == "127.0.0.1"
        assert int(os.environ["MASTER_PORT"]) > 0 and len(os.environ["BDG_LAUNCH_NONCE"]) == 16
        open(os.path.join(sys.argv[1], f"rank{rank}"), "w").write(os.environ["BDG_LAUNCH_NONCE"])
        print("chatter that is not the line")
        print(json.dumps({"n_gpus": world, "rank": rank, "argv": sys.argv[2:]}))
    """)
    bench.launch_ranks(argparse.Namespace(gpus=3), script=stub, argv=[str(tmp_path), "--steps", "7"])
    out = [ln for ln in capsys.readouterr().out.splitlines() if ln.strip()]
    assert len(out) == 1
    d = json.loads(out[0])
    assert d == {"n_gpus": 3, "rank": 0, "argv": ["--steps", "7"]}
    nonces = {(tmp_path / f"rank{r}").read_text() for r in range(3)}
    assert len(nonces) == 1  # one launch, one nonce, three ranks started


def test_launcher_refuses_a_line_with_another_gpu_count(tmp_path):
    import bench
    stub = _stub(tmp_path, """
        import json
        print(json.dumps({"n_gpus": 1}))
    """)
    with pytest.raises(SystemExit) as e:
        bench.launch_ranks(argparse.Namespace(gpus=2), script=stub, argv=[])
    assert "n_gpus=1" in str(e.value)


def test_launcher_propagates_a_failing_rank(tmp_path):
    import bench
    stub = _stub(tmp_path, """
        import json, os, sys
        if os.environ["RANK"] == "1":
            sys.exit(3)
        print(json.dumps({"n_gpus": 2}))
    """)
    with pytest.raises(SystemExit) as e:
        bench.launch_ranks(argparse.Namespace(gpus=2), script=stub, argv=[])
    assert e.value.code == 3


def test_gpus_flag_and_world_size_must_agree():
    """Under a launcher that set WORLD_SIZE, a different --gpus is an error, never a mislabelled line."""
    env = dict(os.environ, RANK="0", LOCAL_RANK="0", WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT="1")
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "4", "--steps", "1", "--warmup", "0"],
                         capture_output=True, text=True, env=env, timeout=300)
    assert out.returncode != 0 and "refusing to mislabel" in out.stderr and "{" not in out.stdout


def test_file_rendezvous_ignores_a_record_older_than_the_launcher(tmp_path, monkeypatch):
    """A 136-byte record left behind by an earlier launch under the same name is not accepted."""
    import struct
    import threading
    import time

    from blitzdg_amd import halo
    monkeypatch.setenv("BDG_RENDEZVOUS_DIR", str(tmp_path))
    monkeypatch.setenv("MASTER_PORT", "4242")
    monkeypatch.setenv("BDG_LAUNCH_NONCE", "feedfacefeedface")
    base = tmp_path / f"bdg_rccl_{os.getuid()}"
    base.mkdir(mode=0o700)
    stale = base / f"id_{os.getppid()}_4242_2_feedfacefeedface"
    stale.write_bytes(struct.pack("<d", 1000.0) + b"S" * 128)
    fresh = bytes(range(128))

    def rank0():
        time.sleep(0.3)
        halo.file_rendezvous(0, 2, lambda: fresh)
    t = threading.Thread(target=rank0)
    t.start()
    uid, path = halo.file_rendezvous(1, 2, None, timeout=20.0)
    t.join()
    assert uid == fresh and path == str(stale)
    assert (os.stat(path).st_mode & 0o777) == 0o600


def test_a_timed_out_in_kernel_wait_makes_every_rank_measure_again_with_event_waits(monkeypatch, capsys):
    """bench.py --gpus N, one rank's view: the barrier after the warm-up reports that an in-kernel wait between the two launches of a
    stage gave up somewhere (the library reports it from every rank's barrier) -- the measurement starts over from the initial state
    with BDG_SW2D_EVENT_SYNC=1 and the line says which form of the dependencies was measured. Any other error is not retried."""
    import bench
    import blitzdg_amd.halo as halo
    import blitzdg_amd._capi as capi

    class Stub:
        global_elements, Np = 1000, 15

        def __init__(self, fail_with):
            self.fail_with, self.calls, self.uploads = fail_with, [], 0

        def set_initial_state(self, fn): self.uploads += 1
        def compute_dt(self, cfl): return 1e-3
        def lserk4_stages(self, dt, n): self.calls.append((n, os.environ.get("BDG_SW2D_EVENT_SYNC")))

        def barrier(self):
            if self.fail_with and len(self.calls) == 1:
                raise RuntimeError(self.fail_with)

        def allreduce_max(self, v): return max(v, 1e-3)
        def allreduce_sum(self, v): return v
        def owned_mass(self, fn=None): return 1.0
        def halo_counts(self): return {"owned": 500}
        def close(self): pass

    class Lib:
        @staticmethod
        def bdg_device_count(): return 1

    monkeypatch.setattr(capi, "lib", Lib, raising=False)
    monkeypatch.setenv("RANK", "0")
    monkeypatch.setenv("WORLD_SIZE", "2")
    monkeypatch.delenv("BDG_SW2D_EVENT_SYNC", raising=False)
    args = argparse.Namespace(steps=7, warmup=3, gpus=2)
    stub = Stub("on another rank: an in-kernel wait between the interior and the partition-boundary launch of an exchanged stage timed out")
    monkeypatch.setattr(halo.NativeDistributedSw2d, "box", classmethod(lambda cls, *a, **k: stub))
    bench.run_distributed_native(args)
    line = json.loads([ln for ln in capsys.readouterr().out.splitlines() if ln.startswith("{")][-1])
    assert stub.uploads == 2 and stub.calls == [(3, None), (3, "1"), (7, "1")]
    assert line["n_gpus"] == 2 and line["config"]["stage_dependencies"].startswith("events (an in-kernel wait timed out")
    monkeypatch.delenv("BDG_SW2D_EVENT_SYNC", raising=False)
    os.environ.pop("BDG_SW2D_EVENT_SYNC", None)

    clean = Stub(None)
    monkeypatch.setattr(halo.NativeDistributedSw2d, "box", classmethod(lambda cls, *a, **k: clean))
    bench.run_distributed_native(args)
    line = json.loads([ln for ln in capsys.readouterr().out.splitlines() if ln.startswith("{")][-1])
    assert clean.uploads == 1 and clean.calls == [(3, None), (7, None)] and line["config"]["stage_dependencies"].startswith("in-kernel")

    broken = Stub("hipErrorLaunchFailure")
    monkeypatch.setattr(halo.NativeDistributedSw2d, "box", classmethod(lambda cls, *a, **k: broken))
    with pytest.raises(RuntimeError, match="hipErrorLaunchFailure"):
        bench.run_distributed_native(args)
    assert broken.uploads == 1
