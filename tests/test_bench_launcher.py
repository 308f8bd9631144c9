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
