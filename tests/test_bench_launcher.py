"""bench.py started plainly with --gpus N is its own launcher: the parent touches no GPU, starts N rank
processes as children with the torchrun environment, relays rank 0's JSON line and fails when a rank fails or hangs."""
import json
import os
import subprocess
import sys
import textwrap

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _script(tmp_path):
    p = tmp_path / "rank_stub.py"
    p.write_text(textwrap.dedent('''
        import json, os, sys, time
        r = int(os.environ["RANK"])
        assert os.environ["MASTER_ADDR"] == "127.0.0.1" and int(os.environ["MASTER_PORT"]) > 0 and os.environ["LOCAL_RANK"] == str(r)
        mode = sys.argv[1] if len(sys.argv) > 1 else "ok"
        if mode == "fail" and r == 1:
            sys.exit(3)
        if mode == "hang":
            time.sleep(100)
        if r == 0:
            print(json.dumps({"rank": r, "world": int(os.environ["WORLD_SIZE"])}))
    '''))
    return str(p)


def test_self_launch_relays_rank0_and_reports_failures(tmp_path, capfd):
    sys.path.insert(0, ROOT)
    import bench
    s = _script(tmp_path)
    assert bench.self_launch(3, argv=["ok"], script=s, timeout_s=60) == 0
    out = capfd.readouterr().out.strip().splitlines()
    assert json.loads(out[-1]) == {"rank": 0, "world": 3}
    assert bench.self_launch(2, argv=["fail"], script=s, timeout_s=60) == 3          # a dead rank ends the job, its code is returned
    assert bench.self_launch(2, argv=["hang"], script=s, timeout_s=1) == 124         # hung ranks are killed by PID


@pytest.mark.gpu
def test_bench_gpus_2_started_plainly():
    """`python bench.py --gpus 2 --steps 3`: two ranks (RCCL with two GPUs, gloo on a shared GPU otherwise), one JSON line."""
    import torch
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "2"], env=env,
                       capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    line = json.loads(r.stdout.strip().splitlines()[-1])
    assert line["n_gpus"] == 2 and line["config"]["rccl_ranks"] == 2 and line["steps"] == 3
    assert line["config"]["captions_per_step"] == 2 * 2 * 32 and line["value"] > 0
    assert line["config"]["dist_backend"] in ("nccl", "gloo")
