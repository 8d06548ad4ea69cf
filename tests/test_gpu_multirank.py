"""Two ranks on one GPU: the N > 1 path of bench.py end to end (set_shard -> render -> pack -> all-gather -> unpack),
with the gathered frame compared bit for bit with an unsharded render.  The exchange goes through gloo and
host memory here (MTR_BENCH_BACKEND=gloo): one card cannot host two RCCL ranks; the driver's 2/4/8-GPU runs use nccl."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.gpu
def test_two_ranks_one_gpu_gathered_frame_matches_unsharded():
    env = dict(os.environ, MTR_BENCH_BACKEND="gloo", MASTER_ADDR="127.0.0.1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", "29533", os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1", "--verify",
           "--config-steps", "3", "--config-instances", "64"]
    r = subprocess.run(cmd, env=env, cwd=ROOT, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    for leg in ("headline", "C4", "C5"):  # --verify covers the frames of the config legs too, on both ranks
        assert r.stderr.count(f"verify {leg}: gathered frame == unsharded frame: True") == 2, (leg, r.stderr[-2000:])
    line = [l for l in r.stdout.splitlines() if l.startswith("{")][-1]
    out = json.loads(line)
    assert out["n_gpus"] == 2 and out["value"] > 0 and out["config"]["sharding"].startswith("bands of bin rows [0, ")
    assert out["frame_stats"]["chunks_culled"] > 0  # rank 0 skipped the geometry of the other rank's band
    assert out["repetitions"] >= 1 and out["steps"] == 3
    # BASELINE configs 4 and 5 are multi-GPU configs: an N > 1 line carries them, sharded next to the same scene unsharded
    # in the same run, so that one SCALE record is self-contained (here: 64 / 8 instances, gloo through host memory)
    assert out["verified"] == {"headline": True, "C4": True, "C5": True}
    for leg in ("C4", "C5"):
        c = out["configs"][leg]
        assert c["unsharded"]["ms_per_frame"] > 0 and c["sharded"]["ms_per_frame"] > 0 and c["sharded"]["n_gpus"] == 2
        assert abs(c["efficiency"] - c["unsharded"]["ms_per_frame"] / (2 * c["sharded"]["ms_per_frame"])) < 1e-3
        assert len(c["sharded"]["bands"]) == 3 and c["sharded"]["ownership"] == "bands-balanced"


@pytest.mark.gpu
@pytest.mark.parametrize("ownership", ["interleaved", "supertiles", "bands-balanced"])
def test_two_ranks_one_gpu_other_ownership_maps(ownership):
    env = dict(os.environ, MTR_BENCH_BACKEND="gloo", MASTER_ADDR="127.0.0.1", HSA_ENABLE_IPC_MODE_LEGACY="0", MTR_BENCH_OWNERSHIP=ownership)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", "29534", os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1", "--verify", "--configs", "off"]
    r = subprocess.run(cmd, env=env, cwd=ROOT, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    assert r.stderr.count("verify headline: gathered frame == unsharded frame: True") == 2, r.stderr[-2000:]
