"""bench.py's N > 1 path on the one-GPU box: two ranks on device 0, gloo with host staging in place of RCCL
(VRT_BENCH_REHEARSE=1).  Rank 0 itself checks that the frame assembled from the gathered row tiles equals an
unsharded render of the same passes bit for bit -- for the 1080p headline and for the two 3840x2160 configs BASELINE.json
defines on N GPUs (dense 128^3, dense 256^3); here: the run succeeds, says so, and prints the one JSON line."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.gpu
@pytest.mark.parametrize("stripes", [0, 32])
def test_bench_two_ranks_rehearsal(stripes):
    """stripes = 32: the same run with the frame split into interleaved 32-row stripes (vrt_set_row_stripes) instead of two contiguous
    row tiles -- SURVEY.md 8e's fallback partition."""
    env = dict(os.environ, VRT_BENCH_REHEARSE="1", MASTER_ADDR="127.0.0.1", VRT_BENCH_STRIPES=str(stripes))
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK"):
        env.pop(k, None)
    import socket
    sock = socket.socket()
    sock.bind(("127.0.0.1", 0))
    port = sock.getsockname()[1]
    sock.close()
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                        "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1"],
                       cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    for label in ("config 2", "config4_dense_4k", "config5_dense256_4k"):   # the 1080p headline and the two 4K frames defined on N GPUs
        assert f"[rehearsal] {label}: gathered frame == unsharded frame: True" in r.stderr, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["scaling"] == "strong" and out["steps"] == 2 and out["value"] > 0
    assert out["cpu_baseline"] is None and out["roofline"]["traffic"] is None
    assert sum(int(x) for x in out["config"]["sharding"].split("[")[1].split("]")[0].split(",")) == 1080
    assert ("interleaved 32-row stripes" in out["config"]["sharding"]) == (stripes == 32)
    sec = {s["name"]: s for s in out["secondary"]}
    assert set(sec) == {"config4_dense_4k_2gpu", "config5_dense256_4k_2gpu"}
    for s in sec.values():
        assert s["n_gpus"] == 2 and s["value"] > 0 and sum(s["tile_rows"]) == 2160
