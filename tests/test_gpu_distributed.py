"""The N > 1 path of bench.py with RCCL underneath, on the one GPU of a test box: torch.distributed.run starts ONE rank and
bench.py --force-distributed takes the code path of N ranks -- init_process_group("nccl", device_id=...), one builder per node
(scene file in /dev/shm), this rank's interleaved bands in one launch (wpt_render_bands_device), blocks.reduce_frame on the
device tensor (an RCCL reduce over a one-rank communicator), blocks.rank_stats (an RCCL all_gather), the comparison with a single
launch and rows of the oracle.  MPICoordinator semantics (mpi.hpp:215-288: getBlock / submitBlock) as bands + one reduce."""
import json
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


@pytest.mark.parametrize("workload,extra", [("cornell_256x256_64spp_lambertian", []),
                                            ("sponza_like_1920x1080_256spp_envmap_is", ["--samples-sqrt", "2"])])
def test_one_rank_under_the_launcher_runs_the_nccl_path(workload, extra):
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    env.pop("WPT_BENCH_BACKEND", None)
    env.pop("WPT_BENCH_DEVICE", None)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=1", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py"), "--gpus", "1", "--force-distributed", "--steps", "2", "--warmup", "1",
           "--workload", workload, "--cpu-seconds", "3"] + extra
    r = subprocess.run(cmd, capture_output=True, timeout=900, cwd=ROOT, env=env)
    assert r.returncode == 0, r.stdout.decode()[-2000:] + r.stderr.decode()[-3000:]
    lines = [l for l in r.stdout.decode().splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout.decode()[-2000:]
    out = json.loads(lines[0])
    assert out["backend"] == "nccl" and out["n_gpus"] == 1
    assert out["frame_equals_single_launch"] is True
    assert out["parity"]["bits_differ"] == 0 and out["parity"]["pixels"] > 0
    assert out["frame_finite"] is True
    assert "bands of" in out["config"]["parallelism"] and "RCCL reduce" in out["config"]["parallelism"]
    assert len(out["per_rank"]["kernel_ms_per_step"]["all"]) == 1 and out["per_rank"]["reduce_ms_per_step"]["max"] >= 0.0
    assert out["value"] > 0 and out["scaling"] == "strong"
