"""The N > 1 path on CPU: two gloo ranks pull pixel blocks from the shared counter
(MPICoordinator::getBlock semantics), render them with the CPU restatement into
zero-initialised full frames, and one reduce sums them onto rank 0.  The result must be
bit-identical to the single-process frame (blocks are disjoint; the rest is exactly 0)."""
import os
import socket
import subprocess
import sys
import tempfile

import numpy as np
import pytest

from wurblpt_amd import blocks, host

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = r'''
import os, sys
sys.path.insert(0, %(root)r)
import numpy as np, torch, torch.distributed as dist
from wurblpt_amd import blocks, host, scenefile
from tests import oracle_loader
dist.init_process_group("gloo")
rank, world = dist.get_rank(), dist.get_world_size()
W, H, S = 48, 40, 3
# one builder: rank 0 builds and flattens the scene, the others map its file (their host library never builds anything)
built = []
def build():
    built.append(rank)
    return host.cornell(W, H, 1, 2)
sc = scenefile.build_once(build, sys.argv[1] + ".scene", rank, dist.barrier, dist.broadcast_object_list)
assert built == ([0] if rank == 0 else []), built
assert isinstance(sc, scenefile.FileScene) == (rank != 0)
assert not os.path.exists(sys.argv[1] + ".scene")      # removed as soon as every rank has mapped it
# a build that fails on rank 0 raises on EVERY rank instead of leaving the others at a barrier
def broken():
    raise ValueError("no such scene")
try:
    scenefile.build_once(broken, sys.argv[1] + ".broken", rank, dist.barrier, dist.broadcast_object_list)
    raise SystemExit("a failed build went unnoticed on rank " + str(rank))
except RuntimeError as e:
    assert "no such scene" in str(e), e
# what bench.py reports per rank for N > 1 (one all_gather)
stats = blocks.rank_stats(10.0 + rank, 0.5 * (rank + 1), 1000 * (rank + 1))
assert stats["kernel_ms_per_step"] == {"min": 10.0, "mean": 10.5, "max": 11.0, "all": [10.0, 11.0]}, stats
assert stats["reduce_ms_per_step"] == {"max": 1.0, "all": [0.5, 1.0]} and len(stats["pixels_per_lane"]) == world
assert abs(stats["pixels_per_lane"][1] - 2000.0 / blocks.LANES_PER_GPU) < 1e-12
orc = oracle_loader.load("portable")
frame = torch.zeros((H, W, 3), dtype=torch.float32)
store = dist.distributed_c10d._get_default_store()
bs = blocks.plan_block_size(W * H, W, world, 2, min_block=96)
queue = blocks.BlockQueue(W * H, bs, store, "q0") if sys.argv[2] == "counter" else blocks.InterleavedBlocks(W * H, bs, rank, world)
def render_block(worker, start, size):
    part, _ = orc.render(sc, S, block=(start, size), threads=1)
    flat = frame.view(-1, 3)
    flat[start:start + size] = torch.from_numpy(part.reshape(-1, 3)[start:start + size])
mine = blocks.render_sharded(queue, render_block, 2)
counts = torch.tensor([len(mine), sum(s for _, s in mine)], dtype=torch.int64)
dist.all_reduce(counts)
blocks.reduce_frame(frame, dst=0)
if rank == 0:
    np.save(sys.argv[1], frame.numpy())
    np.save(sys.argv[1] + ".counts.npy", counts.numpy())
    np.save(sys.argv[1] + ".bs.npy", np.array([bs, queue.n_blocks]))
dist.barrier()
dist.destroy_process_group()
'''


def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def test_plan_block_size():
    assert blocks.plan_block_size(1024 * 1024, 1024, 1, 1) == 1024 * 1024
    assert blocks.plan_block_size(1024 * 1024, 1024, 1, 1, blocks_per_worker=2) == 1024 * 1024 // 2
    bs = blocks.plan_block_size(1024 * 1024, 1024, 8, 8)  # one block per worker: 64 strips of 16 rows
    assert bs == 16 * 1024
    bs = blocks.plan_block_size(1920 * 1080, 1920, 8, 8)
    assert bs % (8 * 1920) == 0 and 4096 <= bs <= 1920 * 1080 // 64
    assert blocks.plan_block_size(100, 10, 8, 4) == 100  # never below the reference's 4096 -> whole frame


def test_block_queue_covers_the_frame_exactly_once():
    q = blocks.BlockQueue(1000, 96)
    seen = np.zeros(1000, np.int32)
    while True:
        b = q.get_block()
        if b is None:
            break
        seen[b[0]:b[0] + b[1]] += 1
    assert (seen == 1).all() and q.get_block() is None


def test_interleaved_blocks_cover_the_frame_exactly_once():
    seen = np.zeros(1000, np.int32)
    per_rank = []
    for rank in range(3):
        q = blocks.InterleavedBlocks(1000, 96, rank, 3)
        mine = []
        while True:
            b = q.get_block()
            if b is None:
                break
            seen[b[0]:b[0] + b[1]] += 1
            mine.append(b[0] // 96)
        per_rank.append(mine)
        assert q.get_block() is None
    assert (seen == 1).all()
    assert per_rank == [[0, 3, 6, 9], [1, 4, 7, 10], [2, 5, 8]]      # a rank's blocks are spread over the picture


@pytest.mark.parametrize("mode", ["counter", "interleaved"])
def test_two_ranks_gloo_match_single_process(oracle, mode):
    with tempfile.TemporaryDirectory() as d:
        script = os.path.join(d, "worker.py")
        open(script, "w").write(WORKER % {"root": ROOT})
        out = os.path.join(d, "frame.npy")
        env = dict(os.environ, MASTER_ADDR="127.0.0.1", OMP_NUM_THREADS="2")
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2",
               "--master-addr", "127.0.0.1", "--master-port", str(free_port()), script, out, mode]
        subprocess.run(cmd, check=True, env=env, timeout=600, cwd=ROOT, stdout=subprocess.DEVNULL, stderr=subprocess.PIPE)
        got = np.load(out)
        counts = np.load(out + ".counts.npy")
        bs, nb = np.load(out + ".bs.npy")
    sc = host.cornell(48, 40, 1, 2)
    ref, _ = oracle.render(sc, 3)
    assert counts[0] == nb and counts[1] == 48 * 40  # every block rendered exactly once over both ranks
    assert np.array_equal(got.view(np.uint32), ref.view(np.uint32))
