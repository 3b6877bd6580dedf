"""Pixel-block distribution over the GPUs of one node.

Mirror of the reference's MPICoordinator (mpi.hpp:152-289): the frame is cut into blocks of
consecutive pixel indices, workers pull the next block index from one shared counter
(getBlock), render it into their own zero-initialised full frame (submitBlock), and the frames
are summed onto rank 0 at the end.  Here the counter lives in the c10d store of
torch.distributed (an atomic fetch-add, so no coordinator thread is needed), one process
drives one GPU, and the final gather is ONE reduce over RCCL/xGMI.  Blocks are disjoint and the
rest of every frame is exactly 0.0f, so the sum is exact in any reduction order.

On GPUs the default hand-out is static instead (InterleavedBlocks): with one block per worker there is nothing
left to balance once every worker holds its block, so which blocks a rank gets decides the balance.  Block i goes
to rank i mod N: every rank's strips are spread evenly over the picture (measured on the Cornell frame, 64 strips
over 8 ranks: slowest rank / mean 1.04 against a median of 1.10 for the order in which a shared counter happens to
hand them out, and 1.27 for contiguous eighths).

Backend agnostic: tests run it with gloo on CPU tensors and the CPU restatement as renderer."""
import threading


def plan_block_size(pixels, width, world, workers_per_rank, min_block=4096, blocks_per_worker=1):
    """Whole rows, about `blocks_per_worker` blocks per worker, never below the reference's
    default block of 4096 pixels (mpi.hpp:178), and a multiple of 8 rows where possible so that
    the kernel can map waves to 8x8 pixel tiles.  One block per worker by default: a lane owns a
    pixel for all its samples, so a GPU is only as busy as the number of its pixels that are in
    flight -- with all workers' blocks launched at once (one HIP stream each) every pixel of the
    rank's share is resident from the start, and the strips of a rank interleave over the image."""
    size = max(min_block, -(-pixels // max(1, world * workers_per_rank * blocks_per_worker)))
    rows = -(-size // width)
    if rows > 8:
        rows -= rows % 8
    return min(pixels, rows * width)


class BlockQueue:
    """MPICoordinator::getBlock over a shared counter."""

    def __init__(self, pixels, block_size, store=None, key="wpt_blocks"):
        self.pixels = pixels
        self.block_size = block_size
        self.n_blocks = -(-pixels // block_size)
        self._store = store
        self._key = key
        self._local = 0
        self._lock = threading.Lock()

    def get_block(self):
        """Returns (start, size) or None when the frame is handed out (blockSize == 0 in the reference)."""
        if self._store is not None:
            index = self._store.add(self._key, 1) - 1
        else:
            with self._lock:
                index = self._local
                self._local += 1
        if index >= self.n_blocks:
            return None
        start = index * self.block_size
        return start, min(self.block_size, self.pixels - start)


class InterleavedBlocks:
    """Static hand-out: block i belongs to rank i mod world; a rank's workers take its blocks in order.
    Same interface as BlockQueue (get_block, n_blocks)."""

    def __init__(self, pixels, block_size, rank=0, world=1):
        self.pixels = pixels
        self.block_size = block_size
        self.n_blocks = -(-pixels // block_size)
        self._mine = list(range(rank, self.n_blocks, world))
        self._next = 0
        self._lock = threading.Lock()

    def get_block(self):
        with self._lock:
            if self._next >= len(self._mine):
                return None
            index = self._mine[self._next]
            self._next += 1
        start = index * self.block_size
        return start, min(self.block_size, self.pixels - start)


def render_sharded(queue, render_block, workers):
    """Each worker (e.g. one per HIP stream) pulls blocks until the queue is empty.
    render_block(worker_index, start, size) must return after the block is in this rank's frame.
    Returns the list of (start, size) this rank rendered."""
    done = []
    errors = []
    lock = threading.Lock()

    def run(w):
        try:
            while True:
                b = queue.get_block()
                if b is None:
                    break
                render_block(w, b[0], b[1])
                with lock:
                    done.append(b)
        except Exception as exc:  # surfaced on the calling thread
            errors.append(exc)

    if workers == 1:
        run(0)
    else:
        threads = [threading.Thread(target=run, args=(w,)) for w in range(workers)]
        for t in threads:
            t.start()
        for t in threads:
            t.join()
    if errors:
        raise errors[0]
    return done


def reduce_frame(frame, dst=0):
    """Final gather: sum of the per-rank frames onto rank `dst` (RCCL over xGMI for CUDA tensors).  Whenever a process group
    exists the collective runs, also over a single rank (bench.py --force-distributed: the communicator and the reduce are RCCL's
    on the one GPU of a test box)."""
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized():
        if frame.is_cuda and dist.get_backend() == "gloo":
            dist.all_reduce(frame, op=dist.ReduceOp.SUM)  # gloo has no reduce for device tensors (rehearsals only)
        else:
            dist.reduce(frame, dst=dst, op=dist.ReduceOp.SUM)
    return frame


LANES_PER_GPU = 256 * 4 * 4 * 64  # lanes an MI355X holds at four waves per SIMD (256 CUs x 4 SIMDs x 4 waves x 64)


def rank_stats(kernel_ms_per_step, reduce_ms_per_step, my_pixels, device="cpu"):
    """What a step's time is made of, rank by rank (one all_gather): each rank's render time per step, its reduce time
    per step and how many of its pixels there are per lane of its GPU -- a pixel is one serial sequence of samples, so
    below one pixel per lane a GPU cannot be filled.  Every rank gets the same record; a single process gets its own."""
    import torch
    import torch.distributed as dist
    mine = torch.tensor([float(kernel_ms_per_step), float(reduce_ms_per_step), float(my_pixels) / LANES_PER_GPU], dtype=torch.float64, device=device)
    rows = [mine]
    if dist.is_available() and dist.is_initialized():
        rows = [torch.zeros_like(mine) for _ in range(dist.get_world_size())]
        dist.all_gather(rows, mine)
    rows = [[float(x) for x in r.cpu().tolist()] for r in rows]
    kms = [r[0] for r in rows]
    return {"kernel_ms_per_step": {"min": min(kms), "mean": sum(kms) / len(kms), "max": max(kms), "all": kms},
            "reduce_ms_per_step": {"max": max(r[1] for r in rows), "all": [r[1] for r in rows]},
            "pixels_per_lane": [r[2] for r in rows],
            "note": "pixels_per_lane: a rank's pixels over the %d lanes its GPU holds at four waves per SIMD; a pixel is one serial "
                    "sequence of samples, so below 1 the GPU cannot be filled" % LANES_PER_GPU}
