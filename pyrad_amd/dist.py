"""One process per GPU: rank discovery, the out-of-band exchange of the RCCL unique id, and
the host-side shard arithmetic of the contiguous-range grid sharding.

The data path has exactly one collective, the RCCL all-gather of the per-rank spectrum
shards (``lbl_allgather_dev``).  Everything here is control plane: it moves 128 bytes once.
The launcher contract is torchrun's (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT
in the environment); the ranks of one node share a filesystem, so the id travels through a
file — no second HIP runtime (PyTorch's bundled one) is pulled into the process.
"""
from __future__ import annotations

import os
import time

import numpy as np


def env_world():
    """(rank, local_rank, world_size) from the launcher's environment (1 process if unset)."""
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", str(rank)))
    return rank, local, world


class FileRendezvous:
    """Single-node rendezvous through a directory.  The key is unique per launch: all workers
    of one torchrun share their parent (the elastic agent) and the master port."""

    def __init__(self, rank: int, world: int, key: str | None = None, root: str | None = None, timeout: float = 120.0):
        self.rank, self.world, self.timeout = rank, world, timeout
        if key is None:
            key = "%s_%s_%s" % (os.environ.get("MASTER_PORT", "0"), os.environ.get("TORCHELASTIC_RUN_ID", "none"),
                                os.getppid())
        root = root or os.environ.get("PYRAD_RENDEZVOUS_DIR", "/tmp")
        self.dir = os.path.join(root, "pyrad_amd_rdzv_%s" % key)
        os.makedirs(self.dir, exist_ok=True)

    def broadcast(self, name: str, data: bytes | None) -> bytes:
        """Rank 0 publishes ``data``; every rank returns it."""
        path = os.path.join(self.dir, name)
        if self.rank == 0:
            tmp = path + ".tmp"
            with open(tmp, "wb") as f:
                f.write(data)
            os.replace(tmp, path)          # atomic publish
            return data
        t0 = time.time()
        while not os.path.isfile(path):
            if time.time() - t0 > self.timeout:
                raise TimeoutError("rendezvous: %s not published within %.0f s" % (path, self.timeout))
            time.sleep(0.005)
        with open(path, "rb") as f:
            return f.read()

    def arrive(self, name: str):
        """File barrier (control plane only; the timed region uses an RCCL barrier)."""
        open(os.path.join(self.dir, "%s.%d" % (name, self.rank)), "wb").close()
        t0 = time.time()
        while True:
            if all(os.path.isfile(os.path.join(self.dir, "%s.%d" % (name, r))) for r in range(self.world)):
                return
            if time.time() - t0 > self.timeout:
                raise TimeoutError("rendezvous barrier '%s' timed out" % name)
            time.sleep(0.005)

    def cleanup(self):
        """Every rank acknowledges that it is past its last wait; rank 0 removes the directory
        only after all acknowledgements (otherwise it could delete files a slower rank still polls)."""
        open(os.path.join(self.dir, "ack.%d" % self.rank), "wb").close()
        if self.rank != 0:
            return
        t0 = time.time()
        while not all(os.path.isfile(os.path.join(self.dir, "ack.%d" % r)) for r in range(self.world)):
            if time.time() - t0 > self.timeout:
                return                      # leave the files rather than hang
            time.sleep(0.005)
        try:
            for f in os.listdir(self.dir):
                os.remove(os.path.join(self.dir, f))
            os.rmdir(self.dir)
        except OSError:
            pass


# ----------------------------------------------------------------------------------------
# shard arithmetic (pure host logic; exercised with gloo on CPU in tests/test_sharding.py)
# ----------------------------------------------------------------------------------------
def shard_bounds(n: int, world_size: int, rank: int):
    """Equal contiguous shards of an n-point grid padded to world_size * S.
    Rank r owns [r*S, min((r+1)*S, n)).  Returns (S, first, count)."""
    S = -(-int(n) // int(world_size))
    first = min(rank * S, n)
    count = max(min((rank + 1) * S, n) - first, 0)
    return S, first, count


def halo_select(lines: dict, range_min, resolution, W, first, count):
    """Lines whose wing support (centre +- (W-2) points, cls:394) can reach grid points
    [first, first+count).  Selected by wavenumber with two grid steps of slack, so the
    truncation of cls:390 can never drop a contributing line; extra lines contribute nothing
    because the kernel clips every line to its exact support."""
    H = max(int(W) - 2, 0)
    lo = range_min + (first - H - 2) * resolution
    hi = range_min + (first + count + H + 2) * resolution
    nu = np.asarray(lines["nu"])
    m = (nu > lo) & (nu < hi)
    return {k: np.asarray(v)[m] for k, v in lines.items()}


def assemble_gathered(parts, n: int):
    """What the in-place all-gather leaves in the padded buffer: rank-ordered shards of S
    points each; the first n points are the spectrum."""
    return np.concatenate([np.asarray(p) for p in parts])[:n]
