"""CPU, world_size 2 over gloo: block-range sharding + the one collective of the path (gather of
per-block compressed sizes -> identical footer index on every rank).  SURVEY.md §8(e)."""
import os
import socket
import sys

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import helpers


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def _worker(rank, world, port, nblocks, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    sys.path.insert(0, helpers.ROOT)
    p = helpers.pkg()
    rng = np.random.default_rng(99)
    all_c = rng.integers(1, 4 << 20, nblocks)                       # the "true" per-block csizes
    lo, hi = p.shard_range(nblocks, rank, world)
    local = torch.from_numpy(all_c[lo:hi].astype(np.int32))
    cs, offs = p.gather_block_index(local, nblocks)
    q.put((rank, lo, hi, cs.tolist(), offs.tolist()))
    dist.barrier(); dist.destroy_process_group()


def test_gather_block_index_world2():
    for nblocks in (7, 8, 1):
        ctx = mp.get_context("spawn"); q = ctx.Queue(); port = _free_port()
        ps = [ctx.Process(target=_worker, args=(r, 2, port, nblocks, q)) for r in range(2)]
        [p.start() for p in ps]
        got = sorted(q.get(timeout=120) for _ in range(2))
        [p.join(60) for p in ps]
        assert all(p.exitcode == 0 for p in ps)
        rng = np.random.default_rng(99); all_c = rng.integers(1, 4 << 20, nblocks)
        want_off = helpers.pkg().container.block_offsets(all_c).tolist()
        covered = []
        for rank, lo, hi, cs, offs in got:
            assert cs == all_c.tolist() and offs == want_off, rank     # same index on every rank
            covered += list(range(lo, hi))
        assert covered == list(range(nblocks))                          # ranges partition the blocks
        # the gathered index is exactly what the footer stores
        p = helpers.pkg()
        foot = p.frame_footer(p.MAGIC_4MC, np.array(want_off, dtype=np.uint64))
        assert list(p.parse_footer(foot, p.MAGIC_4MC)) == want_off


def test_shard_range_partitions():
    p = helpers.pkg()
    for n in (0, 1, 5, 8, 2048, 16384, 16385):
        for w in (1, 2, 3, 8):
            r = [p.shard_range(n, k, w) for k in range(w)]
            assert r[0][0] == 0 and r[-1][1] == n
            assert all(r[i][1] == r[i + 1][0] for i in range(w - 1))
