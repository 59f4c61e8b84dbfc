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


# ---- a REAL .4mc file assembled by two ranks (4mc_amd/csrc/shard.c) ----------------------------------------------------
def _blocks_of_rank(data, lo, hi):
    """what a rank's GPU produces for its block range, restated by the oracle: (usize, csize, xxh32, payload) per block
    with the container's stored fallback (native/4mc.c:301-329)"""
    B = helpers.B
    out = []
    for b in range(lo, hi):
        src = data[b * B:(b + 1) * B]
        r, comp = helpers.orc_compress(src, len(src) - 1)
        payload = comp if 0 < r < len(src) else src
        out.append((len(src), len(payload), helpers.orc_xxh32(payload), np.ascontiguousarray(payload)))
    return out


def _file_worker(rank, world, port, path, q):
    import ctypes as C
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    sys.path.insert(0, helpers.ROOT)
    p = helpers.pkg(); L = p.lib()
    man = __import__("json").load(open(os.path.join(helpers.ROOT, "tests", "golden", "corpus_manifest.json")))
    data = helpers.corpus(man["corpus"]["bytes"], first_block=man["corpus"]["first_block"], seed=man["corpus"]["seed"])
    nblocks = -(-len(data) // helpers.B)
    first, count = C.c_uint64(), C.c_uint64()
    L.fourmc_shard_range(nblocks, rank, world, C.byref(first), C.byref(count))
    lo, n = first.value, count.value
    mine = _blocks_of_rank(data, lo, lo + n)
    # the one collective: compressed sizes, padded to equal counts per rank
    per = -(-nblocks // world)
    pad = torch.zeros(per, dtype=torch.int32); pad[:n] = torch.tensor([m[1] for m in mine], dtype=torch.int32)
    allc = torch.empty(per * world, dtype=torch.int32)
    dist.all_gather_into_tensor(allc, pad)
    cs_all = allc.numpy().astype(np.uint32)
    off_all = np.zeros(nblocks, np.uint64)
    L.fourmc_shard_offsets(cs_all.ctypes.data, nblocks, off_all.ctypes.data)
    usz = np.array([m[0] for m in mine], np.uint32); xs = np.array([m[2] for m in mine], np.uint32)
    pay = np.concatenate([m[3] for m in mine]) if mine else np.zeros(1, np.uint8)
    poff = np.concatenate([[0], np.cumsum([m[1] for m in mine])[:-1]]).astype(np.uint64) if mine else np.zeros(1, np.uint64)
    fd = os.open(path, os.O_WRONLY | os.O_CREAT, 0o644)
    rc = L.fourmc_shard_write(fd, p.MAGIC_4MC, rank, lo, n, nblocks, off_all.ctypes.data, cs_all.ctypes.data, usz.ctypes.data, xs.ctypes.data,
                              pay.ctypes.data, poff.ctypes.data)
    os.close(fd)
    dist.barrier()
    q.put((rank, rc, lo, n))
    dist.destroy_process_group()


def test_two_ranks_write_one_4mc_file(tmp_path):
    """both ranks pwrite their own byte ranges, rank 0 adds header / end mark / footer: the file equals the reference CLI's"""
    import hashlib, json
    helpers.oracle(); helpers.corpus_lib()                       # build the checkers once, not in both ranks at the same time
    man = json.load(open(os.path.join(helpers.ROOT, "tests", "golden", "corpus_manifest.json")))
    path = str(tmp_path / "two_ranks.4mc")
    # an older, LONGER file of that name: nothing of it may survive behind the footer (readers find the footer from the end)
    with open(path, "wb") as f:
        f.write(b"\xEE" * (man["levels"]["4mc-1"]["file_bytes"] + 12345))
    ctx = mp.get_context("spawn"); q = ctx.Queue(); port = _free_port()
    ps = [ctx.Process(target=_file_worker, args=(r, 2, port, path, q)) for r in range(2)]
    [p.start() for p in ps]
    got = sorted(q.get(timeout=300) for _ in range(2))
    [p.join(60) for p in ps]
    assert all(p.exitcode == 0 for p in ps) and all(g[1] == 0 for g in got), got
    assert got[0][2] == 0 and got[0][2] + got[0][3] == got[1][2]            # contiguous ranges
    blob = open(path, "rb").read()
    assert len(blob) == man["levels"]["4mc-1"]["file_bytes"]
    assert hashlib.sha256(blob).hexdigest() == man["levels"]["4mc-1"]["sha256"]


def _report_worker(rank, world, port, nblocks, break_rank, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    sys.path.insert(0, helpers.ROOT)
    p = helpers.pkg()
    rng = np.random.default_rng(7)
    all_c = rng.integers(1, 4 << 20, nblocks)
    lo, hi = p.shard_range(nblocks, rank, world)
    cs, offs = p.gather_block_index(torch.from_numpy(all_c[lo:hi].astype(np.int32)), nblocks)
    mine = {"rank": rank, "first_block_offset": int(offs[lo]) + (4 if rank == break_rank else 0), "shard_bytes": int((cs[lo:hi] + 12).sum()), "blocks": hi - lo}
    try:
        rep = p.container.gather_rank_reports(mine)
        q.put((rank, "ok", [r["rank"] for r in rep], [r["first_block_offset"] for r in rep]))
    except AssertionError as e:
        q.put((rank, "bad", str(e), None))
    dist.barrier(); dist.destroy_process_group()


def test_rank_reports_world2():
    """what bench.py puts into `per_rank` for N > 1: gathered in rank order on every rank, and an offset that does not follow from the
    shard sizes is refused on every rank"""
    for break_rank in (-1, 1):
        ctx = mp.get_context("spawn"); q = ctx.Queue(); port = _free_port()
        ps = [ctx.Process(target=_report_worker, args=(r, 2, port, 9, break_rank, q)) for r in range(2)]
        [p.start() for p in ps]
        got = sorted(q.get(timeout=120) for _ in range(2))
        [p.join(60) for p in ps]
        assert all(p.exitcode == 0 for p in ps)
        for rank, verdict, a, b in got:
            if break_rank < 0:
                assert verdict == "ok" and a == [0, 1] and b[0] == 12 and b[1] > 12, (rank, verdict, a, b)
            else:
                assert verdict == "bad" and "rank 1" in a, (rank, verdict, a)


# ---- a rank that fails must not leave the others waiting in the collective (4mc_amd/csrc/shard.c) ----------
def _failing_worker(rank, world, port, src, out, q):
    import ctypes as C
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), FOURMC_SHARD_FAIL_RANK="0")
    dist.init_process_group("gloo", rank=rank, world_size=world)
    sys.path.insert(0, helpers.ROOT)
    p = helpers.pkg(); L = p.lib()
    AG = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p)
    seen = []
    def allgather(ctx, send, nbytes, recv):
        a = torch.from_numpy(np.ctypeslib.as_array(C.cast(send, C.POINTER(C.c_uint8)), (nbytes,)).copy())
        o = torch.empty(nbytes * world, dtype=torch.uint8)
        dist.all_gather_into_tensor(o, a)
        C.memmove(recv, o.numpy().ctypes.data, nbytes * world)
        seen.append(nbytes)
        return 0
    cb = AG(allgather)
    L.fourmc_file_compress_sharded.restype = C.c_int
    L.fourmc_file_compress_sharded.argtypes = [C.c_char_p, C.c_char_p, C.c_int, C.c_uint32, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
    rc = L.fourmc_file_compress_sharded(src.encode(), out.encode(), 1, p.MAGIC_4MC, rank, world, C.cast(cb, C.c_void_p), None)
    q.put((rank, rc, seen))
    dist.barrier(); dist.destroy_process_group()


def test_a_failing_rank_ends_the_call_on_every_rank(tmp_path):
    """One block, two ranks: rank 0 owns it and reports an engine failure (FOURMC_SHARD_FAIL_RANK, no device needed); rank 1 owns
    nothing and has nothing to fail on.  Both take part in the ONE all-gather (its rows carry a status word), both come back - rank 0
    with its own code (-3), rank 1 with -6 "another rank failed" - and nothing is written."""
    src = tmp_path / "one_block.bin"; src.write_bytes(helpers.corpus(100000).tobytes())
    out = tmp_path / "never.4mc"
    ctx = mp.get_context("spawn"); q = ctx.Queue(); port = _free_port()
    ps = [ctx.Process(target=_failing_worker, args=(r, 2, port, str(src), str(out), q)) for r in range(2)]
    [p.start() for p in ps]
    got = sorted(q.get(timeout=120) for _ in range(2))           # a hang (round 5: the peers waited in the collective) ends here
    [p.join(60) for p in ps]
    assert all(p.exitcode == 0 for p in ps)
    assert [(g[0], g[1]) for g in got] == [(0, -3), (1, -6)], got
    assert all(g[2] == [(1 + 1) * 4] for g in got), got           # one exchange each: {status, one size}
    assert not out.exists()
