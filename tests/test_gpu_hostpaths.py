"""GPU: host-side paths of the drop-in boundary that round 1 left untested - several devices / ranks behind the file API,
concatenated streams, pipe mode, the overwrite refusal, and the whole-file SHA of `4mc -4`.
Reference behaviour: native/4mc.c:164-209 (open), :220-386 (compress), :560-707 + :908-912 (decode, concatenated streams),
native/4mccli.c:190-271; golden files: tests/golden/corpus_manifest.json (written by the reference CLI)."""
import ctypes as C
import hashlib
import json
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

import helpers
from helpers import B

pytestmark = pytest.mark.gpu
G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


@pytest.fixture(scope="module")
def gpu():
    p = helpers.pkg(); p.gpu_init()
    return p


@pytest.fixture(scope="module")
def golden_corpus(tmp_path_factory):
    man = json.load(open(os.path.join(G, "corpus_manifest.json")))
    data = helpers.corpus(man["corpus"]["bytes"], first_block=man["corpus"]["first_block"], seed=man["corpus"]["seed"])
    assert hashlib.sha256(data.tobytes()).hexdigest() == man["corpus"]["sha256"]
    path = tmp_path_factory.mktemp("corpus") / "corpus.bin"
    path.write_bytes(data.tobytes())
    return man, data, path


def _sha(path):
    return hashlib.sha256(open(path, "rb").read()).hexdigest()


def test_cli_ultra_whole_file_sha(gpu, golden_corpus, tmp_path):
    """`4mc -4` (LZ4 HC level 8): the whole file equals the reference CLI's"""
    man, data, src = golden_corpus
    out = tmp_path / "c.4mc"
    r = subprocess.run([gpu.cli_path(), "-4", "-f", str(src), str(out)], capture_output=True)
    assert r.returncode == 0, r.stderr
    assert _sha(out) == man["levels"]["4mc-4"]["sha256"]


@pytest.mark.parametrize("shards", [2, 3, 5])
def test_file_api_over_several_shards(gpu, golden_corpus, tmp_path, shards):
    """FOURMC_GPUS=N: a batch is cut into N contiguous block ranges, each with its own device buffers and stream (round
    robin over the visible devices; on a one-GPU box they share the device).  Files stay byte-identical."""
    man, data, src = golden_corpus
    env = dict(os.environ, FOURMC_GPUS=str(shards), FOURMC_BATCH_BLOCKS="13")
    for flags, key in (([], "4mc-1"), (["-3"], "4mc-3"), (["-z", "-1"], "4mz-1")):
        out = tmp_path / ("s%d_%s" % (shards, key)); back = tmp_path / "back.bin"
        r = subprocess.run([gpu.cli_path(), *flags, "-f", str(src), str(out)], capture_output=True, env=env)
        assert r.returncode == 0, r.stderr
        assert _sha(out) == man["levels"][key]["sha256"], (shards, key)
        dflags = ["-z"] if key.startswith("4mz") else []
        r = subprocess.run([gpu.cli_path(), *dflags, "-d", "-f", str(out), str(back)], capture_output=True, env=env)
        assert r.returncode == 0, r.stderr
        assert _sha(back) == man["corpus"]["sha256"]


_RANK_SCRIPT = r'''
import ctypes as C, os, sys
import numpy as np, torch, torch.distributed as dist
sys.path.insert(0, sys.argv[1]); sys.path.insert(0, os.path.join(sys.argv[1], "tests"))
import helpers
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
dist.init_process_group("gloo", rank=rank, world_size=world)
p = helpers.pkg(); p.gpu_init(0); L = p.lib()
AG = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p)
def allgather(ctx, send, nbytes, recv):
    a = torch.from_numpy(np.ctypeslib.as_array(C.cast(send, C.POINTER(C.c_uint8)), (nbytes,)).copy())
    out = torch.empty(nbytes * world, dtype=torch.uint8)
    dist.all_gather_into_tensor(out, a)
    C.memmove(recv, out.numpy().ctypes.data, nbytes * world)
    return 0
cb = AG(allgather)
magic = p.MAGIC_4MZ if sys.argv[4] == "z" else p.MAGIC_4MC
rc = L.fourmc_file_compress_sharded(sys.argv[2].encode(), sys.argv[3].encode(), int(sys.argv[5]), magic, rank, world, C.cast(cb, C.c_void_p), None)
dist.barrier(); dist.destroy_process_group()
sys.exit(10 + abs(rc) if rc else 0)
'''


@pytest.mark.parametrize("world", [1, 2, 3])
def test_ranks_write_one_file_through_the_c_host(gpu, golden_corpus, tmp_path, world):
    """fourmc_file_compress_sharded: one process per rank (here all on GPU 0), every rank compresses its block range on the
    GPU, ONE all-gather of the compressed sizes (gloo here, RCCL on a multi-GPU node), every rank pwrite()s its byte range."""
    man, data, src = golden_corpus
    script = tmp_path / "rank.py"; script.write_text(_RANK_SCRIPT)
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    for fmt, level, key in (("c", 1, "4mc-1"), ("z", 1, "4mz-1")):
        out = tmp_path / ("w%d_%s" % (world, key))
        ps = [subprocess.Popen([sys.executable, str(script), helpers.ROOT, str(src), str(out), fmt, str(level)],
                               env=dict(os.environ, RANK=str(r), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port + 0)))
              for r in range(world)]
        assert [p.wait(timeout=600) for p in ps] == [0] * world
        assert os.path.getsize(out) == man["levels"][key]["file_bytes"]
        assert _sha(out) == man["levels"][key]["sha256"], (world, key)


def test_a_rank_whose_encode_fails_ends_the_call_on_every_rank(gpu, golden_corpus, tmp_path):
    """Two ranks (both on GPU 0, gloo) write one .4mz file; rank 1's engine cannot have any workspace (FOURMC_WS_FAIL_ABOVE=1: every
    lease of the zstd encoder fails the way hipMalloc would), so ITS encode fails - after rank 0 has compressed its own range.  Both
    ranks take part in the one all-gather (rows carry a status word) and both return within the timeout: rank 1 with the engine
    error (-3 -> exit 13), rank 0 with "another rank failed" (-6 -> exit 16); round 5's writer left rank 0 waiting in the collective
    forever.  Nothing is written."""
    man, data, src = golden_corpus
    script = tmp_path / "rank.py"; script.write_text(_RANK_SCRIPT)
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    out = tmp_path / "never.4mz"
    ps = [subprocess.Popen([sys.executable, str(script), helpers.ROOT, str(src), str(out), "z", "1"],
                           env=dict(os.environ, RANK=str(r), WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                                    **({"FOURMC_WS_FAIL_ABOVE": "1"} if r == 1 else {})))
          for r in range(2)]
    codes = [p.wait(timeout=300) for p in ps]                       # a hang ends here
    assert codes == [16, 13], codes
    assert not out.exists()


@pytest.mark.parametrize("world", [1, 3])
def test_ranks_decompress_one_file_through_the_c_host(gpu, golden_corpus, tmp_path, world):
    """fourmc_file_decompress_sharded: every rank decodes its block range through the footer index and pwrite()s it at
    block index * 4 MiB - no exchange at all; the owner of the last block sets the file size (an older, longer output must not
    survive).  The ranks run one after the other here (one GPU); they touch disjoint byte ranges."""
    import ctypes as C
    man, data, src = golden_corpus
    L = gpu.lib()
    L.fourmc_file_decompress_sharded.restype = C.c_int
    L.fourmc_file_decompress_sharded.argtypes = [C.c_char_p, C.c_char_p, C.c_int, C.c_int, C.POINTER(C.c_longlong)]
    for flags, key in (([], "4mc-1"), (["-z"], "4mz-1")):
        comp = tmp_path / ("d%d_%s" % (world, key))
        r = subprocess.run([gpu.cli_path(), *flags, "-f", str(src), str(comp)], capture_output=True)
        assert r.returncode == 0, r.stderr
        back = tmp_path / ("d%d_%s.back" % (world, key))
        back.write_bytes(b"\xEE" * (len(data) + 54321))
        os.environ["FOURMC_BATCH_BLOCKS"] = "5"                       # several batches per rank on the small corpus
        try:
            for rank in reversed(range(world)):                       # (any order: the ranges are disjoint)
                det = C.c_longlong(0)
                assert L.fourmc_file_decompress_sharded(str(comp).encode(), str(back).encode(), rank, world, C.byref(det)) == 0, (rank, det.value)
        finally:
            del os.environ["FOURMC_BATCH_BLOCKS"]
        assert os.path.getsize(back) == len(data)
        assert _sha(back) == man["corpus"]["sha256"], (world, key)
        # a damaged block: the rank that owns it reports it (-3, detail -4), the others finish
        blob = bytearray(comp.read_bytes()); blob[len(blob) // 2] ^= 0x20; bad = tmp_path / "bad"; bad.write_bytes(bytes(blob))
        codes = []
        for rank in range(world):
            det = C.c_longlong(0)
            codes.append((L.fourmc_file_decompress_sharded(str(bad).encode(), str(back).encode(), rank, world, C.byref(det)), det.value))
        assert sorted(c[0] for c in codes)[0] == -3 and sum(1 for c in codes if c[0] != 0) == 1, codes


def test_concatenated_streams_decode_as_one(gpu, tmp_path):
    """two .4mc files glued together decode to the concatenation of their contents (native/4mc.c:908-912)"""
    a = helpers.corpus(2 * B + 99, first_block=1); b = helpers.corpus(B // 2 + 5, first_block=9)
    parts = []
    for i, d in enumerate((a, b)):
        src = tmp_path / ("p%d" % i); src.write_bytes(d.tobytes())
        out = tmp_path / ("p%d.4mc" % i)
        assert subprocess.run([gpu.cli_path(), "-f", str(src), str(out)], capture_output=True).returncode == 0
        parts.append(out.read_bytes())
    cat = tmp_path / "cat.4mc"; cat.write_bytes(parts[0] + parts[1])
    back = tmp_path / "cat.bin"
    r = subprocess.run([gpu.cli_path(), "-d", "-f", str(cat), str(back)], capture_output=True)
    assert r.returncode == 0, r.stderr
    assert back.read_bytes() == a.tobytes() + b.tobytes()
    ref = helpers.ref_cli()
    if ref:                                                         # the reference CLI reads the same glued file the same way
        rb = tmp_path / "ref.bin"
        assert subprocess.run([ref, "-d", "-f", str(cat), str(rb)], capture_output=True).returncode == 0
        assert rb.read_bytes() == back.read_bytes()


def test_pipe_mode_and_overwrite_refusal(gpu, tmp_path):
    """`4mc -c` / stdin-stdout pipes (native/4mccli.c:215,:262-271) and exit code 3 when the output exists and stdin offers
    no confirmation (native/4mc.c:190-203)"""
    data = helpers.corpus(B + 4321, first_block=2)
    want = helpers.orc_container(data).tobytes()
    r = subprocess.run([gpu.cli_path(), "-c"], input=data.tobytes(), capture_output=True)             # stdin -> stdout
    assert r.returncode == 0, r.stderr
    assert r.stdout == want
    r = subprocess.run([gpu.cli_path(), "-d", "-c"], input=want, capture_output=True)
    assert r.returncode == 0, r.stderr
    assert r.stdout == data.tobytes()
    src = tmp_path / "x"; src.write_bytes(data.tobytes())
    out = tmp_path / "x.4mc"; out.write_bytes(b"already here")
    r = subprocess.run([gpu.cli_path(), "-q", str(src), str(out)], input=b"", capture_output=True)    # quiet: no prompt, refuse
    assert r.returncode == 3
    assert out.read_bytes() == b"already here"
    r = subprocess.run([gpu.cli_path(), str(src), str(out)], input=b"n\n", capture_output=True)       # prompt answered with no
    assert r.returncode == 3 and b"already exists" in r.stderr
    r = subprocess.run([gpu.cli_path(), str(src), str(out)], input=b"y\n", capture_output=True)       # ... and with yes
    assert r.returncode == 0 and out.read_bytes() == want


@pytest.mark.parametrize("mapped", ["1", "0"], ids=["mapped", "streaming"])
def test_cli_mapped_and_streaming_paths_write_the_same_files(gpu, golden_corpus, tmp_path, mapped):
    """regular files take the mapped path (input and output files mapped, 512-block launches, no stdio in between),
    FOURMC_MMAP=0 and pipes the streaming one: same files, same messages, same exit codes, same bytes left behind by a
    file that is corrupt half way through (what the reference CLI leaves: native/4mc.c:637-668)."""
    man, data, src = golden_corpus
    env = dict(os.environ, FOURMC_MMAP=mapped)
    for flags, key in (([], "4mc-1"), (["-2"], "4mc-2"), (["-z", "-1"], "4mz-1")):
        out = tmp_path / ("m%s_%s" % (mapped, key)); back = tmp_path / "back.bin"
        r = subprocess.run([gpu.cli_path(), *flags, "-f", str(src), str(out)], capture_output=True, env=env)
        assert r.returncode == 0, r.stderr
        assert b"Compressed (" in r.stderr
        assert _sha(out) == man["levels"][key]["sha256"], (mapped, key)
        dflags = ["-z"] if key.startswith("4mz") else []
        r = subprocess.run([gpu.cli_path(), *dflags, "-d", "-f", str(out), str(back)], capture_output=True, env=env)
        assert r.returncode == 0, r.stderr
        assert b"Successfully decoded %d bytes" % len(data) in r.stderr
        assert _sha(back) == man["corpus"]["sha256"]
        if key == "4mc-1":
            # a payload byte of a block in the middle flipped: exit 4, the blocks before it are in the output
            img = bytearray(out.read_bytes())
            img[len(img) // 2] ^= 0x40
            bad = tmp_path / "bad.4mc"; bad.write_bytes(bytes(img)); part = tmp_path / ("part%s.bin" % mapped)
            r = subprocess.run([gpu.cli_path(), "-d", "-f", str(bad), str(part)], capture_output=True, env=env)
            assert r.returncode == 4 and b"invalid block checksum detected" in r.stderr, r.stderr
            got = part.read_bytes()
            assert 0 < len(got) < len(data) and len(got) % helpers.B == 0 and got == data[: len(got)].tobytes()
            ref = helpers.ref_cli()
            if ref:
                rp = tmp_path / "part_ref.bin"
                rr = subprocess.run([ref, "-d", "-f", str(bad), str(rp)], capture_output=True)
                assert rr.returncode == 4 and rp.read_bytes() == got


def test_c_launcher_gathers_the_index_through_rccl(gpu, golden_corpus, tmp_path):
    """tools/shard_rccl: the C host of the several-ranks writer with ncclAllGather (librccl.so, loaded at run time) as its one
    collective.  RCCL refuses two ranks on one device, so a one-GPU box runs WORLD_SIZE=1: communicator, stream and call are
    the real ones, the file equals the reference CLI's (more ranks: the same code, byte layout in test_multirank_cpu.py)."""
    exe = os.path.join(helpers.ROOT, "tools", "shard_rccl")
    if not os.path.exists(exe):
        pytest.skip("tools/shard_rccl not built (__graft_entry__.build())")
    man, data, src = golden_corpus
    env = dict(os.environ, RANK="0", WORLD_SIZE="1", LOCAL_RANK="0", FOURMC_BATCH_BLOCKS="5")
    for flags, key in (([], "4mc-1"), (["-z", "-1"], "4mz-1")):
        out = tmp_path / ("rccl_" + key)
        out.write_bytes(b"\xEE" * (man["levels"][key]["file_bytes"] + 999))       # an older, longer file of that name
        r = subprocess.run([exe, *flags, str(src), str(out)], capture_output=True, env=env, timeout=300)
        assert r.returncode == 0, r.stderr
        assert _sha(out) == man["levels"][key]["sha256"], key
        # and back, two launcher processes on the one GPU (the decompress direction has no collective: ranks need not meet)
        back = tmp_path / ("rccl_" + key + ".back")
        back.write_bytes(b"\xEE" * (len(data) + 777))
        ps = [subprocess.Popen([exe, "-d", str(out), str(back)], env=dict(env, RANK=str(rk), WORLD_SIZE="2", LOCAL_RANK="0"), stderr=subprocess.PIPE) for rk in range(2)]
        assert [p.wait(timeout=300) for p in ps] == [0, 0], [p.stderr.read() for p in ps]
        assert _sha(back) == man["corpus"]["sha256"], key


@pytest.mark.parametrize("flags,limit", [(["-z", "-1"], 6 << 20), (["-z", "-4"], 120 << 20), (["-3"], 1 << 20)])
def test_workspace_that_does_not_fit_is_taken_in_pieces(gpu, tmp_path, flags, limit):
    """A launch whose workspace the device cannot give is cut into pieces instead of failing (engine.hip: in_pieces).
    FOURMC_WS_FAIL_ABOVE makes every lease above `limit` bytes fail the way hipMalloc would: the files stay the reference's."""
    data = helpers.corpus(9 * B + 999, first_block=2)
    src = tmp_path / "in.bin"; src.write_bytes(data.tobytes())
    outs = []
    for env in (os.environ, dict(os.environ, FOURMC_WS_FAIL_ABOVE=str(limit))):
        out = tmp_path / f"o{len(outs)}"
        r = subprocess.run([gpu.cli_path()] + flags + ["-f", str(src), str(out)], capture_output=True, env=env)
        assert r.returncode == 0, r.stderr
        outs.append(out.read_bytes())
    assert outs[0] == outs[1]
    back = tmp_path / "back"
    env = dict(os.environ, FOURMC_WS_FAIL_ABOVE=str(20 << 20))          # the decoders' leases: 14 MiB (4mz) / 11 MiB (4mc) per block
    r = subprocess.run([gpu.cli_path(), "-d"] + (["-z"] if "-z" in flags else []) + ["-f", str(tmp_path / "o1"), str(back)], capture_output=True, env=env)
    assert r.returncode == 0, r.stderr
    assert back.read_bytes() == data.tobytes()


@pytest.mark.parametrize("config", ["fast", "ultra_logs"])
def test_bench_line_of_two_ranks(gpu, config):
    """bench.py as the driver launches it for N > 1 (torch.distributed.run, one process per rank), with two ranks: block ranges per
    rank, the footer index gathered, every rank's numbers in rank 0's line (`per_rank`), the gathered index checked against a
    recomputed prefix sum.  RCCL refuses two ranks on one device, so on a one-GPU box the collectives go through gloo
    (FOURMC_BENCH_BACKEND, a test aid: same calls, host copies); a reduced launch (256 / 64 blocks per rank)."""
    import json, socket, sys
    with socket.socket() as so:
        so.bind(("127.0.0.1", 0)); port = so.getsockname()[1]
    nb = 256 if config == "fast" else 64
    env = dict(os.environ, FOURMC_BENCH_BACKEND="gloo", FOURMC_BENCH_NO_PMC="1")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                        "--master-port", str(port), os.path.join(helpers.ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1",
                        "--blocks", str(nb), "--no-cpu", "--no-extras", "--config", config], capture_output=True, text=True, env=env, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith('{"metric"')]
    assert len(lines) == 1, r.stdout[-2000:]                      # rank 0 prints, the others do not
    j = json.loads(lines[0])
    assert j["n_gpus"] == 2 and j["scaling"] == "weak" and j["steps"] == 2
    pr = j["per_rank"]
    assert [g["rank"] for g in pr] == [0, 1] and all(g["blocks"] == pr[0]["blocks"] for g in pr)
    assert pr[0]["first_block_offset"] == 12 and pr[1]["first_block_offset"] == 12 + pr[0]["shard_bytes"]
    assert j["rccl_ranks"]["world_size"] == 2
    total = sum(g["blocks"] for g in pr) * helpers.B
    assert abs(j["value"] - total / (j["ms_per_step"] * 1e-3) / 1e9) < 0.02 * j["value"]      # whole-job aggregate over both ranks
