"""GPU: LZ4 HC (4mc High = level 4, Ultra = level 8) byte parity against the oracle port
(oracle/lz4hc_port.c, itself pinned to the reference's LZ4_compress_HC) and the reference CLI's
golden manifests."""
import hashlib
import json
import os
import subprocess
import time

import numpy as np
import pytest
import torch

import helpers
from helpers import B

pytestmark = pytest.mark.gpu
G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _encode_hc(gpu, srcs, caps, level):
    offs, pos = [], 0
    for s in srcs:
        offs.append(pos); pos += len(s) + 3
    buf = np.zeros(pos + 64, np.uint8)
    for s, o in zip(srcs, offs):
        buf[o:o + len(s)] = s
    dsts, dpos = [], 0
    for c in caps:
        dsts.append(dpos); dpos += c + 40
    batch = gpu.DeviceBatch(gpu.make_blocks(offs, dsts, [len(s) for s in srcs], caps))
    d_out = torch.full((dpos + 64,), 0x5A, dtype=torch.uint8, device="cuda")
    gpu.lz4_compress_hc(torch.from_numpy(buf).cuda(), d_out, batch, level)
    torch.cuda.synchronize()
    res = batch.download()["result"]
    out = d_out.cpu().numpy()
    return res, [out[d:d + max(r, 0)] for d, r in zip(dsts, res)]


@pytest.mark.parametrize("level", [4, 8])
def test_lz4hc_bytes_identical_edge_inputs(gpu, level):
    inputs = helpers.edge_inputs()
    names = list(inputs)
    srcs = [inputs[k] for k in names]
    bound = [helpers.oracle().orc_lz4_compress_bound(len(s)) for s in srcs]
    for mode, caps in (("bound", bound), ("n-1", [max(len(s) - 1, 0) for s in srcs])):
        res, outs = _encode_hc(gpu, srcs, caps, level)
        for k, s, cap, r, o in zip(names, srcs, caps, res, outs):
            want_r, want = helpers.orc_compress_hc(s, level, cap)
            assert r == want_r, (mode, k, r, want_r)
            assert np.array_equal(o, want), (mode, k)
            if r > 0:                                     # and it is a valid LZ4 block
                n, back = helpers.orc_decompress(o, len(s))
                assert n == len(s) and np.array_equal(back, s), (mode, k)


def test_lz4hc4_corpus_blocks_and_golden_manifest(gpu):
    m = json.load(open(os.path.join(G, "corpus_manifest.json")))
    n = m["corpus"]["bytes"]
    data = helpers.corpus(n)
    nb = -(-n // B)
    srcs = [data[b * B: min(n, (b + 1) * B)] for b in range(nb)]
    t0 = time.time()
    res, outs = _encode_hc(gpu, srcs, [len(s) - 1 for s in srcs], 4)
    print("hc4 13 blocks:", time.time() - t0, "s")
    for b, (u, c, x) in enumerate(m["levels"]["4mc-3"]["blocks"]):      # `4mc -3` = LZ4 HC level 4
        payload = outs[b] if res[b] > 0 else srcs[b]
        assert (len(srcs[b]), len(payload), helpers.orc_xxh32(payload)) == (u, c, x), b
    # level 8 on a sample of blocks (128 chain steps per search)
    sample = [0, 3, 5, 12]
    res8, outs8 = _encode_hc(gpu, [srcs[b] for b in sample], [len(srcs[b]) - 1 for b in sample], 8)
    for i, b in enumerate(sample):
        u, c, x = m["levels"]["4mc-4"]["blocks"][b]                       # `4mc -4` = LZ4 HC level 8
        payload = outs8[i] if res8[i] > 0 else srcs[b]
        assert (len(payload), helpers.orc_xxh32(payload)) == (c, x), b


def test_cli_high_level_file_equals_reference(gpu, tmp_path):
    m = json.load(open(os.path.join(G, "corpus_manifest.json")))
    data = helpers.corpus(m["corpus"]["bytes"])
    src = tmp_path / "c.bin"; src.write_bytes(data.tobytes())
    out = tmp_path / "c.4mc"
    r = subprocess.run([gpu.cli_path(), "-3", "-f", str(src), str(out)], capture_output=True)
    assert r.returncode == 0, r.stderr
    assert b"Compressed (high)" in r.stderr
    img = out.read_bytes()
    assert len(img) == m["levels"]["4mc-3"]["file_bytes"]
    assert hashlib.sha256(img).hexdigest() == m["levels"]["4mc-3"]["sha256"], "file differs from the reference CLI's"
    back = tmp_path / "back.bin"
    assert subprocess.run([gpu.cli_path(), "-d", "-f", str(out), str(back)], capture_output=True).returncode == 0
    assert back.read_bytes() == data.tobytes()


# ------------------------------------------------------------------------------------------ 4mc Medium (lz4mc.c)
def _encode_mc(gpu, srcs, caps):
    offs, pos = [], 1
    for s in srcs:
        offs.append(pos); pos += len(s) + 3
    buf = np.zeros(pos + 64, np.uint8)
    for s, o in zip(srcs, offs):
        buf[o:o + len(s)] = s
    dsts, dpos = [], 0
    for s in srcs:
        dsts.append(dpos); dpos += len(s) + len(s) // 255 + 64
    capf = [0xFFFFFFFF if c < 0 else c for c in caps]
    batch = gpu.DeviceBatch(gpu.make_blocks(offs, dsts, [len(s) for s in srcs], capf))
    d_out = torch.full((dpos + 64,), 0x5A, dtype=torch.uint8, device="cuda")
    gpu.lz4_compress_mc(torch.from_numpy(buf).cuda(), d_out, batch)
    torch.cuda.synchronize()
    res = batch.download()["result"]
    out = d_out.cpu().numpy()
    return res, [out[d:d + max(r, 0)] for d, r in zip(dsts, res)]


def test_lz4mc_bytes_identical(gpu):
    inputs = helpers.edge_inputs()
    names = list(inputs)
    srcs = [inputs[k] for k in names]
    for mode, caps in (("unlimited", [-1] * len(srcs)), ("n-1", [max(len(s) - 1, 0) for s in srcs])):
        res, outs = _encode_mc(gpu, srcs, caps)
        for k, s, cap, r, o in zip(names, srcs, caps, res, outs):
            want_r, want = helpers.orc_compress_mc(s, cap)
            assert r == want_r, (mode, k, r, want_r)
            assert np.array_equal(o, want), (mode, k)
            if r > 0 and len(s) > 0:
                n, back = helpers.orc_decompress(o, len(s))
                assert n == len(s) and np.array_equal(back, s), (mode, k)
    # corpus blocks against the reference CLI's `4mc -2` manifest
    m = json.load(open(os.path.join(G, "corpus_manifest.json")))
    n = m["corpus"]["bytes"]
    data = helpers.corpus(n)
    nb = -(-n // B)
    blocks = [data[b * B: min(n, (b + 1) * B)] for b in range(nb)]
    t0 = time.time()
    res, outs = _encode_mc(gpu, blocks, [len(s) - 1 for s in blocks])
    print("mc 13 blocks:", time.time() - t0, "s")
    for b, (u, c, x) in enumerate(m["levels"]["4mc-2"]["blocks"]):
        payload = outs[b] if res[b] > 0 else blocks[b]
        assert (len(payload), helpers.orc_xxh32(payload)) == (c, x), b


def test_cli_medium_file_equals_reference(gpu, tmp_path):
    m = json.load(open(os.path.join(G, "corpus_manifest.json")))
    data = helpers.corpus(m["corpus"]["bytes"])
    src = tmp_path / "c.bin"; src.write_bytes(data.tobytes())
    out = tmp_path / "c.4mc"
    r = subprocess.run([gpu.cli_path(), "-2", "-f", str(src), str(out)], capture_output=True)
    assert r.returncode == 0 and b"Compressed (medium)" in r.stderr, r.stderr
    assert hashlib.sha256(out.read_bytes()).hexdigest() == m["levels"]["4mc-2"]["sha256"]
