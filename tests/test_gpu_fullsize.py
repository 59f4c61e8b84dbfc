"""GPU, BASELINE.json configs[1] size (2048 x 4 MiB = 8 GiB HBM resident): size-independent
properties — replicas encode identically, every payload equals the oracle's for its base block,
decode(encode(x)) == x over the whole corpus, one corrupted byte flags exactly one block."""
import numpy as np
import pytest
import torch

import helpers
from helpers import B

pytestmark = pytest.mark.gpu


def test_8gib_roundtrip_and_replica_consistency(gpu):
    base_n, nb = 48, 2048
    base = helpers.corpus(base_n * B)
    d_src = torch.from_numpy(base).cuda().repeat(-(-nb // base_n))[: nb * B].contiguous()
    offs = np.arange(nb, dtype=np.uint64) * B
    lens = np.full(nb, B, dtype=np.uint32)
    enc = gpu.DeviceBatch(gpu.make_blocks(offs, offs, lens, lens))
    d_stage = torch.empty(nb * B, dtype=torch.uint8, device="cuda")
    gpu.encode_blocks(d_src, d_stage, enc)
    e = enc.download()
    # replicas of one base block give identical (csize, xxh32): idempotence across the batch
    for b in range(nb):
        assert (e["result"][b], e["xxh32"][b]) == (e["result"][b % base_n], e["xxh32"][b % base_n]), b
    # the 48 distinct blocks equal the oracle's bytes (checked through size + checksum + a full compare of 6)
    stage = d_stage[: base_n * B].cpu().numpy()
    for b in range(base_n):
        r, comp = helpers.orc_compress(base[b * B:(b + 1) * B], B - 1)
        want = comp if r > 0 else base[b * B:(b + 1) * B]
        assert e["result"][b] == len(want) and e["xxh32"][b] == helpers.orc_xxh32(want), b
        if b % 8 == 0:
            assert np.array_equal(stage[b * B: b * B + len(want)], want), b
    # pack into one image, decode in place, compare all 8 GiB
    csz = torch.from_numpy(e["result"].astype(np.int64)).cuda()
    img_off = (torch.cumsum(csz + 12, 0) - (csz + 12) + 12).contiguous()
    d_img = torch.zeros(int((csz + 12).sum().item()) + 4096, dtype=torch.uint8, device="cuda")
    gpu.pack_image(d_stage, d_img, enc, img_off)
    assert list(gpu.container.block_offsets(e["result"])) == img_off.cpu().tolist()
    dblocks = gpu.make_blocks(img_off.cpu().numpy().astype(np.uint64) + 12, offs, e["result"].astype(np.uint32), lens, e["xxh32"])
    dec = gpu.DeviceBatch(dblocks)
    d_out = torch.empty(nb * B + 64, dtype=torch.uint8, device="cuda")
    gpu.decode_blocks(d_img, d_out, dec)
    assert bool((torch.from_numpy(dec.download()["result"].astype(np.int64)) == B).all())
    assert torch.equal(d_out[: nb * B], d_src)
    # block headers inside the image are the reference's big-endian triples
    hdr = d_img[int(img_off[5]): int(img_off[5]) + 12].cpu().numpy()
    assert [int.from_bytes(bytes(hdr[i:i + 4]), "big") for i in (0, 4, 8)] == [B, int(e["result"][5]), int(e["xxh32"][5])]
    # one flipped byte -> exactly one BADSUM
    victim = 1234
    pos = int(img_off[victim]) + 12 + 77
    d_img[pos] ^= 0x40
    dec = gpu.DeviceBatch(dblocks)
    gpu.decode_blocks(d_img, d_out, dec)
    r = dec.download()["result"]
    assert r[victim] == gpu.BLK_BADSUM and int((r != B).sum()) == 1


@pytest.mark.parametrize("level", [1, 3])
def test_zstd_full_launch_equals_the_oracle_on_every_replica(gpu, level):
    """4mz Fast / Medium at the bench's size: 2048 blocks in one launch (every CU holds its 8 blocks, the dense-window finders'
    table reads and writes of all of them in flight together); every replica of a base block gives the same frame, and the
    48 distinct frames are the oracle's (zstd_enc_port.c, pinned to the reference's ZSTD_compress)."""
    base_n, nb = 48, 2048
    base = helpers.corpus(base_n * B)
    d_src = torch.from_numpy(base).cuda().repeat(-(-nb // base_n))[: nb * B].contiguous()
    offs = np.arange(nb, dtype=np.uint64) * B
    lens = np.full(nb, B, dtype=np.uint32)
    enc = gpu.DeviceBatch(gpu.make_blocks(offs, offs, lens, lens))
    d_stage = torch.empty(nb * B, dtype=torch.uint8, device="cuda")
    gpu.encode_blocks(d_src, d_stage, enc, codec=gpu.CODEC_ZSTD, level=level)
    e = enc.download()
    for b in range(nb):
        assert (e["result"][b], e["xxh32"][b]) == (e["result"][b % base_n], e["xxh32"][b % base_n]), b
    stage = d_stage[: base_n * B].cpu().numpy()
    for b in range(base_n):
        src = base[b * B:(b + 1) * B]
        r, comp = helpers.orc_zstd_compress(src, level, B - 1)
        want = comp if r > 0 else src
        assert e["result"][b] == len(want) and e["xxh32"][b] == helpers.orc_xxh32(want), b
        if b % 6 == 0:
            assert np.array_equal(stage[b * B: b * B + len(want)], want), b
    # a late replica, byte for byte
    late = 2040
    want_r = int(e["result"][late % base_n])
    assert torch.equal(d_stage[late * B: late * B + want_r], d_stage[(late % base_n) * B: (late % base_n) * B + want_r])


def _full_launch_against_the_oracle(gpu, base, base_n, nb, codec, level, oracle_fn, sample_every):
    """`nb` blocks (replicas of `base_n` distinct ones) in ONE launch: every replica equals its base block's result, the base
    blocks equal the oracle's (size + XXH32 for all, bytes for a sample), and the batch decodes back to the input."""
    d_src = torch.from_numpy(base).cuda().repeat(-(-nb // base_n))[: nb * B].contiguous()
    offs = np.arange(nb, dtype=np.uint64) * B
    lens = np.full(nb, B, dtype=np.uint32)
    enc = gpu.DeviceBatch(gpu.make_blocks(offs, offs, lens, lens))
    d_stage = torch.empty(nb * B, dtype=torch.uint8, device="cuda")
    gpu.encode_blocks(d_src, d_stage, enc, codec=codec, level=level)
    e = enc.download()
    for b in range(nb):
        assert (e["result"][b], e["xxh32"][b]) == (e["result"][b % base_n], e["xxh32"][b % base_n]), b
    stage = d_stage[: base_n * B].cpu().numpy()
    for b in range(base_n):
        src = base[b * B:(b + 1) * B]
        r, comp = oracle_fn(src)
        want = comp if 0 < r < B else src
        assert e["result"][b] == len(want) and e["xxh32"][b] == helpers.orc_xxh32(want), b
        if b % sample_every == 0:
            assert np.array_equal(stage[b * B: b * B + len(want)], want), b
    late = nb - 8
    want_r = int(e["result"][late % base_n])
    assert torch.equal(d_stage[late * B: late * B + want_r], d_stage[(late % base_n) * B: (late % base_n) * B + want_r])
    dec = gpu.DeviceBatch(gpu.make_blocks(offs, offs, e["result"].astype(np.uint32), lens, e["xxh32"]))
    d_out = torch.empty(nb * B + 64, dtype=torch.uint8, device="cuda")
    gpu.decode_blocks(d_stage, d_out, dec, codec=codec)
    assert bool((torch.from_numpy(dec.download()["result"].astype(np.int64)) == B).all())
    assert torch.equal(d_out[: nb * B], d_src)


def test_hc4_full_launch_equals_the_oracle_on_every_replica(gpu):
    """BASELINE configs[3] (4mc High, LZ4 HC level 4) at the bench's size: 2048 blocks in one launch (builder + parser waves of
    eight blocks per CU, 768 MiB of hash / chain tables in flight)."""
    base = helpers.corpus(48 * B)
    _full_launch_against_the_oracle(gpu, base, 48, 2048, gpu.CODEC_LZ4_HC, 4, lambda s: helpers.orc_compress_hc(s, 4, B - 1), 8)


def test_zstd12_logs_full_launch_equals_the_oracle_on_every_replica(gpu):
    """BASELINE configs[4]'s codec and corpus (4mz Ultra, zstd level 12, synthetic logs) in one launch as large as the level's 49 MiB
    of tables per block allow (2048 blocks = 97 GiB on a 288 GB part)."""
    nb = 2048
    free = torch.cuda.mem_get_info()[0]
    while nb > 256 and nb * (49 + 14) * (1 << 20) > 0.8 * free:
        nb //= 2
    logs = helpers.corpus(24 * B, logs=True)
    _full_launch_against_the_oracle(gpu, logs, 24, nb, gpu.CODEC_ZSTD, 12, lambda s: helpers.orc_zstd_compress(s, 12, B - 1), 6)


@pytest.mark.parametrize("path", [2, 11, 13], ids=["exact", "seg", "tile"])
def test_both_lz4_decode_paths_at_full_size(gpu, path):
    """2048 blocks in one launch through each fast path (the default picks by launch size): all 8 GiB equal the input"""
    base_n, nb = 48, 2048
    base = helpers.corpus(base_n * B)
    d_src = torch.from_numpy(base).cuda().repeat(-(-nb // base_n))[: nb * B].contiguous()
    offs = np.arange(nb, dtype=np.uint64) * B
    lens = np.full(nb, B, dtype=np.uint32)
    enc = gpu.DeviceBatch(gpu.make_blocks(offs, offs, lens, lens))
    d_stage = torch.empty(nb * B, dtype=torch.uint8, device="cuda")
    gpu.encode_blocks(d_src, d_stage, enc)
    e = enc.download()
    before = gpu.lib().fourmc_gpu_get_lz4_decode_path()
    gpu.lib().fourmc_gpu_set_lz4_decode_path(path)
    try:
        dec = gpu.DeviceBatch(gpu.make_blocks(offs, offs, e["result"].astype(np.uint32), lens, e["xxh32"]))
        d_out = torch.zeros(nb * B + 64, dtype=torch.uint8, device="cuda")
        gpu.decode_blocks(d_stage, d_out, dec)
        assert bool((torch.from_numpy(dec.download()["result"].astype(np.int64)) == B).all())
        assert torch.equal(d_out[: nb * B], d_src)
    finally:
        gpu.lib().fourmc_gpu_set_lz4_decode_path(before)


@pytest.mark.parametrize("path", [6, 13], ids=["auto", "tile"])
def test_64gib_launch(gpu, path):
    """16 384 blocks in ONE decode call (the 64 GiB configuration of north_star): through `auto` - the segment-parallel path in
    pieces of 8192 blocks, the split of the launch loop in lz4_decode.hip - and through the tile path (one piece); every one of
    the 64 GiB compared with the input.  Needs ~110 GB of HBM (skipped on a smaller device)."""
    base_n, nb, reps = 48, 2048, 8
    nd = nb * reps
    gpu.release_workspaces(); torch.cuda.empty_cache()             # what earlier tests left with the engine's streams and in torch's cache
    free = torch.cuda.mem_get_info()[0]
    need = (226 if path == 6 else 150) * (1 << 30)                 # 64 GiB out, 8 images of 8 GiB, 8 GiB of input; + the workspace: 92 GB of records / 8.6 GB of bitmaps
                                                                   # (less if the engine's stream already holds a workspace from an earlier test)
    if free < need:
        pytest.skip("needs %d GiB of free HBM" % (need >> 30))
    base = helpers.corpus(base_n * B)
    d_src = torch.from_numpy(base).cuda().repeat(-(-nb // base_n))[: nb * B].contiguous()
    offs = np.arange(nb, dtype=np.uint64) * B
    lens = np.full(nb, B, dtype=np.uint32)
    enc = gpu.DeviceBatch(gpu.make_blocks(offs, offs, lens, lens))
    d_stage = torch.empty(nb * B, dtype=torch.uint8, device="cuda")
    gpu.encode_blocks(d_src, d_stage, enc)
    e = enc.download()
    img = d_stage.repeat(reps)                                   # 8 physically distinct copies of the payload slots
    del d_stage
    so = np.tile(offs, reps) + (np.arange(nd, dtype=np.uint64) // nb) * np.uint64(nb * B)
    before = gpu.lib().fourmc_gpu_get_lz4_decode_path()
    gpu.lib().fourmc_gpu_set_lz4_decode_path(path)
    try:
        dec = gpu.DeviceBatch(gpu.make_blocks(so, np.arange(nd, dtype=np.uint64) * B, np.tile(e["result"].astype(np.uint32), reps),
                                              np.full(nd, B, dtype=np.uint32), np.tile(e["xxh32"], reps)))
        big = torch.zeros(nd * B + 64, dtype=torch.uint8, device="cuda")
        gpu.decode_blocks(img, big, dec)
        assert bool((torch.from_numpy(dec.download()["result"].astype(np.int64)) == B).all())
        for k in range(reps):
            assert torch.equal(big[k * nb * B:(k + 1) * nb * B], d_src), k
        assert bool((big[nd * B:] == 0).all())
        if path == 6:
            # the engine keeps the 92 GB of records for the next call of this size; a caller that wants them back says so, and the next
            # decode allocates again and gives the same bytes
            held = torch.cuda.mem_get_info()[0]
            gpu.release_workspaces()
            assert torch.cuda.mem_get_info()[0] - held > (60 << 30)
            small = gpu.DeviceBatch(gpu.make_blocks(so[:nb], np.arange(nb, dtype=np.uint64) * B, e["result"].astype(np.uint32), lens, e["xxh32"]))
            big[: nb * B].zero_()
            gpu.decode_blocks(img, big, small)
            assert torch.equal(big[: nb * B], d_src)
    finally:
        gpu.lib().fourmc_gpu_set_lz4_decode_path(before)
        gpu.release_workspaces()
