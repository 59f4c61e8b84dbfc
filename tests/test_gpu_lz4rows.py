"""Row-parallel LZ4 decode (4mc_amd/csrc/lz4_rows.hip, the default fast path): decoded bytes and return codes against
the oracle on the shapes that stress it - long literal runs (rows the walk skips, literal runs carried across rows),
long and overlapping matches (match-space passes), unaligned outputs, every block class in one launch - and, with
the exact walker switched off (path 5), that regular streams are finished by the row pipeline itself.
Reference behaviour: native/lz4/lz4.c:1936-2339 via oracle/lz4_port.c."""
import numpy as np
import pytest

import helpers
from helpers import B, corpus, orc_compress, orc_decompress, pkg
import test_gpu_lz4par as par

torch = pytest.importorskip("torch")
pytestmark = pytest.mark.gpu
RETRY = -1000000003


@pytest.fixture(scope="module", params=[4, 5, 7, 8, 9, 10, 11, 12, 13, 14, 15, 16],
                ids=["rows", "rows-alone", "lanes", "lanes-alone", "wx", "wx-alone", "seg", "seg-alone", "tile", "tile-alone", "ring", "ring-alone"])
def gpu(request):
    p = pkg(); p.gpu_init()
    research = request.param in (4, 5, 7, 8, 9, 10, 15, 16)   # designs no launch of the product selects: tools/research/, research side build
    if research: p.use_research(True); p.gpu_init()
    before = p.lib().fourmc_gpu_get_lz4_decode_path()
    p.lib().fourmc_gpu_set_lz4_decode_path(request.param)
    assert p.lib().fourmc_gpu_get_lz4_decode_path() == request.param
    p.rows_alone = request.param in (5, 8, 10, 12, 14, 16)
    # below these sizes a path leaves the block to the exact walker (seg: lz4seg.h kMinSrc / kMinCap; "alone" = blocks handed
    # back entirely stay RETRY - the exact walker still finishes the last bytes of the blocks the segment path executed)
    p.min_cap, p.min_src = (256, 256) if request.param in (12, 14, 16) else (64, 8)
    yield p
    p.lib().fourmc_gpu_set_lz4_decode_path(before)
    if research: p.use_research(False)


def test_decode_shapes(gpu):
    ins = par._inputs()
    names = list(ins)
    comps, caps, shifts = [], [], []
    for i, k in enumerate(names):
        r, comp = orc_compress(ins[k])
        comps.append(comp); caps.append(len(ins[k])); shifts.append((i * 29) % 128)
    res, out, doffs = par._decode(gpu, comps, caps, shifts)
    for i, k in enumerate(names):
        if gpu.rows_alone and (caps[i] < gpu.min_cap or len(comps[i]) < gpu.min_src):
            assert res[i] == RETRY, (k, int(res[i]))                 # below the pipeline's sizes: the exact walker's
            continue
        assert res[i] == caps[i], (k, int(res[i]))
        got = out[doffs[i]: doffs[i] + caps[i]]
        bad = np.nonzero(got != ins[k])[0]
        assert len(bad) == 0, (k, int(bad[0]), len(bad))
        assert doffs[i] == 0 or out[doffs[i] - 1] == 0xA5, k
        assert np.all(out[doffs[i] + caps[i]: doffs[i] + caps[i] + 7] == 0xA5), k


def test_decode_larger_capacity_and_hc_streams(gpu):
    par.test_decode_larger_capacity_and_hc_streams(gpu)


def test_decode_many_blocks_one_launch(gpu):
    par.test_decode_many_blocks_one_launch(gpu)


def test_sizes_around_the_row_and_tail_limits(gpu):
    """stream lengths around multiples of 64 and around the tail guard, capacities at and above the decoded size"""
    text = corpus(B)
    comps, caps, want = [], [], []
    for n in list(range(64, 64 + 40)) + [400, 417, 418, 480, 481, 511, 512, 513, 1000, 4095, 4096, 4097, 65535, 65536, 65537, 70001]:
        src = text[1000: 1000 + n]
        r, comp = orc_compress(src)
        comps.append(comp); caps.append(n + (n % 3) * 5); want.append(src)
    res, out, doffs = par._decode(gpu, comps, caps)
    for i in range(len(comps)):
        want_r, w = orc_decompress(comps[i], caps[i])
        if gpu.rows_alone and res[i] == RETRY:
            continue                                                  # handing back is always allowed; wrong answers are not
        assert res[i] == want_r == len(want[i]), (i, int(res[i]), want_r)
        assert np.array_equal(out[doffs[i]: doffs[i] + want_r], want[i]), i


def test_decode_mutated_streams_match_oracle(gpu):
    """corrupt streams: accept / reject, return codes and accepted bytes equal the oracle's; alone, the row pipeline may
    hand a stream back but never returns a wrong size or wrong bytes for one it claims"""
    rng = np.random.default_rng(5)
    text = corpus(B)[: 200000]
    r, good = orc_compress(text)
    comps, caps = [], []
    for i in range(160):
        c = good.copy()
        for _ in range(int(rng.integers(1, 4))):
            c[int(rng.integers(0, len(c)))] = rng.integers(0, 256)
        if i % 5 == 0:
            c = c[: int(rng.integers(1, len(c)))]
        comps.append(c); caps.append(len(text) + (0 if i % 3 else 77))
    res, out, doffs = par._decode(gpu, comps, caps)
    for i, c in enumerate(comps):
        want_r, want = orc_decompress(c, caps[i])
        if gpu.rows_alone and res[i] == RETRY:
            continue
        assert res[i] == want_r, (i, int(res[i]), want_r)
        if want_r > 0:
            assert np.array_equal(out[doffs[i]: doffs[i] + want_r], want), i
        assert doffs[i] == 0 or out[doffs[i] - 1] == 0xA5, i
        assert np.all(out[doffs[i] + caps[i]: doffs[i] + caps[i] + 7] == 0xA5), i


def test_incompressible_full_blocks_raw_mode(gpu):
    """Several consecutive incompressible 4 MiB blocks in RAW mode (the JNI / fourmc_gpu_lz4_decompress route: the container would
    store them).  Their streams are the longest a block can have (one token, 4 MiB of literals), which is where the fused walk of
    the tile path wrote bitmap words beyond its block's workspace slot; a compressible block decoded in the same launch
    right behind each of them shows a neighbour's slot is left alone."""
    rng = np.random.default_rng(0x4D43)
    text = corpus(B)
    ins, comps = [], []
    for i in range(3):
        noise = rng.integers(0, 256, B, dtype=np.uint8)
        r, comp = orc_compress(noise)
        assert r >= B                                                    # an LZ4 stream longer than the block
        ins.append(noise); comps.append(comp)
        r, comp = orc_compress(text)
        ins.append(text); comps.append(comp)
    caps = [B] * len(ins)
    res, out, doffs = par._decode(gpu, comps, caps)
    for i in range(len(ins)):
        if gpu.rows_alone and res[i] == RETRY:
            continue
        assert res[i] == B, (i, int(res[i]))
        bad = np.nonzero(out[doffs[i]: doffs[i] + B] != ins[i])[0]
        assert len(bad) == 0, (i, int(bad[0]), len(bad))
        assert np.all(out[doffs[i] + B: doffs[i] + B + 7] == 0xA5), i
