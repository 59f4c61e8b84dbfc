"""Block-parallel LZ4 decode (4mc_amd/csrc/lz4_parse.hip + lz4_exec.hip): the parser's records against the oracle's
sequence list, and the decoded bytes / return codes against the oracle on shapes that stress the window machinery
(long literal runs, long and overlapping matches, unaligned outputs, many blocks in one launch).
Reference behaviour: native/lz4/lz4.c:1936-2339 via oracle/lz4_port.c."""
import ctypes as C

import numpy as np
import pytest

import helpers
from helpers import B, corpus, edge_inputs, orc_compress, orc_decompress, pkg

torch = pytest.importorskip("torch")
pytestmark = pytest.mark.gpu
WIN = 1024


def _sequences(comp, cap, fields=False):
    """oracle's sequence list: token positions, output positions and (fields=True) [literal start, ll, ml, offset] rows"""
    L = helpers.oracle()
    L.orc_lz4_sequences_ex.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]
    L.orc_lz4_sequences_ex.restype = C.c_int
    mx = len(comp) // 3 + 16
    tok = np.zeros(mx, np.uint32); out = np.zeros(mx, np.uint32); fl = np.zeros((mx, 4), np.uint32); total = C.c_int(0)
    n = L.orc_lz4_sequences_ex(comp.ctypes.data, len(comp), cap, tok.ctypes.data, out.ctypes.data, fl.ctypes.data, mx, C.byref(total))
    if fields:
        return n, tok[:max(n, 0)], out[:max(n, 0)], total.value, fl[:max(n, 0)]
    return n, tok[:max(n, 0)], out[:max(n, 0)], total.value


def _inputs():
    rng = np.random.default_rng(11)
    ed = edge_inputs()
    text = corpus(B)
    out = {k: ed[k] for k in ("hello10", "zeros_64k", "zeros_1m", "period3", "period7", "period37", "period200", "random_100k",
                               "text_300k", "text_60k", "lit_then_run", "two_symbols", "far_repeat")}
    out["text_4m"] = text
    out["binary_4m"] = corpus(B, first_block=1)
    out["pcm_4m"] = corpus(B, first_block=2)
    out["db_4m"] = corpus(B, first_block=5)
    # long literal runs between matches, matches of every length class, offsets of every distance class
    parts = []
    for i in range(300):
        parts.append(rng.integers(0, 256, int(rng.integers(1, 3000)), dtype=np.uint8))
        src = np.concatenate(parts)
        off = int(rng.integers(1, min(len(src), 65535) + 1)); ln = int(rng.integers(4, 5000))
        seg = np.empty(ln, np.uint8)
        for k in range(ln):
            seg[k] = src[len(src) - off + k] if k < off else seg[k - off]
        parts.append(seg)
    out["mixed_runs"] = np.concatenate(parts)[: B]
    out["zeros_4m"] = np.zeros(B, np.uint8)
    out["ff_4m"] = np.full(B, 255, np.uint8)
    out["rle_pieces"] = np.repeat(rng.integers(0, 256, 4000, dtype=np.uint8), rng.integers(1, 2000, 4000))[: B]
    return out


@pytest.fixture(scope="module")
def gpu():
    """the block-parallel path is opt-in (the wave-trio path is faster today): select it for this module"""
    p = pkg(); p.gpu_init()
    p.use_research(True); p.gpu_init()                 # the pair and its debug export are in the research side build only
    before = p.lib().fourmc_gpu_get_lz4_decode_path()
    p.lib().fourmc_gpu_set_lz4_decode_path(1)
    yield p
    p.lib().fourmc_gpu_set_lz4_decode_path(before)
    p.use_research(False)


def test_parser_records_equal_the_oracle_sequence_list(gpu):
    L = gpu.lib()
    layout = (C.c_size_t * 3)()
    for name, src in _inputs().items():
        if len(src) < 64:
            continue                                            # below 64 bytes of capacity the exact kernel decodes
        r, comp = orc_compress(src)
        assert r > 0
        n, tok, opos, total, fl = _sequences(comp, len(src), fields=True)
        assert n > 0 and total == len(src), name
        for shift in (0, 37):                                   # output address alignment moves the window grid
            d_src = torch.from_numpy(comp).cuda()
            d_dst = torch.empty(len(src) + 256, dtype=torch.uint8, device="cuda")
            base = d_dst.data_ptr()
            doff = (-base) % 128 + shift
            blk = gpu.DeviceBatch(gpu.make_blocks([0], [doff], [len(comp)], [len(src)]))
            host = np.zeros(8 << 20, np.uint8)
            gpu.binding.check(L.fourmc_gpu_debug_lz4_parse(d_src.data_ptr(), d_dst.data_ptr(), blk.ptr, 1, 0, host.ctypes.data, host.nbytes, layout), "parse")
            slot, woff, toff = layout[0], layout[1], layout[2]
            hdr = host[:64].view(np.uint32)
            status, nseq, tot, nwin, a0 = int(hdr[0]), int(hdr[1]), int(hdr[2]), int(hdr[3]), int(hdr[4])
            assert status == 0, (name, shift, "parser handed the block back")
            assert (nseq, tot, a0) == (n, total, shift), (name, shift, nseq, n, tot, total, a0)
            assert nwin == (total + a0 + WIN - 1) // WIN
            rec = host[toff: toff + 8 * nseq].view(np.uint32).reshape(-1, 2).astype(np.int64)
            g_lit = rec[:, 0] & 0x7fffff; g_ll = (rec[:, 0] >> 23) | ((rec[:, 1] >> 27) << 9)
            g_ml = (rec[:, 1] >> 16) & 2047; g_off = rec[:, 1] & 0xffff
            want = fl.astype(np.int64)
            assert np.array_equal(g_lit, want[:, 0]), (name, shift, "literal start", int(np.argmax(g_lit != want[:, 0])))
            assert np.array_equal(g_ll, np.minimum(want[:, 1], 16383)), (name, shift, "ll")
            assert np.array_equal(g_ml, np.minimum(want[:, 2], 2047)), (name, shift, "ml")
            assert np.array_equal(g_off, want[:, 3]), (name, shift, "offset")
            wd = host[woff: woff + 32 * (nwin + 1)].view(np.uint32).reshape(-1, 8)
            # window w's descriptor names the sequence that covers (shifted) position w * WIN
            ends = np.concatenate([opos[1:], [total]]).astype(np.int64)
            for w in range(nwin):
                pos = max(w * WIN - a0, 0)
                i = int(np.searchsorted(ends, pos, side="right"))
                while i < n - 1 and ends[i] == opos[i] and pos >= ends[i]:
                    i += 1
                assert wd[w, 0] == i and wd[w, 1] == opos[i] and wd[w, 2] == tok[i], (name, shift, w, wd[w].tolist(), i)
                # the decoded fields of that sequence: literal start / literal length / match length
                nxt_o = int(opos[i + 1]) if i + 1 < n else total
                assert int(wd[w, 4]) + int(wd[w, 5]) == nxt_o - int(opos[i]), (name, shift, w, wd[w].tolist())
                assert [int(wd[w, 3]), int(wd[w, 4]), int(wd[w, 5]), int(wd[w, 6])] == fl[i].tolist(), (name, shift, w, wd[w].tolist(), fl[i].tolist())
            assert wd[nwin, 0] == n - 1


def _decode(gpu, comps, caps, shifts=None):
    offs, pos = [], 0
    for c in comps:
        offs.append(pos); pos += len(c) + 13            # payloads at odd alignments, as inside a .4mc image
    src = np.zeros(pos + 64, np.uint8)
    for c, o in zip(comps, offs):
        src[o:o + len(c)] = c
    doffs, dpos = [], 0
    for i, cap in enumerate(caps):
        sh = 0 if shifts is None else shifts[i]
        dpos = (dpos + 127) // 128 * 128 + sh
        doffs.append(dpos); dpos += cap + 7
    d_src = torch.from_numpy(src).cuda()
    d_dst = torch.full((dpos + 256,), 0xA5, dtype=torch.uint8, device="cuda")
    batch = gpu.DeviceBatch(gpu.make_blocks(offs, doffs, [len(c) for c in comps], caps))
    gpu.lz4_decompress(d_src, d_dst, batch)
    torch.cuda.synchronize()
    res = batch.download()["result"]
    out = d_dst.cpu().numpy()
    return res, out, doffs


def test_decode_window_shapes(gpu):
    ins = _inputs()
    names = list(ins)
    comps, caps, shifts = [], [], []
    for i, k in enumerate(names):
        r, comp = orc_compress(ins[k])
        comps.append(comp); caps.append(len(ins[k])); shifts.append((i * 29) % 128)
    res, out, doffs = _decode(gpu, comps, caps, shifts)
    for i, k in enumerate(names):
        assert res[i] == caps[i], (k, int(res[i]))
        got = out[doffs[i]: doffs[i] + caps[i]]
        bad = np.nonzero(got != ins[k])[0]
        assert len(bad) == 0, (k, int(bad[0]), len(bad))
        # nothing outside the block's output range is touched
        assert doffs[i] == 0 or out[doffs[i] - 1] == 0xA5, k
        assert np.all(out[doffs[i] + caps[i]: doffs[i] + caps[i] + 7] == 0xA5), k


def test_decode_larger_capacity_and_hc_streams(gpu):
    """dst_cap above the decoded size (the JNI / LZ4_decompress_safe contract) and streams of the HC encoder, whose
    parses have longer matches and lazy overlaps."""
    text = corpus(B)[: 1 << 20]
    binary = corpus(B, first_block=1)[: 1 << 20]
    comps, caps, want = [], [], []
    for src in (text, binary):
        for lvl in (0, 4, 8):
            r, comp = orc_compress(src) if lvl == 0 else helpers.orc_compress_hc(src, lvl)
            comps.append(comp); caps.append(len(src) + 1000 * (lvl + 1)); want.append(src)
    res, out, doffs = _decode(gpu, comps, caps)
    for i in range(len(comps)):
        assert res[i] == len(want[i])
        assert np.array_equal(out[doffs[i]: doffs[i] + len(want[i])], want[i])


def test_decode_many_blocks_one_launch(gpu):
    """more blocks than CUs, every class of the corpus, repeated: every workgroup slot gets reused"""
    base = corpus(12 * B)
    comps, caps, srcs = [], [], []
    for rep in range(3):
        for b in range(12):
            src = base[b * B:(b + 1) * B]
            if rep == 0:
                r, comp = orc_compress(src, B - 1)
                if r <= 0:
                    r, comp = orc_compress(src)
                comps.append(comp)
            else:
                comps.append(comps[b])
            caps.append(B); srcs.append(src)
    res, out, doffs = _decode(gpu, comps, caps)
    for i in range(len(comps)):
        assert res[i] == B, (i, int(res[i]))
        assert np.array_equal(out[doffs[i]: doffs[i] + B], srcs[i]), i


def test_decode_mutated_streams_match_oracle(gpu):
    """corrupt streams: accept/reject, return codes and the bytes of accepted outputs equal the oracle's"""
    rng = np.random.default_rng(5)
    text = corpus(B)[: 200000]
    r, good = orc_compress(text)
    comps, caps = [], []
    for i in range(120):
        c = good.copy()
        k = int(rng.integers(1, 4))
        for _ in range(k):
            p = int(rng.integers(0, len(c)))
            c[p] = rng.integers(0, 256)
        if i % 5 == 0:
            c = c[: int(rng.integers(1, len(c)))]
        comps.append(c); caps.append(len(text) + (0 if i % 3 else 77))
    res, out, doffs = _decode(gpu, comps, caps)
    for i, c in enumerate(comps):
        want_r, want = orc_decompress(c, caps[i])
        assert res[i] == want_r, (i, int(res[i]), want_r)
        if want_r > 0:
            assert np.array_equal(out[doffs[i]: doffs[i] + want_r], want), i
