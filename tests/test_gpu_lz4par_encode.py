"""The ratio-tolerance LZ4 encoder (4mc_amd/csrc/lz4_par_encode.hip; fourmc_gpu_set_lz4_encode_mode(1) / FOURMC_LZ4_ENCODE=parallel).

It does not reproduce the reference parse (native/lz4/lz4.c:910-1302), so byte identity with the reference is NOT the bar.  The
bar is the one north_star states for that case ("otherwise compression ratio is reported within a stated tolerance"):
  * every payload is ONE LZ4 block that the reference's LZ4_decompress_safe (native/lz4/lz4.c:2345: oracle/_ref when it was built,
    the oracle's restatement always) accepts and decodes to the input;
  * sizes within TOLERANCE of the reference parse on the S-mix (all 48 distinct blocks);
  * container mode keeps the reference's rules (capacity n - 1, stored blocks, XXH32 of the payload): the image is a .4mc file the
    reference CLI reads;
  * and, as a determinism / specification check, the bytes are the ones tools/model/lz4p_model.c (the executable statement of the
    kernel's rules) produces."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest
import torch

import helpers
from helpers import B

pytestmark = pytest.mark.gpu

TOLERANCE = 0.03            # total size of the 48 S-mix payloads against the reference parse (measured: 2.2 %)


@pytest.fixture()
def par(gpu):
    before = gpu.lib().fourmc_gpu_get_lz4_encode_mode()
    gpu.lib().fourmc_gpu_set_lz4_encode_mode(1)
    yield gpu
    gpu.lib().fourmc_gpu_set_lz4_encode_mode(before)


@pytest.fixture(scope="module")
def model():
    return helpers.lz4p_model()


def _dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


def _inputs():
    inputs = dict(helpers.edge_inputs())
    data = helpers.corpus(12 * B)
    for b in (0, 1, 4, 8, 9):
        inputs[f"corpus{b}"] = data[b * B:(b + 1) * B]
    for n in (1, 12, 13, 31, 32, 33, 36, 63, 64, 65, 100, 1000, 65535, 65536, 65537, 65540, 131072, 200000, B - 1):
        inputs[f"text{n}"] = data[5 * B:5 * B + n]
    return inputs


def _encode(gpu, arrays, caps, shift=0):
    """All arrays in ONE launch; `shift` misaligns the inputs.  Returns (results, payload arrays)."""
    offs, pos = [], shift
    for a in arrays:
        offs.append(pos); pos += len(a) + 3          # (inputs end 3 bytes before the next one starts: ragged on purpose)
    buf = np.zeros(pos + 64, np.uint8)
    for a, o in zip(arrays, offs):
        buf[o:o + len(a)] = a
    doffs, dpos = [], 0
    for c in caps:
        doffs.append(dpos); dpos += c + 16
    d_dst = torch.full((dpos + 64,), 0xA5, dtype=torch.uint8, device="cuda")
    batch = gpu.DeviceBatch(gpu.make_blocks(offs, doffs, [len(a) for a in arrays], caps))
    gpu.lz4_compress_fast(_dev(buf), d_dst, batch)
    res = batch.download()["result"]
    out = d_dst.cpu().numpy()
    for c, o in zip(caps, doffs):                    # nothing is written beyond a block's capacity
        assert np.all(out[o + c:o + c + 16] == 0xA5)
    return [int(r) for r in res], [out[o:o + max(int(r), 0)] for r, o in zip(res, doffs)]


def _decodes_to(payload, want):
    n = len(want)
    r, back = helpers.orc_decompress(payload, n)
    ok = r == n and np.array_equal(back[:n], want)
    ref = helpers.ref()
    if ref is not None and ok:                       # the reference's own decoder, when oracle/_ref was built
        dst = np.zeros(max(n, 1) + 8, np.uint8)
        payload = np.ascontiguousarray(payload)
        r2 = ref.LZ4_decompress_safe(payload.ctypes.data, dst.ctypes.data, len(payload), n)
        ok = r2 == n and np.array_equal(dst[:n], want)
    return ok


def test_payloads_are_lz4_blocks_the_reference_decodes(par, model):
    inputs = _inputs()
    names = [k for k in inputs]
    arrays = [inputs[k] for k in names]
    caps = [helpers.oracle().orc_lz4_compress_bound(len(a)) + 64 for a in arrays]
    for shift in (0, 5):
        res, outs = _encode(par, arrays, caps, shift)
        for k, a, r, o, cap in zip(names, arrays, res, outs, caps):
            assert r > 0, (k, r)
            assert _decodes_to(o, a), (k, shift)
            mr, mb = helpers.lz4p_model_encode(a, cap)
            assert r == mr and np.array_equal(o, mb), (k, shift, r, mr)


def test_sizes_within_tolerance_of_the_reference_parse(par):
    data = helpers.corpus(48 * B)
    arrays = [data[b * B:(b + 1) * B] for b in range(48)]
    cap = helpers.oracle().orc_lz4_compress_bound(B) + 64
    res, outs = _encode(par, arrays, [cap] * 48)
    tot = ref = 0
    for b in range(48):
        assert res[b] > 0 and _decodes_to(outs[b], arrays[b]), b
        want_r, _ = helpers.orc_compress(arrays[b], cap)
        # what a container stores: min(payload, block)
        tot += min(res[b], B); ref += min(want_r, B)
    assert tot <= ref * (1 + TOLERANCE), (tot, ref, tot / ref)


def test_capacity_is_honoured(par):
    """result 0 when the block does not fit dst_cap (LZ4_compress_default's convention), the same bytes when it does"""
    data = helpers.corpus(3 * B)
    srcs = [data[5000:5000 + n] for n in (100, 5000, 70000, 300000)] + [np.zeros(100000, np.uint8)]
    big = [helpers.oracle().orc_lz4_compress_bound(len(s)) + 64 for s in srcs]
    res, outs = _encode(par, srcs, big)
    exact, _ = _encode(par, srcs, res)                               # capacity = size: fits
    assert exact == res
    short, souts = _encode(par, srcs, [r - 1 for r in res])          # one byte less: does not
    assert short == [0] * len(srcs)
    assert all(len(o) == 0 for o in souts)


def test_container_mode_image_is_read_by_the_reference(par, tmp_path):
    n = 9 * B + 12345
    data = helpers.corpus(n, first_block=3)
    nb = -(-n // B)
    lens = [min(B, n - b * B) for b in range(nb)]
    batch = par.DeviceBatch(par.make_blocks([b * B for b in range(nb)], [b * B for b in range(nb)], lens, lens))
    d_dst = torch.zeros(nb * B, dtype=torch.uint8, device="cuda")
    par.encode_blocks(_dev(data), d_dst, batch)
    enc = batch.download()
    out = d_dst.cpu().numpy()
    payloads = [out[b * B: b * B + enc["result"][b]] for b in range(nb)]
    assert any(enc["result"][b] == lens[b] for b in range(nb)), "the corpus should exercise a stored block"
    for b in range(nb):
        assert 0 < enc["result"][b] <= lens[b]
        assert enc["xxh32"][b] == helpers.orc_xxh32(payloads[b])
        if enc["result"][b] == lens[b]:
            assert np.array_equal(payloads[b], data[b * B:b * B + lens[b]])          # stored raw (native/4mc.c:318-329)
    image = par.assemble_container(par.MAGIC_4MC, lens, enc["result"], enc["xxh32"], payloads)
    got, back, _ = helpers.orc_container_decode(np.frombuffer(image, np.uint8), n)
    assert got == n and np.array_equal(back[:n], data)
    ref = helpers.ref_cli()
    if ref:
        f = tmp_path / "x.4mc"; f.write_bytes(image)
        o = tmp_path / "x.bin"
        r = subprocess.run([ref, "-d", "-f", str(f), str(o)], capture_output=True)
        assert r.returncode == 0, r.stderr
        assert o.read_bytes() == data.tobytes()
    # and by this engine's own decoder, in place from HBM
    img = np.frombuffer(image, dtype=np.uint8)
    dblocks, used = par.split_container(img, par.MAGIC_4MC)
    dbatch = par.DeviceBatch(dblocks)
    d_out = torch.zeros(n + 64, dtype=torch.uint8, device="cuda")
    par.decode_blocks(_dev(np.concatenate([img, np.zeros(64, np.uint8)])), d_out, dbatch)
    assert list(dbatch.download()["result"]) == lens
    assert np.array_equal(d_out.cpu().numpy()[:n], data)


def test_cli_with_the_environment_switch(gpu, tmp_path):
    """FOURMC_LZ4_ENCODE=parallel: the CLI writes a .4mc the reference CLI (and this one) reads; without it the bytes are the
    reference's."""
    data = helpers.corpus(2 * B + 777, first_block=6)
    src = tmp_path / "in.bin"; src.write_bytes(data.tobytes())
    out = tmp_path / "par.4mc"; exact = tmp_path / "exact.4mc"
    env = dict(os.environ, FOURMC_LZ4_ENCODE="parallel")
    assert subprocess.run([gpu.cli_path(), "-f", str(src), str(out)], capture_output=True, env=env).returncode == 0
    assert subprocess.run([gpu.cli_path(), "-f", str(src), str(exact)], capture_output=True).returncode == 0
    assert exact.read_bytes() == helpers.orc_container(data).tobytes()
    assert out.read_bytes() != exact.read_bytes()
    back = tmp_path / "back.bin"
    tool = helpers.ref_cli() or gpu.cli_path()
    assert subprocess.run([tool, "-d", "-f", str(out), str(back)], capture_output=True).returncode == 0
    assert back.read_bytes() == data.tobytes()


def test_full_launch_round_trip(par):
    """2048 blocks (BASELINE.json configs[1]'s launch): every payload decoded by the device decoder gives the input back; the
    copies of a corpus block give the same bytes (no dependence on where a block runs)."""
    nb = 2048
    data = helpers.corpus(48 * B)
    src = _dev(data).repeat((nb + 47) // 48)[:nb * B].contiguous()
    cap = B + B // 255 + 64
    stride = (cap + 255) & ~255
    dst = torch.zeros(nb * stride, dtype=torch.uint8, device="cuda")
    batch = par.DeviceBatch(par.make_blocks([i * B for i in range(nb)], [i * stride for i in range(nb)], [B] * nb, [cap] * nb))
    par.lz4_compress_fast(src, dst, batch)
    res = batch.download()["result"]
    assert all(res[i] == res[i % 48] for i in range(nb))
    for i in (48, 1000, 2047):
        assert torch.equal(dst[i * stride:i * stride + int(res[i])], dst[(i % 48) * stride:(i % 48) * stride + int(res[i])])
    back = torch.zeros(nb * B, dtype=torch.uint8, device="cuda")
    dbatch = par.DeviceBatch(par.make_blocks([i * stride for i in range(nb)], [i * B for i in range(nb)], [int(r) for r in res], [B] * nb))
    par.lz4_decompress(dst, back, dbatch)
    assert all(r == B for r in dbatch.download()["result"])
    assert torch.equal(back, src)


def test_fuzz_slice(gpu):
    """tools/fuzz_k2p.py (built inputs around the window / segment / end-of-block limits, misaligned, capacities at and below the
    size): 20 seeds here, thousands by hand (DESIGN.md has the totals)."""
    import sys
    r = subprocess.run([sys.executable, os.path.join(helpers.ROOT, "tools", "fuzz_k2p.py"), "5000", "20"], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
