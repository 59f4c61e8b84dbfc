"""tools/model/lz4p_model.c, the executable statement of the ratio-tolerance LZ4 encoder's rules (lz4_par_encode.hip), checked on
the CPU: what it writes are LZ4 blocks the oracle's LZ4_decompress_safe restatement (and the reference's own decoder, when
oracle/_ref was built) decodes to the input, and their sizes stay within the tolerance the GPU test holds the kernel to.  The GPU
test then asks the kernel for the model's bytes."""
import numpy as np

import helpers
from helpers import B

TOLERANCE = 0.03


def _decodes_to(payload, want):
    n = len(want)
    r, back = helpers.orc_decompress(payload, n)
    ok = r == n and np.array_equal(back[:n], want)
    ref = helpers.ref()
    if ref is not None and ok:
        dst = np.zeros(max(n, 1) + 8, np.uint8)
        payload = np.ascontiguousarray(payload)
        ok = ref.LZ4_decompress_safe(payload.ctypes.data, dst.ctypes.data, len(payload), n) == n and np.array_equal(dst[:n], want)
    return ok


def test_model_payloads_decode():
    inputs = dict(helpers.edge_inputs())
    data = helpers.corpus(2 * B, first_block=4)
    for n in (1, 12, 13, 36, 63, 64, 65, 1000, 65535, 65536, 65537, 200000, B - 1, B):
        inputs[f"db{n}"] = data[:n]
    for k, s in inputs.items():
        cap = helpers.oracle().orc_lz4_compress_bound(len(s)) + 64
        r, out = helpers.lz4p_model_encode(s, cap)
        assert r > 0 and _decodes_to(out, s), k
        r2, _ = helpers.lz4p_model_encode(s, r - 1)                  # one byte short: does not fit
        assert r2 == 0, k


def test_model_sizes_within_tolerance():
    data = helpers.corpus(12 * B)
    tot = ref = 0
    for b in range(12):                                              # one block of every class of the S-mix
        s = data[b * B:(b + 1) * B]
        cap = helpers.oracle().orc_lz4_compress_bound(B) + 64
        r, out = helpers.lz4p_model_encode(s, cap)
        assert _decodes_to(out, s), b
        want_r, _ = helpers.orc_compress(s, cap)
        tot += min(r, B); ref += min(want_r, B)
    assert tot <= ref * (1 + TOLERANCE), (tot, ref)
