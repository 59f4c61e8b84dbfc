#!/usr/bin/env python3
"""Debug aid for the zstd level-1 match finder: encodes ONE input of <= 128 KiB (one inner block) twice - with the
wave-uniform transcription of the reference loop (FOURMC_ZSTD_SERIAL=1) and with the product path - and prints the first
sequences where the two sequence stores differ.    python tools/z1_debug.py text_30k | <size> <class> <seed>"""
import ctypes as C, importlib, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import helpers
p = importlib.import_module("4mc_amd"); p.gpu_init(0)
if not os.environ.get("FOURMC_LIB"): p.use_research(True); p.gpu_init(0)     # debug exports: research side build
SEQCAP = 32768 + 64
TABLES = []
LEVEL = int(os.environ.get("ZL", "1"))          # 1 or 3 (FOURMC_ZSTD_SERIAL=1: the wave-uniform transcription at level 1, the batched search alone at level 3)

def seqs(src, serial):
    os.environ["FOURMC_ZSTD_SERIAL"] = "1" if serial else "0"
    n = len(src)
    batch = p.DeviceBatch(p.make_blocks([0], [0], [n], [n + 1024]))
    out = torch.zeros(n + 2048, dtype=torch.uint8, device="cuda")
    buf = np.zeros(n + 64, np.uint8); buf[:n] = src
    p.zstd_compress(torch.from_numpy(buf).cuda(), out, batch, LEVEL)
    torch.cuda.synchronize()
    r = int(batch.download()["result"][0])
    raw = (C.c_uint32 * (3 * SEQCAP))()
    assert p.lib().fourmc_gpu_debug_read_workspace(raw, 0, 12 * SEQCAP) == 0
    a = np.frombuffer(raw, np.uint32).reshape(3, SEQCAP).copy()
    global TABLES
    nt = (1 << 15) if LEVEL == 1 else (1 << 17) + (1 << 16)     # level 3: long table (2^17) then short table (2^16)
    traw = (C.c_uint32 * nt)()
    assert p.lib().fourmc_gpu_debug_read_workspace(traw, 629312, 4 * nt) == 0    # zstd_encode.hip: kStoreBytes, the hash table(s) behind it
    TABLES.append(np.frombuffer(traw, np.uint32).copy())
    return r, a, out[:max(r, 0)].cpu().numpy()

if sys.argv[1] == "corpus":                                  # block B of the S-mix corpus: the first 128 KiB inner block whose output differs, cut there
    bi = int(sys.argv[2])
    full = helpers.corpus((bi + 1) * (4 << 20))[bi * (4 << 20):].copy()
    def frame(d, serial):
        os.environ["FOURMC_ZSTD_SERIAL"] = "1" if serial else "0"
        n = len(d); batch = p.DeviceBatch(p.make_blocks([0], [0], [n], [n + 4096]))
        out = torch.zeros(n + 8192, dtype=torch.uint8, device="cuda")
        p.zstd_compress(torch.from_numpy(d).cuda(), out, batch, LEVEL); torch.cuda.synchronize()
        r = int(batch.download()["result"][0]); return out[:r].cpu().numpy()
    cut = None
    for k in range(1, 33):
        a, b = frame(full[: k * 131072], True), frame(full[: k * 131072], False)
        if len(a) != len(b) or not np.array_equal(a, b): cut = k; break
    print("first differing inner block:", cut)
    if cut is None: sys.exit(0)
    src = full[: cut * 131072]
elif sys.argv[1] == "edge":                                    # tests/helpers.py: edge_inputs() - lists the mismatching ones, or takes one by name (cut to 128 KiB)
    ed = helpers.edge_inputs()
    if len(sys.argv) == 2:
        for k, d in ed.items():
            d = np.ascontiguousarray(d)
            r1, _, o1 = seqs(d, False) if len(d) else (0, None, None)
            wr, w = helpers.orc_zstd_compress(d, LEVEL, len(d) + 1024)
            if len(d) and not (r1 == wr and np.array_equal(o1, w)): print(k, len(d), "MISMATCH", r1, wr)
        sys.exit(0)
    src = np.ascontiguousarray(ed[sys.argv[2]])[: int(sys.argv[3]) if len(sys.argv) > 3 else 128 * 1024]
elif sys.argv[1] == "sizes":                                   # the inputs of tests/test_gpu_zstd_enc.py::test_zstd_size_classes_and_tails
    rng = np.random.default_rng(5)
    big = helpers.corpus(3 * (4 << 20), first_block=5)
    sizes = [7, 8, 18, 19, 20, 63, 64, 65, 255, 256, 257, 1023, 1024, 1025, 16383, 16384, 16385, 65791, 65792,
             131071, 131072, 131073, 131078, 131079, 131080, 262144, 262145, 262151, 393216 + 3, 524288, 524289, 1500001,
             2 * 1024 * 1024 + 77, 3 * 1024 * 1024]
    offs = rng.integers(0, 4 << 20, len(sizes))
    if len(sys.argv) == 2:
        for n, o in zip(sizes, offs):
            if n > 128 * 1024: continue
            d = big[int(o): int(o) + n].copy()
            r1, _, o1 = seqs(d, False); wr, w = helpers.orc_zstd_compress(d, LEVEL, n + 1024)
            print(n, "ok" if r1 == wr and np.array_equal(o1, w) else "MISMATCH %d %d" % (r1, wr))
        sys.exit(0)
    i = sizes.index(int(sys.argv[2])); src = big[int(offs[i]): int(offs[i]) + sizes[i]].copy()
elif len(sys.argv) == 2:
    src = np.ascontiguousarray(helpers.golden_zstd_inputs()[sys.argv[1]])
else:
    n, cls, seed = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
    src = helpers.corpus(4 << 20, first_block=cls, seed=seed)[:n].copy()
r0, a0, o0 = seqs(src, True)
r1, a1, o1 = seqs(src, False)
want_r, want = helpers.orc_zstd_compress(src, LEVEL, len(src) + 1024)
print("serial", r0, "product", r1, "oracle", want_r, "serial==oracle", r0 == want_r and np.array_equal(o0, want))
pos0 = pos1 = ((len(src) - 1) // 131072) * 131072           # (the sequence store holds the last inner block)
for i in range(SEQCAP):
    t0, t1 = a0[:, i], a1[:, i]
    if (t0 != t1).any():
        print("first difference at sequence", i, "position", pos0)
        for j in range(max(0, i - 3), i + 4):
            print(j, "serial ll/ml/of", a0[0, j], a0[1, j] + 3, a0[2, j], "| product", a1[0, j], a1[1, j] + 3, a1[2, j])
        lo = max(0, pos0 - 16)
        print("input around:", bytes(src[lo:pos0 + 80]))
        break
    pos0 += int(t0[0]) + int(t0[1]) + 3
    if pos0 >= len(src) - 8: print("no difference in", i + 1, "sequences"); break

if os.environ.get("POS"):                                    # both sequence lists around a position
    P = int(os.environ["POS"])
    for name, a in (("serial", a0), ("product", a1)):
        pos = 0
        for i in range(SEQCAP):
            ll, ml, of = int(a[0, i]), int(a[1, i]) + 3, int(a[2, i])
            if pos + ll + ml > P - 80 and pos < P + 40: print(name, i, "at", pos, "ll", ll, "match", pos + ll, "ml", ml, "of", of, "end", pos + ll + ml)
            pos += ll + ml
            if pos > P + 40: break

if os.environ.get("TABLE"):                                  # hash table at the end of the block: slots that differ (value - 2 = position)
    t0, t1 = TABLES[-2], TABLES[-1]                          # (serial, product)
    d = np.nonzero(t0 != t1)[0]
    print(len(d), "slots differ; lowest positions first:")
    for i in sorted(d, key=lambda i: min(int(t0[i]), int(t1[i])))[:12]: print("slot", i, "serial pos", int(t0[i]) - 2, "product pos", int(t1[i]) - 2)
