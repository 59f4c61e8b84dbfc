#!/usr/bin/env python3
"""Debug / timing aid for the ratio-tolerance LZ4 encoder (lz4_par_encode.hip): every payload must decode with the oracle's
LZ4_decompress_safe restatement; sizes against the reference parse and against tools/model/lz4p_model.c; launch timing.
  python tools/k2p_debug.py [--blocks 2048] [--no-edge]"""
import importlib, sys, os, subprocess, ctypes as C, argparse, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import helpers
ap = argparse.ArgumentParser(); ap.add_argument("--blocks", type=int, default=0); ap.add_argument("--no-edge", action="store_true")
args = ap.parse_args()
p = importlib.import_module("4mc_amd"); p.gpu_init(0)
B = p.BLOCKSIZE
so = "/tmp/liblz4p_model.so"
subprocess.check_call(["gcc", "-O2", "-shared", "-fPIC", "-DLZ4P_NO_MAIN", "-o", so, os.path.join(ROOT, "tools/model/lz4p_model.c")])
model = C.CDLL(so); model.lz4p_model_encode.restype = C.c_int
model.lz4p_model_encode.argtypes = [C.c_void_p, C.c_uint32, C.c_void_p, C.c_uint32]
import re
NENT = int(re.search(r"#define FOURMC_PAR_NENT (\d+)", open(os.path.join(ROOT, "4mc_amd/csrc/lz4_par_encode.hip")).read()).group(1))
model.lz4p_model_set(NENT, 16, 4, 3, 12, 0x16)
orc = helpers.oracle()
p.lib().fourmc_gpu_set_lz4_encode_mode(1)

def model_encode(s, cap):
    out = np.zeros(cap + 8, dtype=np.uint8)
    s = np.ascontiguousarray(s)
    r = model.lz4p_model_encode(s.ctypes.data if len(s) else None, len(s), out.ctypes.data, cap)
    return r, out[:max(r, 0)]

if not args.no_edge:
    inputs = dict(helpers.edge_inputs())
    data = helpers.corpus(12 * B)
    for b in range(12): inputs[f"corpus{b}"] = data[b * B:(b + 1) * B]
    for n in (1, 12, 13, 31, 32, 33, 36, 63, 64, 65, 100, 1000, 65535, 65536, 65537, 65540, 131072, 200000, B - 1):
        inputs[f"text{n}"] = data[5 * B:5 * B + n]
    bad = 0; tot = ref = mod = 0
    for k, s in inputs.items():
        n = len(s)
        cap = orc.orc_lz4_compress_bound(n) + 64
        src = torch.from_numpy(np.ascontiguousarray(s)).cuda() if n else torch.zeros(1, dtype=torch.uint8, device="cuda")
        dst = torch.zeros(cap + 64, dtype=torch.uint8, device="cuda")
        batch = p.DeviceBatch(p.make_blocks([0], [0], [n], [cap]))
        p.lz4_compress_fast(src, dst, batch)
        r = int(batch.download()["result"][0])
        got = dst[:max(r, 0)].cpu().numpy()
        want_r, _ = helpers.orc_compress(s, cap)
        ok = r > 0
        if ok:
            dr, back = helpers.orc_decompress(got, n)
            ok = dr == n and np.array_equal(back[:n], s)
        mr, mb = model_encode(s, cap)
        same = mr == r and np.array_equal(mb, got)
        if not ok or not same:
            bad += 1
            m = min(len(got), len(mb)); d = np.nonzero(got[:m] != mb[:m])[0]
            print(f"{k}: n={n} r={r} ref={want_r} model={mr} decodes={ok} first diff vs model at {int(d[0]) if len(d) else m}")
        tot += max(r, 0); ref += want_r; mod += mr
    print(f"edge + corpus inputs: {len(inputs)}, bad {bad}; bytes gpu {tot} model {mod} reference parse {ref} ({100.0 * (tot - ref) / ref:+.2f} %)")

if args.blocks:
    nb = args.blocks
    nrep = (nb + 47) // 48
    data = helpers.corpus(48 * B)
    src = torch.from_numpy(data).cuda().repeat(nrep)[:nb * B].contiguous()
    cap = B + B // 255 + 64
    stride = (cap + 255) & ~255
    dst = torch.zeros(nb * stride, dtype=torch.uint8, device="cuda")
    offs = [i * B for i in range(nb)]; doffs = [i * stride for i in range(nb)]
    batch = p.DeviceBatch(p.make_blocks(offs, doffs, [B] * nb, [cap] * nb))
    for mode in (1, 0):
        p.lib().fourmc_gpu_set_lz4_encode_mode(mode)
        best = 1e9
        for it in range(3):
            torch.cuda.synchronize(); t0 = time.perf_counter()
            p.lz4_compress_fast(src, dst, batch)
            torch.cuda.synchronize(); best = min(best, time.perf_counter() - t0)
        res = batch.download()["result"]
        print(f"mode {mode} ({'parallel' if mode else 'exact'}): {nb} blocks best {best * 1e3:.2f} ms = {nb * B / best / 1e9:.1f} GB/s in, csize {int(res.sum())} ratio {nb * B / float(res.sum()):.4f}")
        if mode == 1:
            # every distinct payload through the oracle decoder
            bad = 0
            for i in range(min(nb, 48)):
                r = int(res[i]); got = dst[i * stride:i * stride + r].cpu().numpy()
                dr, back = helpers.orc_decompress(got, B)
                if dr != B or not np.array_equal(back[:B], data[i * B:(i + 1) * B]): bad += 1
            print(f"  oracle decoder on the first {min(nb, 48)} payloads: bad {bad}")
