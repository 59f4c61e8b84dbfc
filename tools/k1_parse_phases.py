import importlib, sys, os, ctypes as C
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import helpers
p = importlib.import_module("4mc_amd"); p.gpu_init(0)
B = p.BLOCKSIZE
data = helpers.corpus(12 * B)
for b, name in ((0, "text"), (1, "binary"), (3, "sdf")):
    src = torch.from_numpy(data[b * B:(b + 1) * B].copy()).cuda()
    S = (B + B // 255 + 16 + 63) & ~63
    enc = p.DeviceBatch(p.make_blocks([0], [0], [B], [S]))
    stage = torch.empty(S, dtype=torch.uint8, device="cuda")
    p.lz4_compress_fast(src, stage, enc)
    r = enc.download()
    dec = p.DeviceBatch(p.make_blocks([0], [0], r["result"].astype(np.uint32), [B]))
    out = torch.zeros(B + 64, dtype=torch.uint8, device="cuda")
    p.lz4_decompress(stage, out, dec); torch.cuda.synchronize()
    buf = (C.c_uint64 * 8)(); p.lib().fourmc_k1_prof(buf)
    v = list(buf); n = max(v[7], 1)
    names = ["loop/general", "ring read", "bpermute", "walk", "scan+bad", "slot", "write+publish"]
    print(name, "records", v[7], " clk/record:", " ".join(f"{nm} {v[i] / n:.0f}" for i, nm in enumerate(names)), " total", sum(v[:7]) / n)
