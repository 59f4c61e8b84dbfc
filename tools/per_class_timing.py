#!/usr/bin/env python3
"""Times encode / decode of each S-mix block class alone (1 block per launch) and 64 copies of it
(64 blocks per launch) - shows which class sets the kernel time of the mixed batch."""
import importlib, sys, os
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import helpers
p = importlib.import_module("4mc_amd"); p.gpu_init(0)
B = p.BLOCKSIZE
names = ["text", "binary", "pcm6", "sdf", "binary", "db", "text", "code", "pcm11", "dict", "xml", "random"]
data = helpers.corpus(12 * B)
def t(fn):
    s = torch.cuda.Event(enable_timing=True); e = torch.cuda.Event(enable_timing=True)
    fn(); torch.cuda.synchronize(); s.record(); fn(); e.record(); torch.cuda.synchronize(); return s.elapsed_time(e)
for reps in (1, 64):
    print(f"--- {reps} block(s) of one class per launch: ms  [encode_fast, decode, hc4(1 only), mc(1 only)]")
    for b in range(12):
        src = torch.from_numpy(np.tile(data[b * B:(b + 1) * B], reps)).cuda()
        offs = np.arange(reps, dtype=np.uint64) * B; lens = np.full(reps, B, np.uint32)
        enc = p.DeviceBatch(p.make_blocks(offs, offs, lens, lens))
        stage = torch.empty(reps * B, dtype=torch.uint8, device="cuda")
        te = t(lambda: p.encode_blocks(src, stage, enc))
        r = enc.download()
        dec = p.DeviceBatch(p.make_blocks(offs, offs, r["result"].astype(np.uint32), lens, r["xxh32"]))
        out = torch.empty(reps * B + 64, dtype=torch.uint8, device="cuda")
        td = t(lambda: p.decode_blocks(stage, out, dec))
        extra = ""
        if reps == 1:
            th = t(lambda: p.encode_blocks(src, stage, enc, codec=p.CODEC_LZ4_HC, level=4))
            tm = t(lambda: p.encode_blocks(src, stage, enc, codec=p.CODEC_LZ4_MC))
            extra = f"  hc4 {th:8.1f}  mc {tm:8.1f}"
        print(f"{b:2d} {names[b]:7s} csize {int(r['result'][0]):8d}  enc {te:8.2f}  dec {td:8.2f}{extra}")
