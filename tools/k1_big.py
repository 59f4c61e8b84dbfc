#!/usr/bin/env python3
"""The 64 GiB decode leg alone (container mode, 16384 blocks reading 8 copies of the 8 GiB image), for a sweep of launch knobs:
  python tools/k1_big.py [blocks]      env: FOURMC_SEG_BATCH, FOURMC_DECODE"""
import importlib, sys, os, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import helpers
p = importlib.import_module("4mc_amd"); p.gpu_init(0)
B = p.BLOCKSIZE; L = p.lib()
nd = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
nb = 2048
base = helpers.corpus(48 * B)
d_src = torch.from_numpy(base).cuda().repeat(-(-nb // 48))[: nb * B].contiguous()
offs = np.arange(nb, dtype=np.uint64) * B; lens = np.full(nb, B, np.uint32)
enc = p.DeviceBatch(p.make_blocks(offs, offs, lens, lens))
d_stage = torch.empty(nb * B, dtype=torch.uint8, device="cuda")
p.encode_blocks(d_src, d_stage, enc)
e = enc.download()
csz = torch.from_numpy(e["result"].astype(np.int64)).cuda()
img_off = (torch.cumsum(csz + 12, 0) - (csz + 12) + 12).contiguous()
img_bytes = int((csz + 12).sum().item()) + 12
img_pad = (img_bytes + 4095) & ~4095
reps = -(-nd // nb)
img = torch.zeros(reps * img_pad + 4096, dtype=torch.uint8, device="cuda")
one = torch.zeros(img_pad, dtype=torch.uint8, device="cuda")
p.pack_image(d_stage, one, enc, img_off)
for k in range(reps): img[k * img_pad: (k + 1) * img_pad] = one
del d_stage, one
so = np.tile(img_off.cpu().numpy().astype(np.uint64) + 12, reps)[:nd] + (np.arange(nd, dtype=np.uint64) // nb) * np.uint64(img_pad)
dec = p.DeviceBatch(p.make_blocks(so, np.arange(nd, dtype=np.uint64) * B, np.tile(e["result"].astype(np.uint32), reps)[:nd], np.full(nd, B, np.uint32), np.tile(e["xxh32"], reps)[:nd]))
big = torch.empty(nd * B + 64, dtype=torch.uint8, device="cuda")
def run():
    s = torch.cuda.Event(enable_timing=True); t = torch.cuda.Event(enable_timing=True)
    s.record(); p.decode_blocks(img, big, dec); t.record(); torch.cuda.synchronize(); return s.elapsed_time(t)
run()
ts = [run() for _ in range(3)]
ok = all(torch.equal(big[k * B:(k + min(nb, nd - k)) * B], d_src[: min(nb, nd - k) * B]) for k in range(0, nd, nb))
alg = int(csz.sum().item()) * nd / nb + nd * B
print(f"blocks {nd} seg_batch {os.environ.get('FOURMC_SEG_BATCH','default')} decode {os.environ.get('FOURMC_DECODE','auto')}: best {min(ts):.2f} ms ({nd * B / min(ts) / 1e6:.1f} GB/s out, {alg / min(ts) / 1e6 / 8000 * 100:.2f} % of 8 TB/s) roundtrip {'ok' if ok else 'BAD'}", flush=True)
