#!/usr/bin/env python3
"""Does the decode of batch i run beside the encode of batch i + 1?  The exact LZ4 encoder keeps 8 waves per CU (its hash tables fill the
LDS) and is a latency chain per block: three quarters of the chip's wave slots are idle while it runs.  Two streams, two image / output
buffers: stream A encodes + packs batch i + 1 while stream B verifies + decodes batch i.
    python tools/overlap_probe.py [blocks] [steps]"""
import importlib, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import helpers
p = importlib.import_module("4mc_amd"); p.gpu_init(0); L = p.lib()
B = p.BLOCKSIZE
nb = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
K = int(sys.argv[2]) if len(sys.argv) > 2 else 6
base = helpers.corpus(48 * B)
d_src = torch.from_numpy(base).cuda().repeat(-(-nb // 48))[: nb * B].contiguous()
offs = np.arange(nb, dtype=np.uint64) * B; lens = np.full(nb, B, np.uint32)
enc = p.DeviceBatch(p.make_blocks(offs, offs, lens, lens))
d_stage = torch.empty(nb * B, dtype=torch.uint8, device="cuda")
imgs = [torch.empty(nb * (B + 12) + 4096, dtype=torch.uint8, device="cuda") for _ in range(2)]
outs = [torch.empty(nb * B + 64, dtype=torch.uint8, device="cuda") for _ in range(2)]
sa = torch.cuda.current_stream(); sb = torch.cuda.Stream()
spa, spb = int(sa.cuda_stream), int(sb.cuda_stream)
def compress(i):
    p.binding.check(L.fourmc_gpu_4mc_encode_blocks(d_src.data_ptr(), d_stage.data_ptr(), enc.ptr, nb, 0, 0, spa), "encode")
    desc = enc.d.view(torch.int32).view(nb, 8)
    csz = desc[:, 6].to(torch.int64)
    off = (torch.cumsum(csz + 12, 0) - (csz + 12) + 12).contiguous()
    p.binding.check(L.fourmc_gpu_4mc_pack_image(d_stage.data_ptr(), imgs[i & 1].data_ptr(), enc.ptr, off.data_ptr(), nb, spa), "pack")
    dd = torch.empty_like(desc); d64 = dd.view(torch.int64)
    d64[:, 0] = off + 12; d64[:, 1] = torch.arange(nb, device="cuda", dtype=torch.int64) * B
    dd[:, 4] = desc[:, 6]; dd[:, 5] = desc[:, 4]; dd[:, 6] = 0; dd[:, 7] = desc[:, 7]
    return dd, off
def run(overlap):
    freed = [None, None]; keep = []
    torch.cuda.synchronize()
    t0 = torch.cuda.Event(enable_timing=True); t1 = torch.cuda.Event(enable_timing=True)
    t0.record(sa)
    for i in range(K):
        if freed[i & 1] is not None: sa.wait_event(freed[i & 1])         # the image / output pair of batch i - 2 has been decoded
        dd, off = compress(i); keep.append((dd, off))
        if overlap:
            ready = torch.cuda.Event(); ready.record(sa); sb.wait_event(ready)
            p.binding.check(L.fourmc_gpu_4mc_decode_blocks(imgs[i & 1].data_ptr(), outs[i & 1].data_ptr(), dd.data_ptr(), nb, 0, spb), "decode")
            f = torch.cuda.Event(); f.record(sb); freed[i & 1] = f
        else:
            p.binding.check(L.fourmc_gpu_4mc_decode_blocks(imgs[i & 1].data_ptr(), outs[i & 1].data_ptr(), dd.data_ptr(), nb, 0, spa), "decode")
    if overlap: sa.wait_stream(sb)
    t1.record(sa); torch.cuda.synchronize()
    ok = all(bool((dd[:, 6] == B).all()) for dd, _ in keep[-2:]) and all(torch.equal(o[: nb * B], d_src) for o in outs)
    return t0.elapsed_time(t1) / K, ok
run(False); run(True)
for mode in (False, True, False, True):
    ms, ok = run(mode)
    print(f"blocks {nb} steps {K} {'two streams (decode i beside encode i+1)' if mode else 'one stream'}: {ms:8.2f} ms per step  {nb * B / ms / 1e6:6.2f} GB/s  round trip {'ok' if ok else 'BAD'}", flush=True)
