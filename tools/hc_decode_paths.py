import importlib, os, sys
import numpy as np, torch
sys.path.insert(0,'.'); sys.path.insert(0,'tests')
import helpers
p = importlib.import_module("4mc_amd"); p.gpu_init(0)
B=p.BLOCKSIZE; nb=2048
base = helpers.corpus(48*B)
d_src = torch.from_numpy(base).cuda().repeat(-(-nb//48))[:nb*B].contiguous()
offs=np.arange(nb,dtype=np.uint64)*B; lens=np.full(nb,B,dtype=np.uint32)
enc=p.DeviceBatch(p.make_blocks(offs,offs,lens,lens)); st=torch.empty(nb*B,dtype=torch.uint8,device="cuda")
p.encode_blocks(d_src,st,enc,codec=p.CODEC_LZ4_HC,level=4); e=enc.download()
for path,name in ((9,"wx"),(0,"trio"),(4,"rows")):
    p.lib().fourmc_gpu_set_lz4_decode_path(path)
    out=torch.zeros(nb*B+64,dtype=torch.uint8,device="cuda"); ts=[]
    for it in range(3):
        dec=p.DeviceBatch(p.make_blocks(offs,offs,e["result"].astype(np.uint32),lens,e["xxh32"]))
        s=torch.cuda.Event(enable_timing=True); t=torch.cuda.Event(enable_timing=True)
        s.record(); p.decode_blocks(st,out,dec); t.record(); torch.cuda.synchronize(); ts.append(s.elapsed_time(t))
    print(name, f"{min(ts):.2f} ms", torch.equal(out[:nb*B],d_src))
