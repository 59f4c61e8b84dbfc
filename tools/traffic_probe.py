#!/usr/bin/env python3
"""HBM traffic of the headline launches, measured with the PMC counters: bench.py runs this script twice under rocprofv3
(`--pmc FETCH_SIZE`, then `--pmc WRITE_SIZE`: the TCC block cannot hold both in one pass) and reads the result databases.

    python tools/traffic_probe.py --child [--blocks N]         the workload: ONE encode / hash / pack / verify / decode of N blocks
    python tools/traffic_probe.py [--blocks N]                 runs both passes, prints one JSON object

Per kernel: bytes = FETCH_SIZE + WRITE_SIZE (the counters are in KiB).  Calibration (MI355X_MICROARCH.md, HBM section: on gfx950
FETCH_SIZE tallies the 128-byte requests of a wide 16 B / lane stream at 64 bytes; other widths are uncalibrated - calibrate on a
known byte count in your own access pattern): two kernels of the same pass move a KNOWN number of bytes - xxh32 reads every
payload byte exactly once (16 B / lane stream), pack reads the payloads and writes payloads + 12 bytes per block - and give
the factors reported beside the raw numbers."""
import argparse, importlib, json, os, sqlite3, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))


def child(nb):
    import numpy as np, torch
    import helpers
    p = importlib.import_module("4mc_amd"); p.gpu_init(0)
    B = p.BLOCKSIZE
    base = helpers.corpus(48 * B)
    d_src = torch.from_numpy(base).cuda().repeat(-(-nb // 48))[: nb * B].contiguous()
    offs = np.arange(nb, dtype=np.uint64) * B; lens = np.full(nb, B, np.uint32)
    enc = p.DeviceBatch(p.make_blocks(offs, offs, lens, lens))
    d_stage = torch.empty(nb * B, dtype=torch.uint8, device="cuda")
    p.encode_blocks(d_src, d_stage, enc)
    e = enc.download()
    csz = torch.from_numpy(e["result"].astype(np.int64)).cuda()
    img_off = (torch.cumsum(csz + 12, 0) - (csz + 12) + 12).contiguous()
    d_img = torch.zeros(int((csz + 12).sum().item()) + 4096, dtype=torch.uint8, device="cuda")
    p.pack_image(d_stage, d_img, enc, img_off)
    dec = p.DeviceBatch(p.make_blocks(img_off.cpu().numpy().astype(np.uint64) + 12, offs, e["result"].astype(np.uint32), lens, e["xxh32"]))
    d_out = torch.empty(nb * B + 64, dtype=torch.uint8, device="cuda")
    p.decode_blocks(d_img, d_out, dec)
    torch.cuda.synchronize()
    assert torch.equal(d_out[: nb * B], d_src)
    print(json.dumps({"blocks": nb, "csize_sum": int(csz.sum().item()), "usize_sum": nb * B}))


def counters(db):
    cur = sqlite3.connect(db).cursor()
    out = {}
    for k, c, v, n in cur.execute("select kernel_name, counter_name, sum(value), count(distinct dispatch_id) from counters_collection group by kernel_name, counter_name"):
        k = k.split("::")[1].split("(")[0] if "::" in k else k.split("(")[0][:40]
        out.setdefault(k, {})[c] = (float(v), int(n))
    return out


def main():
    ap = argparse.ArgumentParser(); ap.add_argument("--child", action="store_true"); ap.add_argument("--blocks", type=int, default=2048)
    a = ap.parse_args()
    if a.child:
        return child(a.blocks)
    res = {}
    meta = None
    with tempfile.TemporaryDirectory(dir="/tmp") as d:
        env = dict(os.environ, TMPDIR="/tmp")
        for name in ("FETCH_SIZE", "WRITE_SIZE"):
            r = subprocess.run(["rocprofv3", "--pmc", name, "-d", os.path.join(d, name), "-o", name, "--", sys.executable, os.path.abspath(__file__), "--child", "--blocks", str(a.blocks)],
                               capture_output=True, text=True, env=env, cwd="/tmp", timeout=600)
            for l in r.stdout.splitlines():
                if l.startswith("{\"blocks\""): meta = json.loads(l)
            dbs = [os.path.join(dp, f) for dp, _, fs in os.walk(os.path.join(d, name)) for f in fs if f.endswith("_results.db")]
            if not dbs:
                raise RuntimeError("rocprofv3 left no result database: " + r.stderr[-300:])
            for k, v in counters(dbs[0]).items():
                res.setdefault(k, {}).update(v)
    KiB = 1024.0
    def get(kern, c):
        for k, v in res.items():
            if k.startswith(kern) and c in v: return v[c][0] * KiB          # one dispatch per kernel in the child
        return None
    cs, us, nb = meta["csize_sum"], meta["usize_sum"], meta["blocks"]
    # every kernel of the decode path that ran (a path is several launches: the tile / segment-parallel ones are walk + executor + the exact
    # walker's tail; kernels of a path that was launched but left at once move nothing and add nothing): their counters are summed
    dec_kernels = sorted(k for k in res if (k.startswith("lz4_seg_") or k.startswith("lz4_tile_") or k.startswith("lz4_decode_")) and (res[k].get("WRITE_SIZE", (0, 0))[0] + res[k].get("FETCH_SIZE", (0, 0))[0]) > 0)
    dec_kernel = "+".join(dec_kernels)
    def get_dec(c):
        return sum(res[k][c][0] for k in dec_kernels if c in res[k]) * KiB
    # the hash kernel runs twice (over the staging slots after the encode, over the image before the decode): per dispatch
    xx = [k for k in res if k.startswith("xxh32")]
    xf = sum(res[k]["FETCH_SIZE"][0] for k in xx) * KiB / max(sum(res[k]["FETCH_SIZE"][1] for k in xx), 1)
    out = {"blocks": nb, "source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (two passes) around one encode / hash / pack / verify / decode of the launch size, inside this bench run",
           "lz4_encode": {"fetch": get("lz4_encode_fast_kernel", "FETCH_SIZE"), "write": get("lz4_encode_fast_kernel", "WRITE_SIZE"), "algorithmic": us + cs},
           "lz4_decode": {"kernel": dec_kernel, "fetch": get_dec("FETCH_SIZE"), "write": get_dec("WRITE_SIZE"), "algorithmic": us + cs,
                          "per_kernel": {k: {c: res[k][c][0] * KiB for c in ("FETCH_SIZE", "WRITE_SIZE") if c in res[k]} for k in dec_kernels}},
           "calibration": {"xxh32_fetch_reported_over_known": round(xf / cs, 4), "pack_fetch_reported_over_known": round(get("pack_image_kernel", "FETCH_SIZE") / cs, 4),
                           "pack_write_reported_over_known": round(get("pack_image_kernel", "WRITE_SIZE") / (cs + 12 * nb), 4),
                           "decode_write_reported_over_known": round(get_dec("WRITE_SIZE") / us, 4),
                           "note": "known = the bytes the kernel has to move exactly once (xxh32: every payload byte read; pack: payloads read, payloads + 12 B headers written; decode: 4 MiB written per block - the segment-parallel path also writes and reads back 8 bytes per sequence of records (2.4 MB per block on the S-mix), the tile path one bit per stream byte, which this ratio then includes)"}}
    for k in ("lz4_encode", "lz4_decode"):
        out[k]["traffic"] = int(out[k]["fetch"] + out[k]["write"])
        out[k]["traffic_over_algorithmic"] = round(out[k]["traffic"] / out[k]["algorithmic"], 4)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
