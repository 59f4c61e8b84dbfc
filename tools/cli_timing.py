#!/usr/bin/env python3
"""Wall clock of the drop-in CLI next to the reference CLI (oracle/_ref/4mc_ref, built from the reference's sources) on the
same file, both directions, files compared.  usage: python tools/cli_timing.py [--gib 8] [--dir /dev/shm] [--zstd]
Prints one JSON object.  The process start (HIP runtime + device, ~0.1-0.2 s) is inside the GPU CLI's wall clock; it is
also reported alone (`gpu_cli_one_block_s`: the CLI compressing one 4 MiB block - process start, HIP runtime, device, one launch)."""
import argparse, hashlib, json, os, subprocess, sys, tempfile, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import helpers, importlib
import numpy as np

def sha(path):
    h = hashlib.sha256()
    with open(path, "rb") as f:
        while True:
            b = f.read(64 << 20)
            if not b: break
            h.update(b)
    return h.hexdigest()

def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gib", type=float, default=8.0); ap.add_argument("--dir", default="/dev/shm"); ap.add_argument("--zstd", action="store_true")
    ap.add_argument("--modes", default="default,mapped,streaming", help="default = compression mapped, decompression streamed")
    a = ap.parse_args()
    p = importlib.import_module("4mc_amd")
    B = 4 << 20; nblk = int(a.gib * 2**30) // B
    base = helpers.corpus(48 * B)
    out = {"file_GiB": nblk * B / 2**30, "dir": a.dir, "format": "4mz -1" if a.zstd else "4mc -1"}
    flags = ["-z"] if a.zstd else []
    with tempfile.TemporaryDirectory(dir=a.dir) as d:
        src = os.path.join(d, "in.bin")
        with open(src, "wb") as f:
            for k in range(0, nblk, 48):
                f.write(base[: min(48, nblk - k) * B].tobytes())
        one = os.path.join(d, "one"); open(one, "wb").write(base[:B].tobytes())
        t0 = time.perf_counter(); subprocess.run([p.cli_path(), "-f", one, os.path.join(d, "one.4mc")], capture_output=True); out["gpu_cli_one_block_s"] = round(time.perf_counter() - t0, 3)
        runs = [("reference_cli", helpers.ref_cli(), {})]
        for m in a.modes.split(","):
            runs.append(("gpu_cli" if m == "default" else "gpu_cli_" + m, p.cli_path(), {} if m == "default" else {"FOURMC_MMAP": "1" if m == "mapped" else "0"}))
        shas = {}
        for name, exe, env in runs:
            if not exe or not os.path.exists(exe):
                continue
            c = os.path.join(d, name + ".4mc"); back = os.path.join(d, name + ".back")
            e = dict(os.environ, **env)
            t0 = time.perf_counter(); r1 = subprocess.run([exe, *flags, "-f", src, c], capture_output=True, env=e); t1 = time.perf_counter()
            r2 = subprocess.run([exe, *flags, "-d", "-f", c, back], capture_output=True, env=e); t2 = time.perf_counter()
            ok = r1.returncode == 0 and r2.returncode == 0 and os.path.getsize(back) == nblk * B and subprocess.run(["cmp", "-s", back, src]).returncode == 0
            if name == "reference_cli": keep = os.path.join(d, "reference.4mc"); os.replace(c, keep); c = keep; shas[name] = "ref"
            else: shas[name] = "ref" if ("reference_cli" in shas and subprocess.run(["cmp", "-s", c, os.path.join(d, "reference.4mc")]).returncode == 0) else "differs"
            out[name] = {"compress_GBps": round(nblk * B / (t1 - t0) / 1e9, 3), "decompress_GBps": round(nblk * B / (t2 - t1) / 1e9, 3),
                         "compress_s": round(t1 - t0, 3), "decompress_s": round(t2 - t1, 3), "round_trip_ok": ok, "file_bytes": os.path.getsize(c) if os.path.exists(c) else None}
            if not ok: out[name]["stderr"] = (r1.stderr[-200:] + r2.stderr[-200:]).decode(errors="replace")
            os.remove(back)
            if name != "reference_cli": os.remove(c)
        if "reference_cli" in shas:
            out["files_identical_to_reference"] = {k: v == shas["reference_cli"] for k, v in shas.items() if k != "reference_cli"}
            for k in shas:
                if k != "reference_cli" and k in out:
                    out[k]["vs_reference"] = {"compress": round(out[k]["compress_GBps"] / out["reference_cli"]["compress_GBps"], 2),
                                              "decompress": round(out[k]["decompress_GBps"] / out["reference_cli"]["decompress_GBps"], 2)}
    print(json.dumps(out))

if __name__ == "__main__":
    main()
