#!/usr/bin/env python3
"""Where the time of Nomad.get_embeddings_csv goes on a large directory: per-stage time of the probe -> convert/pack ->
launch pipeline (native header probe and row conversion, Python decodes, packing, launch, result fetch).
NOMAD_WAV_THREADS=0 gives the all-Python front end for comparison.
Usage: python tools/predict_pipeline_trace.py [--files 4000] [--precision bf16x3]"""
import argparse, json, os, struct, sys, tempfile, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch


def write_wav(path, x, sr=16000):
    pcm = (np.clip(x, -1, 1) * 32767).astype("<i2").tobytes()
    with open(path, "wb") as f:
        f.write(b"RIFF" + struct.pack("<I", 36 + len(pcm)) + b"WAVEfmt " + struct.pack("<IHHIIHH", 16, 1, 1, sr, sr * 2, 2, 16) +
                b"data" + struct.pack("<I", len(pcm)) + pcm)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--files", type=int, default=4000)
    ap.add_argument("--precision", default="bf16x3")
    a = ap.parse_args()
    import nomad_amd.nomad as NM
    import importlib
    NM = importlib.import_module("nomad_amd.nomad")
    rng = np.random.RandomState(0)
    with tempfile.TemporaryDirectory() as d:
        for i in range(a.files):
            write_wav(f"{d}/f{i:05d}.wav", 0.1 * rng.randn(int(rng.uniform(1.0, 8.0) * 16000)))
        nmd = NM.Nomad(weights="seeded", precision=a.precision)
        nmd.get_embeddings(d)     # warm-up
        torch.cuda.synchronize()
        T = {"probe": 0.0, "read_rows": 0.0, "load": 0.0, "pack": 0.0, "launch": 0.0, "fetch": 0.0, "batches": 0}
        eng = nmd.engine
        orig_load, orig_pack, orig_embed, orig_fetch = nmd.load_processing, eng.pack_ragged_host, eng.embed_ragged, eng.fetch_async

        def timed(name, fn):
            def w(*args, **kw):
                t0 = time.perf_counter()
                r = fn(*args, **kw)
                T[name] += time.perf_counter() - t0
                return r
            return w
        from nomad_amd import wavio
        wavio.probe = timed("probe", wavio.probe)                 # packer thread (native threads inside)
        wavio.read_rows = timed("read_rows", wavio.read_rows)     # packer thread (native threads inside)
        nmd.load_processing = timed("load", orig_load)            # summed over the Python decode threads
        eng.pack_ragged_host = timed("pack", orig_pack)           # packer thread
        eng.embed_ragged = timed("launch", orig_embed)            # consumer thread
        class F:
            def __init__(self, f): self.f = f
            def result(self):
                t0 = time.perf_counter(); r = self.f.result(); T["fetch"] += time.perf_counter() - t0; T["batches"] += 1; return r
        eng.fetch_async = lambda e: F(orig_fetch(e))
        t0 = time.perf_counter()
        nmd.get_embeddings(d)
        torch.cuda.synchronize()
        wall = time.perf_counter() - t0
    print(json.dumps({"files": a.files, "precision": a.precision, "wall_s": round(wall, 3), "files_per_s": round(a.files / wall, 1),
                      **{k: (round(v, 3) if isinstance(v, float) else v) for k, v in T.items()},
                      "decode_threads": nmd.DECODE_THREADS, "native_wav_threads": nmd.NATIVE_WAV_THREADS, "cpus": os.cpu_count()}))


if __name__ == "__main__":
    main()
