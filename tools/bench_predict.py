#!/usr/bin/env python3
"""End-to-end nomad.predict('dir', ...) on a directory of wav files (the reference's own usage, config C1 at scale):
file decode + host packing + H2D + embedding + distances + DataFrame/CSV writing, wall clock.
Usage: python tools/bench_predict.py [--deg 512] [--ref 64]"""
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
    ap.add_argument("--deg", type=int, default=512)
    ap.add_argument("--ref", type=int, default=64)
    ap.add_argument("--precision", default="fp32", choices=("fp32", "bf16x3", "bf16"))
    ap.add_argument("--sr", type=int, default=16000, help="sample rate of the files (other than 16000: the front end resamples)")
    a = ap.parse_args()
    from nomad_amd.nomad import Nomad
    rng = np.random.RandomState(0)
    with tempfile.TemporaryDirectory() as d:
        os.makedirs(d + "/nmr"); os.makedirs(d + "/deg"); os.makedirs(d + "/out0"); os.makedirs(d + "/out1")
        secs = 0.0
        for sub, n in (("nmr", a.ref), ("deg", a.deg)):
            for i in range(n):
                length = int(rng.uniform(1.0, 8.0) * a.sr)
                secs += length / a.sr
                write_wav(f"{d}/{sub}/f{i:05d}.wav", 0.1 * rng.randn(length), a.sr)
        nmd = Nomad(weights="seeded", precision=a.precision)
        nmd.predict("dir", d + "/nmr", d + "/deg", results_path=d + "/out0")  # warm-up (first-touch, allocator)
        torch.cuda.synchronize()
        import resource
        rss0 = resource.getrusage(resource.RUSAGE_SELF).ru_maxrss
        t0 = time.perf_counter()
        avg, mat = nmd.predict("dir", d + "/nmr", d + "/deg", results_path=d + "/out1")
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        rss1 = resource.getrusage(resource.RUSAGE_SELF).ru_maxrss
        # the stages of one predict call, timed separately on the same data
        ts = time.perf_counter()
        e_ref = nmd.get_embeddings(d + "/nmr")
        e_deg = nmd.get_embeddings(d + "/deg")
        torch.cuda.synchronize()
        t_embed_stage = time.perf_counter() - ts
        ts = time.perf_counter()
        from nomad_amd.nomad import _write_rounded_csv
        _write_rounded_csv(mat.reset_index(), d + "/out0/scores_again.csv")
        t_csv = time.perf_counter() - ts
        # where the time goes: decode alone, embedding alone (device-resident inputs)
        import glob
        paths = sorted(glob.glob(d + "/deg/*.wav")) + sorted(glob.glob(d + "/nmr/*.wav"))
        t1 = time.perf_counter()
        waves = [nmd.load_processing(p) for p in paths]
        t_dec = time.perf_counter() - t1
        dev = [w[0].cuda() for w in waves]
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        for i in range(0, len(dev), 256):
            nmd.engine.embed_ragged(dev[i:i + 256], precision=a.precision)
        torch.cuda.synchronize()
        t_emb = time.perf_counter() - t2
    n = a.deg + a.ref
    print(json.dumps({"precision": a.precision, "file_sample_rate": a.sr, "files": n, "audio_s": round(secs, 1), "predict_s": round(dt, 3), "files_per_s": round(n / dt, 1),
                      "audio_s_per_s": round(secs / dt, 1), "decode_only_s": round(t_dec, 3), "embed_only_s": round(t_emb, 3),
                      "get_embeddings_pipeline_s": round(t_embed_stage, 3), "scores_csv_s": round(t_csv, 3),
                      "audio_bytes_all_files_MB": round(secs * 16000 * 4 / 1e6, 1),
                      "peak_rss_MB_before_after": [round(rss0 / 1024, 1), round(rss1 / 1024, 1)],
                      "score_shape": list(mat.shape)}))


if __name__ == "__main__":
    main()
