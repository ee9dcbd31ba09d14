"""``python -m nomad_amd --mode dir --nmr P --deg P`` - CLI of the reference
(/root/reference/src/nomad_audio/__main__.py:4-18); ``--device`` is honoured here (the reference
parses it and ignores it)."""
import argparse

from .nomad import Nomad


def main():
    ap = argparse.ArgumentParser(prog="nomad_amd")
    ap.add_argument("--mode", type=str, default="dir", help="Choose mode dir or csv")
    ap.add_argument("--nmr", "--nmr_path", dest="nmr", type=str, help="Path to non-matching reference files")
    ap.add_argument("--deg", "--test_path", dest="deg", type=str, help="Path to test files")
    ap.add_argument("--results_path", type=str, default=None)
    ap.add_argument("--device", type=str, default=None)
    ap.add_argument("--weights", type=str, default=None, help="checkpoint path, or 'seeded'")
    ap.add_argument("--precision", type=str, default="fp32", choices=("fp32", "bf16x3", "bf16"),
                    help="bf16x3: split-operand bf16 MFMA, scores within ~1e-6 of fp32 at over twice the speed; "
                         "bf16: fastest, for long recordings (scores within ~5e-4 of fp32)")
    a = ap.parse_args()
    nomad_avg, _ = Nomad(device=a.device, weights=a.weights, precision=a.precision).predict(a.mode, a.nmr, a.deg, a.results_path)
    print("Nomad average scores, printing top 5 test files")
    print(nomad_avg.head())


if __name__ == "__main__":
    main()
