"""GPU + real weights: the reference's only published golden values - the README 2 x 4 example table
(/root/reference/README.md:69-81) - reproduced through Nomad.predict on the shipped example wavs.

Needs the real ``nomad_best_model.pt`` (``./pt-models/`` or ``$NOMAD_CHECKPOINT``), which the reference downloads
at import time and which is not available offline: skipped (parity against the README stays UNPINNED, see
DESIGN.md section 2) until a checkpoint is present.  Values are rounded to 3 decimals by the reference, so the
tolerance is 5e-4 + the 1e-4 north-star budget."""
import os

import pytest

from conftest import GOLD

pytestmark = pytest.mark.gpu

README_SCORES = {   # README.md:76-81, by file name (column order in the CSV follows os.listdir)
    "445-123860-0012_NOISE_15": {"MJ60_10": 1.627, "FL67_01": 1.534, "FI53_04": 1.629, "MJ57_01": 1.561},
    "6563-285357-0042_OPUS_64k": {"MJ60_10": 0.23, "FL67_01": 0.414, "FI53_04": 0.186, "MJ57_01": 0.346},
}
README_MEAN = {"445-123860-0012_NOISE_15": 1.587, "6563-285357-0042_OPUS_64k": 0.294}   # README.md:69-74


def test_readme_example_table(built_lib, tmp_path):
    from nomad_amd.weights import find_checkpoint
    if find_checkpoint() is None:
        pytest.skip("nomad_best_model.pt not available offline: README parity unpinned")
    from nomad_amd.nomad import Nomad
    avg, dm = Nomad().predict("dir", os.path.join(GOLD, "wavs", "nmr-data"), os.path.join(GOLD, "wavs", "test-data"),
                              results_path=str(tmp_path))
    for deg, row in README_SCORES.items():
        assert abs(avg.loc[deg, "NOMAD"] - README_MEAN[deg]) <= 1.1e-3
        for ref, val in row.items():
            assert abs(dm.loc[deg, ref] - val) <= 1.1e-3, (deg, ref, dm.loc[deg, ref], val)
