"""The oracle against reference-generated goldens at the BASELINE sizes (1080p / 720p / 15x15 window / 4K).

tests/golden/baseline_sizes.npz holds what the reference itself (imported in the development container by
tests/golden/gen_golden.py) produced on the cfg-2 ... cfg-5 inputs: selected and tracked lists in full, sha256 of the
eigenvalue map and of every pyramid plane, the head of the sorted candidate list.  Everything is bit-exact.
"""
import os

import numpy as np
import pytest

from oracle import klt_oracle as ko
from helpers import baseline_case, golden_feats_equal, params_from_tc, sha_bytes


@pytest.fixture(scope="module")
def big(golden_dir):
    return np.load(os.path.join(golden_dir, "baseline_sizes.npz"))


@pytest.fixture(scope="module", autouse=True)
def threads():
    ko.set_threads(min(8, os.cpu_count() or 1))     # OpenMP over image lines / features: bit-identical results
    yield
    ko.set_threads(1)


@pytest.mark.parametrize("tag", ["cfg2", "cfg4", "cfg3", "cfg5"])
def test_oracle_matches_reference_at_baseline_size(big, tag):
    frames, tc, n = baseline_case(tag)
    p = params_from_tc(tc)
    assert [tc.borderx, tc.bordery] == big[tag + "_border"].tolist()
    for k, f in enumerate(frames):
        assert np.array_equal(sha_bytes(f), big["%s_frame%d_sha" % (tag, k)]), "synthetic frame %d differs" % k
    a0, a1 = frames[0].astype(np.float32), frames[1].astype(np.float32)
    nrows, ncols = a0.shape

    # eigenvalue map (goodFeaturesUtils.pyx:35-73) and candidate order (selectGoodFeatures.py:234-236)
    fl, val = ko.select_good_features(p, a0, n, want_val=True)
    assert val.size == int(big[tag + "_eig_count"][0])
    assert np.array_equal(sha_bytes(val), big[tag + "_eig_sha"]), "eigenvalue map"
    bx, by, _, _ = ko.scan_borders(p)
    c = ko.sorted_candidates(val, ncols, nrows, bx, by, 0)
    assert np.array_equal(c["val"][:4096], big[tag + "_sorted_val"])
    assert np.array_equal(c["x"][:4096], big[tag + "_sorted_x"]) and np.array_equal(c["y"][:4096], big[tag + "_sorted_y"])
    # KLTSelectGoodFeatures (selectGoodFeatures.py:279-294)
    assert int((fl["val"] > 0).sum()) == n
    assert golden_feats_equal(fl, big, tag, "sel"), "selected list"

    # ComputeImagePyramids (trackFeatures.py:146-196)
    P0, P1 = ko.Pyramids(p, a0), ko.Pyramids(p, a1)
    for k, P in enumerate((P0, P1)):
        for l in range(tc.nPyramidLevels):
            for w in ("img", "gx", "gy"):
                assert np.array_equal(sha_bytes(P.level(w, l)), big["%s_p%d_%s_%d_sha" % (tag, k, w, l)]), (k, w, l)
    # KLTTrackFeatures (trackFeatures.py:205-409)
    ko.track_features(p, P0, P1, fl)
    assert golden_feats_equal(fl, big, tag, "trk"), "tracked list"
    assert int((fl["val"] == 0).sum()) > 0.9 * n
