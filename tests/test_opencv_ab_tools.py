"""tools/opencv_ab/ is the route by which a maintainer WITH OpenCV pins the oracle (the oracle is 'parity unpinned'
here). This only checks that the exporter still runs against the current oracle and that the comparer accepts a data
directory without OpenCV outputs; the OpenCV side (dump_opencv.cpp) cannot be built here."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_export_and_compare_round_trip(tmp_path):
    out = str(tmp_path / "ab")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "opencv_ab", "export_inputs.py"), out],
                       capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    man = json.load(open(os.path.join(out, "manifest.json")))
    kinds = {c["kind"] for c in man["cases"]}
    assert kinds == {"pc", "optdft", "lp", "srseq", "resize_quarter", "resize_2x", "gray", "undistort", "homography"}
    lines = open(os.path.join(out, "manifest.txt")).read().strip().splitlines()
    assert len(lines) == len(man["cases"])
    for c in man["cases"]:
        for v in c.values():
            if isinstance(v, dict) and "file" in v:
                assert os.path.getsize(os.path.join(out, v["file"])) > 0
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "opencv_ab", "compare_with_oracle.py"), out],
                       capture_output=True, text=True, timeout=120)
    assert r.returncode == 0 and "0 mismatching" in r.stdout
    src = open(os.path.join(ROOT, "tools", "opencv_ab", "dump_opencv.cpp")).read()
    for call in ("cv::phaseCorrelate", "cv::logPolar", "cvLogPolar", "cv::resize", "cv::cvtColor", "cv::undistortPoints",
                 "cv::findHomography", "cv::decomposeHomographyMat", "cv::remap"):
        assert call in src
