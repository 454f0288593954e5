"""Shared test helpers: golden fixtures (outputs of the reference itself, see tests/golden/make_golden.py)."""
import hashlib
import json
import os

import numpy as np

from hopperrender_amd import synth

GOLDEN_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


class Golden:
    def __init__(self, name):
        self.z = np.load(os.path.join(GOLDEN_DIR, name + ".npz"))
        self.meta = json.loads(str(self.z["meta"]))
        self.case = self.meta["meta"]
        self.keys = [k for k in self.meta if k.startswith("R")]

    def frames(self):
        c = self.case
        sc = synth.Scene(c["H"], c["W"], bool(c["hdr"]), c["seed"], in_stride=c["si"])
        content = c.get("content", "motion")
        if content == "identical":
            f = sc.frame(0)
            fr = [f, f.copy(), f.copy(), f.copy()]
        elif content == "cut":
            other = synth.Scene(c["H"], c["W"], bool(c["hdr"]), c["seed"] + 999, in_stride=c["si"])
            fr = [sc.frame(0), sc.frame(1), other.frame(2), other.frame(3)]
        else:
            fr = [sc.frame(k) for k in range(4)]
        assert [sha(f) for f in fr] == self.meta["inputs_sha"], "synthetic generator drifted from the fixtures"
        return fr

    @staticmethod
    def params(key):
        r, d, n = key.split("_")
        return int(r[1:]), int(d[1:]), int(n[1:])

    def arr(self, key, name):
        return self.z[f"{key}/{name}"]

    def has(self, key, name):
        return f"{key}/{name}" in self.z.files

    def frame_names(self, key):
        return list(self.meta[key]["sha"].keys())

    def frame_sha(self, key, name):
        return self.meta[key]["sha"][name]


ALL_GOLDEN = ["sdr_180p", "hdr_180p", "sdr_360p", "hdr_360p", "sdr_ragged_strided", "hdr_ragged_strided",
              "sdr_722p_rs2", "sdr_identical", "sdr_scenecut", "sdr_1080p", "hdr_2160p", "sdr_widegrid_rs0"]


def parse_frame_name(name):
    """'warp_m2_t0.3996' -> ('warp', 2, 0.3996, (0,255)); 'copy_lv16_235' -> ('copy', None, None, (16,235))."""
    parts = name.split("_")
    kind = parts[0]
    mode, t, lv = None, None, (0.0, 255.0)
    for i, p in enumerate(parts[1:], 1):
        if p.startswith("m") and p[1:].isdigit():
            mode = int(p[1:])
        elif p.startswith("t"):
            t = float(p[1:])
        elif p.startswith("lv"):
            lv = (float(p[2:]), float(parts[i + 1]))
    return kind, mode, t, lv
