"""CPU model of the chain's cross-step SAD reuse (hf_flow.hip "SAD tables"), on the oracle's step functions.

The per-pixel SAD of candidate cz at a step depends only on the pixel, the window's (ox, oy) before the step and the axis
(calcDeltaSumsKernelSDR.h:61-101); bias terms are per-window constants.  A window of step s (axis a, level k) therefore sees exactly the
samples of step s - 2 (same axis, level k - 1) when the two steps in between -- s - 2 itself and s - 1 -- both chose d = 0 for the parent
window:   X step of level k:  X and Y of level k - 1 chose 0 for the parent;   Y step of level k:  Y of level k - 1 chose 0 for the parent
and X of level k chose 0 for the window itself.  The HIP chain keeps the 16 candidate SADs of every 2 x 2 grid block from the last step that
computed them and sums those instead of gathering the phase plane again.  This model walks the oracle's chain step by step and reports,
per step, the share of grid pixels (inside 32 x 32 tiles that lie fully in the grid: the tiles the HIP kernels treat that way) whose
window may reuse; tests/test_flow_reuse_model.py pins the numbers for the bench scene and checks the directed cases.

Run as a script for the table: python tests/flow_reuse_model.py [--hdr 1 --H 2160 --W 3840 --scene bench]
"""
import argparse
import ctypes as C
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import oracle  # noqa: E402  (test infrastructure)


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


def chain_steps(f1, f2, g, R=16, delta=8, nb=6, iterations=0):
    """Yields (level k, axis, window, offsets_before[2,lh,lw], offsets_after) for every step of the oracle's chain."""
    L = oracle.lib()
    ws = L.hfo_initial_window(g.lw, g.lh)
    iters = L.hfo_iterations(ws, iterations)
    off = np.zeros((2, g.lh, g.lw), dtype=np.int16)
    lowest = np.zeros((g.lh, g.lw), dtype=np.uint8)
    for k in range(iters):
        for axis in (0, 1):
            before = off.copy()
            sums, _ = oracle.calc_delta_sums(f1, f2, off, g, ws, R, k, axis, delta, nb)
            sums = np.ascontiguousarray(sums)
            L.hfo_determine_lowest_layer(_p(sums), _p(lowest), ws, R, g.lh, g.lw)
            L.hfo_adjust_offsets(_p(off), _p(lowest), ws, R, g.lh, g.lw, axis)
            yield k, axis, ws, before, off.copy()
        ws = max(ws >> 1, 1)


def reuse_shares(f1, f2, g, R=16, delta=8, nb=6, first_table_window=32):
    """[(window, axis, share of full-tile pixels whose window reuses)] for every step; steps that cannot reuse (no table yet) give 0."""
    lw, lh = g.lw, g.lh
    full = np.zeros((lh, lw), dtype=bool)
    full[: (lh // 32) * 32, : (lw // 32) * 32] = True
    n_full = max(int(full.sum()), 1)
    hist = []   # (k, axis, ws, before, after)
    out = []
    for k, axis, ws, before, after in chain_steps(f1, f2, g, R, delta, nb):
        share = 0.0
        if ws < first_table_window and R == 16:
            same = lambda st: (st[3][st[1]] == st[4][st[1]])          # the step chose d = 0 (per pixel = per window)
            prev_same_axis_other = [h for h in hist if h[0] == k - 1]  # X and Y of level k - 1
            if len(prev_same_axis_other) == 2:
                px, py = prev_same_axis_other
                if axis == 0:
                    ok = same(px) & same(py)
                else:
                    own_x = [h for h in hist if h[0] == k and h[1] == 0][0]
                    ok = same(py) & same(own_x)
                share = float((ok & full).sum()) / n_full
        hist.append((k, axis, ws, before, after))
        out.append((ws, axis, share))
    return out


def scene_frames(name, H, W, hdr, seed=1234):
    """The bench's content classes (bench.py --scene): frames N-1, N of one pair."""
    from hopperrender_amd import synth
    return synth.scene_pair(name, H, W, hdr, seed)


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--hdr", type=int, default=0); ap.add_argument("--H", type=int, default=1080); ap.add_argument("--W", type=int, default=1920)
    ap.add_argument("--scene", default="bench"); ap.add_argument("--seed", type=int, default=1234)
    a = ap.parse_args()
    g = oracle.make_geom(a.hdr, a.H, a.W)
    f1, f2 = scene_frames(a.scene, a.H, a.W, bool(a.hdr), a.seed)
    for ws, axis, share in reuse_shares(f1, f2, g):
        print(f"window {ws:3d} {'XY'[axis]}: reusable {share:.3f}")
