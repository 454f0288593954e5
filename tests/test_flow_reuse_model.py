"""CPU: the model of the chain's SAD reuse decision (tests/flow_reuse_model.py, on the oracle's step functions) -- the shares DESIGN.md
quotes for the bench scene and for the hostile content classes."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))


def _shares(scene):
    from flow_reuse_model import reuse_shares, scene_frames
    from oracle import oracle
    g = oracle.make_geom(0, 1080, 1920)
    f1, f2 = scene_frames(scene, 1080, 1920, False)
    return {(ws, ax): s for ws, ax, s in reuse_shares(f1, f2, g)}


def test_reuse_share_on_the_bench_scene():
    sh = _shares("bench")
    assert sh[(32, 1)] == 0.0                       # the tables start at level 32: nothing to reuse there
    assert sh[(16, 0)] > 0.9 and sh[(8, 0)] > 0.88 and sh[(4, 0)] > 0.88 and sh[(2, 0)] > 0.8
    assert 0.7 < sh[(2, 1)] < 0.8, sh               # the last Y step: its own X step must have chosen 0 too


def test_reuse_share_static_and_hostile():
    assert all(s == 1.0 for (ws, ax), s in _shares("static").items() if ws <= 16)
    assert all(s < 0.2 for (ws, ax), s in _shares("chaotic").items() if ws <= 16)
    assert all(s < 0.1 for (ws, ax), s in _shares("cut").items() if ws <= 16)
