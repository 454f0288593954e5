"""tests/warp_window_model.py -- CPU model of the window decision of warp_wg_kernel (csrc/hf_kernels.hip warp_wg_body) on the bench scene.

For every workgroup tile of the staged period warp (2160p HDR: 128 x 32 luma pixels / 128 x 32 chroma elements) it computes,
from the blurred flow of the oracle on the bench's synthetic scene, the source windows the five outputs of a 24 -> 120 period
need (exactly the kernel's arithmetic: runs as (row, byte offset), window = 16-byte chunk columns x rows) and prints how many
workgroups stage under a given policy.  No GPU needed: this is how the LDS budget / tile shape / per-source policies are chosen
before they are built.  It takes its flow fields from the oracle, so it lives with the tests (only tests/ may use oracle/);
tests/test_window_model.py pins the design assumption it was written to check.

    python tests/warp_window_model.py [--seed 1234] [--frames 3]
"""
import argparse, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hopperrender_amd import synth
from oracle import oracle

ap = argparse.ArgumentParser()
ap.add_argument("--seed", type=int, default=1234)
ap.add_argument("--frames", type=int, default=3, help="consecutive source periods to average over")
ap.add_argument("--speed", type=int, default=48)
a = ap.parse_args([] if __name__ != "__main__" else None)

H, W, SZ, RS = 2160, 3840, 2, 3
g = oracle.make_geom(1, H, W)
lw, lh = g.lw, g.lh
_frames = {}


def scene_frames(scene="bench"):
    """Frames 0 .. of a content class (synth.ContentScene; bench = synth.Scene, the bench's default content)."""
    if scene not in _frames:
        sc = synth.Scene(H, W, True, a.seed, max_rect_speed=a.speed) if scene == "bench" else synth.ContentScene(scene, H, W, True, a.seed)
        _frames[scene] = [sc.frame(k) for k in range(a.frames + 2)]
    return _frames[scene]


TS = [[0.0, 0.1998, 0.3996, 0.5994, 0.7992], [0.199, 0.3988, 0.5986, 0.7984, 0.9982]]


def rnd(x):  # roundf: half away from zero
    return np.where(x >= 0, np.floor(x + 0.5), -np.floor(-x + 0.5)).astype(np.int64)


def windows(flow, ts, cz, tw_px, th_rows):
    """Per workgroup tile: (cols_a, rows_a, cols_b, rows_b, interior) -- window extents in 16-byte chunks and rows."""
    fx, fy = flow[0].astype(np.int64), flow[1].astype(np.int64)
    dim_y = H >> cz
    # one entry per (thread row pair, 16-byte thread): cy0 in steps of 2, cx0 in steps of 8 elements
    cy0 = np.arange(0, dim_y, 2)[:, None]
    cx0 = np.arange(0, W, 8)[None, :]
    ly = ((cy0 >> RS) << 1) if cz else (cy0 >> RS)
    lx = ((cx0 >> RS) & ~1) if cz else (cx0 >> RS)
    ly = np.broadcast_to(ly, (cy0.shape[0], cx0.shape[1])); lx = np.broadcast_to(lx, ly.shape)
    ox12, oy12 = fx[ly, lx], fy[ly, lx]
    py = np.clip(ly - (oy12 >> RS), 0, lh - 1); px = np.clip(lx - (ox12 >> RS), 0, lw - 1)
    ox21, oy21 = fx[py, px], fy[py, px]
    lo = {}; hi = {}
    for s in "ab":
        lo[s] = [np.full(ly.shape, 1 << 30), np.full(ly.shape, 1 << 30)]; hi[s] = [np.full(ly.shape, -(1 << 30)), np.full(ly.shape, -(1 << 30))]
    interior = np.ones(ly.shape, bool)
    for t in ts:
        s12, s21 = np.float32(t), np.float32(1.0) - np.float32(t)
        hy = np.float32(0.5) if cz else np.float32(1.0)
        xa = cx0 + rnd(ox12.astype(np.float32) * s12); xb = cx0 - rnd(ox21.astype(np.float32) * s21)
        ya = cy0 + rnd(oy12.astype(np.float32) * s12 * hy); yb = cy0 - rnd(oy21.astype(np.float32) * s21 * hy)
        for s, x, y in (("a", xa, ya), ("b", xb, yb)):
            xe = (x & ~1) if cz else x
            interior &= (xe >= 1) & (xe + (1 if cz else 0) + 7 <= W - 2) & (y >= 1) & (y + 1 <= dim_y - 2)
            lo[s][0] = np.minimum(lo[s][0], xe * SZ); hi[s][0] = np.maximum(hi[s][0], xe * SZ)
            lo[s][1] = np.minimum(lo[s][1], y); hi[s][1] = np.maximum(hi[s][1], y)
    # reduce over workgroup tiles
    tr, tc = th_rows // 2, tw_px // 8
    nr, nc = -(-ly.shape[0] // tr), -(-ly.shape[1] // tc)
    out = np.zeros((nr, nc, 5), np.int64)
    for r in range(nr):
        for c in range(nc):
            sl = (slice(r * tr, (r + 1) * tr), slice(c * tc, (c + 1) * tc))
            full = ly[sl].shape == (tr, tc)
            ok = full and interior[sl].all()
            v = []
            for s in "ab":
                cmin = lo[s][0][sl].min() >> 4
                C = (((hi[s][0][sl].max() & ~3) + 16 + 3) >> 4) - cmin + 1
                R = hi[s][1][sl].max() + 1 - lo[s][1][sl].min() + 1
                v += [C, R]
            out[r, c] = v + [int(ok)]
    return out


def policy_fixed(w, budget):      # today's kernel: both windows within `budget` chunks each, C <= 64
    ra = ((w[..., 0] * w[..., 1] + 63) & ~63); rb = ((w[..., 2] * w[..., 3] + 63) & ~63)
    return (w[..., 4] == 1) & (ra <= budget) & (rb <= budget) & (w[..., 0] <= 64) & (w[..., 2] <= 64)


def policy_pool(w, budget2):      # one pool shared by both windows
    ra = ((w[..., 0] * w[..., 1] + 63) & ~63); rb = ((w[..., 2] * w[..., 3] + 63) & ~63)
    return (w[..., 4] == 1) & (ra + rb <= budget2) & (w[..., 0] <= 64) & (w[..., 2] <= 64)


def policy_per_source(w, budget):  # fraction of SOURCE windows staged when each source decides for itself
    ra = ((w[..., 0] * w[..., 1] + 63) & ~63); rb = ((w[..., 2] * w[..., 3] + 63) & ~63)
    ok = w[..., 4] == 1
    return (ok & (ra <= budget)).mean() * 0.5 + (ok & (rb <= budget)).mean() * 0.5


def flows_of(n_frames, scene="bench"):
    out = []
    fr = scene_frames(scene)
    for k in range(n_frames):
        _, blur, _, _ = oracle.calculate_optical_flow(fr[k], fr[k + 1], g, 16, 0, 8, 6, 4)
        out.append(blur)
    return out


def tile_table(flows, tw, th):
    rows = []
    for cz in (0, 1):
        ws = [windows(f, TS[i % 2], cz, tw, th) for i, f in enumerate(flows)]
        rows.append(np.concatenate([w.reshape(-1, 5) for w in ws]))
    return np.concatenate(rows)           # luma blocks are 2/3 of all blocks automatically (twice as many rows)


def main():
    flows = flows_of(a.frames)
    for tw, th in ((128, 32), (128, 16), (256, 16), (64, 32), (128, 64), (256, 32)):
        w = tile_table(flows, tw, th)
        tile_chunks = tw * SZ // 16 * th
        line = f"tile {tw:3d}x{th:2d} ({tile_chunks:4d} chunks)  interior {w[:, 4].mean():.3f} "
        for mult in (1.5, 2.0, 2.5, 3.0):
            b = int(tile_chunks * mult) & ~63
            line += f"| x{mult}: fixed {policy_fixed(w, b).mean():.3f} pool {policy_pool(w, 2 * b).mean():.3f} persrc {policy_per_source(w, b):.3f} "
        print(line, flush=True)
        if (tw, th) == (128, 32):
            ok = w[:, 4] == 1
            need = np.maximum(w[:, 0] * w[:, 1], w[:, 2] * w[:, 3])[ok]
            print("   chunks needed per source window (interior tiles), percentiles 50/70/80/90/95/99:",
                  [int(np.percentile(need, p)) for p in (50, 70, 80, 90, 95, 99)], " fetched/tile at 768:",
                  round(float((w[:, 0] * w[:, 1] + w[:, 2] * w[:, 3])[policy_fixed(w, 768)].mean() / (2 * tile_chunks)), 3))


if __name__ == "__main__":
    main()
