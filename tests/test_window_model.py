"""CPU: the design assumption behind the staged period warp's LDS budget (csrc/hf_kernels.hip kWgChunksPerWave), checked on the bench's
synthetic scene with the oracle's flow: a 128 x 32 workgroup tile needs at most 768 16-byte chunks per source window for (almost) every
tile whose runs stay clear of the frame edge -- so the edge tiles (about 10 % of the workgroups, staged through mirror-extended windows
since round 4), not oversize windows, were what sent workgroups to the global path."""
import warp_window_model as m


def test_windows_of_the_bench_scene_fit_the_lds_budget():
    w = m.tile_table(m.flows_of(1), 128, 32)
    interior = w[:, 4] == 1
    assert 0.85 < interior.mean() < 0.93                      # 2 of 30 tile columns + the top / bottom block rows touch the mirror zone
    fits = m.policy_fixed(w, 768)
    assert fits.sum() >= 0.995 * interior.sum()               # interior tiles practically always fit 768 chunks ...
    assert m.policy_fixed(w, 512).sum() < 0.2 * interior.sum()   # ... and practically never the bare tile size: motion needs the margin


def test_windows_of_hostile_content_mostly_do_not_fit():
    """The other end (the fit share must be stated, not assumed, for content that is not benign).  Every 16 x 16 block with its own
    motion up to +-96 px: 29 % of the interior tiles fit their windows into 768 chunks, the rest takes the global path (device counters of the
    same scene inside the pipeline: 0.28 of all workgroups staged, bench.py content legs); a hard cut: 9 % (device: 0.06); a pure 64-px pan
    fits like the bench scene."""
    w = m.tile_table(m.flows_of(1, "chaotic"), 128, 32)
    interior = w[:, 4] == 1
    assert 0.2 < m.policy_fixed(w, 768).sum() / interior.sum() < 0.4
    w = m.tile_table(m.flows_of(1, "cut"), 128, 32)
    assert m.policy_fixed(w, 768).sum() < 0.15 * (w[:, 4] == 1).sum()
    w = m.tile_table(m.flows_of(1, "pan64"), 128, 32)
    assert m.policy_fixed(w, 768).sum() >= 0.995 * (w[:, 4] == 1).sum()
