"""CPU: the exactness condition behind csrc/hf_kernels.h FastDiv -- the scalar multiply-high division the batched kernels decode their unit
index with: u // d == (u * ceil(2^32 / d)) >> 32 whenever u * d < 2^32 (what the launchers check with fastdiv_exact before they pick
a kernel that relies on it)."""
from hypothesis import given, settings, strategies as st


def magic(d):
    return ((1 << 32) + d - 1) // d if d > 1 else 0


def fastdiv(u, d):
    m = magic(d)
    return (u * m) >> 32 if m else u


@settings(max_examples=3000, deadline=None)
@given(st.integers(1, 1 << 20), st.integers(0, (1 << 32) - 1))
def test_fastdiv_is_exact_below_the_bound(d, u):
    if u * d < (1 << 32):
        assert fastdiv(u, d) == u // d


def test_fastdiv_on_the_shapes_the_kernels_use():
    # blocks per member of the staged warp at 2160p HDR with plane blocks, 16 members; tiles of the chain kernels at 480 x 270, 32 pairs
    for d, max_u in ((4080, 4080 * 16 + 8), (120, 4080), (30, 120), (135, 135 * 4 * 32), (15, 135), (8 * 17, 8 * 17 * 32)):
        assert max_u * d < (1 << 32)
        for u in list(range(0, min(max_u, 70000))) + [max_u - 1, max_u]:
            assert fastdiv(u, d) == u // d, (u, d)
