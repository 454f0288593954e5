"""CPU: the exactness condition behind csrc/hf_kernels.h FastDiv -- the scalar multiply-high division the batched kernels decode their unit
index with: u // d == (u * ceil(2^32 / d)) >> 32 whenever u * d < 2^32.  Every launcher builds its dividers with make_fastdiv(d, max_u):
beyond that bound the divider carries magic == 0 and the kernel divides plainly (an 8K frame at rs = 0 has 64,800 tiles of 16 x 32:
a batch of two is already past the bound).  Restated here the way csrc/hf_kernels.h writes it."""
from hypothesis import given, settings, strategies as st


def magic(d):
    return ((1 << 32) + d - 1) // d if d > 1 else 0


def fastdiv(u, d):
    m = magic(d)
    return (u * m) >> 32 if m else u


def make_fastdiv(d, max_u):          # csrc/hf_kernels.h make_fastdiv: (d, magic), magic == 0 => plain division in the kernel
    return (d, magic(d) if d > 1 and max_u * d < (1 << 32) else 0)


def fastdiv_checked(u, f):
    d, m = f
    return (u * m) >> 32 if m else (u // d if d > 1 else u)


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


def test_dividers_beyond_the_bound_fall_back_to_plain_division():
    # 8K at rs = 0: 240 x 270 tiles of 32 x 16 -> 64,800 per pair; two pairs and more are beyond u * d < 2^32
    tiles = 64800
    for n in (1, 2, 5, 32):
        f = make_fastdiv(tiles, tiles * n + 8)
        assert (f[1] != 0) == ((tiles * n + 8) * tiles < (1 << 32))
        for u in (0, 1, tiles - 1, tiles, tiles * n - 1, tiles * n + 7, 66277 if n > 1 else 5):
            assert fastdiv_checked(u, f) == u // tiles
    # the unchecked magic really is wrong there (what ADVICE r4 flagged): the test would be vacuous otherwise
    bad = [u for u in range(tiles * 2 - 70000, tiles * 2) if (u * magic(tiles)) >> 32 != u // tiles]
    assert bad
