"""Counter-based shadow-jitter stream (pure-Python statement of the contract).

TEST INFRASTRUCTURE.  The same function is implemented three times -- here
(used to drive the unmodified reference through wrappers), in
``oracle/rt_oracle.c`` and in the HIP kernels -- and the three must agree bit
for bit; ``tests/test_jitter.py`` pins them against the known-answer table in
``tests/golden/jitter_kat.json``.

The reference draws ``np.random.random()`` twice per shadow sample
(/root/reference/light.py:53-54, first the ``u`` draw, then the ``v`` draw) from
one sequential Mersenne-Twister stream.  A GPU cannot follow that order, so
both sides are driven by a stream keyed on *what* is being sampled instead of
*when*:

    key   = (seed, pixel index y*W+x, ray-tree node id, light index)
    node  = 1 for the camera ray, 2k for the reflection child of node k,
            2k+1 for its transparency child
    draw  = 2*(i*N+j) + axis   (axis 0 = u, 1 = v), i outer / j inner as in
            light.py:51-52

One 32-bit word serves TWO neighbouring samples of a row of the grid, one byte per
coordinate: samples (i, 2q) and (i, 2q+1) share the word of pair p = i*ceil(N/2) + q; the
even sample takes bytes 3 (u) and 2 (v), the odd one bytes 1 (u) and 0 (v).  The word is

    pair_word(p) ^ key        pair_word(p) = mix32(p * GOLDEN + PAIR_SALT)

i.e. ONE table of well-mixed words indexed by the pair alone, scrambled per (pixel, node,
light) grid by XOR with the grid's key (a digital shift).  XOR with a constant permutes the
256 values of a byte, so every coordinate stays uniform whatever the key; two grids differ
by the XOR of their keys, which are full mix32 outputs.  (On the GPU all lanes of a
wavefront walk the table at the same index, so the table word comes through the scalar unit
and a pair of samples costs one vector XOR instead of a hash per lane.)  Every value is an
8-bit integer scaled by 2**-8 -- 1/256 of a grid cell, i.e. 1/4096 of the light's edge at
N = 16 -- so it is exact in float32 and float64 alike, and so is cell index + value.
"""

M32 = 0xFFFFFFFF
GOLDEN = 0x9E3779B9


def mix32(x: int) -> int:
    """32-bit finaliser (xorshift-multiply, two rounds)."""
    x &= M32
    x ^= x >> 16
    x = (x * 0x7FEB352D) & M32
    x ^= x >> 15
    x = (x * 0x846CA68B) & M32
    x ^= x >> 16
    return x


PAIR_SALT = 0x5BF03635


def pair_word(pair: int) -> int:
    """Entry ``pair`` of the jitter table (independent of seed, pixel, node and light)."""
    return mix32((pair * GOLDEN + PAIR_SALT) & M32)


def stream_key(seed: int, pixel: int, node: int, light: int) -> int:
    """Key of the (pixel, node, light) sample grid."""
    k = mix32((seed + GOLDEN) & M32)
    k = mix32(k ^ (pixel & M32))
    k = mix32(k ^ ((node * 0x9E3779B1) & M32))
    k = mix32(k ^ (((light + 1) * 0x85EBCA6B) & M32))
    return k


JITTER_STEPS = 256


def sample_bits(key: int, n: int, s: int):
    """The two 8-bit integers (u, v) of sample ``s = i*n+j`` of an ``n`` x ``n`` grid."""
    i, j = divmod(s, n)
    pair = i * ((n + 1) >> 1) + (j >> 1)
    m = pair_word(pair) ^ key
    if j & 1:
        return (m >> 8) & 0xFF, m & 0xFF
    return m >> 24, (m >> 16) & 0xFF


def uniform(seed: int, pixel: int, node: int, light: int, n: int, draw: int) -> float:
    """Value handed to the reference in place of ``np.random.random()`` (``n`` = grid edge, light.py:35)."""
    u, v = sample_bits(stream_key(seed, pixel, node, light), n, draw >> 1)
    return (v if (draw & 1) else u) * (1.0 / JITTER_STEPS)
