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

One 32-bit hash per sample yields both coordinates: u from its high 16 bits, v
from its low 16 bits.  Every value is a 16-bit integer scaled by 2**-16, so it is
exact in float32 and float64 alike (and so is cell index + value for N < 256).
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


def stream_key(seed: int, pixel: int, node: int, light: int) -> int:
    """Key of the (pixel, node, light) sample grid."""
    k = mix32((seed + GOLDEN) & M32)
    k = mix32(k ^ (pixel & M32))
    k = mix32(k ^ ((node * 0x9E3779B1) & M32))
    k = mix32(k ^ (((light + 1) * 0x85EBCA6B) & M32))
    return k


def sample_bits(key: int, s: int):
    """The two 16-bit integers (u, v) of sample ``s = i*N+j`` of a grid."""
    m = mix32((key + s * GOLDEN) & M32)
    return m >> 16, m & 0xFFFF


def uniform(seed: int, pixel: int, node: int, light: int, draw: int) -> float:
    """Value handed to the reference in place of ``np.random.random()``."""
    u, v = sample_bits(stream_key(seed, pixel, node, light), draw >> 1)
    return (v if (draw & 1) else u) * (1.0 / 65536.0)
