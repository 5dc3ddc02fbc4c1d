"""The keyed jitter stream: Python statement == C oracle == library host function == golden table."""
import random

from helpers import golden_json
from oracle import jitter


def test_python_statement_matches_golden_table():
    for row in golden_json("kat/jitter_kat.json"):
        u, v = jitter.sample_bits(jitter.stream_key(row["seed"], row["pixel"], row["node"], row["light"]), row["n"], row["s"])
        assert (u, v) == (row["u"], row["v"])
        assert 0 <= u < 256 and 0 <= v < 256


def test_uniform_is_8_bit_fraction():
    x = jitter.uniform(3, 1000, 5, 2, 5, 7)
    assert 0.0 <= x < 1.0 and (x * 256.0).is_integer()
    u, v = jitter.sample_bits(jitter.stream_key(3, 1000, 5, 2), 5, 3)
    assert jitter.uniform(3, 1000, 5, 2, 5, 6) == u / 256.0   # even draw = u
    assert jitter.uniform(3, 1000, 5, 2, 5, 7) == v / 256.0   # odd draw = v


def test_two_neighbours_of_a_row_share_one_table_word_scrambled_by_the_key():
    key = jitter.stream_key(9, 4242, 3, 1)
    for n in (1, 2, 5, 16):
        for i in range(n):
            for q in range((n + 1) // 2):
                m = jitter.pair_word(i * ((n + 1) // 2) + q) ^ key
                assert jitter.sample_bits(key, n, i * n + 2 * q) == (m >> 24, (m >> 16) & 255)
                if 2 * q + 1 < n:
                    assert jitter.sample_bits(key, n, i * n + 2 * q + 1) == ((m >> 8) & 255, m & 255)


def test_oracle_c_matches_python(oracle):
    rnd = random.Random(5)
    for _ in range(2000):
        n = rnd.randrange(1, 33)
        a = (rnd.getrandbits(32), rnd.getrandbits(23), rnd.getrandbits(11), rnd.randrange(8), n, rnd.randrange(n * n))
        assert oracle.jitter_bits(*a) == jitter.sample_bits(jitter.stream_key(*a[:4]), a[4], a[5])


def test_library_host_function_matches_python():
    from raytracer_amd import runtime
    rnd = random.Random(6)
    for _ in range(2000):
        n = rnd.randrange(1, 33)
        a = (rnd.getrandbits(32), rnd.getrandbits(23), rnd.getrandbits(11), rnd.randrange(8), n, rnd.randrange(n * n))
        assert runtime.jitter_bits(*a) == jitter.sample_bits(jitter.stream_key(*a[:4]), a[4], a[5])


def test_stream_is_roughly_uniform():
    n = 20000
    us = [jitter.uniform(1, p, 1, 0, 4, 0) for p in range(n // 2)] + [jitter.uniform(1, p, 1, 0, 4, 2) for p in range(n // 2)]
    vs = [jitter.uniform(1, p, 1, 0, 4, 1) for p in range(n // 2)] + [jitter.uniform(1, p, 1, 0, 4, 3) for p in range(n // 2)]
    mu, mv = sum(us) / n, sum(vs) / n
    assert abs(mu - 0.5) < 0.01 and abs(mv - 0.5) < 0.01
    cov = sum((a - mu) * (b - mv) for a, b in zip(us, vs)) / n
    assert abs(cov) < 0.002  # u and v of one sample are uncorrelated (var of U[0,1) is 1/12)


def _pearson(a, b):
    n = len(a)
    ma, mb = sum(a) / n, sum(b) / n
    va = sum((x - ma) ** 2 for x in a)
    vb = sum((y - mb) ** 2 for y in b)
    return sum((x - ma) * (y - mb) for x, y in zip(a, b)) / (va * vb) ** 0.5


def test_neighbouring_pixels_have_uncorrelated_jitter():
    """All grids share one table of pair words and differ by the XOR of their keys: the jitter of the SAME sample in
    neighbouring pixels (right-hand and lower neighbour of a 3840-wide frame), in two lights of one pixel and in the
    two children of one node must not be correlated."""
    width, n = 3840, 16
    rnd = random.Random(11)
    cases = {"right": [], "below": [], "light": [], "node": []}
    for _ in range(6000):
        pixel = rnd.randrange(width * 2159 - 1)
        s = rnd.randrange(n * n)
        u0, v0 = jitter.sample_bits(jitter.stream_key(0, pixel, 1, 0), n, s)
        for name, key in (("right", jitter.stream_key(0, pixel + 1, 1, 0)), ("below", jitter.stream_key(0, pixel + width, 1, 0)),
                          ("light", jitter.stream_key(0, pixel, 1, 1)), ("node", jitter.stream_key(0, pixel, 2, 0))):
            u1, v1 = jitter.sample_bits(key, n, s)
            cases[name].append((u0, v0, u1, v1))
    for name, rows in cases.items():
        u0, v0, u1, v1 = zip(*rows)
        assert abs(_pearson(u0, u1)) < 0.04 and abs(_pearson(v0, v1)) < 0.04 and abs(_pearson(u0, v1)) < 0.04, name


def test_one_grid_is_uniform():
    """Within (pixel, node, light) grids: every byte uniform over the samples of many grids (chi-square against 16
    bins) and no correlation between u and v of a sample."""
    n = 16
    rnd = random.Random(12)
    us, vs = [], []
    for _ in range(150):
        key = jitter.stream_key(rnd.getrandbits(32), rnd.getrandbits(23), 1 + rnd.randrange(30), rnd.randrange(5))
        grid = [jitter.sample_bits(key, n, s) for s in range(n * n)]
        us += [g[0] for g in grid]
        vs += [g[1] for g in grid]
    for vals in (us, vs):
        bins = [0] * 16
        for x in vals:
            bins[x >> 4] += 1
        expect = len(vals) / 16.0
        chi2 = sum((b - expect) ** 2 / expect for b in bins)
        assert chi2 < 45.0, chi2        # 15 degrees of freedom: 45 is far in the tail (p < 1e-4)
    assert abs(_pearson(us, vs)) < 0.02


def test_the_table_of_pair_words_is_well_mixed():
    """The structure every grid shares is the table itself (a grid is the table XOR its key), so the table is what has
    to be free of structure: uniform bytes, no correlation between the four bytes of a word (u / v of the two samples
    of a pair), between neighbouring pairs of a row, or between the pairs above each other (16 x 16 grid: 8 words per
    row).  4096 words: the standard deviation of a sample correlation is 0.016."""
    words = [jitter.pair_word(p) for p in range(4096)]
    byte = [[(w >> (8 * b)) & 255 for w in words] for b in range(4)]
    for b in range(4):
        bins = [0] * 16
        for x in byte[b]:
            bins[x >> 4] += 1
        chi2 = sum((c - 256.0) ** 2 / 256.0 for c in bins)
        assert chi2 < 45.0, (b, chi2)
        for c in range(b + 1, 4):
            assert abs(_pearson(byte[b], byte[c])) < 0.06, (b, c)
        for lag in (1, 8):
            assert abs(_pearson(byte[b][:-lag], byte[b][lag:])) < 0.06, (b, lag)
    assert len(set(words)) == len(words)
