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


def test_two_neighbours_of_a_row_share_one_hash():
    key = jitter.stream_key(9, 4242, 3, 1)
    for n in (1, 2, 5, 16):
        for i in range(n):
            for q in range((n + 1) // 2):
                m = jitter.pair_hash(key + (i * ((n + 1) // 2) + q) * jitter.GOLDEN)
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
