"""The keyed jitter stream: Python statement == C oracle == library host function == golden table."""
import random

from helpers import golden_json
from oracle import jitter


def test_python_statement_matches_golden_table():
    for row in golden_json("kat/jitter_kat.json"):
        u, v = jitter.sample_bits(jitter.stream_key(row["seed"], row["pixel"], row["node"], row["light"]), row["s"])
        assert (u, v) == (row["u"], row["v"])
        assert 0 <= u < 1 << 16 and 0 <= v < 1 << 16


def test_uniform_is_16_bit_fraction():
    x = jitter.uniform(3, 1000, 5, 2, 7)
    assert 0.0 <= x < 1.0 and (x * 65536.0).is_integer()
    u, v = jitter.sample_bits(jitter.stream_key(3, 1000, 5, 2), 3)
    assert jitter.uniform(3, 1000, 5, 2, 6) == u / 65536.0   # even draw = u
    assert jitter.uniform(3, 1000, 5, 2, 7) == v / 65536.0   # odd draw = v


def test_oracle_c_matches_python(oracle):
    rnd = random.Random(5)
    for _ in range(2000):
        a = (rnd.getrandbits(32), rnd.getrandbits(23), rnd.getrandbits(11), rnd.randrange(8), rnd.randrange(1024))
        assert oracle.jitter_bits(*a) == jitter.sample_bits(jitter.stream_key(*a[:4]), a[4])


def test_library_host_function_matches_python():
    from raytracer_amd import runtime
    rnd = random.Random(6)
    for _ in range(2000):
        a = (rnd.getrandbits(32), rnd.getrandbits(23), rnd.getrandbits(11), rnd.randrange(8), rnd.randrange(1024))
        assert runtime.jitter_bits(*a) == jitter.sample_bits(jitter.stream_key(*a[:4]), a[4])


def test_stream_is_roughly_uniform():
    n = 20000
    us = [jitter.uniform(1, p, 1, 0, 0) for p in range(n)]
    vs = [jitter.uniform(1, p, 1, 0, 1) for p in range(n)]
    mu, mv = sum(us) / n, sum(vs) / n
    assert abs(mu - 0.5) < 0.01 and abs(mv - 0.5) < 0.01
    cov = sum((a - mu) * (b - mv) for a, b in zip(us, vs)) / n
    assert abs(cov) < 0.002  # u and v of one sample are uncorrelated (var of U[0,1) is 1/12)
