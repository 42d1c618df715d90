"""The division-step inversion of csrc/fq28_inv.hpp, restated with range-checked Python integers (tests/model/safegcd_model.py): right
answers on edge values, every 64-bit accumulator inside its range, and the fixed 37 rounds enough for everything tried, including the
inputs with the longest known division-step chains (CPU only)."""
import random

import safegcd_model as G

P = G.P


def test_edge_values_and_random_inputs():
    rnd = random.Random(0x5AFE)
    cases = [1, 2, 3, P - 1, P - 2, (P + 1) // 2, (P - 1) // 2, 1 << 380, (1 << 381) - 1, P + 1, P + 5, 2 * P - 1, 2 * P - 2]
    cases += [rnd.randrange(1, P) for _ in range(400)] + [rnd.randrange(P, 2 * P) for _ in range(50)]
    worst = 0
    for x in cases:
        if x % P == 0:
            continue
        inv, used = G.modinv(x)
        assert inv * x % P == 1, hex(x)
        assert G.modinv(x, early_exit=False)[0] == inv          # running all 37 rounds (a lane whose neighbours need them) changes nothing
        worst = max(worst, used)
    assert worst <= G.ROUNDS
    assert G.modinv(0)[0] == 0 and G.modinv(P)[0] == 0            # 0 and p -> 0, as the device function documents


def test_slow_inputs_stay_inside_the_round_budget():
    """Inputs that keep g odd and delta oscillating for long: powers of two +-1, Fibonacci-like ratios f / g ~ golden ratio (the worst case
    of Euclid-type chains), and values near p / 2^k."""
    xs = []
    a, b = 1, 1
    while b < P:
        a, b = b, a + b
        xs.append(b % P)
    for k in range(1, 381, 7):
        xs += [(1 << k) - 1, (1 << k) + 1, P >> k, (P >> k) | 1, P - (1 << k)]
    phi = (P * 0x9E3779B97F4A7C15) >> 64                          # p / golden ratio
    xs += [phi, phi + 1, P - phi]
    worst = 0
    for x in xs:
        x %= P
        if x == 0:
            continue
        inv, used = G.modinv(x)
        assert inv * x % P == 1
        worst = max(worst, used)
    assert worst <= G.ROUNDS
    # the proven bound behind the constant (Bernstein-Yang, Theorem 11.2, delta = 1): 0 <= g < 2p
    import math
    assert (49 * 0.5 * (762 + math.log2(17 / 5)) + 57) / 17 <= 30 * G.ROUNDS
