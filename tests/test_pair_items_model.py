"""CPU model of the texel-pair item lists of the production matchers (cost_volume_fast64.hip / cost_volume_fast.hip, round 5).

Not a test of the kernels (those are the -m gpu parity tests) but of the ALGORITHM they implement with ballots and `v_mbcnt`: given the
quad keys of a wave's lanes in lane order, the gate bits and a travel mode, every gate-open lane must find the two texel pairs of its
own quad in the slots it is sent to — for any key sequence, also ones that contradict the travel mode (then fewer pairs are shared,
never a wrong one), and every slot must be written exactly once."""
import numpy as np
import pytest

CLOSED = 0x40000000          # FKEY_CLOSED (cv_fast_common.hpp): +- a step of the padded map is no valid key


def pair_items(keys, gate, rowm, neg, Wp, reverse_down):
    """reverse_down=True: the batched-view kernel's numbering (slots along increasing coordinate: a view travelling down is numbered
    backwards, the combine is direction-free); False: the per-view kernel's (travel order; the combine mirrors its weight).
    Returns (first slot per lane, {slot: (texel 0, texel 1)}, pair count)."""
    n = len(keys)
    key = [k if g else CLOSED for k, g in zip(keys, gate)]
    prev = [CLOSED] + key[:-1]
    fresh = [g and k != p for g, k, p in zip(gate, key, prev)]                          # run leaders
    step = (Wp if rowm else 1) * (-1 if neg else 1)
    shr = [f and ((p + step) & 0xffffffff) == k for f, p, k in zip(fresh, prev, key)]  # leader re-uses the previous run's second pair
    nl, ns = np.cumsum(fresh), np.cumsum(shr)                                           # mbcnt (inclusive)
    cnt = 2 * int(nl[-1]) - int(ns[-1])
    T = 2 * (nl - 1) - ns                                                               # first pair of the lane's quad, travel order
    oJ, oM = (Wp, 1) if rowm else (1, Wp)
    slots, lane_slot = {}, []

    def put(s, v):
        assert s not in slots, "slot written twice"
        slots[s] = v
    for i in range(n):
        t = int(T[i])
        if reverse_down:
            s = 2 + (cnt - 2 - t if neg else t)
            if fresh[i] and not (shr[i] and not neg):
                put(s, (key[i], key[i] + oM))
            if fresh[i] and not (shr[i] and neg):
                put(s + 1, (key[i] + oJ, key[i] + oJ + oM))
        else:
            s = 2 + t
            first, second = (key[i] + oJ, key[i]) if neg else (key[i], key[i] + oJ)
            if fresh[i] and not shr[i]:
                put(s, (first, first + oM))
            if fresh[i]:
                put(s + 1, (second, second + oM))
        lane_slot.append(s if gate[i] else 0)
    assert sorted(slots) == list(range(2, 2 + cnt))
    return lane_slot, slots, cnt


def _walk(rng, kind, rowm, neg, Wp, n=64):
    x, y, keys = int(rng.integers(5, 40)), int(rng.integers(5, 40)), []
    for i in range(n):
        r = rng.random()
        sgn = -1 if neg else 1
        if kind == 0:                       # consistent with the mode, with side steps along the minor axis
            if r < 0.4:
                x, y = (x, y + sgn) if rowm else (x + sgn, y)
            elif r < 0.5:
                d = int(rng.integers(-1, 2))
                x, y = (x + d, y) if rowm else (x, y + d)
        elif kind == 1:                     # against the mode's direction
            if r < 0.4:
                x, y = (x, y - sgn) if rowm else (x - sgn, y)
        elif kind == 2:                     # random jumps
            x += int(rng.integers(-2, 3)); y += int(rng.integers(-2, 3))
        else:                               # oscillation across a texel boundary
            if r < 0.5:
                x += 1 if i % 2 else -1
        x, y = min(max(x, 0), Wp - 2), min(max(y, 0), 45)
        keys.append(y * Wp + x)
    return keys


@pytest.mark.parametrize("reverse_down", [True, False])
def test_every_open_lane_finds_its_quad(reverse_down):
    rng = np.random.default_rng(0)
    Wp, shared = 50, 0
    for _ in range(4000):
        rowm, neg = bool(rng.integers(2)), bool(rng.integers(2))
        keys = _walk(rng, int(rng.integers(4)), rowm, neg, Wp)
        gate = list(rng.random(64) < rng.choice([0.3, 0.8, 1.0]))
        ls, slots, cnt = pair_items(keys, gate, rowm, neg, Wp, reverse_down)
        oJ, oM = (Wp, 1) if rowm else (1, Wp)
        runs = sum(1 for i in range(64) if gate[i] and (i == 0 or not gate[i - 1] or keys[i - 1] != keys[i]))
        shared += 2 * runs - cnt
        for i in range(64):
            if not gate[i]:
                continue
            lo, hi = (keys[i], keys[i] + oM), (keys[i] + oJ, keys[i] + oJ + oM)
            a, b = slots[ls[i]], slots[ls[i] + 1]
            if reverse_down or not neg:
                assert (a, b) == (lo, hi)                   # coordinate order: the combine interpolates a -> b with the fraction
            else:
                assert (a, b) == (hi, lo)                   # travel order, going down: the kernel mirrors the fraction
    assert shared > 0                                       # the walks consistent with their mode do share pairs


def test_closed_key_is_never_a_neighbour():
    """A closed previous lane must not look like the neighbouring quad: FKEY_CLOSED +- step is outside the 24-bit key range."""
    for step in (1, -1, 1218, -1218, (1 << 24) - 1):
        assert not (0 <= ((CLOSED + step) & 0xffffffff) < (1 << 24))
    ls, slots, cnt = pair_items([0, 1, 1], [False, True, True], False, False, 50, True)
    assert cnt == 2 and ls[1] == ls[2] == 2
