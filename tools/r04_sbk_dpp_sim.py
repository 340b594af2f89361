#!/usr/bin/env python
"""Host check of the DPP form of the by-key wave scan (vexcl/scan_by_key.hpp, VEXCL_SBK_DPP=1) against the shuffle form it replaces:
64 lanes, each with the running value `tail` of its own elements, a flag `head` (the lane holds a run head) and `live` (a prefix of
the lanes holds elements).  Shuffle form: for o = 1, 2, .. 32: lane L adds the value of lane L - o when L - o is not below hl(L), the
nearest head lane at or before L.  DPP form: row_shr 1 / 2 / 4 / 8 inside rows of 16 lanes, row_bcast:15 (lane 15 / 47 into the row
behind it), row_bcast:31 (lane 31 into lanes 32..63), the same rule at every step.  Integers: the sums must be identical."""
import random


def hl_of(head):
    out, cur = [], 0
    for lane, h in enumerate(head):
        if h:
            cur = lane
        out.append(cur)
    return out


def shuffle_form(tail, hl, live):
    t = list(tail)
    o = 1
    while o < 64:
        old = list(t)
        for lane in range(64):
            if lane - o >= hl[lane] and live[lane]:
                t[lane] = old[lane - o] + old[lane]
        o *= 2
    return t


def dpp_form(tail, hl, live):
    t = list(tail)
    for o in (1, 2, 4, 8):                                   # row_shr:o -- a lane whose source lies in another row keeps its own value
        old = list(t)
        for lane in range(64):
            u = old[lane - o] if lane % 16 >= o else old[lane]
            if (lane & 15) >= o and lane - o >= hl[lane] and live[lane]:
                t[lane] = u + old[lane]
    old = list(t)                                            # row_bcast:15 -- lane 15 of a row to every lane of the next row
    for lane in range(64):
        u = old[(lane // 16) * 16 - 1] if lane >= 16 else old[lane]
        if (lane & 16) and (lane & 48) - 1 >= hl[lane] and live[lane]:
            t[lane] = u + old[lane]
    old = list(t)                                            # row_bcast:31 -- lane 31 to rows 2 and 3
    for lane in range(64):
        u = old[31] if lane >= 32 else old[lane]
        if lane >= 32 and 31 >= hl[lane] and live[lane]:
            t[lane] = u + old[lane]
    return t


def main(trials=200000, seed=1):
    rng = random.Random(seed)
    for trial in range(trials):
        p = rng.choice([0.0, 0.02, 0.1, 0.3, 0.7, 1.0])
        nl = rng.choice([64, 64, 64, rng.randint(1, 64)])
        live = [i < nl for i in range(64)]
        head = [rng.random() < p and live[i] for i in range(64)]
        tail = [rng.randint(-1000, 1000) for _ in range(64)]
        hl = hl_of(head)
        a, b = shuffle_form(tail, hl, live), dpp_form(tail, hl, live)
        for lane in range(64):
            assert not live[lane] or a[lane] == b[lane], (trial, lane, hl[lane], a[lane], b[lane])
    print("%d patterns: the two forms agree on every live lane" % trials)


if __name__ == "__main__":
    import sys
    main(int(sys.argv[1]) if len(sys.argv) > 1 else 200000)
