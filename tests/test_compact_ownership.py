"""k_compact's ownership rule (k3_own, csrc/qmvt_kernels.hip), restated and model-checked on the host.

A wave of k_compact owns the chunks of a list (2^CL entries) that BEGIN among the entries of its four tiles, completes the
last of them from the ONE tile behind its own, and stores what the owner in front of it could not reach.  Every wave derives
the range it stores from tile offsets alone -- its own, its neighbours', the tiles behind them -- so the rule must be a
PARTITION of every list, whatever it holds: dense, sparse, empty waves, a ragged first chunk, a list that stops half way.
The GPU tests check the kernel against the oracle (tests/test_gpu_parity.py::test_the_compaction_on_lists_of_every_density);
this restatement of the rule's arithmetic guards the rule itself (its first version lost the ragged first chunk of a list
whose first waves hold no entry: found here, before it ran on a GPU)."""
import random

K3_TILES = 4   # tiles per wave (csrc/qmvt_kernels.hip K3_TILES)


def k3_own(CL, a, a2, aprev, rnext, reach, l0, l1):
    """[lo, hi) of a wave whose tiles hold the entries [a, a2) of a list that spans [l0, l1); aprev: first entry of the wave in
    front; rnext: the list's offset behind this wave's FIRST tile (= the reach of the wave in front); reach: behind the tile
    behind its last one.  The same arithmetic as the kernel's."""
    C = 1 << CL
    M = C - 1
    s = a & ~M
    up, up2 = (a + M) & ~M, (a2 + M) & ~M
    head = a == l0
    begins = up < a2 or (head and a2 > a)
    sb = max(s, l0)
    inherited = (not head) and s != a and aprev <= sb < a
    lo = min(s + C, rnext) if inherited else a
    hi = min(up2, reach, l1) if begins else a2
    return lo, hi


def _waves(off, l1, ntiles):
    for t in range(0, ntiles, K3_TILES):
        a = off[t]
        a2 = off[t + K3_TILES] if t + K3_TILES < ntiles else l1
        aprev = 0 if t == 0 else off[t - K3_TILES]
        rnext = off[t + 1] if t + 1 < ntiles else l1
        reach = off[t + K3_TILES + 1] if t + K3_TILES + 1 < ntiles else l1
        yield t, a, a2, aprev, rnext, reach


def test_every_entry_leaves_exactly_once_and_within_reach():
    rng = random.Random(20260501)
    for trial in range(20000):
        CL = rng.choice([2, 3, 6, 8])                       # (the kernel: 64-entry TP chunks, 256-entry FP chunks)
        ntiles = rng.randint(1, 40)
        dens = rng.choice([0, 0.001, 0.01, 0.1, 1, 5, 50, 300, 1024])
        cnt = [min(1024, int(rng.expovariate(1 / dens))) if dens > 0 and rng.random() < 0.9 else 0 for _ in range(ntiles)]
        if rng.random() < 0.2:
            for i in range(ntiles // 2, ntiles):             # a list that stops: the chunk at its end is never whole
                cnt[i] = 0
        l0 = rng.choice([0, 0, 5, 13, 256, 1000])            # the FP list starts wherever n - fp_total says
        off = [l0]
        for c in cnt:
            off.append(off[-1] + c)
        l1 = off[-1]
        stored = [0] * (l1 + 1)
        for t, a, a2, aprev, rnext, reach in _waves(off, l1, ntiles):
            lo, hi = k3_own(CL, a, a2, aprev, rnext, reach, l0, l1)
            if hi > lo:
                assert lo >= a                                # nothing in front of its own tiles
                assert hi <= off[min(t + K3_TILES + 1, ntiles)]   # nothing beyond the one tile it reads behind its own
                for e in range(lo, hi):
                    stored[e] += 1
        assert all(stored[e] == 1 for e in range(l0, l1)), (CL, cnt, l0)


def test_dense_lists_leave_in_whole_chunks():
    """what the rule is for: on a dense list every wave but the first and the last stores whole chunks only"""
    ntiles, per = 64, 900                                    # 900 of 1 024 records per tile on the list
    off = [0]
    for _ in range(ntiles):
        off.append(off[-1] + per)
    l1 = off[-1]
    for t, a, a2, aprev, rnext, reach in _waves(off, l1, ntiles):
        lo, hi = k3_own(8, a, a2, aprev, rnext, reach, 0, l1)
        if t > 0:
            assert lo % 256 == 0
        if t + K3_TILES < ntiles:
            assert hi % 256 == 0
