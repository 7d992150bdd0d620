"""The 5-comparator network pass_kernel's kNN merge uses after the bitonic halving of two sorted top-5 lists
(limo-velo_amd/csrc/lv_match.hip: order_unimodal5): min(a[i], b[4-i]) is a unimodal sequence, whose threshold images are
0^p 1^m 0^q — a comparator network sorts a class of sequences iff it sorts all their 0-1 threshold images.  Host logic only."""
import itertools
import random
import re
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent


def _network_in_the_kernel():
    src = (ROOT / "limo-velo_amd" / "csrc" / "lv_match.hip").read_text()
    body = re.search(r"void order_unimodal5\(kkey \(&c\)\[5\]\) \{(.*?)\n\}", src, re.S).group(1)
    return [(int(a), int(b)) for a, b in re.findall(r"cswap\(c\[(\d)\], c\[(\d)\]\)", body)]


def _apply(net, v):
    v = list(v)
    for i, j in net:
        if v[i] > v[j]:
            v[i], v[j] = v[j], v[i]
    return v


def test_the_kernels_network_sorts_every_unimodal_sequence_of_five():
    net = _network_in_the_kernel()
    assert len(net) == 5 and all(i < j for i, j in net)
    images = {tuple([0] * p + [1] * m + [0] * (5 - p - m)) for p in range(6) for m in range(6 - p)}
    assert all(_apply(net, im) == sorted(im) for im in images)
    # no network of four comparators does (so five is minimal)
    pairs = [(i, j) for i in range(5) for j in range(i + 1, 5)]
    assert not any(all(_apply(n4, im) == sorted(im) for im in images) for n4 in itertools.product(pairs, repeat=4))


def test_merge_of_two_sorted_top5_lists_with_ties():
    net = _network_in_the_kernel()
    rng = random.Random(5)
    for _ in range(50_000):
        a = sorted(rng.choice([rng.random(), rng.randint(0, 4)]) for _ in range(5))
        b = sorted(rng.choice([rng.random(), rng.randint(0, 4)]) for _ in range(8))   # (the chunk side: 8 sorted, its 5 smallest are used)
        merged = _apply(net, [min(a[i], b[4 - i]) for i in range(5)])
        assert merged == sorted(a + b)[:5]


def test_lds_bitonic_network_steps_and_the_wave_local_claim():
    """lv_ldssort.hpp: pair t of step (k2, j = 1 << lj) is lo = ((t >> lj) << (lj + 1)) | (t & (j - 1)), hi = lo | j, ascending iff
    (lo & k2) == 0; steps whose j and whose successor's j are <= 64 are separated by a wavefront fence only, which is sound iff
    the 64 pairs of a wavefront (t in [64 w, 64 w + 64)) stay inside elements [128 w, 128 w + 128) in both steps."""
    rng = random.Random(11)
    for length in (64, 128, 1024, 2048, 4096):   # (4096: two pairs per thread of a 1024-thread workgroup)
        keys = [rng.randrange(1 << 20) for _ in range(length)]
        want = sorted(keys)
        steps = []
        k2, lk = 2, 1
        while k2 <= length:
            for lj in range(lk - 1, -1, -1):
                steps.append((k2, lj))
            k2, lk = k2 << 1, lk + 1
        for si, (k2, lj) in enumerate(steps):
            j = 1 << lj
            touched = {}
            for t in range(length // 2):
                lo = ((t >> lj) << (lj + 1)) | (t & (j - 1))
                hi = lo | j
                assert hi == lo + j and hi < length
                up = (lo & k2) == 0
                if (keys[lo] > keys[hi]) == up:
                    keys[lo], keys[hi] = keys[hi], keys[lo]
                touched.setdefault(t // 64, set()).update((lo, hi))
            next_j = (j >> 1) if lj > 0 else k2
            last = si == len(steps) - 1
            workgroup_barrier = j > 64 or next_j > 64 or last
            if not workgroup_barrier:   # wave-local: every wavefront stayed inside its own 128 elements
                assert all(min(e) >= 128 * w and max(e) < 128 * w + 128 for w, e in touched.items()), (length, k2, j)
        assert keys == want


def test_the_eight_key_network_sorts(  ):
    """sort8 (every chunk of eight candidates; also the ordering step of the general-K merge, NUM_MATCH_POINTS 3..8:
    order_selected pads the K selected keys with NONE and sorts them with it): 0-1 principle over all 256 images."""
    src = (ROOT / "limo-velo_amd" / "csrc" / "lv_match.hip").read_text()
    body = re.search(r"void sort8\(kkey \(&c\)\[8\]\) \{(.*?)\n\}", src, re.S).group(1)
    net = [(int(a), int(b)) for a, b in re.findall(r"cswap\(c\[(\d)\], c\[(\d)\]\)", body)]
    assert len(net) == 19 and all(i < j for i, j in net)
    assert all(_apply(net, im) == sorted(im) for im in itertools.product((0, 1), repeat=8))
    # the general-K merge: the K smallest of two sorted lists = min(k[i], o[K-1-i]), then any sorter
    rng = random.Random(5)
    for K in (3, 4, 6, 7, 8):
        for _ in range(200):
            k = sorted(rng.sample(range(1000), K))
            o = sorted(rng.sample(range(1000, 2000), 8)) if rng.random() < 0.3 else sorted(rng.sample(range(1000), 8))
            sel = [min(k[i], o[K - 1 - i]) for i in range(K)]
            assert sorted(sel) == sorted(k + o)[:K]
