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
    body = re.search(r"void order_unimodal5\(kkey \(&c\)\[KNN\]\) \{(.*?)\n\}", src, re.S).group(1)
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
