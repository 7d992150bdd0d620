"""Derives and checks the 5-comparator network lv_match.hip::order_unimodal5 uses after the bitonic halving of two sorted top-5
lists: c[i] = min(a[i], b[4-i]) is unimodal (ascending, then descending), its threshold images are 0^p 1^m 0^q, and a
comparator network sorts a sequence iff it sorts all its threshold images.  Exhaustive search over all networks of up to 5
comparators finds none with 4 and (0,4)(1,3)(1,4)(2,4)(3,4) with 5; 200 000 random merges (ties included) confirm it."""
import itertools
import random

pats = sorted({tuple([0] * p + [1] * m + [0] * (5 - p - m)) for p in range(6) for m in range(6 - p)})
pairs = [(i, j) for i in range(5) for j in range(i + 1, 5)]


def apply(net, v):
    v = list(v)
    for i, j in net:
        if v[i] > v[j]:
            v[i], v[j] = v[j], v[i]
    return v


def sorts_all(net):
    return all(apply(net, p) == sorted(p) for p in pats)


for n in range(1, 6):
    found = next((net for net in itertools.product(pairs, repeat=n) if sorts_all(net)), None)
    print(n, "comparators:", found)
NET = ((0, 4), (1, 3), (1, 4), (2, 4), (3, 4))
assert sorts_all(NET)
bad = 0
for _ in range(200000):
    a = sorted(random.choice([random.random(), random.randint(0, 5)]) for _ in range(5))
    b = sorted(random.choice([random.random(), random.randint(0, 5)]) for _ in range(5))
    bad += apply(NET, [min(a[i], b[4 - i]) for i in range(5)]) != sorted(a + b)[:5]
print("random merges with a wrong result:", bad)
