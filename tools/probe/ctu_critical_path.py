#!/usr/bin/env python3
"""Critical path of a picture's CTU wavefront from the requests each CTU made (XMI_EXP_CTU_REQUESTS lines on stderr):
T(y, x) = max(T(y, x - 1), T(y - 1, x + 1)) + r(y, x). Prints, per picture type, the mean path length in requests beside the
mean number of requests of a CTU and what the path would be if every CTU took one request."""
import sys
from collections import defaultdict

acc = defaultdict(list)
for line in sys.stdin:
    if not line.startswith("CTU_REQUESTS"):
        continue
    f = line.split()
    kind, cols = f[1], int(f[2])
    r = list(map(int, f[3:]))
    rows = len(r) // cols
    def path(cost):
        T = [[0] * cols for _ in range(rows)]
        for y in range(rows):
            for x in range(cols):
                a = T[y][x - 1] if x else 0
                b = T[y - 1][min(x + 1, cols - 1)] if y else 0
                T[y][x] = max(a, b) + cost(r[y * cols + x])
        return T[-1][-1]
    hist = defaultdict(int)
    for v in r:
        hist[min(v, 12)] += 1
    acc[kind].append((path(lambda v: v), sum(r) / len(r), path(lambda v: 1), path(lambda v: min(v, 2)), path(lambda v: (v + 1) // 2), dict(sorted(hist.items()))))
for kind, v in acc.items():
    n = len(v)
    print(kind, "pictures", n, "path %.0f" % (sum(a[0] for a in v) / n), "requests per CTU %.2f" % (sum(a[1] for a in v) / n),
          "path at one request per CTU %.0f" % (sum(a[2] for a in v) / n), "path with requests capped at 2: %.0f" % (sum(a[3] for a in v) / n),
          "path with requests halved: %.0f" % (sum(a[4] for a in v) / n))
    print("   requests per CTU, histogram of the last picture (12 = 12 and more):", v[-1][5])
