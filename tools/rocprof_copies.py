import sqlite3, sys, collections
cur = sqlite3.connect(sys.argv[1]).cursor()
rows = list(cur.execute("select name, start, end from kernels order by start"))
adam = [i for i, r in enumerate(rows) if "adam_kernel" in r[0]]
rows = rows[adam[-2] + 1: adam[-1] + 1]
t0 = rows[0][1]
prev = collections.Counter(); nxt = collections.Counter(); n = 0; tot = 0
pos = []
for i, (name, s, e) in enumerate(rows):
    if "copyBuffer" in name:
        n += 1; tot += e - s
        j = i - 1
        while j >= 0 and "copyBuffer" in rows[j][0]: j -= 1
        prev[rows[j][0][:60] if j >= 0 else "start"] += 1
        pos.append((s - t0) / 1e6)
print(n, "copies", tot / 1e3, "us total")
for k, v in prev.most_common(12): print(v, k)
import numpy as np
h = np.histogram(pos, bins=13, range=(0, 130))
print(list(h[0]))
