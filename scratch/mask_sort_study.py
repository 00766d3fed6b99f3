"""Offline: offsets a 32-row wave of K9b must process (|union of its rows' neighbour masks|) under row orders (CPU)."""
import sys, os, numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
from fullysparsefusion_amd import synthetic
f = synthetic.make_frame(10, 0)
p = f['points'][:, :3]
vs = np.array([0.2, 0.2, 0.2], np.float32); lo = np.array([-51.2, -51.2, -5.0], np.float32)
c = np.floor((p - lo) / vs).astype(np.int64)
ok = ((c >= 0) & (c < np.array([512, 512, 40]))).all(1)
c = c[ok]
key = np.unique((c[:, 2] * 512 + c[:, 1]) * 512 + c[:, 0])
z, y, x = key // (512 * 512), (key // 512) % 512, key % 512
def nbr_mask(z, y, x):
    n = len(z); base = (z * 4096 + y) * 4096 + x
    srt = np.argsort(base); bs = base[srt]
    m = np.zeros(n, np.int64); k = 0
    for dz in (-1, 0, 1):
        for dy in (-1, 0, 1):
            for dx in (-1, 0, 1):
                q = ((z + dz) * 4096 + (y + dy)) * 4096 + (x + dx)
                pos = np.searchsorted(bs, q); pos[pos >= n] = n - 1
                m |= (bs[pos] == q).astype(np.int64) << k; k += 1
    return m
def popcount(v):
    return np.array([bin(int(t)).count('1') for t in v])
def study(z, y, x, name, G=32):
    m = nbr_mask(z, y, x); n = len(m)
    print(f'--- {name}: rows {n}, pairs/out {popcount(m[:: max(1, n // 5000)]).mean():.2f}')
    def cost(order, G):
        mm = m[order]; pad = (-n) % G
        mm = np.concatenate([mm, np.zeros(pad, np.int64)]).reshape(-1, G)
        u = np.bitwise_or.reduce(mm, axis=1)
        return popcount(u).mean()
    lex = np.arange(n)
    srt = np.argsort(m, kind='stable')
    # gray-ish: sort by popcount then mask
    pc = popcount(m) if n < 200000 else None
    srt2 = np.lexsort((m, pc))
    # bit-reordered key: most frequent offsets as the most significant bits
    freq = [(int(((m >> b) & 1).sum()), b) for b in range(27)]
    order_bits = [b for _, b in sorted(freq)]  # rarest -> LSB ... most frequent -> MSB
    key2 = np.zeros(n, np.int64)
    for i, b in enumerate(order_bits): key2 |= ((m >> b) & 1) << i
    srt3 = np.argsort(key2, kind='stable')
    key3 = np.zeros(n, np.int64)
    for i, b in enumerate(reversed(order_bits)): key3 |= ((m >> b) & 1) << i   # rarest offsets most significant
    srt4 = np.argsort(key3, kind='stable')
    for G in (16, 32):
        print(f'  group {G}: lexicographic {cost(lex, G):5.2f}   mask value {cost(srt, G):5.2f}   popcount,mask {cost(srt2, G):5.2f}   freq-MSB {cost(srt3, G):5.2f}   rare-MSB {cost(srt4, G):5.2f}   (of 27)')
study(z, y, x, 'level 1 (0.2 m)')
def down(z, y, x):
    outs = []
    for dz in range(3):
        for dy in range(3):
            for dx in range(3):
                oz, oy, ox = z + 1 - dz, y + 1 - dy, x + 1 - dx
                mk = (oz % 2 == 0) & (oy % 2 == 0) & (ox % 2 == 0) & (oz >= 0) & (oy >= 0) & (ox >= 0)
                outs.append(((oz[mk] // 2) * 4096 + oy[mk] // 2) * 4096 + ox[mk] // 2)
    k = np.unique(np.concatenate(outs))
    return k // (4096 * 4096), (k // 4096) % 4096, k % 4096
z2, y2, x2 = down(z, y, x); study(z2, y2, x2, 'level 2 (0.4 m)')
z3, y3, x3 = down(z2, y2, x2); study(z3, y3, x3, 'level 3 (0.8 m)')
z4, y4, x4 = down(z3, y3, x3); study(z4, y4, x4, 'level 4 (1.6 m)')
