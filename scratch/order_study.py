"""Offline: row-block utilisation + tile-neighbour locality of the spconv kernel under different voxel row orders (CPU)."""
import sys, os, numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
from fullysparsefusion_amd import synthetic
f = synthetic.make_frame(10, 0)
p = f['points'][:, :3]
vs = np.array([0.2, 0.2, 0.2], np.float32); lo = np.array([-51.2, -51.2, -5.0], np.float32)
c = np.floor((p - lo) / vs).astype(np.int64)
ok = ((c >= 0) & (c < np.array([512, 512, 40]))).all(1)
c = c[ok]
key = (c[:, 2] * 512 + c[:, 1]) * 512 + c[:, 0]
key = np.unique(key)
z, y, x = key // (512 * 512), (key // 512) % 512, key % 512
print('level-1 voxels', len(key))

def part1by2(v):
    v = v.astype(np.int64) & 0x3ff
    v = (v | (v << 16)) & 0x30000ff
    v = (v | (v << 8)) & 0x300f00f
    v = (v | (v << 4)) & 0x30c30c3
    v = (v | (v << 2)) & 0x9249249
    return v
def part1by1(v):
    v = v.astype(np.int64) & 0xffff
    v = (v | (v << 8)) & 0x00ff00ff
    v = (v | (v << 4)) & 0x0f0f0f0f
    v = (v | (v << 2)) & 0x33333333
    v = (v | (v << 1)) & 0x55555555
    return v

def orders(z, y, x):
    o = {}
    o['lex zyx'] = (z * 4096 + y) * 4096 + x
    o['morton3'] = part1by2(x) | (part1by2(y) << 1) | (part1by2(z) << 2)
    o['z, morton2(y,x)'] = (z << 32) | part1by1(x) | (part1by1(y) << 1)
    o['morton2(y,x), z'] = ((part1by1(x) | (part1by1(y) << 1)) << 8) | z
    for b in (4, 8, 16):
        o[f'yx blk{b}, z, y, x'] = ((((y // b) * 4096 + (x // b)) * 64 + z) * b + (y % b)) * b + (x % b)
        o[f'z, yx blk{b}'] = (((z * 4096 + (y // b)) * 4096 + (x // b)) * b + (y % b)) * b + (x % b)
    o['yx blk8, z/2 ...'] = (((((y // 8) * 4096 + (x // 8)) * 64 + z // 2) * 8 + (y % 8)) * 8 + (x % 8)) * 2 + z % 2
    return o

def study(z, y, x, name, TM=64):
    n = len(z)
    base = (z * 4096 + y) * 4096 + x
    srt = np.argsort(base); bs = base[srt]
    offs = [(dz, dy, dx) for dz in (-1, 0, 1) for dy in (-1, 0, 1) for dx in (-1, 0, 1)]
    nbr = np.full((n, 27), -1, np.int64)
    for k, (dz, dy, dx) in enumerate(offs):
        q = ((z + dz) * 4096 + (y + dy)) * 4096 + (x + dx)
        pos = np.searchsorted(bs, q); pos[pos >= n] = n - 1
        hit = bs[pos] == q
        nbr[hit, k] = srt[pos[hit]]
    print(f'--- {name}: rows {n} pairs/out {np.mean((nbr >= 0).sum(1)):.2f}')
    for oname, okey in orders(z, y, x).items():
        perm = np.argsort(okey, kind='stable')      # new row r = old row perm[r]
        inv = np.empty(n, np.int64); inv[perm] = np.arange(n)
        nb = nbr[perm]
        nb = np.where(nb >= 0, inv[np.maximum(nb, 0)], -1)
        ntile = (n + TM - 1) // TM
        pad = ntile * TM - n
        nbp = np.vstack([nb, np.full((pad, 27), -1, np.int64)]).reshape(ntile, TM, 27)
        cnt = (nbp >= 0).sum(1)                       # [tile, 27]
        blocks = ((cnt + 15) // 16).sum()
        util = cnt.sum() / (blocks * 16)
        stages = (cnt > 0).sum() / ntile
        # locality: distinct 64-row input tiles touched per output tile (gather footprint)
        t_in = np.where(nbp >= 0, nbp // TM, -1).reshape(ntile, -1)
        foot = np.mean([len(np.unique(r[r >= 0])) for r in t_in[:: max(1, ntile // 400)]])
        # in-tile fraction: neighbours that live in the same tile
        same = ((t_in == np.arange(ntile)[:, None]) & (t_in >= 0)).sum() / (t_in >= 0).sum()
        print(f'  {oname:22s} util {util:.3f}  active offsets/tile {stages:5.2f}  in-tile nbr frac {same:.3f}  input tiles touched {foot:6.1f}')

study(z, y, x, 'level-1 SubM (0.2 m)')
# level 2: stride-2 downsample k3 p1 (out = floor((in + 1 - k)/2) for k in 0..2) -> any out with an in inside its window
def down(z, y, x):
    outs = []
    for dz in range(3):
        for dy in range(3):
            for dx in range(3):
                oz, oy, ox = z + 1 - dz, y + 1 - dy, x + 1 - dx
                m = (oz % 2 == 0) & (oy % 2 == 0) & (ox % 2 == 0) & (oz >= 0) & (oy >= 0) & (ox >= 0)
                outs.append(((oz[m] // 2) * 4096 + oy[m] // 2) * 4096 + ox[m] // 2)
    k = np.unique(np.concatenate(outs))
    return k // (4096 * 4096), (k // 4096) % 4096, k % 4096
z2, y2, x2 = down(z, y, x); study(z2, y2, x2, 'level-2 SubM (0.4 m)')
z3, y3, x3 = down(z2, y2, x2); study(z3, y3, x3, 'level-3 SubM (0.8 m)')
