"""CPU restatements of the refine-stage native ops (TEST INFRASTRUCTURE — only tests/, smoke() and bench.py's
cpu_baseline may import this; the product path never does).

PARITY UNPINNED for both: the upstream kernels live in un-vendored packages (TorchEx dynamic_point_pool_ext; mmdet3d 0.x
iou3d) and the reference tree holds no golden vectors for them.  What pins them is (1) the published algorithm
restated here, (2) the call sites and the 13-float layout asserted at
projects/mmdet3d_plugin/models/roi_heads/roi_extractors/dynamic_point_roi_extractor.py:78-92, (3) for NMS an
independent float64 polygon-intersection (intersection points + contained corners, angular sort — the iou3d way) that
the HIP clipper has to agree with.
"""
import numpy as np


# ------------------------------------------------------------------------------------- K17 dynamic point pool
def dynamic_point_pool(rois, pts, extra_wlh, max_inbox_point, max_all_pts=50000, return_margin=False, near_tol=None,
                       stop_at_cap=False):
    """rois [R,7] (cx,cy,cz_bottom,w,l,h,rz), pts [P,3] -> (pts_idx i64 [k], roi_idx i64 [k], feats f32 [k,13]) in
    ascending (roi, point) order, first `max_inbox_point` per roi, first `max_all_pts` overall
    (dynamic_point_pool_op.py:10-51 semantics with the atomics replaced by a canonical order).
    return_margin: also the distance of every kept (point, roi) decision from the nearest decision boundary, and the
    same for the closest rejected pairs — lets tests skip pairs that sit within rounding of a box face.
    near_tol: keep only the margin rows closer than this to a face (the full table is [R * P, 3]: 79 GB at 10.6 k RoIs x
    3.1e5 points).  stop_at_cap: leave the RoI loop once `max_all_pts` pairs are collected — the rows a later RoI would add
    are cut by the final `[:max_all_pts]` anyway, so the result is the same (at the 10-sweep frame's 10.6 k RoIs the cap
    is reached after a few hundred)."""
    rois = np.asarray(rois, dtype=np.float32)
    pts = np.asarray(pts, dtype=np.float32)
    ew, el, eh = [np.float32(v) for v in extra_wlh]
    out_p, out_r, out_f, near = [], [], [], []
    half = np.float32(0.5)
    total = 0
    for r in range(rois.shape[0]):
        if stop_at_cap and total >= max_all_pts:
            break
        cx, cy, czb, w, l, h, rz = rois[r, :7]
        cz = czb + h * half
        rot = np.float32(rz + np.float32(np.pi / 2))
        cosa, sina = np.float32(np.cos(rot)), np.float32(np.sin(rot))
        dx, dy = pts[:, 0] - cx, pts[:, 1] - cy
        lz = pts[:, 2] - cz
        lx = dx * cosa + dy * (-sina)
        ly = dx * sina + dy * cosa
        hl, hw, hh = l * half, w * half, h * half
        lhl, lhw, lhh = (l + el) * half, (w + ew) * half, (h + eh) * half
        in_large = (np.abs(lz) <= lhh) & (lx > -lhl) & (lx < lhl) & (ly > -lhw) & (ly < lhw)
        in_box = in_large & (lx > -hl) & (lx < hl) & (ly > -hw) & (ly < hw) & (np.abs(lz) <= hh)
        if return_margin:
            d_large = np.minimum.reduce([lhh - np.abs(lz), lhl - np.abs(lx), lhw - np.abs(ly)])
            d_box = np.minimum.reduce([hh - np.abs(lz), hl - np.abs(lx), hw - np.abs(ly)])
            mg = np.minimum(np.abs(d_large), np.abs(d_box))
            sel = np.arange(pts.shape[0]) if near_tol is None else np.nonzero(mg < near_tol)[0]
            near.append(np.stack([np.full(sel.shape[0], r, dtype=np.float64), sel.astype(np.float64), mg[sel].astype(np.float64)], 1))
        idx = np.nonzero(in_large)[0][:max_inbox_point]
        feats = np.stack([pts[idx, 0], pts[idx, 1], pts[idx, 2], lx[idx], ly[idx], lz[idx],
                          lx[idx] + hl, ly[idx] + hw, lz[idx] + hh, hl - lx[idx], hw - ly[idx], hh - lz[idx],
                          (~in_box[idx]).astype(np.float32)], 1).astype(np.float32)
        total += idx.shape[0]
        out_p.append(idx)
        out_r.append(np.full(idx.shape[0], r, dtype=np.int64))
        out_f.append(feats)
    if out_p:
        p, r_, f = np.concatenate(out_p)[:max_all_pts], np.concatenate(out_r)[:max_all_pts], np.concatenate(out_f)[:max_all_pts]
    else:
        p, r_, f = np.zeros(0, np.int64), np.zeros(0, np.int64), np.zeros((0, 13), np.float32)
    if return_margin:
        return p.astype(np.int64), r_, f, (np.concatenate(near) if near else np.zeros((0, 3)))
    return p.astype(np.int64), r_, f


# --------------------------------------------------------------------------------------------- K20 BEV NMS
def _corners(box):
    x1, y1, x2, y2, ang = [float(v) for v in box]
    cx, cy = (x1 + x2) / 2, (y1 + y2) / 2
    c, s = np.cos(ang), np.sin(ang)
    pts = []
    for px, py in ((x1, y1), (x2, y1), (x2, y2), (x1, y2)):
        dx, dy = px - cx, py - cy
        pts.append((dx * c + dy * s + cx, -dx * s + dy * c + cy))  # iou3d rotate_around_center
    return np.array(pts)


def _inside(poly, p, eps=1e-9):
    """p inside the convex polygon `poly` (either orientation)."""
    sign = 0
    for i in range(len(poly)):
        a, b = poly[i], poly[(i + 1) % len(poly)]
        cr = (b[0] - a[0]) * (p[1] - a[1]) - (b[1] - a[1]) * (p[0] - a[0])
        if abs(cr) < eps:
            continue
        s = 1 if cr > 0 else -1
        if sign == 0:
            sign = s
        elif s != sign:
            return False
    return True


def _seg_intersection(p0, p1, q0, q1):
    d1, d2 = p1 - p0, q1 - q0
    den = d1[0] * d2[1] - d1[1] * d2[0]
    if abs(den) < 1e-14:
        return None
    t = ((q0[0] - p0[0]) * d2[1] - (q0[1] - p0[1]) * d2[0]) / den
    u = ((q0[0] - p0[0]) * d1[1] - (q0[1] - p0[1]) * d1[0]) / den
    if 0.0 <= t <= 1.0 and 0.0 <= u <= 1.0:
        return p0 + t * d1
    return None


def rotated_overlap(a, b):
    """Intersection area of two rotated rectangles, float64: edge intersections + contained corners, sorted by angle
    around their centroid, shoelace (the iou3d box_overlap recipe)."""
    pa, pb = _corners(a), _corners(b)
    cand = []
    for i in range(4):
        for j in range(4):
            x = _seg_intersection(pa[i], pa[(i + 1) % 4], pb[j], pb[(j + 1) % 4])
            if x is not None:
                cand.append(x)
    cand += [p for p in pa if _inside(pb, p)] + [p for p in pb if _inside(pa, p)]
    if len(cand) < 3:
        return 0.0
    cand = np.array(cand)
    ctr = cand.mean(0)
    order = np.argsort(np.arctan2(cand[:, 1] - ctr[1], cand[:, 0] - ctr[0]))
    poly = cand[order]
    x, y = poly[:, 0], poly[:, 1]
    return 0.5 * abs(np.dot(x, np.roll(y, -1)) - np.dot(y, np.roll(x, -1)))


def _corners_batch(b):
    """[m, 5] (x1, y1, x2, y2, yaw) -> [m, 4, 2], the same expression as `_corners`."""
    cx, cy = (b[:, 0] + b[:, 2]) / 2, (b[:, 1] + b[:, 3]) / 2
    c, s = np.cos(b[:, 4]), np.sin(b[:, 4])
    out = np.empty((b.shape[0], 4, 2))
    for k, (ix, iy) in enumerate(((0, 1), (2, 1), (2, 3), (0, 3))):
        dx, dy = b[:, ix] - cx, b[:, iy] - cy
        out[:, k, 0] = dx * c + dy * s + cx
        out[:, k, 1] = -dx * s + dy * c + cy
    return out


def _inside_batch(poly, p, eps=1e-9):
    """poly [m, 4, 2], p [m, q, 2] -> bool [m, q]: `_inside` for every (polygon, point) pair."""
    a = poly[:, None, :, :]                                   # [m, 1, 4, 2]
    b = np.roll(poly, -1, axis=1)[:, None, :, :]
    cr = (b[..., 0] - a[..., 0]) * (p[:, :, None, 1] - a[..., 1]) - (b[..., 1] - a[..., 1]) * (p[:, :, None, 0] - a[..., 0])
    live = np.abs(cr) >= eps
    pos, neg = live & (cr > 0), live & (cr <= 0)
    return ~(pos.any(-1) & neg.any(-1))


def rotated_overlap_batch(a, b):
    """`rotated_overlap(a, b[k])` for every row of b [m, 5], float64, the same recipe vectorised over the pairs: the 16 edge
    intersections + the 8 contained corners as a masked candidate list, angular sort around the centroid of the valid
    ones, shoelace.  Checked against the scalar function in tests/test_oracle_golden.py."""
    b = np.asarray(b, dtype=np.float64).reshape(-1, 5)
    m = b.shape[0]
    if m == 0:
        return np.zeros(0)
    pb = _corners_batch(b)
    pa = np.broadcast_to(_corners(a)[None], (m, 4, 2))
    cand = np.zeros((m, 24, 2))
    valid = np.zeros((m, 24), dtype=bool)
    k = 0
    for i in range(4):
        p0, p1 = pa[:, i], pa[:, (i + 1) % 4]
        d1 = p1 - p0
        for j in range(4):
            q0, q1 = pb[:, j], pb[:, (j + 1) % 4]
            d2 = q1 - q0
            den = d1[:, 0] * d2[:, 1] - d1[:, 1] * d2[:, 0]
            ok = np.abs(den) >= 1e-14
            sden = np.where(ok, den, 1.0)
            t = ((q0[:, 0] - p0[:, 0]) * d2[:, 1] - (q0[:, 1] - p0[:, 1]) * d2[:, 0]) / sden
            u = ((q0[:, 0] - p0[:, 0]) * d1[:, 1] - (q0[:, 1] - p0[:, 1]) * d1[:, 0]) / sden
            ok &= (t >= 0.0) & (t <= 1.0) & (u >= 0.0) & (u <= 1.0)
            cand[:, k] = p0 + t[:, None] * d1
            valid[:, k] = ok
            k += 1
    cand[:, 16:20], valid[:, 16:20] = pa, _inside_batch(pb, pa)
    cand[:, 20:24], valid[:, 20:24] = pb, _inside_batch(pa, pb)
    cnt = valid.sum(1)
    ctr = (cand * valid[..., None]).sum(1) / np.maximum(cnt, 1)[:, None]
    ang = np.where(valid, np.arctan2(cand[..., 1] - ctr[:, None, 1], cand[..., 0] - ctr[:, None, 0]), np.inf)
    order = np.argsort(ang, axis=1, kind="stable")
    poly = np.take_along_axis(cand, order[..., None], axis=1)
    pv = np.take_along_axis(valid, order, axis=1)
    poly = np.where(pv[..., None], poly, poly[:, :1])  # the invalid tail collapses onto the first vertex: zero-area edges
    x, y = poly[..., 0], poly[..., 1]
    area = 0.5 * np.abs((x * np.roll(y, -1, axis=1)).sum(1) - (y * np.roll(x, -1, axis=1)).sum(1))
    return np.where(cnt >= 3, area, 0.0)


def iou_bev_matrix(boxes, rotated=True):
    boxes = np.asarray(boxes, dtype=np.float64)
    n = boxes.shape[0]
    iou = np.zeros((n, n))
    area = (boxes[:, 2] - boxes[:, 0]) * (boxes[:, 3] - boxes[:, 1])
    for i in range(n):
        for j in range(i + 1, n):
            if rotated:
                ov = rotated_overlap(boxes[i], boxes[j])
            else:
                ov = max(min(boxes[i, 2], boxes[j, 2]) - max(boxes[i, 0], boxes[j, 0]), 0.0) * \
                     max(min(boxes[i, 3], boxes[j, 3]) - max(boxes[i, 1], boxes[j, 1]), 0.0)
            iou[i, j] = iou[j, i] = ov / max(area[i] + area[j] - ov, 1e-8)
    return iou


def nms_from_iou(iou, thresh):
    """Greedy NMS over boxes already in descending score order."""
    n = iou.shape[0]
    removed = np.zeros(n, dtype=bool)
    keep = []
    for i in range(n):
        if removed[i]:
            continue
        keep.append(i)
        removed[i + 1:] |= iou[i, i + 1:] > thresh
    return np.array(keep, dtype=np.int64)


def nms_lazy(boxes, thresh, rotated=True, delta=0.0, near_tol=None):
    """Greedy NMS over boxes [n,5] (x1,y1,x2,y2,yaw) already in descending score order, evaluating the float64 IoU only for
    the pairs the greedy scan actually decides (kept box vs later, not yet removed box) and only when their circumscribed
    circles touch (otherwise the overlap is exactly 0).  Same keep set as nms_from_iou(iou_bev_matrix(boxes)) at a fraction
    of the pair evaluations (one `rotated_overlap_batch` call per kept box).  Also returns the smallest |IoU - thresh| over
    the decisions taken: if it exceeds the perturbation an fp32 implementation can cause, the keep set is the only
    admissible answer.  With `near_tol` a third result lists the decisions closer than that to the threshold as rows
    (kept box, later box, |IoU - thresh|): a caller that only needs a score prefix of the keep list can ignore close calls
    among boxes behind it."""
    b = np.asarray(boxes, dtype=np.float64)
    n = b.shape[0]
    cx, cy = (b[:, 0] + b[:, 2]) / 2, (b[:, 1] + b[:, 3]) / 2
    rad = 0.5 * np.hypot(b[:, 2] - b[:, 0], b[:, 3] - b[:, 1])
    area = (b[:, 2] - b[:, 0]) * (b[:, 3] - b[:, 1])
    removed = np.zeros(n, dtype=bool)
    keep, margin, close = [], np.inf, []
    for i in range(n):
        if removed[i]:
            continue
        keep.append(i)
        later = np.nonzero(~removed[i + 1:])[0] + i + 1
        if later.size == 0:
            continue
        near = later[np.hypot(cx[later] - cx[i], cy[later] - cy[i]) <= rad[later] + rad[i]]
        if near.size == 0:
            continue
        if rotated:
            ov = rotated_overlap_batch(b[i], b[near])
        else:
            ov = np.maximum(np.minimum(b[i, 2], b[near, 2]) - np.maximum(b[i, 0], b[near, 0]), 0.0) * \
                 np.maximum(np.minimum(b[i, 3], b[near, 3]) - np.maximum(b[i, 1], b[near, 1]), 0.0)
        iou = ov / np.maximum(area[i] + area[near] - ov, 1e-8)
        d = np.abs(iou - thresh)
        margin = min(margin, float(d.min()))
        if near_tol is not None:
            for j in np.nonzero(d < near_tol)[0]:
                close.append((i, int(near[j]), float(d[j])))
        removed[near[iou > thresh]] = True
    keep = np.array(keep, dtype=np.int64)
    if near_tol is not None:
        return keep, float(margin), np.array(close, dtype=np.float64).reshape(-1, 3)
    return keep, float(margin)
