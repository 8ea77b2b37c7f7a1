"""A slow, direct numpy statement of the 3-way semi-global matcher the oracle restates (plain integer arithmetic, no
packing, no streaming): pixel cost -> block sum -> three paths -> winner -> right-view check -> median.  Written from
the algorithm's description for small images; used only to cross-check oracle/sgbm.cpp."""
import numpy as np
from scipy import ndimage

INVALID = -16


def prefiltered(img, ftzero=15):
    """(gradient channel, raw channel), both with the table value in the first and the last column."""
    I = img.astype(np.int64)
    h, w = I.shape
    up = np.vstack([I[:1], I[:-1]])
    dn = np.vstack([I[1:], I[-1:]])
    g = np.full((h, w), ftzero, dtype=np.int64)
    g[:, 1:-1] = np.clip(2 * (I[:, 2:] - I[:, :-2]) + (up[:, 2:] - up[:, :-2]) + (dn[:, 2:] - dn[:, :-2]), -ftzero, ftzero) + ftzero
    r = I.copy()
    r[:, 0] = ftzero
    r[:, -1] = ftzero
    return g, r


def half_pixel_range(c):
    """per pixel: value, min and max over {value, mean with left neighbour, mean with right neighbour}"""
    left = np.hstack([c[:, :1], c[:, :-1]])
    right = np.hstack([c[:, 1:], c[:, -1:]])
    a, b = (c + left) // 2, (c + right) // 2
    a[:, 0] = c[:, 0]
    b[:, -1] = c[:, -1]
    return c, np.minimum(np.minimum(a, b), c), np.maximum(np.maximum(a, b), c)


def pixel_cost(left, right, D):
    """pix[y, xc, d] for cost columns xc = x - D, x in [D, W)"""
    h, w = left.shape
    out = np.zeros((h, w - D, D), dtype=np.int64)
    for ch, shift in ((0, 0), (1, 2)):
        u, u0, u1 = half_pixel_range(prefiltered(left)[ch])
        v, v0, v1 = half_pixel_range(prefiltered(right)[ch])
        for d in range(D):
            xs = np.arange(D, w)
            U, U0, U1 = u[:, xs], u0[:, xs], u1[:, xs]
            V, V0, V1 = v[:, xs - d], v0[:, xs - d], v1[:, xs - d]
            c0 = np.maximum(0, np.maximum(U - V1, V0 - U))
            c1 = np.maximum(0, np.maximum(V - U1, U0 - V))
            out[:, :, d] += np.minimum(c0, c1) >> shift
    return out


def path(C_line, P1, P2):
    """one 1-D path over C_line[n, D]: returns L[n, D]"""
    n, D = C_line.shape
    L = np.zeros((n, D), dtype=np.int64)
    prev = np.zeros(D, dtype=np.int64)
    m = 0
    big = 1 << 30
    for i in range(n):
        lo = np.concatenate([[big], prev[:-1]]) + P1
        hi = np.concatenate([prev[1:], [big]]) + P1
        L[i] = C_line[i] + np.minimum(np.minimum(prev, lo), np.minimum(hi, m + P2)) - (m + P2)
        prev = L[i]
        m = prev.min()
    return L


def sgm3way(left, right, D=32, block=11, P1=968, P2=3872, disp12=1):
    h, w = left.shape
    w1 = w - D
    r = block // 2
    pix = pixel_cost(left, right, D)
    stripe_sz = -(-h // 4)
    overlap = (r + 1) + int(np.ceil(0.1 * stripe_sz))
    disp = np.full((h, w), INVALID, dtype=np.int64)
    for s in range(4):
        y0 = max(min(s * stripe_sz - overlap, h), 0)
        y1 = min((s + 1) * stripe_sz, h)
        if y1 <= y0:
            continue
        rows = np.arange(y0, y1)
        # block sum with the row index clamped to [y0, h-1] and the cost column clamped to [0, w1-1]
        C = np.zeros((len(rows), w1, D), dtype=np.int64)
        for dy in range(-r, r + 1):
            yy = np.clip(rows + dy, y0, h - 1)
            for dx in range(-r, r + 1):
                xx = np.clip(np.arange(w1) + dx, 0, w1 - 1)
                C += pix[yy][:, xx]
        Ltop = np.stack([path(C[:, x], P1, P2) for x in range(w1)], axis=1)
        for i, y in enumerate(rows):
            if y < s * stripe_sz:
                continue                      # warm-up row
            Llr = path(C[i], P1, P2)
            Lrl = path(C[i][::-1], P1, P2)[::-1]
            S = Llr + Ltop[i] + Lrl
            best = S.argmin(axis=1)           # lowest d on ties
            minS = S[np.arange(w1), best]
            d16 = best * 16
            for xc in range(w1):
                b = best[xc]
                if 0 < b < D - 1:
                    den = max(S[xc, b - 1] + S[xc, b + 1] - 2 * S[xc, b], 1)
                    num = (S[xc, b - 1] - S[xc, b + 1]) * 16 + den
                    d16[xc] += int(num / (den * 2))          # C division truncates toward zero
            # right view: smallest cost wins, ties to the largest x
            d2 = np.full(w, INVALID, dtype=np.int64)
            c2 = np.full(w, 1 << 30, dtype=np.int64)
            for xc in range(w1 - 1, -1, -1):
                x2 = xc + D - best[xc]
                if c2[x2] > minS[xc]:
                    c2[x2] = minS[xc]; d2[x2] = best[xc]
            row = np.full(w, INVALID, dtype=np.int64)
            for xc in range(w1):
                x = xc + D
                dv = d16[xc]
                _d, d_ = dv >> 4, (dv + 15) >> 4
                _x, x_ = x - _d, x - d_
                bad1 = 0 <= _x < w and d2[_x] >= 0 and abs(d2[_x] - _d) > disp12
                bad2 = 0 <= x_ < w and d2[x_] >= 0 and abs(d2[x_] - d_) > disp12
                row[x] = INVALID if (bad1 and bad2) else dv
            disp[y] = row
    pre = disp.astype(np.int16)
    return ndimage.median_filter(pre, size=3, mode="nearest"), pre
