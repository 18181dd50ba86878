"""CPU-only proofs-by-enumeration of the integer cuts of round 5 (numpy / Python ints; nothing here touches the GPU or the oracle).

The HIP kernels replaced 64-bit arithmetic by 32/24-bit arithmetic and closed forms by stepped forms in three places; every one is
claimed to be EXACT.  The GPU parity tests show that on scenes; these tests show the arithmetic identities themselves over random
inputs drawn from the whole admissible domain, including its corners:

  raster_small        (m2s_devfn.h)  small triangles: edge values from the edge's own vertex, 24 x 24-bit products
  small_coverage      (m2s_devfn.h)  sign bits shifted into rows of a mask, wave-uniform trip counts
  RowWalkerS          (m2s_devfn.h)  the row walker stepped 64 rows at a time (round 6: the 32-bit walker with stride 64 took its place)
  depth sort          (m2s_sort.hip) sorting key - min over the bits of max - min
"""
import numpy as np

RNG = np.random.default_rng(0x5EED5)


def snapped_small_triangles(n, R):
    """Snapped 24.8 coordinates of triangles whose sub-pixel extent is at most 2304, anywhere in (and around) an R x R viewport."""
    cx = RNG.integers(-4096, R * 256 + 4096, n)
    cy = RNG.integers(-4096, R * 256 + 4096, n)
    ext = RNG.integers(0, 2305, (n, 1))
    X = cx[:, None] + RNG.integers(0, 2305, (n, 3)) * ext // 2304
    Y = cy[:, None] + RNG.integers(0, 2305, (n, 3)) * ext // 2304
    # corners of the domain: full extent, degenerate, axis-aligned edges
    X[: n // 50, 0] = cx[: n // 50]; X[: n // 50, 1] = cx[: n // 50] + 2304
    Y[: n // 100, 1] = Y[: n // 100, 0]
    return X.astype(np.int64), Y.astype(np.int64)


def head(X, Y, R):
    xmin, xmax, ymin, ymax = X.min(1), X.max(1), Y.min(1), Y.max(1)
    ext = np.maximum(xmax - xmin, ymax - ymin)
    x0 = np.maximum((xmin - 128 + 255) >> 8, 0)
    x1 = np.minimum((xmax - 128) >> 8, R - 1)
    y0 = np.maximum((ymin - 128 + 255) >> 8, 0)
    y1 = np.minimum((ymax - 128) >> 8, R - 1)
    return ext, x0, x1, y0, y1


def full_setup(X, Y):
    """raster_setup's 64-bit edge functions (Python-int exact through int64: coordinates are below 2^23)."""
    area2 = (X[:, 1] - X[:, 0]) * (Y[:, 2] - Y[:, 0]) - (Y[:, 1] - Y[:, 0]) * (X[:, 2] - X[:, 0])
    sgn = np.where(area2 < 0, -1, 1)
    a, b, c = [], [], []
    for i in range(3):
        ia, ib = (i + 1) % 3, (i + 2) % 3
        dy, dx = Y[:, ib] - Y[:, ia], X[:, ib] - X[:, ia]
        a.append(-dy * sgn); b.append(dx * sgn); c.append((dy * X[:, ia] - dx * Y[:, ia]) * sgn)
    a, b, c = np.stack(a, 1), np.stack(b, 1), np.stack(c, 1)
    bias = ((a > 0) | ((a == 0) & (b > 0))).astype(np.int64)
    return area2, a, b, c, bias


def test_raster_small_equals_the_64_bit_setup_and_stays_inside_24_bits():
    for R in (16, 1024, 4096):
        X, Y = snapped_small_triangles(200_000, R)
        ext, x0, x1, y0, y1 = head(X, Y, R)
        small = (ext <= 2304) & (x0 <= x1) & (y0 <= y1) & (x1 - x0 < 8) & (y1 - y0 < 8)
        assert small.sum() > (50_000 if R >= 1024 else 5_000)
        X, Y, x0, y0 = X[small], Y[small], x0[small], y0[small]
        area2, a, b, c, bias = full_setup(X, Y)
        Px0, Py0 = 256 * x0 + 128, 256 * y0 + 128
        e64 = a * Px0[:, None] + b * Py0[:, None] + c                       # what raster_setup / tri_shade_setup compute
        # raster_small: from the edge's own first vertex
        ia = np.array([1, 2, 0])
        fx, fy = Px0[:, None] - X[:, ia], Py0[:, None] - Y[:, ia]
        assert np.abs(a).max() <= 2304 and np.abs(b).max() <= 2304
        assert np.abs(fx).max() <= 2304 and np.abs(fy).max() <= 2304         # the box-origin centre lies inside the triangle's box
        p1, p2 = a * fx, b * fy
        assert np.abs(p1).max() < 2 ** 23 and np.abs(p2).max() < 2 ** 23     # each a 24 x 24-bit product (v_mul_i32_i24)
        assert np.array_equal(p1 + p2, e64)
        assert np.abs(area2).max() < 2 ** 24                                 # (float)area2 exact, like (float)(long long)
        # every edge value met inside the 8 x 8 box fits 32 bits, so the stepping in small_coverage cannot wrap
        worst = np.abs(e64) + 7 * 256 * (np.abs(a) + np.abs(b))
        assert worst.max() < 2 ** 31


def coverage_reference(e, a, b, bias, w, rows):
    m = 0
    for dy in range(rows):
        for dx in range(w):
            v = [e[i] + bias[i] - 1 + a[i] * 256 * dx + b[i] * 256 * dy for i in range(3)]
            if all(x >= 0 for x in v):
                m |= 1 << (8 * dy + dx)
    return m


def small_coverage_wave(E, A, B, BIAS, W, ROWS):
    """small_coverage for one 'wave' of lanes, operation for operation (32-bit wrap-around included)."""
    wmax, rmax = int(W.max(initial=0)), int(ROWS.max(initial=0))
    n = len(W)
    out = np.zeros(n, np.uint64)
    if not wmax:
        return out
    M32 = 0xFFFFFFFF
    e = ((E + BIAS - 1) & M32).astype(np.uint64)
    ax, by = ((A * 256) & M32).astype(np.uint64), ((B * 256) & M32).astype(np.uint64)
    wmask = ((np.uint64(1) << W.astype(np.uint64)) - np.uint64(1))
    mlo = np.zeros(n, np.uint64); mhi = np.zeros(n, np.uint64)
    for dy in range(rmax):
        r = e.copy()
        outside = np.zeros(n, np.uint64)
        for dx in range(wmax):
            t = r[:, 0] | r[:, 1] | r[:, 2]
            outside = ((outside << np.uint64(1)) | (t >> np.uint64(31))) & np.uint64(M32)      # v_alignbit(outside, t, 31)
            r = (r + ax) & np.uint64(M32)
        inv = (~outside) & np.uint64(M32)
        rev = np.array([int(format(int(v), "032b")[::-1], 2) for v in inv], np.uint64)         # v_bfrev
        row = (rev >> np.uint64(32 - wmax)) & wmask
        row = np.where(dy >= ROWS, np.uint64(0), row)
        if dy < 4:
            mlo |= row << np.uint64(8 * dy)
        else:
            mhi |= row << np.uint64(8 * dy - 32)
        e = (e + by) & np.uint64(M32)
    return mlo | (mhi << np.uint64(32))


def test_small_coverage_equals_the_per_pixel_rule():
    for R in (64, 1024):
        X, Y = snapped_small_triangles(6000, R)
        ext, x0, x1, y0, y1 = head(X, Y, R)
        small = (ext <= 2304) & (x0 <= x1) & (y0 <= y1) & (x1 - x0 < 8) & (y1 - y0 < 8)
        X, Y, x0, x1, y0, y1 = (v[small] for v in (X, Y, x0, x1, y0, y1))
        area2, a, b, c, bias = full_setup(X, Y)
        keep = area2 != 0
        a, b, c, bias, x0, x1, y0, y1 = (v[keep] for v in (a, b, c, bias, x0, x1, y0, y1))
        e = a * (256 * x0 + 128)[:, None] + b * (256 * y0 + 128)[:, None] + c
        W, ROWS = (x1 - x0 + 1), (y1 - y0 + 1)
        n = (len(W) // 64) * 64
        checked = 0
        for s in range(0, min(n, 64 * 40), 64):          # 40 waves of 64 lanes, some lanes switched off like lanes without a small triangle
            sl = slice(s, s + 64)
            Wl, Rl = W[sl].copy(), ROWS[sl].copy()
            off = RNG.random(64) < 0.15
            Wl[off] = 0; Rl[off] = 0
            got = small_coverage_wave(e[sl], a[sl], b[sl], bias[sl], Wl, Rl)
            for k in range(64):
                want = 0 if off[k] else coverage_reference([int(v) for v in e[s + k]], [int(v) for v in a[s + k]], [int(v) for v in b[s + k]],
                                                           [int(v) for v in bias[s + k]], int(W[s + k]), int(ROWS[s + k]))
                assert int(got[k]) == want, (s, k)
                checked += 1
        assert checked >= 64 * 20


def floordiv(n, d):
    return n // d            # Python's // is the mathematical floor for d > 0


def row_span_reference(a, b, c, bias, x0, x1, y):
    Py = 256 * y + 128
    lo, hi = x0, x1
    for i in range(3):
        alpha = 256 * a[i]
        beta = 128 * a[i] + b[i] * Py + c[i] + bias[i]
        if alpha > 0:
            lo = max(lo, floordiv(alpha - beta, alpha))
        elif alpha < 0:
            hi = min(hi, floordiv(beta - 1, -alpha))
        elif beta < 1:
            hi = lo - 1
    return (0, -1) if hi < lo else (lo, hi)


def test_strided_row_walker_equals_the_closed_form():
    """Quotient and remainder of every edge bound stepped by S rows at a time (round 5: RowWalkerS; since round 6 the 32-bit walker with
    stride 64, m2s_devfn.h row_walker32_init — tests/test_round6_math.py has its transcription; the identity is the same)."""
    S = 64
    for _ in range(300):
        R = int(RNG.choice([256, 1024, 4096]))
        X = [int(v) for v in RNG.integers(-2000, R * 256 + 2000, 3)]
        Y = [int(v) for v in RNG.integers(-2000, R * 256 + 2000, 3)]
        area2 = (X[1] - X[0]) * (Y[2] - Y[0]) - (Y[1] - Y[0]) * (X[2] - X[0])
        if area2 == 0:
            continue
        sgn = -1 if area2 < 0 else 1
        a, b, c, bias = [], [], [], []
        for i in range(3):
            ia, ib = (i + 1) % 3, (i + 2) % 3
            dy, dx = Y[ib] - Y[ia], X[ib] - X[ia]
            a.append(-dy * sgn); b.append(dx * sgn); c.append((dy * X[ia] - dx * Y[ia]) * sgn)
            bias.append(1 if (a[i] > 0 or (a[i] == 0 and b[i] > 0)) else 0)
        x0, x1 = max((min(X) - 128 + 255) >> 8, 0), min((max(X) - 128) >> 8, R - 1)
        y0, y1 = max((min(Y) - 128 + 255) >> 8, 0), min((max(Y) - 128) >> 8, R - 1)
        if x0 > x1 or y0 > y1:
            continue
        for lane in (0, 1, 17, 63):
            y = y0 + lane
            if y > y1:
                continue
            # init at row y, stride S
            Py = 256 * y + 128
            q, r, sq, sr, D, beta, bstep, lower = [0] * 3, [0] * 3, [0] * 3, [0] * 3, [0] * 3, [0] * 3, [0] * 3, [False] * 3
            for i in range(3):
                alpha = 256 * a[i]
                be = 128 * a[i] + b[i] * Py + c[i] + bias[i]
                bs = 256 * b[i] * S
                beta[i], bstep[i] = be, bs
                if alpha > 0:
                    lower[i] = True; D[i] = alpha
                    n = alpha - be; q[i] = floordiv(n, alpha); r[i] = n - q[i] * alpha
                    sq[i] = floordiv(-bs, alpha); sr[i] = -bs - sq[i] * alpha
                elif alpha < 0:
                    d = -alpha; D[i] = d
                    n = be - 1; q[i] = floordiv(n, d); r[i] = n - q[i] * d
                    sq[i] = floordiv(bs, d); sr[i] = bs - sq[i] * d
                assert abs(sq[i]) < 2 ** 31 and 0 <= sr[i] < max(D[i], 1) and D[i] < 2 ** 32
            while y <= y1:
                lo, hi = x0, x1
                for i in range(3):
                    if D[i]:
                        if lower[i]: lo = max(lo, q[i])
                        else: hi = min(hi, q[i])
                        r[i] += sr[i]; q[i] += sq[i]
                        if r[i] >= D[i]:
                            r[i] -= D[i]; q[i] += 1
                    else:
                        if beta[i] < 1: hi = lo - 1
                        beta[i] += bstep[i]
                got = (0, -1) if hi < lo else (lo, hi)
                assert got == row_span_reference(a, b, c, bias, x0, x1, y), (X, Y, y)
                y += S


def test_sorting_key_minus_min_over_the_bits_that_differ_is_the_same_sort():
    """m2s_sort.hip: keys of a bounded scene differ in their low bits only; a stable LSD radix sort of (key - min) over
    bits(max - min) bits gives the permutation of a stable sort of the keys."""
    for lo, hi, saves in ((-8.5, -4.75, True), (0.25, 3.0, False), (-1e-3, -1e-6, False), (5.0, 5.0, True), (100.0, 131.0, True)):
        z = RNG.uniform(lo, hi, 20000).astype(np.float32)
        z[::7] = z[0]                                                  # ties: stability matters
        key = z.view(np.uint32).astype(np.int64)
        kmin, kmax = key.min(), key.max()
        bits = max(1, int(kmax - kmin).bit_length())
        assert ((bits + 7) // 8 < 4) == saves                          # C5's view (z in [-8.5, -4.75]): 23 bits, three passes instead of four
        red = key - kmin
        order = np.arange(len(red))
        for shift in range(0, bits, 8):                                # LSD radix, 8-bit digits, stable
            digit = (red[order] >> shift) & (min(255, (1 << (bits - shift)) - 1) if bits - shift < 8 else 255)
            order = order[np.argsort(digit, kind="stable")]
        assert np.array_equal(order, np.argsort(key, kind="stable"))
    mixed = np.array([-1.0, 2.0, -0.5, 0.25], np.float32).view(np.uint32).astype(np.int64)
    assert int(mixed.max() - mixed.min()).bit_length() > 24            # a range that spans the sign keeps all 32 bits (four passes)


def test_row_starts_plus_fill_reproduces_the_pixel_list():
    """k_emit2's expansion (m2s_emit2.hip): every covered row writes ONE entry — at its first record inside [pos, bend), with the x of that
    record — and sets the record's bit; the fill gives every record the nearest start at or before it plus the distance, 64 records at a
    time with the last lane's entry carried into the next block.  Against the straightforward pixel-by-pixel list, for slices that begin
    and end inside rows, rows longer than a block, and single-record rows."""
    for trial in range(300):
        wbase = int(RNG.integers(0, 4)) * 512
        # rows of the triangles that overlap the slice: (slot, y, xa, length), in canonical order; the first may start before the slice
        k = wbase - int(RNG.integers(0, 300))
        rows = []
        while k < wbase + 512 + 100:
            ln = int(RNG.choice([1, 1, 2, 3, 7, 30, 64, 65, 200, 700]))
            rows.append((int(RNG.integers(0, 64)), int(RNG.integers(0, 4096)), int(RNG.integers(0, 4096 - ln)), ln, k))
            k += ln
        pos = wbase + int(RNG.integers(0, 200)) if trial % 3 else wbase
        bend = min(wbase + 512, pos + int(RNG.integers(1, 513)))
        want = {}
        for slot, y, xa, ln, k0 in rows:
            for j in range(ln):
                if pos <= k0 + j < bend:
                    want[k0 + j] = (slot << 24) | (y << 12) | (xa + j)
        assert sorted(want) == list(range(pos, bend))
        # step 1: row starts (mark_row)
        entries = [0xDEADBEEF] * 512
        mask = 0
        for slot, y, xa, ln, k0 in rows:
            if ln and k0 < bend and k0 + ln > pos:
                st = max(k0, pos)
                i = st - wbase
                entries[i] = (slot << 24) | (y << 12) | (xa + (st - k0))
                mask |= 1 << i
        # step 2: fill, block by block
        i0, i1 = pos - wbase, bend - wbase
        carry = 0
        for blk in range(i0 >> 6, ((i1 - 1) >> 6) + 1):
            m = (mask >> (64 * blk)) & ((1 << 64) - 1)
            e_blk = []
            for lane in range(64):
                i = blk * 64 + lane
                me = m & ((1 << (lane + 1)) - 1)
                if me:
                    src = me.bit_length() - 1
                    e = (entries[blk * 64 + src] + (lane - src)) & 0xFFFFFFFF
                else:
                    e = (carry + lane + 1) & 0xFFFFFFFF
                e_blk.append(e)
            for lane in range(64):
                i = blk * 64 + lane
                if i0 <= i < i1:
                    entries[i] = e_blk[lane]
            carry = e_blk[63]
        for kk in range(pos, bend):
            assert entries[kk - wbase] == want[kk], (trial, kk)
